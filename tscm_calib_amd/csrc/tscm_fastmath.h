// tscm_fastmath.h -- fp64 reciprocal / reciprocal square root from the gfx950 hardware seeds.
#pragma once

#include <hip/hip_runtime.h>

namespace tscm {

// 1/sqrt(x) and 1/x to full fp64 accuracy from the hardware seeds (v_rsq_f64 / v_rcp_f64, measured
// relative error 5e-8 on gfx950: tools/rsq_precision.hip) plus ONE third-order correction step
// (error ~ e^3): a shorter dependent chain than two Newton steps and ~3x fewer instructions than the
// IEEE sqrt + divide expansions.
__device__ __forceinline__ double fast_rsqrt(double x)
{
    const double r = __builtin_amdgcn_rsq(x);
    const double e = __builtin_fma(-x * r, r, 1.0);                 // 1 - x r^2
    return __builtin_fma(r * e, __builtin_fma(0.375, e, 0.5), r);   // r (1 + e/2 + 3 e^2 / 8)
}
__device__ __forceinline__ double fast_rcp(double x)
{
    const double r = __builtin_amdgcn_rcp(x);
    const double e = __builtin_fma(-x, r, 1.0);                     // 1 - x r
    return __builtin_fma(r * e, 1.0 + e, r);                        // r (1 + e + e^2)
}
// sqrt(x) and its reciprocal
__device__ __forceinline__ void sqrt_and_inverse(double x, double &s, double &is)
{
    is = fast_rsqrt(x);        // ~1 ulp
    s = x * is;                // ~2 ulp: a refinement step for s would cost 3 more instructions on the serial chain of the
                               // triple-sphere projection (d1 -> d2 -> d3) and buys nothing at the 1e-6 parity bar
}

}  // namespace tscm

// accuracy of the gfx950 v_rsq_f64 / v_rcp_f64 seeds and of 1 / 2 Newton steps (decides how many steps tscm uses)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const double *x, double *o, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = x[i];
    double r0 = __builtin_amdgcn_rsq(v);
    double r1 = r0 * __builtin_fma(-0.5 * v * r0, r0, 1.5);
    double r2 = r1 * __builtin_fma(-0.5 * v * r1, r1, 1.5);
    { const double e = __builtin_fma(-v * r0, r0, 1.0); r1 = __builtin_fma(r0 * e, __builtin_fma(0.375, e, 0.5), r0); }   // third-order step (used)
    double c0 = __builtin_amdgcn_rcp(v);
    double c1 = __builtin_fma(__builtin_fma(-v, c0, 1.0), c0, c0);
    double c2 = __builtin_fma(__builtin_fma(-v, c1, 1.0), c1, c1);
    { const double e = __builtin_fma(-v, c0, 1.0); c1 = __builtin_fma(c0 * e, 1.0 + e, c0); }                              // third-order step (used)
    o[6 * i] = r0; o[6 * i + 1] = r1; o[6 * i + 2] = r2; o[6 * i + 3] = c0; o[6 * i + 4] = c1; o[6 * i + 5] = c2;
}
int main()
{
    const int n = 1 << 20;
    double *hx = new double[n], *ho = new double[6 * n];
    unsigned long long s = 88172645463325252ULL;
    for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; double u = (s >> 11) * (1.0 / 9007199254740992.0); hx[i] = std::exp(40.0 * u - 20.0); }
    double *dx, *dout;
    hipMalloc(&dx, n * 8); hipMalloc(&dout, 6 * n * 8);
    hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
    hipMemcpy(ho, dout, 6 * n * 8, hipMemcpyDeviceToHost);
    double e[6] = { 0, 0, 0, 0, 0, 0 };
    for (int i = 0; i < n; ++i) {
        const long double v = hx[i];
        const long double rs = 1.0L / sqrtl(v), rc = 1.0L / v;
        for (int k2 = 0; k2 < 3; ++k2) { double d = (double)fabsl((ho[6 * i + k2] - rs) / rs); if (d > e[k2]) e[k2] = d; }
        for (int k2 = 0; k2 < 3; ++k2) { double d = (double)fabsl((ho[6 * i + 3 + k2] - rc) / rc); if (d > e[3 + k2]) e[3 + k2] = d; }
    }
    printf("max rel err  rsq: seed %.3e  cubic step %.3e  2 NR %.3e | rcp: seed %.3e  cubic step %.3e  2 NR %.3e\n", e[0], e[1], e[2], e[3], e[4], e[5]);
    return 0;
}

// Does a SIMD of gfx950 run VALU instructions of one wave WHILE the matrix core works on another wave's MFMA?
// 512-thread workgroups = 8 waves = two per SIMD (waves w and w + 4 share one): wave role A runs a chain of MFMAs,
// role B a chain of VALU FMAs.  Times: A alone, B alone, both.  both ~ max(A, B): the pipes overlap; ~ A + B: they do not.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/ubench_overlap tools/ubench_overlap.hip && /tmp/ubench_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MF, int VA>
__global__ __launch_bounds__(512) void k(int iters, int run_a, int run_b, float *out)
{
    const int wave = threadIdx.x >> 6;
    const bool role_a = ((wave >> 2) & 1) == 0;
    float r = 0.f;
    if (role_a) {
        if (!run_a) return;
        if constexpr (MF == 0) {            // v_mfma_f32_16x16x4_f32
            f4 a0 = { 0, 0, 0, 0 }, a1 = a0, a2 = a0, a3 = a0;
            const float x = threadIdx.x * 1e-3f, y = 1.0f;
            for (int i = 0; i < iters; ++i) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
            }
            r = a0[0] + a1[1] + a2[2] + a3[3];
        } else if constexpr (MF == 1) {     // v_mfma_f64_4x4x4_4b_f64
            double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
            const double x = threadIdx.x * 1e-3, y = 1.0;
            for (int i = 0; i < iters; ++i) {
                a0 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, a3, 0, 0, 0);
            }
            r = (float)(a0 + a1 + a2 + a3);
        } else {                            // v_mfma_f64_16x16x4_f64
            d4 a0 = { 0, 0, 0, 0 }, a1 = a0;
            const double x = threadIdx.x * 1e-3, y = 1.0;
            for (int i = 0; i < iters; ++i) {
                a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
            }
            r = (float)(a0[0] + a1[1]);
        }
    } else {
        if (!run_b) return;
        if constexpr (VA == 0) {            // v_fma_f32
            float v[8];
            for (int q = 0; q < 8; ++q) v[q] = threadIdx.x * 1e-3f + q;
            const float m = 0.999f, c = 1e-3f;
            for (int i = 0; i < iters; ++i)
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = __builtin_fmaf(v[q], m, c);
            for (int q = 0; q < 8; ++q) r += v[q];
        } else if constexpr (VA == 1) {     // v_pk_fma_f32
            f2 v[8];
            for (int q = 0; q < 8; ++q) v[q] = f2{ threadIdx.x * 1e-3f + q, 1.f };
            const f2 m = { 0.999f, 0.998f }, c = { 1e-3f, 2e-3f };
            for (int i = 0; i < iters; ++i)
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = __builtin_elementwise_fma(v[q], m, c);
            for (int q = 0; q < 8; ++q) r += v[q].x + v[q].y;
        } else {                            // v_fma_f64
            double v[8];
            for (int q = 0; q < 8; ++q) v[q] = threadIdx.x * 1e-3 + q;
            const double m = 0.999, c = 1e-3;
            for (int i = 0; i < iters; ++i)
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = __builtin_fma(v[q], m, c);
            for (int q = 0; q < 8; ++q) r += (float)v[q];
        }
    }
    if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int MF, int VA>
static void run(const char *name, float *out)
{
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms[3];
    for (int mode = 0; mode < 3; ++mode) {
        const int a = mode != 1, b = mode != 0;
        k<MF, VA><<<256, 512>>>(iters, a, b, out);
        hipEventRecord(e0);
        k<MF, VA><<<256, 512>>>(iters, a, b, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms[mode], e0, e1);
    }
    printf("%-44s MFMA alone %7.3f ms  VALU alone %7.3f ms  both %7.3f ms  (sum %7.3f, max %7.3f)\n", name, ms[0], ms[1], ms[2], ms[0] + ms[1], ms[0] > ms[1] ? ms[0] : ms[1]);
}

int main()
{
    float *out; hipMalloc(&out, 4096);
    run<0, 0>("mfma_f32_16x16x4 (4/iter) + v_fma_f32 (8/iter)", out);
    run<0, 1>("mfma_f32_16x16x4 (4/iter) + v_pk_fma_f32 (8/iter)", out);
    run<0, 2>("mfma_f32_16x16x4 (4/iter) + v_fma_f64 (8/iter)", out);
    run<1, 2>("mfma_f64_4x4x4_4b (4/iter) + v_fma_f64 (8/iter)", out);
    run<1, 0>("mfma_f64_4x4x4_4b (4/iter) + v_fma_f32 (8/iter)", out);
    run<2, 2>("mfma_f64_16x16x4 (2/iter) + v_fma_f64 (8/iter)", out);
    run<2, 0>("mfma_f64_16x16x4 (2/iter) + v_fma_f32 (8/iter)", out);
    return 0;
}

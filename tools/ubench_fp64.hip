// ubench_fp64.hip -- measured fp64 ceilings of the MI355X for the TSCM roofline:
//   v_fma_f64 (VALU) and v_mfma_f64_16x16x4_f64 (matrix core) throughput, all CUs busy.
// build: hipcc --offload-arch=gfx950 -O3 ubench_fp64.hip -o ubench_fp64 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_fma(double *out, int iters, double seed)
{
    double a[NACC];
    for (int i = 0; i < NACC; ++i) a[i] = seed + i + threadIdx.x;
    const double b = 1.0000001, c = 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) a[i] = __builtin_fma(a[i], b, c);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(double *out, int iters, double seed)
{
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{ seed, 0, 0, 0 };
    const double a = 1.0 + threadIdx.x * 1e-6, b = 1.0 - threadIdx.x * 1e-6;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_sqrt_div(double *out, int iters, double seed)
{
    double a = seed + threadIdx.x, b = seed * 3 + threadIdx.x, c = 2.0 + threadIdx.x * 1e-3, d = 5.0 + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        a = sqrt(a + 1.5); b = sqrt(b + 2.5); c = 1.0 / (c + 0.25); d = 1.0 / (d + 0.5);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}

// both pipes in one wave: NF independent FMA chains interleaved with NM MFMA chains
template <int NF, int NM>
__global__ __launch_bounds__(256) void k_mixed(double *out, int iters, double seed)
{
    double a[NF > 0 ? NF : 1];
    d4 acc[NM > 0 ? NM : 1];
    for (int i = 0; i < NF; ++i) a[i] = seed + i + threadIdx.x;
    for (int i = 0; i < NM; ++i) acc[i] = d4{ seed, 0, 0, 0 };
    const double b = 1.0000001, c = 1e-9;
    const double ma = 1.0 + threadIdx.x * 1e-6;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NM; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, ma, acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NF; ++i) a[i] = __builtin_fma(a[i], b, c);
    }
    double s = 0;
    for (int i = 0; i < NF; ++i) s += a[i];
    for (int i = 0; i < NM; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// half of the waves of a block only issue MFMA, the other half only FMA
__global__ __launch_bounds__(512) void k_split(double *out, int iters, double seed, int fma_per_mfma)
{
    const int wave = threadIdx.x >> 6;
    double s = 0;
    if (wave & 1) {
        d4 acc[4];
        for (int i = 0; i < 4; ++i) acc[i] = d4{ seed, 0, 0, 0 };
        const double ma = 1.0 + threadIdx.x * 1e-6;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, ma, acc[i], 0, 0, 0);
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        double a[8];
        for (int i = 0; i < 8; ++i) a[i] = seed + i + threadIdx.x;
        const double b = 1.0000001, c = 1e-9;
        for (int it = 0; it < iters * fma_per_mfma / 2; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = __builtin_fma(a[i], b, c);
        for (int i = 0; i < 8; ++i) s += a[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
static double time_ms(F launch, int reps)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %.0f MHz\n", prop.name, cus, prop.clockRate / 1e3);
    double *out;
    hipMalloc(&out, sizeof(double) * 256 * 8 * cus);
    const int iters = 20000;
    for (int bpc : { 1, 2, 4 }) {
        const int grid = cus * bpc;
        double ms = time_ms([&] { hipLaunchKernelGGL(k_fma<8>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0); }, 5);
        double fl = 2.0 * 8 * iters * 256.0 * grid;
        printf("v_fma_f64      %d blocks/CU (x4 waves): %8.3f ms  %7.2f TFLOP/s\n", bpc, ms, fl / ms / 1e9);
    }
    for (int bpc : { 1, 2 }) {
        const int grid = cus * bpc;
        double ms = time_ms([&] { hipLaunchKernelGGL(k_mfma<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0); }, 5);
        double fl = 2048.0 * 4 * iters * 4.0 * grid;   // 2*16*16*4 flop per wave-instruction, 4 waves per block
        printf("mfma_f64_16x16x4 %d blocks/CU (x4 waves): %8.3f ms  %7.2f TFLOP/s  (%.1f cycles/instr/SIMD at %.0f MHz)\n", bpc, ms,
               fl / ms / 1e9, ms * 1e-3 * prop.clockRate * 1e3 / (4.0 * iters * bpc), prop.clockRate / 1e3);
    }
    for (int bpc : { 4 }) {
        const int grid = cus * bpc;
        double ms = time_ms([&] { hipLaunchKernelGGL(k_mfma<8>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0); }, 3);
        double fl = 2048.0 * 8 * iters * 4.0 * grid;
        printf("mfma_f64_16x16x4 x8 acc, %d blocks/CU: %8.3f ms  %7.2f TFLOP/s\n", bpc, ms, fl / ms / 1e9);
    }
    {
        const int grid = cus * 2;
        double m1 = time_ms([&] { hipLaunchKernelGGL((k_mixed<0, 4>), dim3(grid), dim3(256), 0, 0, out, iters, 1.0); }, 3);
        double m2 = time_ms([&] { hipLaunchKernelGGL((k_mixed<16, 0>), dim3(grid), dim3(256), 0, 0, out, iters, 1.0); }, 3);
        double m3 = time_ms([&] { hipLaunchKernelGGL((k_mixed<16, 4>), dim3(grid), dim3(256), 0, 0, out, iters, 1.0); }, 3);
        printf("same wave, 2 blocks/CU: 4 mfma only %.3f ms | 16 fma only %.3f ms | both %.3f ms  (sum %.3f, max %.3f)\n", m1, m2, m3, m1 + m2, m1 > m2 ? m1 : m2);
        double s1 = time_ms([&] { hipLaunchKernelGGL(k_split, dim3(cus), dim3(512), 0, 0, out, iters, 1.0, 0); }, 3);
        double s2 = time_ms([&] { hipLaunchKernelGGL(k_split, dim3(cus), dim3(512), 0, 0, out, iters, 1.0, 16); }, 3);
        printf("split waves, 1 block(8 waves)/CU: mfma waves alone %.3f ms | with fma waves (16 fma per 4 mfma) %.3f ms\n", s1, s2);
    }
    {
        const int grid = cus * 4;
        double ms = time_ms([&] { hipLaunchKernelGGL(k_sqrt_div, dim3(grid), dim3(256), 0, 0, out, 2000, 1.0); }, 5);
        printf("2 sqrt + 2 div f64 per iter, 4 blocks/CU: %8.3f ms -> %.1f SIMD-cycles per (sqrt+div) pair per wave\n", ms,
               ms * 1e-3 * prop.clockRate * 1e3 / (2000.0 * 2 * 4));
    }
    return 0;
}

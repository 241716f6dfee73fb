// ubench_mfma64.hip -- what the fp64 matrix instructions of gfx950 cost a SIMD, in shader clocks:
//   v_mfma_f64_16x16x4_f64 (2048 flop) against v_mfma_f64_4x4x4_4b_f64 (4 blocks of 4x4x4: 512 flop), independent and
//   dependent chains, 1 and 4 waves per SIMD; and which VALU work overlaps with them (fp64 FMA, 32-bit integer).
// One workgroup per CU-slot, s_memtime around the loop of every wave, the slowest wave of the launch is reported.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma64.hip -o /tmp/ubench_mfma64 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int KIND, int NACC, int NFMA, int NINT>
__global__ __launch_bounds__(256) void k_bench(long long *cyc, double *out, int iters, double seed)
{
    d4 acc4[NACC > 0 ? NACC : 1];
    double acc1[NACC > 0 ? NACC : 1];
    double f[NFMA > 0 ? NFMA : 1];
    int q[NINT > 0 ? NINT : 1];
    for (int i = 0; i < NACC; ++i) { acc4[i] = d4{ seed, 0, 0, 0 }; acc1[i] = seed; }
    for (int i = 0; i < NFMA; ++i) f[i] = seed + i + threadIdx.x;
    for (int i = 0; i < NINT; ++i) q[i] = (int)threadIdx.x + i;
    const double a = 1.0 + threadIdx.x * 1e-6, b = 1.0 - threadIdx.x * 1e-6, fb = 1.0000001, fc = 1e-9;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (KIND == 0) acc4[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc4[i], 0, 0, 0);
            else acc1[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc1[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NFMA; ++i) f[i] = __builtin_fma(f[i], fb, fc);
#pragma unroll
        for (int i = 0; i < NINT; ++i) q[i] = (q[i] ^ (q[i] >> 3)) + it;
    }
    const long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc4[i][0] + acc4[i][1] + acc4[i][2] + acc4[i][3] + acc1[i];
    for (int i = 0; i < NFMA; ++i) s += f[i];
    for (int i = 0; i < NINT; ++i) s += q[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND, int NACC, int NFMA, int NINT>
static void run(const char *name, int cus, int wgs_per_cu, long long *d_cyc, double *d_out)
{
    const int iters = 2000, n = cus * wgs_per_cu;
    hipLaunchKernelGGL((k_bench<KIND, NACC, NFMA, NINT>), dim3(n), dim3(256), 0, 0, d_cyc, d_out, iters, 1.0);
    hipDeviceSynchronize();
    std::vector<long long> c(4 * (size_t)n);
    hipMemcpy(c.data(), d_cyc, sizeof(long long) * c.size(), hipMemcpyDeviceToHost);
    std::sort(c.begin(), c.end());
    const double med = (double)c[c.size() / 2] / iters, mx = (double)c.back() / iters;
    const int per_it = NACC + NFMA + NINT;
    printf("%-58s waves/SIMD %d: %7.1f clocks per iteration (max %7.1f)  = %6.2f per instruction", name, wgs_per_cu, med, mx, med / per_it);
    if (NACC) printf("  | per MFMA if alone %6.1f", med / NACC);
    printf("\n");
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs\n(s_memtime ticks: 100 MHz constant clock or shader clock depending on the part -- compare rows, not absolute values)\n", prop.name, cus);
    long long *d_cyc; double *d_out;
    hipMalloc(&d_cyc, sizeof(long long) * 4 * cus * 4);
    hipMalloc(&d_out, sizeof(double) * 256 * cus * 4);
    for (int w : { 1, 4 }) {
        run<0, 4, 0, 0>("mfma_f64_16x16x4, 4 independent accumulators", cus, w, d_cyc, d_out);
        run<0, 1, 0, 0>("mfma_f64_16x16x4, 1 dependent chain", cus, w, d_cyc, d_out);
        run<1, 8, 0, 0>("mfma_f64_4x4x4_4b, 8 independent accumulators", cus, w, d_cyc, d_out);
        run<1, 1, 0, 0>("mfma_f64_4x4x4_4b, 1 dependent chain", cus, w, d_cyc, d_out);
        run<0, 0, 8, 0>("v_fma_f64, 8 independent chains", cus, w, d_cyc, d_out);
        run<0, 0, 0, 8>("32-bit integer VALU (3 ops per element), 8 chains", cus, w, d_cyc, d_out);
        run<0, 4, 16, 0>("mfma 16x16x4 x4 + 16 v_fma_f64", cus, w, d_cyc, d_out);
        run<0, 4, 0, 16>("mfma 16x16x4 x4 + 16 x 3 integer ops", cus, w, d_cyc, d_out);
        run<1, 8, 16, 0>("mfma 4x4x4 x8 + 16 v_fma_f64", cus, w, d_cyc, d_out);
        run<1, 8, 0, 16>("mfma 4x4x4 x8 + 16 x 3 integer ops", cus, w, d_cyc, d_out);
    }
    return 0;
}

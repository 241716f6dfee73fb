// ubench_issue.hip -- what ONE workgroup gets out of a CU in fp64: issue interval of independent v_fma_f64 per wave
// (1, 2, 4 waves per SIMD), latency of a dependent chain, ds_read_b128 throughput, s_barrier cost.  The single-workgroup
// kernels of the solver (k_solve_reduced) are priced against these numbers.
// build: hipcc --offload-arch=gfx950 -O3 ubench_issue.hip -o ubench_issue ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ long long rt() { return wall_clock64(); }

template <int NCH>
__global__ __launch_bounds__(1024) void k_fma_issue(double *out, long long *ticks, int iters, double seed)
{
    double a[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) a[i] = seed + i + threadIdx.x;
    const double b = 1.0000001, c = 1e-9;
    __syncthreads();
    const long long t0 = rt(), c0 = clock64();
    for (int it = 0; it < iters; it += 16) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NCH; ++i) a[i] = __builtin_fma(a[i], b, c);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NCH; ++i) s += a[i];
    asm volatile("" : "+v"(s));
    __syncthreads();
    const long long t1 = rt();
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) { ticks[0] = t1 - t0; ticks[1] = clock64() - c0; }
}

__global__ __launch_bounds__(1024) void k_lds_read(double *out, long long *ticks, int iters)
{
    __shared__ __attribute__((aligned(16))) double buf[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) buf[i] = i;
    __syncthreads();
    typedef double d2 __attribute__((ext_vector_type(2)));
    const d2 *p = reinterpret_cast<const d2 *>(buf) + (threadIdx.x & 63) * 9;       // conflict-free tile stride (18 doubles)
    d2 acc = { 0.0, 0.0 };
    const long long t0 = rt();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int q = 0; q < 8; ++q) { d2 v = p[q + ((it + u) & 1)]; acc += v; }
    }
    asm volatile("" : "+v"(acc));
    __syncthreads();
    const long long t1 = rt();
    out[threadIdx.x] = acc[0] + acc[1];
    if (threadIdx.x == 0) *ticks = t1 - t0;
}

__global__ __launch_bounds__(1024) void k_barrier(long long *ticks, int iters)
{
    __syncthreads();
    const long long t0 = rt();
    for (int it = 0; it < iters; ++it) __syncthreads();
    const long long t1 = rt();
    if (threadIdx.x == 0) *ticks = t1 - t0;
}

// first-touch cost of global loads in a single-workgroup kernel that follows a producer kernel on the same stream
__global__ void k_produce(double *b1, double *b2, double *b3, double *b4, double v)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    b1[i] = v + i; b2[i] = v - i; b3[i] = v * 2; b4[i] = v * 3;
}
__global__ __launch_bounds__(64) void k_first_load(const double *b1, const double *b2, const double *b3, const double *b4, double *out, long long *ticks)
{
    const long long t0 = rt();
    double v1 = b1[threadIdx.x];
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(v1));
    const long long t1 = rt();
    double v2 = b2[threadIdx.x];
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(v2));
    const long long t2 = rt();
    double v3 = b1[threadIdx.x + 4096];
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(v3));
    const long long t3 = rt();
    double v4 = b3[threadIdx.x], v5 = b4[threadIdx.x];
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(v4), "+v"(v5));
    const long long t4 = rt();
    double v6 = b1[threadIdx.x];
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(v6));
    const long long t5 = rt();
    out[threadIdx.x] = v1 + v2 + v3 + v4 + v5 + v6;
    if (threadIdx.x == 0) { ticks[0] = t1 - t0; ticks[1] = t2 - t1; ticks[2] = t3 - t2; ticks[3] = t4 - t3; ticks[4] = t5 - t4; }
}

int main()
{
    {
        double *b[4], *o; long long *tk, h[5];
        for (int i = 0; i < 4; ++i) hipMalloc(&b[i], 8 * 65536);
        hipMalloc(&o, 8 * 64); hipMalloc(&tk, 8 * 8);
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(k_produce, dim3(256), dim3(256), 0, 0, b[0], b[1], b[2], b[3], 1.0 + rep);
            hipLaunchKernelGGL(k_first_load, dim3(1), dim3(64), 0, 0, b[0], b[1], b[2], b[3], o, tk);
            hipMemcpy(h, tk, 40, hipMemcpyDeviceToHost);
            printf("after producer:  first load %lld0 ns | other buffer %lld0 | same buffer, other line %lld0 | two buffers together %lld0 | same line again %lld0\n", h[0], h[1], h[2], h[3], h[4]);
            hipLaunchKernelGGL(k_first_load, dim3(1), dim3(64), 0, 0, b[0], b[1], b[2], b[3], o, tk);
            hipMemcpy(h, tk, 40, hipMemcpyDeviceToHost);
            printf("relaunched:      first load %lld0 ns | other buffer %lld0 | same buffer, other line %lld0 | two buffers together %lld0 | same line again %lld0\n", h[0], h[1], h[2], h[3], h[4]);
        }
    }
    double *out; long long *ticks, h, h2[2];
    hipMalloc(&out, 8 * 1024); hipMalloc(&ticks, 16);
    const int iters = 20000;
    const int threads[4] = { 64, 256, 512, 1024 };
    for (int t = 0; t < 4; ++t) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_fma_issue<8>, dim3(1), dim3(threads[t]), 0, 0, out, ticks, iters, 1.0);
        hipMemcpy(h2, ticks, 16, hipMemcpyDeviceToHost); h = h2[0];
        printf("fma  8 independent chains, %4d threads: %.2f ns per v_fma_f64 per wave   (s_memtime: %.3f ticks per ns)\n", threads[t], h * 10.0 / (iters * 8.0), h2[1] / (h * 10.0));
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_fma_issue<1>, dim3(1), dim3(threads[t]), 0, 0, out, ticks, iters, 1.0);
        hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
        printf("fma  1 dependent chain,    %4d threads: %.2f ns per v_fma_f64\n", threads[t], h * 10.0 / iters);
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_lds_read, dim3(1), dim3(threads[t]), 0, 0, out, ticks, iters);
        hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
        printf("lds  ds_read_b128 + 2 adds, %4d threads: %.2f ns per read per wave\n", threads[t], h * 10.0 / (iters * 32.0));
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_barrier, dim3(1), dim3(threads[t]), 0, 0, ticks, iters);
        hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
        printf("s_barrier,                  %4d threads: %.2f ns\n", threads[t], h * 10.0 / iters);
    }
    return 0;
}

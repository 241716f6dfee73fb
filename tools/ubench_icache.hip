// ubench_icache.hip -- does straight-line code run slower than the same instructions in a loop (instruction fetch of code
// that is executed once per launch, cold after every kernel boundary)?  One workgroup of 64 threads, N dependent
// v_fma_f64: (a) fully unrolled (8 bytes of code per FMA), (b) a loop of 16.  In-kernel s_memrealtime stamps.
// build: hipcc --offload-arch=gfx950 -O3 ubench_icache.hip -o ubench_icache
#include <hip/hip_runtime.h>
#include <cstdio>

template <int N>
__global__ __launch_bounds__(64) void k_straight(double *out, long long *ticks, double b, double c)
{
    double a = threadIdx.x;
    const long long t0 = wall_clock64();
#pragma unroll
    for (int i = 0; i < N; ++i) a = __builtin_fma(a, b, c);
    asm volatile("" : "+v"(a));
    const long long t1 = wall_clock64();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0) ticks[0] = t1 - t0;
}
__global__ __launch_bounds__(64) void k_loop(double *out, long long *ticks, double b, double c, int n)
{
    double a = threadIdx.x;
    const long long t0 = wall_clock64();
    for (int i = 0; i < n; i += 16) {
#pragma unroll
        for (int u = 0; u < 16; ++u) a = __builtin_fma(a, b, c);
    }
    asm volatile("" : "+v"(a));
    const long long t1 = wall_clock64();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0) ticks[0] = t1 - t0;
}
__global__ void k_other(double *out) { out[threadIdx.x] = 2.0; }

int main()
{
    double *out; long long *ticks, h;
    (void)hipMalloc(&out, 8 * 256); (void)hipMalloc(&ticks, 8);
    for (int rep = 0; rep < 4; ++rep) {
        hipLaunchKernelGGL(k_other, dim3(256), dim3(256), 0, 0, out);
        hipLaunchKernelGGL(k_straight<4096>, dim3(1), dim3(64), 0, 0, out, ticks, 1.0000001, 1e-9);
        (void)hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
        printf("4096 dependent FMAs, straight-line (32 KB of code): %lld0 ns\n", h);
        hipLaunchKernelGGL(k_straight<4096>, dim3(1), dim3(64), 0, 0, out, ticks, 1.0000001, 1e-9);
        hipLaunchKernelGGL(k_straight<4096>, dim3(1), dim3(64), 0, 0, out, ticks, 1.0000001, 1e-9);
        (void)hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
        printf("   ... third launch in a row:                        %lld0 ns\n", h);
        hipLaunchKernelGGL(k_other, dim3(256), dim3(256), 0, 0, out);
        hipLaunchKernelGGL(k_loop, dim3(1), dim3(64), 0, 0, out, ticks, 1.0000001, 1e-9, 4096);
        (void)hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
        printf("4096 dependent FMAs, loop of 16:                     %lld0 ns\n", h);
    }
    return 0;
}

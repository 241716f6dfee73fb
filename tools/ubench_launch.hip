// ubench_launch.hip -- what makes an early-exit launch of the solver's kernels 4.6 us under rocprofv3 when an isolated tiny
// kernel is 1.2-2.5 us: candidates are the register allocation of the waves, the LDS allocation, the size of the code
// and the number of workgroups.  Every kernel below loads a flag and returns when it is set (it is); the body behind
// the return is never executed but shapes the resources.   Run under rocprofv3 --kernel-trace --stats.
// build: hipcc --offload-arch=gfx950 -O3 ubench_launch.hip -o ubench_launch
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ __launch_bounds__(256) void k_plain(const int *flag, double *out)
{
    if (*flag) return;
    out[threadIdx.x] = 1.0;
}
// many live registers behind the early exit
__global__ __launch_bounds__(256) void k_regs(const int *flag, double *out, const double *in)
{
    if (*flag) return;
    double a[96];
#pragma unroll
    for (int i = 0; i < 96; ++i) a[i] = in[threadIdx.x + 256 * i];
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 96; ++i) s = __builtin_fma(s, a[i], a[(i * 7) % 96]);
    out[threadIdx.x] = s;
}
// 60 KB of static LDS
__global__ __launch_bounds__(256) void k_lds(const int *flag, double *out)
{
    __shared__ double buf[7680];
    if (*flag) return;
    for (int i = threadIdx.x; i < 7680; i += 256) buf[i] = i;
    __syncthreads();
    out[threadIdx.x] = buf[(threadIdx.x * 37) % 7680];
}
// a lot of code behind the early exit
template <int N>
__global__ __launch_bounds__(256) void k_code(const int *flag, double *out, const double *in)
{
    if (*flag) return;
    double s = in[threadIdx.x];
#pragma unroll
    for (int i = 0; i < N; ++i) s = __builtin_fma(s, 1.0000001 + i * 1e-9, in[(threadIdx.x + i) & 1023]);
    out[threadIdx.x] = s;
}

int main()
{
    int *flag; double *out, *in;
    (void)hipMalloc(&flag, 4); (void)hipMalloc(&out, 8 * 1024); (void)hipMalloc(&in, 8 * 256 * 128);
    const int one = 1;
    (void)hipMemcpy(flag, &one, 4, hipMemcpyHostToDevice);
    (void)hipMemset(in, 0, 8 * 256 * 128);
    for (int rep = 0; rep < 20; ++rep) {
        for (int grid : { 1, 5, 160, 640 }) {
            hipLaunchKernelGGL(k_plain, dim3(grid), dim3(256), 0, 0, flag, out);
            hipLaunchKernelGGL(k_regs, dim3(grid), dim3(256), 0, 0, flag, out, in);
            hipLaunchKernelGGL(k_lds, dim3(grid), dim3(256), 0, 0, flag, out);
            hipLaunchKernelGGL(k_code<2000>, dim3(grid), dim3(256), 0, 0, flag, out, in);
        }
    }
    (void)hipDeviceSynchronize();
    // the same on a created stream, thousands of launches enqueued ahead (the solver's situation)
    hipStream_t st; (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    for (int rep = 0; rep < 500; ++rep) {
        hipLaunchKernelGGL(k_plain, dim3(7), dim3(256), 0, st, flag, out);
        hipLaunchKernelGGL(k_regs, dim3(7), dim3(256), 0, st, flag, out, in);
        hipLaunchKernelGGL(k_lds, dim3(7), dim3(256), 0, st, flag, out);
        hipLaunchKernelGGL(k_code<2000>, dim3(7), dim3(256), 0, st, flag, out, in);
    }
    (void)hipStreamSynchronize(st);
    // dispatch rate of dependent launches without a profiler: host clock around 4000 early-exit launches
    {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int trial = 0; trial < 3; ++trial) {
            (void)hipEventRecord(e0, st);
            const auto h0 = std::chrono::steady_clock::now();
            for (int rep = 0; rep < 1000; ++rep) {
                hipLaunchKernelGGL(k_plain, dim3(7), dim3(256), 0, st, flag, out);
                hipLaunchKernelGGL(k_regs, dim3(160), dim3(256), 0, st, flag, out, in);
                hipLaunchKernelGGL(k_lds, dim3(640), dim3(256), 0, st, flag, out);
                hipLaunchKernelGGL(k_code<2000>, dim3(7), dim3(256), 0, st, flag, out, in);
            }
            (void)hipEventRecord(e1, st);
            const auto h1 = std::chrono::steady_clock::now();
            (void)hipEventSynchronize(e1);
            printf("   host time to enqueue them: %.2f us per launch\n", std::chrono::duration<double, std::micro>(h1 - h0).count() / 4000.0);
            float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
            printf("4000 early-exit launches in a row: %.2f us per launch\n", ms * 1000.0 / 4000.0);
        }
    }
    printf("done\n");
    return 0;
}

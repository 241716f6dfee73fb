// ubench_kernarg.hip -- what the head of a kernel pays for its arguments: time from the first instruction to the first
// data load's return, (a) pointer in a small argument list, (b) pointer at the end of a ~900-byte argument struct (the
// solver's kernels take DevProblem + DevState by value), (c) the struct behind ONE pointer in device memory.
// build: hipcc --offload-arch=gfx950 -O3 ubench_kernarg.hip -o ubench_kernarg [-mllvm -amdgpu-kernarg-preload-count=16]
#include <hip/hip_runtime.h>
#include <cstdio>

struct Big { int pad[220]; const double *p; long long *ticks; };

__global__ void k_fill(double *b) { b[blockIdx.x * blockDim.x + threadIdx.x] = 1.0 + threadIdx.x; }

__global__ __launch_bounds__(64) void k_small(const double *p, long long *ticks, double *out)
{
    const long long t0 = wall_clock64();
    double v = p[threadIdx.x];
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(v));
    const long long t1 = wall_clock64();
    out[threadIdx.x] = v;
    if (threadIdx.x == 0) ticks[0] = t1 - t0;
}
__global__ __launch_bounds__(64) void k_big(Big b, double *out)
{
    const long long t0 = wall_clock64();
    double v = b.p[threadIdx.x];
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(v));
    const long long t1 = wall_clock64();
    out[threadIdx.x] = v + b.pad[threadIdx.x & 127];
    if (threadIdx.x == 0) b.ticks[0] = t1 - t0;
}
__global__ __launch_bounds__(64) void k_indirect(const Big *pb, double *out)
{
    const long long t0 = wall_clock64();
    const Big &b = *pb;
    double v = b.p[threadIdx.x];
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(v));
    const long long t1 = wall_clock64();
    out[threadIdx.x] = v + b.pad[threadIdx.x & 127];
    if (threadIdx.x == 0) b.ticks[0] = t1 - t0;
}

int main()
{
    double *buf, *out; long long *ticks, h;
    (void)hipMalloc(&buf, 8 * 65536); (void)hipMalloc(&out, 8 * 64); (void)hipMalloc(&ticks, 8);
    Big hb{}; hb.p = buf; hb.ticks = ticks;
    Big *db; (void)hipMalloc(&db, sizeof(Big)); (void)hipMemcpy(db, &hb, sizeof(Big), hipMemcpyHostToDevice);
    for (int rep = 0; rep < 4; ++rep) {
        hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, 0, buf);
        hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, 0, buf, ticks, out);
        (void)hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
        printf("small argument list:   %lld0 ns to the first datum\n", h);
        hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, 0, buf);
        hipLaunchKernelGGL(k_big, dim3(1), dim3(64), 0, 0, hb, out);
        (void)hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
        printf("900-byte argument:     %lld0 ns\n", h);
        hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, 0, buf);
        hipLaunchKernelGGL(k_indirect, dim3(1), dim3(64), 0, 0, db, out);
        (void)hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
        printf("struct behind pointer: %lld0 ns\n", h);
    }
    return 0;
}

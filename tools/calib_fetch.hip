// calib_fetch.hip -- calibration of the rocprofv3 FETCH_SIZE / WRITE_SIZE counters on this box for the access
// widths the library's kernels use (MI355X_MICROARCH.md, "HBM": only 16 B/lane streaming reads are calibrated
// there).  Three kernels stream a buffer of known size: 8 B/lane loads (observations in k_eval_gram), 16 B/lane
// loads, and 8 B/lane stores (per-view records).  Run:
//   hipcc -O3 --offload-arch=gfx950 tools/calib_fetch.hip -o /tmp/calib_fetch
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- /tmp/calib_fetch   (and WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void read8(const double *__restrict__ p, size_t n, double *out)
{
    double s = 0.0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
    if (s == 123.456) out[0] = s;
}
__global__ void read16(const double2 *__restrict__ p, size_t n, double *out)
{
    double s = 0.0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const double2 v = p[i]; s += v.x + v.y; }
    if (s == 123.456) out[0] = s;
}
__global__ void write8(double *__restrict__ p, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (double)i;
}

int main()
{
    const size_t n = (size_t)1 << 27;          // 1 GiB of doubles: past the 256 MiB Infinity Cache
    double *d = nullptr, *o = nullptr;
    if (hipMalloc(&d, n * sizeof(double)) != hipSuccess || hipMalloc(&o, 64) != hipSuccess) { std::printf("alloc failed\n"); return 1; }
    hipMemset(d, 0, n * sizeof(double));
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(read8, dim3(4096), dim3(256), 0, 0, d, n, o);
        hipLaunchKernelGGL(read16, dim3(4096), dim3(256), 0, 0, reinterpret_cast<const double2 *>(d), n / 2, o);
        hipLaunchKernelGGL(write8, dim3(4096), dim3(256), 0, 0, d, n);
    }
    hipDeviceSynchronize();
    std::printf("bytes per kernel: %zu\n", n * sizeof(double));
    return 0;
}

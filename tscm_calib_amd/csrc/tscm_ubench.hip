// tscm_ubench.hip -- measured fp64 ceilings of the device the solver runs on (C ABI: tscm_device_peak_fp64).
//
// MI355X_MICROARCH.md lists no fp64 peak; the datasheet figure (78.6 TFLOP/s, vector = matrix) is not sustained on
// the gpurun boxes (clock / power state).  The benchmark therefore measures, on the same device and in the same
// process as the timed solve, what v_mfma_f64_16x16x4_f64 and v_fma_f64 actually deliver with every CU busy, and
// reports the dominant kernel against both the datasheet peak and this measured ceiling (bench.py).
// Not on any solver path.
#include "tscm/tscm.h"

#include <hip/hip_runtime.h>

#include <string>

int tscm_set_error(int code, const std::string &msg);      // tscm_solver.hip

typedef double d4 __attribute__((ext_vector_type(4)));

namespace {

// eight independent FMA chains per lane: enough to cover the dependent-issue latency at 2+ waves per SIMD
__global__ __launch_bounds__(256) void k_peak_fma(double *out, int iters, double seed)
{
    double a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = seed + i + threadIdx.x;
    const double b = 1.0000001, c = 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = __builtin_fma(a[i], b, c);
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// four independent accumulator tiles per wave
__global__ __launch_bounds__(256) void k_peak_mfma(double *out, int iters, double seed)
{
    d4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = d4{ seed, 0.0, 0.0, 0.0 };
    const double a = 1.0 + threadIdx.x * 1e-6, b = 1.0 - threadIdx.x * 1e-6;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// eight independent 4x4 accumulator blocks per wave: v_mfma_f64_4x4x4_4b_f64, the instruction k_eval_gram4 uses
__global__ __launch_bounds__(256) void k_peak_mfma4(double *out, int iters, double seed)
{
    double acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = seed;
    const double a = 1.0 + threadIdx.x * 1e-6, b = 1.0 - threadIdx.x * 1e-6;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// four independent fp32 accumulator tiles per wave: v_mfma_f32_16x16x4_f32, the instruction k_eval_gram_f32 contracts with
typedef float f4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_peak_mfma_f32(double *out, int iters, float seed)
{
    f4v acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f4v{ seed, 0.f, 0.f, 0.f };
    const float a = 1.f + threadIdx.x * 1e-6f, b = 1.f - threadIdx.x * 1e-6f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
bool time_launches(F launch, int reps, double *ms_per_launch)
{
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return false;
    launch();                                                    // warm-up (code load, clocks)
    launch();
    bool ok = hipDeviceSynchronize() == hipSuccess;
    ok = ok && hipEventRecord(e0, 0) == hipSuccess;
    for (int r = 0; r < reps; ++r) launch();
    ok = ok && hipEventRecord(e1, 0) == hipSuccess && hipEventSynchronize(e1) == hipSuccess;
    float ms = 0.f;
    ok = ok && hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *ms_per_launch = ms / reps;
    return ok && hipGetLastError() == hipSuccess;
}

}  // namespace

// peaks[0] = v_mfma_f64_16x16x4_f64, peaks[1] = v_mfma_f64_4x4x4_4b_f64, peaks[2] = v_fma_f64, TFLOP/s
extern "C" int tscm_device_peak_fp64_ex(int device, double peaks[3])
{
    if (!peaks) return tscm_set_error(TSCM_E_INVALID, "NULL argument");
    if (int rc = tscm_device_peak_fp64(device, &peaks[0], &peaks[2])) return rc;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return tscm_set_error(TSCM_E_HIP, "hipGetDeviceProperties failed");
    const int blocks = prop.multiProcessorCount * 8;
    double *out = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&out), sizeof(double) * 256 * (size_t)blocks) != hipSuccess) return tscm_set_error(TSCM_E_NOMEM, "hipMalloc failed");
    const int iters = 8000, reps = 5;
    double ms = 0.0;
    const bool ok = time_launches([&] { hipLaunchKernelGGL(k_peak_mfma4, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0); }, reps, &ms);
    (void)hipFree(out);
    if (!ok || ms <= 0.0) return tscm_set_error(TSCM_E_HIP, "fp64 peak measurement failed");
    peaks[1] = 512.0 * 8.0 * iters * 4.0 * blocks / (ms * 1e-3) / 1e12;       // 4 blocks x 4x4x4x2 flop per instruction, 8 per wave and iteration, 4 waves
    return 0;
}

// v_mfma_f32_16x16x4_f32 with every CU busy, TFLOP/s: the ceiling of the contraction of the fp32-Jacobian tier
extern "C" int tscm_device_peak_fp32_mfma(int device, double *tflops)
{
    if (!tflops) return tscm_set_error(TSCM_E_INVALID, "NULL argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return tscm_set_error(TSCM_E_NO_DEVICE, "no usable HIP device");
    if (hipSetDevice(device) != hipSuccess) return tscm_set_error(TSCM_E_NO_DEVICE, "hipSetDevice failed");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return tscm_set_error(TSCM_E_HIP, "hipGetDeviceProperties failed");
    const int blocks = prop.multiProcessorCount * 8;
    double *out = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&out), sizeof(double) * 256 * (size_t)blocks) != hipSuccess) return tscm_set_error(TSCM_E_NOMEM, "hipMalloc failed");
    const int iters = 8000, reps = 5;
    double ms = 0.0;
    const bool ok = time_launches([&] { hipLaunchKernelGGL(k_peak_mfma_f32, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f); }, reps, &ms);
    (void)hipFree(out);
    if (!ok || ms <= 0.0) return tscm_set_error(TSCM_E_HIP, "fp32 MFMA peak measurement failed");
    *tflops = 2048.0 * 4.0 * iters * 4.0 * blocks / (ms * 1e-3) / 1e12;       // 16x16x4x2 flop per instruction, 4 per wave and iteration, 4 waves
    return 0;
}

extern "C" int tscm_device_peak_fp64(int device, double *mfma_tflops, double *valu_tflops)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return tscm_set_error(TSCM_E_NO_DEVICE, "no usable HIP device");
    if (hipSetDevice(device) != hipSuccess) return tscm_set_error(TSCM_E_NO_DEVICE, "hipSetDevice failed");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return tscm_set_error(TSCM_E_HIP, "hipGetDeviceProperties failed");
    const int blocks = prop.multiProcessorCount * 8;             // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    double *out = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&out), sizeof(double) * 256 * (size_t)blocks) != hipSuccess) return tscm_set_error(TSCM_E_NOMEM, "hipMalloc failed");
    const int iters = 4000, reps = 5;
    double ms_fma = 0.0, ms_mfma = 0.0;
    const bool ok1 = time_launches([&] { hipLaunchKernelGGL(k_peak_fma, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0); }, reps, &ms_fma);
    const bool ok2 = time_launches([&] { hipLaunchKernelGGL(k_peak_mfma, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0); }, reps, &ms_mfma);
    (void)hipFree(out);
    if (!ok1 || !ok2 || ms_fma <= 0.0 || ms_mfma <= 0.0) return tscm_set_error(TSCM_E_HIP, "fp64 peak measurement failed");
    const double fl_fma = 2.0 * 8.0 * iters * 256.0 * blocks;                  // 8 FMAs per lane and iteration
    const double fl_mfma = 2048.0 * 4.0 * iters * 4.0 * blocks;               // 16x16x4x2 flop per instruction, 4 per wave and iteration, 4 waves
    if (valu_tflops) *valu_tflops = fl_fma / (ms_fma * 1e-3) / 1e12;
    if (mfma_tflops) *mfma_tflops = fl_mfma / (ms_mfma * 1e-3) / 1e12;
    return 0;
}

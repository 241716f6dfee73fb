// tscm_maps.hip -- remap-table generation: TripleSphereCamera::undistort
// (TS.cpp:284-306), the table of undistort_chessboard (TS.cpp:308-330) and the rectification tables
// of EpipolarRectify/rectify.cpp:86-199, all instances of one per-pixel loop (see tscm.h).
// One thread per 4 consecutive output pixels of a row, float4 stores where the row allows it;
// the map descriptor is wave-uniform (blockIdx.y) and read through the scalar path.
#include "tscm/tscm.h"

#include <hip/hip_runtime.h>

#include "tscm_fastmath.h"

#include <string>
#include <vector>

using namespace tscm;

int tscm_set_error(int code, const std::string &msg);   // tscm_solver.hip

#define MAP_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) return tscm_set_error(TSCM_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

namespace {

// the reference's arithmetic, operation by operation (x86-64 gcc without FMA contraction)
__device__ __forceinline__ void map_pixel_exact(const tscm_map_desc &m, double beta, int i, int j, float &mx, float &my)
{
    const double x0 = __ddiv_rn(__dsub_rn((double)j, m.cx), m.fx);
    const double y0 = __ddiv_rn(__dsub_rn((double)i, m.cy), m.fy);
    // cv::Mat product R * p: sum over k in order, starting from the first product
    const double X = __dadd_rn(__dadd_rn(__dmul_rn(m.R[0], x0), __dmul_rn(m.R[1], y0)), m.R[2]);
    const double Y = __dadd_rn(__dadd_rn(__dmul_rn(m.R[3], x0), __dmul_rn(m.R[4], y0)), m.R[5]);
    const double Z = __dadd_rn(__dadd_rn(__dmul_rn(m.R[6], x0), __dmul_rn(m.R[7], y0)), m.R[8]);
    const double rho2 = __dadd_rn(__dmul_rn(X, X), __dmul_rn(Y, Y));
    const double d1 = __dsqrt_rn(__dadd_rn(rho2, __dmul_rn(Z, Z)));
    const double z1 = __dadd_rn(Z, __dmul_rn(m.intr[4], d1));
    const double d2 = __dsqrt_rn(__dadd_rn(rho2, __dmul_rn(z1, z1)));
    const double z2 = __dadd_rn(z1, __dmul_rn(m.intr[5], d2));
    const double d3 = __dsqrt_rn(__dadd_rn(rho2, __dmul_rn(z2, z2)));
    const double ksai = __dadd_rn(z2, __dmul_rn(beta, d3));
    double u = __dadd_rn(__dadd_rn(__ddiv_rn(__dmul_rn(m.intr[0], X), ksai), __ddiv_rn(__dmul_rn(m.intr[7], Y), ksai)), m.intr[2]);
    double v = __dadd_rn(__dadd_rn(__ddiv_rn(__dmul_rn(m.intr[8], X), ksai), __ddiv_rn(__dmul_rn(m.intr[1], Y), ksai)), m.intr[3]);
    if (m.check_w2 && Z <= __dmul_rn(-m.w2, d1)) { u = -1.0; v = -1.0; }
    mx = (float)__dadd_rn(u, m.offset_x);
    my = (float)__dadd_rn(v, m.offset_y);
}

__device__ __forceinline__ void map_pixel_fast(const tscm_map_desc &m, double beta, double ifx, double ify, int i, int j, float &mx, float &my)
{
    const double x0 = ((double)j - m.cx) * ifx, y0 = ((double)i - m.cy) * ify;
    const double X = __builtin_fma(m.R[0], x0, __builtin_fma(m.R[1], y0, m.R[2]));
    const double Y = __builtin_fma(m.R[3], x0, __builtin_fma(m.R[4], y0, m.R[5]));
    const double Z = __builtin_fma(m.R[6], x0, __builtin_fma(m.R[7], y0, m.R[8]));
    const double rho2 = __builtin_fma(Y, Y, X * X);
    const double s1 = __builtin_fma(Z, Z, rho2);
    const double d1 = s1 * fast_rsqrt(s1);
    const double z1 = __builtin_fma(m.intr[4], d1, Z);
    const double s2 = __builtin_fma(z1, z1, rho2);
    const double d2 = s2 * fast_rsqrt(s2);
    const double z2 = __builtin_fma(m.intr[5], d2, z1);
    const double s3 = __builtin_fma(z2, z2, rho2);
    const double d3 = s3 * fast_rsqrt(s3);
    const double ik = fast_rcp(__builtin_fma(beta, d3, z2));
    const double xn = X * ik, yn = Y * ik;
    double u = __builtin_fma(m.intr[0], xn, __builtin_fma(m.intr[7], yn, m.intr[2]));
    double v = __builtin_fma(m.intr[8], xn, __builtin_fma(m.intr[1], yn, m.intr[3]));
    if (m.check_w2 && Z <= -m.w2 * d1) { u = -1.0; v = -1.0; }
    mx = (float)(u + m.offset_x);
    my = (float)(v + m.offset_y);
}

// grid (ceil(max quads per map / 256), n_maps) x 256.  A quad = 4 consecutive output ELEMENTS:
// for a contiguous table (out_stride == width) quads run over the flat element index, crossing row
// ends, so that every store is one aligned float4 when out_offset is a multiple of 4; tables with
// row padding use quads inside a row.
template <bool EXACT>
__global__ __launch_bounds__(256) void k_build_maps(const tscm_map_desc *__restrict__ maps, float *__restrict__ mapx, float *__restrict__ mapy)
{
    const tscm_map_desc m = maps[blockIdx.y];
    const bool flat = m.out_stride == m.width;
    const int qpr = (m.width + 3) >> 2;                       // quads per row (padded tables)
    const long long total = (long long)m.width * m.height;
    const long long nquads = flat ? (total + 3) >> 2 : (long long)qpr * m.height;
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    if (q >= nquads) return;
    // row / column of the quad's first element without an integer division: fp64 quotient + one correction
    const int den = flat ? m.width : qpr;
    const long long num = flat ? 4 * q : q;
    int i = (int)((double)num * fast_rcp((double)den));
    int r = (int)(num - (long long)i * den);
    if (r < 0) { --i; r += den; }
    if (r >= den) { ++i; r -= den; }
    int j = flat ? r : r * 4;
    const double beta = EXACT ? __ddiv_rn(m.intr[6], __dsub_rn(1.0, m.intr[6])) : m.intr[6] * fast_rcp(1.0 - m.intr[6]);
    const double ifx = EXACT ? 0.0 : fast_rcp(m.fx), ify = EXACT ? 0.0 : fast_rcp(m.fy);
    const long long base = m.out_offset + (flat ? 4 * q : (long long)i * m.out_stride + j);
    const int n = flat ? (int)min(4LL, total - 4 * q) : min(4, m.width - j);
    float ox[4], oy[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (EXACT) map_pixel_exact(m, beta, i, j, ox[k], oy[k]);
        else map_pixel_fast(m, beta, ifx, ify, i, j, ox[k], oy[k]);
        if (++j == m.width) { j = 0; ++i; }                  // flat quads continue on the next row
    }
    if (n == 4 && (base & 3) == 0) {
        *reinterpret_cast<float4 *>(mapx + base) = make_float4(ox[0], ox[1], ox[2], ox[3]);
        *reinterpret_cast<float4 *>(mapy + base) = make_float4(oy[0], oy[1], oy[2], oy[3]);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < n) { mapx[base + k] = ox[k]; mapy[base + k] = oy[k]; }
    }
}

template <typename T>
struct DevBuf {
    T *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc(reinterpret_cast<void **>(&p), (n ? n : 1) * sizeof(T)); }
};

}  // namespace

extern "C" int tscm_build_maps(const tscm_map_desc *maps, int n_maps, int device, int exact, float *mapx, float *mapy, size_t n_elems,
                               double *seconds_kernel)
{
    if (n_maps < 0 || (n_maps > 0 && (!maps || !mapx || !mapy))) return tscm_set_error(TSCM_E_INVALID, "NULL argument");
    if (n_maps > 65535) return tscm_set_error(TSCM_E_UNSUPPORTED, "more than 65535 maps in one call");
    long long max_quads = 0;
    unsigned long long covered = 0;
    for (int m = 0; m < n_maps; ++m) {
        const tscm_map_desc &d = maps[m];
        if (d.width < 0 || d.height < 0 || d.out_stride < d.width || d.out_offset < 0) return tscm_set_error(TSCM_E_INVALID, "map " + std::to_string(m) + ": bad geometry");
        if (d.width == 0 || d.height == 0) continue;
        const unsigned long long last = (unsigned long long)d.out_offset + (unsigned long long)(d.height - 1) * d.out_stride + d.width;
        if (last > n_elems) return tscm_set_error(TSCM_E_INVALID, "map " + std::to_string(m) + " does not fit the output arrays");
        max_quads = std::max(max_quads, d.out_stride == d.width ? ((long long)d.width * d.height + 3) / 4 : (long long)((d.width + 3) / 4) * d.height);
        covered += (unsigned long long)d.width * d.height;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return tscm_set_error(TSCM_E_NO_DEVICE, "no HIP device available (tscm_build_maps has no CPU fallback)");
    if (device < 0 || device >= ndev) return tscm_set_error(TSCM_E_NO_DEVICE, "device index out of range");
    MAP_TRY(hipSetDevice(device));
    if (seconds_kernel) *seconds_kernel = 0.0;
    if (n_maps == 0 || max_quads == 0) return 0;
    DevBuf<tscm_map_desc> d_maps;
    DevBuf<float> d_x, d_y;
    MAP_TRY(d_maps.alloc(n_maps)); MAP_TRY(d_x.alloc(n_elems)); MAP_TRY(d_y.alloc(n_elems));
    MAP_TRY(hipMemcpy(d_maps.p, maps, sizeof(tscm_map_desc) * n_maps, hipMemcpyHostToDevice));
    // elements no map covers (row padding, gaps) keep the caller's values
    if (covered < n_elems) {
        MAP_TRY(hipMemcpy(d_x.p, mapx, sizeof(float) * n_elems, hipMemcpyHostToDevice));
        MAP_TRY(hipMemcpy(d_y.p, mapy, sizeof(float) * n_elems, hipMemcpyHostToDevice));
    }
    hipEvent_t e0, e1;
    MAP_TRY(hipEventCreate(&e0)); MAP_TRY(hipEventCreate(&e1));
    MAP_TRY(hipEventRecord(e0, 0));
    const dim3 grid((unsigned)((max_quads + 255) / 256), (unsigned)n_maps);
    if (exact) hipLaunchKernelGGL(k_build_maps<true>, grid, dim3(256), 0, 0, d_maps.p, d_x.p, d_y.p);
    else hipLaunchKernelGGL(k_build_maps<false>, grid, dim3(256), 0, 0, d_maps.p, d_x.p, d_y.p);
    MAP_TRY(hipEventRecord(e1, 0));
    MAP_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    MAP_TRY(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    MAP_TRY(hipGetLastError());
    if (seconds_kernel) *seconds_kernel = 1e-3 * ms;
    MAP_TRY(hipMemcpy(mapx, d_x.p, sizeof(float) * n_elems, hipMemcpyDeviceToHost));
    MAP_TRY(hipMemcpy(mapy, d_y.p, sizeof(float) * n_elems, hipMemcpyDeviceToHost));
    return 0;
}

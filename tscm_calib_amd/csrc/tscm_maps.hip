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

// ------------------------------------------------------------------------------------------------
// cv::remap(src, dst, mapx, mapy, INTER_LINEAR) for 8-bit images (TS.cpp:304, :329), border constant 0, optionally
// followed by BGR2GRAY (findCorner.cpp:9-10).  OpenCV's fixed-point scheme (see oracle/tscm_oracle_remap.c): map
// coordinates rounded to 1/32 pixel, 15-bit weights, (sum + 2^14) >> 15.  Integer arithmetic: bit-identical to the oracle.
namespace {

template <int CH>
__global__ __launch_bounds__(256) void k_remap(const unsigned char *src, int w, int h, int stride, const float *mapx, const float *mapy, int map_w, int map_h,
                                               int to_gray, unsigned char *dst, int dst_stride)
{
    const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (j >= map_w) return;
    const int sx = __float2int_rn(mapx[(size_t)i * map_w + j] * 32.0f), sy = __float2int_rn(mapy[(size_t)i * map_w + j] * 32.0f);
    const int ix = max(-32768, min(32767, sx >> 5)), iy = max(-32768, min(32767, sy >> 5));
    const int fx = sx & 31, fy = sy & 31;
    int wgt[4] = { 32 * (32 - fx) * (32 - fy), 32 * fx * (32 - fy), 32 * (32 - fx) * fy, 32 * fx * fy };
    if (wgt[0] == 32768) { wgt[0] = 32767; wgt[3] = 1; }
    int px[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        int acc = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int x = ix + (k & 1), y = iy + (k >> 1);
            const int p = (x >= 0 && x < w && y >= 0 && y < h) ? src[(size_t)y * stride + (size_t)x * CH + c] : 0;
            acc += wgt[k] * p;
        }
        px[c] = max(0, min(255, (acc + (1 << 14)) >> 15));
    }
    if (CH == 3 && to_gray) dst[(size_t)i * dst_stride + j] = (unsigned char)((px[0] * 1868 + px[CH > 1 ? 1 : 0] * 9617 + px[CH > 2 ? 2 : 0] * 4899 + (1 << 13)) >> 14);
    else {
#pragma unroll
        for (int c = 0; c < CH; ++c) dst[(size_t)i * dst_stride + (size_t)j * CH + c] = (unsigned char)px[c];
    }
}

struct DevBytes {
    void *p = nullptr;
    ~DevBytes() { if (p) (void)hipFree(p); }
};

}  // namespace

extern "C" int tscm_remap(const unsigned char *src, int width, int height, int stride, int channels, const float *mapx, const float *mapy, int map_width,
                          int map_height, int map_stride, int to_gray, int device, unsigned char *dst, int dst_stride)
{
    if (!src || !mapx || !mapy || !dst) return tscm_set_error(TSCM_E_INVALID, "NULL argument");
    if (channels != 1 && channels != 3) return tscm_set_error(TSCM_E_UNSUPPORTED, "remap: 1 or 3 channels");
    const int out_ch = (to_gray || channels == 1) ? 1 : channels;
    if (width < 1 || height < 1 || stride < width * channels || map_width < 0 || map_height < 0 || map_stride < map_width || dst_stride < map_width * out_ch)
        return tscm_set_error(TSCM_E_INVALID, "bad image / map description");
    if (width > 32767 || height > 32767) return tscm_set_error(TSCM_E_UNSUPPORTED, "images beyond 32767 pixels per side");
    if (map_width == 0 || map_height == 0) return 0;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return tscm_set_error(TSCM_E_NO_DEVICE, "no HIP device available (remap has no CPU fallback)");
    if (device < 0 || device >= ndev) return tscm_set_error(TSCM_E_NO_DEVICE, "device index out of range");
    MAP_TRY(hipSetDevice(device));
    DevBytes d_src, d_mx, d_my, d_dst;
    const size_t nmap = (size_t)map_width * map_height, dst_row = (size_t)map_width * out_ch;
    MAP_TRY(hipMalloc(&d_src.p, (size_t)stride * height));
    MAP_TRY(hipMalloc(&d_mx.p, nmap * sizeof(float))); MAP_TRY(hipMalloc(&d_my.p, nmap * sizeof(float)));
    MAP_TRY(hipMalloc(&d_dst.p, dst_row * map_height));
    MAP_TRY(hipMemcpy(d_src.p, src, (size_t)stride * height, hipMemcpyHostToDevice));
    MAP_TRY(hipMemcpy2D(d_mx.p, (size_t)map_width * sizeof(float), mapx, (size_t)map_stride * sizeof(float), (size_t)map_width * sizeof(float), map_height, hipMemcpyHostToDevice));
    MAP_TRY(hipMemcpy2D(d_my.p, (size_t)map_width * sizeof(float), mapy, (size_t)map_stride * sizeof(float), (size_t)map_width * sizeof(float), map_height, hipMemcpyHostToDevice));
    const dim3 grid((map_width + 255) / 256, map_height);
    if (channels == 1)
        hipLaunchKernelGGL(k_remap<1>, grid, dim3(256), 0, nullptr, static_cast<const unsigned char *>(d_src.p), width, height, stride, static_cast<const float *>(d_mx.p),
                           static_cast<const float *>(d_my.p), map_width, map_height, 0, static_cast<unsigned char *>(d_dst.p), (int)dst_row);
    else
        hipLaunchKernelGGL(k_remap<3>, grid, dim3(256), 0, nullptr, static_cast<const unsigned char *>(d_src.p), width, height, stride, static_cast<const float *>(d_mx.p),
                           static_cast<const float *>(d_my.p), map_width, map_height, to_gray ? 1 : 0, static_cast<unsigned char *>(d_dst.p), (int)dst_row);
    MAP_TRY(hipGetLastError());
    MAP_TRY(hipMemcpy2D(dst, (size_t)dst_stride, d_dst.p, dst_row, dst_row, map_height, hipMemcpyDeviceToHost));
    return 0;
}

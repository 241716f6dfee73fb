// tscm_corners.hip -- chessboard-corner candidates of a grey image on the GPU (first stage):
// findCorner() up to and including the score filter (DetectCorner/findCorner.cpp:7-66) plus the sub-pixel fit
// of subPixelLocation (:492-541) for every candidate.  The chessboard structure recovery
// (DetectCorner/chessboard.cpp) consumes exactly this output and is not part of this file.
//
//   k_corner_gradients   3x3 derivative filters on the raw grey values -> edge angle, gradient magnitude; min / max
//   k_gauss_rows/_cols   separable Gaussian (7 sigma + 1 taps) of the normalised image, BORDER_REFLECT_101
//   k_corner_metric      first and second derivatives fused: cxy + c45 (suppression input) and Ixy (sub-pixel fit)
//   k_nms_cells          one thread per (n + 1)^2 cell of nonMaximumSuppression, k_nms_compact keeps the cell order
//   k_corner_describe    one wave per candidate: orientation histogram + mode seeking, three-radius correlation
//                        score, quadratic sub-pixel fit
// Everything per pixel is HBM-bound fp64 streaming.  The arithmetic follows the oracle operation by operation (no
// FMA contraction in this file); transcendental constants (Gaussian taps, normpdf tables, bin directions) are
// computed on the host with the C library the CPU path would use, so the image planes are bit-identical to the
// oracle's except for atan2 (device libm, <= 2 ulp; the exact cases du = 0, dv = 0, |du| = |dv| -- which sit on
// histogram-bin boundaries -- use host constants) and the per-candidate sums (wave-parallel order).
#pragma clang fp contract(off)
#include "tscm/tscm.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

int tscm_set_error(int code, const std::string &msg);   // tscm_solver.hip

#define CRN_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) return tscm_set_error(TSCM_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

namespace {

constexpr int kNmsN = 4, kNmsMargin = 5, kOrientR = 10, kBins = 32;
constexpr double kNmsTau = 0.07;
constexpr double kPi = 3.14159265358979323846;

constexpr int kMmSlots = 64, kMmStride = 32;     // min / max accumulators: 64 slots, 128 bytes apart

struct AtanConsts { double pp, pn, np, nn, p0, n0, zn; };      // atan2(1,1), (1,-1), (-1,1), (-1,-1), (1,0), (-1,0), (0,-1)

struct DescribeTables {
    double smooth[5];                 // normpdf(j, 0, 1), j = -2..2          (findCorner.cpp:195, :296)
    double bcos[kBins], bsin[kBins];  // direction of histogram bin b: angle b pi / 32
    double tsin[kBins], tcos[kBins];  // sin / cos of atan2(bsin, bcos)        (:483, :364-365)
    double X[150];                    // 6 x 25 least-squares operator        (:495-509)
    double npdf[3][513];              // normpdf(sqrt(d2), 0, r / 2) by squared distance, r = 8, 12, 16
};

__device__ __forceinline__ int refl101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
    return i;
}

// min and max of the grey values from the slots; every lane of a full wave calls it
__device__ __forceinline__ void read_extremes(const int *mm, int &mn, int &mx)
{
    const int lane = threadIdx.x & 63;
    mn = mm[kMmStride * lane]; mx = mm[kMmStride * lane + 1];
    for (int off = 32; off > 0; off >>= 1) { mn = min(mn, __shfl_xor(mn, off)); mx = max(mx, __shfl_xor(mx, off)); }
}

// findCorner.cpp:8-29, evaluated where it is needed: the edge angle in [0, pi] and the gradient magnitude of pixel (i, j)
// from the 3x3 derivative filters on the raw grey values (BORDER_REFLECT_101).  Only the 21 x 21 / 33 x 33 windows
// around the ~100 maxima ever read these planes, so they are not materialised for the other 1.4 M pixels.
__device__ __forceinline__ void pixel_gradient(const unsigned char *gray, int w, int h, int stride, int i, int j, double &du, double &dv)
{
    const int im = refl101(i - 1, h), ip = refl101(i + 1, h), jm = refl101(j - 1, w), jp = refl101(j + 1, w);
    const unsigned char *rm = gray + (size_t)im * stride, *r0 = gray + (size_t)i * stride, *rp = gray + (size_t)ip * stride;
    du = ((double)rm[jp] - rm[jm]) + ((double)r0[jp] - r0[jm]) + ((double)rp[jp] - rp[jm]);
    dv = ((double)rp[jm] - rm[jm]) + ((double)rp[j] - rm[j]) + ((double)rp[jp] - rm[jp]);
}
__device__ __forceinline__ double edge_angle(double du, double dv, const AtanConsts &ac)
{
    double a;
    if (dv == 0.0) a = du < 0.0 ? ac.zn : 0.0;
    else if (du == 0.0) a = dv > 0.0 ? ac.p0 : ac.n0;
    else if (fabs(du) == fabs(dv)) a = dv > 0.0 ? (du > 0.0 ? ac.pp : ac.pn) : (du > 0.0 ? ac.np : ac.nn);
    else a = atan2(dv, du);
    if (a < 0) a += kPi;
    if (a > kPi) a -= kPi;
    return a;
}

// min and max of a contiguous image (stride == width): 64 bytes per thread as four 16-byte loads.
// grid (ceil(w*h/16384), n_images) x 256
__global__ __launch_bounds__(256) void k_grey_extremes_flat(const unsigned char *gray, size_t bytes, int *mm)
{
    gray += blockIdx.y * bytes; mm += (size_t)blockIdx.y * kMmSlots * kMmStride;
    int mn = 255, mx = 0;
    const size_t o = ((size_t)blockIdx.x * 256 + threadIdx.x) * 64;
    if (o + 64 <= bytes && ((reinterpret_cast<size_t>(gray) + o) & 15) == 0) {
        const uint4 *r = reinterpret_cast<const uint4 *>(gray + o);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint4 v = r[k];
            const unsigned q[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) { const int g = (q[a] >> (8 * b)) & 0xff; mn = min(mn, g); mx = max(mx, g); }
        }
    } else {
        for (int k = 0; k < 64; ++k) if (o + k < bytes) { const int g = gray[o + k]; mn = min(mn, g); mx = max(mx, g); }
    }
    for (int off = 32; off > 0; off >>= 1) { mn = min(mn, __shfl_xor(mn, off)); mx = max(mx, __shfl_xor(mx, off)); }
    if ((threadIdx.x & 63) == 0) {
        int *slot = mm + kMmStride * ((blockIdx.x * 4 + (threadIdx.x >> 6)) & (kMmSlots - 1));
        const volatile int *cur = slot;
        if (mn < cur[0]) atomicMin(&slot[0], mn);
        if (mx > cur[1]) atomicMax(&slot[1], mx);
    }
}

// min and max of the grey values (normalisation of :30-34).  grid (ceil(w/4096), h, n_images) x 256, 16 pixels of a row per thread
__global__ __launch_bounds__(256) void k_grey_extremes(const unsigned char *gray, int w, int h, int stride, int *mm)
{
    gray += (size_t)blockIdx.z * stride * h; mm += (size_t)blockIdx.z * kMmSlots * kMmStride;
    int mn = 255, mx = 0;
    const unsigned char *row = gray + (size_t)blockIdx.y * stride;
    const int j0 = (blockIdx.x * 256 + threadIdx.x) * 16;
    if (j0 + 16 <= w && ((reinterpret_cast<size_t>(row) + j0) & 3) == 0) {       // four aligned 32-bit loads
        const unsigned *r4 = reinterpret_cast<const unsigned *>(row + j0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned v = r4[k];
#pragma unroll
            for (int b = 0; b < 4; ++b) { const int g = (v >> (8 * b)) & 0xff; mn = min(mn, g); mx = max(mx, g); }
        }
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) if (j0 + k < w) { const int g = row[j0 + k]; mn = min(mn, g); mx = max(mx, g); }
    }
    for (int off = 32; off > 0; off >>= 1) { mn = min(mn, __shfl_xor(mn, off)); mx = max(mx, __shfl_xor(mx, off)); }
    // one atomic pair per wave only while it can still move the extremes (a plain read of a monotone value: a stale
    // one merely costs a redundant atomic), spread over kMmSlots slots on separate cache lines (the consumers reduce
    // the slots): unconditional atomics on two addresses serialise 21 k waves -- 0.5 ms
    if ((threadIdx.x & 63) == 0) {
        int *slot = mm + kMmStride * (((blockIdx.y * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6)) & (kMmSlots - 1));
        const volatile int *cur = slot;
        if (mn < cur[0]) atomicMin(&slot[0], mn);
        if (mx > cur[1]) atomicMax(&slot[1], mx);
    }
}

// rows of GaussianBlur on the normalised image (img - min) / (max - min) (:30-34, :106).  grid (ceil(w/256), h, n_images):
// the 256 + n - 1 normalised inputs of a row segment are staged in LDS once, every thread then sums its n taps
// (in tap order, like the sequential filter).
__global__ __launch_bounds__(256) void k_gauss_rows(const unsigned char *gray, int w, int h, int stride, const int *mm, const double *taps, int n,
                                                    double *tmp, size_t plane)
{
    gray += (size_t)blockIdx.z * stride * h; mm += (size_t)blockIdx.z * kMmSlots * kMmStride; tmp += blockIdx.z * plane;
    __shared__ double seg[256 + 64], k[64];
    int imn, imx;
    read_extremes(mm, imn, imx);
    const double mn = imn, mx = imx;
    __shared__ double lut[256];
    lut[threadIdx.x] = ((double)threadIdx.x - mn) / (mx - mn);          // one division per thread instead of one per staged pixel
    if (threadIdx.x < n) k[threadIdx.x] = taps[threadIdx.x];
    __syncthreads();
    const int i = blockIdx.y, j0 = blockIdx.x * 256, half = n / 2;
    const unsigned char *row = gray + (size_t)i * stride;
    for (int e = threadIdx.x; e < 256 + n - 1; e += 256) {
        const int c = j0 + e - half;
        seg[e] = c < w + half ? lut[row[(c >= 0 && c < w) ? c : refl101(c, w)]] : 0.0;
    }
    __syncthreads();
    const int j = j0 + threadIdx.x;
    if (j >= w) return;
    double s = 0;
    for (int q = 0; q < n; ++q) s += k[q] * seg[threadIdx.x + q];
    tmp[(size_t)i * w + j] = s;
}

// normalisation table of an image: lut[g] = (g - min) / (max - min).  grid n_images x 256
__global__ __launch_bounds__(256) void k_norm_lut(const int *mm, double *lut)
{
    mm += (size_t)blockIdx.x * kMmSlots * kMmStride;
    int imn, imx;
    read_extremes(mm, imn, imx);
    const double mn = imn, mx = imx;
    lut[(size_t)blockIdx.x * 256 + threadIdx.x] = ((double)threadIdx.x - mn) / (mx - mn);
}

// The row pass for a compile-time tap count: a thread computes OPT consecutive outputs from NT + OPT - 1 LDS values
// (~8 LDS reads per output instead of NT) and takes the taps as scalar operands (constant address space: uniform,
// written by the host); a block walks kRowsPerBlock rows so that the table / tap set-up and the launch are amortised.
// Same order of additions per output.  grid (ceil(w/(256 OPT)), ceil(h/kRowsPerBlock), n_images)
constexpr int kRowsPerBlock = 4;
template <int NT, int OPT>
__global__ __launch_bounds__(256) void k_gauss_rows_n(const unsigned char *gray, int w, int h, int stride, const double *lut_g, const double *taps,
                                                      double *tmp, size_t plane)
{
    gray += (size_t)blockIdx.z * stride * h; lut_g += (size_t)blockIdx.z * 256; tmp += blockIdx.z * plane;
    constexpr int H = NT / 2, SEG = 256 * OPT + NT - 1;
    // element idx sits at idx + idx / 32 (bank spreading for the strided window reads)
    __shared__ double seg[SEG + SEG / 32 + 2], lut[256];
    lut[threadIdx.x] = lut_g[threadIdx.x];
    typedef const double __attribute__((address_space(4))) *cptr4;
    const cptr4 kt = (cptr4)taps;
    const int j0 = blockIdx.x * 256 * OPT, j = j0 + OPT * threadIdx.x;
    for (int rr = 0; rr < kRowsPerBlock; ++rr) {
        const int i = blockIdx.y * kRowsPerBlock + rr;
        if (i >= h) break;                                   // block-uniform
        const unsigned char *row = gray + (size_t)i * stride;
        __syncthreads();                                     // the previous row's windows have been read (and lut is there)
        for (int e = threadIdx.x; e < SEG; e += 256) {
            const int c = j0 + e - H;
            seg[e + (e >> 5)] = c < w + H ? lut[row[(c >= 0 && c < w) ? c : refl101(c, w)]] : 0.0;
        }
        __syncthreads();
        if (j >= w) continue;
        double win[NT + OPT - 1];
#pragma unroll
        for (int e = 0; e < NT + OPT - 1; ++e) { const int idx = OPT * threadIdx.x + e; win[e] = seg[idx + (idx >> 5)]; }
        double *dst = tmp + (size_t)i * w + j;
#pragma unroll
        for (int u = 0; u < OPT; ++u) {
            double sum = 0;
#pragma unroll
            for (int q = 0; q < NT; ++q) sum += kt[q] * win[u + q];
            if (j + u < w) dst[u] = sum;
        }
    }
}

// columns (symmetric kernel: centre tap, then pairs).  Generic version: grid (ceil(w/256), h, n_images)
__global__ __launch_bounds__(256) void k_gauss_cols(const double *tmp, int w, int h, const double *taps, int n, double *Ig, size_t plane)
{
    tmp += blockIdx.z * plane; Ig += blockIdx.z * plane;
    __shared__ double k[64];
    if (threadIdx.x < n) k[threadIdx.x] = taps[threadIdx.x];
    __syncthreads();
    const int i = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
    if (j >= w) return;
    const int half = n / 2;
    double s = k[half] * tmp[(size_t)i * w + j];
    for (int q = 1; q <= half; ++q) s += k[half + q] * (tmp[(size_t)refl101(i + q, h) * w + j] + tmp[(size_t)refl101(i - q, h) * w + j]);
    Ig[(size_t)i * w + j] = s;
}

// The same for a compile-time tap count: a thread owns one column of a strip of kStrip rows and keeps the
// kStrip + NT - 1 inputs in registers (2.75 loads per output at 29 taps instead of 57).  Same order of additions.
// grid (ceil(w/256), ceil(h/kStrip), n_images)
constexpr int kStrip = 16;
template <int NT>
__global__ __launch_bounds__(256) void k_gauss_cols_strip(const double *tmp, int w, int h, const double *taps, double *Ig, size_t plane)
{
    tmp += blockIdx.z * plane; Ig += blockIdx.z * plane;
    constexpr int H = NT / 2;
    __shared__ double k[NT];
    if (threadIdx.x < NT) k[threadIdx.x] = taps[threadIdx.x];
    __syncthreads();
    const int i0 = blockIdx.y * kStrip, j = blockIdx.x * 256 + threadIdx.x;
    if (j >= w) return;
    double win[kStrip + NT - 1];
#pragma unroll
    for (int e = 0; e < kStrip + NT - 1; ++e) {
        const int r = i0 + e - H;
        win[e] = r < h + H ? tmp[(size_t)refl101(r, h) * w + j] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < kStrip; ++u) {
        if (i0 + u >= h) break;
        double s = k[H] * win[u + H];
#pragma unroll
        for (int q = 1; q <= H; ++q) s += k[H + q] * (win[u + H + q] + win[u + H - q]);
        Ig[(size_t)(i0 + u) * w + j] = s;
    }
}

// secondDerivCornerMetric :108-141 fused: every intermediate plane (Ix, Iy, I_45, ...) is a reflected 3-tap stencil of
// the previous one, so each output pixel reads a 5x5 neighbourhood of Ig.  grid (ceil(w/256), h)
constexpr int kMetricRows = 8;
__global__ __launch_bounds__(256) void k_corner_metric(const double *Ig, int w, int h, int sigma, double c4, double cn4, double s4, double sn4,
                                                       double *metric, double *Ixy, size_t plane)
{
    Ig += blockIdx.z * plane; metric += blockIdx.z * plane; Ixy += blockIdx.z * plane;
    // a block computes kMetricRows rows of a 256-column segment: the (kMetricRows + 4) x (256 + 4) neighbourhood is
    // staged in LDS once (1.5 reads of Ig per output instead of 5); reflected indices of pixels in the block always
    // land inside this window
    __shared__ double tile[kMetricRows + 4][256 + 4];
    const int i0 = blockIdx.y * kMetricRows, j0 = blockIdx.x * 256;
    for (int e = threadIdx.x; e < (kMetricRows + 4) * 260; e += 256) {
        const int r = e / 260, c = e % 260, gi = i0 - 2 + r, gj = j0 - 2 + c;
        tile[r][c] = (gi >= 0 && gi < h && gj >= 0 && gj < w) ? Ig[(size_t)gi * w + gj] : 0.0;
    }
    __syncthreads();
    const int j = j0 + threadIdx.x;
    if (j >= w) return;
    auto G = [&](int r, int c) { return tile[r - (i0 - 2)][c - (j0 - 2)]; };
    for (int u = 0; u < kMetricRows; ++u) {
        const int i = i0 + u;
        if (i >= h) break;
        double ix, iy, i45, ixy, i45x, i45y;
        if (i >= 2 && i < h - 2 && j >= 2 && j < w - 2) {
            // interior: no reflection, fixed tile offsets (t(dr, dc) = Ig(i + dr, j + dc)); same operations in the same order
            const int tc = threadIdx.x + 2, tr = u + 2;
            auto t = [&](int dr, int dc) { return tile[tr + dr][tc + dc]; };
            auto ixo = [&](int dr, int dc) { return t(dr, dc - 1) - t(dr, dc + 1); };
            auto iyo = [&](int dr, int dc) { return t(dr - 1, dc) - t(dr + 1, dc); };
            auto i45o = [&](int dr, int dc) { return ixo(dr, dc) * c4 + iyo(dr, dc) * s4; };
            ix = ixo(0, 0); iy = iyo(0, 0); i45 = ix * c4 + iy * s4;
            ixy = ixo(-1, 0) - ixo(1, 0);
            i45x = i45o(0, -1) - i45o(0, 1);
            i45y = i45o(-1, 0) - i45o(1, 0);
        } else {
            auto IX = [&](int r, int c) { return G(r, refl101(c - 1, w)) - G(r, refl101(c + 1, w)); };       // du = (1 0 -1)
            auto IY = [&](int r, int c) { return G(refl101(r - 1, h), c) - G(refl101(r + 1, h), c); };
            auto I45 = [&](int r, int c) { return IX(r, c) * c4 + IY(r, c) * s4; };
            const int im = refl101(i - 1, h), ip = refl101(i + 1, h), jm = refl101(j - 1, w), jp = refl101(j + 1, w);
            ix = IX(i, j); iy = IY(i, j); i45 = ix * c4 + iy * s4;
            ixy = IX(im, j) - IX(ip, j);
            i45x = I45(i, jm) - I45(i, jp);
            i45y = I45(im, j) - I45(ip, j);
        }
        const double i4545 = i45x * cn4 + i45y * sn4;
        const double in45 = ix * cn4 + iy * sn4;
        double cxy = sigma * sigma * fabs(ixy) - 1.5 * sigma * (fabs(i45) + fabs(in45));
        if (cxy < 0) cxy = 0;
        double c45 = sigma * sigma * fabs(i4545) - 1.5 * sigma * (fabs(ix) + fabs(iy));
        if (c45 < 0) c45 = 0;
        metric[(size_t)i * w + j] = cxy + c45;
        Ixy[(size_t)i * w + j] = ixy;
    }
}

// nonMaximumSuppression :144-193, one thread per cell; cells are numbered column-major like the reference's loops
// (x outer, y inner).  cell[c] = (maxi << 16) | maxj, or -1.
__global__ __launch_bounds__(256) void k_nms_cells(const double *img, int width, int height, int ncx, int ncy, int *cell, size_t plane)
{
    img += blockIdx.y * plane; cell += (size_t)blockIdx.y * ncx * ncy;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= ncx * ncy) return;
    constexpr int n = kNmsN, margin = kNmsMargin;
    const int cxi = t % ncx, cyi = t / ncx;          // neighbouring threads: neighbouring cells of a row (coalescing) ...
    const int c = cxi * ncy + cyi;                   // ... stored in the reference's column-major cell order
    const int i = n + margin + cxi * (n + 1), j = n + margin + cyi * (n + 1);
    int maxi = i, maxj = j;
    double maxval = img[(size_t)j * width + i];
    for (int i2 = i; i2 <= i + n; ++i2)
        for (int j2 = j; j2 <= j + n; ++j2) {
            const double v = img[(size_t)j2 * width + i2];
            if (v > maxval) { maxi = i2; maxj = j2; maxval = v; }
        }
    if (!(maxval >= kNmsTau)) { cell[c] = -1; return; }      // (nearly all cells: the neighbourhood test cannot matter)
    bool failed = false;
    const int i_end = min(maxi + n, width - margin), j_end = min(maxj + n, height - margin);
    for (int i2 = maxi - n; i2 < i_end && !failed; ++i2)
        for (int j2 = maxj - n; j2 < j_end; ++j2) {
            const double v = img[(size_t)j2 * width + i2];
            if (v > maxval && (i2 < i || i2 > i + n || j2 < j || j2 > j + n)) { failed = true; break; }
        }
    cell[c] = (maxval >= kNmsTau && !failed) ? ((maxi << 16) | maxj) : -1;
}

// order-preserving compaction of the cell results: one 1024-thread workgroup, each of the 16 waves owns a contiguous
// range of cells and walks it 64 cells at a time (coalesced, eight loads in flight), positions by ballot + popcount.
// count[0] = number of maxima
__global__ __launch_bounds__(1024) void k_nms_compact(const int *cell, int ncell, int cap, int *cand, int *count)
{
    cell += (size_t)blockIdx.x * ncell; cand += (size_t)blockIdx.x * ncell; count += blockIdx.x;          // one workgroup per image
    __shared__ int wtot[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int per = ((ncell + 15) / 16 + 63) & ~63;           // cells per wave, whole 64-cell steps
    const int b = min(ncell, wave * per), e = min(ncell, b + per);
    int total = 0;
    for (int q0 = b; q0 < e; q0 += 512) {
        int v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int q = q0 + 64 * u + lane; v[u] = q < e ? cell[q] : -1; }
#pragma unroll
        for (int u = 0; u < 8; ++u) total += __popcll(__ballot(v[u] >= 0));
    }
    if (lane == 0) wtot[wave] = total;
    __syncthreads();
    int pos = 0;
    for (int k = 0; k < wave; ++k) pos += wtot[k];
    for (int q0 = b; q0 < e; q0 += 512) {
        int v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int q = q0 + 64 * u + lane; v[u] = q < e ? cell[q] : -1; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned long long m = __ballot(v[u] >= 0);
            if (v[u] >= 0) { const int p = pos + __popcll(m & ((1ull << lane) - 1)); if (p < cap) cand[p] = v[u]; }
            pos += __popcll(m);
        }
    }
    if (threadIdx.x == 1023) count[0] = pos;
}

__device__ __forceinline__ double wave_sum(double v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// block-wide sum of up to eight values (256 threads: wave butterflies, then the four wave totals in wave order)
template <int NV>
__device__ __forceinline__ void block_sum4(double (&v)[NV], double *sh)      // sh: 4 * NV doubles
{
#pragma unroll
    for (int m = 0; m < NV; ++m) v[m] = wave_sum(v[m]);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int m = 0; m < NV; ++m) sh[(threadIdx.x >> 6) * NV + m] = v[m];
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < NV; ++m) v[m] = (sh[m] + sh[NV + m]) + (sh[2 * NV + m] + sh[3 * NV + m]);
}

// cornerCorrelationScore :428-490 (+ createCorrelationPatch :351-389) for one radius, by the whole 256-thread block
template <int R>
__device__ __forceinline__ double score_radius(const unsigned char *gray, int stride, int width, int height, int cu, int cv, double gmn, double gmx,
                                               double v1x, double v1y, double v2x, double v2y, double s1a, double c1a, double s2a, double c2a,
                                               const double *npdf, double *sh)
{
    constexpr int n = 2 * R + 1, N = n * n, PER = (N + 255) / 256;
    double wv[PER], fv[PER];
    double acc2[2] = { 0, 0 };
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int e = threadIdx.x + 256 * u;
        wv[u] = 0; fv[u] = 0;
        if (e < N) {
            const int y = e / n, x = e % n;
            const double p0 = x - R, p1 = y - R;
            const double a = p0 * v1x + p1 * v1y, b = p0 * v2x + p1 * v2y;
            const double q0 = p0 - a * v1x, q1 = p1 - a * v1y, t0 = p0 - b * v2x, t1 = p1 - b * v2y;
            fv[u] = (sqrt(q0 * q0 + q1 * q1) <= 1.5 || sqrt(t0 * t0 + t1 * t1) <= 1.5) ? 1.0 : -1.0;
            double du, dv;
            pixel_gradient(gray, width, height, stride, cv - R + y, cu - R + x, du, dv);
            wv[u] = sqrt(dv * dv + du * du);
            acc2[0] += wv[u]; acc2[1] += fv[u];
        }
    }
    block_sum4<2>(acc2, sh);
    const double mw = acc2[0] / N, mf = acc2[1] / N;
    double var2[2] = { 0, 0 };
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        if (threadIdx.x + 256 * u < N) { const double dw = wv[u] - mw, df = fv[u] - mf; var2[0] += dw * dw; var2[1] += df * df; }
    }
    block_sum4<2>(var2, sh);
    const double sdw = sqrt(var2[0] / N), sdf = sqrt(var2[1] / N);
    double acc[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };      // correlation, 4 template sums, 4 template norms
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int e = threadIdx.x + 256 * u;
        if (e < N) {
            const int y = e / n, x = e % n;
            acc[0] += ((wv[u] - mw) / sdw) * ((fv[u] - mf) / sdf);
            const int du = x - R, dv = y - R;
            const double e1 = -du * s1a + dv * c1a, e2 = -du * s2a + dv * c2a;
            int which = -1;
            if (e1 <= -0.1 && e2 <= -0.1) which = 0;
            else if (e1 >= 0.1 && e2 >= 0.1) which = 1;
            else if (e1 <= -0.1 && e2 >= 0.1) which = 2;
            else if (e1 >= 0.1 && e2 <= -0.1) which = 3;
            if (which >= 0) {
                const double g = npdf[du * du + dv * dv];
                const double px = ((double)gray[(size_t)(cv - R + y) * stride + (cu - R + x)] - gmn) / (gmx - gmn);
#pragma unroll
                for (int m = 0; m < 4; ++m) if (m == which) { acc[5 + m] += g; acc[1 + m] += g * px; }
            }
        }
    }
    block_sum4<9>(acc, sh);
    const double score_gradient = fmax(acc[0] / (N - 1), 0.0);
    double tt[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) tt[m] = acc[5 + m] > 2.220446049250313e-16 ? acc[1 + m] / acc[5 + m] : 0.0;
    const double mu = (tt[0] + tt[1] + tt[2] + tt[3]) / 4;
    const double score_1 = fmin(fmin(tt[0] - mu, tt[1] - mu), fmin(mu - tt[2], mu - tt[3]));
    const double score_2 = fmin(fmin(mu - tt[0], mu - tt[1]), fmin(tt[2] - mu, tt[3] - mu));
    return score_gradient * fmax(fmax(score_1, score_2), 0.0);
}

// getOrientations + scoreCorners + subPixelLocation for one candidate per 256-thread workgroup
__global__ __launch_bounds__(256) void k_corner_describe(const unsigned char *gray, int stride, const int *mm, AtanConsts ac,
                                                         const double *Ixy, int width, int height, const int *cand, const DescribeTables *T,
                                                         double *out_v, double *out_score, double *out_sub, size_t plane, int ncell, const int *count)
{
    // blockIdx.y = image; candidates past this image's count (the grid is sized for the fullest image): nothing to do
    if ((int)blockIdx.x >= count[blockIdx.y]) return;
    gray += (size_t)blockIdx.y * stride * height; mm += (size_t)blockIdx.y * kMmSlots * kMmStride;
    Ixy += blockIdx.y * plane;
    cand += (size_t)blockIdx.y * ncell; out_v += (size_t)blockIdx.y * 4 * ncell; out_score += (size_t)blockIdx.y * ncell; out_sub += (size_t)blockIdx.y * 2 * ncell;
    __shared__ double hist[kBins], sm[kBins], mv[kBins], vsh[4], red[36];
    __shared__ int bsh[2], climb[kBins], mb[kBins];
    __shared__ double wwgt[(2 * kOrientR + 1) * (2 * kOrientR + 1)];
    __shared__ unsigned char wbin[(2 * kOrientR + 1) * (2 * kOrientR + 1)];
    const int tid = threadIdx.x, q = blockIdx.x;
    const int cu = cand[q] >> 16, cv = cand[q] & 0xffff;
    // ---- :200-279  orientation histogram ------------------------------------------------------------------------------
    {
        const int y1 = min(cv + kOrientR, height - 1), y0 = max(cv - kOrientR, 0), x1 = min(cu + kOrientR, width - 1), x0 = max(cu - kOrientR, 0);
        // the window (<= 21 x 21) is staged in LDS by all threads, then thread b < 32 owns bin b and adds its pixels in
        // the reference's order (row by row): the histogram is bit-identical to the sequential one
        const int ww = x1 - x0 + 1, wn = ww * (y1 - y0 + 1);
        for (int e = tid; e < wn; e += 256) {
            const int i = y0 + e / ww, j = x0 + e % ww;
            double du, dv;
            pixel_gradient(gray, width, height, stride, i, j, du, dv);
            double a = edge_angle(du, dv, ac) + kPi / 2;
            if (a > kPi) a -= kPi;
            int bin = (int)floor(a / (kPi / kBins));
            wbin[e] = (unsigned char)max(min(bin, kBins - 1), 0);
            wwgt[e] = sqrt(dv * dv + du * du);
        }
        __syncthreads();
        if (tid < kBins) {
            double hsum = 0;
            for (int e = 0; e < wn; ++e) if (wbin[e] == tid) hsum += wwgt[e];
            hist[tid] = hsum;
        }
        __syncthreads();
        // :286-349 (the histogram is circular: indices wrap, see the oracle's header for the reference's out-of-range
        // reads).  Smoothing and hill climbing: thread i does bin i; the list of distinct modes in discovery order and
        // its ordering: thread 0, on LDS arrays.
        if (tid < kBins) {
            double sum = 0;
            for (int j = -2; j <= 2; ++j) sum += hist[((tid + j) % kBins + kBins) % kBins] * T->smooth[j + 2];
            sm[tid] = sum;
        }
        __syncthreads();
        if (tid < kBins) {
            int j = tid;
            for (;;) {
                const double h0 = sm[j];
                const int j1 = (j + 1) % kBins, j2 = (j - 1 + kBins) % kBins;
                const double h1 = sm[j1], h2 = sm[j2];
                if (h1 >= h0 && h1 >= h2) j = j1;
                else if (h2 > h0 && h2 > h1) j = j2;
                else break;
            }
            climb[tid] = j;
        }
        __syncthreads();
        if (tid == 0) {
            bool flat = true;
            for (int i = 1; i < kBins; ++i) if (fabs(sm[i] - sm[0]) > 1e-5) { flat = false; break; }
            int nm = 0;
            if (!flat) {
                for (int i = 0; i < kBins; ++i) {
                    const int j = climb[i];
                    bool seen = false;
                    for (int k = 0; k < nm; ++k) if (mb[k] == j) { seen = true; break; }
                    if (!seen) { mb[nm] = j; mv[nm] = sm[j]; ++nm; }
                }
                for (int a = 1; a < nm; ++a) {          // strongest first, stable
                    const int b = mb[a]; const double v = mv[a];
                    int k = a - 1;
                    while (k >= 0 && mv[k] < v) { mb[k + 1] = mb[k]; mv[k + 1] = mv[k]; --k; }
                    mb[k + 1] = b; mv[k + 1] = v;
                }
            }
            int b1 = -1, b2 = -1;                      // histogram bins of v1 and v2
            if (nm > 1) {
                const double z0 = mb[0] * kPi / kBins, z1 = mb[1] * kPi / kBins;
                if (z0 > z1) {
                    b1 = mb[1]; b2 = mb[0];
                    if (fmin(z0 - z1, z1 + kPi - z0) <= 0.3 && nm > 2) b1 = mb[2];
                } else {
                    b1 = mb[0]; b2 = mb[1];
                    if (fmin(z1 - z0, z0 + kPi - z1) <= 0.3 && nm > 2) b2 = mb[2];
                }
            }
            bsh[0] = b1; bsh[1] = b2;
            vsh[0] = b1 >= 0 ? T->bcos[b1] : 0.0; vsh[1] = b1 >= 0 ? T->bsin[b1] : 0.0;
            vsh[2] = b2 >= 0 ? T->bcos[b2] : 0.0; vsh[3] = b2 >= 0 ? T->bsin[b2] : 0.0;
        }
        __syncthreads();
    }
    const double v1x = vsh[0], v1y = vsh[1], v2x = vsh[2], v2y = vsh[3];
    const int b1 = bsh[0], b2 = bsh[1];
    // template angles atan2(v.y, v.x) and their sin / cos (:483, :364-365); v = (0, 0): atan2 = 0
    const double s1a = b1 >= 0 ? T->tsin[b1] : 0.0, c1a = b1 >= 0 ? T->tcos[b1] : 1.0;
    const double s2a = b2 >= 0 ? T->tsin[b2] : 0.0, c2a = b2 >= 0 ? T->tcos[b2] : 1.0;
    // ---- :391-426  correlation score, best of the radii that fit (block-uniform conditions) ---------------------------
    int imn, imx;
    read_extremes(mm, imn, imx);
    const double gmn = imn, gmx = imx;
    auto fits = [&](int r) { return cu >= r && cu < width - r && cv >= r && cv < height - r; };
    double best = 0, sc = 0;
    if (fits(8)) sc = score_radius<8>(gray, stride, width, height, cu, cv, gmn, gmx, v1x, v1y, v2x, v2y, s1a, c1a, s2a, c2a, T->npdf[0], red);
    best = sc;
    sc = 0;
    if (fits(12)) sc = score_radius<12>(gray, stride, width, height, cu, cv, gmn, gmx, v1x, v1y, v2x, v2y, s1a, c1a, s2a, c2a, T->npdf[1], red);
    if (sc > best) best = sc;
    sc = 0;
    if (fits(16)) sc = score_radius<16>(gray, stride, width, height, cu, cv, gmn, gmx, v1x, v1y, v2x, v2y, s1a, c1a, s2a, c2a, T->npdf[2], red);
    if (sc > best) best = sc;
    // ---- :510-539  quadratic fit of the 5x5 neighbourhood of Ixy (thread a < 6: coefficient a, sums in the reference's order)
    double beta = 0;
    if (tid < 6) {
        int cnt = 0;
        for (int j = cu - 2; j <= cu + 2; ++j)
            for (int k = cv - 2; k <= cv + 2; ++k) beta += T->X[tid * 25 + cnt++] * Ixy[(size_t)k * width + j];
    }
    if (tid < 64) {
        const double A = __shfl(beta, 0), B = __shfl(beta, 1), C = __shfl(beta, 2), D = __shfl(beta, 3), E = __shfl(beta, 4);
        if (tid == 0) {
            double x = -(2 * B * C - D * E) / (4 * A * B - E * E);
            double y = -(2 * A * D - C * E) / (4 * A * B - E * E);
            if (fabs(x) > 2 || fabs(y) > 2) { x = 0; y = 0; }
            out_sub[2 * q] = cu + x; out_sub[2 * q + 1] = cv + y;
            out_score[q] = best;
            out_v[4 * q] = v1x; out_v[4 * q + 1] = v1y; out_v[4 * q + 2] = v2x; out_v[4 * q + 3] = v2y;
        }
    }
}

double normpdf_i(double x, int mu, int sigma) { return std::exp(-(x - mu) * (x - mu) / 2 / sigma / sigma) / std::sqrt(2 * kPi) / sigma; }

// Working set of one call (~70 MB of fp64 planes for 1280 x 1080): carved out of ONE grow-only arena per device that
// stays allocated between calls -- hipMalloc / hipFree of the planes cost 1.4 ms per image, six times the kernels.
// Calls on the same device are serialised by the arena's mutex.
struct Arena {
    std::mutex mu;
    void *base = nullptr;
    size_t bytes = 0;
};
Arena g_arena[16];

struct ArenaCursor {
    char *p; size_t left;
    template <typename T> T *take(size_t n)
    {
        const size_t b = ((n ? n : 1) * sizeof(T) + 255) & ~(size_t)255;
        if (b > left) return nullptr;
        T *r = reinterpret_cast<T *>(p); p += b; left -= b;
        return r;
    }
};

}  // namespace

extern "C" void tscm_corner_candidates_free(tscm_corner_candidates *c)
{
    if (!c) return;
    std::free(c->x); std::free(c->y); std::free(c->v1); std::free(c->v2); std::free(c->score); std::free(c->sub);
    c->x = c->y = c->v1 = c->v2 = c->score = c->sub = nullptr;
    c->n = 0;
}

extern "C" int tscm_detect_corners_batch(const unsigned char *const *images, int n_images, int width, int height, int stride, int sigma, double min_score,
                                         int device, tscm_corner_candidates *out)
{
    if (n_images < 0 || (n_images > 0 && (!out || !images))) return tscm_set_error(TSCM_E_INVALID, "NULL argument");
    if (n_images == 0) return 0;
    std::memset(out, 0, sizeof(*out) * (size_t)n_images);
    for (int q = 0; q < n_images; ++q) if (!images[q]) return tscm_set_error(TSCM_E_INVALID, "NULL image");
    if (width < 1 || height < 1 || stride < width) return tscm_set_error(TSCM_E_INVALID, "bad image description");
    if (n_images > 65535) return tscm_set_error(TSCM_E_UNSUPPORTED, "more than 65535 images per batch");
    if (width > 32767 || height > 32767) return tscm_set_error(TSCM_E_UNSUPPORTED, "images beyond 32767 pixels per side");
    const int ntap = 7 * sigma + 1;
    if (sigma < 1 || ntap % 2 == 0 || ntap > 64) return tscm_set_error(TSCM_E_UNSUPPORTED, "sigma must be even and at most 8 (cv::GaussianBlur needs an odd 7 sigma + 1; the reference uses 4)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return tscm_set_error(TSCM_E_NO_DEVICE, "no HIP device available (the corner detector has no CPU fallback)");
    if (device < 0 || device >= ndev) return tscm_set_error(TSCM_E_NO_DEVICE, "device index out of range");
    CRN_TRY(hipSetDevice(device));

    // ---- host-side constants (the C library the CPU path would use) -----------------------------------------------
    std::vector<double> taps(ntap);
    {
        const double scale2x = -0.5 / ((double)sigma * sigma);
        double sum = 0;
        for (int i = 0; i < ntap; ++i) { const double x = i - (ntap - 1) * 0.5; taps[i] = std::exp(scale2x * x * x); sum += taps[i]; }
        sum = 1. / sum;
        for (int i = 0; i < ntap; ++i) taps[i] *= sum;
    }
    const AtanConsts ac = { std::atan2(1.0, 1.0), std::atan2(1.0, -1.0), std::atan2(-1.0, 1.0), std::atan2(-1.0, -1.0),
                            std::atan2(1.0, 0.0), std::atan2(-1.0, 0.0), std::atan2(0.0, -1.0) };
    std::vector<DescribeTables> tab(1);
    DescribeTables &T = tab[0];
    for (int j = -2; j <= 2; ++j) T.smooth[j + 2] = normpdf_i(j, 0, 1);
    for (int b = 0; b < kBins; ++b) {
        const double z = b * kPi / kBins;
        T.bcos[b] = std::cos(z); T.bsin[b] = std::sin(z);
        const double a = std::atan2(T.bsin[b], T.bcos[b]);
        T.tsin[b] = std::sin(a); T.tcos[b] = std::cos(a);
    }
    for (int k = 0; k < 3; ++k) {
        const int r = 8 + 4 * k;
        for (int d2 = 0; d2 <= 512; ++d2) T.npdf[k][d2] = d2 <= 2 * r * r ? normpdf_i(std::sqrt((double)d2), 0, r / 2) : 0.0;
    }
    {   // X = (A^T A)^-1 A^T, Gauss-Jordan with partial pivoting
        double A[25][6], M[6][12];
        for (int y = -2; y <= 2; ++y)
            for (int x = -2; x <= 2; ++x) {
                const int idx = (x + 2) * 5 + y + 2;
                A[idx][0] = x * x; A[idx][1] = y * y; A[idx][2] = x; A[idx][3] = y; A[idx][4] = x * y; A[idx][5] = 1;
            }
        for (int a = 0; a < 6; ++a)
            for (int b = 0; b < 6; ++b) {
                double s = 0;
                for (int q = 0; q < 25; ++q) s += A[q][a] * A[q][b];
                M[a][b] = s; M[a][6 + b] = a == b ? 1.0 : 0.0;
            }
        for (int c = 0; c < 6; ++c) {
            int p = c;
            for (int q = c + 1; q < 6; ++q) if (std::fabs(M[q][c]) > std::fabs(M[p][c])) p = q;
            if (p != c) for (int q = 0; q < 12; ++q) std::swap(M[c][q], M[p][q]);
            const double d = M[c][c];
            for (int q = 0; q < 12; ++q) M[c][q] /= d;
            for (int rr = 0; rr < 6; ++rr) {
                if (rr == c) continue;
                const double f = M[rr][c];
                for (int q = 0; q < 12; ++q) M[rr][q] -= f * M[c][q];
            }
        }
        for (int a = 0; a < 6; ++a)
            for (int q = 0; q < 25; ++q) {
                double s = 0;
                for (int b = 0; b < 6; ++b) s += M[a][6 + b] * A[q][b];
                T.X[a * 25 + q] = s;
            }
    }

    // ---- device buffers --------------------------------------------------------------------------------------------
    const size_t N = (size_t)width * height;
    constexpr int n = kNmsN, margin = kNmsMargin;
    const int span_x = width - 2 * (n + margin), span_y = height - 2 * (n + margin);
    const int ncx = span_x > 0 ? (span_x + n) / (n + 1) : 0, ncy = span_y > 0 ? (span_y + n) / (n + 1) : 0;
    const int ncell = ncx * ncy;
    if (device >= 16) return tscm_set_error(TSCM_E_UNSUPPORTED, "device index beyond 15");
    Arena &arena = g_arena[device];
    std::lock_guard<std::mutex> lock(arena.mu);
    const size_t B = (size_t)n_images;
    const size_t cells_n = (size_t)(ncell > 0 ? ncell : 1) * B;
    const size_t gbytes = (size_t)stride * height;
    const size_t need = 256 * 26 + B * gbytes + 4 * B * N * sizeof(double) + 64 * sizeof(double) + B * 256 * sizeof(double) + B * kMmSlots * kMmStride * sizeof(int)
                      + 2 * cells_n * sizeof(int) + B * sizeof(int) + sizeof(DescribeTables) + 7 * cells_n * sizeof(double);
    if (arena.bytes < need) {
        if (arena.base) { (void)hipFree(arena.base); arena.base = nullptr; arena.bytes = 0; }
        CRN_TRY(hipMalloc(&arena.base, need));
        arena.bytes = need;
    }
    ArenaCursor cur = { static_cast<char *>(arena.base), arena.bytes };
    struct { unsigned char *p; } d_gray = { cur.take<unsigned char>(B * gbytes) };
    struct { double *p; } d_tmp = { cur.take<double>(B * N) }, d_Ig = { cur.take<double>(B * N) },
                          d_metric = { cur.take<double>(B * N) }, d_Ixy = { cur.take<double>(B * N) }, d_taps = { cur.take<double>(64) }, d_lut = { cur.take<double>(B * 256) },
                          d_v = { cur.take<double>(4 * cells_n) }, d_score = { cur.take<double>(cells_n) }, d_sub = { cur.take<double>(2 * cells_n) };
    struct { int *p; } d_mm = { cur.take<int>(B * kMmSlots * kMmStride) }, d_cell = { cur.take<int>(cells_n) }, d_cand = { cur.take<int>(cells_n) },
                       d_count = { cur.take<int>(B) };
    struct { DescribeTables *p; } d_tab = { cur.take<DescribeTables>(1) };
    if (!d_tab.p || !d_count.p || !d_sub.p) return tscm_set_error(TSCM_E_HIP, "internal error: arena too small");
    // a strided view (cv::Mat ROI, numpy column slice) only guarantees (height - 1) * stride + width bytes behind its pointer
    const size_t host_bytes = (size_t)(height - 1) * (size_t)stride + (size_t)width;
    for (size_t q = 0; q < B; ++q) CRN_TRY(hipMemcpy(d_gray.p + q * gbytes, images[q], std::min(gbytes, host_bytes), hipMemcpyHostToDevice));
    CRN_TRY(hipMemcpy(d_taps.p, taps.data(), sizeof(double) * ntap, hipMemcpyHostToDevice));
    CRN_TRY(hipMemcpy(d_tab.p, tab.data(), sizeof(DescribeTables), hipMemcpyHostToDevice));
    {
        std::vector<int> mm0(B * kMmSlots * kMmStride, 0);
        for (size_t q = 0; q < B * kMmSlots; ++q) { mm0[q * kMmStride] = 255; mm0[q * kMmStride + 1] = 0; }
        CRN_TRY(hipMemcpy(d_mm.p, mm0.data(), sizeof(int) * mm0.size(), hipMemcpyHostToDevice));
    }
    CRN_TRY(hipMemset(d_count.p, 0, sizeof(int) * B));

    hipEvent_t e0, e1;
    CRN_TRY(hipEventCreate(&e0)); CRN_TRY(hipEventCreate(&e1));
    CRN_TRY(hipEventRecord(e0, nullptr));
    const dim3 grid2((width + 255) / 256, height, n_images);
    if (stride == width)
        hipLaunchKernelGGL(k_grey_extremes_flat, dim3((unsigned)((gbytes + 16383) / 16384), n_images), dim3(256), 0, nullptr, d_gray.p, gbytes, d_mm.p);
    else
        hipLaunchKernelGGL(k_grey_extremes, dim3((width + 4095) / 4096, height, n_images), dim3(256), 0, nullptr, d_gray.p, width, height, stride, d_mm.p);
    if (ntap == 29) {
        hipLaunchKernelGGL(k_norm_lut, dim3(n_images), dim3(256), 0, nullptr, d_mm.p, d_lut.p);
        const int rows_y = (height + kRowsPerBlock - 1) / kRowsPerBlock;
        if (width > 1024 && width <= 1280)       // one block per row at the reference's image width
            hipLaunchKernelGGL((k_gauss_rows_n<29, 5>), dim3(1, rows_y, n_images), dim3(256), 0, nullptr, d_gray.p, width, height, stride, d_lut.p, d_taps.p, d_tmp.p, N);
        else
            hipLaunchKernelGGL((k_gauss_rows_n<29, 4>), dim3((width + 1023) / 1024, rows_y, n_images), dim3(256), 0, nullptr, d_gray.p, width, height, stride, d_lut.p,
                               d_taps.p, d_tmp.p, N);
    } else
        hipLaunchKernelGGL(k_gauss_rows, grid2, dim3(256), 0, nullptr, d_gray.p, width, height, stride, d_mm.p, d_taps.p, ntap, d_tmp.p, N);
    if (ntap == 29)
        hipLaunchKernelGGL(k_gauss_cols_strip<29>, dim3((width + 255) / 256, (height + kStrip - 1) / kStrip, n_images), dim3(256), 0, nullptr, d_tmp.p, width, height,
                           d_taps.p, d_Ig.p, N);
    else
        hipLaunchKernelGGL(k_gauss_cols, grid2, dim3(256), 0, nullptr, d_tmp.p, width, height, d_taps.p, ntap, d_Ig.p, N);
    hipLaunchKernelGGL(k_corner_metric, dim3((width + 255) / 256, (height + kMetricRows - 1) / kMetricRows, n_images), dim3(256), 0, nullptr, d_Ig.p, width, height, sigma, std::cos(kPi / 4), std::cos(-kPi / 4), std::sin(kPi / 4),
                       std::sin(-kPi / 4), d_metric.p, d_Ixy.p, N);
    std::vector<int> counts(B, 0);
    int n_top = 0;
    if (ncell > 0) {
        hipLaunchKernelGGL(k_nms_cells, dim3((ncell + 255) / 256, n_images), dim3(256), 0, nullptr, d_metric.p, width, height, ncx, ncy, d_cell.p, N);
        hipLaunchKernelGGL(k_nms_compact, dim3(n_images), dim3(1024), 0, nullptr, d_cell.p, ncell, ncell, d_cand.p, d_count.p);
        CRN_TRY(hipMemcpy(counts.data(), d_count.p, sizeof(int) * B, hipMemcpyDeviceToHost));
        for (int c : counts) n_top = std::max(n_top, c);
        if (n_top > 0)
            hipLaunchKernelGGL(k_corner_describe, dim3(n_top, n_images), dim3(256), 0, nullptr, d_gray.p, stride, d_mm.p, ac, d_Ixy.p, width, height,
                               d_cand.p, d_tab.p, d_v.p, d_score.p, d_sub.p, N, ncell, d_count.p);
    }
    CRN_TRY(hipEventRecord(e1, nullptr));
    CRN_TRY(hipEventSynchronize(e1));
    CRN_TRY(hipGetLastError());
    float ms = 0;
    CRN_TRY(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    // ---- score filter (findCorner.cpp:47-66) per image, order preserved ---------------------------------------------
    for (size_t b = 0; b < B; ++b) {
        tscm_corner_candidates *o = out + b;
        const int n_max = counts[b];
        o->seconds = ms * 1e-3 / (double)B;          // share of the batch's device time
        o->n_maxima = n_max;
        std::vector<int> cand(n_max);
        std::vector<double> v(4 * (size_t)n_max), score(n_max), sub(2 * (size_t)n_max);
        if (n_max > 0) {
            CRN_TRY(hipMemcpy(cand.data(), d_cand.p + b * ncell, sizeof(int) * n_max, hipMemcpyDeviceToHost));
            CRN_TRY(hipMemcpy(v.data(), d_v.p + b * 4 * ncell, sizeof(double) * 4 * n_max, hipMemcpyDeviceToHost));
            CRN_TRY(hipMemcpy(score.data(), d_score.p + b * ncell, sizeof(double) * n_max, hipMemcpyDeviceToHost));
            CRN_TRY(hipMemcpy(sub.data(), d_sub.p + b * 2 * ncell, sizeof(double) * 2 * n_max, hipMemcpyDeviceToHost));
        }
        int keep = 0;
        for (int q = 0; q < n_max; ++q) if (!(score[q] < min_score)) ++keep;
        const size_t kk = keep ? keep : 1;
        o->x = static_cast<double *>(std::calloc(kk, sizeof(double))); o->y = static_cast<double *>(std::calloc(kk, sizeof(double)));
        o->v1 = static_cast<double *>(std::calloc(2 * kk, sizeof(double))); o->v2 = static_cast<double *>(std::calloc(2 * kk, sizeof(double)));
        o->score = static_cast<double *>(std::calloc(kk, sizeof(double))); o->sub = static_cast<double *>(std::calloc(2 * kk, sizeof(double)));
        if (!o->x || !o->y || !o->v1 || !o->v2 || !o->score || !o->sub) {
            for (size_t q = 0; q <= b; ++q) tscm_corner_candidates_free(out + q);
            return tscm_set_error(TSCM_E_NOMEM, "out of memory");
        }
        int w = 0;
        for (int q = 0; q < n_max; ++q) {
            if (score[q] < min_score) continue;
            o->x[w] = cand[q] >> 16; o->y[w] = cand[q] & 0xffff;
            o->v1[2 * w] = v[4 * q]; o->v1[2 * w + 1] = v[4 * q + 1]; o->v2[2 * w] = v[4 * q + 2]; o->v2[2 * w + 1] = v[4 * q + 3];
            o->score[w] = score[q];
            o->sub[2 * w] = sub[2 * q]; o->sub[2 * w + 1] = sub[2 * q + 1];
            ++w;
        }
        o->n = keep;
    }
    return 0;
}

extern "C" int tscm_detect_corners(const unsigned char *gray, int width, int height, int stride, int sigma, double min_score, int device,
                                   tscm_corner_candidates *out)
{
    if (!out) return tscm_set_error(TSCM_E_INVALID, "NULL argument");
    std::memset(out, 0, sizeof(*out));
    if (!gray) return tscm_set_error(TSCM_E_INVALID, "bad image description");
    return tscm_detect_corners_batch(&gray, 1, width, height, stride, sigma, min_score, device, out);
}

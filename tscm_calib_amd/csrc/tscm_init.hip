// tscm_init.hip -- focal-length initialisation on the device (mono part):
// TripleSphereCamera::estimate_focal (TS.cpp:110-168).  One thread per (image, board row): the
// width x 4 circle-fit design matrix is reduced to its 4 x 4 triangular factor by Householder
// reflections in thread-private memory, and the null vector (cv::SVD::solveZ) comes from a
// one-sided Jacobi SVD of that factor held in registers.  The accepted samples are averaged on the
// host in the reference's (image, row) order.
#include "tscm/tscm.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <string>
#include <vector>

int tscm_set_error(int code, const std::string &msg);   // tscm_solver.hip

#define INIT_TRY(expr)                                                                              \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) return tscm_set_error(TSCM_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

namespace {

constexpr int kMaxBoardW = 32;

// gamma of one board row, or a negative marker: -1 image without board, -2 rejected (TS.cpp:149, :153)
__global__ __launch_bounds__(64) void k_focal_rows(const double *__restrict__ pu, const double *__restrict__ pv, const int *__restrict__ count,
                                                   int n_views, int width, int height, double cx, double cy, double *__restrict__ gamma)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_views * height) return;
    const int k = t / height, i = t - k * height;
    if (count[k] == 0) { gamma[t] = -1.0; return; }
    double A[kMaxBoardW][4];
    const size_t base = ((size_t)k * height + i) * width;
    for (int j = 0; j < width; ++j) {
        const double x = pu[base + j] - cx, y = pv[base + j] - cy;
        A[j][0] = x; A[j][1] = y; A[j][2] = 0.5; A[j][3] = -0.5 * (x * x + y * y);
    }
    // Householder QR: A -> R (upper triangular, rows 0..3); Q is not needed for a null vector
    for (int c = 0; c < 4; ++c) {
        double s = 0.0;
        for (int r = c; r < width; ++r) s += A[r][c] * A[r][c];
        const double nrm = sqrt(s);
        if (nrm == 0.0) continue;
        const double alpha = A[c][c] > 0 ? -nrm : nrm;
        const double v0 = A[c][c] - alpha;
        const double vtv = s - A[c][c] * A[c][c] + v0 * v0;
        if (vtv == 0.0) continue;
        for (int cc = c + 1; cc < 4; ++cc) {
            double dot = v0 * A[c][cc];
            for (int r = c + 1; r < width; ++r) dot += A[r][c] * A[r][cc];
            const double f = 2.0 * dot / vtv;
            A[c][cc] -= f * v0;
            for (int r = c + 1; r < width; ++r) A[r][cc] -= f * A[r][c];
        }
        A[c][c] = alpha;                                     // the entries below (the reflector) are not read again
    }
    // one-sided Jacobi on the columns of R, V accumulates the right singular vectors
    double R[4][4], V[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) { R[r][c] = c >= r ? A[r][c] : 0.0; V[r][c] = r == c ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = false;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int q = p + 1; q < 4; ++q) {
                double a = 0, b = 0, g = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) { a += R[r][p] * R[r][p]; b += R[r][q] * R[r][q]; g += R[r][p] * R[r][q]; }
                if (g == 0.0 || fabs(g) <= 1e-17 * sqrt(a * b)) continue;
                rotated = true;
                const double zeta = (b - a) / (2.0 * g);
                const double tt = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double cs = 1.0 / sqrt(1.0 + tt * tt), sn = cs * tt;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double x = R[r][p], y = R[r][q];
                    R[r][p] = cs * x - sn * y; R[r][q] = sn * x + cs * y;
                    const double vx = V[r][p], vy = V[r][q];
                    V[r][p] = cs * vx - sn * vy; V[r][q] = sn * vx + cs * vy;
                }
            }
        if (!rotated) break;
    }
    double smin = INFINITY, c1 = 0, c2 = 0, c3 = 0, c4 = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        double s = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) s += R[r][j] * R[r][j];
        if (s < smin) { smin = s; c1 = V[0][j]; c2 = V[1][j]; c3 = V[2][j]; c4 = V[3][j]; }
    }
    const double tq = c1 * c1 + c2 * c2 + c3 * c4;                 // TS.cpp:148-156
    if (tq < 0) { gamma[t] = -2.0; return; }
    const double d = sqrt(1 / tq);
    const double nx = c1 * d, ny = c2 * d;
    if (nx * nx + ny * ny > 0.95) { gamma[t] = -2.0; return; }
    const double nz = sqrt(1 - nx * nx - ny * ny);
    gamma[t] = fabs(c3 * d / nz);
}

}  // namespace

extern "C" int tscm_estimate_focal(const double *pix_u, const double *pix_v, const int *count, int n_views, int board_w, int board_h,
                                   double cx, double cy, int device, double *focal, int *n_used)
{
    if (!focal || !n_used || n_views < 0 || (n_views > 0 && (!pix_u || !pix_v || !count))) return tscm_set_error(TSCM_E_INVALID, "NULL argument");
    if (board_w < 4 || board_h < 1) return tscm_set_error(TSCM_E_UNSUPPORTED, "estimate_focal needs boards at least 4 corners wide");
    if (board_w > kMaxBoardW) return tscm_set_error(TSCM_E_UNSUPPORTED, "boards wider than 32 corners");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return tscm_set_error(TSCM_E_NO_DEVICE, "no HIP device available (tscm_estimate_focal has no CPU fallback)");
    if (device < 0 || device >= ndev) return tscm_set_error(TSCM_E_NO_DEVICE, "device index out of range");
    INIT_TRY(hipSetDevice(device));
    *focal = 0.0; *n_used = 0;
    const size_t rows = (size_t)n_views * board_h, npix = rows * board_w;
    if (rows == 0) return 0;
    double *d_u = nullptr, *d_v = nullptr, *d_g = nullptr;
    int *d_c = nullptr;
    std::vector<double> g(rows);
    int rc = 0;
    auto body = [&]() -> int {
        INIT_TRY(hipMalloc(reinterpret_cast<void **>(&d_u), npix * sizeof(double)));
        INIT_TRY(hipMalloc(reinterpret_cast<void **>(&d_v), npix * sizeof(double)));
        INIT_TRY(hipMalloc(reinterpret_cast<void **>(&d_g), rows * sizeof(double)));
        INIT_TRY(hipMalloc(reinterpret_cast<void **>(&d_c), (size_t)n_views * sizeof(int)));
        INIT_TRY(hipMemcpy(d_u, pix_u, npix * sizeof(double), hipMemcpyHostToDevice));
        INIT_TRY(hipMemcpy(d_v, pix_v, npix * sizeof(double), hipMemcpyHostToDevice));
        INIT_TRY(hipMemcpy(d_c, count, (size_t)n_views * sizeof(int), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_focal_rows, dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, 0, d_u, d_v, d_c, n_views, board_w, board_h, cx, cy, d_g);
        INIT_TRY(hipGetLastError());
        INIT_TRY(hipMemcpy(g.data(), d_g, rows * sizeof(double), hipMemcpyDeviceToHost));
        return 0;
    };
    rc = body();
    (void)hipFree(d_u); (void)hipFree(d_v); (void)hipFree(d_g); (void)hipFree(d_c);
    if (rc) return rc;
    double f = 0.0;
    int total = 0;
    for (size_t r = 0; r < rows; ++r) {                     // focal_ += gamma in (image, row) order (:155-156)
        if (g[r] < 0.0) continue;                           // markers; NaN samples are summed like the reference does
        f += g[r]; ++total;
    }
    if (total > 0) f /= total;
    *focal = f; *n_used = total;
    return 0;
}

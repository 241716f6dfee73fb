// tscm_init.hip -- focal-length initialisation on the device (mono part):
// TripleSphereCamera::estimate_focal (TS.cpp:110-168).  One thread per (image, board row): the
// width x 4 circle-fit design matrix is reduced to its 4 x 4 triangular factor by Householder
// reflections in thread-private memory, and the null vector (cv::SVD::solveZ) comes from a
// one-sided Jacobi SVD of that factor held in registers.  The accepted samples are averaged on the
// host in the reference's (image, row) order.
#include "tscm/tscm.h"

#include <hip/hip_runtime.h>

#include "tscm_math.h"

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

// ---------------------------------------------------------------------------------------------------
// estimate_extrinsic (TS.cpp:170-203), one thread per image.  cv::solvePnPRansac (external, randomised)
// is replaced by the deterministic planar PnP its iterative method performs on an all-inlier set:
// DLT homography (Hartley-normalised board points, h33 = 1, 8x8 normal equations), pose from the
// columns, polar orthonormalisation, Gauss-Newton on the 6 pose parameters with the analytic Jacobian.
// ---------------------------------------------------------------------------------------------------
template <int NN>
__device__ bool chol_solve_dev(double *A, double *b)
{
    for (int j = 0; j < NN; ++j) {
        double d = A[j * NN + j];
        for (int k = 0; k < j; ++k) d -= A[j * NN + k] * A[j * NN + k];
        if (!(d > 0.0)) return false;
        d = sqrt(d);
        A[j * NN + j] = d;
        for (int i = j + 1; i < NN; ++i) {
            double s = A[i * NN + j];
            for (int k = 0; k < j; ++k) s -= A[i * NN + k] * A[j * NN + k];
            A[i * NN + j] = s / d;
        }
    }
    for (int i = 0; i < NN; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= A[i * NN + k] * b[k]; b[i] = s / A[i * NN + i]; }
    for (int i = NN - 1; i >= 0; --i) { double s = b[i]; for (int k = i + 1; k < NN; ++k) s -= A[k * NN + i] * b[k]; b[i] = s / A[i * NN + i]; }
    return true;
}

// orthogonal polar factor by Newton iteration X <- (X + X^-T) / 2, then the angle-axis vector (cv::Rodrigues)
__device__ void rotation_vector_dev(const double *Min, double *rv)
{
    double X[9];
    for (int i = 0; i < 9; ++i) X[i] = Min[i];
    for (int it = 0; it < 50; ++it) {
        const double c00 = X[4] * X[8] - X[5] * X[7], c01 = X[5] * X[6] - X[3] * X[8], c02 = X[3] * X[7] - X[4] * X[6];
        const double c10 = X[2] * X[7] - X[1] * X[8], c11 = X[0] * X[8] - X[2] * X[6], c12 = X[1] * X[6] - X[0] * X[7];
        const double c20 = X[1] * X[5] - X[2] * X[4], c21 = X[2] * X[3] - X[0] * X[5], c22 = X[0] * X[4] - X[1] * X[3];
        const double det = X[0] * c00 + X[1] * c01 + X[2] * c02;
        const double cof[9] = { c00, c01, c02, c10, c11, c12, c20, c21, c22 };
        double diff = 0.0;
        for (int i = 0; i < 9; ++i) { const double y = 0.5 * (X[i] + cof[i] / det); diff = fmax(diff, fabs(y - X[i])); X[i] = y; }
        if (diff < 1e-16) break;
    }
    double rx = X[7] - X[5], ry = X[2] - X[6], rz = X[3] - X[1];
    const double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (X[0] + X[4] + X[8] - 1.0) * 0.5;
    c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
    double theta = acos(c);
    if (s < 1e-5) {
        if (c > 0) { rv[0] = rv[1] = rv[2] = 0.0; return; }
        double t = (X[0] + 1.0) * 0.5;
        rx = sqrt(fmax(t, 0.0));
        t = (X[4] + 1.0) * 0.5;
        ry = sqrt(fmax(t, 0.0)) * (X[1] < 0 ? -1.0 : 1.0);
        t = (X[8] + 1.0) * 0.5;
        rz = sqrt(fmax(t, 0.0)) * (X[2] < 0 ? -1.0 : 1.0);
        if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && ((X[5] > 0) != (ry * rz > 0))) rz = -rz;
        theta /= sqrt(rx * rx + ry * ry + rz * rz);
        rv[0] = rx * theta; rv[1] = ry * theta; rv[2] = rz * theta;
    } else {
        const double vth = theta / (2.0 * s);
        rv[0] = rx * vth; rv[1] = ry * vth; rv[2] = rz * vth;
    }
}

__global__ __launch_bounds__(64) void k_estimate_extrinsic(const double *__restrict__ intr, const double *__restrict__ pu, const double *__restrict__ pv,
                                                           const int *__restrict__ count, int n_views, const double *__restrict__ worlds, int n, int board_w,
                                                           double *__restrict__ Rt_out, unsigned char *__restrict__ ok_out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_views) return;
    ok_out[k] = 0;
    if (count[k] == 0) return;
    double I[9];
    for (int i = 0; i < 9; ++i) I[i] = intr[i];
    const double *u = pu + (size_t)k * n, *v = pv + (size_t)k * n;
    // transform = R2 * R1 turning the camera towards the board (:175-187)
    double p[3];
    const int ref = n / 2 - board_w / 2 - 1;
    tscm::unproject_pixel(I, u[ref], v[ref], p);
    const double alpha = atan2(p[0], p[2]), beta = asin(p[1]);
    const double ca = cos(alpha), sa = sin(alpha), cb = cos(beta), sb = sin(beta);
    const double T[9] = { ca, 0.0, -sa, -sb * sa, cb, -sb * ca, cb * sa, sb, cb * ca };       // R2 * R1
    auto normalised = [&](int i, double &x, double &y) {
        double q[3];
        tscm::unproject_pixel(I, u[i], v[i], q);
        const double X = T[0] * q[0] + T[1] * q[1] + T[2] * q[2], Y = T[3] * q[0] + T[4] * q[1] + T[5] * q[2], Z = T[6] * q[0] + T[7] * q[1] + T[8] * q[2];
        x = X / Z; y = Y / Z;
    };
    // Hartley normalisation of the board points
    double cx = 0, cy = 0, md = 0;
    for (int i = 0; i < n; ++i) { cx += worlds[3 * i]; cy += worlds[3 * i + 1]; }
    cx /= n; cy /= n;
    for (int i = 0; i < n; ++i) { const double dx = worlds[3 * i] - cx, dy = worlds[3 * i + 1] - cy; md += sqrt(dx * dx + dy * dy); }
    md /= n;
    if (!(md > 0.0)) return;
    const double s = sqrt(2.0) / md;
    // DLT: 8x8 normal equations
    double A[64], b[8];
    for (int i = 0; i < 64; ++i) A[i] = 0.0;
    for (int i = 0; i < 8; ++i) b[i] = 0.0;
    for (int i = 0; i < n; ++i) {
        double x, y;
        normalised(i, x, y);
        const double X = (worlds[3 * i] - cx) * s, Y = (worlds[3 * i + 1] - cy) * s;
        const double r1[8] = { X, Y, 1, 0, 0, 0, -x * X, -x * Y }, r2[8] = { 0, 0, 0, X, Y, 1, -y * X, -y * Y };
        for (int a = 0; a < 8; ++a) {
            for (int c = 0; c < 8; ++c) A[8 * a + c] += r1[a] * r1[c] + r2[a] * r2[c];
            b[a] += r1[a] * x + r2[a] * y;
        }
    }
    if (!chol_solve_dev<8>(A, b)) return;
    const double Hn[9] = { b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7], 1.0 };
    double H[9];
    for (int r = 0; r < 3; ++r) {
        H[3 * r] = Hn[3 * r] * s; H[3 * r + 1] = Hn[3 * r + 1] * s;
        H[3 * r + 2] = Hn[3 * r + 2] - s * (Hn[3 * r] * cx + Hn[3 * r + 1] * cy);
    }
    const double n1 = sqrt(H[0] * H[0] + H[3] * H[3] + H[6] * H[6]), n2 = sqrt(H[1] * H[1] + H[4] * H[4] + H[7] * H[7]);
    if (!(n1 > 0.0) || !(n2 > 0.0)) return;
    double lam = 2.0 / (n1 + n2);
    if (H[8] < 0) lam = -lam;
    double M[9], t[3], rv[3];
    for (int r = 0; r < 3; ++r) { M[3 * r] = lam * H[3 * r]; M[3 * r + 1] = lam * H[3 * r + 1]; t[r] = lam * H[3 * r + 2]; }
    M[2] = M[3] * M[7] - M[6] * M[4]; M[5] = M[6] * M[1] - M[0] * M[7]; M[8] = M[0] * M[4] - M[3] * M[1];
    rotation_vector_dev(M, rv);
    // Gauss-Newton on (rv, t), analytic Jacobian
    double R[9], dR[27];
    for (int it = 0; it < 10; ++it) {
        tscm::rotation_and_derivatives(rv, R, dR);
        double JtJ[36], Jtr[6];
        for (int i = 0; i < 36; ++i) JtJ[i] = 0.0;
        for (int i = 0; i < 6; ++i) Jtr[i] = 0.0;
        for (int i = 0; i < n; ++i) {
            double x, y;
            normalised(i, x, y);
            const double wx = worlds[3 * i], wy = worlds[3 * i + 1];
            const double PX = R[0] * wx + R[1] * wy + t[0], PY = R[3] * wx + R[4] * wy + t[1], PZ = R[6] * wx + R[7] * wy + t[2];
            const double iz = 1.0 / PZ, rx = PX * iz - x, ry = PY * iz - y;
            double J[2][6];
            for (int q = 0; q < 3; ++q) {                                   // d P / d w_q = dR_q (wx, wy, 0)
                const double dX = dR[9 * q] * wx + dR[9 * q + 1] * wy, dY = dR[9 * q + 3] * wx + dR[9 * q + 4] * wy, dZ = dR[9 * q + 6] * wx + dR[9 * q + 7] * wy;
                J[0][q] = (dX - PX * iz * dZ) * iz; J[1][q] = (dY - PY * iz * dZ) * iz;
            }
            J[0][3] = iz; J[0][4] = 0.0; J[0][5] = -PX * iz * iz;
            J[1][3] = 0.0; J[1][4] = iz; J[1][5] = -PY * iz * iz;
            for (int a = 0; a < 6; ++a) {
                for (int c = 0; c < 6; ++c) JtJ[6 * a + c] += J[0][a] * J[0][c] + J[1][a] * J[1][c];
                Jtr[a] += J[0][a] * rx + J[1][a] * ry;
            }
        }
        for (int a = 0; a < 6; ++a) JtJ[7 * a] *= 1.0 + 1e-12;
        if (!chol_solve_dev<6>(JtJ, Jtr)) break;
        double step = 0.0;
        for (int a = 0; a < 3; ++a) { rv[a] -= Jtr[a]; t[a] -= Jtr[3 + a]; step += Jtr[a] * Jtr[a] + Jtr[3 + a] * Jtr[3 + a] / fmax(1.0, t[a] * t[a]); }
        if (step < 1e-24) break;
    }
    tscm::rotation_and_derivatives(rv, R, dR);
    double *o = Rt_out + 9 * (size_t)k;                                     // Rt = transform^T [r1 r2 t]  (:195-200)
    for (int r = 0; r < 3; ++r) {
        o[3 * r] = T[r] * R[0] + T[3 + r] * R[3] + T[6 + r] * R[6];
        o[3 * r + 1] = T[r] * R[1] + T[3 + r] * R[4] + T[6 + r] * R[7];
        o[3 * r + 2] = T[r] * t[0] + T[3 + r] * t[1] + T[6 + r] * t[2];
    }
    ok_out[k] = 1;
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

extern "C" int tscm_estimate_extrinsic(const double *intr9, const double *pix_u, const double *pix_v, const int *count, int n_views,
                                       const double *worlds, int n_points, int board_w, int device, double *Rt, int *n_estimated)
{
    if (!intr9 || !worlds || n_views < 0 || n_points < 4 || (n_views > 0 && (!pix_u || !pix_v || !count || !Rt))) return tscm_set_error(TSCM_E_INVALID, "NULL or inconsistent argument");
    if (board_w < 1 || n_points / 2 - board_w / 2 - 1 < 0) return tscm_set_error(TSCM_E_INVALID, "board width does not fit the corner count");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return tscm_set_error(TSCM_E_NO_DEVICE, "no HIP device available (tscm_estimate_extrinsic has no CPU fallback)");
    if (device < 0 || device >= ndev) return tscm_set_error(TSCM_E_NO_DEVICE, "device index out of range");
    INIT_TRY(hipSetDevice(device));
    if (n_estimated) *n_estimated = 0;
    if (n_views == 0) return 0;
    const size_t npix = (size_t)n_views * n_points;
    double *d_i = nullptr, *d_u = nullptr, *d_v = nullptr, *d_w = nullptr, *d_rt = nullptr;
    int *d_c = nullptr;
    unsigned char *d_ok = nullptr;
    std::vector<unsigned char> ok(n_views);
    auto body = [&]() -> int {
        INIT_TRY(hipMalloc(reinterpret_cast<void **>(&d_i), 9 * sizeof(double)));
        INIT_TRY(hipMalloc(reinterpret_cast<void **>(&d_u), npix * sizeof(double)));
        INIT_TRY(hipMalloc(reinterpret_cast<void **>(&d_v), npix * sizeof(double)));
        INIT_TRY(hipMalloc(reinterpret_cast<void **>(&d_w), 3 * (size_t)n_points * sizeof(double)));
        INIT_TRY(hipMalloc(reinterpret_cast<void **>(&d_rt), 9 * (size_t)n_views * sizeof(double)));
        INIT_TRY(hipMalloc(reinterpret_cast<void **>(&d_c), (size_t)n_views * sizeof(int)));
        INIT_TRY(hipMalloc(reinterpret_cast<void **>(&d_ok), (size_t)n_views));
        INIT_TRY(hipMemcpy(d_i, intr9, 9 * sizeof(double), hipMemcpyHostToDevice));
        INIT_TRY(hipMemcpy(d_u, pix_u, npix * sizeof(double), hipMemcpyHostToDevice));
        INIT_TRY(hipMemcpy(d_v, pix_v, npix * sizeof(double), hipMemcpyHostToDevice));
        INIT_TRY(hipMemcpy(d_w, worlds, 3 * (size_t)n_points * sizeof(double), hipMemcpyHostToDevice));
        INIT_TRY(hipMemcpy(d_rt, Rt, 9 * (size_t)n_views * sizeof(double), hipMemcpyHostToDevice));      // views without a pose keep the caller's values
        INIT_TRY(hipMemcpy(d_c, count, (size_t)n_views * sizeof(int), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_estimate_extrinsic, dim3((unsigned)((n_views + 63) / 64)), dim3(64), 0, 0, d_i, d_u, d_v, d_c, n_views, d_w, n_points, board_w, d_rt, d_ok);
        INIT_TRY(hipGetLastError());
        INIT_TRY(hipMemcpy(Rt, d_rt, 9 * (size_t)n_views * sizeof(double), hipMemcpyDeviceToHost));
        INIT_TRY(hipMemcpy(ok.data(), d_ok, (size_t)n_views, hipMemcpyDeviceToHost));
        return 0;
    };
    const int rc = body();
    (void)hipFree(d_i); (void)hipFree(d_u); (void)hipFree(d_v); (void)hipFree(d_w); (void)hipFree(d_rt); (void)hipFree(d_c); (void)hipFree(d_ok);
    if (rc) return rc;
    int done = 0;
    for (unsigned char f : ok) done += f;
    if (n_estimated) *n_estimated = done;
    return 0;
}

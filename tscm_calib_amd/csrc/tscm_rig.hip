// tscm_rig.hip -- rig initialisation on the MI355X: MultiCalib::MultiCalib
// (multi_calib.cpp:6-153).  The quadratic hypothesis test (every common board's pose hypothesis
// scored on every common board, both cameras) runs on the GPU, one thread per (hypothesis, board)
// pair with board-major coalesced pixel loads; the 3x3 bookkeeping stays on the host.
#include "tscm/tscm.h"
#include "tscm_math.h"
#include "tscm_fastmath.h"

#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

using namespace tscm;

int tscm_set_error(int code, const std::string &msg);   // tscm_solver.hip

#define RIG_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) return tscm_set_error(TSCM_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

namespace {

struct M3 { double a[9]; };
struct V3 { double a[3]; };

__host__ __device__ inline M3 mul(const M3 &A, const M3 &B)
{
    M3 C;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) C.a[3 * i + j] = A.a[3 * i] * B.a[j] + A.a[3 * i + 1] * B.a[3 + j] + A.a[3 * i + 2] * B.a[6 + j];
    return C;
}
__host__ __device__ inline M3 transpose(const M3 &A)
{
    M3 T;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) T.a[3 * i + j] = A.a[3 * j + i];
    return T;
}
__host__ __device__ inline V3 mul(const M3 &A, const V3 &x)
{
    V3 y;
    for (int i = 0; i < 3; ++i) y.a[i] = A.a[3 * i] * x.a[0] + A.a[3 * i + 1] * x.a[1] + A.a[3 * i + 2] * x.a[2];
    return y;
}
__host__ __device__ inline V3 sub(const V3 &x, const V3 &y) { return V3{ { x.a[0] - y.a[0], x.a[1] - y.a[1], x.a[2] - y.a[2] } }; }
__host__ __device__ inline V3 add(const V3 &x, const V3 &y) { return V3{ { x.a[0] + y.a[0], x.a[1] + y.a[1], x.a[2] + y.a[2] } }; }

// multi_calib.h:130-137 -- R from float32 r1, r2 and their float cross product
void Rt_to_R_t(const double *Rt, M3 &R, V3 &t)
{
    volatile float r1[3], r2[3], p, q;     // float32 roundings and products must survive optimisation
    float r3[3];
    for (int i = 0; i < 3; ++i) { r1[i] = (float)Rt[3 * i]; r2[i] = (float)Rt[3 * i + 1]; }
    p = r1[1] * r2[2]; q = r1[2] * r2[1]; r3[0] = p - q;
    p = r1[2] * r2[0]; q = r1[0] * r2[2]; r3[1] = p - q;
    p = r1[0] * r2[1]; q = r1[1] * r2[0]; r3[2] = p - q;
    for (int i = 0; i < 3; ++i) { R.a[3 * i] = r1[i]; R.a[3 * i + 1] = r2[i]; R.a[3 * i + 2] = r3[i]; }
    t.a[0] = Rt[2]; t.a[1] = Rt[5]; t.a[2] = Rt[8];
}

// orthogonal polar factor U V^T of a nearly orthogonal 3x3 matrix by Newton iteration
// X <- (X + X^{-T}) / 2  (the same matrix cv::Rodrigues gets from its SVD)
M3 polar_orthogonal(const M3 &A)
{
    M3 X = A;
    for (int it = 0; it < 50; ++it) {
        const double *x = X.a;
        const double c00 = x[4] * x[8] - x[5] * x[7], c01 = x[5] * x[6] - x[3] * x[8], c02 = x[3] * x[7] - x[4] * x[6];
        const double c10 = x[2] * x[7] - x[1] * x[8], c11 = x[0] * x[8] - x[2] * x[6], c12 = x[1] * x[6] - x[0] * x[7];
        const double c20 = x[1] * x[5] - x[2] * x[4], c21 = x[2] * x[3] - x[0] * x[5], c22 = x[0] * x[4] - x[1] * x[3];
        const double det = x[0] * c00 + x[1] * c01 + x[2] * c02;
        const double cof[9] = { c00, c01, c02, c10, c11, c12, c20, c21, c22 };     // cofactors: X^{-T} = cof / det
        M3 Y;
        double diff = 0.0;
        for (int i = 0; i < 9; ++i) { Y.a[i] = 0.5 * (x[i] + cof[i] / det); diff = std::fmax(diff, std::fabs(Y.a[i] - x[i])); }
        X = Y;
        if (diff < 1e-16) break;
    }
    return X;
}

// cv::Rodrigues, matrix -> vector (external, OpenCV calib3d), see tscm.h
void rodrigues_inverse(const M3 &Rin, double *r)
{
    const M3 Rm = polar_orthogonal(Rin);
    const double *R = Rm.a;
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    const double s = std::sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1.0) * 0.5;
    c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
    double theta = std::acos(c);
    if (s < 1e-5) {
        if (c > 0) { r[0] = r[1] = r[2] = 0.0; return; }
        double t = (R[0] + 1.0) * 0.5;
        rx = std::sqrt(std::fmax(t, 0.0));
        t = (R[4] + 1.0) * 0.5;
        ry = std::sqrt(std::fmax(t, 0.0)) * (R[1] < 0 ? -1.0 : 1.0);
        t = (R[8] + 1.0) * 0.5;
        rz = std::sqrt(std::fmax(t, 0.0)) * (R[2] < 0 ? -1.0 : 1.0);
        if (std::fabs(rx) < std::fabs(ry) && std::fabs(rx) < std::fabs(rz) && ((R[5] > 0) != (ry * rz > 0))) rz = -rz;
        theta /= std::sqrt(rx * rx + ry * ry + rz * rz);
        r[0] = rx * theta; r[1] = ry * theta; r[2] = rz * theta;
    } else {
        const double vth = theta / (2.0 * s);
        r[0] = rx * vth; r[1] = ry * vth; r[2] = rz * vth;
    }
}

// TS.h:58-69 for one board: SUM of pixel errors, pixels stored point-major with stride `stride`
__device__ inline double reproject_error_sum(const double *I, const double *pu, const double *pv, size_t stride,
                                             const double *worlds, int n, const M3 &R, const V3 &t)
{
    double err = 0.0;
    for (int c = 0; c < n; ++c) {
        const double x = worlds[3 * c], y = worlds[3 * c + 1], z = worlds[3 * c + 2];
        const double X = R.a[0] * x + R.a[1] * y + R.a[2] * z + t.a[0];
        const double Y = R.a[3] * x + R.a[4] * y + R.a[5] * z + t.a[1];
        const double Z = R.a[6] * x + R.a[7] * y + R.a[8] * z + t.a[2];
        double u, v;
        project_point(I, X, Y, Z, u, v);
        const double du = pu[c * stride] - u, dv = pv[c * stride] - v;
        err += sqrt(du * du + dv * dv);
    }
    return err;
}

// One board point of one common board as the hypothesis kernel consumes it: the point in the frame
// of the camera that detected the board (R_k w + t_k, prepared on the host) and the pixel the OTHER
// camera of the pair observed.  64 bytes = one scalar load of 16 dwords.
struct HypPoint { double qx, qy, qz, pu, pv, pad[3]; };

struct StageArgs {
    int J, K, n, ksplit;
    const double *Rs, *ts;                 // [J*9], [J*3] hypotheses
    const HypPoint *pts;                   // [K][2][n]: direction 0 = seen by camera i, scored in camera i-1; 1 = the reverse
    double intrP[9], intrI[9];             // intrinsics of camera i-1 / camera i
    M3 Rp; V3 tp;                          // pose of camera i-1
    double *partial;                       // [ksplit][J]
};

// TS.cpp:332-344 + TS.h:65 for one point: pixel error of (X, Y, Z) against (pu, pv).
// sqrt(x) = x * rsqrt(x) and 1/ksai from the hardware seeds with one third-order step (~1 ulp).
template <bool SKEW>
__device__ __forceinline__ double pixel_error(const double *I, double beta, double X, double Y, double Z, double pu, double pv)
{
    const double rho2 = __builtin_fma(Y, Y, X * X);
    const double s1 = __builtin_fma(Z, Z, rho2);
    const double d1 = s1 * fast_rsqrt(s1);
    const double z1 = __builtin_fma(I[4], d1, Z);
    const double s2 = __builtin_fma(z1, z1, rho2);
    const double d2 = s2 * fast_rsqrt(s2);
    const double z2 = __builtin_fma(I[5], d2, z1);
    const double s3 = __builtin_fma(z2, z2, rho2);
    const double d3 = s3 * fast_rsqrt(s3);
    const double ik = fast_rcp(__builtin_fma(beta, d3, z2));
    const double mx = X * ik, my = Y * ik;
    double du, dv;
    if (SKEW) {
        du = pu - __builtin_fma(I[0], mx, __builtin_fma(I[7], my, I[2]));
        dv = pv - __builtin_fma(I[8], mx, __builtin_fma(I[1], my, I[3]));
    } else {
        du = pu - __builtin_fma(I[0], mx, I[2]);
        dv = pv - __builtin_fma(I[1], my, I[3]);
    }
    const double e2 = fmax(__builtin_fma(dv, dv, du * du), 1e-300);      // rsq(0) = inf
    return e2 * fast_rsqrt(e2);
}

// multi_calib.cpp:50-85.  One LANE per pose hypothesis j, one wave per (64 hypotheses, slice of the
// common boards): everything that depends on the board only -- the prepared points and pixels --
// is wave-uniform and arrives through the scalar data path, so the vector ALUs do nothing but
// P = A_j q + a_j and the projection.  Per lane the errors are summed in the reference's order
// (corner by corner, direction 0 then 1, board by board).  grid (ceil(J/64), ksplit) x 64
template <bool SKEW>
__global__ __launch_bounds__(64) void k_rig_hyp_errors(StageArgs s)
{
    const int j = min((int)(blockIdx.x * 64 + threadIdx.x), s.J - 1);
    M3 Rsj; V3 tsj;
    for (int i = 0; i < 9; ++i) Rsj.a[i] = s.Rs[9 * (size_t)j + i];
    for (int i = 0; i < 3; ++i) tsj.a[i] = s.ts[3 * (size_t)j + i];
    // direction 0: R_ki = camera_R_k * Rs[j].t(), t_ki = camera_t_k - R_ki * ts[j]   (:57-60)
    // direction 1: R_ki = Rs[j] * camera_R_k.t(), t_ki = ts[j] - R_ki * camera_t_k   (:69-72)
    M3 A[2]; V3 a[2];
    A[0] = mul(s.Rp, transpose(Rsj));
    a[0] = sub(s.tp, mul(A[0], tsj));
    A[1] = mul(Rsj, transpose(s.Rp));
    a[1] = sub(tsj, mul(A[1], s.tp));
    const double betaP = s.intrP[6] / (1.0 - s.intrP[6]), betaI = s.intrI[6] / (1.0 - s.intrI[6]);
    const int per = (s.K + s.ksplit - 1) / s.ksplit;
    const int k0 = blockIdx.y * per, k1 = min(s.K, k0 + per);
    double error = 0.0;
    for (int k = k0; k < k1; ++k) {
        const HypPoint *__restrict__ pt = s.pts + (size_t)k * 2 * s.n;
#pragma unroll
        for (int dir = 0; dir < 2; ++dir) {
            const double *I = dir == 0 ? s.intrP : s.intrI;
            const double beta = dir == 0 ? betaP : betaI;
            double eb = 0.0;
#pragma unroll 2
            for (int c = 0; c < s.n; ++c) {
                const HypPoint q = pt[dir * s.n + c];
                const double X = __builtin_fma(A[dir].a[0], q.qx, __builtin_fma(A[dir].a[1], q.qy, __builtin_fma(A[dir].a[2], q.qz, a[dir].a[0])));
                const double Y = __builtin_fma(A[dir].a[3], q.qx, __builtin_fma(A[dir].a[4], q.qy, __builtin_fma(A[dir].a[5], q.qz, a[dir].a[1])));
                const double Z = __builtin_fma(A[dir].a[6], q.qx, __builtin_fma(A[dir].a[7], q.qy, __builtin_fma(A[dir].a[8], q.qz, a[dir].a[2])));
                eb += pixel_error<SKEW>(I, beta, X, Y, Z, q.pu, q.pv);
            }
            error += eb;
        }
    }
    if ((int)(blockIdx.x * 64 + threadIdx.x) < s.J) s.partial[(size_t)blockIdx.y * s.J + j] = error;
}

// prepared points of one stage: thread per (common board h, corner c)
__global__ void k_rig_points(int K, int n, int B, int i, const int *__restrict__ common, const double *__restrict__ pose,
                             const double *__restrict__ pu, const double *__restrict__ pv, const double *__restrict__ worlds,
                             HypPoint *__restrict__ pts)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= K * n) return;
    const int h = t / n, c = t - h * n, j = common[h];
    const size_t vi = (size_t)i * B + j, vp = (size_t)(i - 1) * B + j;
    const double *Pi = pose + 12 * vi, *Pp = pose + 12 * vp;
    const double x = worlds[3 * c], y = worlds[3 * c + 1], z = worlds[3 * c + 2];
    HypPoint a, b;
    a.qx = Pi[0] * x + Pi[1] * y + Pi[2] * z + Pi[9];  a.qy = Pi[3] * x + Pi[4] * y + Pi[5] * z + Pi[10]; a.qz = Pi[6] * x + Pi[7] * y + Pi[8] * z + Pi[11];
    b.qx = Pp[0] * x + Pp[1] * y + Pp[2] * z + Pp[9];  b.qy = Pp[3] * x + Pp[4] * y + Pp[5] * z + Pp[10]; b.qz = Pp[6] * x + Pp[7] * y + Pp[8] * z + Pp[11];
    a.pu = pu[vp * n + c]; a.pv = pv[vp * n + c];       // seen by camera i, scored against camera i-1's pixels
    b.pu = pu[vi * n + c]; b.pv = pv[vi * n + c];
    a.pad[0] = a.pad[1] = a.pad[2] = b.pad[0] = b.pad[1] = b.pad[2] = 0.0;
    pts[((size_t)h * 2) * n + c] = a;
    pts[((size_t)h * 2 + 1) * n + c] = b;
}

__global__ void k_rig_hyp_reduce(const double *partial, int J, int ksplit, double *err)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= J) return;
    double e = 0.0;
    for (int b = 0; b < ksplit; ++b) e += partial[(size_t)b * J + j];
    err[j] = e;
}

constexpr int kRigMaxCam = 32;      // same limit as the solver (tscm_kernels.h: kMaxCam)

// multi_calib.cpp:90-151, one thread per board
__global__ void k_rig_boards(int C, int B, int n, const unsigned char *has, const double *pose, const double *pu, const double *pv,
                             const double *worlds, const double *intr, const double *cam_R, const double *cam_t,
                             double *board_R, double *board_t, unsigned char *initial)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int ids[kRigMaxCam], nc = 0;
    for (int m = 0; m < C; ++m) if (has[(size_t)m * B + b]) ids[nc++] = m;
    initial[b] = nc > 0;
    for (int i = 0; i < 9; ++i) board_R[9 * (size_t)b + i] = 0.0;
    for (int i = 0; i < 3; ++i) board_t[3 * (size_t)b + i] = 0.0;
    if (nc == 0) return;
    M3 Rs[kRigMaxCam]; V3 ts[kRigMaxCam];
    for (int q = 0; q < nc; ++q) {
        const int m = ids[q];
        M3 cR, chR; V3 ct, cht;
        for (int i = 0; i < 9; ++i) { cR.a[i] = cam_R[9 * m + i]; chR.a[i] = pose[12 * ((size_t)m * B + b) + i]; }
        for (int i = 0; i < 3; ++i) { ct.a[i] = cam_t[3 * m + i]; cht.a[i] = pose[12 * ((size_t)m * B + b) + 9 + i]; }
        const M3 cRt = transpose(cR);
        Rs[q] = mul(cRt, chR);                       // camera_R.t() * chess_R
        ts[q] = mul(cRt, sub(cht, ct));              // camera_R.t() * (chess_t - camera_t)
    }
    int best = 0;
    if (nc > 1) {
        double min_error = 1e10;
        best = -1;
        for (int q = 0; q < nc; ++q) {
            double error = 0.0;
            for (int k = 0; k < nc; ++k) {
                const int m = ids[k];
                M3 cR; V3 ct;
                double I[9];
                for (int i = 0; i < 9; ++i) { cR.a[i] = cam_R[9 * m + i]; I[i] = intr[9 * m + i]; }
                for (int i = 0; i < 3; ++i) ct.a[i] = cam_t[3 * m + i];
                const size_t base = (size_t)n * ((size_t)m * B + b);
                error += reproject_error_sum(I, pu + base, pv + base, 1, worlds, n, mul(cR, Rs[q]), add(mul(cR, ts[q]), ct));
            }
            if (error < min_error) { min_error = error; best = q; }
        }
        if (best < 0) { initial[b] = 255; return; }     // every hypothesis scored >= 1e10: the reference indexes Rs[-1] here
    }
    for (int i = 0; i < 9; ++i) board_R[9 * (size_t)b + i] = Rs[best].a[i];
    for (int i = 0; i < 3; ++i) board_t[3 * (size_t)b + i] = ts[best].a[i];
}

template <typename T>
struct DevBuf {
    T *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc(reinterpret_cast<void **>(&p), (n ? n : 1) * sizeof(T)); }
    hipError_t upload(const T *h, size_t n) { hipError_t e = alloc(n); if (e != hipSuccess || n == 0) return e; return hipMemcpy(p, h, n * sizeof(T), hipMemcpyHostToDevice); }
    hipError_t upload(const std::vector<T> &h) { hipError_t e = alloc(h.size()); if (e != hipSuccess || h.empty()) return e; return hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice); }
};

double wall() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

extern "C" int tscm_poses_from_r1r2t(const double *Rt, const unsigned char *has, int n, double *rt)
{
    if (n < 0 || (n > 0 && (!Rt || !rt))) return tscm_set_error(TSCM_E_INVALID, "NULL argument");
    for (int i = 0; i < n; ++i) {
        if (has && !has[i]) continue;                       // TS.cpp:64-65
        M3 R; V3 t;
        Rt_to_R_t(Rt + 9 * (size_t)i, R, t);                // the same float32 construction (TS.cpp:66-69)
        rodrigues_inverse(R, rt + 6 * (size_t)i);
        std::memcpy(rt + 6 * (size_t)i + 3, t.a, sizeof(t.a));
    }
    return 0;
}

extern "C" int tscm_rig_init(const tscm_rig_input *in, int device, tscm_rig_result *out)
{
    if (!in || !out) return tscm_set_error(TSCM_E_INVALID, "NULL argument");
    const int C = in->n_cameras, B = in->n_boards, n = in->n_points;
    if (C < 1 || B < 0 || n < 1) return tscm_set_error(TSCM_E_INVALID, "bad rig dimensions");
    if (C > kRigMaxCam) return tscm_set_error(TSCM_E_UNSUPPORTED, "more than 32 cameras");
    if (!in->worlds || !in->intr || !in->has || !in->Rt || !in->pix_u || !in->pix_v) return tscm_set_error(TSCM_E_INVALID, "NULL input array");
    if (!out->cam_R || !out->cam_t || !out->cam_rt || !out->board_R || !out->board_t || !out->board_rt || !out->board_initial) return tscm_set_error(TSCM_E_INVALID, "NULL output array");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return tscm_set_error(TSCM_E_NO_DEVICE, "no HIP device available (tscm_rig_init has no CPU fallback)");
    if (device < 0 || device >= ndev) return tscm_set_error(TSCM_E_NO_DEVICE, "device index out of range");
    RIG_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    RIG_TRY(hipGetDeviceProperties(&prop, device));
    const double t_start = wall();
    out->seconds_hypotheses = 0.0; out->n_projections = 0;

    // Rt_to_R_t of every (camera, board) with a detection
    std::vector<double> pose(12 * (size_t)C * B, 0.0);
    for (int m = 0; m < C; ++m)
        for (int j = 0; j < B; ++j) {
            if (!in->has[(size_t)m * B + j]) continue;
            M3 R; V3 t;
            Rt_to_R_t(in->Rt + 9 * ((size_t)m * B + j), R, t);
            std::memcpy(&pose[12 * ((size_t)m * B + j)], R.a, sizeof(R.a));
            std::memcpy(&pose[12 * ((size_t)m * B + j) + 9], t.a, sizeof(t.a));
        }
    // everything the two selection loops read goes to the device once
    DevBuf<double> d_worlds, d_intr, d_pose, d_pu, d_pv;
    DevBuf<unsigned char> d_has;
    RIG_TRY(d_worlds.upload(in->worlds, 3 * (size_t)n));
    RIG_TRY(d_intr.upload(in->intr, 9 * (size_t)C));
    RIG_TRY(d_pose.upload(pose));
    RIG_TRY(d_pu.upload(in->pix_u, (size_t)C * B * n));
    RIG_TRY(d_pv.upload(in->pix_v, (size_t)C * B * n));
    RIG_TRY(d_has.upload(in->has, (size_t)C * B));

    std::vector<M3> camR(C); std::vector<V3> camt(C);
    for (int i = 0; i < C; ++i) {
        if (out->cam_choice) out->cam_choice[i] = -1;
        if (out->cam_min_error) out->cam_min_error[i] = 0.0;
        if (i == 0) {
            camR[0] = M3{ { 1, 0, 0, 0, 1, 0, 0, 0, 1 } }; camt[0] = V3{ { 0, 0, 0 } };
            continue;
        }
        std::vector<int> common;
        for (int j = 0; j < B; ++j) if (in->has[(size_t)(i - 1) * B + j] && in->has[(size_t)i * B + j]) common.push_back(j);
        const int K = (int)common.size();
        if (K == 0) return tscm_set_error(TSCM_E_INVALID, "adjacent cameras " + std::to_string(i - 1) + " and " + std::to_string(i) + " share no board");
        // hypotheses (:29-48) on the host, the prepared points of the stage on the device
        std::vector<double> Rs(9 * (size_t)K), ts(3 * (size_t)K);
        for (int h = 0; h < K; ++h) {
            const int j = common[h];
            M3 Ri, Rk; V3 ti, tk;
            std::memcpy(Ri.a, &pose[12 * ((size_t)i * B + j)], sizeof(Ri.a)); std::memcpy(ti.a, &pose[12 * ((size_t)i * B + j) + 9], sizeof(ti.a));
            std::memcpy(Rk.a, &pose[12 * ((size_t)(i - 1) * B + j)], sizeof(Rk.a)); std::memcpy(tk.a, &pose[12 * ((size_t)(i - 1) * B + j) + 9], sizeof(tk.a));
            const M3 Rik = mul(Ri, transpose(Rk));
            const V3 tik = sub(ti, mul(Rik, tk));
            const M3 Rh = mul(Rik, camR[i - 1]);
            const V3 th = add(mul(Rik, camt[i - 1]), tik);
            std::memcpy(&Rs[9 * (size_t)h], Rh.a, sizeof(Rh.a)); std::memcpy(&ts[3 * (size_t)h], th.a, sizeof(th.a));
        }
        DevBuf<double> dRs, dts, dpart, derr;
        DevBuf<HypPoint> dpts;
        DevBuf<int> dcommon;
        RIG_TRY(dRs.upload(Rs)); RIG_TRY(dts.upload(ts)); RIG_TRY(dcommon.upload(common));
        RIG_TRY(dpts.alloc((size_t)K * 2 * n));
        hipLaunchKernelGGL(k_rig_points, dim3((unsigned)(((size_t)K * n + 255) / 256)), dim3(256), 0, 0, K, n, B, i, dcommon.p, d_pose.p, d_pu.p, d_pv.p,
                           d_worlds.p, dpts.p);
        const bool skew = in->intr[9 * i + 7] != 0.0 || in->intr[9 * i + 8] != 0.0 || in->intr[9 * (i - 1) + 7] != 0.0 || in->intr[9 * (i - 1) + 8] != 0.0;
        auto kern = skew ? k_rig_hyp_errors<true> : k_rig_hyp_errors<false>;
        // one round of resident waves: slices of the boards so that (hypothesis groups x slices) fills the chip once
        int per_cu = 0;
        RIG_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 64, 0));
        const int jgroups = (K + 63) / 64;
        const int resident = std::max(1, per_cu) * prop.multiProcessorCount;
        StageArgs s{};
        s.J = K; s.K = K; s.n = n;
        s.ksplit = std::max(1, std::min(K, resident / jgroups));
        s.Rs = dRs.p; s.ts = dts.p; s.pts = dpts.p;
        std::memcpy(s.intrI, in->intr + 9 * i, sizeof(s.intrI)); std::memcpy(s.intrP, in->intr + 9 * (i - 1), sizeof(s.intrP));
        s.Rp = camR[i - 1]; s.tp = camt[i - 1];
        RIG_TRY(dpart.alloc((size_t)K * s.ksplit)); RIG_TRY(derr.alloc((size_t)K));
        s.partial = dpart.p;
        hipEvent_t e0, e1;
        RIG_TRY(hipEventCreate(&e0)); RIG_TRY(hipEventCreate(&e1));
        RIG_TRY(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kern, dim3(jgroups, s.ksplit), dim3(64), 0, 0, s);
        hipLaunchKernelGGL(k_rig_hyp_reduce, dim3((K + 255) / 256), dim3(256), 0, 0, dpart.p, K, s.ksplit, derr.p);
        RIG_TRY(hipEventRecord(e1, 0));
        RIG_TRY(hipEventSynchronize(e1));
        float ms = 0.f;
        RIG_TRY(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        RIG_TRY(hipGetLastError());
        out->seconds_hypotheses += 1e-3 * ms;
        out->n_projections += 2LL * n * (long long)K * K;
        std::vector<double> err(K);
        RIG_TRY(hipMemcpy(err.data(), derr.p, sizeof(double) * K, hipMemcpyDeviceToHost));
        double min_error = 1e10; int min_id = -1;
        for (int j = 0; j < K; ++j) if (err[j] < min_error) { min_error = err[j]; min_id = j; }     // strict <: first minimum (:79-83)
        if (min_id < 0) return tscm_set_error(TSCM_E_INVALID, "no pose hypothesis with a finite reprojection error < 1e10 (the reference indexes Rs[-1] here)");
        std::memcpy(camR[i].a, &Rs[9 * (size_t)min_id], sizeof(camR[i].a));
        std::memcpy(camt[i].a, &ts[3 * (size_t)min_id], sizeof(camt[i].a));
        if (out->cam_choice) out->cam_choice[i] = min_id;
        if (out->cam_min_error) out->cam_min_error[i] = min_error;
    }
    for (int i = 0; i < C; ++i) {
        std::memcpy(out->cam_R + 9 * i, camR[i].a, sizeof(camR[i].a));
        std::memcpy(out->cam_t + 3 * i, camt[i].a, sizeof(camt[i].a));
        rodrigues_inverse(camR[i], out->cam_rt + 6 * i);
        std::memcpy(out->cam_rt + 6 * i + 3, camt[i].a, sizeof(camt[i].a));
    }
    // boards (:90-151)
    if (B > 0) {
        std::vector<double> cR(9 * (size_t)C), ct(3 * (size_t)C);
        for (int i = 0; i < C; ++i) { std::memcpy(&cR[9 * (size_t)i], camR[i].a, sizeof(camR[i].a)); std::memcpy(&ct[3 * (size_t)i], camt[i].a, sizeof(camt[i].a)); }
        DevBuf<unsigned char> dinit;
        DevBuf<double> dcR, dct, dbR, dbt;
        RIG_TRY(dcR.upload(cR)); RIG_TRY(dct.upload(ct));
        RIG_TRY(dbR.alloc(9 * (size_t)B)); RIG_TRY(dbt.alloc(3 * (size_t)B)); RIG_TRY(dinit.alloc((size_t)B));
        hipLaunchKernelGGL(k_rig_boards, dim3((B + 127) / 128), dim3(128), 0, 0, C, B, n, d_has.p, d_pose.p, d_pu.p, d_pv.p, d_worlds.p, d_intr.p,
                           dcR.p, dct.p, dbR.p, dbt.p, dinit.p);
        RIG_TRY(hipDeviceSynchronize());
        RIG_TRY(hipGetLastError());
        RIG_TRY(hipMemcpy(out->board_R, dbR.p, sizeof(double) * 9 * (size_t)B, hipMemcpyDeviceToHost));
        RIG_TRY(hipMemcpy(out->board_t, dbt.p, sizeof(double) * 3 * (size_t)B, hipMemcpyDeviceToHost));
        RIG_TRY(hipMemcpy(out->board_initial, dinit.p, (size_t)B, hipMemcpyDeviceToHost));
        for (int b = 0; b < B; ++b) {
            std::memset(out->board_rt + 6 * (size_t)b, 0, 6 * sizeof(double));
            if (!out->board_initial[b]) continue;
            if (out->board_initial[b] == 255) return tscm_set_error(TSCM_E_INVALID, "board " + std::to_string(b) + ": no pose hypothesis with a reprojection error < 1e10");
            M3 R;
            std::memcpy(R.a, out->board_R + 9 * (size_t)b, sizeof(R.a));
            rodrigues_inverse(R, out->board_rt + 6 * (size_t)b);
            std::memcpy(out->board_rt + 6 * (size_t)b + 3, out->board_t + 3 * (size_t)b, 3 * sizeof(double));
            int ncam = 0;
            for (int m = 0; m < C; ++m) ncam += in->has[(size_t)m * B + b] ? 1 : 0;
            if (ncam > 1) out->n_projections += (long long)ncam * ncam * n;
        }
    }
    out->seconds_total = wall() - t_start;
    return 0;
}

/*
 * tscm_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 * See tscm_oracle.h for scope, provenance and the "parity unpinned" statement.
 *
 * Single-threaded on purpose: the reference never sets options.num_threads
 * (TS.cpp:271-274, multi_calib.cpp:209-212), so Ceres runs on one thread.
 */
#define _POSIX_C_SOURCE 200809L
#include "tscm_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ======================================================================== *
 *  Forward-mode dual numbers == ceres::Jet<double, N>  (external: jet.h)
 *  N = 15 for the mono functor (TS.cpp:261-264: <2, 9, 6>) and
 *  N = 21 for the multi functor (multi_calib.cpp:177-180: <2, 6, 6, 9>).
 * ======================================================================== */
#define JW 21
typedef struct { double a; double v[JW]; } jet;

static inline jet j_const(double s, int n) { jet r; r.a = s; for (int i = 0; i < n; ++i) r.v[i] = 0.0; return r; }
static inline jet j_var(double s, int k, int n) { jet r = j_const(s, n); r.v[k] = 1.0; return r; }
static inline jet j_add(jet f, jet g, int n) { jet r; r.a = f.a + g.a; for (int i = 0; i < n; ++i) r.v[i] = f.v[i] + g.v[i]; return r; }
static inline jet j_sub(jet f, jet g, int n) { jet r; r.a = f.a - g.a; for (int i = 0; i < n; ++i) r.v[i] = f.v[i] - g.v[i]; return r; }
/* Jet(f.a * g.a, f.a * g.v + f.v * g.a) */
static inline jet j_mul(jet f, jet g, int n) { jet r; r.a = f.a * g.a; for (int i = 0; i < n; ++i) r.v[i] = f.a * g.v[i] + f.v[i] * g.a; return r; }
/* g_a_inverse = 1/g.a; f_a_by_g_a = f.a*g_a_inverse; Jet(f_a_by_g_a, (f.v - f_a_by_g_a*g.v)*g_a_inverse) */
static inline jet j_div(jet f, jet g, int n)
{
    jet r; const double gi = 1.0 / g.a; const double fg = f.a * gi;
    r.a = fg; for (int i = 0; i < n; ++i) r.v[i] = (f.v[i] - fg * g.v[i]) * gi; return r;
}
/* tmp = sqrt(f.a); two_a_inverse = 1/(2*tmp); Jet(tmp, f.v*two_a_inverse) */
static inline jet j_sqrt(jet f, int n)
{
    jet r; const double t = sqrt(f.a); const double h = 1.0 / (2.0 * t);
    r.a = t; for (int i = 0; i < n; ++i) r.v[i] = f.v[i] * h; return r;
}
static inline jet j_sin(jet f, int n) { jet r; const double c = cos(f.a); r.a = sin(f.a); for (int i = 0; i < n; ++i) r.v[i] = c * f.v[i]; return r; }
static inline jet j_cos(jet f, int n) { jet r; const double s = -sin(f.a); r.a = cos(f.a); for (int i = 0; i < n; ++i) r.v[i] = s * f.v[i]; return r; }

/* ======================================================================== *
 *  ceres::AngleAxisRotatePoint (external: ceres/rotation.h), call sites
 *  TS.h:112, multi_calib.h:158,164.  Two branches: Rodrigues formula when
 *  theta^2 > DBL_EPSILON, first-order Taylor (pt + w x pt) otherwise.
 * ======================================================================== */
void orc_angle_axis_rotate_point(const double *aa, const double *pt, double *out)
{
    const double theta2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
    if (theta2 > DBL_EPSILON) {
        const double theta = sqrt(theta2);
        const double costheta = cos(theta);
        const double sintheta = sin(theta);
        const double theta_inverse = 1.0 / theta;
        const double w[3] = { aa[0] * theta_inverse, aa[1] * theta_inverse, aa[2] * theta_inverse };
        const double wxp[3] = { w[1] * pt[2] - w[2] * pt[1], w[2] * pt[0] - w[0] * pt[2], w[0] * pt[1] - w[1] * pt[0] };
        const double tmp = (w[0] * pt[0] + w[1] * pt[1] + w[2] * pt[2]) * (1.0 - costheta);
        out[0] = pt[0] * costheta + wxp[0] * sintheta + w[0] * tmp;
        out[1] = pt[1] * costheta + wxp[1] * sintheta + w[1] * tmp;
        out[2] = pt[2] * costheta + wxp[2] * sintheta + w[2] * tmp;
    } else {
        const double wxp[3] = { aa[1] * pt[2] - aa[2] * pt[1], aa[2] * pt[0] - aa[0] * pt[2], aa[0] * pt[1] - aa[1] * pt[0] };
        out[0] = pt[0] + wxp[0]; out[1] = pt[1] + wxp[1]; out[2] = pt[2] + wxp[2];
    }
}

static inline void aa_rotate_jet(const jet *aa, const jet *pt, jet *out, int n)
{
    const jet theta2 = j_add(j_add(j_mul(aa[0], aa[0], n), j_mul(aa[1], aa[1], n), n), j_mul(aa[2], aa[2], n), n);
    if (theta2.a > DBL_EPSILON) {
        const jet theta = j_sqrt(theta2, n);
        const jet costheta = j_cos(theta, n);
        const jet sintheta = j_sin(theta, n);
        const jet theta_inverse = j_div(j_const(1.0, n), theta, n);
        const jet w[3] = { j_mul(aa[0], theta_inverse, n), j_mul(aa[1], theta_inverse, n), j_mul(aa[2], theta_inverse, n) };
        const jet wxp[3] = {
            j_sub(j_mul(w[1], pt[2], n), j_mul(w[2], pt[1], n), n),
            j_sub(j_mul(w[2], pt[0], n), j_mul(w[0], pt[2], n), n),
            j_sub(j_mul(w[0], pt[1], n), j_mul(w[1], pt[0], n), n) };
        const jet dot = j_add(j_add(j_mul(w[0], pt[0], n), j_mul(w[1], pt[1], n), n), j_mul(w[2], pt[2], n), n);
        const jet tmp = j_mul(dot, j_sub(j_const(1.0, n), costheta, n), n);
        for (int i = 0; i < 3; ++i)
            out[i] = j_add(j_add(j_mul(pt[i], costheta, n), j_mul(wxp[i], sintheta, n), n), j_mul(w[i], tmp, n), n);
    } else {
        const jet wxp[3] = {
            j_sub(j_mul(aa[1], pt[2], n), j_mul(aa[2], pt[1], n), n),
            j_sub(j_mul(aa[2], pt[0], n), j_mul(aa[0], pt[2], n), n),
            j_sub(j_mul(aa[0], pt[1], n), j_mul(aa[1], pt[0], n), n) };
        for (int i = 0; i < 3; ++i) out[i] = j_add(pt[i], wxp[i], n);
    }
}

/* cv::Rodrigues(rvec -> R): R = cos I + (1-cos) k k^T + sin [k]x, identity for theta==0.
 * Used by the write-back (multi_calib.h:42-57,104-108; TS.cpp:90-105). */
void orc_rodrigues(const double *aa, double *R)
{
    const double theta = sqrt(aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2]);
    if (theta < DBL_EPSILON) {
        R[0] = 1; R[1] = 0; R[2] = 0; R[3] = 0; R[4] = 1; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1;
        return;
    }
    const double c = cos(theta), s = sin(theta), c1 = 1.0 - c, it = 1.0 / theta;
    const double kx = aa[0] * it, ky = aa[1] * it, kz = aa[2] * it;
    R[0] = c + c1 * kx * kx;      R[1] = c1 * kx * ky - s * kz; R[2] = c1 * kx * kz + s * ky;
    R[3] = c1 * kx * ky + s * kz; R[4] = c + c1 * ky * ky;      R[5] = c1 * ky * kz - s * kx;
    R[6] = c1 * kx * kz - s * ky; R[7] = c1 * ky * kz + s * kx; R[8] = c + c1 * kz * kz;
}

/* ======================================================================== *
 *  Triple-sphere projection tail shared by both functors
 *  (TS.h:117-125 == multi_calib.h:170-178).  Skew terms are commented out in
 *  the functors (TS.h:122-123, multi_calib.h:175-176): b, c are unused.
 * ======================================================================== */
static inline void ts_tail_double(const double *I, const double *P, const double *obs, double *res)
{
    const double one = 1.0;
    const double d1 = sqrt(P[0] * P[0] + P[1] * P[1] + P[2] * P[2]);
    const double d2 = sqrt(P[0] * P[0] + P[1] * P[1] + (P[2] + I[4] * d1) * (P[2] + I[4] * d1));
    const double d3 = sqrt(P[0] * P[0] + P[1] * P[1] + (P[2] + I[4] * d1 + I[5] * d2) * (P[2] + I[4] * d1 + I[5] * d2));
    const double ksai = P[2] + I[4] * d1 + I[5] * d2 + I[6] / (one - I[6]) * d3;
    const double pixel_x = I[0] * P[0] / ksai + I[2];
    const double pixel_y = I[1] * P[1] / ksai + I[3];
    res[0] = obs[0] - pixel_x;
    res[1] = obs[1] - pixel_y;
}

static inline void ts_tail_jet(const jet *I, const jet *P, const double *obs, jet *res, int n)
{
#define ADD(x, y) j_add((x), (y), n)
#define MUL(x, y) j_mul((x), (y), n)
    const jet one = j_const(1.0, n);
    const jet xx_yy = ADD(MUL(P[0], P[0]), MUL(P[1], P[1]));
    const jet d1 = j_sqrt(ADD(xx_yy, MUL(P[2], P[2])), n);
    const jet z1 = ADD(P[2], MUL(I[4], d1));
    const jet d2 = j_sqrt(ADD(xx_yy, MUL(z1, z1)), n);
    const jet z2 = ADD(z1, MUL(I[5], d2));
    const jet d3 = j_sqrt(ADD(xx_yy, MUL(z2, z2)), n);
    const jet ksai = ADD(z2, MUL(j_div(I[6], j_sub(one, I[6], n), n), d3));
    const jet pixel_x = ADD(j_div(MUL(I[0], P[0]), ksai, n), I[2]);
    const jet pixel_y = ADD(j_div(MUL(I[1], P[1]), ksai, n), I[3]);
    res[0] = j_sub(j_const(obs[0], n), pixel_x, n);
    res[1] = j_sub(j_const(obs[1], n), pixel_y, n);
#undef ADD
#undef MUL
}

/* TS.h:100-131 with T = double */
void orc_mono_residual(const double *intr, const double *rt, const double *obs,
                       const double *board_pt, double *res)
{
    const double p[3] = { board_pt[0], board_pt[1], 0.0 };
    double P[3];
    orc_angle_axis_rotate_point(rt, p, P);
    P[0] += rt[3]; P[1] += rt[4]; P[2] += rt[5];
    ts_tail_double(intr, P, obs, res);
}

/* multi_calib.h:146-195 with T = double */
void orc_multi_residual(const double *cam_rt, const double *board_rt, const double *intr,
                        const double *obs, const double *board_pt, double *res)
{
    const double p[3] = { board_pt[0], board_pt[1], 0.0 };
    double Pw[3], Pc[3];
    orc_angle_axis_rotate_point(board_rt, p, Pw);
    Pw[0] += board_rt[3]; Pw[1] += board_rt[4]; Pw[2] += board_rt[5];
    orc_angle_axis_rotate_point(cam_rt, Pw, Pc);
    Pc[0] += cam_rt[3]; Pc[1] += cam_rt[4]; Pc[2] += cam_rt[5];
    ts_tail_double(intr, Pc, obs, res);
}

/* TS.h:100-131 with T = Jet<double,15>; parameter blocks (intrinsic 9, rt 6)
 * in the AddResidualBlock order of TS.cpp:266-267. */
void orc_mono_autodiff(const double *intr, const double *rt, const double *obs,
                       const double *board_pt, double *res, double *J_intr, double *J_rt)
{
    enum { N = 15 };
    jet I[9], q[6], p[3], P[3], r[2];
    for (int i = 0; i < 9; ++i) I[i] = j_var(intr[i], i, N);
    for (int i = 0; i < 6; ++i) q[i] = j_var(rt[i], 9 + i, N);
    p[0] = j_const(board_pt[0], N); p[1] = j_const(board_pt[1], N); p[2] = j_const(0.0, N);
    aa_rotate_jet(q, p, P, N);
    for (int i = 0; i < 3; ++i) P[i] = j_add(P[i], q[3 + i], N);
    ts_tail_jet(I, P, obs, r, N);
    for (int k = 0; k < 2; ++k) {
        if (res) res[k] = r[k].a;
        if (J_intr) for (int i = 0; i < 9; ++i) J_intr[k * 9 + i] = r[k].v[i];
        if (J_rt) for (int i = 0; i < 6; ++i) J_rt[k * 6 + i] = r[k].v[9 + i];
    }
}

/* multi_calib.h:146-195 with T = Jet<double,21>; parameter blocks
 * (camera_rt 6, chessboard_rt 6, intrinsic 9) as in multi_calib.cpp:182-184. */
void orc_multi_autodiff(const double *cam_rt, const double *board_rt, const double *intr,
                        const double *obs, const double *board_pt, double *res,
                        double *J_cam, double *J_board, double *J_intr)
{
    enum { N = 21 };
    jet c[6], b[6], I[9], p[3], Pw[3], Pc[3], r[2];
    for (int i = 0; i < 6; ++i) c[i] = j_var(cam_rt[i], i, N);
    for (int i = 0; i < 6; ++i) b[i] = j_var(board_rt[i], 6 + i, N);
    for (int i = 0; i < 9; ++i) I[i] = j_var(intr[i], 12 + i, N);
    p[0] = j_const(board_pt[0], N); p[1] = j_const(board_pt[1], N); p[2] = j_const(0.0, N);
    aa_rotate_jet(b, p, Pw, N);
    for (int i = 0; i < 3; ++i) Pw[i] = j_add(Pw[i], b[3 + i], N);
    aa_rotate_jet(c, Pw, Pc, N);
    for (int i = 0; i < 3; ++i) Pc[i] = j_add(Pc[i], c[3 + i], N);
    ts_tail_jet(I, Pc, obs, r, N);
    for (int k = 0; k < 2; ++k) {
        if (res) res[k] = r[k].a;
        if (J_cam) for (int i = 0; i < 6; ++i) J_cam[k * 6 + i] = r[k].v[i];
        if (J_board) for (int i = 0; i < 6; ++i) J_board[k * 6 + i] = r[k].v[6 + i];
        if (J_intr) for (int i = 0; i < 9; ++i) J_intr[k * 9 + i] = r[k].v[12 + i];
    }
}

/* ======================================================================== *
 *  Plain projection / unprojection (with skew terms)
 * ======================================================================== */
/* TS.cpp:332-344 */
void orc_project(const double *I, const double *P, double *uv)
{
    const double X = P[0], Y = P[1], Z = P[2];
    const double fx = I[0], fy = I[1], cx = I[2], cy = I[3], xi = I[4], lamda = I[5], alpha = I[6], b = I[7], c = I[8];
    const double d1 = sqrt(X * X + Y * Y + Z * Z);
    const double d2 = sqrt(X * X + Y * Y + pow(Z + xi * d1, 2));
    const double d3 = sqrt(X * X + Y * Y + pow(Z + xi * d1 + lamda * d2, 2));
    const double ksai = Z + xi * d1 + lamda * d2 + alpha / (1 - alpha) * d3;
    uv[0] = fx * X / ksai + b * Y / ksai + cx;
    uv[1] = c * X / ksai + fy * Y / ksai + cy;
}

/* TS.h:39-57, transform = identity */
void orc_unproject(const double *I, const double *uv, double *ray)
{
    const double fx = I[0], fy = I[1], cx = I[2], cy = I[3], xi = I[4], lamda = I[5], alpha = I[6], b = I[7], c = I[8];
    double x = uv[0] - cx;
    double y = uv[1] - cy;
    const double mx = (fy * x - b * y) / (fx * fy - b * c);
    const double my = (-c * x + fx * y) / (fx * fy - b * c);
    const double ksai = alpha / (1 - alpha);
    const double r_square = mx * mx + my * my;
    const double gamma = (ksai + sqrt(1 + (1 - ksai * ksai) * r_square)) / (r_square + 1);
    const double yita = lamda * (gamma - ksai) + sqrt(((gamma - ksai) * (gamma - ksai) - 1) * lamda * lamda + 1);
    const double mz = yita * (gamma - ksai);
    const double mu = xi * (mz - lamda) + sqrt(xi * xi * ((mz - lamda) * (mz - lamda) - 1) + 1);
    ray[0] = mu * yita * gamma * mx;
    ray[1] = mu * yita * gamma * my;
    ray[2] = mu * (mz - lamda) - xi;
}

/* ======================================================================== *
 *  Optional multi-threading (OpenMP).  The reference never sets
 *  Solver::Options::num_threads (TS.cpp:271-274, multi_calib.cpp:209-212), so
 *  the checker and the 1-core CPU baseline run with ONE thread -- that path
 *  is the sequential code below, untouched.  orc_set_num_threads(n > 1)
 *  switches the O(N) passes to OpenMP loops (what Ceres does with
 *  num_threads = n: evaluation over residual blocks, Schur elimination over
 *  e-blocks with per-thread accumulators): the "all host cores" baseline of
 *  SURVEY 8d.  Sums are then associated per thread / per view, so results
 *  agree with the sequential path to rounding, not bit for bit.
 * ======================================================================== */
#ifdef _OPENMP
#include <omp.h>
#endif
static int g_orc_threads = 1;
void orc_set_num_threads(int n) { g_orc_threads = n < 1 ? 1 : n; }
int orc_get_num_threads(void) { return g_orc_threads; }
int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ======================================================================== *
 *  Batched evaluation
 * ======================================================================== */
static int total_corners(const orc_problem *p)
{
    int n = 0;
    for (int v = 0; v < p->n_views; ++v) n += p->view_count[v];
    return n;
}

/* Evaluate into caller arrays indexed by the *program order* used by the solver:
 * views in the given order, corners in order.  Row index = position in obs arrays is
 * NOT assumed contiguous, so outputs are indexed by a running counter. */
static double evaluate_view(const orc_problem *p, int v, long k, int use_jets, double *residuals,
                            double *J_cam, double *J_board, double *J_intr, double cost)
{
    const int m = p->view_camera[v], b = p->view_board[v];
    const double *crt = p->cam_rt ? p->cam_rt + 6 * m : NULL;
    const double *brt = p->board_rt + 6 * b;
    const double *I = p->intr + 9 * m;
    for (int j = 0; j < p->view_count[v]; ++j, ++k) {
        const double obs[2] = { p->obs_u[p->view_offset[v] + j], p->obs_v[p->view_offset[v] + j] };
        const double *bp = p->board_xy + 2 * j;
        double r[2];
        if (!use_jets) {
            if (p->mono) orc_mono_residual(I, brt, obs, bp, r);
            else orc_multi_residual(crt, brt, I, obs, bp, r);
        } else if (p->mono) {
            orc_mono_autodiff(I, brt, obs, bp, r, J_intr ? J_intr + 18 * k : NULL, J_board ? J_board + 12 * k : NULL);
            if (J_cam) memset(J_cam + 12 * k, 0, 12 * sizeof(double));
        } else {
            orc_multi_autodiff(crt, brt, I, obs, bp, r, J_cam ? J_cam + 12 * k : NULL,
                               J_board ? J_board + 12 * k : NULL, J_intr ? J_intr + 18 * k : NULL);
        }
        if (residuals) { residuals[2 * k] = r[0]; residuals[2 * k + 1] = r[1]; }
        /* ResidualBlock::Evaluate: cost = 0.5 * squaredNorm; evaluator sums block costs */
        cost += 0.5 * (r[0] * r[0] + r[1] * r[1]);
    }
    return cost;
}

double orc_evaluate(const orc_problem *p, int use_jets, double *residuals,
                    double *J_cam, double *J_board, double *J_intr)
{
    if (g_orc_threads > 1 && p->n_views > 1) {
        long *row = (long *)malloc(sizeof(long) * (size_t)p->n_views);
        double *part = (double *)malloc(sizeof(double) * (size_t)p->n_views);
        if (!row || !part) { fprintf(stderr, "oracle: out of memory\n"); abort(); }
        { long k = 0; for (int v = 0; v < p->n_views; ++v) { row[v] = k; k += p->view_count[v]; } }
#pragma omp parallel for schedule(static) num_threads(g_orc_threads)
        for (int v = 0; v < p->n_views; ++v) part[v] = evaluate_view(p, v, row[v], use_jets, residuals, J_cam, J_board, J_intr, 0.0);
        double cost = 0.0;
        for (int v = 0; v < p->n_views; ++v) cost += part[v];
        free(row); free(part);
        return cost;
    }
    double cost = 0.0;
    long k = 0;
    for (int v = 0; v < p->n_views; ++v) {
        cost = evaluate_view(p, v, k, use_jets, residuals, J_cam, J_board, J_intr, cost);     /* one running sum, program order */
        k += p->view_count[v];
    }
    return cost;
}

/* ======================================================================== *
 *  Small dense helpers (Eigen LLT stand-ins)
 * ======================================================================== */
/* In-place lower Cholesky of a row-major n x n SPD matrix (reads lower+diag).
 * Returns 0 on success, -1 if a pivot is not positive (Eigen::NumericalIssue). */
static int chol_lower(double *A, int n)
{
    for (int j = 0; j < n; ++j) {
        double d = A[j * n + j];
        for (int k = 0; k < j; ++k) d -= A[j * n + k] * A[j * n + k];
        if (!(d > 0.0)) return -1;
        d = sqrt(d);
        A[j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = A[i * n + j];
            for (int k = 0; k < j; ++k) s -= A[i * n + k] * A[j * n + k];
            A[i * n + j] = s / d;
        }
    }
    return 0;
}

static void chol_solve(const double *L, int n, double *x)
{
    for (int i = 0; i < n; ++i) {
        double s = x[i];
        for (int k = 0; k < i; ++k) s -= L[i * n + k] * x[k];
        x[i] = s / L[i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = x[i];
        for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * x[k];
        x[i] = s / L[i * n + i];
    }
}

/* InvertPSDMatrix<6>: m.selfadjointView<Upper>().llt().solve(Identity) */
static int invert_psd6(const double *M, double *inv)
{
    double L[36];
    memcpy(L, M, sizeof(L));
    if (chol_lower(L, 6) != 0) return -1;
    for (int c = 0; c < 6; ++c) {
        double e[6] = { 0, 0, 0, 0, 0, 0 };
        e[c] = 1.0;
        chol_solve(L, 6, e);
        for (int r = 0; r < 6; ++r) inv[r * 6 + c] = e[r];
    }
    return 0;
}

/* ======================================================================== *
 *  Options
 * ======================================================================== */
void orc_default_options(orc_options *o, int mono)
{
    o->max_num_iterations = mono ? 100 : 50;   /* TS.cpp:274 ; Ceres default (multi_calib.cpp:212 commented out) */
    o->function_tolerance = 1e-6;
    o->gradient_tolerance = 1e-10;
    o->parameter_tolerance = 1e-8;
    o->initial_trust_region_radius = 1e4;
    o->max_trust_region_radius = 1e16;
    o->min_trust_region_radius = 1e-32;
    o->min_relative_decrease = 1e-3;
    o->min_lm_diagonal = 1e-6;
    o->max_lm_diagonal = 1e32;
    o->max_num_consecutive_invalid_steps = 5;
    o->jacobi_scaling = 1;
}

/* ======================================================================== *
 *  The minimiser: ceres::Solve with TRUST_REGION / LEVENBERG_MARQUARDT /
 *  DENSE_SCHUR (external), driven exactly as TS.cpp:247-282 and
 *  multi_calib.cpp:155-218 configure it.
 *
 *  Reduced program: parameter blocks that appear in no residual block or are
 *  constant (multi_calib.cpp:186) are removed.  Schur ordering: the e-blocks
 *  are the board poses (mutually independent; every residual touches exactly
 *  one -- multi_calib.cpp:182-184, TS.cpp:266-267); f-blocks are the camera
 *  poses followed by the intrinsics (all 9 columns, b and c included with
 *  structurally zero Jacobian columns).
 * ======================================================================== */
typedef struct {
    const orc_problem *p;
    const orc_options *o;
    int C, B, N;
    /* program structure */
    int *cam_pose_col;   /* [C] column offset of the camera pose in the f-part, -1 if constant/inactive */
    int *intr_col;       /* [C] column offset of the intrinsics in the f-part, -1 if inactive */
    int *board_active;   /* [B] the board's pose is a free e-block: it has views and is not constant */
    int *board_seen;     /* [B] the board has views (its residual blocks exist even when its pose is constant) */
    int nf;              /* reduced-system size */
    int *bv_ptr, *bv_idx;/* board -> views adjacency (CSR) */
    long *view_row;      /* [n_views] first corner row (program order) of each view */
    /* state */
    double *x_cam, *x_intr, *x_board;       /* current point  */
    double *c_cam, *c_intr, *c_board;       /* candidate      */
    double *res;                            /* [2N] */
    double *Jc, *Jb, *Ji;                   /* [12N],[12N],[18N] (scaled in place like ScaleColumns) */
    double *g_f, *g_b;                      /* gradient (unscaled): [nf], [6B] */
    double *s_f, *s_b;                      /* jacobi scaling */
    double *d_f, *d_b;                      /* LM diagonal^2 source: clamped squared column norms */
    double *step_f, *step_b;                /* trust_region_step (scaled space) */
    double *lhs, *rhs;                      /* reduced system */
    double *inv_ete;                        /* [36B] */
} lm_state;

static void *xcalloc(size_t n, size_t s) { void *q = calloc(n ? n : 1, s); if (!q) { fprintf(stderr, "oracle: out of memory\n"); abort(); } return q; }

/* Evaluate at (cam,intr,board) -> cost; optionally residuals+Jacobians (jets). */
static double lm_eval(lm_state *S, const double *cam, const double *intr, const double *board, int with_jac)
{
    orc_problem q = *S->p;
    q.cam_rt = (double *)cam; q.intr = (double *)intr; q.board_rt = (double *)board;
    if (with_jac) return orc_evaluate(&q, 1, S->res, S->Jc, S->Jb, S->Ji);
    return orc_evaluate(&q, 0, NULL, NULL, NULL, NULL);
}

/* gradient = J^T r (unscaled Jacobian) */
static void gradient_view(lm_state *S, int v, double *g_f, double *g_b)
{
    const orc_problem *p = S->p;
    const int m = p->view_camera[v], b = p->view_board[v];
    const int cc = S->cam_pose_col[m], ic = S->intr_col[m];
    for (int j = 0; j < p->view_count[v]; ++j) {
        const long k = S->view_row[v] + j;
        for (int r = 0; r < 2; ++r) {
            const double rr = S->res[2 * k + r];
            if (cc >= 0) for (int i = 0; i < 6; ++i) g_f[cc + i] += S->Jc[12 * k + 6 * r + i] * rr;
            for (int i = 0; i < 9; ++i) g_f[ic + i] += S->Ji[18 * k + 9 * r + i] * rr;
            for (int i = 0; i < 6; ++i) g_b[6 * b + i] += S->Jb[12 * k + 6 * r + i] * rr;
        }
    }
}

/* threads > 1: boards are dealt to the threads in contiguous blocks (a board's rows are written by one thread), the
 * f-part is accumulated per thread and the per-thread sums are added in thread order */
#define ORC_MT_BOARD_LOOP(NF_ACC, BODY)                                                          \
    do {                                                                                         \
        const int T_ = g_orc_threads;                                                            \
        double *acc_ = (double *)xcalloc((size_t)T_ * (size_t)(NF_ACC), sizeof(double));         \
        _Pragma("omp parallel num_threads(T_)")                                                  \
        {                                                                                        \
            int t_ = 0;                                                                          \
            ORC_THREAD_ID(t_);                                                                   \
            double *accf = acc_ + (size_t)t_ * (size_t)(NF_ACC);                                 \
            _Pragma("omp for schedule(static)")                                                  \
            for (int b = 0; b < S->B; ++b)                                                       \
                for (int q_ = S->bv_ptr[b]; q_ < S->bv_ptr[b + 1]; ++q_) { const int v = S->bv_idx[q_]; BODY; } \
        }                                                                                        \
        for (int t_ = 0; t_ < T_; ++t_) for (int i_ = 0; i_ < (NF_ACC); ++i_) ORC_MT_OUT[i_] += acc_[(size_t)t_ * (size_t)(NF_ACC) + i_]; \
        free(acc_);                                                                              \
    } while (0)
#ifdef _OPENMP
#define ORC_THREAD_ID(t) (t) = omp_get_thread_num()
#else
#define ORC_THREAD_ID(t) (t) = 0
#endif

static void lm_gradient(lm_state *S)
{
    const orc_problem *p = S->p;
    memset(S->g_f, 0, sizeof(double) * (size_t)S->nf);
    memset(S->g_b, 0, sizeof(double) * 6 * (size_t)S->B);
    if (g_orc_threads > 1) {
#define ORC_MT_OUT S->g_f
        ORC_MT_BOARD_LOOP(S->nf, gradient_view(S, v, accf, S->g_b));
#undef ORC_MT_OUT
        return;
    }
    for (int v = 0; v < p->n_views; ++v) gradient_view(S, v, S->g_f, S->g_b);
}

/* SquaredColumnNorm of the Jacobian currently stored */
static void sq_col_norm_view(lm_state *S, int v, double *nf, double *nb)
{
    const orc_problem *p = S->p;
    const int m = p->view_camera[v], b = p->view_board[v];
    const int cc = S->cam_pose_col[m], ic = S->intr_col[m];
    for (int j = 0; j < p->view_count[v]; ++j) {
        const long k = S->view_row[v] + j;
        for (int r = 0; r < 2; ++r) {
            if (cc >= 0) for (int i = 0; i < 6; ++i) { const double a = S->Jc[12 * k + 6 * r + i]; nf[cc + i] += a * a; }
            for (int i = 0; i < 9; ++i) { const double a = S->Ji[18 * k + 9 * r + i]; nf[ic + i] += a * a; }
            for (int i = 0; i < 6; ++i) { const double a = S->Jb[12 * k + 6 * r + i]; nb[6 * b + i] += a * a; }
        }
    }
}

static void lm_sq_col_norm(lm_state *S, double *nf, double *nb)
{
    const orc_problem *p = S->p;
    memset(nf, 0, sizeof(double) * (size_t)S->nf);
    memset(nb, 0, sizeof(double) * 6 * (size_t)S->B);
    if (g_orc_threads > 1) {
#define ORC_MT_OUT nf
        ORC_MT_BOARD_LOOP(S->nf, sq_col_norm_view(S, v, accf, nb));
#undef ORC_MT_OUT
        return;
    }
    for (int v = 0; v < p->n_views; ++v) sq_col_norm_view(S, v, nf, nb);
}

/* jacobian->ScaleColumns(jacobian_scaling) */
static void lm_scale_columns(lm_state *S)
{
    const orc_problem *p = S->p;
#pragma omp parallel for schedule(static) num_threads(g_orc_threads) if (g_orc_threads > 1)
    for (int v = 0; v < p->n_views; ++v) {
        const int m = p->view_camera[v], b = p->view_board[v];
        const int cc = S->cam_pose_col[m], ic = S->intr_col[m];
        for (int j = 0; j < p->view_count[v]; ++j) {
            const long k = S->view_row[v] + j;
            for (int r = 0; r < 2; ++r) {
                if (cc >= 0) for (int i = 0; i < 6; ++i) S->Jc[12 * k + 6 * r + i] *= S->s_f[cc + i];
                for (int i = 0; i < 9; ++i) S->Ji[18 * k + 9 * r + i] *= S->s_f[ic + i];
                for (int i = 0; i < 6; ++i) S->Jb[12 * k + 6 * r + i] *= S->s_b[6 * b + i];
            }
        }
    }
}

/* SchurEliminator::Eliminate + dense Cholesky of the reduced system +
 * SchurEliminator::BackSubstitute.  Solves (J^T J + D^T D) y = J^T r for the scaled
 * Jacobian; y goes to step_f / step_b (sign flipped by the caller).
 * Returns 0 ok, -1 LINEAR_SOLVER_FAILURE. */
/* Elimination of ONE e-block (board b): accumulates F^T F, F^T b and the Schur complement terms into lhs / rhs.
 * `buffer` [6 * nf] and `touched` [C] are scratch.  Returns 0, or -1 if E^T E + D^2 is not positive definite. */
static int schur_eliminate_board(lm_state *S, int b, double radius, double *lhs, double *rhs, double *buffer, int *touched)
{
    const orc_problem *p = S->p;
    const int nf = S->nf;
    if (!S->board_seen[b]) return 0;
    if (!S->board_active[b]) {
        /* constant pose block (removed from the program): its residual blocks only have f-columns */
        for (int q = S->bv_ptr[b]; q < S->bv_ptr[b + 1]; ++q) {
            const int v = S->bv_idx[q];
            const int m = p->view_camera[v];
            const int cc = S->cam_pose_col[m], ic = S->intr_col[m];
            for (int j = 0; j < p->view_count[v]; ++j) {
                const long k = S->view_row[v] + j;
                for (int r = 0; r < 2; ++r) {
                    const double *Fc = S->Jc + 12 * k + 6 * r;
                    const double *Fi = S->Ji + 18 * k + 9 * r;
                    const double rr = S->res[2 * k + r];
                    if (cc >= 0) {
                        for (int i = 0; i < 6; ++i) {
                            for (int l = 0; l < 6; ++l) lhs[(cc + i) * nf + cc + l] += Fc[i] * Fc[l];
                            for (int l = 0; l < 9; ++l) { const double t = Fc[i] * Fi[l]; lhs[(cc + i) * nf + ic + l] += t; lhs[(ic + l) * nf + cc + i] += t; }
                            rhs[cc + i] += Fc[i] * rr;
                        }
                    }
                    for (int i = 0; i < 9; ++i) {
                        for (int l = 0; l < 9; ++l) lhs[(ic + i) * nf + ic + l] += Fi[i] * Fi[l];
                        rhs[ic + i] += Fi[i] * rr;
                    }
                }
            }
        }
        return 0;
    }
    double ete[36], g[6];
    memset(ete, 0, sizeof(ete)); memset(g, 0, sizeof(g));
    for (int i = 0; i < 6; ++i) { const double D = sqrt(S->d_b[6 * b + i] / radius); ete[i * 6 + i] = D * D; }
    int nt = 0;
    for (int q = S->bv_ptr[b]; q < S->bv_ptr[b + 1]; ++q) {
        const int v = S->bv_idx[q];
        const int m = p->view_camera[v];
        const int cc = S->cam_pose_col[m], ic = S->intr_col[m];
        touched[nt++] = m;
        if (cc >= 0) for (int i = 0; i < 6; ++i) memset(buffer + i * nf + cc, 0, 6 * sizeof(double));
        for (int i = 0; i < 6; ++i) memset(buffer + i * nf + ic, 0, 9 * sizeof(double));
    }
    for (int q = S->bv_ptr[b]; q < S->bv_ptr[b + 1]; ++q) {
        const int v = S->bv_idx[q];
        const int m = p->view_camera[v];
        const int cc = S->cam_pose_col[m], ic = S->intr_col[m];
        for (int j = 0; j < p->view_count[v]; ++j) {
            const long k = S->view_row[v] + j;
            for (int r = 0; r < 2; ++r) {
                const double *E = S->Jb + 12 * k + 6 * r;
                const double *Fc = S->Jc + 12 * k + 6 * r;
                const double *Fi = S->Ji + 18 * k + 9 * r;
                const double rr = S->res[2 * k + r];
                for (int i = 0; i < 6; ++i) {
                    for (int l = 0; l < 6; ++l) ete[i * 6 + l] += E[i] * E[l];
                    g[i] += E[i] * rr;
                    if (cc >= 0) for (int l = 0; l < 6; ++l) buffer[i * nf + cc + l] += E[i] * Fc[l];
                    for (int l = 0; l < 9; ++l) buffer[i * nf + ic + l] += E[i] * Fi[l];
                }
                /* lhs += F^T F ; rhs += F^T b */
                if (cc >= 0) {
                    for (int i = 0; i < 6; ++i) {
                        for (int l = 0; l < 6; ++l) lhs[(cc + i) * nf + cc + l] += Fc[i] * Fc[l];
                        for (int l = 0; l < 9; ++l) { const double t = Fc[i] * Fi[l]; lhs[(cc + i) * nf + ic + l] += t; lhs[(ic + l) * nf + cc + i] += t; }
                        rhs[cc + i] += Fc[i] * rr;
                    }
                }
                for (int i = 0; i < 9; ++i) {
                    for (int l = 0; l < 9; ++l) lhs[(ic + i) * nf + ic + l] += Fi[i] * Fi[l];
                    rhs[ic + i] += Fi[i] * rr;
                }
            }
        }
    }
    double *inv = S->inv_ete + 36 * b;
    if (invert_psd6(ete, inv) != 0) return -1;
    /* lhs -= buffer^T inv buffer ; rhs -= buffer^T inv g   (only over touched f-blocks) */
    double ig[6];
    for (int i = 0; i < 6; ++i) { double s = 0; for (int l = 0; l < 6; ++l) s += inv[i * 6 + l] * g[l]; ig[i] = s; }
    for (int t1 = 0; t1 < nt; ++t1) {
        const int m1 = touched[t1];
        for (int part1 = 0; part1 < 2; ++part1) {
            const int c1 = part1 ? S->intr_col[m1] : S->cam_pose_col[m1];
            const int w1 = part1 ? 9 : 6;
            if (c1 < 0) continue;
            for (int a = 0; a < w1; ++a) {
                double ib[6]; /* inv * buffer[:, c1+a] */
                for (int i = 0; i < 6; ++i) { double s = 0; for (int l = 0; l < 6; ++l) s += inv[i * 6 + l] * buffer[l * nf + c1 + a]; ib[i] = s; }
                double sg = 0; for (int i = 0; i < 6; ++i) sg += buffer[i * nf + c1 + a] * ig[i];
                rhs[c1 + a] -= sg;
                for (int t2 = 0; t2 < nt; ++t2) {
                    const int m2 = touched[t2];
                    for (int part2 = 0; part2 < 2; ++part2) {
                        const int c2 = part2 ? S->intr_col[m2] : S->cam_pose_col[m2];
                        const int w2 = part2 ? 9 : 6;
                        if (c2 < 0) continue;
                        for (int bcol = 0; bcol < w2; ++bcol) {
                            double s = 0; for (int i = 0; i < 6; ++i) s += buffer[i * nf + c2 + bcol] * ib[i];
                            lhs[(c2 + bcol) * nf + c1 + a] -= s;
                        }
                    }
                }
            }
        }
    }
    return 0;
}

/* SchurEliminator::BackSubstitute for ONE e-block: y_e = inv_ete * sum_rows E^T (b - F z) */
static void schur_backsubstitute_board(lm_state *S, int b)
{
    const orc_problem *p = S->p;
    double *y = S->step_b + 6 * b;
    for (int i = 0; i < 6; ++i) y[i] = 0.0;
    if (!S->board_active[b]) return;
    double acc[6] = { 0, 0, 0, 0, 0, 0 };
    for (int q = S->bv_ptr[b]; q < S->bv_ptr[b + 1]; ++q) {
        const int v = S->bv_idx[q];
        const int m = p->view_camera[v];
        const int cc = S->cam_pose_col[m], ic = S->intr_col[m];
        for (int j = 0; j < p->view_count[v]; ++j) {
            const long k = S->view_row[v] + j;
            for (int r = 0; r < 2; ++r) {
                double sj = S->res[2 * k + r];
                if (cc >= 0) for (int l = 0; l < 6; ++l) sj -= S->Jc[12 * k + 6 * r + l] * S->step_f[cc + l];
                for (int l = 0; l < 9; ++l) sj -= S->Ji[18 * k + 9 * r + l] * S->step_f[ic + l];
                for (int i = 0; i < 6; ++i) acc[i] += S->Jb[12 * k + 6 * r + i] * sj;
            }
        }
    }
    const double *inv = S->inv_ete + 36 * b;
    for (int i = 0; i < 6; ++i) { double s = 0; for (int l = 0; l < 6; ++l) s += inv[i * 6 + l] * acc[l]; y[i] = s; }
}

/* SchurEliminator::Eliminate + dense Cholesky of the reduced system +
 * SchurEliminator::BackSubstitute.  Solves (J^T J + D^T D) y = J^T r for the scaled
 * Jacobian; y goes to step_f / step_b (sign flipped by the caller).
 * Returns 0 ok, -1 LINEAR_SOLVER_FAILURE. */
static int lm_schur_solve(lm_state *S, double radius)
{
    const orc_problem *p = S->p;
    const int nf = S->nf;
    memset(S->lhs, 0, sizeof(double) * (size_t)nf * nf);
    memset(S->rhs, 0, sizeof(double) * (size_t)nf);
    /* lm_diagonal = sqrt(diagonal / radius); D^T D = diagonal / radius */
    for (int i = 0; i < nf; ++i) { const double D = sqrt(S->d_f[i] / radius); S->lhs[i * nf + i] = D * D; }

    if (g_orc_threads > 1) {
        /* e-blocks dealt to the threads in contiguous blocks, per-thread lhs / rhs added in thread order */
        const int T = g_orc_threads;
        const size_t stride = (size_t)nf * nf + (size_t)nf;
        double *acc = (double *)xcalloc((size_t)T * stride, sizeof(double));
        int failed = 0;
#pragma omp parallel num_threads(T)
        {
            int t = 0;
            ORC_THREAD_ID(t);
            double *buffer = (double *)xcalloc((size_t)6 * nf, sizeof(double));
            int *touched = (int *)xcalloc((size_t)p->n_cameras, sizeof(int));
            double *lhs = acc + (size_t)t * stride, *rhs = lhs + (size_t)nf * nf;
#pragma omp for schedule(static)
            for (int b = 0; b < S->B; ++b)
                if (schur_eliminate_board(S, b, radius, lhs, rhs, buffer, touched) != 0) {
#pragma omp atomic write
                    failed = 1;
                }
            free(buffer); free(touched);
        }
        for (int t = 0; t < T; ++t) {
            const double *lhs = acc + (size_t)t * stride, *rhs = lhs + (size_t)nf * nf;
            for (size_t i = 0; i < (size_t)nf * nf; ++i) S->lhs[i] += lhs[i];
            for (int i = 0; i < nf; ++i) S->rhs[i] += rhs[i];
        }
        free(acc);
        if (failed) return -1;
    } else {
        double *buffer = (double *)xcalloc((size_t)6 * nf, sizeof(double)); /* E^T F, dense over f-columns */
        int *touched = (int *)xcalloc((size_t)p->n_cameras, sizeof(int));
        for (int b = 0; b < S->B; ++b)
            if (schur_eliminate_board(S, b, radius, S->lhs, S->rhs, buffer, touched) != 0) { free(buffer); free(touched); return -1; }
        free(buffer); free(touched);
    }

    /* DenseSchurComplementSolver::SolveReducedLinearSystem: Eigen LLT */
    double *L = (double *)xcalloc((size_t)nf * nf, sizeof(double));
    memcpy(L, S->lhs, sizeof(double) * (size_t)nf * nf);
    if (chol_lower(L, nf) != 0) { free(L); return -1; }
    memcpy(S->step_f, S->rhs, sizeof(double) * (size_t)nf);
    chol_solve(L, nf, S->step_f);
    free(L);

#pragma omp parallel for schedule(static) num_threads(g_orc_threads) if (g_orc_threads > 1)
    for (int b = 0; b < S->B; ++b) schur_backsubstitute_board(S, b);
    return 0;
}

/* model_cost_change = -(J step)^T (r + J step / 2)  (scaled J, scaled step) */
static double model_cost_change_view(lm_state *S, int v, double acc)
{
    const orc_problem *p = S->p;
    const int m = p->view_camera[v], b = p->view_board[v];
    const int cc = S->cam_pose_col[m], ic = S->intr_col[m];
    for (int j = 0; j < p->view_count[v]; ++j) {
        const long k = S->view_row[v] + j;
        for (int r = 0; r < 2; ++r) {
            double mr = 0.0;
            if (cc >= 0) for (int l = 0; l < 6; ++l) mr += S->Jc[12 * k + 6 * r + l] * S->step_f[cc + l];
            for (int l = 0; l < 9; ++l) mr += S->Ji[18 * k + 9 * r + l] * S->step_f[ic + l];
            for (int l = 0; l < 6; ++l) mr += S->Jb[12 * k + 6 * r + l] * S->step_b[6 * b + l];
            acc += mr * (S->res[2 * k + r] + mr / 2.0);
        }
    }
    return acc;
}

static double lm_model_cost_change(lm_state *S)
{
    const orc_problem *p = S->p;
    double acc = 0.0;
    if (g_orc_threads > 1) {
        double *part = (double *)xcalloc((size_t)p->n_views, sizeof(double));
#pragma omp parallel for schedule(static) num_threads(g_orc_threads)
        for (int v = 0; v < p->n_views; ++v) part[v] = model_cost_change_view(S, v, 0.0);
        for (int v = 0; v < p->n_views; ++v) acc += part[v];
        free(part);
        return -acc;
    }
    for (int v = 0; v < p->n_views; ++v) acc = model_cost_change_view(S, v, acc);
    return -acc;
}

static int all_finite(const double *x, long n) { for (long i = 0; i < n; ++i) if (!isfinite(x[i])) return 0; return 1; }

/* norms over the reduced program's parameter vector */
static double prog_norm_diff(lm_state *S, const double *acam, const double *aintr, const double *aboard,
                             const double *bcam, const double *bintr, const double *bboard, int maxnorm)
{
    double acc = 0.0;
    for (int m = 0; m < S->C; ++m) {
        if (S->cam_pose_col[m] >= 0) for (int i = 0; i < 6; ++i) { const double d = acam[6 * m + i] - (bcam ? bcam[6 * m + i] : 0.0); if (maxnorm) { if (fabs(d) > acc) acc = fabs(d); } else acc += d * d; }
        if (S->intr_col[m] >= 0) for (int i = 0; i < 9; ++i) { const double d = aintr[9 * m + i] - (bintr ? bintr[9 * m + i] : 0.0); if (maxnorm) { if (fabs(d) > acc) acc = fabs(d); } else acc += d * d; }
    }
    for (int b = 0; b < S->B; ++b) {
        if (!S->board_active[b]) continue;
        for (int i = 0; i < 6; ++i) { const double d = aboard[6 * b + i] - (bboard ? bboard[6 * b + i] : 0.0); if (maxnorm) { if (fabs(d) > acc) acc = fabs(d); } else acc += d * d; }
    }
    return maxnorm ? acc : sqrt(acc);
}

int orc_solve(const orc_problem *p, const orc_options *opt, orc_summary *sum)
{
    lm_state S;
    memset(&S, 0, sizeof(S));
    memset(sum, 0, sizeof(*sum));
    S.p = p; S.o = opt; S.C = p->n_cameras; S.B = p->n_boards; S.N = total_corners(p);
    const int C = S.C, B = S.B; const long N = S.N;
    sum->n_residual_blocks = (int)N;
    if (p->mono && C != 1) { snprintf(sum->message, sizeof(sum->message), "mono problem needs exactly one camera"); sum->termination_type = ORC_FAILURE; return -1; }

    /* ---- reduced program ------------------------------------------------*/
    S.cam_pose_col = (int *)xcalloc((size_t)C, sizeof(int));
    S.intr_col = (int *)xcalloc((size_t)C, sizeof(int));
    S.board_active = (int *)xcalloc((size_t)B, sizeof(int));
    S.board_seen = (int *)xcalloc((size_t)B, sizeof(int));
    int *cam_active = (int *)xcalloc((size_t)C, sizeof(int));
    S.view_row = (long *)xcalloc((size_t)p->n_views, sizeof(long));
    S.bv_ptr = (int *)xcalloc((size_t)B + 1, sizeof(int));
    S.bv_idx = (int *)xcalloc((size_t)p->n_views, sizeof(int));
    { long k = 0; for (int v = 0; v < p->n_views; ++v) { S.view_row[v] = k; k += p->view_count[v]; if (p->view_count[v] > 0) { cam_active[p->view_camera[v]] = 1; S.board_seen[p->view_board[v]] = 1; S.bv_ptr[p->view_board[v] + 1]++; } } }
    for (int b = 0; b < B; ++b) S.board_active[b] = S.board_seen[b] && !(p->board_pose_constant && p->board_pose_constant[b]);
    for (int b = 0; b < B; ++b) S.bv_ptr[b + 1] += S.bv_ptr[b];
    { int *fill = (int *)xcalloc((size_t)B, sizeof(int)); for (int v = 0; v < p->n_views; ++v) if (p->view_count[v] > 0) { const int b = p->view_board[v]; S.bv_idx[S.bv_ptr[b] + fill[b]++] = v; } free(fill); }
    int col = 0;
    for (int m = 0; m < C; ++m) {
        const int constant = p->mono || (p->cam_pose_constant && p->cam_pose_constant[m]);
        S.cam_pose_col[m] = (cam_active[m] && !constant) ? col : -1;
        if (S.cam_pose_col[m] >= 0) col += 6;
    }
    for (int m = 0; m < C; ++m) { S.intr_col[m] = cam_active[m] ? col : -1; if (cam_active[m]) col += 9; }
    S.nf = col;
    const int nf = S.nf;

    /* ---- storage ---------------------------------------------------------*/
    double *zero_cam = (double *)xcalloc((size_t)6 * C, sizeof(double));
    S.x_cam = (double *)xcalloc((size_t)6 * C, sizeof(double)); S.c_cam = (double *)xcalloc((size_t)6 * C, sizeof(double));
    S.x_intr = (double *)xcalloc((size_t)9 * C, sizeof(double)); S.c_intr = (double *)xcalloc((size_t)9 * C, sizeof(double));
    S.x_board = (double *)xcalloc((size_t)6 * B, sizeof(double)); S.c_board = (double *)xcalloc((size_t)6 * B, sizeof(double));
    memcpy(S.x_cam, p->cam_rt ? p->cam_rt : zero_cam, sizeof(double) * 6 * (size_t)C);
    memcpy(S.x_intr, p->intr, sizeof(double) * 9 * (size_t)C);
    memcpy(S.x_board, p->board_rt, sizeof(double) * 6 * (size_t)B);
    S.res = (double *)xcalloc((size_t)2 * N, sizeof(double));
    S.Jc = (double *)xcalloc((size_t)12 * N, sizeof(double));
    S.Jb = (double *)xcalloc((size_t)12 * N, sizeof(double));
    S.Ji = (double *)xcalloc((size_t)18 * N, sizeof(double));
    S.g_f = (double *)xcalloc((size_t)nf, sizeof(double)); S.g_b = (double *)xcalloc((size_t)6 * B, sizeof(double));
    S.s_f = (double *)xcalloc((size_t)nf, sizeof(double)); S.s_b = (double *)xcalloc((size_t)6 * B, sizeof(double));
    S.d_f = (double *)xcalloc((size_t)nf, sizeof(double)); S.d_b = (double *)xcalloc((size_t)6 * B, sizeof(double));
    S.step_f = (double *)xcalloc((size_t)nf, sizeof(double)); S.step_b = (double *)xcalloc((size_t)6 * B, sizeof(double));
    S.lhs = (double *)xcalloc((size_t)nf * nf, sizeof(double)); S.rhs = (double *)xcalloc((size_t)nf, sizeof(double));
    S.inv_ete = (double *)xcalloc((size_t)36 * B, sizeof(double));
    for (int i = 0; i < nf; ++i) S.s_f[i] = 1.0;
    for (long i = 0; i < 6L * B; ++i) S.s_b[i] = 1.0;

    const double t_start = now_s();
    double t_jac = 0.0, t_lin = 0.0;

    /* ---- LevenbergMarquardtStrategy state --------------------------------*/
    double radius = opt->initial_trust_region_radius;
    double decrease_factor = 2.0;
    int reuse_diagonal = 0;

    /* ---- IterationZero / EvaluateGradientAndJacobian ---------------------*/
    orc_iteration it;
    memset(&it, 0, sizeof(it));
    double x_cost, candidate_cost = 0.0, model_cost_change = 0.0;
    double x_norm = prog_norm_diff(&S, S.x_cam, S.x_intr, S.x_board, NULL, NULL, NULL, 0);
    int num_consecutive_invalid_steps = 0;
    int term = -1;
    int iteration = 0;

#define EVAL_GRAD_JAC()                                                                         \
    do {                                                                                        \
        const double t0_ = now_s();                                                             \
        x_cost = lm_eval(&S, S.x_cam, S.x_intr, S.x_board, 1);                                  \
        lm_gradient(&S);                                                                        \
        if (opt->jacobi_scaling) {                                                              \
            if (iteration == 0) {                                                               \
                lm_sq_col_norm(&S, S.s_f, S.s_b);                                               \
                for (int i_ = 0; i_ < nf; ++i_) S.s_f[i_] = 1.0 / (1.0 + sqrt(S.s_f[i_]));     \
                for (long i_ = 0; i_ < 6L * B; ++i_) S.s_b[i_] = 1.0 / (1.0 + sqrt(S.s_b[i_]));\
            }                                                                                   \
            lm_scale_columns(&S);                                                               \
        }                                                                                       \
        /* |x - Plus(x, -gradient)| */                                                          \
        {                                                                                       \
            double gmax_ = 0.0, gsq_ = 0.0;                                                     \
            for (int m_ = 0; m_ < C; ++m_) {                                                    \
                if (S.cam_pose_col[m_] >= 0) for (int i_ = 0; i_ < 6; ++i_) { const double x_ = S.x_cam[6 * m_ + i_]; const double d_ = x_ - (x_ + (-S.g_f[S.cam_pose_col[m_] + i_])); if (fabs(d_) > gmax_) gmax_ = fabs(d_); gsq_ += d_ * d_; } \
                if (S.intr_col[m_] >= 0) for (int i_ = 0; i_ < 9; ++i_) { const double x_ = S.x_intr[9 * m_ + i_]; const double d_ = x_ - (x_ + (-S.g_f[S.intr_col[m_] + i_])); if (fabs(d_) > gmax_) gmax_ = fabs(d_); gsq_ += d_ * d_; } \
            }                                                                                   \
            for (int b_ = 0; b_ < B; ++b_) if (S.board_active[b_]) for (int i_ = 0; i_ < 6; ++i_) { const double x_ = S.x_board[6 * b_ + i_]; const double d_ = x_ - (x_ + (-S.g_b[6 * b_ + i_])); if (fabs(d_) > gmax_) gmax_ = fabs(d_); gsq_ += d_ * d_; } \
            it.gradient_max_norm = gmax_; it.gradient_norm = sqrt(gsq_);                        \
        }                                                                                       \
        it.cost = x_cost;                                                                       \
        t_jac += now_s() - t0_;                                                                 \
    } while (0)

    it.iteration = 0;
    EVAL_GRAD_JAC();
    sum->initial_cost = x_cost;
    it.step_is_valid = 1; it.step_is_successful = 1;

    /* TrustRegionStepEvaluator (monotonic: max_consecutive_nonmonotonic_steps = 0) */
    double se_minimum_cost = x_cost, se_current_cost = x_cost, se_reference_cost = x_cost, se_candidate_cost = x_cost;
    double se_acc_ref = 0.0, se_acc_cand = 0.0;

    for (;;) {
        /* ---- FinalizeIterationAndCheckIfMinimizerCanContinue --------------*/
        if (it.step_is_successful) {
            ++sum->num_successful_steps;
            /* x_cost < minimum_cost always holds for monotonic steps: publish x to the user */
        } else {
            ++sum->num_unsuccessful_steps;
        }
        it.trust_region_radius = radius;
        if (sum->num_iterations < 256) sum->iterations[sum->num_iterations] = it;
        ++sum->num_iterations;
        if (it.iteration >= opt->max_num_iterations) { term = ORC_NO_CONVERGENCE; snprintf(sum->message, sizeof(sum->message), "Maximum number of iterations reached."); break; }
        if (it.step_is_successful && it.gradient_max_norm <= opt->gradient_tolerance) { term = ORC_CONVERGENCE; snprintf(sum->message, sizeof(sum->message), "Gradient tolerance reached."); break; }
        if (radius <= opt->min_trust_region_radius) { term = ORC_CONVERGENCE; snprintf(sum->message, sizeof(sum->message), "Minimum trust region radius reached."); break; }

        const double prev_gmax = it.gradient_max_norm, prev_gnorm = it.gradient_norm;
        iteration = it.iteration + 1;
        memset(&it, 0, sizeof(it));
        it.iteration = iteration;

        /* ---- ComputeTrustRegionStep -------------------------------------*/
        if (!reuse_diagonal) {
            lm_sq_col_norm(&S, S.d_f, S.d_b);
            for (int i = 0; i < nf; ++i) S.d_f[i] = fmin(fmax(S.d_f[i], opt->min_lm_diagonal), opt->max_lm_diagonal);
            for (long i = 0; i < 6L * B; ++i) S.d_b[i] = fmin(fmax(S.d_b[i], opt->min_lm_diagonal), opt->max_lm_diagonal);
        }
        const double tl0 = now_s();
        int lin_ok = (lm_schur_solve(&S, radius) == 0);
        if (lin_ok && !(all_finite(S.step_f, nf) && all_finite(S.step_b, 6L * B))) lin_ok = 0;
        reuse_diagonal = 1;
        it.step_is_valid = 0;
        if (lin_ok) {
            for (int i = 0; i < nf; ++i) S.step_f[i] *= -1.0;
            for (long i = 0; i < 6L * B; ++i) S.step_b[i] *= -1.0;
            model_cost_change = lm_model_cost_change(&S);
            it.step_is_valid = (model_cost_change > 0.0);
        }
        t_lin += now_s() - tl0;

        if (!it.step_is_valid) {
            /* ---- HandleInvalidStep ----------------------------------------*/
            if (++num_consecutive_invalid_steps >= opt->max_num_consecutive_invalid_steps) {
                term = ORC_FAILURE; snprintf(sum->message, sizeof(sum->message), "Number of consecutive invalid steps more than Solver::Options::max_num_consecutive_invalid_steps."); break;
            }
            /* strategy_->StepIsInvalid() == StepRejected(0) */
            radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diagonal = 1;
            it.cost = x_cost; it.cost_change = 0.0; it.gradient_max_norm = prev_gmax; it.gradient_norm = prev_gnorm;
            it.step_norm = 0.0; it.relative_decrease = 0.0; it.step_is_successful = 0;
            continue;
        }
        num_consecutive_invalid_steps = 0;

        /* delta = step .* jacobian_scaling ; candidate = Plus(x, delta) */
        for (int m = 0; m < C; ++m) {
            for (int i = 0; i < 6; ++i) S.c_cam[6 * m + i] = S.x_cam[6 * m + i] + (S.cam_pose_col[m] >= 0 ? S.step_f[S.cam_pose_col[m] + i] * S.s_f[S.cam_pose_col[m] + i] : 0.0);
            for (int i = 0; i < 9; ++i) S.c_intr[9 * m + i] = S.x_intr[9 * m + i] + (S.intr_col[m] >= 0 ? S.step_f[S.intr_col[m] + i] * S.s_f[S.intr_col[m] + i] : 0.0);
        }
        for (int b = 0; b < B; ++b) for (int i = 0; i < 6; ++i) S.c_board[6 * b + i] = S.x_board[6 * b + i] + (S.board_active[b] ? S.step_b[6 * b + i] * S.s_b[6 * b + i] : 0.0);

        /* ---- ComputeCandidatePointAndEvaluateCost (double functor) --------*/
        { const double t0 = now_s(); candidate_cost = lm_eval(&S, S.c_cam, S.c_intr, S.c_board, 0); t_jac += now_s() - t0; }
        if (!isfinite(candidate_cost)) candidate_cost = DBL_MAX;

        /* ---- ParameterToleranceReached ----------------------------------*/
        it.step_norm = prog_norm_diff(&S, S.x_cam, S.x_intr, S.x_board, S.c_cam, S.c_intr, S.c_board, 0);
        it.gradient_max_norm = prev_gmax; it.gradient_norm = prev_gnorm;
        if (it.step_norm <= opt->parameter_tolerance * (x_norm + opt->parameter_tolerance)) {
            term = ORC_CONVERGENCE; snprintf(sum->message, sizeof(sum->message), "Parameter tolerance reached.");
            it.cost = x_cost; it.cost_change = x_cost - candidate_cost; it.trust_region_radius = radius;
            break;
        }
        /* ---- FunctionToleranceReached -----------------------------------*/
        it.cost_change = x_cost - candidate_cost;
        if (fabs(it.cost_change) <= opt->function_tolerance * x_cost) {
            term = ORC_CONVERGENCE; snprintf(sum->message, sizeof(sum->message), "Function tolerance reached.");
            it.cost = x_cost; it.trust_region_radius = radius;
            break;
        }
        /* ---- IsStepSuccessful (StepQuality) -----------------------------*/
        {
            double q;
            if (candidate_cost >= DBL_MAX) q = -DBL_MAX;
            else {
                const double rel = (se_current_cost - candidate_cost) / model_cost_change;
                const double hist = (se_reference_cost - candidate_cost) / (se_acc_ref + model_cost_change);
                q = rel > hist ? rel : hist;
            }
            it.relative_decrease = q;
        }
        if (it.relative_decrease > opt->min_relative_decrease) {
            /* ---- HandleSuccessfulStep -------------------------------------*/
            memcpy(S.x_cam, S.c_cam, sizeof(double) * 6 * (size_t)C);
            memcpy(S.x_intr, S.c_intr, sizeof(double) * 9 * (size_t)C);
            memcpy(S.x_board, S.c_board, sizeof(double) * 6 * (size_t)B);
            x_norm = prog_norm_diff(&S, S.x_cam, S.x_intr, S.x_board, NULL, NULL, NULL, 0);
            { const double sn = it.step_norm, cc = it.cost_change, rd = it.relative_decrease;
              EVAL_GRAD_JAC();
              it.step_norm = sn; it.cost_change = cc; it.relative_decrease = rd; }
            it.step_is_valid = 1; it.step_is_successful = 1;
            /* strategy_->StepAccepted(step_quality) */
            radius = radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * it.relative_decrease - 1.0, 3));
            radius = fmin(opt->max_trust_region_radius, radius);
            decrease_factor = 2.0; reuse_diagonal = 0;
            /* step_evaluator_->StepAccepted(candidate_cost, model_cost_change) */
            se_current_cost = candidate_cost; se_acc_cand += model_cost_change; se_acc_ref += model_cost_change;
            if (se_current_cost < se_minimum_cost) { se_minimum_cost = se_current_cost; se_candidate_cost = se_current_cost; se_acc_cand = 0.0; }
            else if (se_current_cost > se_candidate_cost) { se_candidate_cost = se_current_cost; se_acc_cand = 0.0; }
            se_reference_cost = se_candidate_cost; se_acc_ref = se_acc_cand;
        } else {
            it.step_is_successful = 0;
            it.cost = candidate_cost;
            /* strategy_->StepRejected */
            radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diagonal = 1;
        }
    }

    /* On the two tolerance exits Ceres returns before FinalizeIteration..., so the last
     * iteration summary is not appended; the user's parameters hold the last accepted x. */
    sum->termination_type = term;
    sum->final_cost = x_cost;
    sum->seconds_total = now_s() - t_start;
    sum->seconds_jacobian = t_jac;
    sum->seconds_linear = t_lin;
    if (p->cam_rt) for (int m = 0; m < C; ++m) if (S.cam_pose_col[m] >= 0) memcpy(p->cam_rt + 6 * m, S.x_cam + 6 * m, 6 * sizeof(double));
    for (int m = 0; m < C; ++m) if (S.intr_col[m] >= 0) memcpy(p->intr + 9 * m, S.x_intr + 9 * m, 9 * sizeof(double));
    for (int b = 0; b < B; ++b) if (S.board_active[b]) memcpy(p->board_rt + 6 * b, S.x_board + 6 * b, 6 * sizeof(double));

    free(S.cam_pose_col); free(S.intr_col); free(S.board_active); free(S.board_seen); free(cam_active); free(S.view_row); free(S.bv_ptr); free(S.bv_idx);
    free(zero_cam); free(S.x_cam); free(S.c_cam); free(S.x_intr); free(S.c_intr); free(S.x_board); free(S.c_board);
    free(S.res); free(S.Jc); free(S.Jb); free(S.Ji); free(S.g_f); free(S.g_b); free(S.s_f); free(S.s_b); free(S.d_f); free(S.d_b);
    free(S.step_f); free(S.step_b); free(S.lhs); free(S.rhs); free(S.inv_ete);
    return 0;
#undef EVAL_GRAD_JAC
}

/* ======================================================================== *
 *  Error reports
 * ======================================================================== */
/* multi_calib.cpp:233-283 (and main.cpp:245-288): p = R_b*w + t_b ; p = R_c*p + t_c with
 * Rodrigues matrices (update_param), projection with skew terms, mean Euclidean error. */
double orc_mean_reprojection_error(const orc_problem *p, double *per_camera)
{
    double error_sum = 0.0; long cntt = 0;
    double *err = (double *)xcalloc((size_t)p->n_cameras, sizeof(double));
    long *cnt = (long *)xcalloc((size_t)p->n_cameras, sizeof(long));
    for (int v = 0; v < p->n_views; ++v) {
        const int m = p->view_camera[v], b = p->view_board[v];
        double Rb[9], Rc[9];
        const double zero6[6] = { 0, 0, 0, 0, 0, 0 };
        const double *crt = (p->mono || !p->cam_rt) ? zero6 : p->cam_rt + 6 * m;
        orc_rodrigues(p->board_rt + 6 * b, Rb);
        orc_rodrigues(crt, Rc);
        const double *tb = p->board_rt + 6 * b + 3, *tc = crt + 3;
        for (int j = 0; j < p->view_count[v]; ++j) {
            const double w[3] = { p->board_xy[2 * j], p->board_xy[2 * j + 1], 0.0 };
            double q[3], P[3], uv[2];
            for (int i = 0; i < 3; ++i) q[i] = Rb[3 * i] * w[0] + Rb[3 * i + 1] * w[1] + Rb[3 * i + 2] * w[2] + tb[i];
            for (int i = 0; i < 3; ++i) P[i] = Rc[3 * i] * q[0] + Rc[3 * i + 1] * q[1] + Rc[3 * i + 2] * q[2] + tc[i];
            orc_project(p->intr + 9 * m, P, uv);
            const double du = p->obs_u[p->view_offset[v] + j] - uv[0], dv = p->obs_v[p->view_offset[v] + j] - uv[1];
            err[m] += sqrt(du * du + dv * dv); cnt[m]++; cntt++;
        }
    }
    for (int m = 0; m < p->n_cameras; ++m) { error_sum += err[m]; if (per_camera) per_camera[m] = cnt[m] ? err[m] / (double)cnt[m] : 0.0; }
    free(err); free(cnt);
    return cntt ? error_sum / (double)cntt : 0.0;
}

double orc_rmse(const orc_problem *p)
{
    const int N = total_corners(p);
    const double cost = orc_evaluate(p, 0, NULL, NULL, NULL, NULL);
    return N ? sqrt(2.0 * cost / (double)N) : 0.0;
}

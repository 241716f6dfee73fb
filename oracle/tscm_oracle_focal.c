/*
 * tscm_oracle_focal.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE), focal initialisation.
 *
 * Plain-C restatement of TripleSphereCamera::estimate_focal (TS.cpp:110-168): every row of every
 * detected board is the image of a straight line, i.e. (for xi = lambda = 0, alpha = 0.5) a circle
 * x*c1 + y*c2 + 0.5*c3 - 0.5*(x^2+y^2)*c4 = 0; the coefficient vector is the null vector of the
 * width x 4 design matrix (cv::SVD::solveZ, external: OpenCV core -- the right singular vector of
 * the smallest singular value; restated with a one-sided Jacobi SVD, the algorithm OpenCV's
 * JacobiSVD also uses), from which gamma = |c3 d / nz| is one focal-length sample (:146-156).
 * focal = mean of the accepted samples, summed in (view, row) order.
 * Also TS.cpp:62-74: [r1 r2 t] -> rt_ = [cv::Rodrigues(R), t] with the float32 cross product.
 * PARITY UNPINNED (no OpenCV here, see tscm_oracle.h).
 */
#include <math.h>
#include <string.h>

#include "tscm_oracle.h"

#define ORC_MAX_W 64

/* right singular vector of the smallest singular value of the m x 4 matrix A (row-major) */
static void null_vector4(const double *A_in, int m, double *c)
{
    double A[ORC_MAX_W * 4], V[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 };
    memcpy(A, A_in, sizeof(double) * 4 * m);
    for (int sweep = 0; sweep < 60; ++sweep) {
        int rotated = 0;
        for (int p = 0; p < 3; ++p)
            for (int q = p + 1; q < 4; ++q) {
                double a = 0, b = 0, g = 0;
                for (int r = 0; r < m; ++r) { a += A[4 * r + p] * A[4 * r + p]; b += A[4 * r + q] * A[4 * r + q]; g += A[4 * r + p] * A[4 * r + q]; }
                if (g == 0.0 || fabs(g) <= 1e-17 * sqrt(a * b)) continue;
                rotated = 1;
                const double zeta = (b - a) / (2.0 * g);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
                for (int r = 0; r < m; ++r) {
                    const double x = A[4 * r + p], y = A[4 * r + q];
                    A[4 * r + p] = cs * x - sn * y; A[4 * r + q] = sn * x + cs * y;
                }
                for (int r = 0; r < 4; ++r) {
                    const double x = V[4 * r + p], y = V[4 * r + q];
                    V[4 * r + p] = cs * x - sn * y; V[4 * r + q] = sn * x + cs * y;
                }
            }
        if (!rotated) break;
    }
    int jmin = 0; double smin = INFINITY;
    for (int j = 0; j < 4; ++j) {
        double s = 0;
        for (int r = 0; r < m; ++r) s += A[4 * r + j] * A[4 * r + j];
        if (s < smin) { smin = s; jmin = j; }
    }
    for (int r = 0; r < 4; ++r) c[r] = V[4 * r + jmin];
}

/* one row of one board: returns 1 and *gamma when the sample is accepted (TS.cpp:129-156) */
int orc_focal_sample(const double *pu, const double *pv, int width, double cx, double cy, double *gamma)
{
    double P[ORC_MAX_W * 4], C[4];
    for (int j = 0; j < width; ++j) {
        const double x = pu[j] - cx, y = pv[j] - cy;
        P[4 * j] = x; P[4 * j + 1] = y; P[4 * j + 2] = 0.5; P[4 * j + 3] = -0.5 * (x * x + y * y);
    }
    null_vector4(P, width, C);
    const double c1 = C[0], c2 = C[1], c3 = C[2], c4 = C[3];
    const double t = c1 * c1 + c2 * c2 + c3 * c4;
    if (t < 0) return 0;
    const double d = sqrt(1 / t);
    const double nx = c1 * d, ny = c2 * d;
    if (nx * nx + ny * ny > 0.95) return 0;
    const double nz = sqrt(1 - nx * nx - ny * ny);
    *gamma = fabs(c3 * d / nz);
    return 1;
}

/* pix_u/pix_v [n_views][width*height]; count[k] = pixels[k].size() (0: no board in image k) */
int orc_estimate_focal(const double *pix_u, const double *pix_v, const int *count, int n_views, int width, int height,
                       double cx, double cy, double *focal, int *total_num)
{
    if (width < 4 || width > ORC_MAX_W) return -1;
    double f = 0;
    int total = 0;
    const int n = width * height;
    for (int k = 0; k < n_views; ++k) {
        if (count[k] == 0) continue;
        for (int i = 0; i < height; ++i) {
            double gamma;
            if (!orc_focal_sample(pix_u + (size_t)k * n + i * width, pix_v + (size_t)k * n + i * width, width, cx, cy, &gamma)) continue;
            f += gamma;
            total++;
        }
    }
    if (total > 0) f /= total;
    *focal = f; *total_num = total;
    return 0;
}

/* TS.cpp:62-74: Rt_[i] (row-major 3x3 [r1 r2 t]) -> rt_[i] */
void orc_Rt_to_rt(const double *Rt, double *rt)
{
    double R[9], t[3];
    orc_Rt_to_R_t(Rt, R, t);          /* the same float32 construction as multi_calib.h:130-137 */
    orc_rodrigues_inverse(R, rt);
    rt[3] = t[0]; rt[4] = t[1]; rt[5] = t[2];
}

/* ==================================================================================================
 * estimate_extrinsic (TS.cpp:170-203).  The reference calls cv::solvePnPRansac (external, OpenCV
 * calib3d, randomised) on the board points and the corners un-projected onto a plane that a
 * rotation `transform` (:175-184) has turned towards the board.  Restated here as the deterministic
 * planar PnP that solvePnP's iterative method performs when every point is an inlier: homography by
 * DLT (Hartley-normalised board points, h33 = 1), pose from its columns, polar orthonormalisation,
 * then Gauss-Newton on the 6 pose parameters against the normalised image points.  NOT bit-comparable
 * with OpenCV (RANSAC draws, termination rules); documented deviation, see DESIGN.md.
 * ================================================================================================== */
static void m3mul(const double *A, const double *B, double *C)
{
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

static int chol_solve(double *A, double *b, int n)      /* A (row-major, n x n, SPD) is overwritten */
{
    for (int j = 0; j < n; ++j) {
        double d = A[j * n + j];
        for (int k = 0; k < j; ++k) d -= A[j * n + k] * A[j * n + k];
        if (!(d > 0.0)) return 0;
        d = sqrt(d);
        A[j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = A[i * n + j];
            for (int k = 0; k < j; ++k) s -= A[i * n + k] * A[j * n + k];
            A[i * n + j] = s / d;
        }
    }
    for (int i = 0; i < n; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= A[i * n + k] * b[k]; b[i] = s / A[i * n + i]; }
    for (int i = n - 1; i >= 0; --i) { double s = b[i]; for (int k = i + 1; k < n; ++k) s -= A[k * n + i] * b[k]; b[i] = s / A[i * n + i]; }
    return 1;
}

/* planar pose from n correspondences (X, Y, 0) -> (x, y) in the normalised image plane.
 * out: R (row-major 3x3), t.  returns 0 on a degenerate configuration. */
int orc_planar_pnp(const double *worlds, const double *xn, const double *yn, int n, double *R, double *t)
{
    double cx = 0, cy = 0, md = 0;
    for (int i = 0; i < n; ++i) { cx += worlds[3 * i]; cy += worlds[3 * i + 1]; }
    cx /= n; cy /= n;
    for (int i = 0; i < n; ++i) md += sqrt((worlds[3 * i] - cx) * (worlds[3 * i] - cx) + (worlds[3 * i + 1] - cy) * (worlds[3 * i + 1] - cy));
    md /= n;
    if (!(md > 0.0)) return 0;
    const double s = sqrt(2.0) / md;
    double A[64], b[8];
    memset(A, 0, sizeof(A)); memset(b, 0, sizeof(b));
    for (int i = 0; i < n; ++i) {
        const double X = (worlds[3 * i] - cx) * s, Y = (worlds[3 * i + 1] - cy) * s, x = xn[i], y = yn[i];
        const double r1[8] = { X, Y, 1, 0, 0, 0, -x * X, -x * Y }, r2[8] = { 0, 0, 0, X, Y, 1, -y * X, -y * Y };
        for (int p = 0; p < 8; ++p) {
            for (int q = 0; q < 8; ++q) A[8 * p + q] += r1[p] * r1[q] + r2[p] * r2[q];
            b[p] += r1[p] * x + r2[p] * y;
        }
    }
    if (!chol_solve(A, b, 8)) return 0;
    /* H = Hn * T,  T = [s 0 -s cx; 0 s -s cy; 0 0 1] */
    const double Hn[9] = { b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7], 1.0 };
    double H[9];
    for (int r = 0; r < 3; ++r) {
        H[3 * r] = Hn[3 * r] * s; H[3 * r + 1] = Hn[3 * r + 1] * s;
        H[3 * r + 2] = Hn[3 * r + 2] - s * (Hn[3 * r] * cx + Hn[3 * r + 1] * cy);
    }
    const double n1 = sqrt(H[0] * H[0] + H[3] * H[3] + H[6] * H[6]), n2 = sqrt(H[1] * H[1] + H[4] * H[4] + H[7] * H[7]);
    if (!(n1 > 0.0) || !(n2 > 0.0)) return 0;
    double lam = 2.0 / (n1 + n2);
    if (H[8] < 0) lam = -lam;                                   /* board in front of the camera: t_z > 0 */
    double M[9];
    for (int r = 0; r < 3; ++r) { M[3 * r] = lam * H[3 * r]; M[3 * r + 1] = lam * H[3 * r + 1]; t[r] = lam * H[3 * r + 2]; }
    M[2] = M[3] * M[7] - M[6] * M[4]; M[5] = M[6] * M[1] - M[0] * M[7]; M[8] = M[0] * M[4] - M[3] * M[1];   /* r3 = r1 x r2 */
    double rv[3];
    orc_rodrigues_inverse(M, rv);                               /* includes the orthonormalisation (SVD polar factor) */
    /* Gauss-Newton on (rv, t) */
    for (int it = 0; it < 10; ++it) {
        double Rm[9], JtJ[36], Jtr[6];
        orc_rodrigues(rv, Rm);
        memset(JtJ, 0, sizeof(JtJ)); memset(Jtr, 0, sizeof(Jtr));
        for (int i = 0; i < n; ++i) {
            const double p[3] = { worlds[3 * i], worlds[3 * i + 1], 0.0 };
            double res[2] = { 0.0, 0.0 }, J[2][6];
            /* numerical-free Jacobian through the oracle's dual numbers would be overkill: central differences
             * of this smooth 6-parameter map at h = 1e-6 are accurate to 1e-10, far below the GN tolerance */
            for (int q = -1; q < 6; ++q) {
                double fv[2][2];
                for (int sgn = 0; sgn < (q < 0 ? 1 : 2); ++sgn) {
                    double w[3] = { rv[0], rv[1], rv[2] }, tt[3] = { t[0], t[1], t[2] };
                    const double hh = q < 0 ? 0.0 : (sgn ? -1e-6 : 1e-6);
                    if (q >= 0 && q < 3) w[q] += hh; else if (q >= 3) tt[q - 3] += hh * fmax(1.0, fabs(t[q - 3]));
                    double P[3];
                    orc_angle_axis_rotate_point(w, p, P);
                    P[0] += tt[0]; P[1] += tt[1]; P[2] += tt[2];
                    fv[sgn][0] = P[0] / P[2] - xn[i]; fv[sgn][1] = P[1] / P[2] - yn[i];
                }
                if (q < 0) { res[0] = fv[0][0]; res[1] = fv[0][1]; }
                else {
                    const double den = 2e-6 * (q >= 3 ? fmax(1.0, fabs(t[q - 3])) : 1.0);
                    J[0][q] = (fv[0][0] - fv[1][0]) / den; J[1][q] = (fv[0][1] - fv[1][1]) / den;
                }
            }
            for (int a = 0; a < 6; ++a) {
                for (int c = 0; c < 6; ++c) JtJ[6 * a + c] += J[0][a] * J[0][c] + J[1][a] * J[1][c];
                Jtr[a] += J[0][a] * res[0] + J[1][a] * res[1];
            }
        }
        for (int a = 0; a < 6; ++a) JtJ[7 * a] *= 1.0 + 1e-12;
        if (!chol_solve(JtJ, Jtr, 6)) break;
        double step = 0;
        for (int a = 0; a < 3; ++a) { rv[a] -= Jtr[a]; t[a] -= Jtr[3 + a]; step += Jtr[a] * Jtr[a] + Jtr[3 + a] * Jtr[3 + a] / fmax(1.0, t[a] * t[a]); }
        if (step < 1e-24) break;
    }
    orc_rodrigues(rv, R);
    return 1;
}

/* TS.cpp:170-203 for every image with a board.  pix_u/pix_v [n_views][n]; Rt_out [n_views][9] row-major
 * [r1 r2 t] (untouched where count[k] == 0).  returns the number of views whose pose was estimated. */
int orc_estimate_extrinsic(const double *intr, const double *pix_u, const double *pix_v, const int *count, int n_views,
                           const double *worlds, int n, int board_w, double *Rt_out)
{
    int done = 0;
    double xs[1024], ys[1024];
    if (n > 1024) return -1;
    for (int k = 0; k < n_views; ++k) {
        if (count[k] == 0) continue;
        const double *u = pix_u + (size_t)k * n, *v = pix_v + (size_t)k * n;
        const int ref = n / 2 - board_w / 2 - 1;                       /* :178 */
        double uv[2] = { u[ref], v[ref] }, p[3];
        orc_unproject(intr, uv, p);
        const double alpha = atan2(p[0], p[2]), beta = asin(p[1]);     /* :179-180 */
        const double R1[9] = { cos(alpha), 0, -sin(alpha), 0, 1, 0, sin(alpha), 0, cos(alpha) };
        const double R2[9] = { 1, 0, 0, 0, cos(beta), -sin(beta), 0, sin(beta), cos(beta) };
        double T[9];
        m3mul(R2, R1, T);                                              /* transform = R2 * R1, :187 */
        for (int i = 0; i < n; ++i) {
            double q[3], uvi[2] = { u[i], v[i] };
            orc_unproject(intr, uvi, q);
            const double X = T[0] * q[0] + T[1] * q[1] + T[2] * q[2], Y = T[3] * q[0] + T[4] * q[1] + T[5] * q[2], Z = T[6] * q[0] + T[7] * q[1] + T[8] * q[2];
            xs[i] = X / Z; ys[i] = Y / Z;                              /* :190-191 */
        }
        double R[9], t[3];
        if (!orc_planar_pnp(worlds, xs, ys, n, R, t)) continue;
        double *o = Rt_out + 9 * (size_t)k;                            /* Rt = transform^T * [R | t], third column = t  (:195-200) */
        for (int r = 0; r < 3; ++r) {
            o[3 * r] = T[r] * R[0] + T[3 + r] * R[3] + T[6 + r] * R[6];
            o[3 * r + 1] = T[r] * R[1] + T[3 + r] * R[4] + T[6 + r] * R[7];
            o[3 * r + 2] = T[r] * t[0] + T[3 + r] * t[1] + T[6 + r] * t[2];
        }
        ++done;
    }
    return done;
}

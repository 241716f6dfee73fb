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

/*
 * tscm_oracle_rig.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE):
 * restatement of the rig initialisation MultiCalib::MultiCalib (multi_calib.cpp:6-153).
 * cv::Rodrigues / cv::SVD are OpenCV (not installed, not under /root/reference): restated from
 * the published algorithm (calib3d Rodrigues: R <- U V^T of the SVD, axis from the antisymmetric
 * part, special case near 0 / pi).  PARITY UNPINNED like the rest of the oracle.
 */
#include "tscm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static void mat3_mul(const double *A, const double *B, double *C)
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
static void mat3_mul_bt(const double *A, const double *B, double *C)   /* A * B^T */
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[3 * j] + A[3 * i + 1] * B[3 * j + 1] + A[3 * i + 2] * B[3 * j + 2];
}
static void mat3_tmul(const double *A, const double *B, double *C)     /* A^T * B */
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) C[3 * i + j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
}
static void mat3_vec(const double *A, const double *x, double *y)
{
    for (int i = 0; i < 3; ++i) y[i] = A[3 * i] * x[0] + A[3 * i + 1] * x[1] + A[3 * i + 2] * x[2];
}
static void mat3_tvec(const double *A, const double *x, double *y)     /* A^T x */
{
    for (int i = 0; i < 3; ++i) y[i] = A[i] * x[0] + A[3 + i] * x[1] + A[6 + i] * x[2];
}

/* multi_calib.h:130-137: cv::Vec3f r1, r2 (float32 copies), r3 = r1.cross(r2) in float */
void orc_Rt_to_R_t(const double *Rt, double *R, double *t)
{
    /* volatile: gcc 11 -O3 was seen to drop the double->float->double rounding of some elements */
    volatile float r1[3], r2[3];
    float r3[3];
    for (int i = 0; i < 3; ++i) { r1[i] = (float)Rt[3 * i]; r2[i] = (float)Rt[3 * i + 1]; }
    { const float a = r1[1] * r2[2], b = r1[2] * r2[1]; r3[0] = a - b; }
    { const float a = r1[2] * r2[0], b = r1[0] * r2[2]; r3[1] = a - b; }
    { const float a = r1[0] * r2[1], b = r1[1] * r2[0]; r3[2] = a - b; }
    for (int i = 0; i < 3; ++i) { R[3 * i] = r1[i]; R[3 * i + 1] = r2[i]; R[3 * i + 2] = r3[i]; }
    t[0] = Rt[2]; t[1] = Rt[5]; t[2] = Rt[8];
}

/* one-sided Jacobi SVD of a 3x3 matrix: A = U diag(w) V^T; returns U V^T in Q */
static void orthonormal_factor(const double *A, double *Q)
{
    double U[9], V[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    memcpy(U, A, sizeof(U));
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double a = 0, b = 0, c = 0;
                for (int i = 0; i < 3; ++i) { a += U[3 * i + p] * U[3 * i + p]; b += U[3 * i + q] * U[3 * i + q]; c += U[3 * i + p] * U[3 * i + q]; }
                off = fmax(off, fabs(c) / sqrt(fmax(a * b, 1e-300)));
                if (fabs(c) <= 1e-300) continue;
                const double zeta = (b - a) / (2.0 * c);
                const double tt = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double cs = 1.0 / sqrt(1.0 + tt * tt), sn = cs * tt;
                for (int i = 0; i < 3; ++i) {
                    const double up = U[3 * i + p], uq = U[3 * i + q];
                    U[3 * i + p] = cs * up - sn * uq; U[3 * i + q] = sn * up + cs * uq;
                    const double vp = V[3 * i + p], vq = V[3 * i + q];
                    V[3 * i + p] = cs * vp - sn * vq; V[3 * i + q] = sn * vp + cs * vq;
                }
            }
        if (off < 1e-17) break;
    }
    for (int j = 0; j < 3; ++j) {
        double w = 0;
        for (int i = 0; i < 3; ++i) w += U[3 * i + j] * U[3 * i + j];
        w = sqrt(w);
        for (int i = 0; i < 3; ++i) U[3 * i + j] = w > 0 ? U[3 * i + j] / w : U[3 * i + j];
    }
    mat3_mul_bt(U, V, Q);
}

/* cv::Rodrigues, matrix -> vector branch (external: OpenCV calib3d) */
void orc_rodrigues_inverse(const double *Rin, double *r)
{
    double R[9];
    orthonormal_factor(Rin, R);
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    const double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1.0) * 0.5;
    c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
    double theta = acos(c);
    if (s < 1e-5) {
        if (c > 0) { r[0] = r[1] = r[2] = 0.0; return; }
        double t = (R[0] + 1.0) * 0.5;
        rx = sqrt(fmax(t, 0.0));
        t = (R[4] + 1.0) * 0.5;
        ry = sqrt(fmax(t, 0.0)) * (R[1] < 0 ? -1.0 : 1.0);
        t = (R[8] + 1.0) * 0.5;
        rz = sqrt(fmax(t, 0.0)) * (R[2] < 0 ? -1.0 : 1.0);
        if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && ((R[5] > 0) != (ry * rz > 0))) rz = -rz;
        theta /= sqrt(rx * rx + ry * ry + rz * rz);
        r[0] = rx * theta; r[1] = ry * theta; r[2] = rz * theta;
    } else {
        double vth = 1.0 / (2.0 * s);
        vth *= theta;
        r[0] = rx * vth; r[1] = ry * vth; r[2] = rz * vth;
    }
}

/* TS.h:58-69: SUM over all board points of the Euclidean pixel error, P = R*world + t */
static double reproject_error_sum(const double *I, const double *pu, const double *pv, const double *worlds, int n,
                                  const double *R, const double *t)
{
    double error = 0.0;
    for (int i = 0; i < n; ++i) {
        double P[3], uv[2];
        mat3_vec(R, worlds + 3 * i, P);
        P[0] += t[0]; P[1] += t[1]; P[2] += t[2];
        orc_project(I, P, uv);
        error += sqrt((pu[i] - uv[0]) * (pu[i] - uv[0]) + (pv[i] - uv[1]) * (pv[i] - uv[1]));
    }
    return error;
}

#define HAS(m, j) (in->has[(size_t)(m) * in->n_boards + (j)])
#define RT(m, j) (in->Rt + 9 * ((size_t)(m) * in->n_boards + (j)))
#define PU(m, j) (in->pix_u + (size_t)in->n_points * ((size_t)(m) * in->n_boards + (j)))
#define PV(m, j) (in->pix_v + (size_t)in->n_points * ((size_t)(m) * in->n_boards + (j)))

/* multi_calib.cpp:52-78: summed reprojection error of pose hypotheses (Rs[j], ts[j]) for camera i
 * over every board cameras i-1 and i share, both directions.  (Rp, tp) = pose of camera i-1.  */
void orc_rig_hypothesis_errors(const orc_rig_input *in, int i, const double *Rp, const double *tp,
                               const double *Rs, const double *ts, int nj, double *errors)
{
    const int B = in->n_boards, n = in->n_points;
    for (int j = 0; j < nj; ++j) {
        double error = 0.0;
        for (int k = 0; k < B; ++k) {
            if (!(HAS(i - 1, k) && HAS(i, k))) continue;
            double Ri[9], ti[3], Rk[9], tk[3], Rki[9], tki[3], tmp[3];
            orc_Rt_to_R_t(RT(i, k), Ri, ti);
            mat3_mul_bt(Rp, Rs + 9 * j, Rki);              /* camera_R_k * Rs[j].t() */
            mat3_vec(Rki, ts + 3 * j, tmp);
            for (int a = 0; a < 3; ++a) tki[a] = tp[a] - tmp[a];
            mat3_mul(Rki, Ri, Rk);
            mat3_vec(Rki, ti, tmp);
            for (int a = 0; a < 3; ++a) tk[a] = tmp[a] + tki[a];
            error += reproject_error_sum(in->intr + 9 * (i - 1), PU(i - 1, k), PV(i - 1, k), in->worlds, n, Rk, tk);
            orc_Rt_to_R_t(RT(i - 1, k), Rk, tk);
            mat3_mul_bt(Rs + 9 * j, Rp, Rki);              /* Rs[j] * camera_R_k.t() */
            mat3_vec(Rki, tp, tmp);
            for (int a = 0; a < 3; ++a) tki[a] = ts[3 * j + a] - tmp[a];
            mat3_mul(Rki, Rk, Ri);
            mat3_vec(Rki, tk, tmp);
            for (int a = 0; a < 3; ++a) ti[a] = tmp[a] + tki[a];
            error += reproject_error_sum(in->intr + 9 * i, PU(i, k), PV(i, k), in->worlds, n, Ri, ti);
        }
        errors[j] = error;
    }
}

int orc_rig_init(const orc_rig_input *in, double *cam_R, double *cam_t, double *cam_rt,
                 double *board_R, double *board_t, double *board_rt, unsigned char *board_initial,
                 int *cam_choice, double *cam_min_error)
{
    const int C = in->n_cameras, B = in->n_boards, n = in->n_points;
    for (int i = 0; i < C; ++i) {
        double *R = cam_R + 9 * i, *t = cam_t + 3 * i;
        if (cam_choice) cam_choice[i] = -1;
        if (cam_min_error) cam_min_error[i] = 0.0;
        if (i == 0) {                                              /* multi_calib.cpp:19-23 */
            const double I3[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
            memcpy(R, I3, sizeof(I3)); t[0] = t[1] = t[2] = 0.0;
        } else {
            const double *Rp = cam_R + 9 * (i - 1), *tp = cam_t + 3 * (i - 1);
            int nh = 0;
            for (int j = 0; j < B; ++j) if (HAS(i - 1, j) && HAS(i, j)) ++nh;
            if (nh == 0) return -1;                                /* reference: Rs[-1], UB (:51,:86) */
            double *Rs = (double *)malloc(sizeof(double) * 9 * nh), *ts = (double *)malloc(sizeof(double) * 3 * nh);
            int h = 0;
            for (int j = 0; j < B; ++j) {                          /* :29-48 */
                if (!(HAS(i - 1, j) && HAS(i, j))) continue;
                double Ri[9], ti[3], Rk[9], tk[3], Rik[9], tik[3], tmp[3];
                orc_Rt_to_R_t(RT(i, j), Ri, ti);
                orc_Rt_to_R_t(RT(i - 1, j), Rk, tk);
                mat3_mul_bt(Ri, Rk, Rik);
                mat3_vec(Rik, tk, tmp);
                for (int a = 0; a < 3; ++a) tik[a] = ti[a] - tmp[a];
                mat3_mul(Rik, Rp, Rs + 9 * h);
                mat3_vec(Rik, tp, tmp);
                for (int a = 0; a < 3; ++a) ts[3 * h + a] = tmp[a] + tik[a];
                ++h;
            }
            double min_error = 1e10; int min_id = -1;
            for (int j = 0; j < nh; ++j) {                         /* :50-85 */
                double error;
                orc_rig_hypothesis_errors(in, i, Rp, tp, Rs + 9 * j, ts + 3 * j, 1, &error);
                if (error < min_error) { min_error = error; min_id = j; }
            }
            if (min_id < 0) { free(Rs); free(ts); return -1; }
            memcpy(R, Rs + 9 * min_id, sizeof(double) * 9);
            memcpy(t, ts + 3 * min_id, sizeof(double) * 3);
            if (cam_choice) cam_choice[i] = min_id;
            if (cam_min_error) cam_min_error[i] = min_error;
            free(Rs); free(ts);
        }
        orc_rodrigues_inverse(R, cam_rt + 6 * i);                   /* multi_calib.h:16-18 */
        memcpy(cam_rt + 6 * i + 3, t, sizeof(double) * 3);
    }
    for (int i = 0; i < B; ++i) {                                   /* :90-151 */
        double *R = board_R + 9 * i, *t = board_t + 3 * i;
        int ids[64], nc = 0;
        for (int j = 0; j < C && nc < 64; ++j) if (HAS(j, i)) ids[nc++] = j;
        board_initial[i] = nc > 0;
        memset(R, 0, sizeof(double) * 9); memset(t, 0, sizeof(double) * 3); memset(board_rt + 6 * i, 0, sizeof(double) * 6);
        if (nc == 0) continue;
        double Rs[64 * 9], ts[64 * 3];
        for (int j = 0; j < nc; ++j) {
            double cR[9], ct[3], tmp[3];
            orc_Rt_to_R_t(RT(ids[j], i), cR, ct);
            mat3_tmul(cam_R + 9 * ids[j], cR, Rs + 9 * j);         /* camera_R.t() * chess_R */
            for (int a = 0; a < 3; ++a) tmp[a] = ct[a] - cam_t[3 * ids[j] + a];
            mat3_tvec(cam_R + 9 * ids[j], tmp, ts + 3 * j);
        }
        int min_id = 0;
        if (nc > 1) {
            double min_error = 1e10; min_id = -1;
            for (int j = 0; j < nc; ++j) {
                double error = 0.0;
                for (int k = 0; k < nc; ++k) {
                    double cR[9], ct[3];
                    mat3_mul(cam_R + 9 * ids[k], Rs + 9 * j, cR);
                    mat3_vec(cam_R + 9 * ids[k], ts + 3 * j, ct);
                    for (int a = 0; a < 3; ++a) ct[a] += cam_t[3 * ids[k] + a];
                    error += reproject_error_sum(in->intr + 9 * ids[k], PU(ids[k], i), PV(ids[k], i), in->worlds, n, cR, ct);
                }
                if (error < min_error) { min_error = error; min_id = j; }
            }
            if (min_id < 0) return -1;
        }
        memcpy(R, Rs + 9 * min_id, sizeof(double) * 9);
        memcpy(t, ts + 3 * min_id, sizeof(double) * 3);
        orc_rodrigues_inverse(R, board_rt + 6 * i);                 /* multi_calib.h:94-96 */
        memcpy(board_rt + 6 * i + 3, t, sizeof(double) * 3);
    }
    return 0;
}

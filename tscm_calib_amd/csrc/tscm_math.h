// tscm_math.h -- Triple-Sphere camera model math shared by all kernels (fp64).
//
// What it computes (reference file:line):
//   * rotation matrix of an angle-axis vector and its three partial derivatives, with the
//     same two branches as ceres::AngleAxisRotatePoint (call sites TS.h:112,
//     multi_calib.h:158,164): Rodrigues for theta^2 > DBL_EPSILON, I + [w]x otherwise;
//   * residual and analytic 2xK Jacobian of one corner for the multi-camera functor
//     (multi_calib.h:146-195); the mono functor (TS.h:100-131) is the same with an
//     identity, constant camera pose (R_c = I + [0]x, t_c = 0: bit-identical point);
//   * plain projection with skew terms (TS.cpp:332-344) and unprojection (TS.h:39-57).
//
// The Jacobian is hand-derived (no dual numbers on the device):
//   rho2 = X^2+Y^2, d1 = sqrt(rho2+Z^2), z1 = Z+xi d1, d2 = sqrt(rho2+z1^2),
//   z2 = z1+lambda d2, d3 = sqrt(rho2+z2^2), beta = alpha/(1-alpha), k = z2+beta d3
//   c1 = 1+xi Z/d1, c2 = 1+lambda z1/d2, c3 = 1+beta z2/d3
//   dk/dX = X q, dk/dY = Y q, q = beta/d3 + c3 (lambda/d2 + c2 xi/d1);  dk/dZ = c1 c2 c3
//   dk/dxi = c3 c2 d1, dk/dlambda = c3 d2, dk/dalpha = d3/(1-alpha)^2
//   u = fx X/k + cx, v = fy Y/k + cy, residual = observed - (u, v).
#pragma once

#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>

#define TSCM_HD __host__ __device__ __forceinline__

namespace tscm {

// Column layout of the per-corner Jacobian used by the Gram kernels.
//   F (camera side, 16 wide): 0-2 w_c, 3-5 t_c, 6 fx, 7 fy, 8 cx, 9 cy, 10 xi, 11 lambda,
//                             12 alpha, 13 = residual, 14-15 zero padding
//   E (board side, 6 wide):   0-2 w_b, 3-5 t_b
constexpr int kF = 16;
constexpr int kFA = 13;   // active camera-side parameter columns
constexpr int kFR = 13;   // index of the residual column
constexpr int kE = 6;

// R (row-major 3x3) and dR[k] = dR/dw_k (row-major 3x3 each) of the angle-axis vector w.
TSCM_HD void rotation_and_derivatives(const double w[3], double R[9], double dR[27])
{
    const double theta2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    if (theta2 > DBL_EPSILON) {
        const double theta = sqrt(theta2);
        const double c = cos(theta), s = sin(theta);
        const double it = 1.0 / theta;
        const double k[3] = { w[0] * it, w[1] * it, w[2] * it };
        const double c1 = 1.0 - c;
        // R = c I + s [k]x + (1-c) k k^T
        R[0] = c + c1 * k[0] * k[0];        R[1] = c1 * k[0] * k[1] - s * k[2]; R[2] = c1 * k[0] * k[2] + s * k[1];
        R[3] = c1 * k[0] * k[1] + s * k[2]; R[4] = c + c1 * k[1] * k[1];        R[5] = c1 * k[1] * k[2] - s * k[0];
        R[6] = c1 * k[0] * k[2] - s * k[1]; R[7] = c1 * k[1] * k[2] + s * k[0]; R[8] = c + c1 * k[2] * k[2];
        // dR/dw_i = -s k_i I + c k_i [k]x + s [dk_i]x + s k_i k k^T + (1-c)(dk_i k^T + k dk_i^T),
        // dk_i = (e_i - k_i k)/theta
        for (int i = 0; i < 3; ++i) {
            double dk[3] = { -k[i] * k[0] * it, -k[i] * k[1] * it, -k[i] * k[2] * it };
            dk[i] += it;
            const double a = -s * k[i], b = c * k[i], e = s * k[i];
            double *D = dR + 9 * i;
            for (int r = 0; r < 3; ++r)
                for (int q = 0; q < 3; ++q)
                    D[3 * r + q] = e * k[r] * k[q] + c1 * (dk[r] * k[q] + k[r] * dk[q]);
            D[0] += a; D[4] += a; D[8] += a;
            // b [k]x + s [dk]x
            const double x0 = b * k[0] + s * dk[0], x1 = b * k[1] + s * dk[1], x2 = b * k[2] + s * dk[2];
            D[1] -= x2; D[2] += x1;
            D[3] += x2; D[5] -= x0;
            D[6] -= x1; D[7] += x0;
        }
    } else {
        // R p = p + w x p  ->  R = I + [w]x,  dR/dw_i = [e_i]x
        R[0] = 1.0;   R[1] = -w[2]; R[2] = w[1];
        R[3] = w[2];  R[4] = 1.0;   R[5] = -w[0];
        R[6] = -w[1]; R[7] = w[0];  R[8] = 1.0;
        for (int i = 0; i < 27; ++i) dR[i] = 0.0;
        dR[0 * 9 + 5] = -1.0; dR[0 * 9 + 7] = 1.0;   // [e_x]x
        dR[1 * 9 + 2] = 1.0;  dR[1 * 9 + 6] = -1.0;  // [e_y]x
        dR[2 * 9 + 1] = -1.0; dR[2 * 9 + 3] = 1.0;   // [e_z]x
    }
}

// Per-board constants consumed by the corner kernel: only the first two columns of R_b and
// of dR_b/dw_k are needed because board points have z = 0 (TS.h:109, multi_calib.h:156).
//   [0..2] r1, [3..5] r2, then for k = 0..2: [6+6k .. 8+6k] dR_k col 0, [9+6k .. 11+6k] dR_k col 1
constexpr int kBoardConst = 24;
// Per-camera constants: R_c (9 row-major), dR_c/dw_k (27)
constexpr int kCamConst = 36;

TSCM_HD void board_constants(const double rt[6], double out[kBoardConst])
{
    double R[9], dR[27];
    rotation_and_derivatives(rt, R, dR);
    out[0] = R[0]; out[1] = R[3]; out[2] = R[6];
    out[3] = R[1]; out[4] = R[4]; out[5] = R[7];
    for (int k = 0; k < 3; ++k) {
        const double *D = dR + 9 * k;
        out[6 + 6 * k + 0] = D[0]; out[6 + 6 * k + 1] = D[3]; out[6 + 6 * k + 2] = D[6];
        out[6 + 6 * k + 3] = D[1]; out[6 + 6 * k + 4] = D[4]; out[6 + 6 * k + 5] = D[7];
    }
}

TSCM_HD void camera_constants(const double rt[6], double out[kCamConst])
{
    rotation_and_derivatives(rt, out, out + 9);
}

// The camera rotation in the form the Gram kernels consume since round 3: R (row-major), the three vectors a_k with
// dR/dw_k = [a_k]x R  (the columns of the rotation's left Jacobian; a_k = vee(dR_k R^T), taken from the same dR the
// rounds before used), so that  dR/dw_k P_w = a_k x (R P_w) = a_k x Q,  Q = P_c - t_c:  the camera-rotation columns
// of a corner are  n . (a_k x Q) = a_k . (Q x n)  -- one cross product per Jacobian row and three dot products, against
// three matrix-vector products with 27 constants.  Small-angle branch of ceres::AngleAxisRotatePoint (theta^2 <=
// DBL_EPSILON: R p = p + w x p, dR/dw_k = [e_k]x): a_k = e_k and the kernels use Q' = Q - w x Q  (= P_w up to
// O(|w|^2) ~ 1e-16 relative); wsm = w there and 0 otherwise.  Returns 1 in the small-angle branch.
TSCM_HD int camera_rotation_constants(const double w[3], double R[9], double a[9], double wsm[3])
{
    double dR[27];
    rotation_and_derivatives(w, R, dR);
    const double theta2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    if (theta2 > DBL_EPSILON) {
        for (int k = 0; k < 3; ++k) {
            const double *D = dR + 9 * k;
            double A[9];
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) A[3 * r + c] = D[3 * r] * R[3 * c] + D[3 * r + 1] * R[3 * c + 1] + D[3 * r + 2] * R[3 * c + 2];
            a[3 * k + 0] = 0.5 * (A[7] - A[5]);
            a[3 * k + 1] = 0.5 * (A[2] - A[6]);
            a[3 * k + 2] = 0.5 * (A[3] - A[1]);
        }
        wsm[0] = wsm[1] = wsm[2] = 0.0;
        return 0;
    }
    for (int i = 0; i < 9; ++i) a[i] = 0.0;
    a[0] = a[4] = a[8] = 1.0;
    wsm[0] = w[0]; wsm[1] = w[1]; wsm[2] = w[2];
    return 1;
}

// Everything a corner needs that is uniform over one (camera, board) view.
struct ViewConst {
    double r1[3], r2[3], tb[3];     // board: R_b columns 0,1 and translation
    double db[3][6];                // board: dR_b/dw_k columns 0,1
    double Rc[9], tc[3];            // camera rotation (row-major) and translation
    double dRc[27];                 // dR_c/dw_k
    double fx, fy, cx, cy, xi, lam, al;
};

// residual r[2] and Jacobian rows JE[2][6] (board pose), JF[2][13] (camera pose 6, intrinsics 7)
// of one corner with board point (x, y, 0) and observation (ou, ov).
TSCM_HD void corner_residual_jacobian(const ViewConst &vc, double x, double y, double ou, double ov,
                                      double r[2], double JE[2][kE], double JF[2][kFA])
{
    // board -> world -> camera  (multi_calib.h:158-167)
    const double Pw0 = x * vc.r1[0] + y * vc.r2[0] + vc.tb[0];
    const double Pw1 = x * vc.r1[1] + y * vc.r2[1] + vc.tb[1];
    const double Pw2 = x * vc.r1[2] + y * vc.r2[2] + vc.tb[2];
    const double X = vc.Rc[0] * Pw0 + vc.Rc[1] * Pw1 + vc.Rc[2] * Pw2 + vc.tc[0];
    const double Y = vc.Rc[3] * Pw0 + vc.Rc[4] * Pw1 + vc.Rc[5] * Pw2 + vc.tc[1];
    const double Z = vc.Rc[6] * Pw0 + vc.Rc[7] * Pw1 + vc.Rc[8] * Pw2 + vc.tc[2];
    // triple sphere (multi_calib.h:170-178)
    const double rho2 = X * X + Y * Y;
    const double d1 = sqrt(rho2 + Z * Z);
    const double z1 = Z + vc.xi * d1;
    const double d2 = sqrt(rho2 + z1 * z1);
    const double z2 = z1 + vc.lam * d2;
    const double d3 = sqrt(rho2 + z2 * z2);
    const double oma = 1.0 - vc.al;
    const double beta = vc.al / oma;
    const double k = z2 + beta * d3;
    const double ik = 1.0 / k;
    const double mx = X * ik, my = Y * ik;
    r[0] = ou - (vc.fx * mx + vc.cx);
    r[1] = ov - (vc.fy * my + vc.cy);

    const double id1 = 1.0 / d1, id2 = 1.0 / d2, id3 = 1.0 / d3;
    const double c1 = 1.0 + vc.xi * Z * id1;
    const double c2 = 1.0 + vc.lam * z1 * id2;
    const double c3 = 1.0 + beta * z2 * id3;
    const double q = beta * id3 + c3 * (vc.lam * id2 + c2 * vc.xi * id1);
    const double kz = c1 * c2 * c3;
    const double fxk = vc.fx * ik, fyk = vc.fy * ik;
    // A = d(u,v)/dPc
    const double a00 = fxk * (1.0 - X * mx * q), a01 = -fxk * mx * Y * q, a02 = -fxk * mx * kz;
    const double a10 = -fyk * my * X * q, a11 = fyk * (1.0 - Y * my * q), a12 = -fyk * my * kz;

    // camera pose: dPc/dw_c[k] = dRc_k Pw ; dPc/dt_c = I
    for (int kk = 0; kk < 3; ++kk) {
        const double *D = vc.dRc + 9 * kk;
        const double g0 = D[0] * Pw0 + D[1] * Pw1 + D[2] * Pw2;
        const double g1 = D[3] * Pw0 + D[4] * Pw1 + D[5] * Pw2;
        const double g2 = D[6] * Pw0 + D[7] * Pw1 + D[8] * Pw2;
        JF[0][kk] = -(a00 * g0 + a01 * g1 + a02 * g2);
        JF[1][kk] = -(a10 * g0 + a11 * g1 + a12 * g2);
    }
    JF[0][3] = -a00; JF[0][4] = -a01; JF[0][5] = -a02;
    JF[1][3] = -a10; JF[1][4] = -a11; JF[1][5] = -a12;
    // intrinsics: fx fy cx cy xi lambda alpha
    JF[0][6] = -mx;  JF[1][6] = 0.0;
    JF[0][7] = 0.0;  JF[1][7] = -my;
    JF[0][8] = -1.0; JF[1][8] = 0.0;
    JF[0][9] = 0.0;  JF[1][9] = -1.0;
    const double hu = fxk * mx, hv = fyk * my;          // -du/dk, -dv/dk
    const double kxi = c3 * c2 * d1, klam = c3 * d2, kal = d3 / (oma * oma);
    JF[0][10] = hu * kxi;  JF[1][10] = hv * kxi;
    JF[0][11] = hu * klam; JF[1][11] = hv * klam;
    JF[0][12] = hu * kal;  JF[1][12] = hv * kal;
    // board pose: AR = A Rc ; dPw/dw_b[k] = x dRb_k[:,0] + y dRb_k[:,1] ; dPw/dt_b = I
    const double ar00 = a00 * vc.Rc[0] + a01 * vc.Rc[3] + a02 * vc.Rc[6];
    const double ar01 = a00 * vc.Rc[1] + a01 * vc.Rc[4] + a02 * vc.Rc[7];
    const double ar02 = a00 * vc.Rc[2] + a01 * vc.Rc[5] + a02 * vc.Rc[8];
    const double ar10 = a10 * vc.Rc[0] + a11 * vc.Rc[3] + a12 * vc.Rc[6];
    const double ar11 = a10 * vc.Rc[1] + a11 * vc.Rc[4] + a12 * vc.Rc[7];
    const double ar12 = a10 * vc.Rc[2] + a11 * vc.Rc[5] + a12 * vc.Rc[8];
    for (int kk = 0; kk < 3; ++kk) {
        const double h0 = x * vc.db[kk][0] + y * vc.db[kk][3];
        const double h1 = x * vc.db[kk][1] + y * vc.db[kk][4];
        const double h2 = x * vc.db[kk][2] + y * vc.db[kk][5];
        JE[0][kk] = -(ar00 * h0 + ar01 * h1 + ar02 * h2);
        JE[1][kk] = -(ar10 * h0 + ar11 * h1 + ar12 * h2);
    }
    JE[0][3] = -ar00; JE[0][4] = -ar01; JE[0][5] = -ar02;
    JE[1][3] = -ar10; JE[1][4] = -ar11; JE[1][5] = -ar12;
}

// TS.cpp:332-344 (with skew terms b, c)
TSCM_HD void project_point(const double I[9], double X, double Y, double Z, double &u, double &v)
{
    const double d1 = sqrt(X * X + Y * Y + Z * Z);
    const double z1 = Z + I[4] * d1;
    const double d2 = sqrt(X * X + Y * Y + z1 * z1);
    const double z2 = z1 + I[5] * d2;
    const double d3 = sqrt(X * X + Y * Y + z2 * z2);
    const double ksai = z2 + I[6] / (1.0 - I[6]) * d3;
    u = I[0] * X / ksai + I[7] * Y / ksai + I[2];
    v = I[8] * X / ksai + I[1] * Y / ksai + I[3];
}

// TS.h:39-57 with transform = identity
TSCM_HD void unproject_pixel(const double I[9], double px, double py, double ray[3])
{
    const double fx = I[0], fy = I[1], cx = I[2], cy = I[3], xi = I[4], lam = I[5], al = I[6], b = I[7], c = I[8];
    const double x = px - cx, y = py - cy;
    const double det = fx * fy - b * c;
    const double mx = (fy * x - b * y) / det;
    const double my = (-c * x + fx * y) / det;
    const double ksai = al / (1.0 - al);
    const double r2 = mx * mx + my * my;
    const double gamma = (ksai + sqrt(1.0 + (1.0 - ksai * ksai) * r2)) / (r2 + 1.0);
    const double gk = gamma - ksai;
    const double yita = lam * gk + sqrt((gk * gk - 1.0) * lam * lam + 1.0);
    const double mz = yita * gk;
    const double ml = mz - lam;
    const double mu = xi * ml + sqrt(xi * xi * (ml * ml - 1.0) + 1.0);
    ray[0] = mu * yita * gamma * mx;
    ray[1] = mu * yita * gamma * my;
    ray[2] = mu * ml - xi;
}

}  // namespace tscm

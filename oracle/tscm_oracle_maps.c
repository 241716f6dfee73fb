/*
 * tscm_oracle_maps.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE), remap tables.
 *
 * Plain-C restatement of the per-pixel map loops of the reference:
 *   TripleSphereCamera::undistort             TS.cpp:284-306
 *   undistort_chessboard (table only)         TS.cpp:308-330 (cv::remap itself is OpenCV, external)
 *   Remap::init_remap                         EpipolarRectify/rectify.cpp:86-199, TScamera::project :22-36
 * all of the form  ray = R * ((j-cx)/fx, (i-cy)/fy, 1), pixel = project(ray), map = (float)(pixel + offset).
 * cv::Mat products are restated as the plain sum a0*b0 + a1*b1 + a2*b2 in that order.
 * PARITY UNPINNED (no reference build, see tscm_oracle.h); pinned by numpy restatements in tests/.
 */
#include <math.h>
#include <stddef.h>

#include "tscm_oracle.h"

void orc_build_map(const orc_map_desc *m, float *mapx, float *mapy)
{
    const double *I = m->intr, *R = m->R;
    for (int i = 0; i < m->height; ++i) {
        for (int j = 0; j < m->width; ++j) {
            const double x0 = (j - m->cx) / m->fx, y0 = (i - m->cy) / m->fy;
            const double X = R[0] * x0 + R[1] * y0 + R[2] * 1.0;
            const double Y = R[3] * x0 + R[4] * y0 + R[5] * 1.0;
            const double Z = R[6] * x0 + R[7] * y0 + R[8] * 1.0;
            const double d1 = sqrt(X * X + Y * Y + Z * Z);
            double u, v;
            if (m->check_w2 && Z <= -m->w2 * d1) {                     /* rectify.cpp:27 */
                u = -1.0; v = -1.0;
            } else {
                const double d2 = sqrt(X * X + Y * Y + pow(Z + I[4] * d1, 2));
                const double d3 = sqrt(X * X + Y * Y + pow(Z + I[4] * d1 + I[5] * d2, 2));
                const double ksai = Z + I[4] * d1 + I[5] * d2 + I[6] / (1 - I[6]) * d3;
                u = I[0] * X / ksai + I[7] * Y / ksai + I[2];
                v = I[8] * X / ksai + I[1] * Y / ksai + I[3];
            }
            const size_t o = (size_t)m->out_offset + (size_t)i * m->out_stride + j;
            mapx[o] = (float)(u + m->offset_x);
            mapy[o] = (float)(v + m->offset_y);
        }
    }
}

static void normalize3(double *v)                                       /* rectify.cpp:214-221 */
{
    const double norm = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    if (norm == 0) return;
    v[0] /= norm; v[1] /= norm; v[2] /= norm;
}

/* Remap::calc_R, rectify.cpp:234-248: rectifying rotation of a camera pair from the two centres */
void orc_rectify_pair_rotation(const double *t1, const double *t2, double *R)
{
    double x[3] = { t2[0] - t1[0], t2[1] - t1[1], t2[2] - t1[2] };
    normalize3(x);
    double z[3] = { -x[2], 0, x[0] };
    normalize3(z);
    double y[3];
    y[0] = -z[2] * x[1] + z[1] * x[2];                                   /* product(z, x, y), :223-232 */
    y[1] = z[2] * x[0] - z[0] * x[2];
    y[2] = -z[1] * x[0] + z[0] * x[1];
    normalize3(y);
    for (int r = 0; r < 3; ++r) { R[3 * r] = x[r]; R[3 * r + 1] = y[r]; R[3 * r + 2] = z[r]; }
}

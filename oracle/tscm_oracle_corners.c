/*
 * tscm_oracle_corners.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE), corner candidates.
 *
 * Plain-C restatement of the per-image part of the reference's chessboard-corner detector:
 *   findCorner                 DetectCorner/findCorner.cpp:7-66   (gradient angle/weight, normalisation, call order)
 *   secondDerivCornerMetric    :103-142
 *   nonMaximumSuppression      :144-193
 *   getOrientations            :200-234, edgeOrientations :236-279, findModesMeanShift :286-349
 *   scoreCorners               :391-426, cornerCorrelationScore :428-490, createCorrelationPatch :351-389
 *   subPixelLocation           :492-541
 * The chessboard structure recovery (DetectCorner/chessboard.cpp) is not part of this file.
 *
 * OpenCV calls are restated from their documented semantics (OpenCV is an external dependency of the reference,
 * not vendored: CMakeLists.txt:6): filter2D = correlation with the anchor at the kernel centre and
 * BORDER_REFLECT_101; GaussianBlur(ksize = 7 sigma + 1) = separable filter with the getGaussianKernel
 * coefficients exp(-x^2 / (2 sigma^2)) / sum; meanStdDev = population moments; normalize(NORM_L1) = division by
 * the sum of absolute values (left alone when that sum is 0).
 *
 * Places where the reference reads out of bounds (undefined behaviour) and what is done instead:
 *   findModesMeanShift :294, :319  fmod(i + j, n) is negative for i + j < 0  -> the histogram index wraps around
 *                                  (what the libcbdetect original, written with MATLAB's mod, does);
 *   edgeOrientations :264, :272    modes[2] is read when only two modes exist -> the two modes are kept.
 * PARITY UNPINNED (no reference build, no OpenCV here; see tscm_oracle.h).
 */
#include <math.h>
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#include <stdlib.h>
#include <string.h>

#include "tscm_oracle.h"

static int refl101(int i, int n)            /* BORDER_REFLECT_101: gfedcb|abcdefgh|gfedcba */
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
    return i;
}

static double normpdf_i(double x, int mu, int sigma)                     /* findCorner.cpp:195-198 */
{
    return exp(-(x - mu) * (x - mu) / 2 / sigma / sigma) / sqrt(2 * M_PI) / sigma;
}

/* :8-29 -- 3x3 derivative filters on the raw grey values, edge angle in [0, pi] and gradient magnitude */
void orc_corner_gradients(const unsigned char *gray, int w, int h, int stride, double *angle, double *weight)
{
    for (int i = 0; i < h; ++i) {
        const int im = refl101(i - 1, h), ip = refl101(i + 1, h);
        for (int j = 0; j < w; ++j) {
            const int jm = refl101(j - 1, w), jp = refl101(j + 1, w);
            const double du = ((double)gray[(size_t)im * stride + jp] - gray[(size_t)im * stride + jm])
                            + ((double)gray[(size_t)i * stride + jp] - gray[(size_t)i * stride + jm])
                            + ((double)gray[(size_t)ip * stride + jp] - gray[(size_t)ip * stride + jm]);
            const double dv = ((double)gray[(size_t)ip * stride + jm] - gray[(size_t)im * stride + jm])
                            + ((double)gray[(size_t)ip * stride + j] - gray[(size_t)im * stride + j])
                            + ((double)gray[(size_t)ip * stride + jp] - gray[(size_t)im * stride + jp]);
            double a = atan2(dv, du);
            if (a < 0) a += M_PI;
            if (a > M_PI) a -= M_PI;
            angle[(size_t)i * w + j] = a;
            weight[(size_t)i * w + j] = sqrt(dv * dv + du * du);
        }
    }
}

/* :30-34 -- (img - min) / (max - min) */
void orc_corner_normalise(const unsigned char *gray, int w, int h, int stride, double *img)
{
    double mn = 1e300, mx = -1e300;
    for (int i = 0; i < h; ++i)
        for (int j = 0; j < w; ++j) { const double v = gray[(size_t)i * stride + j]; if (v < mn) mn = v; if (v > mx) mx = v; }
    for (int i = 0; i < h; ++i)
        for (int j = 0; j < w; ++j) img[(size_t)i * w + j] = (gray[(size_t)i * stride + j] - mn) / (mx - mn);
}

void orc_gaussian_kernel(int sigma, double *k)        /* ksize = 7 sigma + 1 coefficients */
{
    const int n = 7 * sigma + 1;
    const double scale2x = -0.5 / ((double)sigma * sigma);
    double sum = 0;
    for (int i = 0; i < n; ++i) { const double x = i - (n - 1) * 0.5; k[i] = exp(scale2x * x * x); sum += k[i]; }
    sum = 1. / sum;
    for (int i = 0; i < n; ++i) k[i] *= sum;
}

/* :103-142 -- metric = cxy + c45 (the image the non-maximum suppression runs on) and Ixy (sub-pixel fit) */
int orc_corner_metric(const double *I, int w, int h, int sigma, double *metric, double *Ixy)
{
    const int n = 7 * sigma + 1, half = n / 2;
    const size_t N = (size_t)w * h;
    if (sigma < 1 || n % 2 == 0) return -2;            /* cv::GaussianBlur asserts an odd kernel size: sigma must be even (main.cpp:32 uses 4) */
    double *k = (double *)malloc(sizeof(double) * n);
    double *tmp = (double *)malloc(sizeof(double) * N), *Ig = (double *)malloc(sizeof(double) * N);
    double *Ix = (double *)malloc(sizeof(double) * N), *Iy = (double *)malloc(sizeof(double) * N), *I45 = (double *)malloc(sizeof(double) * N);
    if (!k || !tmp || !Ig || !Ix || !Iy || !I45) { free(k); free(tmp); free(Ig); free(Ix); free(Iy); free(I45); return -1; }
    orc_gaussian_kernel(sigma, k);
    for (int i = 0; i < h; ++i)                                   /* rows */
        for (int j = 0; j < w; ++j) {
            double s = 0;
            for (int q = 0; q < n; ++q) s += k[q] * I[(size_t)i * w + refl101(j + q - half, w)];
            tmp[(size_t)i * w + j] = s;
        }
    for (int i = 0; i < h; ++i)                                   /* columns (symmetric kernel) */
        for (int j = 0; j < w; ++j) {
            double s = k[half] * tmp[(size_t)i * w + j];
            for (int q = 1; q <= half && half + q < n; ++q)
                s += k[half + q] * (tmp[(size_t)refl101(i + q, h) * w + j] + tmp[(size_t)refl101(i - q, h) * w + j]);
            Ig[(size_t)i * w + j] = s;
        }
    const double c4 = cos(M_PI / 4), cn4 = cos(-M_PI / 4), s4 = sin(M_PI / 4), sn4 = sin(-M_PI / 4);
    for (int i = 0; i < h; ++i)
        for (int j = 0; j < w; ++j) {
            const size_t o = (size_t)i * w + j;
            Ix[o] = Ig[(size_t)i * w + refl101(j - 1, w)] - Ig[(size_t)i * w + refl101(j + 1, w)];     /* du = (1 0 -1) */
            Iy[o] = Ig[(size_t)refl101(i - 1, h) * w + j] - Ig[(size_t)refl101(i + 1, h) * w + j];
            I45[o] = Ix[o] * c4 + Iy[o] * s4;
        }
    for (int i = 0; i < h; ++i)
        for (int j = 0; j < w; ++j) {
            const size_t o = (size_t)i * w + j;
            const int im = refl101(i - 1, h), ip = refl101(i + 1, h), jm = refl101(j - 1, w), jp = refl101(j + 1, w);
            const double ixy = Ix[(size_t)im * w + j] - Ix[(size_t)ip * w + j];
            const double i45x = I45[(size_t)i * w + jm] - I45[(size_t)i * w + jp];
            const double i45y = I45[(size_t)im * w + j] - I45[(size_t)ip * w + j];
            const double i4545 = i45x * cn4 + i45y * sn4;
            const double in45 = Ix[o] * cn4 + Iy[o] * sn4;
            double cxy = sigma * sigma * fabs(ixy) - 1.5 * sigma * (fabs(I45[o]) + fabs(in45));
            if (cxy < 0) cxy = 0;
            double c45 = sigma * sigma * fabs(i4545) - 1.5 * sigma * (fabs(Ix[o]) + fabs(Iy[o]));
            if (c45 < 0) c45 = 0;
            metric[o] = cxy + c45;
            Ixy[o] = ixy;
        }
    free(k); free(tmp); free(Ig); free(Ix); free(Iy); free(I45);
    return 0;
}

/* :144-193 -- returns the number of maxima, at most cap are stored (x = column, y = row) */
int orc_corner_nms(const double *img, int width, int height, int n, double tau, int margin, int cap, double *px, double *py)
{
    int count = 0;
    for (int i = n + margin; i < width - n - margin; i += n + 1) {
        for (int j = n + margin; j < height - n - margin; j += n + 1) {
            int maxi = i, maxj = j;
            double maxval = img[(size_t)j * width + i];
            for (int i2 = i; i2 <= i + n; ++i2)
                for (int j2 = j; j2 <= j + n; ++j2) {
                    const double c = img[(size_t)j2 * width + i2];
                    if (c > maxval) { maxi = i2; maxj = j2; maxval = c; }
                }
            int failed = 0;
            const int i_end = maxi + n < width - margin ? maxi + n : width - margin;
            const int j_end = maxj + n < height - margin ? maxj + n : height - margin;
            for (int i2 = maxi - n; i2 < i_end && !failed; ++i2)
                for (int j2 = maxj - n; j2 < j_end; ++j2) {
                    const double c = img[(size_t)j2 * width + i2];
                    if (c > maxval && (i2 < i || i2 > i + n || j2 < j || j2 > j + n)) { failed = 1; break; }
                }
            if (maxval >= tau && !failed) {
                if (count < cap) { px[count] = maxi; py[count] = maxj; }
                ++count;
            }
        }
    }
    return count;
}

/* :286-349 -- modes of the smoothed circular histogram, strongest first; returns their number */
static int find_modes(const double *hist, int nb, int sigma, int *mode_bin, double *mode_val)
{
    double sm[64];
    const int reach = (int)round(2 * sigma);
    for (int i = 0; i < nb; ++i) {
        double sum = 0;
        for (int j = -reach; j <= reach; ++j) {
            const int idx = ((i + j) % nb + nb) % nb;
            sum += hist[idx] * normpdf_i(j, 0, sigma);
        }
        sm[i] = sum;
    }
    int failed = 1;
    for (int i = 1; i < nb; ++i) if (fabs(sm[i] - sm[0]) > 1e-5) { failed = 0; break; }
    if (failed) return 0;
    int nm = 0;
    for (int i = 0; i < nb; ++i) {
        int j = i;
        for (;;) {
            const double h0 = sm[j];
            const int j1 = (j + 1) % nb, j2 = (j - 1 + nb) % nb;
            const double h1 = sm[j1], h2 = sm[j2];
            if (h1 >= h0 && h1 >= h2) j = j1;
            else if (h2 > h0 && h2 > h1) j = j2;
            else break;
        }
        int seen = 0;
        for (int q = 0; q < nm; ++q) if (mode_bin[q] == j) { seen = 1; break; }
        if (!seen) { mode_bin[nm] = j; mode_val[nm] = sm[j]; ++nm; }
    }
    for (int a = 1; a < nm; ++a) {                 /* strongest first (insertion sort: stable, the reference's std::sort is not) */
        const int b = mode_bin[a]; const double v = mode_val[a];
        int q = a - 1;
        while (q >= 0 && mode_val[q] < v) { mode_bin[q + 1] = mode_bin[q]; mode_val[q + 1] = mode_val[q]; --q; }
        mode_bin[q + 1] = b; mode_val[q + 1] = v;
    }
    return nm;
}

/* :200-279 -- the two dominant edge directions around (cu, cv); v = (v1x, v1y, v2x, v2y) */
void orc_corner_orientation(const double *angle, const double *weight, int width, int height, int cu, int cv, int r, double *v)
{
    const int nb = 32;
    double hist[32];
    for (int b = 0; b < nb; ++b) hist[b] = 0;
    const int y1 = cv + r < height - 1 ? cv + r : height - 1, y0 = cv - r > 0 ? cv - r : 0;
    const int x1 = cu + r < width - 1 ? cu + r : width - 1, x0 = cu - r > 0 ? cu - r : 0;
    for (int i = y0; i <= y1; ++i)
        for (int j = x0; j <= x1; ++j) {
            double a = angle[(size_t)i * width + j] + M_PI / 2;
            if (a > M_PI) a -= M_PI;
            int bin = (int)floor(a / (M_PI / nb));
            if (bin > nb - 1) bin = nb - 1;
            if (bin < 0) bin = 0;
            hist[bin] += weight[(size_t)i * width + j];
        }
    v[0] = v[1] = v[2] = v[3] = 0;
    int mb[32]; double mv[32];
    const int nm = find_modes(hist, nb, 1, mb, mv);
    if (nm <= 1) return;
    const double z0 = mb[0] * M_PI / nb, z1 = mb[1] * M_PI / nb, z2 = nm > 2 ? mb[2] * M_PI / nb : 0;
    if (z0 > z1) {
        v[0] = cos(z1); v[1] = sin(z1); v[2] = cos(z0); v[3] = sin(z0);
        const double d = fmin(z0 - z1, z1 + M_PI - z0);
        if (d <= 0.3 && nm > 2) { v[0] = cos(z2); v[1] = sin(z2); }
    } else {
        v[0] = cos(z0); v[1] = sin(z0); v[2] = cos(z1); v[3] = sin(z1);
        const double d = fmin(z1 - z0, z0 + M_PI - z1);
        if (d <= 0.3 && nm > 2) { v[2] = cos(z2); v[3] = sin(z2); }
    }
}

/* :428-490 with :351-389 -- gradient score x intensity score of the (2 r + 1)^2 window centred on (u, v) */
double orc_corner_correlation_score(const double *img, const double *weight, int width, int u, int v, int r, const double *vv)
{
    const int n = 2 * r + 1, N = n * n;
    const double v1x = vv[0], v1y = vv[1], v2x = vv[2], v2y = vv[3];
    double *f = (double *)malloc(sizeof(double) * N), *wn = (double *)malloc(sizeof(double) * N);
    double mw = 0, mf = 0;
    for (int y = 0; y < n; ++y)
        for (int x = 0; x < n; ++x) {
            const double p0 = x - r, p1 = y - r;
            const double a = p0 * v1x + p1 * v1y, b = p0 * v2x + p1 * v2y;
            const double q0 = p0 - a * v1x, q1 = p1 - a * v1y, s0 = p0 - b * v2x, s1 = p1 - b * v2y;
            f[y * n + x] = (sqrt(q0 * q0 + q1 * q1) <= 1.5 || sqrt(s0 * s0 + s1 * s1) <= 1.5) ? 1.0 : -1.0;
            wn[y * n + x] = weight[(size_t)(v - r + y) * width + (u - r + x)];
            mw += wn[y * n + x]; mf += f[y * n + x];
        }
    mw /= N; mf /= N;
    double sw = 0, sf = 0;
    for (int q = 0; q < N; ++q) { sw += (wn[q] - mw) * (wn[q] - mw); sf += (f[q] - mf) * (f[q] - mf); }
    sw = sqrt(sw / N); sf = sqrt(sf / N);
    double sum = 0;
    for (int q = 0; q < N; ++q) sum += ((wn[q] - mw) / sw) * ((f[q] - mf) / sf);
    const double score_gradient = fmax(sum / (N - 1), 0.0);
    /* correlation patch */
    const double ang1 = atan2(v1y, v1x), ang2 = atan2(v2y, v2x);
    double t[4] = { 0, 0, 0, 0 }, nrm[4] = { 0, 0, 0, 0 };
    for (int y = 0; y < n; ++y)
        for (int x = 0; x < n; ++x) {
            const int du = x - r, dv = y - r;
            const double dist = sqrt((double)(du * du + dv * dv));
            const double s1 = -du * sin(ang1) + dv * cos(ang1), s2 = -du * sin(ang2) + dv * cos(ang2);
            int which = -1;
            if (s1 <= -0.1 && s2 <= -0.1) which = 0;
            else if (s1 >= 0.1 && s2 >= 0.1) which = 1;
            else if (s1 <= -0.1 && s2 >= 0.1) which = 2;
            else if (s1 >= 0.1 && s2 <= -0.1) which = 3;
            if (which < 0) continue;
            const double g = normpdf_i(dist, 0, r / 2);
            nrm[which] += g;
            t[which] += g * img[(size_t)(v - r + y) * width + (u - r + x)];
        }
    for (int q = 0; q < 4; ++q) t[q] = nrm[q] > 2.220446049250313e-16 ? t[q] / nrm[q] : 0.0;
    const double a1 = t[0], a2 = t[1], b1 = t[2], b2 = t[3];
    const double mu = (a1 + a2 + b1 + b2) / 4;
    const double score_1 = fmin(fmin(a1 - mu, a2 - mu), fmin(mu - b1, mu - b2));
    const double score_2 = fmin(fmin(mu - a1, mu - a2), fmin(b1 - mu, b2 - mu));
    const double score_intensity = fmax(fmax(score_1, score_2), 0.0);
    free(f); free(wn);
    return score_gradient * score_intensity;
}

/* :391-426 -- best of the three radii that fit into the image */
double orc_corner_score(const double *img, const double *weight, int width, int height, double px, double py, const double *vv)
{
    static const int radius[3] = { 8, 12, 16 };
    const int u = (int)round(px), v = (int)round(py);
    double best = 0;
    for (int j = 0; j < 3; ++j) {
        double s = 0;
        if (u >= radius[j] && u < width - radius[j] && v >= radius[j] && v < height - radius[j])
            s = orc_corner_correlation_score(img, weight, width, u, v, radius[j], vv);
        if (j == 0 || s > best) best = s;      /* sort + last element = maximum (NaN scores: see the tests) */
    }
    return best;
}

/* the 6 x 25 least-squares operator X = (A^T A)^-1 A^T of :495-509 (row index = (x + 2) * 5 + y + 2) */
void orc_subpixel_operator(double *X)
{
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
    for (int c = 0; c < 6; ++c) {                 /* Gauss-Jordan with partial pivoting */
        int p = c;
        for (int q = c + 1; q < 6; ++q) if (fabs(M[q][c]) > fabs(M[p][c])) p = q;
        if (p != c) for (int q = 0; q < 12; ++q) { const double t = M[c][q]; M[c][q] = M[p][q]; M[p][q] = t; }
        const double d = M[c][c];
        for (int q = 0; q < 12; ++q) M[c][q] /= d;
        for (int rr = 0; rr < 6; ++rr) {
            if (rr == c) continue;
            const double fct = M[rr][c];
            for (int q = 0; q < 12; ++q) M[rr][q] -= fct * M[c][q];
        }
    }
    for (int a = 0; a < 6; ++a)
        for (int q = 0; q < 25; ++q) {
            double s = 0;
            for (int b = 0; b < 6; ++b) s += M[a][6 + b] * A[q][b];
            X[a * 25 + q] = s;
        }
}

/* :510-539 -- quadratic fit of the 5x5 neighbourhood of Ixy around the (integer) corner; out = refined (x, y) */
void orc_corner_subpixel(const double *Ixy, int width, const double *X, double px, double py, double *out)
{
    double patch[25], beta[6];
    int cnt = 0;
    for (int j = (int)(px - 2); j <= px + 2; ++j)
        for (int k = (int)(py - 2); k <= py + 2; ++k) patch[cnt++] = Ixy[(size_t)k * width + j];
    for (int a = 0; a < 6; ++a) { double s = 0; for (int q = 0; q < 25; ++q) s += X[a * 25 + q] * patch[q]; beta[a] = s; }
    const double A = beta[0], B = beta[1], C = beta[2], D = beta[3], E = beta[4];
    double x = -(2 * B * C - D * E) / (4 * A * B - E * E);
    double y = -(2 * A * D - C * E) / (4 * A * B - E * E);
    if (fabs(x) > 2 || fabs(y) > 2) { x = 0; y = 0; }
    out[0] = px + x; out[1] = py + y;
}

/* findCorner :7-46 + :84 for every candidate: candidates in the order the suppression finds them.
 * Returns the number of candidates (at most cap are written).  v: [4 * cap], sub: [2 * cap]. */
int orc_detect_corners(const unsigned char *gray, int width, int height, int stride, int sigma, int cap,
                       double *px, double *py, double *v, double *score, double *sub, double *metric_out, double *ixy_out)
{
    const size_t N = (size_t)width * height;
    double *angle = (double *)malloc(sizeof(double) * N), *weight = (double *)malloc(sizeof(double) * N);
    double *img = (double *)malloc(sizeof(double) * N), *metric = (double *)malloc(sizeof(double) * N), *Ixy = (double *)malloc(sizeof(double) * N);
    int n = -1;
    if (angle && weight && img && metric && Ixy) {
        orc_corner_gradients(gray, width, height, stride, angle, weight);
        orc_corner_normalise(gray, width, height, stride, img);
        if (orc_corner_metric(img, width, height, sigma, metric, Ixy) == 0) {
            n = orc_corner_nms(metric, width, height, 4, 0.07, 5, cap, px, py);
            const int m = n < cap ? n : cap;
            double X[150];
            orc_subpixel_operator(X);
            for (int q = 0; q < m; ++q) {
                orc_corner_orientation(angle, weight, width, height, (int)px[q], (int)py[q], 10, v + 4 * q);
                score[q] = orc_corner_score(img, weight, width, height, px[q], py[q], v + 4 * q);
                orc_corner_subpixel(Ixy, width, X, px[q], py[q], sub + 2 * q);
            }
            if (metric_out) memcpy(metric_out, metric, sizeof(double) * N);
            if (ixy_out) memcpy(ixy_out, Ixy, sizeof(double) * N);
        }
    }
    free(angle); free(weight); free(img); free(metric); free(Ixy);
    return n;
}

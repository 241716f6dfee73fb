/*
 * tscm_oracle_boards.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE), chessboard structure recovery.
 *
 * Plain-C restatement of DetectCorner/chessboard.cpp (second stage):
 *   chessboardsFromCorners :3-103, initChessboard :105-149, average / stdd :151-170, directionalNeighbor :172-215,
 *   chessboardEnergy :217-253, growChessboard :255-398, predictCorners :400-414, assignClosestCorners :416-447.
 * Boards are matrices of corner indices; like the reference, index 0 doubles as "empty cell" (corner 0 is never
 * counted as used, :185, :270), the column-direction energy terms go through an integer cv::Point (:244: values
 * rounded half to even), and a board replaced by a better overlapping one is zeroed and dropped (:59-76).
 * PARITY UNPINNED (no reference build, see tscm_oracle.h).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "tscm_oracle.h"

typedef struct { int rows, cols; int *c; } board_t;       /* c[r * cols + k] */

typedef struct { int n; const double *px, *py, *v1, *v2; } corners_t;

static board_t board_new(int rows, int cols)
{
    board_t b; b.rows = rows; b.cols = cols; b.c = (int *)calloc((size_t)rows * cols, sizeof(int));
    return b;
}
static board_t board_clone(const board_t *a)
{
    board_t b = board_new(a->rows, a->cols);
    memcpy(b.c, a->c, sizeof(int) * (size_t)a->rows * a->cols);
    return b;
}
static void board_free(board_t *b) { free(b->c); b->c = NULL; b->rows = b->cols = 0; }

/* indices not on the board, ascending (:174-197, :259-282) */
static int unused_list(const board_t *b, int n, int *unused)
{
    int m = 0;
    for (int i = 0; i < n; ++i) {
        int used = 0;
        if (i != 0)
            for (int q = 0; q < b->rows * b->cols; ++q) if (b->c[q] == i) { used = 1; break; }
        if (!used) unused[m++] = i;
    }
    return m;
}

static void directional_neighbor(int idx, double vx, double vy, const board_t *b, const corners_t *cs, int *neighbor_idx, double *min_dist)
{
    int *unused = (int *)malloc(sizeof(int) * cs->n);
    const int m = unused_list(b, cs->n, unused);
    int best = 0; double bd = 0;
    for (int i = 0; i < m; ++i) {
        const double dx = cs->px[unused[i]] - cs->px[idx], dy = cs->py[unused[i]] - cs->py[idx];
        double d = dx * vx + dy * vy;
        const double ex = dx - d * vx, ey = dy - d * vy;
        const double de = sqrt(ex * ex + ey * ey);
        if (d < 0) d = 1e10;
        d = d + 5 * de;
        if (i == 0 || d < bd) { bd = d; best = i; }
    }
    *min_dist = bd; *neighbor_idx = unused[best];
    free(unused);
}

static double average(const double *a, int n) { double s = 0; for (int i = 0; i < n; ++i) s += a[i]; return s / n; }
static double stdd(const double *a, int n, double mean)
{
    double s = 0;
    for (int i = 0; i < n; ++i) s += (a[i] - mean) * (a[i] - mean);
    return sqrt(s / (n - 1));
}

static board_t init_chessboard(const corners_t *cs, int idx)
{
    board_t b = board_new(3, 3);
    if (cs->n < 9) return b;
    const double v1x = cs->v1[2 * idx], v1y = cs->v1[2 * idx + 1], v2x = cs->v2[2 * idx], v2y = cs->v2[2 * idx + 1];
    double d1[2] = { 0, 0 }, d2[6] = { 0, 0, 0, 0, 0, 0 };
    int nb = 0;
    b.c[1 * 3 + 1] = idx;
    directional_neighbor(idx, v1x, v1y, &b, cs, &nb, &d1[0]); b.c[1 * 3 + 2] = nb;
    directional_neighbor(idx, -v1x, -v1y, &b, cs, &nb, &d1[1]); b.c[1 * 3 + 0] = nb;
    directional_neighbor(idx, v2x, v2y, &b, cs, &nb, &d2[0]); b.c[2 * 3 + 1] = nb;
    directional_neighbor(idx, -v2x, -v2y, &b, cs, &nb, &d2[1]); b.c[0 * 3 + 1] = nb;
    directional_neighbor(b.c[1 * 3 + 0], -v2x, -v2y, &b, cs, &nb, &d2[2]); b.c[0 * 3 + 0] = nb;
    directional_neighbor(b.c[1 * 3 + 0], v2x, v2y, &b, cs, &nb, &d2[3]); b.c[2 * 3 + 0] = nb;
    directional_neighbor(b.c[1 * 3 + 2], -v2x, -v2y, &b, cs, &nb, &d2[4]); b.c[0 * 3 + 2] = nb;
    directional_neighbor(b.c[1 * 3 + 2], v2x, v2y, &b, cs, &nb, &d2[5]); b.c[2 * 3 + 2] = nb;
    const double a1 = average(d1, 2), s1 = stdd(d1, 2, a1);
    const double a2 = average(d2, 6), s2 = stdd(d2, 6, a2);
    if (s1 / a1 > 0.3 || s2 / a2 > 0.3) memset(b.c, 0, sizeof(int) * 9);
    return b;
}

static double chessboard_energy(const board_t *b, const corners_t *cs)
{
    double E = 0;
    for (int j = 0; j < b->rows; ++j)
        for (int k = 0; k < b->cols - 2; ++k) {
            const int i0 = b->c[j * b->cols + k], i1 = b->c[j * b->cols + k + 1], i2 = b->c[j * b->cols + k + 2];
            const double ax = cs->px[i0] + cs->px[i2] - 2 * cs->px[i1], ay = cs->py[i0] + cs->py[i2] - 2 * cs->py[i1];
            const double bx = cs->px[i0] - cs->px[i2], by = cs->py[i0] - cs->py[i2];
            const double r = sqrt(ax * ax + ay * ay) / sqrt(bx * bx + by * by);
            if (E < r) E = r;
        }
    for (int j = 0; j < b->cols; ++j)
        for (int k = 0; k < b->rows - 2; ++k) {
            const int i0 = b->c[k * b->cols + j], i1 = b->c[(k + 1) * b->cols + j], i2 = b->c[(k + 2) * b->cols + j];
            /* cv::Point (int): saturate_cast<int>(double) = nearest, ties to even; the products are int products (:244-248) */
            const int ax = (int)lrint(cs->px[i0] + cs->px[i2] - 2 * cs->px[i1]), ay = (int)lrint(cs->py[i0] + cs->py[i2] - 2 * cs->py[i1]);
            const double n1 = sqrt((double)(ax * ax + ay * ay));
            const int bx = (int)lrint(cs->px[i0] - cs->px[i2]), by = (int)lrint(cs->py[i0] - cs->py[i2]);
            const double n2 = sqrt((double)(bx * bx + by * by));
            const double r = n1 / n2;
            if (E < r) E = r;
        }
    return b->rows * b->cols * (E - 1);
}

static void predict_corner(const corners_t *cs, int i1, int i2, int i3, double *out)
{
    const double v1x = cs->px[i2] - cs->px[i1], v1y = cs->py[i2] - cs->py[i1];
    const double v2x = cs->px[i3] - cs->px[i2], v2y = cs->py[i3] - cs->py[i2];
    const double a1 = atan2(v1y, v1x), a2 = atan2(v2y, v2x), a3 = 2 * a2 - a1;
    const double s1 = sqrt(v1x * v1x + v1y * v1y), s2 = sqrt(v2x * v2x + v2y * v2y), s3 = 2 * s2 - s1;
    out[0] = cs->px[i3] + 0.75 * s3 * cos(a3);
    out[1] = cs->py[i3] + 0.75 * s3 * sin(a3);
}

/* :416-447 -- greedy global-minimum assignment; returns 0 when there are fewer candidates than predictions */
static int assign_closest(const corners_t *cs, const int *unused, int m, const double *pred, int np, int *idx)
{
    if (m < np) return 0;
    double *D = (double *)malloc(sizeof(double) * (size_t)m * np);      /* D[j * np + i]: candidate j, prediction i */
    for (int i = 0; i < np; ++i)
        for (int j = 0; j < m; ++j) {
            const double dx = cs->px[unused[j]] - pred[2 * i], dy = cs->py[unused[j]] - pred[2 * i + 1];
            D[(size_t)j * np + i] = sqrt(dx * dx + dy * dy);
        }
    for (int it = 0; it < np; ++it) {
        int bj = 0, bi = 0; double bv = D[0];
        for (int j = 0; j < m; ++j)
            for (int i = 0; i < np; ++i) if (D[(size_t)j * np + i] < bv) { bv = D[(size_t)j * np + i]; bj = j; bi = i; }
        idx[bi] = bj;
        for (int i = 0; i < np; ++i) D[(size_t)bj * np + i] = 1e10;
        for (int j = 0; j < m; ++j) D[(size_t)j * np + bi] = 1e10;
    }
    free(D);
    return 1;
}

static board_t grow_chessboard(const board_t *b, const corners_t *cs, int border)
{
    if (b->c[0] == 0 && b->c[1] == 0) return board_clone(b);
    int *unused = (int *)malloc(sizeof(int) * cs->n);
    const int m = unused_list(b, cs->n, unused);
    const int R = b->rows, Cn = b->cols;
    const int np = (border == 0 || border == 2) ? R : Cn;
    double *pred = (double *)malloc(sizeof(double) * 2 * np);
    int *idx = (int *)malloc(sizeof(int) * np);
    for (int i = 0; i < np; ++i) {
        switch (border) {
        case 0: predict_corner(cs, b->c[i * Cn + Cn - 3], b->c[i * Cn + Cn - 2], b->c[i * Cn + Cn - 1], pred + 2 * i); break;
        case 1: predict_corner(cs, b->c[(R - 3) * Cn + i], b->c[(R - 2) * Cn + i], b->c[(R - 1) * Cn + i], pred + 2 * i); break;
        case 2: predict_corner(cs, b->c[i * Cn + 2], b->c[i * Cn + 1], b->c[i * Cn + 0], pred + 2 * i); break;
        default: predict_corner(cs, b->c[2 * Cn + i], b->c[1 * Cn + i], b->c[0 * Cn + i], pred + 2 * i); break;
        }
    }
    board_t out;
    if (!assign_closest(cs, unused, m, pred, np, idx)) out = board_clone(b);
    else {
        const int nr = (border == 1 || border == 3) ? R + 1 : R, nc = (border == 0 || border == 2) ? Cn + 1 : Cn;
        out = board_new(nr, nc);
        const int r0 = border == 3 ? 1 : 0, c0 = border == 2 ? 1 : 0;
        for (int r = 0; r < R; ++r) for (int c = 0; c < Cn; ++c) out.c[(r + r0) * nc + c + c0] = b->c[r * Cn + c];
        for (int i = 0; i < np; ++i) {
            const int v = unused[idx[i]];
            if (border == 0) out.c[i * nc + Cn] = v;
            else if (border == 1) out.c[R * nc + i] = v;
            else if (border == 2) out.c[i * nc + 0] = v;
            else out.c[0 * nc + i] = v;
        }
    }
    free(unused); free(pred); free(idx);
    return out;
}

/* :3-103.  Returns the number of boards (at most max_boards are written); board q: rows[q] x cols[q] indices at
 * cells + q * max_cells (row-major). */
int orc_chessboards_from_corners(int n, const double *px, const double *py, const double *v1, const double *v2,
                                 int max_boards, int max_cells, int *rows, int *cols, int *cells)
{
    corners_t cs = { n, px, py, v1, v2 };
    board_t *list = NULL; int nl = 0;
    for (int i = 0; i < n; ++i) {
        board_t b = init_chessboard(&cs, i);
        if ((b.c[0] == 0 && b.c[1] == 0) || chessboard_energy(&b, &cs) > 0) { board_free(&b); continue; }
        for (;;) {
            const double energy = chessboard_energy(&b, &cs);
            board_t prop[4]; double pe[4];
            int mi = 0;
            for (int j = 0; j < 4; ++j) { prop[j] = grow_chessboard(&b, &cs, j); pe[j] = chessboard_energy(&prop[j], &cs); if (pe[j] < pe[mi]) mi = j; }
            const int better = pe[mi] < energy;
            if (better) { board_free(&b); b = prop[mi]; }
            for (int j = 0; j < 4; ++j) if (!(better && j == mi)) board_free(&prop[j]);
            if (!better) break;
        }
        const double eb = chessboard_energy(&b, &cs);
        int keep = 0;
        if (eb < -10) {
            if (nl > 0) {
                int overlapped = 0, lower = 0;
                for (int j = 0; j < nl; ++j) {
                    int shared = 0;
                    for (int k = 0; k < list[j].rows * list[j].cols && !shared; ++k)
                        for (int q = 0; q < b.rows * b.cols; ++q) if (b.c[q] == list[j].c[k]) { shared = 1; break; }
                    if (!shared) continue;
                    overlapped = 1;
                    if (chessboard_energy(&list[j], &cs) > eb) { board_free(&list[j]); lower = 1; }        /* zeroed, dropped below */
                }
                keep = !overlapped || lower;
            } else keep = 1;
        }
        if (keep) { list = (board_t *)realloc(list, sizeof(board_t) * (nl + 1)); list[nl++] = b; } else board_free(&b);
        int w = 0;
        for (int j = 0; j < nl; ++j) if (list[j].c) list[w++] = list[j];
        nl = w;
    }
    for (int q = 0; q < nl; ++q) {                      /* :81-101: at least as many columns as rows */
        board_t *b = &list[q];
        if (b->cols < b->rows) {
            board_t t = board_new(b->cols, b->rows);
            for (int j = 0; j < t.rows; ++j) for (int k = 0; k < t.cols; ++k) t.c[j * t.cols + k] = b->c[(b->rows - k - 1) * b->cols + j];
            board_free(b); *b = t;
        }
        if (q < max_boards && b->rows * b->cols <= max_cells) {
            rows[q] = b->rows; cols[q] = b->cols;
            memcpy(cells + (size_t)q * max_cells, b->c, sizeof(int) * (size_t)b->rows * b->cols);
        } else if (q < max_boards) { rows[q] = cols[q] = 0; }
    }
    const int total = nl;
    for (int q = 0; q < nl; ++q) board_free(&list[q]);
    free(list);
    return total;
}

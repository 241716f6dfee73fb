// tscm_boards.cpp -- chessboard structure recovery from corner candidates (second stage):
// chessboardsFromCorners of DetectCorner/chessboard.cpp:3-103 with its helpers (initChessboard :105-149,
// directionalNeighbor :172-215, chessboardEnergy :217-253, growChessboard :255-398, predictCorners :400-414,
// assignClosestCorners :416-447).  Sequential host logic on a few hundred candidates at most (the reference runs it
// on the CPU as well); its input is the output of tscm_detect_corners.
//
// Conventions kept from the reference because they decide which boards come out: candidate 0 doubles as the "empty
// cell" marker and is therefore never counted as used; the column-direction terms of the energy are evaluated on
// integer-rounded differences (cv::Point, :244); of two overlapping boards the one with the lower energy survives.
#include "tscm/tscm.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

int tscm_set_error(int code, const std::string &msg);   // tscm_solver.hip

namespace {

struct Board {
    int rows = 0, cols = 0;
    std::vector<int> c;                                   // row-major candidate indices
    int &at(int r, int k) { return c[(size_t)r * cols + k]; }
    int at(int r, int k) const { return c[(size_t)r * cols + k]; }
    bool blank() const { return c.size() < 2 || (c[0] == 0 && c[1] == 0); }      // the reference's "zero board" test (:10, :257)
};

struct Cands {
    int n;
    const double *x, *y, *v1, *v2;
};

// candidates that are not on the board, ascending; candidate 0 is always among them
void free_candidates(const Board &b, int n, std::vector<int> &out)
{
    std::vector<char> used((size_t)n, 0);
    for (int v : b.c) if (v > 0 && v < n) used[v] = 1;
    out.clear();
    for (int i = 0; i < n; ++i) if (!used[i]) out.push_back(i);
}

// nearest free candidate in direction v from candidate idx: distance along v + 5 x distance across (:198-214)
void neighbour_towards(const Cands &cs, const Board &b, int idx, double vx, double vy, std::vector<int> &scratch, int &who, double &cost)
{
    free_candidates(b, cs.n, scratch);
    who = scratch[0]; cost = 0;
    bool first = true;
    for (int i : scratch) {
        const double dx = cs.x[i] - cs.x[idx], dy = cs.y[i] - cs.y[idx];
        double along = dx * vx + dy * vy;
        const double ex = dx - along * vx, ey = dy - along * vy;
        const double across = std::sqrt(ex * ex + ey * ey);
        if (along < 0) along = 1e10;
        const double d = along + 5 * across;
        if (first || d < cost) { cost = d; who = i; first = false; }
    }
}

double spread(const double *a, int n)         // sample standard deviation / mean (:129-141)
{
    double mean = 0;
    for (int i = 0; i < n; ++i) mean += a[i];
    mean /= n;
    double s = 0;
    for (int i = 0; i < n; ++i) s += (a[i] - mean) * (a[i] - mean);
    return std::sqrt(s / (n - 1)) / mean;
}

Board seed_board(const Cands &cs, int idx, std::vector<int> &scratch)
{
    Board b; b.rows = b.cols = 3; b.c.assign(9, 0);
    if (cs.n < 9) return b;
    const double ax = cs.v1[2 * idx], ay = cs.v1[2 * idx + 1], bx = cs.v2[2 * idx], by = cs.v2[2 * idx + 1];
    double d1[2], d2[6];
    b.at(1, 1) = idx;
    neighbour_towards(cs, b, idx, ax, ay, scratch, b.at(1, 2), d1[0]);
    neighbour_towards(cs, b, idx, -ax, -ay, scratch, b.at(1, 0), d1[1]);
    neighbour_towards(cs, b, idx, bx, by, scratch, b.at(2, 1), d2[0]);
    neighbour_towards(cs, b, idx, -bx, -by, scratch, b.at(0, 1), d2[1]);
    neighbour_towards(cs, b, b.at(1, 0), -bx, -by, scratch, b.at(0, 0), d2[2]);
    neighbour_towards(cs, b, b.at(1, 0), bx, by, scratch, b.at(2, 0), d2[3]);
    neighbour_towards(cs, b, b.at(1, 2), -bx, -by, scratch, b.at(0, 2), d2[4]);
    neighbour_towards(cs, b, b.at(1, 2), bx, by, scratch, b.at(2, 2), d2[5]);
    if (spread(d1, 2) > 0.3 || spread(d2, 6) > 0.3) b.c.assign(9, 0);
    return b;
}

// rows * cols * (max over consecutive triples of |x0 + x2 - 2 x1| / |x0 - x2|  -  1)
double energy(const Cands &cs, const Board &b)
{
    double worst = 0;
    for (int r = 0; r < b.rows; ++r)
        for (int k = 0; k + 2 < b.cols; ++k) {
            const int i0 = b.at(r, k), i1 = b.at(r, k + 1), i2 = b.at(r, k + 2);
            const double ux = cs.x[i0] + cs.x[i2] - 2 * cs.x[i1], uy = cs.y[i0] + cs.y[i2] - 2 * cs.y[i1];
            const double wx = cs.x[i0] - cs.x[i2], wy = cs.y[i0] - cs.y[i2];
            const double q = std::sqrt(ux * ux + uy * uy) / std::sqrt(wx * wx + wy * wy);
            if (worst < q) worst = q;
        }
    for (int k = 0; k < b.cols; ++k)
        for (int r = 0; r + 2 < b.rows; ++r) {
            const int i0 = b.at(r, k), i1 = b.at(r + 1, k), i2 = b.at(r + 2, k);
            const long ux = std::lrint(cs.x[i0] + cs.x[i2] - 2 * cs.x[i1]), uy = std::lrint(cs.y[i0] + cs.y[i2] - 2 * cs.y[i1]);     // integer cv::Point
            const long wx = std::lrint(cs.x[i0] - cs.x[i2]), wy = std::lrint(cs.y[i0] - cs.y[i2]);
            const double q = std::sqrt((double)((int)ux * (int)ux + (int)uy * (int)uy)) / std::sqrt((double)((int)wx * (int)wx + (int)wy * (int)wy));
            if (worst < q) worst = q;
        }
    return b.rows * b.cols * (worst - 1);
}

// extrapolation of three consecutive corners: turn and stretch continue, 3/4 of the step (:400-414)
void extrapolate(const Cands &cs, int i1, int i2, int i3, double &px, double &py)
{
    const double ax = cs.x[i2] - cs.x[i1], ay = cs.y[i2] - cs.y[i1], bx = cs.x[i3] - cs.x[i2], by = cs.y[i3] - cs.y[i2];
    const double a1 = std::atan2(ay, ax), a2 = std::atan2(by, bx), a3 = 2 * a2 - a1;
    const double s3 = 2 * std::sqrt(bx * bx + by * by) - std::sqrt(ax * ax + ay * ay);
    px = cs.x[i3] + 0.75 * s3 * std::cos(a3);
    py = cs.y[i3] + 0.75 * s3 * std::sin(a3);
}

// one more row or column on side `side` (0 right, 1 bottom, 2 left, 3 top); the board itself when the free
// candidates do not suffice (:255-398, :416-447)
Board grown(const Cands &cs, const Board &b, int side, std::vector<int> &freec, std::vector<double> &D)
{
    if (b.blank()) return b;
    free_candidates(b, cs.n, freec);
    const int R = b.rows, Cn = b.cols, np = (side == 0 || side == 2) ? R : Cn, m = (int)freec.size();
    if (m < np) return b;
    std::vector<double> pred(2 * (size_t)np);
    for (int i = 0; i < np; ++i) {
        switch (side) {
        case 0: extrapolate(cs, b.at(i, Cn - 3), b.at(i, Cn - 2), b.at(i, Cn - 1), pred[2 * i], pred[2 * i + 1]); break;
        case 1: extrapolate(cs, b.at(R - 3, i), b.at(R - 2, i), b.at(R - 1, i), pred[2 * i], pred[2 * i + 1]); break;
        case 2: extrapolate(cs, b.at(i, 2), b.at(i, 1), b.at(i, 0), pred[2 * i], pred[2 * i + 1]); break;
        default: extrapolate(cs, b.at(2, i), b.at(1, i), b.at(0, i), pred[2 * i], pred[2 * i + 1]); break;
        }
    }
    // greedy assignment: repeatedly the globally closest (candidate, prediction) pair, first one in candidate-major order
    D.resize((size_t)m * np);
    for (int j = 0; j < m; ++j)
        for (int i = 0; i < np; ++i) {
            const double dx = cs.x[freec[j]] - pred[2 * i], dy = cs.y[freec[j]] - pred[2 * i + 1];
            D[(size_t)j * np + i] = std::sqrt(dx * dx + dy * dy);
        }
    std::vector<int> pick(np, 0);
    for (int it = 0; it < np; ++it) {
        size_t arg = 0;
        for (size_t q = 1; q < D.size(); ++q) if (D[q] < D[arg]) arg = q;
        const int bj = (int)(arg / np), bi = (int)(arg % np);
        pick[bi] = bj;
        for (int i = 0; i < np; ++i) D[(size_t)bj * np + i] = 1e10;
        for (int j = 0; j < m; ++j) D[(size_t)j * np + bi] = 1e10;
    }
    Board g;
    g.rows = (side == 1 || side == 3) ? R + 1 : R;
    g.cols = (side == 0 || side == 2) ? Cn + 1 : Cn;
    g.c.assign((size_t)g.rows * g.cols, 0);
    const int r0 = side == 3 ? 1 : 0, c0 = side == 2 ? 1 : 0;
    for (int r = 0; r < R; ++r) for (int k = 0; k < Cn; ++k) g.at(r + r0, k + c0) = b.at(r, k);
    for (int i = 0; i < np; ++i) {
        const int v = freec[pick[i]];
        if (side == 0) g.at(i, Cn) = v; else if (side == 1) g.at(R, i) = v; else if (side == 2) g.at(i, 0) = v; else g.at(0, i) = v;
    }
    return g;
}

}  // namespace

extern "C" void tscm_chessboards_free(tscm_chessboards *b)
{
    if (!b) return;
    std::free(b->rows); std::free(b->cols); std::free(b->offset); std::free(b->cells);
    b->rows = b->cols = b->offset = b->cells = nullptr;
    b->n_boards = 0;
}

extern "C" int tscm_chessboards_from_corners(int n, const double *x, const double *y, const double *v1, const double *v2, tscm_chessboards *out)
{
    if (!out) return tscm_set_error(TSCM_E_INVALID, "NULL argument");
    std::memset(out, 0, sizeof(*out));
    if (n < 0 || (n > 0 && (!x || !y || !v1 || !v2))) return tscm_set_error(TSCM_E_INVALID, "NULL argument");
    if (n > 65535) return tscm_set_error(TSCM_E_UNSUPPORTED, "more than 65535 candidates (the reference stores the indices as uint16)");
    const Cands cs = { n, x, y, v1, v2 };
    std::vector<Board> boards;
    std::vector<int> scratch;
    std::vector<double> D;
    for (int i = 0; i < n; ++i) {
        Board b = seed_board(cs, i, scratch);
        if (b.blank() || energy(cs, b) > 0) continue;
        for (;;) {                                           // grow on the side that lowers the energy most (:14-32)
            const double e = energy(cs, b);
            Board best; double be = 0; bool have = false;
            for (int side = 0; side < 4; ++side) {
                Board g = grown(cs, b, side, scratch, D);
                const double ge = energy(cs, g);
                if (!have || ge < be) { best.rows = g.rows; best.cols = g.cols; best.c.swap(g.c); be = ge; have = true; }
            }
            if (be < e) b = best; else break;
        }
        const double eb = energy(cs, b);
        if (!(eb < -10)) continue;
        bool overlapped = false, replaced = false;
        for (Board &o : boards) {
            bool shared = false;
            for (int v : o.c) if (std::find(b.c.begin(), b.c.end(), v) != b.c.end()) { shared = true; break; }
            if (!shared) continue;
            overlapped = true;
            if (energy(cs, o) > eb) { o.c.clear(); o.rows = o.cols = 0; replaced = true; }
        }
        if (!overlapped || replaced) boards.push_back(b);
        boards.erase(std::remove_if(boards.begin(), boards.end(), [](const Board &o) { return o.c.empty(); }), boards.end());
    }
    // at least as many columns as rows (:81-101)
    for (Board &b : boards) {
        if (b.cols >= b.rows) continue;
        Board t; t.rows = b.cols; t.cols = b.rows; t.c.assign(b.c.size(), 0);
        for (int j = 0; j < t.rows; ++j) for (int k = 0; k < t.cols; ++k) t.at(j, k) = b.at(b.rows - k - 1, j);
        b = t;
    }
    const size_t nb = boards.size();
    size_t total = 0;
    for (const Board &b : boards) total += b.c.size();
    out->rows = static_cast<int *>(std::calloc(nb ? nb : 1, sizeof(int))); out->cols = static_cast<int *>(std::calloc(nb ? nb : 1, sizeof(int)));
    out->offset = static_cast<int *>(std::calloc(nb + 1, sizeof(int))); out->cells = static_cast<int *>(std::calloc(total ? total : 1, sizeof(int)));
    if (!out->rows || !out->cols || !out->offset || !out->cells) { tscm_chessboards_free(out); return tscm_set_error(TSCM_E_NOMEM, "out of memory"); }
    size_t o = 0;
    for (size_t q = 0; q < nb; ++q) {
        out->rows[q] = boards[q].rows; out->cols[q] = boards[q].cols; out->offset[q] = (int)o;
        std::memcpy(out->cells + o, boards[q].c.data(), sizeof(int) * boards[q].c.size());
        o += boards[q].c.size();
    }
    out->offset[nb] = (int)o;
    out->n_boards = (int)nb;
    return 0;
}

// CPU check of the reduced solver's elimination plan (tscm_calib_amd/csrc/tscm_nd_plan.h): builds a random SPD camera
// system with a given camera-pair graph, runs the tile algorithm of k_solve_nd phase by phase exactly as the plan schedules
// it -- the solved panel columns published by their owners (step A), the trailing updates as products of two of them
// (step B), look-ahead factorisation of the next phase's diagonal tiles from the raw tiles handed to the look-ahead lanes,
// the per-tile update masks, the packed factor and the level-wise back-substitution -- and compares the solution with a
// dense Cholesky solve.
// Host logic only (no GPU): what the kernel does per thread is done here per tile, between the same barriers.
//   usage: nd_plan_check <C> <graph: ring|chain|complete|star|pairs:a-b,c-d,...> <const_mask> <inactive_mask> <dense_only> <seed>
// prints one JSON line.
#include "../../tscm_calib_amd/csrc/tscm_nd_plan.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <random>
#include <string>

using namespace tscm;

struct Tile4 { double v[4][4]; };
static void solve_tile(const Tile4 &A, const double L[4][4], const double il[4], Tile4 &X)
{
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            double v = A.v[r][c];
            for (int q = 0; q < c; ++q) v -= X.v[r][q] * L[c][q];
            X.v[r][c] = v * il[c];
        }
}
static bool factor_tile(Tile4 &t, double L[4][4], double il[4])
{
    bool ok = true;
    for (int c = 0; c < 4; ++c) {
        double d = t.v[c][c];
        for (int q = 0; q < c; ++q) d -= t.v[c][q] * t.v[c][q];
        if (!(d > 0.0)) { ok = false; d = 1.0; }
        const double isd = 1.0 / std::sqrt(d);
        t.v[c][c] = d * isd; il[c] = isd;
        for (int r = c + 1; r < 4; ++r) {
            double v = t.v[r][c];
            for (int q = 0; q < c; ++q) v -= t.v[r][q] * t.v[c][q];
            t.v[r][c] = v * isd;
        }
    }
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) L[r][c] = c <= r ? t.v[r][c] : 0.0;
    return ok;
}

int main(int argc, char **argv)
{
    if (argc < 7) { std::fprintf(stderr, "usage\n"); return 2; }
    const int C = std::atoi(argv[1]);
    const std::string graph = argv[2];
    const unsigned const_mask = (unsigned)std::strtoul(argv[3], nullptr, 0), inactive = (unsigned)std::strtoul(argv[4], nullptr, 0);
    const bool dense_only = std::atoi(argv[5]) != 0;
    std::mt19937_64 rng((unsigned long long)std::atoll(argv[6]));
    std::vector<unsigned char> pair((size_t)C * C, 0);
    for (int m = 0; m < C; ++m) pair[m * C + m] = 1;
    auto link = [&](int a, int b) { if (a > b) std::swap(a, b); if (a != b) pair[a * C + b] = 1; };
    if (graph == "ring") for (int m = 0; m < C; ++m) link(m, (m + 1) % C);
    else if (graph == "chain") for (int m = 0; m + 1 < C; ++m) link(m, m + 1);
    else if (graph == "complete") for (int a = 0; a < C; ++a) for (int b = a + 1; b < C; ++b) link(a, b);
    else if (graph == "star") for (int m = 1; m < C; ++m) link(0, m);
    else if (graph.rfind("pairs:", 0) == 0) {
        const char *s = graph.c_str() + 6;
        while (*s) { int a = 0, b = 0, n = 0; if (std::sscanf(s, "%d-%d%n", &a, &b, &n) != 2) break; link(a, b); s += n; if (*s == ',') ++s; }
    }
    std::vector<int> ncols(C), col0(C), bid_of((size_t)C * C, -1);
    for (int m = 0; m < C; ++m) {
        const bool cst = (const_mask >> m) & 1u, off = (inactive >> m) & 1u;
        ncols[m] = off ? 0 : cst ? 7 : 13;
        col0[m] = 16 * m + (cst ? 6 : 0);
    }
    int n_bids = 0;
    for (int a = 0; a < C; ++a) for (int b = a; b < C; ++b) if (pair[a * C + b]) bid_of[a * C + b] = n_bids++;
    // the two plans exactly as tscm_solver_create builds them (graph order with the dense plan as its fall-back)
    NdPlan plans[2];
    bool fell_back = false;
    if (!nd_build_plans(C, ncols.data(), col0.data(), pair.data(), bid_of.data(), plans, &fell_back)) { std::printf("{\"ok\": false, \"reason\": \"does not fit\"}\n"); return 0; }
    const NdPlan &pl = plans[dense_only ? 1 : 0];
    fell_back = fell_back && !dense_only;
    // ---- a random SPD system on the padded columns with exactly this block structure ------------------------------------------
    const int n_pad = 16 * C;
    std::vector<double> A((size_t)n_pad * n_pad, 0.0), b(n_pad, 0.0);
    std::normal_distribution<double> nd(0.0, 1.0);
    std::vector<int> free_cols;
    for (int m = 0; m < C; ++m) for (int j = 0; j < ncols[m]; ++j) free_cols.push_back(col0[m] + j);
    {
        // A = sum over present pairs of G_p^T G_p (G_p touches the two cameras' free columns) + diagonal
        for (int a = 0; a < C; ++a)
            for (int c2 = a; c2 < C; ++c2) {
                if (!pair[a * C + c2] || !ncols[a] || !ncols[c2]) continue;
                std::vector<int> cols;
                for (int j = 0; j < ncols[a]; ++j) cols.push_back(col0[a] + j);
                if (c2 != a) for (int j = 0; j < ncols[c2]; ++j) cols.push_back(col0[c2] + j);
                for (int rep = 0; rep < 30; ++rep) {
                    std::vector<double> g(cols.size());
                    for (auto &x : g) x = nd(rng);
                    for (size_t i = 0; i < cols.size(); ++i) for (size_t j = 0; j < cols.size(); ++j) A[(size_t)cols[i] * n_pad + cols[j]] += g[i] * g[j];
                }
            }
        for (int i : free_cols) { A[(size_t)i * n_pad + i] += 1.0; b[i] = nd(rng); }
    }
    // ---- reference: dense Cholesky on the free columns ---------------------------------------------------------------------------
    const int nf = (int)free_cols.size();
    std::vector<double> x_ref(n_pad, 0.0);
    {
        std::vector<double> M((size_t)nf * nf), y(nf);
        for (int i = 0; i < nf; ++i) for (int j = 0; j < nf; ++j) M[(size_t)i * nf + j] = A[(size_t)free_cols[i] * n_pad + free_cols[j]];
        for (int j = 0; j < nf; ++j) {
            for (int k = 0; k < j; ++k) M[(size_t)j * nf + j] -= M[(size_t)j * nf + k] * M[(size_t)j * nf + k];
            M[(size_t)j * nf + j] = std::sqrt(M[(size_t)j * nf + j]);
            for (int i = j + 1; i < nf; ++i) {
                for (int k = 0; k < j; ++k) M[(size_t)i * nf + j] -= M[(size_t)i * nf + k] * M[(size_t)j * nf + k];
                M[(size_t)i * nf + j] /= M[(size_t)j * nf + j];
            }
        }
        for (int i = 0; i < nf; ++i) { double v = b[free_cols[i]]; for (int k = 0; k < i; ++k) v -= M[(size_t)i * nf + k] * y[k]; y[i] = v / M[(size_t)i * nf + i]; }
        for (int i = nf - 1; i >= 0; --i) { double v = y[i]; for (int k = i + 1; k < nf; ++k) v -= M[(size_t)k * nf + i] * y[k]; y[i] = v / M[(size_t)i * nf + i]; x_ref[free_cols[i]] = y[i]; }
    }
    // ---- the kernel's algorithm, tile by tile ---------------------------------------------------------------------------------------
    const int NP = pl.NP, nt = (int)pl.tiles.size();
    std::vector<Tile4> a(nt);
    for (int t = 0; t < nt; ++t) {
        const NdTile &tl = pl.tiles[t];
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) {
                double v = 0.0;
                const int j = pl.pcol[4 * tl.cj + c];
                if (tl.kind == 1) {
                    const int i = pl.pcol[4 * tl.ri + r];
                    if (i >= 0 && j >= 0) v = A[(size_t)i * n_pad + j];
                    if (tl.ri == tl.cj && r == c && i < 0) v = 1.0;          // identity padding
                } else if (r == 0 && j >= 0) v = b[j];
                a[t].v[r][c] = v;
            }
    }
    // structural check: every entry of A outside the tile set must be zero
    {
        std::map<std::pair<int, int>, int> has;
        for (const auto &tl : pl.tiles) if (tl.kind == 1) has[{ tl.ri, tl.cj }] = 1;
        for (int p = 0; p < NP; ++p) for (int k = 0; k <= p; ++k) if (!has.count({ p, k }))
            for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) {
                const int i = pl.pcol[4 * p + r], j = pl.pcol[4 * k + c];
                if (i >= 0 && j >= 0 && A[(size_t)i * n_pad + j] != 0.0) { std::printf("{\"ok\": false, \"reason\": \"nonzero outside the tile set\"}\n"); return 0; }
            }
    }
    const int SL = pl.max_slots;
    std::vector<Tile4> Xs((size_t)SL * (NP + 1));                           // solved panel columns of the current phase: [slot][row panel]
    auto xs = [&](int slot, int row) -> Tile4 & { return Xs[(size_t)slot * (NP + 1) + row]; };
    Tile4 s_dt[2][4], s_dr[2][4][4];
    std::vector<double> Ld((size_t)NP * 16), il((size_t)NP * 4), wp((size_t)4 * NP, 0.0);
    bool fail = false;
    struct Meta { int ri, cj, kind, phase_c, slot_c, dr, lt; unsigned um; };
    auto meta = [&](int t) {
        const NdTile &tl = pl.tiles[t];
        Meta m;
        m.ri = tl.ri; m.cj = tl.cj; m.kind = tl.kind; m.lt = tl.lt;
        m.um = tl.kind == 1 ? (pl.lmask[tl.ri] & pl.lmask[tl.cj]) : pl.lmask[tl.cj];
        m.phase_c = pl.phase_of[m.cj]; m.slot_c = pl.slot_of[m.cj];
        m.dr = tl.ri != tl.cj ? pl.dr_slot(tl.ri, tl.cj) : 0xff;
        return m;
    };
    for (int t = 0; t < nt; ++t) {
        const Meta m = meta(t);
        if (m.ri == m.cj && m.phase_c <= 1) s_dt[m.phase_c][m.slot_c] = a[t];
        if (m.dr != 0xff && m.phase_c == 0) s_dr[0][m.dr][m.slot_c] = a[t];
    }
    auto LdM = [&](int k) { return reinterpret_cast<double (*)[4]>(&Ld[(size_t)16 * k]); };
    for (int q = 0; q < 4; ++q) { const int k = (pl.phase_panels[0] >> (8 * q)) & 0xff; if (k == 0xff) continue; Tile4 d = s_dt[0][q]; if (!factor_tile(d, LdM(k), &il[4 * k])) fail = true; }
    for (int ph = 0; ph < pl.n_phases; ++ph) {
        const unsigned sp = pl.phase_panels[ph], spn = ph + 1 < pl.n_phases ? pl.phase_panels[ph + 1] : 0xffffffffu;
        // look-ahead lanes (read the state of the barrier before this phase)
        std::vector<std::pair<int, Tile4>> newfac;
        for (int ql = 0; ql < 4; ++ql) {
            const int kp = (spn >> (8 * ql)) & 0xff;
            if (kp == 0xff) continue;
            Tile4 dt = s_dt[(ph + 1) & 1][ql];
            for (int q = 0; q < 4; ++q) {
                const int k = (sp >> (8 * q)) & 0xff;
                if (k == 0xff || !((pl.lmask[kp] >> k) & 1u)) continue;
                Tile4 x; solve_tile(s_dr[ph & 1][ql][q], LdM(k), &il[4 * k], x);
                for (int r = 0; r < 4; ++r) for (int c = 0; c <= r; ++c) for (int e = 0; e < 4; ++e) dt.v[r][c] -= x.v[r][e] * x.v[c][e];
            }
            newfac.push_back({ kp, dt });
        }
        // step A: the tiles of the panel columns are solved and published
        for (int t = 0; t < nt; ++t) {
            const Meta m = meta(t);
            if (m.phase_c != ph || m.ri == m.cj) continue;
            Tile4 x; solve_tile(a[t], LdM(m.cj), &il[4 * m.cj], x);
            a[t] = x; xs(m.slot_c, m.ri) = x;
            if (m.kind == 2) for (int c = 0; c < 4; ++c) wp[4 * m.cj + c] = x.v[0][c];
        }
        // barrier A; step B: trailing updates, then what the next phases need
        std::vector<std::pair<Tile4 *, Tile4>> pub;
        for (int t = 0; t < nt; ++t) {
            const Meta m = meta(t);
            const bool dskip = m.ri == m.cj && m.phase_c == ph + 1;
            for (int q = 0; q < 4; ++q) {
                const int k = (sp >> (8 * q)) & 0xff;
                if (k == 0xff || !(k < m.cj && ((m.um >> k) & 1u)) || dskip) continue;
                const Tile4 &xi = xs(q, m.ri), &xj = xs(q, m.cj);
                for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) for (int e = 0; e < 4; ++e) a[t].v[r][c] -= xi.v[r][e] * xj.v[c][e];
            }
            if (m.ri == m.cj && m.phase_c == ph + 2) pub.push_back({ &s_dt[(ph + 2) & 1][m.slot_c], a[t] });
            if (m.dr != 0xff && m.phase_c == ph + 1) pub.push_back({ &s_dr[(ph + 1) & 1][m.dr][m.slot_c], a[t] });
        }
        // barrier B
        for (auto &p : pub) *p.first = p.second;
        for (auto &nf2 : newfac) { Tile4 d = nf2.second; if (!factor_tile(d, LdM(nf2.first), &il[4 * nf2.first])) fail = true; }
    }
    // packed factor, row-major by (row panel, column panel), tiles transposed: Lt[lt][c][r]
    std::vector<double> Lt((size_t)16 * std::max(pl.n_lt, 1), 0.0);
    for (int t = 0; t < nt; ++t) if (pl.tiles[t].lt >= 0)
        for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) Lt[(size_t)16 * pl.tiles[t].lt + 4 * c + r] = a[t].v[r][c];
    // back-substitution L^T y = w, phases in reverse, left-looking: y_k = L_kk^-T (w_k - sum_{i > k} L_ik^T y_i), the tiles of a
    // panel's column gathered by the 16 lanes of its slot (NdPlan::bs_tab) and summed over the lanes
    std::vector<double> w(wp);
    for (int ph = pl.n_phases - 1; ph >= 0; --ph) {
        const unsigned sp = pl.phase_panels[ph];
        double acc[64][4];
        for (int l = 0; l < 64; ++l) for (int c = 0; c < 4; ++c) acc[l][c] = 0.0;
        for (int rd = 0; rd < pl.bs_rounds; ++rd)
            for (int l = 0; l < 64; ++l) {
                const int e = pl.bs_tab[((size_t)ph * pl.bs_rounds + rd) * 64 + l];
                if (e < 0) continue;
                const int lt = e & 0xffff, i = e >> 16;
                for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) acc[l][c] += Lt[(size_t)16 * lt + 4 * c + r] * w[4 * i + r];
            }
        for (int q = 0; q < 4; ++q) {
            const int k = (sp >> (8 * q)) & 0xff;
            if (k == 0xff) continue;
            double v[4];
            for (int c = 0; c < 4; ++c) { double sum = 0.0; for (int j = 0; j < kNdBsGroup; ++j) sum += acc[kNdBsGroup * q + j][c]; v[c] = w[4 * k + c] - sum; }
            for (int c = 3; c >= 0; --c) {
                double x = v[c];
                for (int e = c + 1; e < 4; ++e) x -= LdM(k)[e][c] * v[e];
                v[c] = x * il[4 * k + c];
            }
            for (int c = 0; c < 4; ++c) w[4 * k + c] = v[c];
        }
    }
    double err = 0.0, ref = 0.0;
    for (int i = 0; i < 4 * NP; ++i) { const int pc = pl.pcol[i]; if (pc >= 0) { err = std::max(err, std::fabs(w[i] - x_ref[pc])); ref = std::max(ref, std::fabs(x_ref[pc])); } }
    int covered = 0;
    for (int i = 0; i < 4 * NP; ++i) if (pl.pcol[i] >= 0) ++covered;
    std::printf("{\"ok\": %s, \"NP\": %d, \"phases\": %d, \"tiles\": %d, \"tpt\": %d, \"max_slots\": %d, \"n_lt\": %d, \"dense\": %s, \"free_cols\": %d, \"covered\": %d, \"rel_err\": %.3e, \"lds_bytes\": %zu, \"fell_back\": %s, \"levels\": \"",
                (!fail && covered == nf) ? "true" : "false", NP, pl.n_phases, nt, pl.tpt, pl.max_slots, pl.n_lt, pl.dense ? "true" : "false", nf, covered, ref > 0 ? err / ref : err, pl.lds_doubles * 8, fell_back ? "true" : "false");
    for (size_t l = 0; l < pl.levels.size(); ++l) { std::printf("%s", l ? "|" : ""); for (size_t i = 0; i < pl.levels[l].size(); ++i) std::printf("%s%d", i ? "," : "", pl.levels[l][i]); }
    std::printf("\"}\n");
    return 0;
}

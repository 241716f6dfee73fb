// tscm_nd_plan.h -- host side of the reduced camera system's solver (k_solve_nd): elimination order, phase schedule,
// tile list and per-thread operand map, computed once per solver from the camera-pair graph.
//
// The reduced system S_c (H_cc - T) S_c + D^2 (what Ceres' DENSE_SCHUR hands to its dense Cholesky, multi_calib.cpp:209-216)
// has a 13 x 13 block per camera (7 x 7 for a camera whose pose is constant, multi_calib.cpp:186) and an off-diagonal
// block for every camera PAIR that shares a board.  The reference requires adjacent cameras to share boards
// (multi_calib.cpp:31-36) and nothing else: in the rigs it is used on the pair graph is a ring or a chain, and the
// matrix is block-cyclic-tridiagonal, not dense.  Instead of one dense Cholesky over all 13 C - 6 columns the plan
//   * orders the cameras by nested dissection of the pair graph: a LEVEL is a set of cameras no two of which are
//     adjacent (in the graph with the fill of the levels before it), so their blocks are eliminated CONCURRENTLY;
//     what remains once the rest is a clique is one dense block ("final level"), packed without padding;
//   * cuts every camera block of a level into panels of 4 columns (padded with identity columns) and runs the panels
//     with the same position of all cameras of a level in ONE phase of the blocked right-looking factorisation;
//   * keeps only the 4 x 4 tiles of the lower factor that are structurally non-zero.
// Ring of 8 cameras (BASELINE config 5): levels {1,3,5,7}, {2,6}, then the dense block {0,4}: 4 + 4 + 5 = 13 phases
// instead of 25 and 227 matrix tiles instead of 325; ring of 4 (config 4): {1,3}, then {0,2}: 9 phases instead of 12.  A complete
// graph gives the single dense block -- the algorithm of the rounds before.
#pragma once

#include <algorithm>
#include <cstdint>
#include <vector>

namespace tscm {

constexpr int kNdThreads = 256;        // workgroup of k_solve_nd: three waves of tile threads, one wave for the look-ahead lanes
constexpr int kNdTileThreads = 192;
constexpr int kNdSlots = 4;            // panels per phase at most (cameras per level)
constexpr int kNdMaxPanels = 32;       // 8 cameras x 4 panels
constexpr int kNdMaxPhases = 32;       // one panel per phase at least
constexpr int kNdUnitInts = 48;        // per thread and tile: 16 offsets into H, 16 into T, 4 + 4 scaling indices, 4 meta words, 4 words of update masks
                                       // (phases 8 w .. 8 w + 7 in word w, four bits each: the slots of that phase whose panel updates the tile)
constexpr int kNdMapOne = 1 << 30;     // "scaling 1": the row of a right-hand side tile
constexpr int kNdXT = 18;              // doubles per published tile (16 + 2: 16 lanes read 16 tiles on distinct bank pairs)
// table block (ints) read by every thread at the head of the kernel
constexpr int kNdTabPhasePanels = 0,                                   // [kNdMaxPhases] four panel numbers, 8 bits each (slot = camera of the level), 0xff: none
              kNdTabLmask = kNdTabPhasePanels + kNdMaxPhases,          // [33] row structure: bit k of entry p: tile (p, k), k < p, exists; entry NP: the rhs row
              kNdTabRowStart = kNdTabLmask + kNdMaxPanels + 1,         // [33] first index of row p's tiles in the packed factor (row-major, columns ascending)
              kNdTabPcol = kNdTabRowStart + kNdMaxPanels + 1,          // [128] padded column of each column in elimination order, -1: identity padding
              kNdTabDmask = kNdTabPcol + 4 * kNdMaxPanels,             // [4][4] look-ahead lane q, word w: for the phases 8 w .. 8 w + 7, four bits each: the slots of
                                                                       // that phase whose panel updates the diagonal tile lane q factors for the NEXT phase
              kNdTabInts = (kNdTabDmask + kNdSlots * 4 + 3) & ~3;
// what the kernel needs of the plan before its first memory access: kernel arguments
struct NdDims { int NP, n_phases, slots, n_lt, bs_rounds; };
constexpr int kNdBsGroup = 16;         // lanes per panel in the back-substitution (one DPP row): lane 16 q + j takes the j-th tile of slot q's column

struct NdTile { int ri, cj, kind, lt; };       // kind 1: matrix tile (ri >= cj), 2: right-hand side tile (ri = NP); lt: index in the packed factor (-1: none)

struct NdPlan {
    int NP = 0, n_phases = 0, max_slots = 1, n_lt = 0, tpt = 1;
    bool dense = true;
    std::vector<std::vector<int>> levels;      // cameras of each concurrent level; the last entry is the dense block
    std::vector<int> pcol;                     // [4 NP]
    std::vector<unsigned> lmask;               // [NP + 1]
    std::vector<int> rowstart;                 // [NP + 1]
    std::vector<int> phase_of, slot_of;        // [NP]
    std::vector<unsigned> phase_panels;        // [n_phases]
    std::vector<NdTile> tiles;                 // in thread order: tile i belongs to thread i % kNdTileThreads, unit i / kNdTileThreads
    std::vector<int> tab;                      // [kNdTabInts]
    int bs_rounds = 1;                         // back-substitution: tiles per panel column / kNdBsGroup, rounded up
    std::vector<int> bs_tab;                   // [n_phases][bs_rounds][64] lt | row panel << 16 of the tile lane l gathers in that phase, -1: none
    std::vector<int> map;                      // [tpt][kNdUnitInts / 4][kNdThreads] int4
    size_t lds_doubles = 0;
    NdDims dims() const { return NdDims{ NP, n_phases, max_slots, n_lt, bs_rounds }; }
    int lt_of(int i, int k) const { return rowstart[i] + __builtin_popcount(lmask[i] & ((1u << k) - 1u)); }
    // slot of row panel i's look-ahead lane if tile (i, k) is the one that lane needs raw (i is a panel of the phase after k's), else 0xff
    int dr_slot(int i, int k) const { return i < NP && phase_of[i] == phase_of[k] + 1 ? slot_of[i] : 0xff; }
};

// C cameras; ncols[m]: free columns of camera m (0: no views, 7: constant pose, 13), col0[m]: its first free padded column;
// pair[mi * C + mj] (mi <= mj): the pair shares a board; bid_of[mi * C + mj]: its tile of T (-1: none).
// dense_only: one dense block (the ordering of the rounds before; TSCM_EXEC_DENSE_REDUCED_ORDER).
// Returns false when the system does not fit the kernel (more than kNdMaxPanels panels or 2 x kNdTileThreads tiles).
inline bool nd_build_plan(int C, const int *ncols, const int *col0, const unsigned char *pair, const int *bid_of, bool dense_only, NdPlan &pl)
{
    pl = NdPlan();
    std::vector<int> act;
    for (int m = 0; m < C; ++m) if (ncols[m] > 0) act.push_back(m);
    std::vector<unsigned> adj(C, 0u);
    for (int mi = 0; mi < C; ++mi)
        for (int mj = mi + 1; mj < C; ++mj)
            if (pair[mi * C + mj] && ncols[mi] > 0 && ncols[mj] > 0) { adj[mi] |= 1u << mj; adj[mj] |= 1u << mi; }
    // ---- levels: greedy independent sets by (degree, columns descending, index), with fill --------------------------------
    unsigned rem = 0;
    for (int m : act) rem |= 1u << m;
    std::vector<unsigned> nb_at_elim(C, 0u);     // neighbours of a camera among those eliminated AFTER it (structure of its column block)
    auto popc = [](unsigned x) { return __builtin_popcount(x); };
    auto is_clique = [&](unsigned set) {
        for (int m = 0; m < C; ++m) if ((set >> m) & 1u) if (((adj[m] | (1u << m)) & set) != set) return false;
        return true;
    };
    std::vector<std::vector<int>> levels;
    if (!dense_only) {
        while (rem && !is_clique(rem)) {
            std::vector<int> cand;
            for (int m = 0; m < C; ++m) if ((rem >> m) & 1u) cand.push_back(m);
            std::stable_sort(cand.begin(), cand.end(), [&](int x, int y) {
                const int dx = popc(adj[x] & rem), dy = popc(adj[y] & rem);
                if (dx != dy) return dx < dy;
                if (ncols[x] != ncols[y]) return ncols[x] > ncols[y];
                return x < y;
            });
            std::vector<int> lvl;
            unsigned blocked = 0;
            for (int m : cand) {
                if ((int)lvl.size() == kNdSlots || ((blocked >> m) & 1u)) continue;
                lvl.push_back(m);
                blocked |= adj[m] | (1u << m);
            }
            unsigned after = rem;
            for (int m : lvl) after &= ~(1u << m);
            for (int m : lvl) {
                const unsigned nb = adj[m] & after;
                nb_at_elim[m] = nb;
                for (int i = 0; i < C; ++i) if ((nb >> i) & 1u) adj[i] |= nb & ~(1u << i);       // fill among the neighbours
            }
            rem = after;
            std::sort(lvl.begin(), lvl.end());
            levels.push_back(lvl);
        }
    }
    std::vector<int> fin;
    for (int m = 0; m < C; ++m) if ((rem >> m) & 1u) fin.push_back(m);
    pl.dense = levels.empty();
    // ---- panels and phases (slot = position of the camera in its level: stable over the level's phases) --------------------
    std::vector<int> panel_cam;                // camera of a panel of a concurrent level
    for (size_t l = 0; l < levels.size(); ++l) {
        const auto &lvl = levels[l];
        int depth = 0;
        for (int m : lvl) depth = std::max(depth, (ncols[m] + 3) / 4);
        std::vector<int> first(lvl.size());
        for (size_t q = 0; q < lvl.size(); ++q) {
            const int m = lvl[q], np = (ncols[m] + 3) / 4;
            first[q] = pl.NP;
            for (int j = 0; j < np; ++j) {
                panel_cam.push_back(m);
                for (int c = 0; c < 4; ++c) pl.pcol.push_back(4 * j + c < ncols[m] ? col0[m] + 4 * j + c : -1);
                ++pl.NP;
            }
        }
        pl.phase_of.resize(pl.NP); pl.slot_of.resize(pl.NP);
        pl.max_slots = std::max(pl.max_slots, (int)lvl.size());
        for (int j = 0; j < depth; ++j) {
            unsigned packed = 0xffffffffu;
            for (size_t q = 0; q < lvl.size(); ++q) {
                const int m = lvl[q];
                if (j >= (ncols[m] + 3) / 4) continue;
                const int p = first[q] + j;
                pl.phase_of[p] = pl.n_phases; pl.slot_of[p] = (int)q;
                packed = (packed & ~(0xffu << (8 * q))) | ((unsigned)p << (8 * q));
            }
            pl.phase_panels.push_back(packed);
            ++pl.n_phases;
        }
    }
    const int fin_p0 = pl.NP;
    {
        std::vector<int> cols;
        for (int m : fin) for (int j = 0; j < ncols[m]; ++j) cols.push_back(col0[m] + j);
        const int np = ((int)cols.size() + 3) / 4;
        for (int j = 0; j < np; ++j) {
            for (int c = 0; c < 4; ++c) { const int i = 4 * j + c; pl.pcol.push_back(i < (int)cols.size() ? cols[i] : -1); }
            pl.phase_of.push_back(pl.n_phases); pl.slot_of.push_back(0);
            pl.phase_panels.push_back(0xffffff00u | (unsigned)pl.NP);
            ++pl.NP; ++pl.n_phases;
        }
    }
    if (pl.NP > kNdMaxPanels || pl.NP == 0 || pl.n_phases > kNdMaxPhases) return false;
    // ---- structure of the factor (panel level) ---------------------------------------------------------------------------------
    pl.lmask.assign(pl.NP + 1, 0u);
    auto cams_of_panel = [&](int p, std::vector<int> &out) {
        out.clear();
        if (p < fin_p0) { out.push_back(panel_cam[p]); return; }
        for (int c = 0; c < 4; ++c) { const int pc = pl.pcol[4 * p + c]; if (pc >= 0 && (out.empty() || out.back() != (pc >> 4))) out.push_back(pc >> 4); }
    };
    std::vector<int> cp, ck;
    for (int p = 0; p < pl.NP; ++p) {
        cams_of_panel(p, cp);
        for (int k = 0; k < p; ++k) {
            cams_of_panel(k, ck);
            bool nz = false;
            if (p >= fin_p0 && k >= fin_p0) nz = true;                               // the dense block
            else if (k < fin_p0) {
                const int mk = ck[0];
                for (int mp : cp) if (mp == mk || ((nb_at_elim[mk] >> mp) & 1u)) nz = true;   // same camera block, or a neighbour of camera mk when it is eliminated
            }
            if (nz) pl.lmask[p] |= 1u << k;
        }
    }
    pl.lmask[pl.NP] = pl.NP >= 32 ? 0xffffffffu : ((1u << pl.NP) - 1u);
    pl.rowstart.assign(pl.NP + 1, 0);
    for (int p = 0; p < pl.NP; ++p) pl.rowstart[p + 1] = pl.rowstart[p] + popc(pl.lmask[p]);
    pl.n_lt = pl.rowstart[pl.NP];
    // ---- tiles, in the order their column panel is eliminated (waves retire early).  Grouping them by the level and camera of their
    //      column instead -- so that the lanes of a wave take their updates from few slots -- was measured at config 5 and is no
    //      better (358.9-364.9 against 357.7 us per iteration) ---------------------------------------------------------------------
    std::vector<NdTile> tiles;
    for (int p = 0; p < pl.NP; ++p) {
        tiles.push_back({ p, p, 1, -1 });
        for (int k = 0; k < p; ++k) if ((pl.lmask[p] >> k) & 1u) tiles.push_back({ p, k, 1, pl.lt_of(p, k) });
        tiles.push_back({ pl.NP, p, 2, -1 });
    }
    std::stable_sort(tiles.begin(), tiles.end(), [&](const NdTile &x, const NdTile &y) {
        if (pl.phase_of[x.cj] != pl.phase_of[y.cj]) return pl.phase_of[x.cj] < pl.phase_of[y.cj];
        if (x.cj != y.cj) return x.cj < y.cj;
        return x.ri < y.ri;
    });
    if ((int)tiles.size() > 2 * kNdTileThreads) return false;
    pl.tpt = (int)tiles.size() > kNdTileThreads ? 2 : 1;
    pl.tiles = tiles;
    pl.levels = levels;
    pl.levels.push_back(fin);
    // ---- LDS of the solver workgroup (doubles): diagonal factors, rhs / solution / scaling vectors, the tiles handed to the
    //      look-ahead lanes (diagonal: [2][4][16], below-diagonal raw: [2][4][4][16]), tables, then the solved panel columns
    //      [slots][NP + 1][XT] -- the packed factor takes their place for the back-substitution -----------------------------------
    const size_t xs = (size_t)pl.max_slots * (pl.NP + 1) * kNdXT;
    // ---- back-substitution table: the tiles (i, k), i > k, of every panel column, spread over the 16 lanes of the panel's slot
    {
        int most = 1;
        for (int k = 0; k < pl.NP; ++k) { int cnt = 0; for (int i = k + 1; i < pl.NP; ++i) if ((pl.lmask[i] >> k) & 1u) ++cnt; most = std::max(most, cnt); }
        pl.bs_rounds = (most + kNdBsGroup - 1) / kNdBsGroup;
        pl.bs_tab.assign((size_t)pl.n_phases * pl.bs_rounds * 64, -1);
        for (int t = 0; t < pl.n_phases; ++t)
            for (int q = 0; q < kNdSlots; ++q) {
                const int k = (pl.phase_panels[t] >> (8 * q)) & 0xff;
                if (k == 0xff) continue;
                int j = 0;
                for (int i = k + 1; i < pl.NP; ++i) {
                    if (!((pl.lmask[i] >> k) & 1u)) continue;
                    pl.bs_tab[((size_t)t * pl.bs_rounds + j / kNdBsGroup) * 64 + kNdBsGroup * q + j % kNdBsGroup] = pl.lt_of(i, k) | (i << 16);
                    ++j;
                }
            }
    }
    const size_t bs_doubles = ((size_t)pl.n_phases * pl.bs_rounds * 64 + 1) / 2;
    pl.lds_doubles = 20 * (size_t)kNdMaxPanels + 4 * 128 + 2 * kNdSlots * 16 + 2 * kNdSlots * kNdSlots * 16 + kNdTabInts / 2 + std::max(xs, kNdXT * (size_t)pl.n_lt) + bs_doubles;
    // ---- tables ----------------------------------------------------------------------------------------------------------------
    pl.tab.assign(kNdTabInts, -1);
    for (int t = 0; t < kNdMaxPhases; ++t) pl.tab[kNdTabPhasePanels + t] = t < pl.n_phases ? (int)pl.phase_panels[t] : -1;
    for (int p = 0; p <= kNdMaxPanels; ++p) { pl.tab[kNdTabLmask + p] = p <= pl.NP ? (int)pl.lmask[p] : 0; pl.tab[kNdTabRowStart + p] = p <= pl.NP ? pl.rowstart[p] : 0; }
    for (int i = 0; i < 4 * kNdMaxPanels; ++i) pl.tab[kNdTabPcol + i] = i < 4 * pl.NP ? pl.pcol[i] : -1;
    for (int q = 0; q < kNdSlots * 4; ++q) pl.tab[kNdTabDmask + q] = 0;
    for (int t = 0; t + 1 < pl.n_phases; ++t)
        for (int ql = 0; ql < kNdSlots; ++ql) {
            const int kp = (pl.phase_panels[t + 1] >> (8 * ql)) & 0xff;
            if (kp == 0xff) continue;
            unsigned bits = 0;
            for (int q = 0; q < kNdSlots; ++q) { const int k = (pl.phase_panels[t] >> (8 * q)) & 0xff; if (k != 0xff && ((pl.lmask[kp] >> k) & 1u)) bits |= 1u << q; }
            pl.tab[kNdTabDmask + 4 * ql + t / 8] |= (int)(bits << (4 * (t % 8)));
        }
    // ---- operand map -----------------------------------------------------------------------------------------------------------
    auto t_offset = [&](int i, int j) -> int {          // T(i, j) for padded columns i, j; the lower blocks are the transposed upper ones
        int lo = i >> 4, hi = j >> 4, a = i & 15, b = j & 15;
        if (lo > hi) { std::swap(lo, hi); std::swap(a, b); }
        const int tile = bid_of[lo * C + hi];
        return tile >= 0 ? 256 * tile + a * 16 + b : -1;
    };
    constexpr int kFRcol = 13;                            // kFR (tscm_math.h): the gradient column of a camera tile
    pl.map.assign((size_t)pl.tpt * kNdUnitInts * kNdThreads, -1);
    for (int u = 0; u < pl.tpt; ++u)
        for (int t = 0; t < kNdThreads; ++t) {
            int off[kNdUnitInts];
            for (int q = 0; q < kNdUnitInts; ++q) off[q] = -1;
            off[40] = 0xff << 24; off[41] = 0; off[42] = 0xffff; off[43] = -1;
            off[44] = off[45] = off[46] = off[47] = 0;
            const size_t ti = (size_t)u * kNdTileThreads + t;
            if (t < kNdTileThreads && ti < tiles.size()) {
                const NdTile &tl = tiles[ti];
                int mi[4], mj[4];
                for (int r = 0; r < 4; ++r) { mi[r] = tl.kind == 1 ? pl.pcol[4 * tl.ri + r] : -1; mj[r] = pl.pcol[4 * tl.cj + r]; }
                for (int r = 0; r < 4; ++r) { off[32 + r] = mi[r]; off[36 + r] = mj[r]; }
                if (tl.kind == 1) {
                    for (int r = 0; r < 4; ++r)
                        for (int c = 0; c < 4; ++c) {
                            const int i = mi[r], j = mj[c];
                            if (i < 0 || j < 0) continue;
                            if ((i >> 4) == (j >> 4)) off[r * 4 + c] = 256 * (i >> 4) + (i & 15) * 16 + (j & 15);
                            off[16 + r * 4 + c] = t_offset(i, j);
                        }
                } else {
                    // right-hand side tile: row 0 = g - t_r of the panel's columns (column kFR of H and of the diagonal tile of T)
                    for (int c = 0; c < 4; ++c) {
                        const int j = mj[c];
                        if (j < 0) continue;
                        const int m = j >> 4, b = j & 15;
                        off[c] = 256 * m + b * 16 + kFRcol;
                        off[16 + c] = t_offset(j, m * 16 + kFRcol);
                    }
                    off[32] = kNdMapOne;
                }
                const unsigned um = tl.kind == 1 ? (pl.lmask[tl.ri] & pl.lmask[tl.cj]) : pl.lmask[tl.cj];
                // meta: row panel | column panel << 8 | kind << 16 | slot of the look-ahead lane that needs this tile raw << 24 (0xff: none);
                // the update mask; phase | slot << 8 of the column panel; index in the packed factor
                off[40] = tl.ri | (tl.cj << 8) | (tl.kind << 16) | ((tl.ri != tl.cj ? pl.dr_slot(tl.ri, tl.cj) : 0xff) << 24);
                off[41] = (int)um;
                off[42] = pl.phase_of[tl.cj] | (pl.slot_of[tl.cj] << 8);
                off[43] = tl.lt;
                // per phase: the slots whose panel k updates this tile (k < column panel, both factor tiles exist; the diagonal tile of the NEXT
                // phase's panel is brought up to date by its look-ahead lane instead)
                for (int t = 0; t < pl.phase_of[tl.cj]; ++t) {
                    if (tl.ri == tl.cj && pl.phase_of[tl.cj] == t + 1) continue;
                    unsigned bits = 0;
                    for (int q = 0; q < kNdSlots; ++q) { const int k = (pl.phase_panels[t] >> (8 * q)) & 0xff; if (k != 0xff && k < tl.cj && ((um >> k) & 1u)) bits |= 1u << q; }
                    off[44 + t / 8] |= (int)(bits << (4 * (t % 8)));
                }
            }
            for (int q = 0; q < kNdUnitInts / 4; ++q)
                for (int e = 0; e < 4; ++e) pl.map[(((size_t)u * (kNdUnitInts / 4) + q) * kNdThreads + t) * 4 + e] = off[4 * q + e];
        }
    return true;
}

// The two plans a solver keeps: [0] along the camera-pair graph, [1] the whole system as one dense block.  The graph plan pads
// every camera block to whole panels, so a dense but incomplete pair graph of 8 free cameras can exceed the tile budget that
// the dense packing of the same system meets: such a graph is solved on the dense plan (fell_back).  False only if the dense
// plan does not fit either.
inline bool nd_build_plans(int C, const int *ncols, const int *col0, const unsigned char *pair, const int *bid_of, NdPlan (&pl)[2], bool *fell_back = nullptr)
{
    if (fell_back) *fell_back = false;
    if (!nd_build_plan(C, ncols, col0, pair, bid_of, /*dense_only=*/true, pl[1])) return false;
    if (!nd_build_plan(C, ncols, col0, pair, bid_of, /*dense_only=*/false, pl[0])) { pl[0] = pl[1]; if (fell_back) *fell_back = true; }
    return true;
}

}  // namespace tscm

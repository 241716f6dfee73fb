// tscm_solve_dense4.h -- k_solve_reduced: the reduced camera system of rigs of up to FOUR cameras (n_pad <= 64) as one dense
// block: blocked right-looking Cholesky with the matrix in registers, one barrier per panel (rounds 1-3; included by
// tscm_kernels.h).  A ring of four cameras is nearly dense -- only the block between cameras 1 and 3 is empty -- and every
// tile of the last block receives an update from every panel before it whatever the order: measured on config 4, the
// schedule along the camera-pair graph (k_solve_nd, tscm_solve_nd.h: 9 phases, two panels in each of the first four) takes
// 11.5-11.8 us for the factorisation against 10.5 us for the 12 panels here, so rigs of up to four cameras stay with this
// kernel and k_solve_nd takes the rigs of five to eight, where the structure pays (13 phases instead of 25).
#pragma once
#include <type_traits>

// T(i, j) for padded columns i, j of the camera side; the lower blocks are the transposed upper ones
__device__ __forceinline__ double load_T_small(const DevProblem &P, const double *T, int i, int j)
{
    int lo = i >> 4, hi = j >> 4, a = i & 15, b = j & 15;
    if (lo > hi) { const int t = lo; lo = hi; hi = t; const int u = a; a = b; b = u; }
    const int bit = lo * 8 + hi;
    const unsigned long long m = P.pair_mask;
    const int tile = __popcll(m & ((1ull << bit) - 1ull));
    return ((m >> bit) & 1ull) ? T[256 * tile + a * 16 + b] : 0.0;
}

// compact index of the reduced system -> padded column (-1 past the last free column), from kernel arguments only
__device__ __forceinline__ int compact_to_padded(const DevProblem &P, int ci)
{
    int base = 0, c0 = P.cam_col0[0];
#pragma unroll
    for (int q = 1; q < kMaxCamLds; ++q) { const bool ge = ci >= P.cam_pre[q]; base = ge ? P.cam_pre[q] : base; c0 = ge ? P.cam_col0[q] : c0; }
    return ci < P.n_act ? c0 + (ci - base) : -1;
}
// Tile of thread tid in k_solve_reduced's G x G grid (NP panels of free columns).  Lower tile (ti, tj) of the matrix on
// thread ti * G + tj.  The right-hand side tiles (NP, p), p < NP, go to threads that own no matrix tile, counted
// downwards from the end of the last wave that holds matrix tiles: tile p is needed up to panel step p, so the
// longest-lived ones share a wave with the longest-lived matrix rows and the early waves retire early.  The
// look-ahead thread (the last thread of the workgroup) is never used.
struct SolveTile { bool mine, rhsrow; int ri, cj; };
template <int TS, int G>
__device__ __forceinline__ SolveTile solve_tile(int tid, int NP)
{
    constexpr int NT = (G * G + 63) / 64 * 64;
    auto owns = [&](int t) { const int ti = t / G, tj = t % G; return tj <= ti && ti < NP; };
    SolveTile t;
    t.mine = owns(tid); t.rhsrow = false;
    t.ri = tid / G; t.cj = tid % G;
    if (t.mine || tid == NT - 1) return t;
    const int last = min(NT - 2, ((NP - 1) * G + NP - 1) | 63);        // end of the wave of tile (NP-1, NP-1)
    if (tid > last) return t;
    int rank = 0;                                                       // free threads in (tid, last]
    for (int u = tid + 1; u <= last; ++u) rank += owns(u) ? 0 : 1;
    if (rank < NP) { t.rhsrow = true; t.ri = NP; t.cj = NP - 1 - rank; }
    return t;
}
// slots of the per-thread operand map (ints): offsets into H[cur] and T per tile element (-1: the element is 0),
// s_c indices of the tile's rows and columns (-1: padding / rhs row, where 1 is used through kMapOne)
constexpr int kMapH = 0, kMapT = 16, kMapSci = 32, kMapScj = 36, kMapTile = 40, kSolveMapSlots = 44;   // kMapTile: row, column, 1 = matrix tile / 2 = rhs tile
constexpr int kMapOne = 1 << 30;       // "scaling 1": the row of a right-hand side tile

// Operand map of k_solve_reduced<TS, G> (run once per solver: the map depends on the camera/pair structure only).
// grid 1 x NT
template <int TS, int G>
__global__ __launch_bounds__((G * G + 63) / 64 * 64) void k_solve_map(DevProblem P, int4 *map)
{
    static_assert(TS == 4, "the map holds 4 x 4 tiles");
    constexpr int NT = (G * G + 63) / 64 * 64;
    const int tid = threadIdx.x;
    const int NP = (P.n_act + TS - 1) / TS;
    const SolveTile tl = solve_tile<TS, G>(tid, NP);
    auto cmap = [&](int ci) -> int { return compact_to_padded(P, ci); };
    auto t_offset = [&](int i, int j) -> int {                  // as load_T_small
        int lo = i >> 4, hi = j >> 4, a = i & 15, b = j & 15;
        if (lo > hi) { const int t = lo; lo = hi; hi = t; const int u = a; a = b; b = u; }
        const int bit = lo * 8 + hi;
        const unsigned long long m = P.pair_mask;
        return ((m >> bit) & 1ull) ? 256 * __popcll(m & ((1ull << bit) - 1ull)) + a * 16 + b : -1;
    };
    int off[kSolveMapSlots];
    for (int q = 0; q < kSolveMapSlots; ++q) off[q] = -1;
    off[kMapTile] = tl.ri; off[kMapTile + 1] = tl.cj; off[kMapTile + 2] = tl.mine ? 1 : tl.rhsrow ? 2 : 0;
    if (tl.mine || tl.rhsrow) {
        int mi[TS], mj[TS];
        for (int r = 0; r < TS; ++r) { mi[r] = tl.mine ? cmap(tl.ri * TS + r) : -1; mj[r] = cmap(tl.cj * TS + r); }
        for (int r = 0; r < TS; ++r) { off[kMapSci + r] = mi[r]; off[kMapScj + r] = mj[r]; }
        if (tl.mine) {
            for (int r = 0; r < TS; ++r)
                for (int c = 0; c < TS; ++c) {
                    const int i = mi[r], j = mj[c];
                    if (i < 0 || j < 0) continue;
                    if ((i >> 4) == (j >> 4)) off[kMapH + r * TS + c] = 256 * (i >> 4) + (i & 15) * 16 + (j & 15);
                    off[kMapT + r * TS + c] = t_offset(i, j);
                }
        } else {
            // right-hand side tile: row 0 = g - t_r of the panel's columns (the fused column kFR of H and T)
            for (int c = 0; c < TS; ++c) {
                const int j = mj[c];
                if (j < 0) continue;
                const int m = j >> 4, b = j & 15;
                off[kMapH + c] = 256 * m + b * 16 + kFR;
                off[kMapT + c] = t_offset(j, m * 16 + kFR);
            }
            off[kMapSci] = kMapOne;
        }
    }
    for (int q = 0; q < kSolveMapSlots / 4; ++q) map[q * NT + tid] = make_int4(off[4 * q], off[4 * q + 1], off[4 * q + 2], off[4 * q + 3]);
}

// ---------------------------------------------------------------------------------------------
// Reduced camera system (DenseSchurComplementSolver): one 256-thread workgroup.
//   A = S_c (H_cc - T) S_c + D_c^2, rhs = S_c (g_c - t_r); inactive columns (tile padding,
//   constant camera pose, cameras without views) become identity rows.
// Blocked right-looking Cholesky with the matrix held in REGISTERS: thread (ti, tj) owns the
// TS x TS tile (TS = N/16).  Per panel: the diagonal thread factors and inverts its tile and forward-
// substitutes its slice of the rhs while the column threads publish their raw tiles; after ONE
// barrier every trailing thread forms the needed X = A L_kk^{-T} tiles itself and applies the rank-TS
// update -- 16 barriers in total.
// Back-substitution: one wave, w in registers, rows of L streamed from LDS.  Writes
// yhat = S_c y (camera step = -yhat) and the candidate camera parameters.
// grid 1 x 256, dynamic LDS N*(N+2) + 2*G*(TS*TS+2) + 2*N + 3*NPD doubles.
// ---------------------------------------------------------------------------------------------
// G x G threads, thread (ti, tj) owns the TS x TS tile (ti, tj) of the COMPACT system (N = G * TS >= n_act columns);
// NPD >= n_pad is the capacity of the arrays indexed by padded column
// FUSED (256-thread variant, one GPU): the launch carries the T reduction as workgroups 1 .. n_bids * 256 / kFusedEntries;
// workgroup 0 is the solver and waits for their tiles behind an arrival counter (release -> counter -> acquire) after it
// has requested everything else.  Unlike the producers of the earlier hand-off experiments these have written 20 KB, not
// megabytes, when they release -- and a launch with its 5 us is gone.
// n_prod: workgroups 1 .. n_prod are the T reduction (FUSED); n_bs > 0: workgroups behind them are the back-substitution
// of this step (backsub_body<256, true>): their loads are in flight and their registers full while the solver works
// MF (round 6): the factorisation on ONE wave with the matrix in the accumulator layout of v_mfma_f64_16x16x4 -- see "MF" below
template <int TS, int G = 16, int NPD = 64, bool FUSED = false, bool MF = false>
__global__ __launch_bounds__((G * G + 63) / 64 * 64, FUSED ? 3 : 1) void k_solve_reduced(DevProblem P, DevState S, int epoch, int withhold, int n_prod, int n_bs, int with_floats)
{
    constexpr int NT = (G * G + 63) / 64 * 64;      // whole waves; threads past G * G own no tile
    if constexpr (FUSED) {
        static_assert(NT == kFusedEntries * kTSlices, "the T reduction runs in the solver's workgroup shape");
        if ((int)blockIdx.x > n_prod) {
            TL_ONLY(
            KtlScope ktl_bs(5, S.ctrl);
            ktl_bs.blk = (int)blockIdx.x - 1 - n_prod;
            )
            backsub_body<256, true>(P, S, with_floats, (int)blockIdx.x - 1 - n_prod, n_bs, epoch, epoch * n_prod);
            return;
        }
    }
    KTL(4);
    if constexpr (FUSED) {
        if (blockIdx.x > 0) {
            constexpr int kParts = 256 / kFusedEntries;
            const int bid = ((int)blockIdx.x - 1) / kParts, part = ((int)blockIdx.x - 1) % kParts;
            const int cb = P.bid_part_small[bid], ce = P.bid_part_small[bid + 1];      // kernel arguments: the partial tiles are the first thing requested
            if (S.ctrl->done) return;
            __shared__ double red[kTSlices][kFusedEntries];
            t_reduce_block<kFusedEntries>(S, bid, part, cb, ce, red);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0 && !(withhold && blockIdx.x == 1))        // (withhold: fault injection, TSCM_EXEC_TEST_WITHHOLD_HANDOFF)
                __hip_atomic_fetch_add(S.t_count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // (no release fence: handoff_store)
            return;
        }
    }
    // control block and the static column tables are requested together (one memory round trip); the early
    // exit is taken once they are there
    PHASE_STAMP(ts0);
    const int ctrl_done = S.ctrl->done;
    constexpr int N = G * TS;
    constexpr int LD = N + 2;                       // even: the rows of a diagonal tile are read as 16-byte pairs
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *Lm = lds;                 // [N][LD] lower factor: the diagonal tiles as they are factored, the rest at the end
    constexpr int XT = TS * TS + 2;   // tile stride of the panel column: 16 lanes tj read 16 tiles at once -- a stride of 16 doubles would put them on two bank pairs
    double *Xb = Lm + N * LD;         // [2][G][XT] raw panel column, one TS x TS tile per row block (double-buffered)
    double *wp = Xb + 2 * G * XT;     // [N] forward-substituted rhs  w = L^{-1} b
    double *idg = wp + N;             // [N] 1 / L_kk
    double *yv = idg + N;             // [NPD] solution by padded column
    double *s_sc = yv + NPD;          // [NPD]
    double *s_yh = s_sc + NPD;        // [NPD]
    __shared__ int s_fail;
    __shared__ unsigned char s_act[NPD];
    __shared__ double sred[256];
    const int n = P.n_pad;            // <= N
    const int tid = threadIdx.x;
    const int NP = (P.n_act + TS - 1) / TS;       // panels that hold free columns
    // ---- operands of my tile (lower tiles only) -------------------------------------------------------
    // The right-hand side rides along as tile row NP: row 0 of tile (NP, p) is the rhs slice of panel p (rows
    // 1..TS-1 are zero), so the forward substitution w = L^{-1} b falls out of the panel solves and trailing
    // updates and no thread treats it specially.  Those tiles live on idle threads of the last wave that holds
    // matrix tiles (k_solve_map): the fewer waves take part in a panel step, the less they queue at the LDS.
    // Where a thread's operands sit in H, T and s_c depends on the problem's structure only: k_solve_map wrote
    // the offsets once, so the head of this kernel is two memory round trips (offsets + control block, then the
    // operands) and next to no index arithmetic.
    int off[kSolveMapSlots];
#pragma unroll
    for (int q = 0; q < kSolveMapSlots / 4; ++q) {
        const int4 v = P.solve_map[q * NT + tid];
        off[4 * q] = v.x; off[4 * q + 1] = v.y; off[4 * q + 2] = v.z; off[4 * q + 3] = v.w;
    }
    const int ri = off[kMapTile], cj = off[kMapTile + 1];                // tile (row, column) of this thread
    const bool mine = off[kMapTile + 2] == 1, rhsrow = off[kMapTile + 2] == 2;
    const int cur = S.ctrl->cur;
    const double radius = S.ctrl->radius;
    const double dmin = S.ctrl->opt.min_lm_diagonal, dmax = S.ctrl->opt.max_lm_diagonal;
    const int ctrl_fail = S.ctrl->lin_fail | *S.fac_fail;       // (fac_fail is cleared in the tail, by the one workgroup that solves)
    for (int i = tid; i < NPD; i += NT) { s_sc[i] = i < n ? S.s_c[i] : 1.0; s_act[i] = i < n ? P.col_active[i] : 0; yv[i] = 0.0; }
    if (ctrl_done) return;
    const double *H = S.H[cur];
    double sci[TS], scj[TS], hh[TS][TS], tt[TS][TS];
#pragma unroll
    for (int r = 0; r < TS; ++r) {
        const int oi = off[kMapSci + r], oj = off[kMapScj + r];
        sci[r] = oi == kMapOne ? 1.0 : oi >= 0 ? S.s_c[oi] : 0.0;
        scj[r] = oj >= 0 ? S.s_c[oj] : 0.0;
    }
#pragma unroll
    for (int r = 0; r < TS; ++r)
#pragma unroll
        for (int c = 0; c < TS; ++c) { const int oh = off[kMapH + r * TS + c]; hh[r][c] = oh >= 0 ? H[oh] : 0.0; }
    if constexpr (FUSED) {
        // everything that does not depend on T is in flight; now the tiles of the other workgroups
        __shared__ int s_late;
        if (tid == 0) {
            const int need = epoch * n_prod;
            const long long t_start = wall_clock64();
            int late = 0;
            while (__hip_atomic_load(S.t_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
                __builtin_amdgcn_s_sleep(2);
                if (wall_clock64() - t_start > kHandoffTimeoutTicks) { late = 1; break; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if (late) {
                // not a failed linear solve (that would merely shrink the trust region and go on): the stream's work stops here
                S.ctrl->fault = 1; S.ctrl->term_type = 2; S.ctrl->done = 1;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                // (the workgroups waiting for the camera step are let go: they see ctrl->done)
                if (n_bs > 0) __hip_atomic_store(S.y_flag, 2 * epoch + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            s_late = late;
        }
        __syncthreads();
        if (s_late) return;
    }
#pragma unroll
    for (int r = 0; r < TS; ++r)
#pragma unroll
        for (int c = 0; c < TS; ++c) { const int ot = off[kMapT + r * TS + c]; tt[r][c] = ot >= 0 ? (FUSED ? handoff_load(&S.T[ot]) : S.T[ot]) : 0.0; }
    PHASE_STAMP(ts0b);
    if (tid == 0) s_fail = ctrl_fail;
    double a[TS][TS];
    const double inv_radius = 1.0 / radius;
#pragma unroll
    for (int r = 0; r < TS; ++r) {
#pragma unroll
        for (int c = 0; c < TS; ++c) {
            // matrix tiles: S_c (H - T) S_c, damped diagonal; identity on the padding columns.  rhs tiles: row 0 of
            // the map holds (g, t_r) of the panel's columns, s_c of the ROW is stored as 1 there.
            const bool dg = mine && ri == cj && r == c;
            double v = sci[r] * scj[c] * (hh[r][c] - tt[r][c]);
            if (dg) v = (off[kMapSci + r] >= 0 && off[kMapSci + r] != kMapOne) ? v + fmin(fmax(sci[r] * sci[r] * hh[r][c], dmin), dmax) * inv_radius : 1.0;
            a[r][c] = v;
        }
    }
    PHASE_STAMP(ts1);
    PH_ONLY(const long long cy1 = clock64();)
    if constexpr (MF) {
    // ---- MF: right-looking Cholesky of the 48 x 48 system on ONE wave, rank-4 updates on the matrix cores (round 6) ----------
    // The compact system of a rig of up to four cameras has at most 46 free columns: padded to 48 it is SIX lower 16 x 16 tiles --
    // 48 accumulator registers of one wave in the D layout of v_mfma_f64_16x16x4_f64 (lane (col, kq), register r: row kq + 4 r).
    // A panel is 4 columns = exactly the instruction's K, so the trailing update of a panel is one MFMA per tile that is still
    // alive (6 | 3 | 1 by tile column: 40 for the 12 panels) instead of 64 FMAs per thread and panel behind a workgroup barrier.
    // Per panel: the lanes that hold the panel's columns put them into LDS (12 stores); every lane takes ITS ROW of the panel
    // and the 4 x 4 diagonal block (broadcast reads), factors the block itself -- the same 30 operations on every lane, no
    // hand-off -- and solves its row (x = a L_kk^-T); the rows go back to LDS in row-major order, which is both the factor the
    // back-substitution reads and, fetched as lane (row, k), the A / B operand of the update.  The right-hand side rides along as
    // row 47 (w = L^-1 b falls out of the updates, as with the 4 x 4 tiles of rounds 1-5); columns past the last free one are
    // identity.  No workgroup barrier inside the factorisation; the other three waves wait at the one behind it.
        static_assert(TS == 4 && G == 16 && N >= 48, "six 16 x 16 tiles");
        constexpr int N48 = 48;
        for (int i = tid; i < N48 * LD; i += NT) Lm[i] = 0.0;
        __syncthreads();
        if (mine) {
#pragma unroll
            for (int r = 0; r < TS; ++r)
#pragma unroll
                for (int c = 0; c < TS; ++c)
                    // (free rows / columns only: the padding behind them is identity by construction below, and row 47 belongs to the
                    // right-hand side -- the last tile row's padding rows must not race with it)
                    if ((ri != cj || c <= r) && TS * ri + r < P.n_act && TS * cj + c < P.n_act) { Lm[(TS * ri + r) * LD + TS * cj + c] = a[r][c]; Lm[(TS * cj + c) * LD + TS * ri + r] = a[r][c]; }
        } else if (rhsrow) {
#pragma unroll
            for (int c = 0; c < TS; ++c) Lm[(N48 - 1) * LD + TS * cj + c] = a[0][c];
        }
        __syncthreads();
        // Two waves share the factorisation (the first version had ONE wave do everything: 2,270 clocks per panel against the tile
        // kernel's 2,100 -- the 4 x 4 diagonal factor's four rsq chains sat on the critical path of a wave with nothing else to issue):
        //   wave 0  holds the six tiles.  Per panel: dump the panel's columns (and the NEXT panel's diagonal block as it stands) to LDS,
        //           barrier, take its row and the READY factor L_kk, solve the row, write it back row-major (the factor the
        //           back-substitution reads, and -- fetched as lane (row, k) -- the MFMA operand), one MFMA per live tile;
        //   wave 3  is the look-ahead: behind the same barrier it takes the four rows below the diagonal block from the dump, brings the
        //           next diagonal block up to date (x = a L_kk^-T, t = d - x x^T) and factors it -- the operations of the tile kernel's
        //           look-ahead thread, on every lane -- while wave 0 solves and updates.  ONE workgroup barrier per panel.
        const int lane = tid & 63, wv = tid >> 6, col = lane & 15, kq = lane >> 4;
        const int n_act = P.n_act;
        double *PnB = Xb;                                      // [2][48][4]  the panel's columns, row-major, by parity of the panel
        double *DnB = Xb + 2 * 4 * N48;                        // [2][16]     the next panel's diagonal block, updated through the panel before
        double *LkB = DnB + 32;                                // [2][20]     L_kk (row-major, lower) and 1 / diag of the panel
        static_assert(2 * 4 * N48 + 32 + 40 <= 2 * G * XT, "the panel buffers fit the space of the tile kernel's panel column");
        int fail = 0;
        // Cholesky of a 4 x 4 block (lower part of t) as factor_diag does it; columns past the last free one are identity
        auto factor4 = [&](double (&t)[4][4], int tk, double *Lk) {
            double il[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const bool real = 4 * tk + c < n_act;
                double d = t[c][c];
#pragma unroll
                for (int u = 0; u < c; ++u) d -= t[c][u] * t[c][u];
                if (real && !(d > 0.0)) { fail = 1; d = 1.0; }
                if (!real) d = 1.0;
                const double isd = fast_rsqrt(d);
                t[c][c] = d * isd; il[c] = isd;
#pragma unroll
                for (int r = c + 1; r < 4; ++r) {
                    double v = t[r][c];
#pragma unroll
                    for (int u = 0; u < c; ++u) v -= t[r][u] * t[c][u];
                    t[r][c] = real ? v * isd : 0.0;
                }
            }
            if (lane == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    idg[4 * tk + r] = il[r]; Lk[16 + r] = il[r];
#pragma unroll
                    for (int c = 0; c < 4; ++c) Lk[4 * r + c] = c <= r ? t[r][c] : 0.0;
                }
            }
        };
        d4 acc00 = { 0, 0, 0, 0 }, acc10 = acc00, acc20 = acc00, acc11 = acc00, acc21 = acc00, acc22 = acc00;
        if (wv == 0) {
            auto ld_tile = [&](int I, int Jc) { d4 v; for (int r = 0; r < 4; ++r) v[r] = Lm[(16 * I + kq + 4 * r) * LD + 16 * Jc + col]; return v; };
            acc00 = ld_tile(0, 0); acc10 = ld_tile(1, 0); acc20 = ld_tile(2, 0); acc11 = ld_tile(1, 1); acc21 = ld_tile(2, 1); acc22 = ld_tile(2, 2);
        } else if (wv == 3) {
            double t[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) t[r][c] = Lm[r * LD + c];
            factor4(t, 0, LkB);
        }
        const int row = min(lane, N48 - 1);
        auto dump = [&](double *Pn, const d4 &v, int I, int pc) {
#pragma unroll
            for (int r = 0; r < 4; ++r) Pn[(16 * I + kq + 4 * r) * 4 + pc] = v[r];
        };
        // wave 0, in front of the barrier: the panel's columns and the next panel's diagonal block
        auto dump_step = [&](auto Jtag, int q) {
            constexpr int J = decltype(Jtag)::value;
            const int tk = 4 * J + q;
            double *Pn = PnB + (tk & 1) * 4 * N48, *Dn = DnB + (tk & 1) * 16;
            if ((col >> 2) == q) {
                const int pc = col & 3;
                if constexpr (J == 0) { dump(Pn, acc00, 0, pc); dump(Pn, acc10, 1, pc); dump(Pn, acc20, 2, pc); }
                if constexpr (J == 1) { dump(Pn, acc11, 1, pc); dump(Pn, acc21, 2, pc); }
                if constexpr (J == 2) { dump(Pn, acc22, 2, pc); }
            }
            // block rows / columns 4 (tk + 1) ..: register q + 1 of tile (J, J), lanes of columns 4 (q + 1) ..; across a tile
            // boundary (q = 3) register 0 of tile (J + 1, J + 1), columns 0 .. 3
            const int qn = (q + 1) & 3;
            if ((col >> 2) == qn) {
                double v = 0.0;
                if constexpr (J == 0) v = q == 0 ? acc00[1] : q == 1 ? acc00[2] : q == 2 ? acc00[3] : acc11[0];
                if constexpr (J == 1) v = q == 0 ? acc11[1] : q == 1 ? acc11[2] : q == 2 ? acc11[3] : acc22[0];
                if constexpr (J == 2) v = q == 0 ? acc22[1] : q == 1 ? acc22[2] : acc22[3];
                Dn[4 * kq + (col & 3)] = v;
            }
        };
        // wave 0, behind the barrier: my row of the panel with the ready factor, then the update
        auto solve_step = [&](auto Jtag, int q) {
            constexpr int J = decltype(Jtag)::value;
            const int tk = 4 * J + q;
            const double *Pn = PnB + (tk & 1) * 4 * N48, *Lk = LkB + (tk & 1) * 20;
            const d2 a01 = *reinterpret_cast<const d2 *>(Pn + 4 * row), a23 = *reinterpret_cast<const d2 *>(Pn + 4 * row + 2);
            double l[4][4], il[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const d2 u = *reinterpret_cast<const d2 *>(Lk + 4 * r), v = *reinterpret_cast<const d2 *>(Lk + 4 * r + 2);
                l[r][0] = u[0]; l[r][1] = u[1]; l[r][2] = v[0]; l[r][3] = v[1];
            }
            { const d2 u = *reinterpret_cast<const d2 *>(Lk + 16), v = *reinterpret_cast<const d2 *>(Lk + 18); il[0] = u[0]; il[1] = u[1]; il[2] = v[0]; il[3] = v[1]; }
            const double a4[4] = { a01[0], a01[1], a23[0], a23[1] };
            double x[4];
            const int rel = row - 4 * tk;                  // < 0: a row that is finished; 0..3: a row of the diagonal block
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                double v = a4[c];
#pragma unroll
                for (int u = 0; u < c; ++u) v -= x[u] * l[c][u];
                x[c] = (rel < 0 || (rel < 4 && c > rel)) ? 0.0 : v * il[c];
            }
            if (lane < N48) {
                *reinterpret_cast<d2 *>(Lm + row * LD + 4 * tk) = d2{ x[0], x[1] };
                *reinterpret_cast<d2 *>(Lm + row * LD + 4 * tk + 2) = d2{ x[2], x[3] };
            }
            if (lane == N48 - 1) {
#pragma unroll
                for (int c = 0; c < 4; ++c) wp[4 * tk + c] = 4 * tk + c < n_act ? x[c] : 0.0;
            }
            wave_lds_fence();
            double x0 = 0.0, x1 = 0.0, x2 = 0.0;
            if constexpr (J == 0) x0 = Lm[(col) * LD + 4 * tk + kq];
            if constexpr (J <= 1) x1 = Lm[(16 + col) * LD + 4 * tk + kq];
            x2 = Lm[(32 + col) * LD + 4 * tk + kq];
            if constexpr (J == 0) {
                acc00 = __builtin_amdgcn_mfma_f64_16x16x4f64(-x0, x0, acc00, 0, 0, 0);
                acc10 = __builtin_amdgcn_mfma_f64_16x16x4f64(-x1, x0, acc10, 0, 0, 0);
                acc20 = __builtin_amdgcn_mfma_f64_16x16x4f64(-x2, x0, acc20, 0, 0, 0);
            }
            if constexpr (J <= 1) {
                acc11 = __builtin_amdgcn_mfma_f64_16x16x4f64(-x1, x1, acc11, 0, 0, 0);
                acc21 = __builtin_amdgcn_mfma_f64_16x16x4f64(-x2, x1, acc21, 0, 0, 0);
            }
            acc22 = __builtin_amdgcn_mfma_f64_16x16x4f64(-x2, x2, acc22, 0, 0, 0);
        };
        // wave 3, behind the barrier of panel tk: the diagonal block of panel tk + 1 (the tile kernel's look-ahead thread)
        auto lookahead = [&](int tk) {
            const double *Pn = PnB + (tk & 1) * 4 * N48, *Dn = DnB + (tk & 1) * 16, *Lk = LkB + (tk & 1) * 20;
            double araw[4][4], dt[4][4], f[4][4], fil[4], x[4][4], t[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const d2 a = *reinterpret_cast<const d2 *>(Pn + 4 * (4 * tk + 4 + r)), b = *reinterpret_cast<const d2 *>(Pn + 4 * (4 * tk + 4 + r) + 2);
                araw[r][0] = a[0]; araw[r][1] = a[1]; araw[r][2] = b[0]; araw[r][3] = b[1];
                const d2 c = *reinterpret_cast<const d2 *>(Dn + 4 * r), d = *reinterpret_cast<const d2 *>(Dn + 4 * r + 2);
                dt[r][0] = c[0]; dt[r][1] = c[1]; dt[r][2] = d[0]; dt[r][3] = d[1];
                const d2 u = *reinterpret_cast<const d2 *>(Lk + 4 * r), v = *reinterpret_cast<const d2 *>(Lk + 4 * r + 2);
                f[r][0] = u[0]; f[r][1] = u[1]; f[r][2] = v[0]; f[r][3] = v[1];
            }
            { const d2 u = *reinterpret_cast<const d2 *>(Lk + 16), v = *reinterpret_cast<const d2 *>(Lk + 18); fil[0] = u[0]; fil[1] = u[1]; fil[2] = v[0]; fil[3] = v[1]; }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    double v = araw[r][c];
#pragma unroll
                    for (int u = 0; u < c; ++u) v -= x[r][u] * f[c][u];
                    x[r][c] = v * fil[c];
                }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c <= r; ++c) {
                    double v = dt[r][c];
#pragma unroll
                    for (int u = 0; u < 4; ++u) v -= x[r][u] * x[c][u];
                    t[r][c] = v;
                }
            factor4(t, tk + 1, LkB + ((tk + 1) & 1) * 20);
        };
        PH_ONLY(long long mf_t[6] = { 0, 0, 0, 0, 0, 0 };)
        for (int tk = 0; tk < NP; ++tk) {
            const int q = tk & 3;
            PH_ONLY(if (tk == 1 || tk == 9) mf_t[0] = clock64();)
            if (wv == 0) {
                if (tk < 4) dump_step(std::integral_constant<int, 0>{}, q);
                else if (tk < 8) dump_step(std::integral_constant<int, 1>{}, q);
                else dump_step(std::integral_constant<int, 2>{}, q);
            }
            PH_ONLY(if (tk == 1 || tk == 9) mf_t[1] = clock64();)
            __syncthreads();
            PH_ONLY(if (tk == 1 || tk == 9) mf_t[2] = clock64();)
            if (wv == 0) {
                if (tk < 4) solve_step(std::integral_constant<int, 0>{}, q);
                else if (tk < 8) solve_step(std::integral_constant<int, 1>{}, q);
                else solve_step(std::integral_constant<int, 2>{}, q);
                PH_ONLY(if (tk == 1 || tk == 9) { mf_t[3] = clock64(); asm volatile("" : "+v"(acc22[0])); mf_t[4] = clock64(); if (lane == 0) printf("  MF panel %d wave0: dump %lld  barrier %lld  solve+operands+mfma issue %lld  mfma done %lld\n", tk, mf_t[1] - mf_t[0], mf_t[2] - mf_t[1], mf_t[3] - mf_t[2], mf_t[4] - mf_t[3]); })
            } else if (wv == 3 && tk + 1 < NP) {
                lookahead(tk);
                PH_ONLY(if (tk == 1 || tk == 9) { mf_t[3] = clock64(); if (lane == 0) printf("  MF panel %d wave3: look-ahead %lld\n", tk, mf_t[3] - mf_t[2]); })
            }
        }
        if (fail) s_fail = 1;
        __syncthreads();
    } else {
    // ---- factorisation: one barrier per panel, diagonal tiles factored one panel ahead ---------------------
    // State at the top of step tk: L_kk (factor of diagonal tile tk) and 1 / diag are in Lm / idg; the tiles of
    // column tk (rows below the diagonal, the rhs row among them), updated through panel tk-1, are in Ar; the
    // diagonal tile tk+1, updated through panel tk-1, is in dt.
    //   * trailing threads (ti > tk, tk <= tj <= ti): X_i = A_i L_kk^{-T} and X_j by forward substitution from the
    //     raw tiles (every thread forms the two it needs itself: no second barrier), then A_ij -= X_i X_j^T; the
    //     threads of column tk keep X_i -- their tile of L.  Column tk+1 and diagonal tile tk+2 are published for
    //     the next step.
    //   * one thread of the otherwise idle last wave applies panel tk's update to diagonal tile tk+1 alone and
    //     factors it WHILE the others run the trailing update: the per-panel critical path is
    //     max(factor, update) instead of their sum.
    auto publish_tile = [&](double *dst, const double (&t)[TS][TS]) {
#pragma unroll
        for (int r = 0; r < TS; ++r)
#pragma unroll
            for (int c = 0; c < TS; ++c) dst[r * TS + c] = t[r][c];
    };
    // Cholesky of the TS x TS tile t (lower part, in place); factor -> Lm, inverse diagonal -> idg
    auto factor_diag = [&](double (&t)[TS][TS], int tk) {
        double il[TS];
#pragma unroll
        for (int c = 0; c < TS; ++c) {
            double d = t[c][c];
#pragma unroll
            for (int q = 0; q < c; ++q) d -= t[c][q] * t[c][q];
            if (!(d > 0.0)) { s_fail = 1; d = 1.0; }
            const double isd = fast_rsqrt(d);
            t[c][c] = d * isd; il[c] = isd;
#pragma unroll
            for (int r = c + 1; r < TS; ++r) {
                double v = t[r][c];
#pragma unroll
                for (int q = 0; q < c; ++q) v -= t[r][q] * t[c][q];
                t[r][c] = v * isd;
            }
        }
#pragma unroll
        for (int r = 0; r < TS; ++r) {
            idg[tk * TS + r] = il[r];
#pragma unroll
            for (int c = 0; c < TS; ++c) Lm[(tk * TS + r) * LD + tk * TS + c] = c <= r ? t[r][c] : 0.0;
        }
    };
    struct PanelFactor { double l[TS][TS], il[TS]; };
    auto load_factor = [&](PanelFactor &f, int tk) {
#pragma unroll
        for (int r = 0; r < TS; ++r) {
            f.il[r] = idg[tk * TS + r];
#pragma unroll
            for (int c = 0; c < TS; ++c) f.l[r][c] = Lm[(tk * TS + r) * LD + tk * TS + c];
        }
    };
    // X = A L^{-T}:  x[r][c] = (A[r][c] - sum_{q < c} x[r][q] L[c][q]) / L[c][c]
    auto load_tile = [&](const double *src, double (&t)[TS][TS]) {
#pragma unroll
        for (int r = 0; r < TS; ++r)
#pragma unroll
            for (int c = 0; c < TS; ++c) t[r][c] = src[r * TS + c];
    };
    auto panel_solve = [&](const double (&At)[TS][TS], const PanelFactor &f, double (&x)[TS][TS]) {
#pragma unroll
        for (int r = 0; r < TS; ++r)
#pragma unroll
            for (int c = 0; c < TS; ++c) {
                double v = At[r][c];
#pragma unroll
                for (int q = 0; q < c; ++q) v -= x[r][q] * f.l[c][q];
                x[r][c] = v * f.il[c];
            }
    };
    __shared__ __attribute__((aligned(16))) double s_dt[2][TS * TS];
    const bool dthread = tid == NT - 1;                     // no tile of its own: NP < G (host-checked)
    if (cj == 0 && ri > 0 && (mine || rhsrow)) publish_tile(Xb + ri * XT, a);
    if (ri == 1 && cj == 1 && mine) publish_tile(s_dt[0], a);
    if (tid == 0) factor_diag(a, 0);
    __syncthreads();
    for (int tk = 0; tk < NP; ++tk) {
        const double *Ar = Xb + (tk & 1) * (G * XT);
        double *ArN = Xb + ((tk + 1) & 1) * (G * XT);
        PH_ONLY(
        __shared__ long long s_ph[8];
        if (tk == 3 && (dthread || tid == 11 * G + 5)) s_ph[dthread ? 0 : 4] = wall_clock64();
        )
        if (dthread && tk + 1 < NP) {
            // (all LDS operands requested before the first use: one exposed latency instead of one per group)
            PanelFactor f;
            double araw[TS][TS], dt[TS][TS], x[TS][TS], t[TS][TS];
            load_factor(f, tk);
            load_tile(Ar + (tk + 1) * XT, araw);
            load_tile(s_dt[tk & 1], dt);
            __builtin_amdgcn_sched_barrier(0);
            panel_solve(araw, f, x);
#pragma unroll
            for (int r = 0; r < TS; ++r)
#pragma unroll
                for (int c = 0; c <= r; ++c) {
                    double v = dt[r][c];
#pragma unroll
                    for (int q = 0; q < TS; ++q) v -= x[r][q] * x[c][q];
                    t[r][c] = v;
                }
            PH_ONLY(if (tk == 3) { asm volatile("" : "+v"(t[3][3])); s_ph[1] = wall_clock64(); })
            factor_diag(t, tk + 1);
            PH_ONLY(if (tk == 3) s_ph[2] = wall_clock64();)
        }
        if ((mine || rhsrow) && ri > tk && cj >= tk && !(ri == cj && ri == tk + 1)) {
            PanelFactor f;
            double ai[TS][TS], aj[TS][TS], xi[TS][TS];
            load_factor(f, tk);
            load_tile(Ar + ri * XT, ai);
            load_tile(Ar + cj * XT, aj);
            __builtin_amdgcn_sched_barrier(0);
            panel_solve(ai, f, xi);
            if (cj == tk) {
                // my tile IS the panel column: keep the final factor entries (rhs row: w of this panel)
#pragma unroll
                for (int r = 0; r < TS; ++r)
#pragma unroll
                    for (int c = 0; c < TS; ++c) a[r][c] = xi[r][c];
                if (rhsrow) {
#pragma unroll
                    for (int c = 0; c < TS; ++c) wp[tk * TS + c] = xi[0][c];
                }
            } else {
                double xj[TS][TS];
                panel_solve(aj, f, xj);                    // (diagonal tiles: the same operations as xi -- no divergent copy)
#pragma unroll
                for (int r = 0; r < TS; ++r)
#pragma unroll
                    for (int c = 0; c < TS; ++c) {
                        double v = a[r][c];
#pragma unroll
                        for (int q = 0; q < TS; ++q) v -= xi[r][q] * xj[c][q];
                        a[r][c] = v;
                    }
            }
        }
        {
            // column tk+1 for the next step and diagonal tile tk+2 for the look-ahead thread: ONE store sequence
            const bool col = cj == tk + 1 && ri > tk + 1 && (mine || rhsrow), dg = ri == tk + 2 && cj == tk + 2 && mine;
            if (col || dg) publish_tile(col ? ArN + ri * XT : s_dt[(tk + 1) & 1], a);
        }
        PH_ONLY(if (tk == 3 && tid == 11 * G + 5) { asm volatile("" : "+v"(a[3][3])); s_ph[5] = wall_clock64(); })
        if (tk + 1 < NP) __syncthreads();
        PH_ONLY(if (tk == 3 && tid == 0) { s_ph[6] = wall_clock64(); printf("  panel 3: D x,t %lld  factor %lld | update %lld | to barrier exit %lld\n", s_ph[1] - s_ph[0], s_ph[2] - s_ph[1], s_ph[5] - s_ph[4], s_ph[6] - s_ph[0]); })
    }
    // ---- publish L (the diagonal tiles are there already), back-substitute L^T y = w with one wave -----------
    if (cj < ri && mine) {
#pragma unroll
        for (int r = 0; r < TS; ++r)
#pragma unroll
            for (int c = 0; c < TS; ++c) Lm[(ri * TS + r) * LD + cj * TS + c] = a[r][c];
    }
    __syncthreads();
    }   // (!MF)
    PHASE_STAMP(ts2);
    PH_ONLY(const long long cy2 = clock64();)
    TailOperands tail_ops;
    tail_prefetch(P, S, cur, H, tail_ops);        // in flight during the back-substitution (the tiles' registers are free now)
    if (tid < 64) {
        // blocked back-substitution, TS unknowns per step: all lanes solve the TS x TS upper-triangular
        // diagonal system redundantly (operands by broadcast), then lane i applies the TS columns to w[i]
        constexpr int R = (N + 63) / 64;
        double w[R];
#pragma unroll
        for (int q = 0; q < R; ++q) w[q] = tid + 64 * q < NP * TS ? wp[tid + 64 * q] : 0.0;
        for (int tk = NP - 1; tk >= 0; --tk) {
            const int k0 = tk * TS;
            double y[TS];
#pragma unroll
            for (int c = 0; c < TS; ++c) {
                const int k = k0 + c;
                double v = 0.0;
#pragma unroll
                for (int q = 0; q < R; ++q) if ((k >> 6) == q) v = w[q];
                y[c] = __shfl(v, k & 63);
            }
#pragma unroll
            for (int c = TS - 1; c >= 0; --c) {
                double v = y[c];
#pragma unroll
                for (int q = c + 1; q < TS; ++q) v -= Lm[(k0 + q) * LD + k0 + c] * y[q];
                y[c] = v * idg[k0 + c];
            }
#pragma unroll
            for (int q = 0; q < R; ++q) {
                const int i = tid + 64 * q;
                if (i < k0) {
                    double v = w[q];
#pragma unroll
                    for (int c = 0; c < TS; ++c) v -= Lm[(k0 + c) * LD + i] * y[c];
                    w[q] = v;
                } else if (i < k0 + TS) {
#pragma unroll
                    for (int c = 0; c < TS; ++c) if (i == k0 + c) w[q] = y[c];
                }
            }
        }
#pragma unroll
        for (int q = 0; q < R; ++q) { const int pi = tid + 64 * q < N ? compact_to_padded(P, tid + 64 * q) : -1; if (pi >= 0) yv[pi] = w[q]; }    // back to padded columns
    }
    __syncthreads();
    PHASE_STAMP(ts3);
    reduced_solution_tail(P, S, cur, s_fail, tail_ops, yv, s_sc, s_yh, s_act, sred, FUSED && n_bs > 0 ? epoch : 0);
    PH_ONLY(if (tid == 0) printf("solve_reduced: ctrl %lld operands %lld  factor %lld (%lld shader clocks)  backsub %lld  tail %lld [10 ns]\n", ts0b - ts0, ts1 - ts0b, ts2 - ts1, cy2 - cy1, ts3 - ts2, wall_clock64() - ts3);)
}


// tscm_solve_nd.h -- k_solve_nd: the reduced camera system (what DENSE_SCHUR hands to its dense Cholesky: TS.cpp:271-278,
// multi_calib.cpp:209-216), up to 8 cameras, factored along the camera-pair graph.  Included by tscm_kernels.h.
//
//   A = S_c (H_cc - T) S_c + D_c^2,  rhs = S_c (g_c - t_r)  on the free camera-side columns.
//
// The plan (tscm_nd_plan.h, host, once per solver) orders the cameras by nested dissection of the pair graph -- a LEVEL is a
// set of cameras that are not adjacent, eliminated concurrently; what is left when the rest is a clique is one dense
// block -- cuts their blocks into panels of 4 columns and lists the PHASES: up to four panels, one per camera of a level.
// The kernel is a blocked right-looking Cholesky with the matrix in REGISTERS, one 4 x 4 tile per thread (two where a rig
// of 8 cameras has more than 192 tiles), run over that schedule:
//   * threads 0..191 hold the structurally non-zero tiles (and the right-hand side as an extra tile row, so the
//     forward substitution falls out of the panel solves), in the order their column panel is eliminated: early
//     waves retire early;
//   * a phase has two steps.  A: the owners of the tiles in the phase's panel columns solve X = A L_kk^-T from their
//     registers and publish X (LDS, [slot][row panel]) -- their final factor entries.  B: a trailing tile (i, j) takes
//     A_ij -= X_ik X_jk^T for every panel k of the phase it has both tiles for: two tile loads and 64 FMAs per update,
//     so the tiles next to two cameras of a level (two updates per phase) cost little more than the others.  (Rounds
//     1-3 and the first version here published the RAW column and let every trailing thread solve the two X tiles it
//     needed itself -- one barrier per phase instead of two, but 144 dependent FMAs per update: 0.7 us, and a level's
//     phases with two updates per tile took 1.6 us each.  A version with left-looking "boundary" updates per level paid
//     for its structure lookups on every phase's dependent chain: 1.45 us per phase.)
//   * lanes 0..3 of the fourth wave are the look-ahead lanes: lane q brings the diagonal tile of slot q of the NEXT
//     phase up to date -- from the raw tiles (next panel, this phase's panels) their owners hand it one phase ahead --
//     and factors it (4 x 4 Cholesky, rsq seeds) while the tile threads work: same instructions in the four lanes;
//   * no table lookup on a phase's dependent chain: a tile's coordinates, its update mask and the rows it reads are
//     in its registers, the phase's panels in an SGPR.
// Ring of 8 cameras: 13 phases instead of 25, 227 matrix tiles instead of 325; ring of 4: 9 phases instead of 12.
// Back-substitution: one wave, lane = unknown (two per lane), phases in reverse, the factor tile-packed and transposed in
// LDS (a lane reads its column of a tile as two 16-byte loads; the tile's index follows from the row's structure mask by
// a population count).  Then the common tail (yhat, candidate camera parameters, camera part of the model cost change).
//
// FUSED: the launch also carries the T reduction (workgroups 1 .. n_prod) and the back-substitution workgroups (behind
// them), both in this kernel's 256-thread shape; see the hand-off notes at handoff_store() in tscm_kernels.h.
// grid (1 + n_prod + n_bs) x 256, dynamic LDS NdPlan::lds_doubles (solver) / BsGeom<256>::kLds (back-substitution)
#pragma once

// workgroups of the fused launch per CU: 2.  At 3 the two-tiles-per-thread variant spills ten doubles inside the phase loop, and
// config 5 runs 364.5 us per iteration (130.7 us per rank on 8 shards, all back-substitution groups riding) against 359.9 (126.3,
// none riding) at 2 -- measured in one call, tools/variants.sh
#ifndef TSCM_ND_WGS_PER_CU
#define TSCM_ND_WGS_PER_CU 2
#endif
template <int TPT, bool FUSED>
__global__ __launch_bounds__(kNdThreads, FUSED ? TSCM_ND_WGS_PER_CU : 1) void k_solve_nd(DevProblem P, DevState S, const int4 *__restrict__ nd_map, const int *__restrict__ nd_tab, const int *__restrict__ nd_bs, NdDims nd,
                                                                       int epoch, int withhold, int n_prod, int n_bs, int with_floats)
{
    constexpr int NT = kNdThreads, TS = 4, XT = kNdXT, NPD = 128;
    static_assert(NT == kFusedEntries * kTSlices, "the T reduction runs in the solver's workgroup shape");
    if constexpr (FUSED) {
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
            if (threadIdx.x == 0 && !(withhold && blockIdx.x == 1))        // (withhold: fault injection, tscm_solver_debug_withhold_handoff)
                __hip_atomic_fetch_add(S.t_count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // (no release fence: handoff_store)
            return;
        }
    }
    PHASE_STAMP(ts0);
    const int ctrl_done = S.ctrl->done;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *Ld = lds;                         // [32][20]  factor of each diagonal tile (16, row-major) and 1 / diag (4)
    double *wp = Ld + 20 * kNdMaxPanels;      // [128] forward-substituted rhs  w = L^-1 b, in elimination order
    double *yv = wp + NPD;                    // [128] solution by padded column
    double *s_sc = yv + NPD;                  // [128]
    double *s_yh = s_sc + NPD;                // [128]
    double *s_dt = s_yh + NPD;                // [2][4][16] diagonal tiles handed to the look-ahead lanes (by the phase they are factored for)
    double *s_dr = s_dt + 2 * kNdSlots * 16;  // [2][4][4][16] raw tiles (next phase's panel, this phase's panel) for the same lanes
    int *s_tab = reinterpret_cast<int *>(s_dr + 2 * kNdSlots * kNdSlots * 16);
    static_assert(kNdTabInts % 4 == 0, "the tiles behind the tables start on a 16-byte boundary");
    double *Xs = reinterpret_cast<double *>(s_tab + kNdTabInts);
    const int xs_doubles = max(nd.slots * (nd.NP + 1) * XT, XT * nd.n_lt);
    int *s_bs = reinterpret_cast<int *>(Xs + xs_doubles);                          // [n_phases][bs_rounds][64] the back-substitution's tile table      // [slots][NP + 1][XT] solved panel columns; later the packed factor
    __shared__ int s_fail;
    __shared__ unsigned char s_act[NPD];
    __shared__ double sred[256];
    const int n = P.n_pad;            // <= 128
    const int tid = threadIdx.x;
    const int NP = nd.NP, n_phases = nd.n_phases;
    const int xs_slot = (NP + 1) * XT;
    // ---- head: the operand map and the tables (structure only: written once per solver), the control block ----------------
    // (the tile's coordinates stay packed in two words and are extracted where they are used)
    struct Unit {
        double a[TS][TS];
        int pm[4];                      // update masks: phases 8 w .. 8 w + 7 in word w, four bits each
        int m0, m1, lt;                 // m0: row panel | column panel << 8 | kind << 16 | look-ahead slot that needs this tile raw << 24; m1: phase | slot << 8 of the column panel
        __device__ __forceinline__ int ri() const { return m0 & 0xff; }
        __device__ __forceinline__ int cj() const { return (m0 >> 8) & 0xff; }
        __device__ __forceinline__ int kind() const { return (m0 >> 16) & 0xff; }
        __device__ __forceinline__ int dr() const { return (m0 >> 24) & 0xff; }
        __device__ __forceinline__ int phase_c() const { return m1 & 0xff; }
        __device__ __forceinline__ int slot_c() const { return (m1 >> 8) & 0xff; }
        __device__ __forceinline__ bool diag() const { return ((m0 ^ (m0 >> 8)) & 0xff) == 0; }
    };
    Unit U[TPT];
    int off[TPT][kNdUnitInts];
#pragma unroll
    for (int u = 0; u < TPT; ++u)
#pragma unroll
        for (int q = 0; q < kNdUnitInts / 4; ++q) {
            const int4 v = nd_map[(u * (kNdUnitInts / 4) + q) * NT + tid];
            off[u][4 * q] = v.x; off[u][4 * q + 1] = v.y; off[u][4 * q + 2] = v.z; off[u][4 * q + 3] = v.w;
        }
    for (int i = tid; i < kNdTabInts; i += NT) s_tab[i] = nd_tab[i];
    for (int i = tid; i < nd.n_phases * nd.bs_rounds * 64; i += NT) s_bs[i] = nd_bs[i];
    const int cur = S.ctrl->cur;
    const double radius = S.ctrl->radius;
    const double dmin = S.ctrl->opt.min_lm_diagonal, dmax = S.ctrl->opt.max_lm_diagonal;
    const int ctrl_fail = S.ctrl->lin_fail | *S.fac_fail;       // (fac_fail is cleared in the tail, by the one workgroup that solves)
    for (int i = tid; i < NPD; i += NT) { s_sc[i] = i < n ? S.s_c[i] : 1.0; s_act[i] = i < n ? P.col_active[i] : 0; yv[i] = 0.0; }
    if (ctrl_done) return;
    PHASE_STAMP(tsA);
    const double *H = S.H[cur];
    double sci[TPT][TS], scj[TPT][TS], hh[TPT][TS][TS];
#pragma unroll
    for (int u = 0; u < TPT; ++u) {
#pragma unroll
        for (int r = 0; r < TS; ++r) {
            const int oi = off[u][32 + r], oj = off[u][36 + r];
            sci[u][r] = oi == kNdMapOne ? 1.0 : oi >= 0 ? S.s_c[oi] : 0.0;
            scj[u][r] = oj >= 0 ? S.s_c[oj] : 0.0;
        }
#pragma unroll
        for (int r = 0; r < TS; ++r)
#pragma unroll
            for (int c = 0; c < TS; ++c) { const int oh = off[u][r * TS + c]; hh[u][r][c] = oh >= 0 ? H[oh] : 0.0; }
    }
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
    PHASE_STAMP(tsB);
    const double inv_radius = 1.0 / radius;
#pragma unroll
    for (int u = 0; u < TPT; ++u) {
        U[u].m0 = off[u][40]; U[u].m1 = off[u][42]; U[u].lt = off[u][43];
#pragma unroll
        for (int w = 0; w < 4; ++w) U[u].pm[w] = off[u][44 + w];
#pragma unroll
        for (int r = 0; r < TS; ++r)
#pragma unroll
            for (int c = 0; c < TS; ++c) {
                const int ot = off[u][16 + r * TS + c];
                const double tt = ot >= 0 ? (FUSED ? handoff_load(&S.T[ot]) : S.T[ot]) : 0.0;
                // matrix tiles: S_c (H - T) S_c, damped diagonal; identity on the padding columns.  rhs tiles: row 0 of
                // the map holds (g, t_r) of the panel's columns, s_c of the ROW is stored as 1 there.
                const bool dg = U[u].kind() == 1 && U[u].diag() && r == c;
                double v = sci[u][r] * scj[u][c] * (hh[u][r][c] - tt);
                if (dg) v = off[u][32 + r] >= 0 ? v + fmin(fmax(sci[u][r] * sci[u][r] * hh[u][r][c], dmin), dmax) * inv_radius : 1.0;
                U[u].a[r][c] = v;
            }
    }
    PHASE_STAMP(ts1);
    if (tid == 0) s_fail = ctrl_fail;
    // ---- helpers ----------------------------------------------------------------------------------------------------------------
    // tiles move as 16-byte pieces (every tile in LDS starts on a 16-byte boundary): ds_read_b128 / ds_write_b128.  Left to
    // itself the compiler pairs the 8-byte accesses into ds_read2_b64, which the LDS serves in 16-lane groups with two-way
    // bank conflicts (MI355X_MICROARCH.md, LDS; what cost k_eval_gram 4 M conflict cycles in round 2)
    auto publish_tile = [&](double *dst, const double (&t)[TS][TS]) {
        d2 *p = reinterpret_cast<d2 *>(dst);
#pragma unroll
        for (int r = 0; r < TS; ++r) { p[2 * r] = d2{ t[r][0], t[r][1] }; p[2 * r + 1] = d2{ t[r][2], t[r][3] }; }
    };
    auto load_tile = [&](const double *src, double (&t)[TS][TS]) {
        const d2 *p = reinterpret_cast<const d2 *>(src);
#pragma unroll
        for (int r = 0; r < TS; ++r) { const d2 lo = p[2 * r], hi = p[2 * r + 1]; t[r][0] = lo[0]; t[r][1] = lo[1]; t[r][2] = hi[0]; t[r][3] = hi[1]; }
    };
    // Cholesky of the 4 x 4 tile t (lower part, in place); factor and inverse diagonal -> Ld[k]
    auto factor_diag = [&](double (&t)[TS][TS], int k) {
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
        double *dst = Ld + 20 * k;
#pragma unroll
        for (int r = 0; r < TS; ++r) {
            dst[16 + r] = il[r];
#pragma unroll
            for (int c = 0; c < TS; ++c) dst[r * TS + c] = c <= r ? t[r][c] : 0.0;
        }
    };
    struct PanelFactor { double l[TS][TS], il[TS]; };
    auto load_factor = [&](PanelFactor &f, int k) {
        const double *src = Ld + 20 * k;
#pragma unroll
        for (int r = 0; r < TS; ++r) {
            f.il[r] = src[16 + r];
#pragma unroll
            for (int c = 0; c < TS; ++c) f.l[r][c] = src[r * TS + c];
        }
    };
    // X = A L^-T:  x[r][c] = (A[r][c] - sum_{q < c} x[r][q] L[c][q]) / L[c][c]
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
    const bool dlane = tid >= kNdTileThreads && tid < kNdTileThreads + kNdSlots;
    const int dq = tid - kNdTileThreads;                      // slot of a look-ahead lane
    // The schedule in REGISTERS: lane l of every wave holds the panels of phase l and the row structure of panel l; a phase
    // reads them with v_readlane (wave-uniform index) -- a table in LDS would put two or three dependent LDS round trips
    // (0.1-0.15 us each) on every phase's critical path.  The look-ahead lanes hold their update masks as four words.
    const int lane64 = tid & 63;
    const int sp_v = lane64 < kNdMaxPhases ? nd_tab[kNdTabPhasePanels + lane64] : -1;
    int dm_w[4] = { 0, 0, 0, 0 };
    if (dlane) {
#pragma unroll
        for (int w = 0; w < 4; ++w) dm_w[w] = nd_tab[kNdTabDmask + 4 * dq + w];
    }
    __syncthreads();                                          // the tables are in LDS
    // ---- before the first phase: the diagonal tiles of phases 0 and 1, the raw tiles the look-ahead lanes need in phase 0 ----
#pragma unroll
    for (int u = 0; u < TPT; ++u) {
        if (!U[u].kind()) continue;
        if (U[u].diag() && U[u].phase_c() <= 1) publish_tile(s_dt + (U[u].phase_c() * kNdSlots + U[u].slot_c()) * 16, U[u].a);
        if (U[u].dr() != 0xff && U[u].phase_c() == 0) publish_tile(s_dr + (U[u].dr() * kNdSlots + U[u].slot_c()) * 16, U[u].a);
    }
    __syncthreads();
    if (dlane) {
        const int k = (__builtin_amdgcn_readlane(sp_v, 0) >> (8 * dq)) & 0xff;
        if (k != 0xff) {
            double t[TS][TS];
            load_tile(s_dt + dq * 16, t);
            factor_diag(t, k);
        }
    }
    __syncthreads();
    PH_ONLY(
    __shared__ long long s_pht[kNdMaxPhases + 1];
    if (tid == 0) s_pht[0] = wall_clock64();
    )
    // ---- factorisation: two barriers per phase ----------------------------------------------------------------------------------
    // State at the top of phase t: the factors of its panels are in Ld; every tile is updated through phase t - 1;
    // s_dt[(t + 1) & 1] holds the diagonal tiles of phase t + 1 (updated through t - 1), s_dr[t & 1] the raw tiles (panel of
    // phase t + 1, panel of phase t).
    unsigned sp = (unsigned)__builtin_amdgcn_readlane(sp_v, 0);
    for (int ph = 0; ph < n_phases; ++ph) {
        const unsigned spn = ph + 1 < n_phases ? (unsigned)__builtin_amdgcn_readlane(sp_v, ph + 1) : 0xffffffffu;
        if (tid >= kNdTileThreads) {
            // look-ahead lanes: the diagonal tile of slot dq of the next phase, brought up to date through this phase, factored
            double t[TS][TS];
            const int kp = (spn >> (8 * dq)) & 0xff;
            const bool on = dlane && kp != 0xff;
            {
                const int w8 = ph >> 3;                                          // (wave-uniform selects)
                const unsigned dmask = on ? ((unsigned)(w8 == 0 ? dm_w[0] : w8 == 1 ? dm_w[1] : w8 == 2 ? dm_w[2] : dm_w[3]) >> (4 * (ph & 7))) & 15u : 0u;
                if (on) load_tile(s_dt + (((ph + 1) & 1) * kNdSlots + dq) * 16, t);
                // (ONE block of code whatever slot a lane's update comes from: a wave executes every block any of its lanes needs,
                // and four blocks with one lane each were four times the instructions on the phase's critical path)
                for (unsigned mm = dmask; __builtin_amdgcn_ballot_w64(mm != 0u) != 0ull; mm &= mm - 1u) {
                    if (mm != 0u) {
                        const int q = __builtin_ctz(mm);
                        const int k = (sp >> (8 * q)) & 0xff;
                        PanelFactor f;
                        double araw[TS][TS], x[TS][TS];
                        load_factor(f, k);
                        load_tile(s_dr + (((ph & 1) * kNdSlots + dq) * kNdSlots + q) * 16, araw);
                        panel_solve(araw, f, x);
#pragma unroll
                        for (int r = 0; r < TS; ++r)
#pragma unroll
                            for (int c = 0; c <= r; ++c) {
                                double v = t[r][c];
#pragma unroll
                                for (int e = 0; e < TS; ++e) v -= x[r][e] * x[c][e];
                                t[r][c] = v;
                            }
                    }
                }
            }
            __syncthreads();                                  // (barrier A: nothing of step A concerns these lanes)
            if (on) factor_diag(t, kp);
        } else {
            // step A: the tiles of this phase's panel columns
#pragma unroll
            for (int u = 0; u < TPT; ++u) {
                Unit &T = U[u];
                if (T.kind() && T.phase_c() == ph && !T.diag()) {
                    PanelFactor f;
                    double x[TS][TS];
                    load_factor(f, T.cj());
                    panel_solve(T.a, f, x);
#pragma unroll
                    for (int r = 0; r < TS; ++r)
#pragma unroll
                        for (int c = 0; c < TS; ++c) T.a[r][c] = x[r][c];
                    publish_tile(Xs + T.slot_c() * xs_slot + T.ri() * XT, x);
                    if (T.kind() == 2) {
#pragma unroll
                        for (int c = 0; c < TS; ++c) wp[T.cj() * TS + c] = x[0][c];       // w of this panel
                    }
                }
            }
            __syncthreads();                                  // barrier A
            // step B: trailing updates, then what the next two phases need.  (One block of code per slot; a loop in which every
            // lane walks its own slots -- as the look-ahead lanes do -- was measured slower here: 24.6 against 23.5 us for the 13
            // phases of an 8-camera ring, the per-lane addressing costs more than the blocks a wave skips.)
#pragma unroll
            for (int u = 0; u < TPT; ++u) {
                Unit &T = U[u];
                // (which slots: four bits of the plan's per-tile table -- deriving them from the phase's panels and the tile's mask was 48
                // instructions per phase, at 3 ns each on a wave that is alone on its SIMD)
                const int w8 = ph >> 3;
                const unsigned mm = ((unsigned)(w8 == 0 ? T.pm[0] : w8 == 1 ? T.pm[1] : w8 == 2 ? T.pm[2] : T.pm[3]) >> (4 * (ph & 7))) & 15u;
#pragma unroll
                for (int q = 0; q < kNdSlots; ++q) {
                    if ((mm >> q) & 1u) {
                        const double *base = Xs + q * xs_slot;
                        double xi[TS][TS], xj[TS][TS];
                        load_tile(base + T.ri() * XT, xi);
                        load_tile(base + T.cj() * XT, xj);
#pragma unroll
                        for (int r = 0; r < TS; ++r)
#pragma unroll
                            for (int c = 0; c < TS; ++c) {
                                double v = T.a[r][c];
#pragma unroll
                                for (int e = 0; e < TS; ++e) v -= xi[r][e] * xj[c][e];
                                T.a[r][c] = v;
                            }
                    }
                }
                if (T.kind() && T.phase_c() > ph) {
                    if (T.diag() && T.phase_c() == ph + 2) publish_tile(s_dt + (((ph + 2) & 1) * kNdSlots + T.slot_c()) * 16, T.a);
                    if (T.dr() != 0xff && T.phase_c() == ph + 1) publish_tile(s_dr + ((((ph + 1) & 1) * kNdSlots + T.dr()) * kNdSlots + T.slot_c()) * 16, T.a);
                }
            }
        }
        __syncthreads();                                      // barrier B
        sp = spn;
        PH_ONLY(if (tid == 0) s_pht[ph + 1] = wall_clock64();)
    }
    PHASE_STAMP(ts2);
    // ---- the packed factor (row-major by (row panel, column panel), tiles transposed: a lane of the back-substitution reads one
    //      COLUMN of a tile) -------------------------------------------------------------------------------------------------------
    // (tile pitch XT = 18 doubles, like the panel columns: at a pitch of 16 doubles = 32 banks the 16 lanes of a row that gather a
    // panel's column in the back-substitution hit two bank groups -- a 32-way conflict on every ds_read_b128, 0.6 us per phase)
    double *Lt = Xs;
#pragma unroll
    for (int u = 0; u < TPT; ++u)
        if (U[u].kind() == 1 && U[u].lt >= 0) {
#pragma unroll
            for (int r = 0; r < TS; ++r)
#pragma unroll
                for (int c = 0; c < TS; ++c) Lt[XT * U[u].lt + 4 * c + r] = U[u].a[r][c];
        }
    __syncthreads();
    PHASE_STAMP(ts2b);
    PH_ONLY(__shared__ long long s_bst[kNdMaxPhases + 2];)
    TailOperands tail_ops;
    tail_prefetch(P, S, cur, H, tail_ops);        // in flight during the back-substitution (the tiles' registers are free now)
    if (tid < 64) {
        // L^T y = w, phases in reverse, LEFT-looking: y_k = L_kk^-T (w_k - sum_{i > k} L_ik^T y_i).  The panels of a phase do not
        // depend on each other; the 16 lanes of a DPP row take the tiles of ONE panel's column (table s_bs, from the plan), one
        // tile per lane and round: 16 FMAs, a row-wide sum of the four components (DPP butterfly, fixed order), the 4 x 4
        // upper triangular solve -- about a hundred instructions per phase whatever the number of panels in it.  (The first
        // version walked the panels one after the other with lane = unknown: six hundred instructions per phase of four
        // panels on the one wave that runs this, 1.2 us per phase.)  y overwrites w in LDS (one wave: LDS accesses in order).
        const int lane = tid, q = lane >> 4;
        if (nd.bs_rounds == 1) {
            // one tile per lane and phase (every plan but the dense one of 7-8 cameras).  What does NOT depend on the solution so far
            // -- the phase's table entry, this lane's factor tile, the diagonal factor its group's leader solves with -- is
            // requested one phase AHEAD: a phase's dependent chain is then the load of y_i, 16 FMAs, the row sum, the 4 x 4 solve
            // and the store of y_k (one LDS round trip instead of four)
            // (two operand buffers that take turns -- the loop is unrolled by two -- instead of a copy of 26 doubles per phase: on this
            // one wave an instruction is 2.2-2.9 ns whatever it does)
            struct BsOps { int e; unsigned sp; double lt[TS][TS], ld[10]; };
            auto prefetch = [&](BsOps &o, int ph) {
                o.e = s_bs[ph * 64 + lane];
                o.sp = (unsigned)__builtin_amdgcn_readlane(sp_v, ph);
                load_tile(Lt + XT * (max(o.e, 0) & 0xffff), o.lt);
                const int k = (o.sp >> (8 * q)) & 0xff;
                const double *ld = Ld + 20 * (k == 0xff ? 0 : k);
                o.ld[0] = ld[1 * TS + 0]; o.ld[1] = ld[2 * TS + 0]; o.ld[2] = ld[2 * TS + 1]; o.ld[3] = ld[3 * TS + 0]; o.ld[4] = ld[3 * TS + 1]; o.ld[5] = ld[3 * TS + 2];
#pragma unroll
                for (int c = 0; c < TS; ++c) o.ld[6 + c] = ld[16 + c];
            };
            const int b0 = lane & 1, b1 = (lane >> 1) & 1;
            auto phase = [&](const BsOps &o, BsOps &nxt, int ph) {
                const int e = o.e;
                const int k = (o.sp >> (8 * q)) & 0xff;
                const bool lead = (lane & 15) == 0 && k != 0xff;
                // the loads ON the chain first: y_i of this lane's tile, w_k of the group's leader (LDS returns a wave's loads in order)
                d2 y01 = { 0.0, 0.0 }, y23 = { 0.0, 0.0 }, w01 = { 0.0, 0.0 }, w23 = { 0.0, 0.0 };
                if (e >= 0) { const d2 *yp = reinterpret_cast<const d2 *>(wp + TS * (e >> 16)); y01 = yp[0]; y23 = yp[1]; }
                if (lead) { const d2 *wq = reinterpret_cast<const d2 *>(wp + TS * k); w01 = wq[0]; w23 = wq[1]; }
                __builtin_amdgcn_sched_barrier(0);
                if (ph > 0) prefetch(nxt, ph - 1);
                double acc[TS] = { 0.0, 0.0, 0.0, 0.0 };
                if (e >= 0) {
                    const double y[TS] = { y01[0], y01[1], y23[0], y23[1] };
#pragma unroll
                    for (int c = 0; c < TS; ++c)
#pragma unroll
                        for (int r = 0; r < TS; ++r) acc[c] += o.lt[c][r] * y[r];
                }
                // sum of the four components over the 16 lanes of the row by recursive halving: after two exchange steps a lane
                // holds ONE component summed over its quad (lane & 3 = 0, 1, 2, 3 <-> component 0, 2, 1, 3), two rotations by 4 and 8
                // lanes complete it -- 27 instructions instead of the 48 of four full butterflies; fixed order
                double kA = b0 ? acc[2] : acc[0], kB = b0 ? acc[3] : acc[1];
                const double sA = b0 ? acc[0] : acc[2], sB = b0 ? acc[1] : acc[3];
                kA += dpp_f64<0xB1>(sA); kB += dpp_f64<0xB1>(sB);                  // quad_perm [1,0,3,2]
                double kv = b1 ? kB : kA;
                const double sv = b1 ? kA : kB;
                kv += dpp_f64<0x4E>(sv);                                           // quad_perm [2,3,0,1]
                kv += dpp_f64<0x124>(kv);                                          // row_ror:4
                kv += dpp_f64<0x128>(kv);                                          // row_ror:8
                const double s2 = dpp_f64<0x55>(kv), s1 = dpp_f64<0xAA>(kv), s3 = dpp_f64<0xFF>(kv);      // lanes 1, 2, 3 of the quad -> components 2, 1, 3
                if (lead) {
                    double v[TS] = { w01[0] - kv, w01[1] - s1, w23[0] - s2, w23[1] - s3 };
                    v[3] = v[3] * o.ld[9];
                    v[2] = (v[2] - o.ld[5] * v[3]) * o.ld[8];
                    v[1] = (v[1] - o.ld[2] * v[2] - o.ld[4] * v[3]) * o.ld[7];
                    v[0] = (v[0] - o.ld[0] * v[1] - o.ld[1] * v[2] - o.ld[3] * v[3]) * o.ld[6];
                    d2 *wq = reinterpret_cast<d2 *>(wp + TS * k);
                    wq[0] = d2{ v[0], v[1] }; wq[1] = d2{ v[2], v[3] };
                }
                // (y_k is read by other lanes of THIS wave in the next phase: LDS operations of a wave are served in order, only
                // the compiler must not move them -- a fence here would also wait for the tail's global prefetch)
                asm volatile("" ::: "memory");
                PH_ONLY(if (tid == 0) s_bst[ph] = wall_clock64();)
            };
            BsOps oa, ob;
            prefetch(oa, n_phases - 1);
            PH_ONLY(if (tid == 0) s_bst[n_phases] = wall_clock64();)
            for (int ph = n_phases - 1; ph >= 0; ph -= 2) {
                phase(oa, ob, ph);
                if (ph > 0) phase(ob, oa, ph - 1);
            }
        } else
        for (int ph = n_phases - 1; ph >= 0; --ph) {
            const unsigned spb = (unsigned)__builtin_amdgcn_readlane(sp_v, ph);
            const int k = (spb >> (8 * q)) & 0xff;
            double acc[TS] = { 0.0, 0.0, 0.0, 0.0 };
            for (int rd = 0; rd < nd.bs_rounds; ++rd) {
                const int e = s_bs[(ph * nd.bs_rounds + rd) * 64 + lane];
                if (e >= 0) {
                    double lt[TS][TS], y[TS];                       // lt[c][r] = L_ik[r][c]
                    load_tile(Lt + XT * (e & 0xffff), lt);
                    const d2 *yp = reinterpret_cast<const d2 *>(wp + TS * (e >> 16));
                    const d2 y01 = yp[0], y23 = yp[1];
                    y[0] = y01[0]; y[1] = y01[1]; y[2] = y23[0]; y[3] = y23[1];
#pragma unroll
                    for (int c = 0; c < TS; ++c)
#pragma unroll
                        for (int r = 0; r < TS; ++r) acc[c] += lt[c][r] * y[r];
                }
            }
#pragma unroll
            for (int c = 0; c < TS; ++c) acc[c] = row16_allsum(acc[c]);
            if ((lane & 15) == 0 && k != 0xff) {
                const double *ld = Ld + 20 * k;
                double v[TS];
#pragma unroll
                for (int c = 0; c < TS; ++c) v[c] = wp[TS * k + c] - acc[c];
#pragma unroll
                for (int c = TS - 1; c >= 0; --c) {
                    double x = v[c];
#pragma unroll
                    for (int e = c + 1; e < TS; ++e) x -= ld[e * TS + c] * v[e];
                    v[c] = x * ld[16 + c];
                }
#pragma unroll
                for (int c = 0; c < TS; ++c) wp[TS * k + c] = v[c];
            }
            wave_lds_fence();
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) { const int i = tid + 64 * e; const int pi = i < NP * TS ? s_tab[kNdTabPcol + i] : -1; if (pi >= 0) yv[pi] = wp[i]; }    // back to padded columns
    }
    __syncthreads();
    PHASE_STAMP(ts3);
    reduced_solution_tail(P, S, cur, s_fail, tail_ops, yv, s_sc, s_yh, s_act, sred, FUSED && n_bs > 0 ? epoch : 0);
    PH_ONLY(
    if (tid == 0) {
        printf("solve_nd: ctrl %lld  T wait %lld  operands %lld  to first phase %lld  factor %lld (%d phases)  backsub %lld  tail %lld [10 ns]\n", tsA - ts0, tsB - tsA, ts1 - tsB, s_pht[0] - ts1, ts2 - s_pht[0], n_phases, ts3 - ts2, wall_clock64() - ts3);
        for (int ph = 0; ph < n_phases; ++ph) printf("  phase %d: %lld\n", ph, s_pht[ph + 1] - s_pht[ph]);
        printf("  backsub: factor store + barrier %lld, to loop %lld, phases (last first):", ts2b - ts2, s_bst[n_phases] - ts2b);
        for (int ph = n_phases - 1; ph >= 0; --ph) printf(" %lld", s_bst[ph] - s_bst[ph + 1]);
        printf("\n");
    }
    )
}

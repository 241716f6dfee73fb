// tscm_eval_gram4.h -- the dominant kernel with the Gram contraction on v_mfma_f64_4x4x4_4b_f64 (round 3).
//
// Same inputs, same outputs, same bits as k_eval_gram<58> (tscm_kernels.h); what changes is the matrix instruction.
// Measured on gfx950 (tools/ubench_mfma64.hip, four waves per SIMD): v_mfma_f64_16x16x4_f64 occupies the fp64 pipe for
// ~90 clocks (2048 flop: 23 flop/clock/SIMD -- the 49 TFLOP/s ceiling bench.py reports), v_mfma_f64_4x4x4_4b_f64 for
// ~12.5 clocks (4 blocks of 4x4x4 = 512 flop: 41 flop/clock/SIMD), and the 16x16 tile computes every off-diagonal
// block of the symmetric Gram twice.  With the 16 tile columns in four groups of four, the ten block pairs (I <= J)
// fit THREE 4x4x4 instructions per four rows (twelve block slots): 37.5 clocks instead of 90.
//
// Operand layout of the instruction (tools/probe_mfma4x4.hip): lane l: k = l / 16, block = (l % 16) / 4, A_b[i][k] with
// i = l % 4, B_b[k][j] with j = l % 4; result D_b[i][j] in lane 16 i + 4 b + j.  So lane (c = l % 16, k = l / 16)
// supplies element (tile column c, row 4 t + k) as A operand -- the vector N the 16x16 kernel reads -- block b of the
// result holds rows of column group b, and the B operand decides which column group they meet:
//   q0  B = N                         pairs (0,0) (1,1) (2,2) (3,3)
//   q1  B = N rotated by one group     pairs (0,1) (1,2) (2,3) (3,0)
//   q2  B = groups [2, 3, 1, 1]        pairs (0,2) (1,3) (2,1) (3,1)
// i.e. all ten unordered pairs, and every pair (b, 1) with the column group of the t_c columns as the COLUMN group:
// there the three t_c entries of a tile column sit in the three adjacent lanes of a quad, and the t_b rows
// (J_tb = J_tc R_c) are three quad-broadcast DPP moves and three FMAs per register -- no LDS crossbar, no cross-row move.
// Entries of the 4x4x4 result are bit-identical to those of the 16x16x4 tile (tools/check_mfma4x4.hip), the epilogue
// performs the same operations in the same order, and the camera tile is handed on in the 16x16 layout and column
// numbering of k_eval_gram: the whole solve is bit-identical (tools/regress_bits.py).
//
// Tile columns here (groups of four):  0-2 w_b, 3 r | 4-6 t_c, 7 alpha | 8-10 w_c, 11 f* | 12 one*, 13 xi, 14 lambda, 15 zero.
// LDS tile: element (k-step t, column c, row k) at double t * 68 + 4 c + (k ^ 2 (c >> 3)):
//   * the three operand vectors are conflict-free ds_read_b64 (32-lane groups, banks mod 64: columns c and c + 8 would
//     share a bank pair, the xor puts them on complementary halves);
//   * a corner's lane writes row (t, k) = (lane / 4, lane % 4) of a column: 16 lanes x 8 bytes on 32 distinct banks
//     (k-step stride 68 doubles = 8 dwords mod 32).
//
// Every board size (round 6; rounds 3-5: boards of 49..56 corners only, everything else ran k_eval_gram<0>).  A pass holds
// 4 KS <= 56 rows, KS a template parameter, so a board of n <= 56 corners issues exactly ceil(n / 4) k-steps (6 x 5: 8, not
// 14 zero-padded ones); larger boards -- the reference's own 11 x 8 (main.cpp:190-191) -- take ceil(n / 56) BALANCED passes
// per view (88 = 2 x 44 -> KS = 11), the accumulators carried across the passes, one epilogue per view (MULTI).  The
// corners of a pass are a multiple of four, so the k-steps of a view contract the same groups of four rows in the same
// order as k_eval_gram<0>'s 64-row passes do: same bits.
#pragma once
// (included from tscm_kernels.h inside namespace tscm)

constexpr int kG4Stride = 68;                   // doubles per k-step of the tile
// (kG4MaxKS, g4_plan: tscm_kernels.h, in front of the fp32-Jacobian tier's kernel, which runs on the same pass plan)
// doubles of LDS per wave: the tile (>= the 512-double camera-tile exchange), then the board points
__host__ __device__ constexpr int g4_tile_doubles(int ks) { return ks * kG4Stride > 512 ? ks * kG4Stride : 512; }
__host__ __device__ inline int eval_gram4_lds_doubles(int n_points, int ks) { return g4_tile_doubles(ks) + 2 * n_points; }

// tile columns of this kernel
constexpr int kG4Wb = 0, kG4R = 3, kG4Tc = 4, kG4Al = 7, kG4Wc = 8, kG4F = 11, kG4One = 12, kG4Xi = 13, kG4Lam = 14;
// ... -> tile column of k_eval_gram (the camera tile is handed on in that numbering)
__device__ __forceinline__ constexpr int g4_old_col(int c)
{
    return c < 3 ? kTcWb + c : c == 3 ? kTcR : c < 7 ? tc_tc(c - 4) : c == 7 ? kTcAl : c < 11 ? kTcWc + (c - 8) : c == 11 ? kTcF
         : c == 12 ? kTcOne : c == 13 ? kTcXi : c == 14 ? kTcLam : 15;
}
// ... -> W column (F index) of the record; f* / one* are the first of a (u-part, v-part) pair; -1: none (w_b, zero)
__device__ __forceinline__ constexpr int g4_wcol(int c)
{
    return c < 3 ? -1 : c == 3 ? kFR : c < 7 ? kWcolTc + (c - 4) : c == 7 ? 12 : c < 11 ? c - 8 : c == 11 ? 6 : c == 12 ? 8 : c == 13 ? 10 : c == 14 ? 11 : -1;
}
// column group block b of the result meets in instruction q
__device__ __forceinline__ constexpr int g4_cgroup(int b, int q) { return q == 0 ? b : q == 1 ? ((b + 1) & 3) : (b == 0 ? 2 : b == 1 ? 3 : 1); }
__device__ __forceinline__ int g4_elem(int c, int k) { return 4 * c + (k ^ ((c >> 3) << 1)); }     // within a k-step

// one k-step: three operand vectors, three instructions; operands requested D k-steps ahead
template <int KS, int D, bool ACC, int T = 0>
__device__ __forceinline__ void gram4_steps(unsigned aN, unsigned aR, unsigned aB, double (&n)[KS], double (&r)[KS], double (&b)[KS], double (&acc)[3])
{
    if constexpr (T < KS) {
        if constexpr (T + D < KS) {
            n[T + D] = ds_read_f64<8 * kG4Stride * (T + D)>(aN);
            r[T + D] = ds_read_f64<8 * kG4Stride * (T + D)>(aR);
            b[T + D] = ds_read_f64<8 * kG4Stride * (T + D)>(aB);
        }
        constexpr int newer = 3 * (KS - 1 - T < D ? KS - 1 - T : D);      // requests younger than this k-step's three
        lgkm_wait<newer + 2>(n[T]);
        if constexpr (T == 0 && !ACC) acc[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(n[T], n[T], 0.0, 0, 0, 0);
        else acc[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(n[T], n[T], acc[0], 0, 0, 0);
        lgkm_wait<newer + 1>(r[T]);
        if constexpr (T == 0 && !ACC) acc[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(n[T], r[T], 0.0, 0, 0, 0);
        else acc[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(n[T], r[T], acc[1], 0, 0, 0);
        lgkm_wait<newer>(b[T]);
        if constexpr (T == 0 && !ACC) acc[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(n[T], b[T], 0.0, 0, 0, 0);
        else acc[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(n[T], b[T], acc[2], 0, 0, 0);
        gram4_steps<KS, D, ACC, T + 1>(aN, aR, aB, n, r, b, acc);
    }
}
template <int KS, int D, int T = 0>
__device__ __forceinline__ void gram4_prime(unsigned aN, unsigned aR, unsigned aB, double (&n)[KS], double (&r)[KS], double (&b)[KS])
{
    if constexpr (T < D && T < KS) {
        n[T] = ds_read_f64<8 * kG4Stride * T>(aN);
        r[T] = ds_read_f64<8 * kG4Stride * T>(aR);
        b[T] = ds_read_f64<8 * kG4Stride * T>(aB);
        gram4_prime<KS, D, T + 1>(aN, aR, aB, n, r, b);
    }
}
// ACC: the accumulators carry the earlier passes of the view (otherwise the first k-step starts from C = 0)
template <int KS, bool ACC>
__device__ __forceinline__ void gram4_full(unsigned aN, unsigned aR, unsigned aB, double (&acc)[3])
{
    constexpr int D = KS < 2 ? KS : 2;      // (round 5: 1, 2, 3 and 4 k-steps ahead measure the same to 0.1 us -- the MFMA phases do not wait for the LDS)
    double n[KS], r[KS], b[KS];
    gram4_prime<KS, D>(aN, aR, aB, n, r, b);
    gram4_steps<KS, D, ACC>(aN, aR, aB, n, r, b, acc);
}

// t_b entry of this lane's quad: lanes j = 0, 1, 2 of a quad hold the t_c0, t_c1, t_c2 entries of one tile column; lane
// j = l gets R_c[0][l] x0 + R_c[1][l] x1 + R_c[2][l] x2 in the operation order of store_view_record (c_m = R_c[m][l])
__device__ __forceinline__ double quad_tb(double x, double c0, double c1, double c2)
{
    const double x0 = dpp_f64<0x00>(x), x1 = dpp_f64<0x55>(x), x2 = dpp_f64<0xAA>(x);
    return fma(c2, x2, fma(c0, x0, c1 * x1));
}

// KS: k-steps of a pass (4 KS rows >= the corners of a pass); MULTI: boards of more than 56 corners, P.g4_per corners per pass
template <int KS, bool MULTI>
__global__ __launch_bounds__(256, 4) void k_eval_gram4(DevProblem P, DevState S, int cand)
{
    static_assert(KS >= 1 && KS <= kG4MaxKS, "a pass holds at most 64 rows");
    constexpr int kTile = g4_tile_doubles(KS);
    KTL(0);
    const int ctrl_done = S.ctrl->done, ctrl_cur = S.ctrl->cur;
    TL_ONLY(
    const long long tl_t0 = wall_clock64();
    const int tl_iter = S.ctrl->iteration;
    long long tl_ph[5] = { 0, 0, 0, 0, 0 };      // shader clocks per phase, summed over the chunk: geometry, MFMA u, copy, MFMA v, epilogue
    // wall-clock stamps (10 ns) of this wave: [0] view loop reached, [1] first MFMA, [2] the wave's very end (camera tile handed
    // on; written at the end of the kernel), [3] last record stored, [4 + i] start of view i (i < kTlViews)
    long long tl_w[4] = { 0, 0, 0, 0 };
    int tl_nv = 0;
    )
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lds_wave = eval_gram4_lds_doubles(P.n_points, KS);
    double *lds = lds_all + (size_t)wave * lds_wave;
    double *Fl = lds;                          // the tile: holds the u-rows, then the v-rows
    double *bxy = lds + kTile;
    const int lane = threadIdx.x & 63;
    const int chunk = blockIdx.x * 4 + wave;
    // Head in TWO dependent round trips (round 5; it was six: control block | board points, twice | view range | camera
    // constants | view metadata | observations -- 7 us from a wave's start to its first MFMA, with every wave of the chip in
    // the same place at the same time).  Trip 1, scalar: the control block and the chunk's 16-byte descriptor; vector: the
    // lane's board point.  Trip 2, everything that hangs on the descriptor at once: R_c, the metadata of the chunk's first
    // 64 views, the first view's observations (all 64 lanes: a lane without a corner reads the next view's value or, past
    // the end, zero, and never uses it), and one dword of every 64-byte line of the first view's and of the camera's
    // constants through the SCALAR cache, so that the geometry's scalar loads hit there.
    const v4i cd = *(const v4i __attribute__((address_space(4))) *)(const void *)(P.chunk_desc + chunk);
    const int cam = cd[0], vb = cd[1], ve = cd[2];
    const d2 my_xy = *reinterpret_cast<const d2 *>(P.board_xy + 2 * min(lane, P.n_points - 1));
    double *const cc_buf[2] = { S.cconst[0], S.cconst[1] };
    double *const rec_buf[2] = { S.rec[0], S.rec[1] };
    double camU[3] = { 0.0, 0.0, 0.0 }, camV[3] = { 0.0, 0.0, 0.0 };
    for (int i = lane; i < KS * kG4Stride; i += 64) Fl[i] = 0.0;     // rows of lanes without a corner, the zero column, the padding
    int prev_nv = 0;
    double pf_u = 0.0, pf_v = 0.0;
    int warm = 0;
    if (ctrl_done) return;
    const int tgt = cand ? (ctrl_cur ^ 1) : ctrl_cur;
    const __amdgpu_buffer_rsrc_t r_rec = make_rsrc(tgt ? rec_buf[1] : rec_buf[0], sizeof(double) * (size_t)kRec * P.V);
    const cptr4 ccs = (cptr4)((tgt ? cc_buf[1] : cc_buf[0]) + kCStride * cam);
    auto CC = [&](int k) { return ccs[k]; };
    const __amdgpu_buffer_rsrc_t r_vc = make_rsrc(S.vconst, sizeof(double) * (size_t)kVStride * P.V);
    const __amdgpu_buffer_rsrc_t r_u = make_rsrc(P.obs_u, sizeof(double) * (size_t)P.N), r_v = make_rsrc(P.obs_v, sizeof(double) * (size_t)P.N);
    int off_next = cd[3];
    int m_cnt0 = 0, m_slot0 = 0;
    if (vb + lane < min(ve, vb + 64)) { m_cnt0 = P.view_count[vb + lane]; m_slot0 = P.view_slot[vb + lane]; }
    if (vb < ve) { pf_u = buf_load_f64(r_u, 8u * lane, 8u * (unsigned)off_next); pf_v = buf_load_f64(r_v, 8u * lane, 8u * (unsigned)off_next); }
    // the scalar cache's lines of the first view's 27 constants (4 lines) and of the camera's 48 (6 lines); retired in front of the loop
    typedef const int __attribute__((address_space(4))) *cptr4i;
    const cptr4i vc0 = (cptr4i)(S.vconst + (size_t)kVStride * min(vb, P.V - 1)), cc0 = (cptr4i)ccs;
    // (as dwords on purpose: the same words loaded as doubles are merged with the geometry's own loads of them, which moves
    // contraction decisions in the geometry and with them the last bits of 8 of the 13 fingerprints of tools/regress_bits.py)
    const int kw0 = vc0[0], kw1 = vc0[16], kw2 = vc0[32], kw3 = vc0[48], kw4 = cc0[16], kw5 = cc0[32], kw6 = cc0[48], kw7 = cc0[64], kw8 = cc0[80];
    // ---- lane roles ------------------------------------------------------------------------------------------------
    // as a corner: row (t, k) of the tile
    double *fu_lo = Fl + (lane >> 2) * kG4Stride + (lane & 3), *fu_hi = Fl + (lane >> 2) * kG4Stride + ((lane & 3) ^ 2);
    // as an MFMA lane: k = i = lane / 16, c16 = 4 b + j
    const int li = lane >> 4, c16 = lane & 15, lb = c16 >> 2, lj = c16 & 3;
    const unsigned lds0 = lds_addr(Fl);
    const unsigned aN = lds0 + 8u * (unsigned)g4_elem(c16, li);
    const unsigned aR = lds0 + 8u * (unsigned)g4_elem((c16 + 4) & 15, li);
    const unsigned aB = lds0 + 8u * (unsigned)g4_elem(4 * g4_cgroup(lb, 2) + lj, li);
    // R_c[m][j] of this lane's quad position (t_b rows), fixed for the chunk
    const double rc0 = lj == 0 ? ccs[0] : lj == 1 ? ccs[1] : ccs[2];
    const double rc1 = lj == 0 ? ccs[3] : lj == 1 ? ccs[4] : ccs[5];
    const double rc2 = lj == 0 ? ccs[6] : lj == 1 ? ccs[7] : ccs[8];
    // record offsets of this lane's entries in the seven stores of a view (bytes inside the view's W / E record; lanes
    // without an entry store past the end of the buffer).  Columns by group: see g4_wcol.
    constexpr unsigned BAD = 0xffffe000u;
    unsigned o1e = BAD, o1w = BAD, o2 = BAD, o3e = BAD, o4 = BAD, o5 = BAD, o6 = BAD, o7a = BAD, o7b = BAD;
    {
        const int b = lb, j = lj, i = li;
        auto W = [](int wcol, int row) { return 8u * (unsigned)(6 * wcol + row); };
        const int f1 = g4_wcol(4 + i), f1j = g4_wcol(4 + j), f2 = g4_wcol(8 + i), f2j = g4_wcol(8 + j), f3 = g4_wcol(12 + i);
        if (b == 0 && j < 3 && i < 3) { o1e = 8u * (unsigned)(6 * j + i); o3e = 8u * (unsigned)(6 * i + 3 + j); }
        if (b == 0 && j == 3 && i < 3) { o1w = W(kFR, i); o7a = 8u * (unsigned)i; }                 // w_b rows x r (and once more in the G region)
        if (b == 1 && j < 3) o1w = W(f1, 3 + j);                             // t_b rows x (t_c | alpha)
        if (b == 0 && i < 3) o2 = W(f1j, i);                                 // w_b rows x (t_c | alpha)
        if (b == 0 && i == 3 && j < 3) { o2 = W(kFR, 3 + j); o7b = 8u * (unsigned)(3 + j); }       // t_b rows x r (G region)
        if (b == 3 && j < 3 && i < 3) o2 = W(f3, j);                         // w_b rows x (one* | xi | lambda), held transposed
        if (b == 0 && i < 3) o4 = W(f2j, i);                                 // w_b rows x (w_c | f*)
        if (b == 2 && j < 3) o4 = W(f2, 3 + j);                              // t_b rows x (w_c | f*)
        if (b == 3 && j < 3 && i < 3) o4 = W(f3, 3 + j);                     // t_b rows x (one* | xi | lambda)
        if (b == 3 && i == 0 && j < 3) { o5 = W(9, j); o6 = W(9, 3 + j); }   // v-row parts: cy
        if (b == 0 && j == 3 && i < 3) o6 = W(7, i);                         // ... fy
        if (b == 2 && i == 3 && j < 3) o6 = W(7, 3 + j);
    }
    if (lane < P.n_points) *reinterpret_cast<d2 *>(bxy + 2 * lane) = my_xy;
    if constexpr (MULTI) {
        for (int j = lane + 64; j < P.n_points; j += 64) *reinterpret_cast<d2 *>(bxy + 2 * j) = *reinterpret_cast<const d2 *>(P.board_xy + 2 * j);
    }
    const int per = MULTI ? P.g4_per : 4 * KS;          // corners of a pass
    asm volatile("" :: "s"(kw0), "s"(kw1), "s"(kw2), "s"(kw3), "s"(kw4), "s"(kw5), "s"(kw6), "s"(kw7), "s"(kw8));
    TL_ONLY(
    tl_w[0] = wall_clock64();
    const bool tl_on = lane == 0 && tl_iter == 5 && cand && chunk < kTimelineWaves;
    )
    for (int vbase = vb; vbase < ve; vbase += 64) {
    const int vend = min(ve, vbase + 64);
    int m_cnt = 0, m_slot = 0;
    if (vbase == vb) { m_cnt = m_cnt0; m_slot = m_slot0; }
    else {
        if (vbase + lane < vend) { m_cnt = P.view_count[vbase + lane]; m_slot = P.view_slot[vbase + lane]; }
        pf_u = buf_load_f64(r_u, 8u * lane, 8u * (unsigned)off_next); pf_v = buf_load_f64(r_v, 8u * lane, 8u * (unsigned)off_next);
    }
    asm volatile("" : "+v"(m_cnt), "+v"(m_slot));
    for (int view = vbase; view < vend; ++view) {
        const int cnt = __builtin_amdgcn_readlane(m_cnt, view - vbase);
        const int off = off_next;               // (MULTI: the later passes' observations)
        off_next += cnt;
        wave_lds_fence();                       // the previous view's MFMA phase has finished with the tile
#if TSCM_PRIO
        // priority by progress: see k_eval_gram.  Short chunks (round 5): the last three views step down 3 | 2 | 1 (geometry), 0 (the
        // rest of the last view), so that the four waves of a SIMD -- served oldest first among equals -- enter the last
        // equal-priority stretch half a view apart, not a view and a quarter: -0.4 us at 10 views per wave; at 40 the stretch is a
        // twentieth of the chunk either way and the wrap of the levels in front of it costs more than it brings (+1.6 us)
        const int left = ve - 1 - view;
        const bool tail_steps = left <= 2 && ve - vb <= 16;
        set_prio(tail_steps ? left + 1 : 3 - min(3, 8 * (view - vb) / max(1, ve - vb) % 4));
#endif
        TL_ONLY(
        if (tl_on && tl_nv < kTlViews) g_tlv[(size_t)(4 + kTlViews) * chunk + 4 + tl_nv] = wall_clock64();
        ++tl_nv;
        )
        const cptr4 vcs = (cptr4)(S.vconst + (size_t)kVStride * view);
        auto VC = [&](int k) { return vcs[k]; };
        double accU[3] = { 0.0, 0.0, 0.0 }, accV[3] = { 0.0, 0.0, 0.0 };
        int pbv = 0;                            // first corner of the pass (MULTI)
        do {
        const int pb = MULTI ? pbv : 0;
        TL_STAMP(ts0);
        if constexpr (MULTI) { if (pb) wave_lds_fence(); }      // ... and so has the previous pass's
        const bool valid = lane < (MULTI ? min(per, cnt - pb) : cnt);
        double fv[16];                          // v-rows wait in registers until the u-rows have been consumed (index = tile column)
        auto PUT = [&](int c, double u, double v) { (c < 8 ? fu_lo : fu_hi)[4 * c] = u; fv[c] = v; };
        if (valid) {
            const double x = bxy[2 * (pb + lane)], y = bxy[2 * (pb + lane) + 1];
            // semantic column -> tile column of this kernel
            constexpr int tcol[15] = { kG4Wb, kG4Wb + 1, kG4Wb + 2, kG4Tc, kG4Tc + 1, kG4Tc + 2, kG4Wc, kG4Wc + 1, kG4Wc + 2,
                                       kG4F, kG4One, kG4Xi, kG4Lam, kG4Al, kG4R };
            corner_geometry(x, y, pf_u, pf_v, VC, CC, [&](int gc, double u, double v) { PUT(tcol[gc], u, v); });
        } else if (lane < prev_nv) {
#pragma unroll
            for (int c = 0; c < kTcols; ++c) (c < 8 ? fu_lo : fu_hi)[4 * c] = 0.0;
        }
        {
            // prefetch of the next view (see k_eval_gram): observations, and the constant record into the L2
            const int vn = min(view + 1, vend - 1);
            const int cn = view + 1 < vend ? __builtin_amdgcn_readlane(m_cnt, vn - vbase) : 0;
            if (!MULTI || pb == 0) warm = __builtin_amdgcn_raw_buffer_load_b32(r_vc, lane < 4 ? 64 * lane : (int)0xffffe000u, (int)(8u * (unsigned)kVStride * (unsigned)vn), 0);
            // MULTI: the next pass of this view, or the first pass of the next one
            const bool more = MULTI && pb + per < cnt;
            const int ncnt = MULTI ? (more ? min(per, cnt - pb - per) : min(per, cn)) : cn;
            const unsigned noff = more ? (unsigned)(off + pb + per) : (unsigned)off_next;
            pf_u = buf_load_f64(r_u, lane < ncnt ? 8u * lane : 0xffffe000u, 8u * noff);
            pf_v = buf_load_f64(r_v, lane < ncnt ? 8u * lane : 0xffffe000u, 8u * noff);
        }
        prev_nv = MULTI ? min(per, max(cnt - pb, 0)) : cnt;
        wave_lds_fence();
#if TSCM_PRIO
        if (tail_steps && left == 0) set_prio(0);
#endif
        TL_STAMP(ts1);
        TL_ONLY(if (tl_nv == 1) tl_w[1] = wall_clock64();)
        gram4_full<KS, MULTI>(aN, aR, aB, accU);
        wave_lds_fence();
        TL_STAMP(ts2);
        if (valid) {
#pragma unroll
            for (int c = 0; c < kTcols; ++c) (c < 8 ? fu_lo : fu_hi)[4 * c] = fv[c];
        }
        wave_lds_fence();
        TL_STAMP(ts3);
        gram4_full<KS, MULTI>(aN, aR, aB, accV);
        TL_STAMP(ts4);
        TL_ADD(0, ts0, ts1); TL_ADD(1, ts1, ts2); TL_ADD(2, ts2, ts3); TL_ADD(3, ts3, ts4);
        pbv += per;
        } while (MULTI && pbv < cnt);
        TL_STAMP(ts4e);
        asm volatile("" :: "v"(warm));       // the warming load retires here, before this view's record stores
#pragma unroll
        for (int q = 0; q < 3; ++q) { camU[q] += accU[q]; camV[q] += accV[q]; }
        // ---- epilogue: the view's record (same values, same operation order as store_view_record) ------------------
        {
            int le = lane;
            asm volatile("" : "+v"(le));         // lane predicates are rebuilt per view instead of living in SGPR pairs
            const int b = (le >> 2) & 3, i = le >> 4;
            const unsigned slot = (unsigned)__builtin_amdgcn_readlane(m_slot, view - vbase);
            const unsigned offW = 8u * (unsigned)kRecW * slot, offE = 8u * ((unsigned)kRecW * (unsigned)P.V + (unsigned)kRecE * slot);
            const double T0 = accU[0] + accV[0], T1 = accU[1] + accV[1], T2 = accU[2] + accV[2];
            const double tb0 = quad_tb(T0, rc0, rc1, rc2), tb1 = quad_tb(T1, rc0, rc1, rc2), tb2 = quad_tb(T2, rc0, rc1, rc2);
            const double tbU2 = quad_tb(accU[2], rc0, rc1, rc2);
            const bool split1 = b == 3 && i == 0, split2 = (le & 3) == 3 ? b == 0 : (b == 2 ? i == 3 : (b == 3 && i == 0));
            buf_store_f64(r_rec, o1e, offE, T0);
            buf_store_f64(r_rec, o1w, offW, b == 1 ? tb0 : T0);
            buf_store_f64(r_rec, o2, offW, split1 ? accU[1] : (i == 3 ? tb1 : T1));
            buf_store_f64(r_rec, o3e, offE, tb1);
            const double a4 = b != 0 ? tb2 : T2, u4 = b != 0 ? tbU2 : accU[2];
            buf_store_f64(r_rec, o4, offW, split2 ? u4 : a4);
            buf_store_f64(r_rec, o5, offW, T1 - accU[1]);
            buf_store_f64(r_rec, o6, offW, a4 - u4);
            const unsigned offG = 8u * ((unsigned)(kRecW + kRecE) * (unsigned)P.V + (unsigned)kRecG * slot);
            buf_store_f64(r_rec, o7a, offG, T0);          // E^T r once more, compact (k_reduce_stats reads it there)
            buf_store_f64(r_rec, o7b, offG, tb1);
        }
        TL_ONLY(
        { TL_STAMP(ts5); TL_ADD(4, ts4e, ts5); }
        if (view + 1 == ve) tl_w[3] = wall_clock64();
        )
    }
    }   // block of <= 64 views
    TL_ONLY(
    if (lane == 0 && tl_iter == 5 && cand && chunk < kTimelineWaves) {
        g_timeline[4 * chunk] = (long long)__builtin_amdgcn_s_getreg(63492);
        g_timeline[4 * chunk + 1] = (long long)__builtin_amdgcn_s_getreg(6164);
        g_timeline[4 * chunk + 2] = tl_t0;
        g_timeline[4 * chunk + 3] = wall_clock64();
        for (int k = 0; k < 5; ++k) g_phase[5 * chunk + k] = tl_ph[k];
        for (int k = 0; k < 4; ++k) g_tlv[(size_t)(4 + kTlViews) * chunk + k] = tl_w[k];
    }
    )
    // the camera tile leaves in the 16x16 layout and column numbering of k_eval_gram (both triangles: every entry is
    // written by the lane that holds it and, mirrored, by the same lane; the doubly held pairs carry the same bits)
    wave_lds_fence();
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int r = g4_old_col(4 * lb + li), c = g4_old_col(4 * g4_cgroup(lb, q) + lj);
        lds[16 * r + c] = camU[q]; lds[16 * c + r] = camU[q];
        lds[256 + 16 * r + c] = camV[q]; lds[256 + 16 * c + r] = camV[q];
    }
    __syncthreads();
    {
        const int t = threadIdx.x;
        const size_t st = lds_wave;
        double *part = S.campart + (size_t)512 * blockIdx.x;
        part[t] = (lds_all[t] + lds_all[st + t]) + (lds_all[2 * st + t] + lds_all[3 * st + t]);
        part[256 + t] = (lds_all[256 + t] + lds_all[st + 256 + t]) + (lds_all[2 * st + 256 + t] + lds_all[3 * st + 256 + t]);
    }
    TL_ONLY(if (lane == 0 && tl_iter == 5 && cand && chunk < kTimelineWaves) g_tlv[(size_t)(4 + kTlViews) * chunk + 2] = wall_clock64();)     // [2]: the wave's very end (camera tile handed on)
}

// ---------------------------------------------------------------------------------------------------------------------------
// k_eval_gram4p<KS, M>: SMALL boards (round 6) -- M consecutive views of the chunk share a pass.
// A board of n <= 32 corners leaves half of the 64 lanes of k_eval_gram4<KS, false> without a corner, and the geometry (263
// VALU instructions per pass whatever the number of live lanes) is a third of the kernel: at 6 x 5 a corner cost 35.9 ps against
// 24.9 at 9 x 6.  Here lane l works for view l / (4 KS) of the pass, corner l % (4 KS): one geometry pass serves M views.
//   * the views' 27 constants are no longer wave-uniform: the M records are staged in LDS (one double per lane, prefetched a
//     pass ahead) and read with the lane's own base -- broadcast reads inside a view's lanes;
//   * the observations of a camera's consecutive views are contiguous: the lane's offset is the running sum of the counts
//     of the views in front of its own (v_readlane of the block's metadata, M - 1 selects);
//   * the tile holds the M views' rows one after the other (KS k-steps each, M KS <= 16); every view keeps ITS OWN accumulators
//     (its record needs its own Gram), so the contraction issues exactly the MFMAs of M separate views: the k-steps of view
//     s are tile rows 4 KS s ..; u-rows of all M views, then v-rows, epilogue per view.
// Same operations per corner and per view in the same order as k_eval_gram4<KS, false> -- only WHO computes them changes: the
// whole solve is bit-identical (tests/test_gpu_parity.py).  Views whose corner count differs from n (ragged) are handled: the
// speculative first load of a block assumes full views and is repeated if that was wrong.
// ---------------------------------------------------------------------------------------------------------------------------
__host__ __device__ constexpr int g4p_views(int ks) { return ks > 8 ? 1 : (16 / ks > 4 ? 4 : 16 / ks); }       // views per pass: M KS <= 16, at most 4
__host__ __device__ inline int eval_gram4p_lds_doubles(int n_points, int ks, int m) { return g4_tile_doubles(ks * m) + 2 * n_points + 32 * m; }

template <int KS, int M>
__global__ __launch_bounds__(256, 4) void k_eval_gram4p(DevProblem P, DevState S, int cand)
{
    static_assert(M >= 2 && M <= 4 && KS * M <= 16, "M views of KS k-steps each in the 64 rows of a pass");
    constexpr int RV = 4 * KS;                      // rows (= lanes) of a view
    constexpr int kTile = g4_tile_doubles(KS * M);
    constexpr int BL = 64 / M * M;                  // views per metadata block: whole passes
    KTL(0);
    const int ctrl_done = S.ctrl->done, ctrl_cur = S.ctrl->cur;
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lds_wave = eval_gram4p_lds_doubles(P.n_points, KS, M);
    double *lds = lds_all + (size_t)wave * lds_wave;
    double *Fl = lds;
    double *bxy = lds + kTile;
    double *vcl = bxy + 2 * P.n_points;             // [M][32] the views' constants of the current pass
    const int lane = threadIdx.x & 63;
    const int sv = lane / RV, jl = lane - sv * RV;  // my view of the pass, my corner
    const int chunk = blockIdx.x * 4 + wave;
    const v4i cd = *(const v4i __attribute__((address_space(4))) *)(const void *)(P.chunk_desc + chunk);
    const int cam = cd[0], vb = cd[1], ve = cd[2];
    const d2 my_xy = *reinterpret_cast<const d2 *>(P.board_xy + 2 * min(jl, P.n_points - 1));
    double *const cc_buf[2] = { S.cconst[0], S.cconst[1] };
    double *const rec_buf[2] = { S.rec[0], S.rec[1] };
    double camU[3] = { 0.0, 0.0, 0.0 }, camV[3] = { 0.0, 0.0, 0.0 };
    for (int i = lane; i < KS * M * kG4Stride; i += 64) Fl[i] = 0.0;
    bool prev_valid = false;
    double pf_u = 0.0, pf_v = 0.0;
    if (ctrl_done) return;
    const int tgt = cand ? (ctrl_cur ^ 1) : ctrl_cur;
    const __amdgpu_buffer_rsrc_t r_rec = make_rsrc(tgt ? rec_buf[1] : rec_buf[0], sizeof(double) * (size_t)kRec * P.V);
    const cptr4 ccs = (cptr4)((tgt ? cc_buf[1] : cc_buf[0]) + kCStride * cam);
    auto CC = [&](int k) { return ccs[k]; };
    const __amdgpu_buffer_rsrc_t r_vc = make_rsrc(S.vconst, sizeof(double) * (size_t)kVStride * P.V);
    const __amdgpu_buffer_rsrc_t r_u = make_rsrc(P.obs_u, sizeof(double) * (size_t)P.N), r_v = make_rsrc(P.obs_v, sizeof(double) * (size_t)P.N);
    int off_next = cd[3];
    constexpr unsigned BAD = 0xffffe000u;
    // the constants of the views of a pass: lane l (and l + 64 for M > 2) fetches double l % 32 of view l / 32 of the pass
    double vc_pf0 = 0.0, vc_pf1 = 0.0;
    auto request_vc = [&](int view0, int vend_) {
        const int s0 = lane >> 5, k = lane & 31;
        vc_pf0 = buf_load_f64(r_vc, (s0 < M && view0 + s0 < vend_ && k < kVConst) ? 8u * (unsigned)(kVStride * s0 + k) : BAD, 8u * (unsigned)kVStride * (unsigned)view0);
        if constexpr (M > 2) vc_pf1 = buf_load_f64(r_vc, (s0 + 2 < M && view0 + s0 + 2 < vend_ && k < kVConst) ? 8u * (unsigned)(kVStride * (s0 + 2) + k) : BAD, 8u * (unsigned)kVStride * (unsigned)view0);
    };
    int m_cnt0 = 0, m_slot0 = 0;
    if (vb + lane < min(ve, vb + BL)) { m_cnt0 = P.view_count[vb + lane]; m_slot0 = P.view_slot[vb + lane]; }
    request_vc(vb, ve);
    // the first pass's observations, requested before the counts are known: full views assumed (checked at the top of the block)
    if (vb < ve) {
        const unsigned o = (jl < P.n_points && sv < M) ? 8u * (unsigned)(sv * P.n_points + jl) : BAD;
        pf_u = buf_load_f64(r_u, o, 8u * (unsigned)off_next); pf_v = buf_load_f64(r_v, o, 8u * (unsigned)off_next);
    }
    // ---- lane roles (as k_eval_gram4) ------------------------------------------------------------------------------------
    double *fu_lo = Fl + (lane >> 2) * kG4Stride + (lane & 3), *fu_hi = Fl + (lane >> 2) * kG4Stride + ((lane & 3) ^ 2);
    const int li = lane >> 4, c16 = lane & 15, lb = c16 >> 2, lj = c16 & 3;
    const unsigned lds0 = lds_addr(Fl);
    const unsigned aN = lds0 + 8u * (unsigned)g4_elem(c16, li);
    const unsigned aR = lds0 + 8u * (unsigned)g4_elem((c16 + 4) & 15, li);
    const unsigned aB = lds0 + 8u * (unsigned)g4_elem(4 * g4_cgroup(lb, 2) + lj, li);
    const double rc0 = lj == 0 ? ccs[0] : lj == 1 ? ccs[1] : ccs[2];
    const double rc1 = lj == 0 ? ccs[3] : lj == 1 ? ccs[4] : ccs[5];
    const double rc2 = lj == 0 ? ccs[6] : lj == 1 ? ccs[7] : ccs[8];
    unsigned o1e = BAD, o1w = BAD, o2 = BAD, o3e = BAD, o4 = BAD, o5 = BAD, o6 = BAD, o7a = BAD, o7b = BAD;
    {
        const int b = lb, j = lj, i = li;
        auto W = [](int wcol, int row) { return 8u * (unsigned)(6 * wcol + row); };
        const int f1 = g4_wcol(4 + i), f1j = g4_wcol(4 + j), f2 = g4_wcol(8 + i), f2j = g4_wcol(8 + j), f3 = g4_wcol(12 + i);
        if (b == 0 && j < 3 && i < 3) { o1e = 8u * (unsigned)(6 * j + i); o3e = 8u * (unsigned)(6 * i + 3 + j); }
        if (b == 0 && j == 3 && i < 3) { o1w = W(kFR, i); o7a = 8u * (unsigned)i; }
        if (b == 1 && j < 3) o1w = W(f1, 3 + j);
        if (b == 0 && i < 3) o2 = W(f1j, i);
        if (b == 0 && i == 3 && j < 3) { o2 = W(kFR, 3 + j); o7b = 8u * (unsigned)(3 + j); }
        if (b == 3 && j < 3 && i < 3) o2 = W(f3, j);
        if (b == 0 && i < 3) o4 = W(f2j, i);
        if (b == 2 && j < 3) o4 = W(f2, 3 + j);
        if (b == 3 && j < 3 && i < 3) o4 = W(f3, 3 + j);
        if (b == 3 && i == 0 && j < 3) { o5 = W(9, j); o6 = W(9, 3 + j); }
        if (b == 0 && j == 3 && i < 3) o6 = W(7, i);
        if (b == 2 && i == 3 && j < 3) o6 = W(7, 3 + j);
    }
    if (lane < P.n_points) *reinterpret_cast<d2 *>(bxy + 2 * lane) = *reinterpret_cast<const d2 *>(P.board_xy + 2 * lane);
    (void)my_xy;
    const double *vcm = vcl + 32 * min(sv, M - 1);          // my view's constants
    auto VC = [&](int k) { return vcm[k]; };
    // the record of one view from its two accumulator sets (the epilogue of k_eval_gram4, same operations, same order)
    auto store_view = [&](const double (&aU)[3], const double (&aV)[3], unsigned slot) {
        int le = lane;
        asm volatile("" : "+v"(le));
        const int b = (le >> 2) & 3, i = le >> 4;
        const unsigned offW = 8u * (unsigned)kRecW * slot, offE = 8u * ((unsigned)kRecW * (unsigned)P.V + (unsigned)kRecE * slot);
        const double T0 = aU[0] + aV[0], T1 = aU[1] + aV[1], T2 = aU[2] + aV[2];
        const double tb0 = quad_tb(T0, rc0, rc1, rc2), tb1 = quad_tb(T1, rc0, rc1, rc2), tb2 = quad_tb(T2, rc0, rc1, rc2);
        const double tbU2 = quad_tb(aU[2], rc0, rc1, rc2);
        const bool split1 = b == 3 && i == 0, split2 = (le & 3) == 3 ? b == 0 : (b == 2 ? i == 3 : (b == 3 && i == 0));
        buf_store_f64(r_rec, o1e, offE, T0);
        buf_store_f64(r_rec, o1w, offW, b == 1 ? tb0 : T0);
        buf_store_f64(r_rec, o2, offW, split1 ? aU[1] : (i == 3 ? tb1 : T1));
        buf_store_f64(r_rec, o3e, offE, tb1);
        const double a4 = b != 0 ? tb2 : T2, u4 = b != 0 ? tbU2 : aU[2];
        buf_store_f64(r_rec, o4, offW, split2 ? u4 : a4);
        buf_store_f64(r_rec, o5, offW, T1 - aU[1]);
        buf_store_f64(r_rec, o6, offW, a4 - u4);
        const unsigned offG = 8u * ((unsigned)(kRecW + kRecE) * (unsigned)P.V + (unsigned)kRecG * slot);
        buf_store_f64(r_rec, o7a, offG, T0);
        buf_store_f64(r_rec, o7b, offG, tb1);
    };
    for (int vbase = vb; vbase < ve; vbase += BL) {
    const int vend = min(ve, vbase + BL);
    int m_cnt = 0, m_slot = 0;
    if (vbase == vb) { m_cnt = m_cnt0; m_slot = m_slot0; }
    else if (vbase + lane < vend) { m_cnt = P.view_count[vbase + lane]; m_slot = P.view_slot[vbase + lane]; }
    asm volatile("" : "+v"(m_cnt), "+v"(m_slot));
    {
        // the block's first pass: its observations were requested assuming full views (the head; the previous block's last pass
        // requests nothing across the block boundary) -- (re)request them with the real counts where that is not what is there
        bool full = vbase == vb;
        int o = off_next, my_o = 0, my_c = 0;
#pragma unroll
        for (int s = 0; s < M; ++s) {
            const int c = vbase + s < vend ? __builtin_amdgcn_readlane(m_cnt, s) : 0;          // (m_cnt: lane i = view vbase + i)
            full = full && (c == P.n_points || vbase + s >= vend);
            if (sv == s) { my_o = o; my_c = c; }
            o += c;
        }
        if (!full) {
            const unsigned ol = (jl < my_c && sv < M) ? 8u * (unsigned)(my_o + jl) : BAD;       // (the offset differs from lane to lane: all of it in the vector part)
            pf_u = buf_load_f64(r_u, ol, 0u); pf_v = buf_load_f64(r_v, ol, 0u);
        }
    }
    for (int view = vbase; view < vend; view += M) {
        const int nvp = min(M, vend - view);                 // views of this pass
        int cnt[M], my_cnt = 0;
#pragma unroll
        for (int s = 0; s < M; ++s) {
            cnt[s] = s < nvp ? __builtin_amdgcn_readlane(m_cnt, min(view - vbase + s, 63)) : 0;
            if (sv == s) my_cnt = cnt[s];
            off_next += cnt[s];
        }
        wave_lds_fence();                       // the previous pass has finished with the tile and the constants
#if TSCM_PRIO
        set_prio(3 - min(3, 8 * (view - vb) / max(1, ve - vb) % 4));      // priority by progress: see k_eval_gram
#endif
        // the views' constants into LDS
        if (lane < 32 * min(M, 2)) vcl[lane] = vc_pf0;
        if constexpr (M > 2) { if (lane < 32 * (M - 2)) vcl[64 + lane] = vc_pf1; }
        wave_lds_fence();
        const bool valid = sv < nvp && jl < my_cnt;
        double fv[16];
        auto PUT = [&](int c, double u, double v) { (c < 8 ? fu_lo : fu_hi)[4 * c] = u; fv[c] = v; };
        if (valid) {
            const double x = bxy[2 * jl], y = bxy[2 * jl + 1];
            constexpr int tcol[15] = { kG4Wb, kG4Wb + 1, kG4Wb + 2, kG4Tc, kG4Tc + 1, kG4Tc + 2, kG4Wc, kG4Wc + 1, kG4Wc + 2,
                                       kG4F, kG4One, kG4Xi, kG4Lam, kG4Al, kG4R };
            corner_geometry(x, y, pf_u, pf_v, VC, CC, [&](int gc, double u, double v) { PUT(tcol[gc], u, v); });
        } else if (prev_valid) {
#pragma unroll
            for (int c = 0; c < kTcols; ++c) (c < 8 ? fu_lo : fu_hi)[4 * c] = 0.0;
        }
        prev_valid = valid;
        {
            // the next pass of this block: its observations (the lane's own offset: the running sum of the counts in front of its
            // view) and its views' constants
            const int vn = view + M;
            int o = off_next, my_o = 0, my_c = 0;
#pragma unroll
            for (int s = 0; s < M; ++s) {
                const int c = vn + s < vend ? __builtin_amdgcn_readlane(m_cnt, min(vn + s - vbase, 63)) : 0;
                if (sv == s) { my_o = o; my_c = c; }
                o += c;
            }
            const unsigned ol = (jl < my_c && sv < M) ? 8u * (unsigned)(my_o + jl) : BAD;
            pf_u = buf_load_f64(r_u, ol, 0u); pf_v = buf_load_f64(r_v, ol, 0u);
            if (vn < vend) request_vc(vn, vend);
            else if (vend < ve) request_vc(vend, ve);          // (the next block's first pass: its constants need no counts)
        }
        wave_lds_fence();
        double accU[M][3];
#pragma unroll
        for (int s = 0; s < M; ++s) {
            accU[s][0] = accU[s][1] = accU[s][2] = 0.0;
            if (s < nvp) gram4_full<KS, false>(aN + 8u * (unsigned)(kG4Stride * KS * s), aR + 8u * (unsigned)(kG4Stride * KS * s), aB + 8u * (unsigned)(kG4Stride * KS * s), accU[s]);
        }
        wave_lds_fence();
        if (valid) {
#pragma unroll
            for (int c = 0; c < kTcols; ++c) (c < 8 ? fu_lo : fu_hi)[4 * c] = fv[c];
        }
        wave_lds_fence();
#pragma unroll
        for (int s = 0; s < M; ++s) {
            if (s < nvp) {
                double accV[3] = { 0.0, 0.0, 0.0 };
                gram4_full<KS, false>(aN + 8u * (unsigned)(kG4Stride * KS * s), aR + 8u * (unsigned)(kG4Stride * KS * s), aB + 8u * (unsigned)(kG4Stride * KS * s), accV);
#pragma unroll
                for (int q = 0; q < 3; ++q) { camU[q] += accU[s][q]; camV[q] += accV[q]; }
                store_view(accU[s], accV, (unsigned)__builtin_amdgcn_readlane(m_slot, view - vbase + s));
            }
        }
    }
    }   // block of BL views
    wave_lds_fence();
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int r = g4_old_col(4 * lb + li), c = g4_old_col(4 * g4_cgroup(lb, q) + lj);
        lds[16 * r + c] = camU[q]; lds[16 * c + r] = camU[q];
        lds[256 + 16 * r + c] = camV[q]; lds[256 + 16 * c + r] = camV[q];
    }
    __syncthreads();
    {
        const int t = threadIdx.x;
        const size_t st = lds_wave;
        double *part = S.campart + (size_t)512 * blockIdx.x;
        part[t] = (lds_all[t] + lds_all[st + t]) + (lds_all[2 * st + t] + lds_all[3 * st + t]);
        part[256 + t] = (lds_all[256 + t] + lds_all[st + 256 + t]) + (lds_all[2 * st + 256 + t] + lds_all[3 * st + 256 + t]);
    }
}

#include "tscm_eval_gram4s.h"      // (an experiment of round 6: the views of a chunk as one stream of k-steps -- measured slower, opt-in)

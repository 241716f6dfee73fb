// tscm_eval_gram4s.h -- k_eval_gram4s, an EXPERIMENT of round 6 (included by tscm_eval_gram4.h; tscm_debug_experiment(TSCM_EXPERIMENT_GRAM_STREAM, 1) in front of
// tscm_solver_create selects it, the default never does): built, bit-identical to the other Gram kernels, and SLOWER than one view
// per pass on every board measured (7 x 6: 61.8 us against 48.3; 8 x 6: 58.8 / 50.0; 10 x 7: 84.5 / 81.6; 11 x 8: 91.1 / 85.4; 9 x 6:
// 67.4 / 53.7 at config-4 size) -- the passes it saves cost less than what every pass pays for the per-lane view bookkeeping, the
// constants through LDS and the branch per k-step in the contraction (HISTORY A.7).
#pragma once

// ---------------------------------------------------------------------------------------------------------------------------
// k_eval_gram4s: the views of a chunk as ONE STREAM of k-steps (round 6) -- for boards whose passes leave lanes without a corner.
// A view of n corners is KSV = ceil(n / 4) k-steps; k_eval_gram4<KS, MULTI> gives every view its own pass(es): 7 x 6 (42 corners, 11
// k-steps) fills 44 of the 64 rows of its pass, 10 x 7 (70) fills 2 x 36 of 2 x 64 -- and the geometry (a third of the kernel)
// costs a pass the same whatever the number of live lanes.  Here a pass takes the NEXT k-steps of the chunk's stream, whichever
// views they belong to: 10 views of 11 k-steps are 8 passes instead of 10.  A pass touches two views at most (the rest of the one
// it starts in and the next; a third accumulator set spilled): lane l works for the view its k-step l / 4 falls in (a compare
// against the pass's segment boundary), the views' constants come through LDS and the observation offset from the running sums as in k_eval_gram4p, the
// contraction sweeps the 16 k-steps once per row half with the accumulator set chosen per k-step by a wave-uniform branch (the
// operand reads are the same unconditional software-pipelined sequence as gram4_steps'), and the accumulators of a view that is
// not finished at the end of the pass are carried into the next.  Every view contracts ITS k-steps in order into ITS accumulators:
// the same bits as the other Gram kernels.  The stream is cut at the 64-view metadata blocks.
// ---------------------------------------------------------------------------------------------------------------------------
// doubles of LDS per wave (tile + the constants of three views) and per workgroup (four waves + ONE copy of the board points)
constexpr int kG4sWave = 16 * kG4Stride + 2 * 32;
__host__ __device__ inline int eval_gram4s_lds_doubles(int n_points) { return 4 * kG4sWave + 2 * n_points; }

// one sweep over the 16 k-steps of the pass: k-steps [0, e0) -> acc0, [e0, 16) -> acc1 (rows past the pass's end are zero)
template <int D, int T = 0>
__device__ __forceinline__ void gram4_sweep(unsigned aN, unsigned aR, unsigned aB, int e0, double (&n)[16], double (&r)[16], double (&b)[16],
                                            double (&acc0)[3], double (&acc1)[3])
{
    if constexpr (T < 16) {
        if constexpr (T + D < 16) {
            n[T + D] = ds_read_f64<8 * kG4Stride * (T + D)>(aN);
            r[T + D] = ds_read_f64<8 * kG4Stride * (T + D)>(aR);
            b[T + D] = ds_read_f64<8 * kG4Stride * (T + D)>(aB);
        }
        constexpr int newer = 3 * (16 - 1 - T < D ? 16 - 1 - T : D);
        lgkm_wait<newer>(b[T]);          // (one wait per k-step: the three operands arrive together)
        asm volatile("" : "+v"(n[T]), "+v"(r[T]));
        if (T < e0) {
            acc0[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(n[T], n[T], acc0[0], 0, 0, 0);
            acc0[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(n[T], r[T], acc0[1], 0, 0, 0);
            acc0[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(n[T], b[T], acc0[2], 0, 0, 0);
        } else {
            acc1[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(n[T], n[T], acc1[0], 0, 0, 0);
            acc1[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(n[T], r[T], acc1[1], 0, 0, 0);
            acc1[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(n[T], b[T], acc1[2], 0, 0, 0);
        }
        gram4_sweep<D, T + 1>(aN, aR, aB, e0, n, r, b, acc0, acc1);
    }
}
__device__ __forceinline__ void gram4_sweep_full(unsigned aN, unsigned aR, unsigned aB, int e0, double (&acc0)[3], double (&acc1)[3])
{
    constexpr int D = 2;
    double n[16], r[16], b[16];
    gram4_prime<16, D>(aN, aR, aB, n, r, b);
    gram4_sweep<D>(aN, aR, aB, e0, n, r, b, acc0, acc1);
}

__global__ __launch_bounds__(256, 4) void k_eval_gram4s(DevProblem P, DevState S, int cand)
{
    constexpr int BL = 64;                          // views per metadata block
    KTL(0);
    const int ctrl_done = S.ctrl->done, ctrl_cur = S.ctrl->cur;
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int lds_wave = kG4sWave;
    double *lds = lds_all + (size_t)wave * lds_wave;
    double *Fl = lds;
    double *vcl = lds + 16 * kG4Stride;             // [2][32] the constants of the two views of the pass
    double *bxy = lds_all + 4 * kG4sWave;           // board points: one copy, written by every wave with the same values
    const int lane = threadIdx.x & 63;
    const int tk = lane >> 2;                       // my k-step of the pass
    const int KSV = P.g4s_ksv;                      // k-steps of a view: ceil(n / 4)
    const int chunk = blockIdx.x * 4 + wave;
    const v4i cd = *(const v4i __attribute__((address_space(4))) *)(const void *)(P.chunk_desc + chunk);
    const int cam = cd[0], vb = cd[1], ve = cd[2];
    double *const cc_buf[2] = { S.cconst[0], S.cconst[1] };
    double *const rec_buf[2] = { S.rec[0], S.rec[1] };
    double camU[3] = { 0.0, 0.0, 0.0 }, camV[3] = { 0.0, 0.0, 0.0 };
    for (int i = lane; i < 16 * kG4Stride; i += 64) Fl[i] = 0.0;
    bool prev_valid = false;
    double pf_u = 0.0, pf_v = 0.0;
    if (ctrl_done) return;
    const int tgt = cand ? (ctrl_cur ^ 1) : ctrl_cur;
    const __amdgpu_buffer_rsrc_t r_rec = make_rsrc(tgt ? rec_buf[1] : rec_buf[0], sizeof(double) * (size_t)kRec * P.V);
    const cptr4 ccs = (cptr4)((tgt ? cc_buf[1] : cc_buf[0]) + kCStride * cam);
    auto CC = [&](int k) { return ccs[k]; };
    const __amdgpu_buffer_rsrc_t r_vc = make_rsrc(S.vconst, sizeof(double) * (size_t)kVStride * P.V);
    const __amdgpu_buffer_rsrc_t r_u = make_rsrc(P.obs_u, sizeof(double) * (size_t)P.N), r_v = make_rsrc(P.obs_v, sizeof(double) * (size_t)P.N);
    int off_first = cd[3];                          // observation offset of view `vfirst` (below)
    constexpr unsigned BAD = 0xffffe000u;
    // the constants of the two views of a pass: lane l fetches double l % 32 of view l / 32
    double vc_pf0 = 0.0;
    auto request_vc = [&](int view0, int vend_) {
        const int s0 = lane >> 5, k = lane & 31;
        vc_pf0 = buf_load_f64(r_vc, (view0 + s0 < vend_ && k < kVConst) ? 8u * (unsigned)(kVStride * s0 + k) : BAD, 8u * (unsigned)kVStride * (unsigned)view0);
    };
    int m_cnt0 = 0, m_slot0 = 0;
    if (vb + lane < min(ve, vb + BL)) { m_cnt0 = P.view_count[vb + lane]; m_slot0 = P.view_slot[vb + lane]; }
    request_vc(vb, ve);
    // lane roles as an MFMA lane / record offsets / R_c: as k_eval_gram4
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
    for (int j = lane; j < P.n_points; j += 64) *reinterpret_cast<d2 *>(bxy + 2 * j) = *reinterpret_cast<const d2 *>(P.board_xy + 2 * j);
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
    const int nvb = vend - vbase;
    auto CNT = [&](int v) { return v < nvb ? __builtin_amdgcn_readlane(m_cnt, min(v, 63)) : 0; };      // corners of view v of the block
    // my place in a pass that starts kin k-steps into view vf (of the block), whose observations start at offset of0:
    // view (0..2 relative to vf), first observation of my view, its corner count, my corner
    // (a pass touches TWO views at most -- the rest of the one it starts in and the next: with a third accumulator set the kernel
    // spilled 120 bytes per lane inside the view loop and ran 1.8 x slower than one view per pass; a pass that ends with its second
    // view leaves the k-steps behind it empty)
    struct Place { int sv, off, cnt, j; };
    auto place = [&](int vf, int kin, int of0) {
        const int e0 = KSV - kin, e1 = e0 + KSV;
        const int c0 = CNT(vf), c1 = CNT(vf + 1);
        Place p;
        p.sv = tk >= e0 ? 1 : 0;
        const int kk = p.sv == 0 ? kin + tk : tk - e0;
        p.j = 4 * kk + (lane & 3);
        p.off = p.sv == 0 ? of0 : of0 + c0;
        p.cnt = tk >= e1 ? 0 : (p.sv == 0 ? c0 : c1);
        return p;
    };
    int vfirst = 0, kin = 0;                        // the pass starts kin k-steps into view vfirst of the block
    {
        // the block's first pass: its observations (the previous block's last pass requests nothing across the boundary)
        const Place p = place(0, 0, off_first);
        const unsigned ol = p.j < p.cnt ? 8u * (unsigned)(p.off + p.j) : BAD;
        pf_u = buf_load_f64(r_u, ol, 0u); pf_v = buf_load_f64(r_v, ol, 0u);
    }
    double carU[3] = { 0.0, 0.0, 0.0 }, carV[3] = { 0.0, 0.0, 0.0 };       // accumulators of the view the pass starts inside of
    while (vfirst < nvb) {
        const Place me = place(vfirst, kin, off_first);
        const int e0 = KSV - kin, e1 = e0 + KSV;     // tile k-steps [0, e0): view vfirst, [e0, min(e1, 16)): view vfirst + 1
        wave_lds_fence();                           // the previous pass has finished with the tile and the constants
#if TSCM_PRIO
        set_prio(3 - min(3, 8 * (vbase + vfirst - vb) / max(1, ve - vb) % 4));      // priority by progress: see k_eval_gram
#endif
        vcl[lane] = vc_pf0;
        wave_lds_fence();
        const bool valid = me.j < me.cnt;           // (views past the block's end have no corners)
        const double *vcm = vcl + 32 * me.sv;
        auto VC = [&](int k) { return vcm[k]; };
        double fv[16];
        auto PUT = [&](int c, double u, double v) { (c < 8 ? fu_lo : fu_hi)[4 * c] = u; fv[c] = v; };
        if (valid) {
            const double x = bxy[2 * me.j], y = bxy[2 * me.j + 1];
            constexpr int tcol[15] = { kG4Wb, kG4Wb + 1, kG4Wb + 2, kG4Tc, kG4Tc + 1, kG4Tc + 2, kG4Wc, kG4Wc + 1, kG4Wc + 2,
                                       kG4F, kG4One, kG4Xi, kG4Lam, kG4Al, kG4R };
            corner_geometry(x, y, pf_u, pf_v, VC, CC, [&](int gc, double u, double v) { PUT(tcol[gc], u, v); });
        } else if (prev_valid) {
#pragma unroll
            for (int c = 0; c < kTcols; ++c) (c < 8 ? fu_lo : fu_hi)[4 * c] = 0.0;
        }
        prev_valid = valid;
        // where the next pass starts; its observations and its views' constants
        int vnext = vfirst, kin_next = kin + min(16, e1), off_next = off_first;
        while (kin_next >= KSV) { kin_next -= KSV; off_next += CNT(vnext); ++vnext; }
        {
            const bool more = vnext < nvb;
            const Place p = place(vnext, kin_next, off_next);
            const unsigned ol = (more && p.j < p.cnt) ? 8u * (unsigned)(p.off + p.j) : BAD;
            pf_u = buf_load_f64(r_u, ol, 0u); pf_v = buf_load_f64(r_v, ol, 0u);
            if (more) request_vc(vbase + vnext, vend);
            else if (vend < ve) request_vc(vend, ve);            // (the next block's first pass)
        }
        wave_lds_fence();
        double aU0[3] = { carU[0], carU[1], carU[2] }, aU1[3] = { 0.0, 0.0, 0.0 };
        gram4_sweep_full(aN, aR, aB, e0, aU0, aU1);
        wave_lds_fence();
        if (valid) {
#pragma unroll
            for (int c = 0; c < kTcols; ++c) (c < 8 ? fu_lo : fu_hi)[4 * c] = fv[c];
        }
        wave_lds_fence();
        double aV0[3] = { carV[0], carV[1], carV[2] }, aV1[3] = { 0.0, 0.0, 0.0 };
        gram4_sweep_full(aN, aR, aB, e0, aV0, aV1);
        // a view whose last k-step lies in this pass is finished: its record; otherwise its accumulators are carried
        auto finish = [&](int s, int end_k, const double (&aU)[3], const double (&aV)[3]) {
            if (vfirst + s >= nvb) return;
            if (end_k <= 16) {
#pragma unroll
                for (int q = 0; q < 3; ++q) { camU[q] += aU[q]; camV[q] += aV[q]; }
                store_view(aU, aV, (unsigned)__builtin_amdgcn_readlane(m_slot, min(vfirst + s, 63)));
            } else {
#pragma unroll
                for (int q = 0; q < 3; ++q) { carU[q] = aU[q]; carV[q] = aV[q]; }
            }
        };
#pragma unroll
        for (int q = 0; q < 3; ++q) { carU[q] = 0.0; carV[q] = 0.0; }
        finish(0, e0, aU0, aV0);
        if (e0 < 16) finish(1, e1, aU1, aV1);
        vfirst = vnext; kin = kin_next; off_first = off_next;
    }
    // (off_first is the observation offset of the next block's first view now: vnext = nvb at the end of the stream)
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

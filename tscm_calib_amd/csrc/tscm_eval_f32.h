// tscm_eval_f32.h -- mixed-precision variant of k_eval_gram (north_star's "fp32 Jacobian" tier).
//
// Same data flow, tile columns, record layout and fp64 epilogue as k_eval_gram (tscm_kernels.h);
// what changes is the arithmetic type of the Jacobian:
//   fp64  board -> camera transform, triple-sphere projection, residual and the cost r^T r
//   fp32  every derivative (the 2 x 15 tile row of a corner), computed two at a time (u-row, v-row)
//         with packed fp32 math, staged in LDS as (u, v) PAIRS and contracted with
//         v_mfma_f32_16x16x4_f32; one view's Gram (<= 2 x 64 rows) accumulates in fp32, is converted
//         once and everything downstream (records, camera tiles, Schur elimination, reduced solve)
//         stays fp64.
// The fp32 MFMA returns rows 4*(lane>>4)+reg per lane, the fp64 one rows (lane>>4)+4*reg; feeding
// the A operand with tile column pi(i) = (i>>2) + 4*(i&3) makes the fp32 result land exactly where
// the fp64 epilogue expects it.
//
// On gfx950 the fp32 (and fp64) MFMAs run at the VECTOR rate and do not overlap with another wave's VALU work on the
// same SIMD (tools/ubench_overlap.hip: MFMA waves + FMA waves on one SIMD take the SUM of their times, in every
// combination of f32 / f64): the kernel's time is the sum of its instruction cycles, and the layout below is about
// spending none on moving operands (round 4):
//   * tile T[column c][row position p] of float2 (u-row entry, v-row entry): a corner writes its 15 entries as
//     ds_write_b64 -- both row halves at once, no v-row copy pass, no v-rows waiting in registers;
//   * the contraction index of the MFMA is free to permute: with K k-steps per pass, corner j sits at position
//     KB (j / K) + j % K, so that lane (column, kq) finds ITS rows of all K k-steps -- positions KB kq .. KB kq + K - 1 --
//     adjacent: K / 2 ds_read_b128 per operand and view instead of 2 x K ds_read_b32.  Boards of up to 56 corners
//     (K = KB = 14 at compile time): the position is the lane itself, 58 positions per column -- 116 dwords = 4 * 29, the 16
//     columns of one kq cover the 64 banks, and a workgroup's four tiles + ONE copy of the board points are 30.6 KB.
//     Every other board (round 6): the pass plan of k_eval_gram4 (g4_plan) -- K = KS k-steps at compile time, exactly
//     ceil(n / 4) of them up to 56 corners, ceil(n / 56) balanced passes above (11 x 8 = 2 x 44: 22 k-steps per view where the
//     64-row passes of rounds 1-5 ran 32); KB = K rounded up to even (16-byte block starts), 4 KB + 2 positions per column;
//   * the u- and v-row MFMAs alternate: two independent accumulators, so the 40-cycle dependent latency of the
//     instruction (32 to issue) is not paid 2 K times per view.
// Measured (round 4, config 4, same box, alternating): 46.0 -> 43.5 us per launch; five workgroups per CU (86 VGPRs and
// the smaller tile allow it: make variant EXTRA="-DTSCM_F32_WGS=5 -DTSCM_EVAL_WAVES=5") give nothing -- the SIMD's one
// pipe is busy, not waiting.
#pragma once
// (included from tscm_kernels.h inside namespace tscm)

// tile geometry: KB row positions per kq block (corner j at position KB (j / K) + j % K: the identity when KB = K),
// P2 positions per tile column with 2 P2 = 4 * odd dwords (the 16 columns of an operand fetch cover the 64 banks)
template <int KS> struct F32Tile {
    static constexpr int KB = KS + (KS & 1);
    static constexpr int P2 = 4 * KB + 2;
    static constexpr int kDoubles = 16 * P2 > 512 ? 16 * P2 : 512;        // per wave; at least the 512-double camera-tile exchange
    static_assert(4 * KB <= P2 && (2 * P2) % 8 == 4 && (KB * 8) % 16 == 0, "block starts are 16-byte aligned, columns 4 * odd dwords apart");
};

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

// dynamic LDS of a workgroup, in doubles: the four waves' tiles, then ONE copy of the board points
__host__ __device__ inline int eval_f32_lds_doubles(int n_points, int ks)
{
    const int kb = ks + (ks & 1), tile = 16 * (4 * kb + 2);
    return 4 * (tile > 512 ? tile : 512) + 2 * n_points;
}

#ifndef TSCM_F32_WGS
#define TSCM_F32_WGS 4
#endif

// KS: k-steps of a pass; MULTI: boards of more than 56 corners, P.g4_per corners per pass (the pass plan of k_eval_gram4: g4_plan)
template <int KS, bool MULTI>
__global__ __launch_bounds__(256, TSCM_F32_WGS) void k_eval_gram_f32(DevProblem P, DevState S, int cand)
{
    static_assert(KS >= 1 && KS <= kG4MaxKS, "a pass holds at most 64 rows");
    // the control block is read together with the static chunk tables (one memory round trip, not two);
    // the early exit is taken right before the first view
    const int ctrl_done = S.ctrl->done, ctrl_cur = S.ctrl->cur;
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int kP2 = F32Tile<KS>::P2, KB = F32Tile<KS>::KB, kTile32 = F32Tile<KS>::kDoubles;
    constexpr int lds_wave = kTile32;
    double *lds = lds_all + (size_t)wave * lds_wave;
    f2 *T2 = reinterpret_cast<f2 *>(lds);                   // [16][kP2] (u, v) pairs
    double *bxy = lds_all + 4 * kTile32;                    // board points: one copy, written by every wave with the same values
    const int lane = threadIdx.x & 63;
    const int chunk = blockIdx.x * 4 + wave;
    // head in two dependent round trips, as in k_eval_gram4 (round 5): control block + 16-byte chunk descriptor + the first 128
    // board coordinates; then everything that hangs on the descriptor at once (below)
    const v4i cd = *(const v4i __attribute__((address_space(4))) *)(const void *)(P.chunk_desc + chunk);
    const int cam = cd[0], vb = cd[1], ve = cd[2];
    const int n2 = 2 * P.n_points;
    const double bx0 = P.board_xy[min(lane, n2 - 1)], bx1 = P.board_xy[min(lane + 64, n2 - 1)];
    // wave-uniform constants through the constant address space: scalar loads into SGPR operands; the
    // float copies k_view_prep stores behind the doubles feed the fp32 derivative math directly
    typedef const float __attribute__((address_space(4))) *fptr4;
    double *const cc_buf[2] = { S.cconst[0], S.cconst[1] };
    double *const rec_buf[2] = { S.rec[0], S.rec[1] };
    const int col = lane & 15, kq = lane >> 4;
    d4 camU = { 0.0, 0.0, 0.0, 0.0 }, camV = { 0.0, 0.0, 0.0, 0.0 };
    double rr = 0.0;                               // this lane's share of r^T r, fp64
    for (int i = lane; i < kTile32; i += 64) lds[i] = 0.0;  // all positions incl. the all-zero 16th tile column and the ones no corner maps to
    int prev_nv = 0;
    double pf_u = 0.0, pf_v = 0.0;
    int warm = 0;
    // k-steps of a pass (kernel-uniform) and this lane's row position as a corner; as an MFMA lane (column, kq): the rows
    // of its K k-steps are positions KB kq .. KB kq + K - 1 of tile column col (B operand) and pi(col) (A operand)
    constexpr int K = KS;
    const int per = MULTI ? P.g4_per : 4 * KS;                            // corners of a pass
    f2 *fw = T2 + (KB == KS ? lane : KB * (lane / K) + lane % K);         // (lanes >= 4 K never hold a corner)
    typedef const f4 __attribute__((address_space(3))) *lds_f4;
    const lds_f4 pB = (lds_f4)(T2 + col * kP2 + KB * kq);
    const lds_f4 pA = (lds_f4)(T2 + ((col >> 2) + 4 * (col & 3)) * kP2 + KB * kq);
    if (ctrl_done) return;
    const int tgt = cand ? (ctrl_cur ^ 1) : ctrl_cur;
    const __amdgpu_buffer_rsrc_t r_rec = make_rsrc(tgt ? rec_buf[1] : rec_buf[0], sizeof(double) * (size_t)kRec * P.V);
    const cptr4 cc = (cptr4)((tgt ? cc_buf[1] : cc_buf[0]) + kCStride * cam);
    const fptr4 cf = (fptr4)((tgt ? cc_buf[1] : cc_buf[0]) + kCStride * cam + kCConst);
    const RecLane rl = rec_lane(lane, (unsigned)P.V);
    const __amdgpu_buffer_rsrc_t r_vc = make_rsrc(S.vconst, sizeof(double) * (size_t)kVStride * P.V);
    const __amdgpu_buffer_rsrc_t r_u = make_rsrc(P.obs_u, sizeof(double) * (size_t)P.N), r_v = make_rsrc(P.obs_v, sizeof(double) * (size_t)P.N);
    // Per-view metadata (corner count, record slot) of a block of <= 64 views sits in lane registers and is
    // read with v_readlane; the observations of a camera's views are contiguous, so the offset is a running
    // sum.  No dependent global load -- and therefore no in-order vmcnt wait behind the previous view's
    // record stores -- is left inside the view loop.
    int off_next = cd[3];
    int m_cnt0 = 0, m_slot0 = 0;
    if (vb + lane < min(ve, vb + 64)) { m_cnt0 = P.view_count[vb + lane]; m_slot0 = P.view_slot[vb + lane]; }
    if (vb < ve) { pf_u = buf_load_f64(r_u, 8u * lane, 8u * (unsigned)off_next); pf_v = buf_load_f64(r_v, 8u * lane, 8u * (unsigned)off_next); }
    {
        // one dword of every 64-byte line of the first view's and the camera's constants (doubles and floats) through the scalar cache
        typedef const int __attribute__((address_space(4))) *cptr4i;
        const cptr4i v0 = (cptr4i)(S.vconst + (size_t)kVStride * min(vb, P.V - 1)), c0 = (cptr4i)cc;
        const int k0 = v0[0], k1 = v0[16], k2 = v0[32], k3 = v0[48], k4 = v0[64], k5 = v0[80], k6 = c0[16], k7 = c0[48], k8 = c0[80], k9 = c0[112], k10 = c0[128];
        asm volatile("" :: "s"(k0), "s"(k1), "s"(k2), "s"(k3), "s"(k4), "s"(k5), "s"(k6), "s"(k7), "s"(k8), "s"(k9), "s"(k10));
    }
    // the board coordinates: every wave writes the same values into the workgroup's one copy
    if (lane < n2) bxy[lane] = bx0;
    if (lane + 64 < n2) bxy[lane + 64] = bx1;
    for (int i = lane + 128; i < n2; i += 64) bxy[i] = P.board_xy[i];
    for (int vbase = vb; vbase < ve; vbase += 64) {
    const int vend = min(ve, vbase + 64);
    int m_cnt = 0, m_slot = 0;
    if (vbase == vb) { m_cnt = m_cnt0; m_slot = m_slot0; }
    else {
        if (vbase + lane < vend) { m_cnt = P.view_count[vbase + lane]; m_slot = P.view_slot[vbase + lane]; }
        pf_u = buf_load_f64(r_u, 8u * lane, 8u * (unsigned)off_next); pf_v = buf_load_f64(r_v, 8u * lane, 8u * (unsigned)off_next);
    }
    asm volatile("" : "+v"(m_cnt), "+v"(m_slot));       // the loads complete here, outside the view loop
    for (int view = vbase; view < vend; ++view) {
        const int cnt = __builtin_amdgcn_readlane(m_cnt, view - vbase);
        const int off = off_next;
        off_next = off + cnt;
        wave_lds_fence();                       // previous view's epilogue has finished with LDS
#if TSCM_PRIO
        set_prio(3 - min(3, 8 * (view - vb) / max(1, ve - vb) % 4));      // priority by progress: see k_eval_gram
#endif
        const cptr4 cst = (cptr4)(S.vconst + (size_t)kVStride * view);                 // this view's constants, doubles
        const fptr4 cs = (fptr4)(S.vconst + (size_t)kVStride * view + kVFloatOff);     // ... and floats
        d4 accU = { 0.0, 0.0, 0.0, 0.0 }, accV = { 0.0, 0.0, 0.0, 0.0 };
        for (int c0 = 0; c0 < cnt; c0 += per) {
            const int j = c0 + lane;
            const bool valid = lane < (MULTI ? min(per, cnt - c0) : cnt);
            auto PUT = [&](int c, float u, float v) { fw[c * kP2] = f2{ u, v }; };
            if (valid) {
                const double x = bxy[2 * j], y = bxy[2 * j + 1];
                const double ou = pf_u, ov = pf_v;
                // ---- fp64: board -> world -> camera, triple sphere, residual (multi_calib.h:158-193) ----
                const double X = fma(x, cst[0], fma(y, cst[3], cst[6]));
                const double Y = fma(x, cst[1], fma(y, cst[4], cst[7]));
                const double Z = fma(x, cst[2], fma(y, cst[5], cst[8]));
                const double rho2 = X * X + Y * Y;
                const double s1 = rho2 + Z * Z;
                const double id1 = fast_rsqrt(s1), d1 = s1 * id1;
                const double z1 = Z + cc[43] * d1;
                const double s2 = rho2 + z1 * z1;
                const double id2 = fast_rsqrt(s2), d2 = s2 * id2;
                const double z2 = z1 + cc[44] * d2;
                const double s3 = rho2 + z2 * z2;
                const double id3 = fast_rsqrt(s3), d3 = s3 * id3;
                const double ik = fast_rcp(z2 + cc[45] * d3);
                const double mx = X * ik, my = Y * ik;
                const double ru = ou - (cc[39] * mx + cc[41]);
                const double rv = ov - (cc[40] * my + cc[42]);
                rr += ru * ru + rv * rv;
                // ---- fp32: derivatives, (u-row, v-row) pairs ---------------------------------------------
                const float xf = (float)x, yf = (float)y;
                const float Xf = (float)X, Yf = (float)Y, Zf = (float)Z, z1f = (float)z1, z2f = (float)z2;
                const float i1 = (float)id1, i2 = (float)id2, i3 = (float)id3, ikf = (float)ik;
                const float e1 = (float)d1, e2 = (float)d2, e3 = (float)d3;
                const float mxf = (float)mx, myf = (float)my;
                // Q' = P_c - t_c (minus w x Q in the small-angle branch of the camera rotation): corner_geometry
                float Q0 = (float)(X - cc[9]), Q1 = (float)(Y - cc[10]), Q2 = (float)(Z - cc[11]);
                if (cc[24] != 0.0) {
                    const float w0 = cf[21], w1 = cf[22], w2 = cf[23];
                    const float s0 = w1 * Q2 - w2 * Q1, s1 = w2 * Q0 - w0 * Q2, s2 = w0 * Q1 - w1 * Q0;
                    Q0 -= s0; Q1 -= s1; Q2 -= s2;
                }
                const float xi = cf[43], lam = cf[44], beta = cf[45];
                const float c1 = 1.f + xi * Zf * i1;
                const float c2 = 1.f + lam * z1f * i2;
                const float c3 = 1.f + beta * z2f * i3;
                const float q = beta * i3 + c3 * (lam * i2 + c2 * xi * i1);
                const float kz = c1 * c2 * c3;
                const f2 fk = { cf[39] * ikf, cf[40] * ikf };              // (fx/k, fy/k)
                const f2 HM = { fk.x * mxf, fk.y * myf };                  // (fx mx / k, fy my / k)
                const f2 N0 = HM * (Xf * q) - (f2){ fk.x, 0.f };           // -d(u,v)/dX
                const f2 N1 = HM * (Yf * q) - (f2){ 0.f, fk.y };           // -d(u,v)/dY
                const f2 N2 = HM * kz;                                     // -d(u,v)/dZ
                PUT(tc_tc(0), N0.x, N0.y);
                PUT(tc_tc(1), N1.x, N1.y);
                PUT(tc_tc(2), N2.x, N2.y);
#pragma unroll
                for (int kk = 0; kk < 3; ++kk) {                           // w_b: -A (x e_k0 + y e_k1)
                    const float h0 = xf * cs[9 + 6 * kk] + yf * cs[12 + 6 * kk];
                    const float h1 = xf * cs[10 + 6 * kk] + yf * cs[13 + 6 * kk];
                    const float h2 = xf * cs[11 + 6 * kk] + yf * cs[14 + 6 * kk];
                    const f2 w = N0 * h0 + N1 * h1 + N2 * h2;
                    PUT((kTcWb + kk), w.x, w.y);
                }
                {                                                          // w_c: -A (dR_c/dw_k Pw) = a_k . (Q' x n), both rows at once
                    const f2 c0 = N2 * Q1 - N1 * Q2, c1 = N0 * Q2 - N2 * Q0, c2 = N1 * Q0 - N0 * Q1;
#pragma unroll
                    for (int kk = 0; kk < 3; ++kk) {
                        const f2 w = c0 * cf[12 + 3 * kk] + c1 * cf[13 + 3 * kk] + c2 * cf[14 + 3 * kk];
                        PUT((kTcWc + kk), w.x, w.y);
                    }
                }
                PUT(kTcF, -mxf, -myf);
                PUT(kTcOne, -1.f, -1.f);
                const float kxi = c3 * c2 * e1, klam = c3 * e2, kal = e3 * cf[46];
                const f2 a = HM * kxi, b = HM * klam, c = HM * kal;
                PUT(kTcXi, a.x, a.y);
                PUT(kTcLam, b.x, b.y);
                PUT(kTcAl, c.x, c.y);
                PUT(kTcR, (float)ru, (float)rv);
            } else if (lane < prev_nv) {
#pragma unroll
                for (int c = 0; c < kTcols; ++c) PUT(c, 0.f, 0.f);
            }
            {
                // Prefetch of the next view, issued once the current view's observations have been consumed: the
                // loads reuse the same registers (no copy that would have to wait for them), and everything
                // between here and their use at the top of the next view is four unconditional stores.
                // Always issued (the block's last view re-reads itself; lanes past the corner count read past
                // the end of the buffer, i.e. zero): unconditional loads keep the vmcnt bookkeeping exact.
                const int vn = min(view + 1, vend - 1);
                const int cn = view + 1 < vend ? __builtin_amdgcn_readlane(m_cnt, vn - vbase) : 0;
                // pull the next view's 384-byte constant record into the L2 (one tracked load, lanes 0..5)
                if (!MULTI || c0 == 0) warm = __builtin_amdgcn_raw_buffer_load_b32(r_vc, lane < 6 ? 64 * lane : (int)0xffffe000u, (int)(8u * (unsigned)kVStride * (unsigned)vn), 0);
                // MULTI: the next pass of this view, or the first pass of the next one
                const bool more = MULTI && c0 + per < cnt;
                const int ncnt = MULTI ? (more ? min(per, cnt - c0 - per) : min(per, cn)) : cn;
                const unsigned noff = more ? (unsigned)(off + c0 + per) : (unsigned)off_next;
                pf_u = buf_load_f64(r_u, lane < ncnt ? 8u * lane : 0xffffe000u, 8u * noff);
                pf_v = buf_load_f64(r_v, lane < ncnt ? 8u * lane : 0xffffe000u, 8u * noff);
            }
            wave_lds_fence();
            const int nv = min(per, cnt - c0);
            prev_nv = nv;
            f4 aU = { 0.f, 0.f, 0.f, 0.f }, aV = { 0.f, 0.f, 0.f, 0.f };
            {
                // rows of lanes without a corner are zero: all K k-steps run whatever the view's corner count
                f4 A[4] = {}, B[4] = {};
#pragma unroll
                for (int m = 0; m < 4; ++m) if (2 * m < K) { A[m] = pA[m]; B[m] = pB[m]; }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        const f4 a = A[m], b = B[m];
                        if (h == 0 && 8 + 2 * m < K) { A[m] = pA[4 + m]; B[m] = pB[4 + m]; }      // the second half's operands, behind their last use
                        const int t = 8 * h + 2 * m;
                        if (t < K) {
                            aU = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, aU, 0, 0, 0);
                            aV = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, aV, 0, 0, 0);
                        }
                        if (t + 1 < K) {
                            aU = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, aU, 0, 0, 0);
                            aV = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, aV, 0, 0, 0);
                        }
                    }
                }
            }
            wave_lds_fence();
            // one pass (<= 64 corners) of fp32 accumulation, then fp64
#pragma unroll
            for (int r = 0; r < 4; ++r) { accU[r] += (double)aU[r]; accV[r] += (double)aV[r]; }
            if (!MULTI) break;               // (n_points <= 4 KS: one pass)
        }
        asm volatile("" :: "v"(warm));       // the warming load retires here, before this view's record stores
        camU += accU; camV += accV;
        store_view_record(r_rec, lane, accU, accV, cc, (unsigned)__builtin_amdgcn_readlane(m_slot, view - vbase), rl);
    }
    }   // block of <= 64 views
    // r^T r of the camera tile (entry [14][14] = lane (col 14, kq 2), reg 3) comes from the fp64 sum
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) rr += __shfl_xor(rr, o);
    if (col == 14 && kq == 2) { camU[3] = rr; camV[3] = 0.0; }
    wave_lds_fence();
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) { lds[(kq + 4 * rg) * 16 + col] = camU[rg]; lds[256 + (kq + 4 * rg) * 16 + col] = camV[rg]; }
    __syncthreads();
    {
        const int t = threadIdx.x;
        const size_t st = lds_wave;
        double *part = S.campart + (size_t)512 * blockIdx.x;
        part[t] = (lds_all[t] + lds_all[st + t]) + (lds_all[2 * st + t] + lds_all[3 * st + t]);
        part[256 + t] = (lds_all[256 + t] + lds_all[st + 256 + t]) + (lds_all[2 * st + 256 + t] + lds_all[3 * st + 256 + t]);
    }
}


// tscm_eval_f32.h -- mixed-precision variant of k_eval_gram (north_star's "fp32 Jacobian" tier).
//
// Same data flow, tile columns, record layout and fp64 epilogue as k_eval_gram (tscm_kernels.h);
// what changes is the arithmetic type of the Jacobian:
//   fp64  board -> camera transform, triple-sphere projection, residual and the cost r^T r
//   fp32  every derivative (the 2 x 15 tile row of a corner), computed two at a time (u-row, v-row)
//         with packed fp32 math, staged in a float LDS tile and contracted with
//         v_mfma_f32_16x16x4_f32 (half the issue cycles of the fp64 MFMA); one view's Gram
//         (<= 2 x 64 rows) accumulates in fp32, is converted once and everything downstream
//         (records, camera tiles, Schur elimination, reduced solve) stays fp64.
// The fp32 MFMA returns rows 4*(lane>>4)+reg per lane, the fp64 one rows (lane>>4)+4*reg; feeding
// the A operand with tile column pi(i) = (i>>2) + 4*(i&3) makes the fp32 result land exactly where
// the fp64 epilogue expects it.
#pragma once
// (included from tscm_kernels.h inside namespace tscm)

constexpr int kRP32 = 68;          // float pitch of the fp32 tile: 68 = 4 (mod 64) -> 16 columns x 4 k-rows hit 64 distinct banks
constexpr int kTile32 = 16 * kRP32 * 4 / 8;   // the tile, in doubles (544): also covers the 512-double camera-tile exchange

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

__host__ __device__ inline int eval_f32_lds_doubles(int n_points) { return kTile32 + 2 * n_points; }

__global__ __launch_bounds__(256, 4) void k_eval_gram_f32(DevProblem P, DevState S, int cand)
{
    // the control block is read together with the static chunk tables (one memory round trip, not two);
    // the early exit is taken right before the first view
    const int ctrl_done = S.ctrl->done, ctrl_cur = S.ctrl->cur;
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lds_wave = eval_f32_lds_doubles(P.n_points);
    double *lds = lds_all + (size_t)wave * lds_wave;
    float *Fl = reinterpret_cast<float *>(lds);            // [16][kRP32]: the u-rows, then the v-rows
    double *bxy = lds + kTile32;                            // board points
    constexpr int RP = kRP32;
    const int lane = threadIdx.x & 63;
    const int chunk = blockIdx.x * 4 + wave;
    const int cam = P.chunk_cam[chunk];
    for (int i = lane; i < 2 * P.n_points; i += 64) bxy[i] = P.board_xy[i];
    // wave-uniform constants through the constant address space: scalar loads into SGPR operands; the
    // float copies k_view_prep stores behind the doubles feed the fp32 derivative math directly
    typedef const float __attribute__((address_space(4))) *fptr4;
    const int vb = P.chunk_vb[chunk], ve = P.chunk_ve[chunk];
    const int col = lane & 15, kq = lane >> 4;
    d4 camU = { 0.0, 0.0, 0.0, 0.0 }, camV = { 0.0, 0.0, 0.0, 0.0 };
    double rr = 0.0;                               // this lane's share of r^T r, fp64
#pragma unroll
    for (int c = 0; c < 16; ++c) Fl[c * RP + lane] = 0.f;  // all 64 rows incl. the all-zero 16th tile column
    int prev_nv = 0;
    double pf_u = 0.0, pf_v = 0.0;
    int warm = 0;
    const float *fpB = Fl + col * RP + kq;                              // B operand: tile column col
    const float *fpA = Fl + ((col >> 2) + 4 * (col & 3)) * RP + kq;     // A operand: tile column pi(col)
    if (ctrl_done) return;
    const int tgt = cand ? (ctrl_cur ^ 1) : ctrl_cur;
    const __amdgpu_buffer_rsrc_t r_rec = make_rsrc(S.rec[tgt], sizeof(double) * (size_t)kRec * P.V);
    const cptr4 cc = (cptr4)(S.cconst[tgt] + kCStride * cam);
    const fptr4 cf = (fptr4)(S.cconst[tgt] + kCStride * cam + kCConst);
    const RecLane rl = rec_lane(lane, (unsigned)P.V);
    const __amdgpu_buffer_rsrc_t r_vc = make_rsrc(S.vconst, sizeof(double) * (size_t)kVStride * P.V);
    const __amdgpu_buffer_rsrc_t r_u = make_rsrc(P.obs_u, sizeof(double) * (size_t)P.N), r_v = make_rsrc(P.obs_v, sizeof(double) * (size_t)P.N);
    // Per-view metadata (corner count, record slot) of a block of <= 64 views sits in lane registers and is
    // read with v_readlane; the observations of a camera's views are contiguous, so the offset is a running
    // sum.  No dependent global load -- and therefore no in-order vmcnt wait behind the previous view's
    // record stores -- is left inside the view loop.
    int off_next = vb < ve ? P.view_obs[vb] : 0;
    for (int vbase = vb; vbase < ve; vbase += 64) {
    const int vend = min(ve, vbase + 64);
    int m_cnt = 0, m_slot = 0;
    if (vbase + lane < vend) { m_cnt = P.view_count[vbase + lane]; m_slot = P.view_slot[vbase + lane]; }
    asm volatile("" : "+v"(m_cnt), "+v"(m_slot));       // the loads complete here, outside the view loop
    {
        const int c0n = __builtin_amdgcn_readlane(m_cnt, 0);
        if (lane < c0n) { pf_u = buf_load_f64(r_u, 8u * lane, 8u * (unsigned)off_next); pf_v = buf_load_f64(r_v, 8u * lane, 8u * (unsigned)off_next); }
    }
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
        for (int c0 = 0; c0 < cnt; c0 += 64) {
            const int j = c0 + lane;
            const bool valid = j < cnt;
            float *fu = Fl + lane;
            float fv[kTcols];
            if (valid) {
                const double x = bxy[2 * j], y = bxy[2 * j + 1];
                const double ou = c0 ? buf_load_f64(r_u, 8u * j, 8u * (unsigned)off) : pf_u, ov = c0 ? buf_load_f64(r_v, 8u * j, 8u * (unsigned)off) : pf_v;
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
                fu[tc_tc(0) * RP] = N0.x; fv[tc_tc(0)] = N0.y;
                fu[tc_tc(1) * RP] = N1.x; fv[tc_tc(1)] = N1.y;
                fu[tc_tc(2) * RP] = N2.x; fv[tc_tc(2)] = N2.y;
#pragma unroll
                for (int kk = 0; kk < 3; ++kk) {                           // w_b: -A (x e_k0 + y e_k1)
                    const float h0 = xf * cs[9 + 6 * kk] + yf * cs[12 + 6 * kk];
                    const float h1 = xf * cs[10 + 6 * kk] + yf * cs[13 + 6 * kk];
                    const float h2 = xf * cs[11 + 6 * kk] + yf * cs[14 + 6 * kk];
                    const f2 w = N0 * h0 + N1 * h1 + N2 * h2;
                    fu[(kTcWb + kk) * RP] = w.x; fv[kTcWb + kk] = w.y;
                }
                {                                                          // w_c: -A (dR_c/dw_k Pw) = a_k . (Q' x n), both rows at once
                    const f2 c0 = N2 * Q1 - N1 * Q2, c1 = N0 * Q2 - N2 * Q0, c2 = N1 * Q0 - N0 * Q1;
#pragma unroll
                    for (int kk = 0; kk < 3; ++kk) {
                        const f2 w = c0 * cf[12 + 3 * kk] + c1 * cf[13 + 3 * kk] + c2 * cf[14 + 3 * kk];
                        fu[(kTcWc + kk) * RP] = w.x; fv[kTcWc + kk] = w.y;
                    }
                }
                fu[kTcF * RP] = -mxf;  fv[kTcF] = -myf;
                fu[kTcOne * RP] = -1.f; fv[kTcOne] = -1.f;
                const float kxi = c3 * c2 * e1, klam = c3 * e2, kal = e3 * cf[46];
                const f2 a = HM * kxi, b = HM * klam, c = HM * kal;
                fu[kTcXi * RP] = a.x;  fv[kTcXi] = a.y;
                fu[kTcLam * RP] = b.x; fv[kTcLam] = b.y;
                fu[kTcAl * RP] = c.x;  fv[kTcAl] = c.y;
                fu[kTcR * RP] = (float)ru; fv[kTcR] = (float)rv;
            } else if (lane < prev_nv) {
#pragma unroll
                for (int c = 0; c < kTcols; ++c) fu[c * RP] = 0.f;
            }
            if (c0 == 0) {
                // Prefetch of the next view, issued once the current view's observations have been consumed: the
                // loads reuse the same registers (no copy that would have to wait for them), and everything
                // between here and their use at the top of the next view is four unconditional stores.
                // Always issued (the block's last view re-reads itself; lanes past the corner count read past
                // the end of the buffer, i.e. zero): unconditional loads keep the vmcnt bookkeeping exact.
                const int vn = min(view + 1, vend - 1);
                const int cn = view + 1 < vend ? __builtin_amdgcn_readlane(m_cnt, vn - vbase) : 0;
                // pull the next view's 384-byte constant record into the L2 (one tracked load, lanes 0..5)
                warm = __builtin_amdgcn_raw_buffer_load_b32(r_vc, lane < 6 ? 64 * lane : (int)0xffffe000u, (int)(8u * (unsigned)kVStride * (unsigned)vn), 0);
                pf_u = buf_load_f64(r_u, lane < cn ? 8u * lane : 0xffffe000u, 8u * (unsigned)off_next);
                pf_v = buf_load_f64(r_v, lane < cn ? 8u * lane : 0xffffe000u, 8u * (unsigned)off_next);
            }
            wave_lds_fence();
            const int nv = min(64, cnt - c0);
            prev_nv = nv;
            const int ksteps = (nv + 3) >> 2;
            f4 aU = { 0.f, 0.f, 0.f, 0.f }, aV = { 0.f, 0.f, 0.f, 0.f };
            {
                float a0 = fpA[0], b0 = fpB[0], a1 = fpA[4], b1 = fpB[4];
                for (int t = 0; t < ksteps; t += 2) {
                    const int tn = min(t + 2, 14);
                    const float na0 = fpA[4 * tn], nb0 = fpB[4 * tn], na1 = fpA[4 * tn + 4], nb1 = fpB[4 * tn + 4];
                    aU = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, aU, 0, 0, 0);
                    aU = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, aU, 0, 0, 0);
                    a0 = na0; b0 = nb0; a1 = na1; b1 = nb1;
                }
            }
            wave_lds_fence();
            if (valid) {
#pragma unroll
                for (int c = 0; c < kTcols; ++c) fu[c * RP] = fv[c];
            }
            wave_lds_fence();
            {
                float a0 = fpA[0], b0 = fpB[0], a1 = fpA[4], b1 = fpB[4];
                for (int t = 0; t < ksteps; t += 2) {
                    const int tn = min(t + 2, 14);
                    const float na0 = fpA[4 * tn], nb0 = fpB[4 * tn], na1 = fpA[4 * tn + 4], nb1 = fpB[4 * tn + 4];
                    aV = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, aV, 0, 0, 0);
                    aV = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, aV, 0, 0, 0);
                    a0 = na0; b0 = nb0; a1 = na1; b1 = nb1;
                }
            }
            wave_lds_fence();
            // one pass (<= 64 corners) of fp32 accumulation, then fp64
#pragma unroll
            for (int r = 0; r < 4; ++r) { accU[r] += (double)aU[r]; accV[r] += (double)aV[r]; }
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


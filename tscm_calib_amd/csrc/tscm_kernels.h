// tscm_kernels.h -- device-side data layout and HIP kernels of the TSCM LM solver (gfx950).
//
// HBM layout (all fp64, device-resident for the whole solve):
//   obs_u/obs_v      SoA corner observations, re-packed so the views of one camera are
//                    contiguous and sorted by board: a wave streams them with unit stride.
//   rec[2][V][132]   per-view Schur pieces written by the Gram kernel:
//                      [0..95]   E^T [F | r]   6 x 16  (cols 0-12 = W, col 13 = E^T r)
//                      [96..131] E^T E         6 x 6
//                    double buffered: index ctrl->cur = system at x, cur^1 = candidate.
//   H_stage          per camera a 16x16 tile [F | r]^T [F | r]  (13x13 Gram, col 13 =
//                    F^T r, [13][13] = r^T r) + 8 scalars; fixed address so that RCCL can
//                    all-reduce it without knowing the device-side buffer index.
//   Y[V][96], L[B][21], z[B][6], D2[B][6]   e-block factors kept for back-substitution.
//   T[n_pad^2]       Schur complement sum_b Y_b^T Y_b as dense 16x16 blocks per camera pair.
// The LM control state (trust-region radius, accept/reject, termination, iteration log)
// lives in `Ctrl` in device memory; every kernel starts with `if (ctrl->done) return`.
#pragma once

#include "tscm_math.h"

namespace tscm {

constexpr int kRec = 132;          // doubles per view record
constexpr int kRecEE = 96;         // offset of E^T E inside a record
constexpr int kRP = 130;           // LDS row pitch (doubles) of the column-major Jacobian tile:
                                   // 2*kRP = 260 = 4 (mod 64) dwords -> conflict-free ds_read_b64
constexpr int kFcols = 14;         // Jacobian columns staged for F (13 params + residual)
constexpr int kScal = 8;           // scalars appended to H_stage
constexpr int kCamG1 = 16;         // first-level fan-in of the per-camera tile reduction
constexpr int kMaxCam = 8;         // n_pad = 16*C <= 128 (reduced system kept in LDS)
constexpr int kMaxLog = 256;

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

struct Options {
    int max_num_iterations;
    double function_tolerance, gradient_tolerance, parameter_tolerance;
    double initial_radius, max_radius, min_radius;
    double min_relative_decrease, min_lm_diagonal, max_lm_diagonal;
    int max_invalid;
    int jacobi_scaling;
};

struct IterLog {
    int iteration, step_is_valid, step_is_successful, pad;
    double cost, cost_change, gradient_max_norm, gradient_norm, step_norm, relative_decrease, radius;
};

enum TermReason { kNone = 0, kMaxIter, kGradTol, kMinRadius, kParamTol, kFuncTol, kInvalidSteps };

struct Ctrl {
    // ---- header (polled by the host) ----
    int done, term_type, term_reason, iteration;
    int cur, lin_fail, num_successful, num_unsuccessful;
    int num_invalid, n_log, lm_iterations, pad0;
    double radius, decrease_factor;
    double x_cost, x_norm, gmax, gnorm;
    double model_cam, stepsq_cam;
    double se_min, se_cur, se_ref, se_cand, se_acc_ref, se_acc_cand;
    double initial_cost;
    Options opt;
    IterLog log[kMaxLog];
};

struct DevProblem {
    int C, B, n_points, V, N, n_pad;
    int n_chunks, n_pairs, n_pchunks, n_bids;
    const double *board_xy;
    const int *view_cam, *view_board, *view_obs, *view_count;
    const double *obs_u, *obs_v;
    const int *chunk_vb, *chunk_ve, *chunk_cam, *cam_chunk_ptr;
    const int *bv_ptr, *bv_idx;
    const int *pair_i, *pair_j;
    const int *pc_begin, *pc_end, *bid_pc_ptr, *bid_mi, *bid_mj;
    const unsigned char *cam_const, *cam_active;
};

struct DevState {
    double *cam_rt[2], *intr[2], *board_rt[2];
    double *board_pc, *cam_pc;
    double *rec[2];
    double *campart, *campart2;
    double *H[2], *H_stage, *M_stage;
    double *s_b, *s_c;
    double *L, *z, *D2, *Y;
    double *pairpart, *T;
    double *yhat;
    double *bs_part, *st_part;
    int n_bs_blocks, n_st_blocks;
    Ctrl *ctrl;
};

// ---------------------------------------------------------------------------------------------
// pose constants of the evaluation target (rotations and their derivatives; the two sincos
// per pose are hoisted out of the per-corner work).  grid: ceil((B + C)/256) x 256
// ---------------------------------------------------------------------------------------------
__global__ void k_pose_prep(DevProblem P, DevState S, int cand)
{
    if (S.ctrl->done) return;
    const int tgt = cand ? (S.ctrl->cur ^ 1) : S.ctrl->cur;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < P.B) {
        double rt[6], out[kBoardConst];
        for (int k = 0; k < 6; ++k) rt[k] = S.board_rt[tgt][6 * i + k];
        board_constants(rt, out);
        for (int k = 0; k < kBoardConst; ++k) S.board_pc[(size_t)kBoardConst * i + k] = out[k];
    } else if (i < P.B + P.C) {
        const int m = i - P.B;
        double rt[6], out[kCamConst];
        for (int k = 0; k < 6; ++k) rt[k] = S.cam_rt[tgt][6 * m + k];
        camera_constants(rt, out);
        for (int k = 0; k < kCamConst; ++k) S.cam_pc[kCamConst * m + k] = out[k];
    }
}

__device__ __forceinline__ void load_view_const(const DevProblem &P, const DevState &S, int tgt, int cam, int board, ViewConst &vc)
{
    const double *bp = S.board_pc + (size_t)kBoardConst * board;
    for (int k = 0; k < 3; ++k) { vc.r1[k] = bp[k]; vc.r2[k] = bp[3 + k]; vc.tb[k] = S.board_rt[tgt][6 * board + 3 + k]; }
    for (int k = 0; k < 3; ++k) for (int q = 0; q < 6; ++q) vc.db[k][q] = bp[6 + 6 * k + q];
    const double *cp = S.cam_pc + kCamConst * cam;
    for (int k = 0; k < 9; ++k) vc.Rc[k] = cp[k];
    for (int k = 0; k < 27; ++k) vc.dRc[k] = cp[9 + k];
    for (int k = 0; k < 3; ++k) vc.tc[k] = S.cam_rt[tgt][6 * cam + 3 + k];
    const double *I = S.intr[tgt] + 9 * cam;
    vc.fx = I[0]; vc.fy = I[1]; vc.cx = I[2]; vc.cy = I[3]; vc.xi = I[4]; vc.lam = I[5]; vc.al = I[6];
}

// ---------------------------------------------------------------------------------------------
// THE HOT KERNEL: per-corner TSCM projection + analytic 2x19 Jacobian + residual, then the
// Gram contractions  [F|r]^T[F|r] (per camera),  E^T[F|r] and E^T E (per view)  on the f64
// matrix cores (v_mfma_f64_16x16x4_f64).
//   one wave (= one 64-thread workgroup) per chunk of consecutive views of ONE camera;
//   lane = corner: coalesced SoA loads of u[], v[]; board points staged in LDS;
//   the 2 x 20 Jacobian rows of 64 corners are transposed through LDS (column-major,
//   conflict-free pitch) into MFMA operand layout: lane (c, k) feeds J[row 4t+k][col c];
//   the 16x16 camera tile stays in the accumulator across all views of the chunk.
// dynamic LDS: (14 + 6) * kRP + 2 * n_points doubles.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_eval_gram(DevProblem P, DevState S, int cand)
{
    if (S.ctrl->done) return;
    const int tgt = cand ? (S.ctrl->cur ^ 1) : S.ctrl->cur;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *Fl = lds;
    double *El = lds + kFcols * kRP;
    double *bxy = El + kE * kRP;
    const int lane = threadIdx.x;
    const int chunk = blockIdx.x;
    for (int i = lane; i < 2 * P.n_points; i += 64) bxy[i] = P.board_xy[i];
    const int cam = P.chunk_cam[chunk];
    const int vb = P.chunk_vb[chunk], ve = P.chunk_ve[chunk];
    const int col = lane & 15, kq = lane >> 4;
    d4 accFF = { 0.0, 0.0, 0.0, 0.0 };
    __syncthreads();
    for (int view = vb; view < ve; ++view) {
        const int board = P.view_board[view];
        const int cnt = P.view_count[view];
        const int off = P.view_obs[view];
        ViewConst vc;
        load_view_const(P, S, tgt, cam, board, vc);
        d4 accEF = { 0.0, 0.0, 0.0, 0.0 };
        d4 accEE = { 0.0, 0.0, 0.0, 0.0 };
        for (int c0 = 0; c0 < cnt; c0 += 64) {
            const int j = c0 + lane;
            const bool valid = j < cnt;
            double r[2] = { 0.0, 0.0 }, JE[2][kE], JF[2][kFA];
            if (valid) {
                corner_residual_jacobian(vc, bxy[2 * j], bxy[2 * j + 1], P.obs_u[off + j], P.obs_v[off + j], r, JE, JF);
            } else {
                for (int c = 0; c < kE; ++c) { JE[0][c] = 0.0; JE[1][c] = 0.0; }
                for (int c = 0; c < kFA; ++c) { JF[0][c] = 0.0; JF[1][c] = 0.0; }
            }
#pragma unroll
            for (int c = 0; c < kFA; ++c) *reinterpret_cast<d2 *>(Fl + c * kRP + 2 * lane) = d2{ JF[0][c], JF[1][c] };
            *reinterpret_cast<d2 *>(Fl + kFR * kRP + 2 * lane) = d2{ r[0], r[1] };
#pragma unroll
            for (int c = 0; c < kE; ++c) *reinterpret_cast<d2 *>(El + c * kRP + 2 * lane) = d2{ JE[0][c], JE[1][c] };
            __syncthreads();
            const int nv = min(64, cnt - c0);
            const int ksteps = (2 * nv + 3) >> 2;
            const double *fp = Fl + (col < kFcols ? col : 0) * kRP + kq;
            const double *ep = El + (col < kE ? col : 0) * kRP + kq;
            for (int t = 0; t < ksteps; ++t) {
                const double af = col < kFcols ? fp[4 * t] : 0.0;
                const double ae = col < kE ? ep[4 * t] : 0.0;
                accFF = __builtin_amdgcn_mfma_f64_16x16x4f64(af, af, accFF, 0, 0, 0);
                accEF = __builtin_amdgcn_mfma_f64_16x16x4f64(ae, af, accEF, 0, 0, 0);
                accEE = __builtin_amdgcn_mfma_f64_16x16x4f64(ae, ae, accEE, 0, 0, 0);
            }
            __syncthreads();
        }
        // D[row = kq + 4*reg][col]: rows 0..5 of E^T[F|r] and E^T E
        double *rec = S.rec[tgt] + (size_t)kRec * view;
        rec[kq * 16 + col] = accEF[0];
        if (kq < 2) rec[(kq + 4) * 16 + col] = accEF[1];
        if (col < kE) {
            rec[kRecEE + kq * 6 + col] = accEE[0];
            if (kq < 2) rec[kRecEE + (kq + 4) * 6 + col] = accEE[1];
        }
    }
    double *part = S.campart + (size_t)256 * chunk;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) part[(kq + 4 * rg) * 16 + col] = accFF[rg];
}

// per-camera tile reduction, level 1: grid (C * kCamG1) x 256
__global__ void k_cam_reduce1(DevProblem P, DevState S)
{
    if (S.ctrl->done) return;
    const int cam = blockIdx.x / kCamG1, g = blockIdx.x % kCamG1;
    const int cb = P.cam_chunk_ptr[cam], ce = P.cam_chunk_ptr[cam + 1];
    const int n = ce - cb;
    const int per = (n + kCamG1 - 1) / kCamG1;
    const int b = cb + g * per, e = min(ce, b + per);
    const int t = threadIdx.x;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int c = b;
    for (; c + 3 < e; c += 4) {
        a0 += S.campart[(size_t)256 * c + t];
        a1 += S.campart[(size_t)256 * (c + 1) + t];
        a2 += S.campart[(size_t)256 * (c + 2) + t];
        a3 += S.campart[(size_t)256 * (c + 3) + t];
    }
    for (; c < e; ++c) a0 += S.campart[(size_t)256 * c + t];
    S.campart2[(size_t)256 * blockIdx.x + t] = (a0 + a1) + (a2 + a3);
}

// deterministic block reductions (256 threads)
__device__ __forceinline__ double block_sum256(double v, double *sm)
{
    const int t = threadIdx.x;
    sm[t] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (t < s) sm[t] += sm[t + s]; __syncthreads(); }
    const double r = sm[0];
    __syncthreads();
    return r;
}
__device__ __forceinline__ double block_max256(double v, double *sm)
{
    const int t = threadIdx.x;
    sm[t] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (t < s) sm[t] = fmax(sm[t], sm[t + s]); __syncthreads(); }
    const double r = sm[0];
    __syncthreads();
    return r;
}

// per-board gradient / norm statistics of the evaluation target (and, at iteration 0, the
// Jacobi scaling of the board columns: s = 1/(1 + ||J_col||)).  grid ceil(B/256) x 256
__global__ void k_board_stats(DevProblem P, DevState S, int cand, int init)
{
    if (S.ctrl->done) return;
    __shared__ double sm[256];
    const int tgt = cand ? (S.ctrl->cur ^ 1) : S.ctrl->cur;
    const int b = blockIdx.x * 256 + threadIdx.x;
    double gmax = 0.0, gsq = 0.0, xsq = 0.0;
    if (b < P.B) {
        const int q0 = P.bv_ptr[b], q1 = P.bv_ptr[b + 1];
        if (q1 > q0) {
            double g[6] = { 0, 0, 0, 0, 0, 0 }, dg[6] = { 0, 0, 0, 0, 0, 0 };
            for (int q = q0; q < q1; ++q) {
                const double *rec = S.rec[tgt] + (size_t)kRec * P.bv_idx[q];
                for (int i = 0; i < 6; ++i) { g[i] += rec[i * 16 + kFR]; dg[i] += rec[kRecEE + i * 7]; }
            }
            for (int i = 0; i < 6; ++i) {
                const double x = S.board_rt[tgt][6 * b + i];
                const double d = x - (x + (-g[i]));   // |x - Plus(x, -gradient)| like Ceres
                gmax = fmax(gmax, fabs(d)); gsq += d * d; xsq += x * x;
                if (init) S.s_b[6 * b + i] = S.ctrl->opt.jacobi_scaling ? 1.0 / (1.0 + sqrt(dg[i])) : 1.0;
            }
        } else if (init) {
            for (int i = 0; i < 6; ++i) S.s_b[6 * b + i] = 1.0;
        }
    }
    const double m = block_max256(gmax, sm);
    const double s1 = block_sum256(gsq, sm);
    const double s2 = block_sum256(xsq, sm);
    if (threadIdx.x == 0) { S.st_part[3 * blockIdx.x] = m; S.st_part[3 * blockIdx.x + 1] = s1; S.st_part[3 * blockIdx.x + 2] = s2; }
}

// level-2 camera reduction into H_stage + reduction of the per-block scalar partials.
// grid (C + 1) x 256.  H_stage scal: [0] model_b [1] stepsq_b [2] xsq_b [3] gsq_b ; M_stage[0] gmax_b
__global__ void k_finalize_eval(DevProblem P, DevState S, int have_backsub)
{
    if (S.ctrl->done) return;
    __shared__ double sm[256];
    const int t = threadIdx.x;
    if ((int)blockIdx.x < P.C) {
        const int cam = blockIdx.x;
        double a = 0.0;
        for (int g = 0; g < kCamG1; ++g) a += S.campart2[(size_t)256 * (cam * kCamG1 + g) + t];
        S.H_stage[256 * cam + t] = a;
        return;
    }
    double mb = 0.0, ss = 0.0;
    if (have_backsub) for (int i = t; i < S.n_bs_blocks; i += 256) { mb += S.bs_part[2 * i]; ss += S.bs_part[2 * i + 1]; }
    double gm = 0.0, gs = 0.0, xs = 0.0;
    for (int i = t; i < S.n_st_blocks; i += 256) { gm = fmax(gm, S.st_part[3 * i]); gs += S.st_part[3 * i + 1]; xs += S.st_part[3 * i + 2]; }
    mb = block_sum256(mb, sm); ss = block_sum256(ss, sm); gs = block_sum256(gs, sm); xs = block_sum256(xs, sm);
    gm = block_max256(gm, sm);
    if (t == 0) {
        double *sc = S.H_stage + 256 * P.C;
        sc[0] = mb; sc[1] = ss; sc[2] = xs; sc[3] = gs; sc[4] = 0.0; sc[5] = 0.0; sc[6] = 0.0; sc[7] = 0.0;
        S.M_stage[0] = gm;
    }
}

// ---------------------------------------------------------------------------------------------
// e-block elimination (SchurEliminator, one 6x6 block per board): 16 lanes per board.
//   V = sum_views E^T E, Jacobi-scaled, damped with D^2 = clamp(diag)/radius, Cholesky L L^T;
//   lane a solves column a of  L Y = S_b W  for every view (a = 13: z = L^{-1} S_b E^T r).
// grid ceil(B/16) x 256
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool chol6(const double M[21], double L[21])
{
    // packed lower: idx(i,j) = i(i+1)/2 + j
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double d = M[j * (j + 1) / 2 + j];
#pragma unroll
        for (int k = 0; k < j; ++k) d -= L[j * (j + 1) / 2 + k] * L[j * (j + 1) / 2 + k];
        if (!(d > 0.0)) { ok = false; d = 1.0; }
        const double sd = sqrt(d);
        const double inv = 1.0 / sd;
        L[j * (j + 1) / 2 + j] = sd;
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
            double s = M[i * (i + 1) / 2 + j];
#pragma unroll
            for (int k = 0; k < j; ++k) s -= L[i * (i + 1) / 2 + k] * L[j * (j + 1) / 2 + k];
            L[i * (i + 1) / 2 + j] = s * inv;
        }
    }
    return ok;
}

__global__ __launch_bounds__(256) void k_schur_factor(DevProblem P, DevState S)
{
    if (S.ctrl->done) return;
    const int cur = S.ctrl->cur;
    const int b = (blockIdx.x * 256 + threadIdx.x) >> 4;
    const int a = threadIdx.x & 15;
    if (b >= P.B) return;
    const int q0 = P.bv_ptr[b], q1 = P.bv_ptr[b + 1];
    if (q1 == q0) return;
    const double radius = S.ctrl->radius;
    const double dmin = S.ctrl->opt.min_lm_diagonal, dmax = S.ctrl->opt.max_lm_diagonal;
    double sb[6];
    for (int i = 0; i < 6; ++i) sb[i] = S.s_b[6 * b + i];
    double M[21], g[6] = { 0, 0, 0, 0, 0, 0 };
    for (int i = 0; i < 21; ++i) M[i] = 0.0;
    for (int q = q0; q < q1; ++q) {
        const double *rec = S.rec[cur] + (size_t)kRec * P.bv_idx[q];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
#pragma unroll
            for (int j = 0; j <= i; ++j) M[i * (i + 1) / 2 + j] += rec[kRecEE + i * 6 + j];
            g[i] += rec[i * 16 + kFR];
        }
    }
    double D2[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
        for (int j = 0; j <= i; ++j) M[i * (i + 1) / 2 + j] *= sb[i] * sb[j];
        D2[i] = fmin(fmax(M[i * (i + 1) / 2 + i], dmin), dmax) / radius;
        M[i * (i + 1) / 2 + i] += D2[i];
    }
    double L[21];
    if (!chol6(M, L)) S.ctrl->lin_fail = 1;
    double il[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) il[i] = 1.0 / L[i * (i + 1) / 2 + i];
    for (int q = q0; q < q1; ++q) {
        const int v = P.bv_idx[q];
        const double *rec = S.rec[cur] + (size_t)kRec * v;
        double y[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            double w = (a == kFR ? g[i] : rec[i * 16 + a]) * sb[i];
#pragma unroll
            for (int k = 0; k < i; ++k) w -= L[i * (i + 1) / 2 + k] * y[k];
            y[i] = w * il[i];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) S.Y[(size_t)96 * v + i * 16 + a] = y[i];
        if (a == kFR && q == q0) for (int i = 0; i < 6; ++i) S.z[6 * b + i] = y[i];
    }
    if (a == 0) {
        for (int i = 0; i < 21; ++i) S.L[(size_t)21 * b + i] = L[i];
        for (int i = 0; i < 6; ++i) S.D2[6 * b + i] = D2[i];
    }
}

// Schur complement contributions  T(mi, mj) += Y'_i^T Y'_j  over pairs of views of one board.
// Pairs are pre-sorted by camera-pair block; one wave per chunk of pairs of a single block keeps
// its 16x16 tile in registers (4 entries per lane).  grid n_pchunks x 64
__global__ __launch_bounds__(64) void k_pair_gram(DevProblem P, DevState S)
{
    if (S.ctrl->done) return;
    const int pc = blockIdx.x;
    const int lane = threadIdx.x;
    const int a = lane & 15, bg = lane >> 4;
    double acc[4] = { 0.0, 0.0, 0.0, 0.0 };
    for (int p = P.pc_begin[pc]; p < P.pc_end[pc]; ++p) {
        const double *Yi = S.Y + (size_t)96 * P.pair_i[p];
        const double *Yj = S.Y + (size_t)96 * P.pair_j[p];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const double ya = Yi[k * 16 + a];
            const d4 yb = *reinterpret_cast<const d4 *>(Yj + k * 16 + 4 * bg);
            acc[0] += ya * yb[0]; acc[1] += ya * yb[1]; acc[2] += ya * yb[2]; acc[3] += ya * yb[3];
        }
    }
    double *out = S.pairpart + (size_t)256 * pc + a * 16 + 4 * bg;
    out[0] = acc[0]; out[1] = acc[1]; out[2] = acc[2]; out[3] = acc[3];
}

// grid n_bids x 256: sum the pair-chunk tiles of one camera-pair block into T
__global__ void k_T_reduce(DevProblem P, DevState S)
{
    if (S.ctrl->done) return;
    const int bid = blockIdx.x, t = threadIdx.x;
    double a0 = 0.0, a1 = 0.0;
    int c = P.bid_pc_ptr[bid];
    const int e = P.bid_pc_ptr[bid + 1];
    for (; c + 1 < e; c += 2) { a0 += S.pairpart[(size_t)256 * c + t]; a1 += S.pairpart[(size_t)256 * (c + 1) + t]; }
    if (c < e) a0 += S.pairpart[(size_t)256 * c + t];
    const int mi = P.bid_mi[bid], mj = P.bid_mj[bid];
    S.T[(size_t)(mi * 16 + (t >> 4)) * P.n_pad + mj * 16 + (t & 15)] = a0 + a1;
}

// ---------------------------------------------------------------------------------------------
// Reduced camera system (DenseSchurComplementSolver): one workgroup, matrix in LDS.
//   A = S_c (H_cc - T) S_c + D_c^2, rhs = S_c (g_c - t_r); inactive columns (padding, constant
//   camera pose, cameras without views) are replaced by identity rows.  Left-looking Cholesky
//   on the matrix augmented with the rhs row (forward substitution for free), then a
//   single-wave back-substitution.  Writes yhat = S_c y (the camera step is -yhat) and the
//   candidate camera parameters.   grid 1 x 256, dynamic LDS (n+1)*(n+4) doubles
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_solve_reduced(DevProblem P, DevState S)
{
    if (S.ctrl->done) return;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ int s_fail;
    __shared__ double sred[256];
    const int n = P.n_pad, ld = n + 4;
    double *A = lds;                 // (n+1) rows x ld
    const int tid = threadIdx.x;
    const int cur = S.ctrl->cur;
    const double radius = S.ctrl->radius;
    const double dmin = S.ctrl->opt.min_lm_diagonal, dmax = S.ctrl->opt.max_lm_diagonal;
    const double *H = S.H[cur];
    if (tid == 0) s_fail = S.ctrl->lin_fail;
    auto active = [&](int i) -> bool {
        const int m = i >> 4, a = i & 15;
        return a < kFA && P.cam_active[m] && !(a < 6 && P.cam_const[m]);
    };
    for (int idx = tid; idx < (n + 1) * n; idx += 256) {
        const int i = idx / n, j = idx % n;
        double v;
        if (i < n) {
            if (j > i) continue;                 // lower triangle only
            const bool ai = active(i), aj = active(j);
            if (ai && aj) {
                const int mi = i >> 4, a = i & 15, mj = j >> 4, b = j & 15;
                const double h = (mi == mj) ? H[256 * mi + a * 16 + b] : 0.0;
                // T holds upper blocks (mj <= mi here -> block (mj, mi), transposed)
                const double tt = (mi == mj) ? S.T[(size_t)i * n + j] : S.T[(size_t)j * n + i];
                v = S.s_c[i] * S.s_c[j] * (h - tt);
                if (i == j) v += fmin(fmax(S.s_c[i] * S.s_c[i] * h, dmin), dmax) / radius;
            } else {
                v = (i == j) ? 1.0 : 0.0;
            }
        } else {
            // rhs row
            if (active(j)) { const int mj = j >> 4, b = j & 15; v = S.s_c[j] * (H[256 * mj + b * 16 + kFR] - S.T[(size_t)j * n + mj * 16 + kFR]); }
            else v = 0.0;
        }
        A[i * ld + j] = v;
    }
    __syncthreads();
    // left-looking Cholesky, 4 threads per row, rows in passes of 64
    const int seg = tid & 3, r0 = tid >> 2;
    for (int j = 0; j < n; ++j) {
        double sj = 0.0;
        for (int k = seg; k < j; k += 4) sj += A[j * ld + k] * A[j * ld + k];
        sj += __shfl_xor(sj, 1, 4); sj += __shfl_xor(sj, 2, 4);
        double pj = A[j * ld + j] - sj;
        if (!(pj > 0.0)) { if (tid == 0) s_fail = 1; pj = 1.0; }
        const double invp = 1.0 / sqrt(pj);
        for (int r = r0; r <= n; r += 64) {
            if (r <= j) continue;
            double s = 0.0;
            for (int k = seg; k < j; k += 4) s += A[r * ld + k] * A[j * ld + k];
            s += __shfl_xor(s, 1, 4); s += __shfl_xor(s, 2, 4);
            if (seg == 0) A[r * ld + j] = (A[r * ld + j] - s) * invp;
        }
        __syncthreads();
        if (tid == 0) A[j * ld + j] = pj * invp;
        // (diagonal is only read again in the back-substitution; the barrier of the next
        //  column orders this store before that)
    }
    __syncthreads();
    // back-substitution L^T y = w (w = row n), single wave, lane i owns y[i], y[i+64]
    double *yv = A + (size_t)n * ld;     // w in place
    if (tid < 64) {
        for (int k = n - 1; k >= 0; --k) {
            const double yk = yv[k] / A[k * ld + k];
            __builtin_amdgcn_wave_barrier();
            if (tid == 0) yv[k] = yk;
            for (int i = tid; i < k; i += 64) yv[i] -= A[k * ld + i] * yk;
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
    // yhat, candidate camera parameters, camera part of the model cost change / step norm
    double model = 0.0, stepsq = 0.0;
    const int fail = s_fail;
    for (int i = tid; i < n; i += 256) {
        const int m = i >> 4, a = i & 15;
        const bool act = active(i) && !fail;
        const double yh = act ? S.s_c[i] * yv[i] : 0.0;
        S.yhat[i] = yh;
        if (a < 6) {
            const double x = S.cam_rt[cur][6 * m + a];
            const double xn = x + (-yh);
            S.cam_rt[cur ^ 1][6 * m + a] = xn;
            const double d = x - xn; stepsq += d * d;
        } else if (a < kFA) {
            const double x = S.intr[cur][9 * m + (a - 6)];
            const double xn = x + (-yh);
            S.intr[cur ^ 1][9 * m + (a - 6)] = xn;
            const double d = x - xn; stepsq += d * d;
        } else if (a < 15) {
            S.intr[cur ^ 1][9 * m + (a - 6)] = S.intr[cur][9 * m + (a - 6)];   // b, c are inert
        }
    }
    __syncthreads();
    // model_cam = yhat^T g_c - 1/2 yhat^T H_cc yhat   (block diagonal H_cc)
    for (int i = tid; i < n; i += 256) {
        const int m = i >> 4, a = i & 15;
        if (a >= kFA) continue;
        const double yi = S.yhat[i];
        if (yi == 0.0) continue;
        double hy = 0.0;
        for (int b = 0; b < kFA; ++b) hy += H[256 * m + a * 16 + b] * S.yhat[m * 16 + b];
        model += yi * (H[256 * m + a * 16 + kFR] - 0.5 * hy);
    }
    model = block_sum256(model, sred);
    stepsq = block_sum256(stepsq, sred);
    if (tid == 0) { S.ctrl->model_cam = model; S.ctrl->stepsq_cam = stepsq; S.ctrl->lin_fail = fail; }
}

// back-substitution of the board steps (SchurEliminator::BackSubstitute), 16 lanes per board:
//   y_b = L^{-T} (z - sum_v Y_v yhat[m_v]);  delta_b = -s_b y_b;  candidate = x + delta.
// grid ceil(B/16) x 256
__global__ __launch_bounds__(256) void k_backsub(DevProblem P, DevState S)
{
    if (S.ctrl->done) return;
    __shared__ double sm_m[16], sm_s[16];
    const int cur = S.ctrl->cur;
    const int fail = S.ctrl->lin_fail;
    const int grp = threadIdx.x >> 4;
    const int b = blockIdx.x * 16 + grp;
    const int a = threadIdx.x & 15;
    double mb = 0.0, ss = 0.0;
    if (b < P.B) {
        const int q0 = P.bv_ptr[b], q1 = P.bv_ptr[b + 1];
        if (q1 == q0 || fail) {
            if (a < 6) S.board_rt[cur ^ 1][6 * b + a] = S.board_rt[cur][6 * b + a];
        } else {
            double p[6] = { 0, 0, 0, 0, 0, 0 };
            for (int q = q0; q < q1; ++q) {
                const int v = P.bv_idx[q];
                const double yh = (a < kFA) ? S.yhat[P.view_cam[v] * 16 + a] : 0.0;
                const double *Yv = S.Y + (size_t)96 * v;
#pragma unroll
                for (int k = 0; k < 6; ++k) p[k] += Yv[k * 16 + a] * yh;
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                p[k] += __shfl_xor(p[k], 1, 16); p[k] += __shfl_xor(p[k], 2, 16);
                p[k] += __shfl_xor(p[k], 4, 16); p[k] += __shfl_xor(p[k], 8, 16);
            }
            double t[6], y[6], L[21];
            for (int i = 0; i < 21; ++i) L[i] = S.L[(size_t)21 * b + i];
#pragma unroll
            for (int k = 0; k < 6; ++k) t[k] = S.z[6 * b + k] - p[k];
#pragma unroll
            for (int i = 5; i >= 0; --i) {
                double w = t[i];
#pragma unroll
                for (int k = i + 1; k < 6; ++k) w -= L[k * (k + 1) / 2 + i] * y[k];
                y[i] = w / L[i * (i + 1) / 2 + i];
            }
            double m = 0.0, s = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                m += 0.5 * t[k] * t[k] + 0.5 * S.D2[6 * b + k] * y[k] * y[k];
                const double x = S.board_rt[cur][6 * b + k];
                const double xn = x + (-(S.s_b[6 * b + k] * y[k]));
                const double d = x - xn;
                s += d * d;
                if (a == k) S.board_rt[cur ^ 1][6 * b + k] = xn;
            }
            mb = m; ss = s;
        }
    }
    if (a == 0) { sm_m[grp] = mb; sm_s[grp] = ss; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double m = 0.0, s = 0.0;
        for (int i = 0; i < 16; ++i) { m += sm_m[i]; s += sm_s[i]; }
        S.bs_part[2 * blockIdx.x] = m; S.bs_part[2 * blockIdx.x + 1] = s;
    }
}

// ---------------------------------------------------------------------------------------------
// LM control (TrustRegionMinimizer + LevenbergMarquardtStrategy + TrustRegionStepEvaluator),
// one thread.  `init` = IterationZero; otherwise the tail of one loop iteration followed by
// FinalizeIterationAndCheckIfMinimizerCanContinue.
// ---------------------------------------------------------------------------------------------
__device__ void camera_norms(const DevProblem &P, const DevState &S, int idx, const double *H,
                             double &gmax, double &gsq, double &xsq)
{
    gmax = 0.0; gsq = 0.0; xsq = 0.0;
    for (int m = 0; m < P.C; ++m) {
        if (!P.cam_active[m]) continue;
        for (int a = 0; a < 15; ++a) {
            if (a < 6 && P.cam_const[m]) continue;
            const double x = a < 6 ? S.cam_rt[idx][6 * m + a] : S.intr[idx][9 * m + (a - 6)];
            const double g = a < kFA ? H[256 * m + a * 16 + kFR] : 0.0;   // b, c: zero gradient
            const double d = x - (x + (-g));
            gmax = fmax(gmax, fabs(d)); gsq += d * d; xsq += x * x;
        }
    }
}

__global__ __launch_bounds__(256) void k_control(DevProblem P, DevState S, int init)
{
    Ctrl &c = *S.ctrl;
    if (c.done) return;
    const Options &o = c.opt;
    const int tgt = init ? c.cur : (c.cur ^ 1);
    // publish the staged (all-reduced) camera tiles as the target system's H
    for (int i = threadIdx.x; i < 256 * P.C; i += 256) S.H[tgt][i] = S.H_stage[i];
    if (threadIdx.x != 0) return;
    const double *H = S.H_stage;
    const double *sc = S.H_stage + 256 * P.C;
    double cost = 0.0;
    for (int m = 0; m < P.C; ++m) cost += H[256 * m + kFR * 16 + kFR];
    cost *= 0.5;
    double gmax_c, gsq_c, xsq_c;
    camera_norms(P, S, tgt, H, gmax_c, gsq_c, xsq_c);
    const double gmax_t = fmax(gmax_c, S.M_stage[0]);
    const double gnorm_t = sqrt(gsq_c + sc[3]);
    const double xnorm_t = sqrt(xsq_c + sc[2]);

    IterLog it;
    it.pad = 0;
    if (init) {
        for (int i = 0; i < P.n_pad; ++i) {
            const int m = i >> 4, a = i & 15;
            const double hii = (a < kFA) ? H[256 * m + a * 16 + a] : 0.0;
            S.s_c[i] = o.jacobi_scaling ? 1.0 / (1.0 + sqrt(hii)) : 1.0;
        }
        c.x_cost = cost; c.initial_cost = cost; c.x_norm = xnorm_t; c.gmax = gmax_t; c.gnorm = gnorm_t;
        c.se_min = c.se_cur = c.se_ref = c.se_cand = cost; c.se_acc_ref = 0.0; c.se_acc_cand = 0.0;
        it.iteration = 0; it.step_is_valid = 1; it.step_is_successful = 1;
        it.cost = cost; it.cost_change = 0.0; it.gradient_max_norm = gmax_t; it.gradient_norm = gnorm_t;
        it.step_norm = 0.0; it.relative_decrease = 0.0;
        c.iteration = 0;
    } else {
        c.iteration += 1;
        c.lm_iterations += 1;
        it.iteration = c.iteration;
        const double model = sc[0] + c.model_cam;
        const double step_norm = sqrt(sc[1] + c.stepsq_cam);
        const bool valid = !c.lin_fail && isfinite(model) && isfinite(step_norm) && model > 0.0;
        c.lin_fail = 0;
        it.step_is_valid = valid ? 1 : 0;
        it.gradient_max_norm = c.gmax; it.gradient_norm = c.gnorm;
        if (!valid) {
            // HandleInvalidStep
            if (++c.num_invalid >= o.max_invalid) { c.done = 1; c.term_type = 2; c.term_reason = kInvalidSteps; return; }
            c.radius = c.radius / c.decrease_factor; c.decrease_factor *= 2.0;
            it.cost = c.x_cost; it.cost_change = 0.0; it.step_norm = 0.0; it.relative_decrease = 0.0; it.step_is_successful = 0;
        } else {
            c.num_invalid = 0;
            double cand = cost;
            if (!isfinite(cand)) cand = DBL_MAX;
            it.step_norm = step_norm;
            it.cost_change = c.x_cost - cand;
            it.cost = c.x_cost;
            it.relative_decrease = 0.0;
            it.step_is_successful = 0;
            // ParameterToleranceReached / FunctionToleranceReached: return before accepting
            if (step_norm <= o.parameter_tolerance * (c.x_norm + o.parameter_tolerance)) {
                c.done = 1; c.term_type = 0; c.term_reason = kParamTol; return;
            }
            if (fabs(it.cost_change) <= o.function_tolerance * c.x_cost) {
                c.done = 1; c.term_type = 0; c.term_reason = kFuncTol; return;
            }
            double q;
            if (cand >= DBL_MAX) q = -DBL_MAX;
            else {
                const double rel = (c.se_cur - cand) / model;
                const double hist = (c.se_ref - cand) / (c.se_acc_ref + model);
                q = rel > hist ? rel : hist;
            }
            it.relative_decrease = q;
            if (q > o.min_relative_decrease) {
                // HandleSuccessfulStep
                c.cur = tgt;
                c.x_cost = cand; c.x_norm = xnorm_t; c.gmax = gmax_t; c.gnorm = gnorm_t;
                it.cost = cand; it.gradient_max_norm = gmax_t; it.gradient_norm = gnorm_t;
                it.step_is_successful = 1;
                c.radius = c.radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * q - 1.0, 3.0));
                c.radius = fmin(o.max_radius, c.radius);
                c.decrease_factor = 2.0;
                c.se_cur = cand; c.se_acc_cand += model; c.se_acc_ref += model;
                if (c.se_cur < c.se_min) { c.se_min = c.se_cur; c.se_cand = c.se_cur; c.se_acc_cand = 0.0; }
                else if (c.se_cur > c.se_cand) { c.se_cand = c.se_cur; c.se_acc_cand = 0.0; }
                c.se_ref = c.se_cand; c.se_acc_ref = c.se_acc_cand;
            } else {
                it.cost = cand;
                c.radius = c.radius / c.decrease_factor; c.decrease_factor *= 2.0;
            }
        }
    }
    // FinalizeIterationAndCheckIfMinimizerCanContinue
    if (it.step_is_successful) ++c.num_successful; else ++c.num_unsuccessful;
    it.radius = c.radius;
    if (c.n_log < kMaxLog) c.log[c.n_log] = it;
    ++c.n_log;
    if (it.iteration >= o.max_num_iterations) { c.done = 1; c.term_type = 1; c.term_reason = kMaxIter; return; }
    if (it.step_is_successful && it.gradient_max_norm <= o.gradient_tolerance) { c.done = 1; c.term_type = 0; c.term_reason = kGradTol; return; }
    if (c.radius <= o.min_radius) { c.done = 1; c.term_type = 0; c.term_reason = kMinRadius; return; }
}

// ---------------------------------------------------------------------------------------------
// operator-level kernels (not on the LM hot path)
// ---------------------------------------------------------------------------------------------
// one thread per corner: residual + Jacobian in Ceres' block layout. corner order = device order.
__global__ void k_eval_functor(DevProblem P, DevState S, const int *corner_view, double *res,
                               double *Jc, double *Jb, double *Ji)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= P.N) return;
    const int view = corner_view[k];
    const int j = k - P.view_obs[view];
    ViewConst vc;
    load_view_const(P, S, 0, P.view_cam[view], P.view_board[view], vc);
    double r[2], JE[2][kE], JF[2][kFA];
    corner_residual_jacobian(vc, P.board_xy[2 * j], P.board_xy[2 * j + 1], P.obs_u[k], P.obs_v[k], r, JE, JF);
    res[2 * k] = r[0]; res[2 * k + 1] = r[1];
    for (int row = 0; row < 2; ++row) {
        if (Jc) for (int i = 0; i < 6; ++i) Jc[12 * (size_t)k + 6 * row + i] = JF[row][i];
        if (Jb) for (int i = 0; i < 6; ++i) Jb[12 * (size_t)k + 6 * row + i] = JE[row][i];
        if (Ji) { for (int i = 0; i < 7; ++i) Ji[18 * (size_t)k + 9 * row + i] = JF[row][6 + i]; Ji[18 * (size_t)k + 9 * row + 7] = 0.0; Ji[18 * (size_t)k + 9 * row + 8] = 0.0; }
    }
}

// multi_calib.cpp:233-283: per-view sums of Euclidean pixel error and squared error, with
// cv::Rodrigues matrices and the skew projection.  one wave per view.
__global__ __launch_bounds__(64) void k_reproj_error(DevProblem P, const double *cam_rt, const double *intr,
                                                    const double *board_rt, double *view_err, double *view_sq)
{
    const int view = blockIdx.x, lane = threadIdx.x;
    const int cam = P.view_cam[view], board = P.view_board[view];
    double Rb[9], Rc[9], dummy[27], I[9];
    // cv::Rodrigues == exact Rodrigues; below DBL_EPSILON the I + [w]x branch differs by O(theta^2) ~ 1e-32
    rotation_and_derivatives(board_rt + 6 * board, Rb, dummy);
    rotation_and_derivatives(cam_rt + 6 * cam, Rc, dummy);
    for (int i = 0; i < 9; ++i) I[i] = intr[9 * cam + i];
    const double *tb = board_rt + 6 * board + 3, *tc = cam_rt + 6 * cam + 3;
    double e = 0.0, sq = 0.0;
    for (int j = lane; j < P.view_count[view]; j += 64) {
        const double x = P.board_xy[2 * j], y = P.board_xy[2 * j + 1];
        double q[3], Pc[3];
        for (int i = 0; i < 3; ++i) q[i] = Rb[3 * i] * x + Rb[3 * i + 1] * y + tb[i];
        for (int i = 0; i < 3; ++i) Pc[i] = Rc[3 * i] * q[0] + Rc[3 * i + 1] * q[1] + Rc[3 * i + 2] * q[2] + tc[i];
        double u, v;
        project_point(I, Pc[0], Pc[1], Pc[2], u, v);
        const double du = P.obs_u[P.view_obs[view] + j] - u, dv = P.obs_v[P.view_obs[view] + j] - v;
        e += sqrt(du * du + dv * dv); sq += du * du + dv * dv;
    }
    for (int s = 32; s > 0; s >>= 1) { e += __shfl_xor(e, s); sq += __shfl_xor(sq, s); }
    if (lane == 0) { view_err[view] = e; view_sq[view] = sq; }
}

__global__ void k_project(const double *intr, const double *pts, int n, double *uv)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double I[9];
    for (int k = 0; k < 9; ++k) I[k] = intr[k];
    project_point(I, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], uv[2 * i], uv[2 * i + 1]);
}

__global__ void k_unproject(const double *intr, const double *uv, int n, double *rays)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double I[9], r[3];
    for (int k = 0; k < 9; ++k) I[k] = intr[k];
    unproject_pixel(I, uv[2 * i], uv[2 * i + 1], r);
    rays[3 * i] = r[0]; rays[3 * i + 1] = r[1]; rays[3 * i + 2] = r[2];
}

}  // namespace tscm

// tscm_kernels.h -- device-side data layout and HIP kernels of the TSCM LM solver (gfx950).
//
// HBM layout (all fp64, device-resident for the whole solve):
//   obs_u/obs_v      SoA corner observations, re-packed so the views of one camera are
//                    contiguous and sorted by board: a wave streams them with unit stride.
//   rec[2]           per-view Schur pieces written by the Gram kernel, two regions in one allocation:
//                      W region [V][84]  E^T [F | r]   COLUMN-major: 14 columns (F order, column 13 = E^T r) x 6 rows
//                      E region [V][18]  E^T E_wb      3 columns (the w_b columns) x 6 rows: rows 0-2 w_b x w_b,
//                                                      rows 3-5 t_b x w_b
//                      G region [V][6]   E^T r         column 13 of W once more, compact (board statistics)
//                    The t_b x t_b block of E^T E is not stored: it is (E_tb^T F_tc) R_c, three FMAs per entry from
//                    the t_c columns of W and the camera rotation the view was evaluated with (tb_tb; cconst is
//                    double-buffered like the records for that reason).
//                    views in board-major slot order; the Schur-complement and back-substitution kernels stream
//                    the W region, the e-block factorisation and the board statistics read the E region and two
//                    pieces of W.  double buffered: index ctrl->cur = system at x, cur^1 = candidate.
//   H_stage          per camera a 16x16 tile [F | r]^T [F | r]  (13x13 Gram, col 13 =
//                    F^T r, [13][13] = r^T r) + 8 scalars + one gradient-max slot per rank; fixed address so that the
//                    exchange back-end can all-reduce it without knowing the device-side buffer index.
//   fac[B][56]       per-board e-block factor (k_schur_gram; k_schur_factor for boards seen by > 3 cameras): multipliers L_ik / L_ii and
//                    s_i / L_ii, L itself for the back-substitution, 1 / L_ii, z = L^-1 S_b E^T r, the damping D^2.
//                    Y = L^-1 S_b W is never stored: k_schur_gram / k_backsub_prep re-derive it from W (21 FMAs a column).
//   T[n_bids][256]   Schur complement sum_b Y_b^T Y_b, one 16x16 tile per camera pair that shares a board.
// The LM control state (trust-region radius, accept/reject, termination, iteration log)
// lives in `Ctrl` in device memory; every kernel starts with `if (ctrl->done) return`.
#pragma once

#include "tscm_math.h"
#include "tscm_fastmath.h"
#include "tscm_nd_plan.h"

namespace tscm {

constexpr int kRecW = 84;          // doubles per view in the W region: E^T [F | r], 14 columns x 6 rows, column-major
constexpr int kRecE = 18;          // doubles per view in the E region: E^T E_wb, 3 columns x 6 rows, column-major
constexpr int kRecG = 6;           // doubles per view in the G region: E^T r once more, compact (round 5): the board statistics read 48
                                   // contiguous bytes per view instead of the last 48 of every 672-byte W record
constexpr int kRec = kRecW + kRecE + kRecG;  // doubles per view over the regions (allocation size, offset limits)
constexpr int kWcolTc = 3;         // W columns of t_c: F index 3, 4, 5 (the gradient column E^T r is F index kFR = 13)
// per-board factor record (doubles)
constexpr int kFac = 56;
constexpr int kFacM = 0;           // [15] L_ik / L_ii, i > k, packed i (i - 1) / 2 + k
constexpr int kFacC = 15;          // [6]  s_i / L_ii
constexpr int kFacL = 21;          // [15] L_ik, same packing (back-substitution)
constexpr int kFacI = 36;          // [6]  1 / L_ii
constexpr int kFacZ = 42;          // [6]  z = L^-1 S_b E^T r
constexpr int kFacD = 48;          // [6]  D^2 (damping of the scaled block)
constexpr int kTcols = 15;         // columns of the single MFMA Gram tile (see k_eval_gram)
// Tile columns (= tile rows): 0-2 w_b | 3 t_c0 | 4-6 w_c | 7 t_c1 | 8 f* | 9 one* | 10 xi | 11 t_c2 | 12 lambda | 13 alpha |
// 14 r | 15 zero.  The accumulator of v_mfma_f64_16x16x4 keeps rows kq + 4 * reg of column col in lane (col, kq): with
// the t_c rows at 3, 7, 11 ONE lane (kq = 3) holds all three of them in registers 0, 1, 2 -- the t_b rows
// (J_tb = J_tc R_c) are nine FMAs with scalar operands there, no cross-lane traffic -- and the w_b rows 0, 1, 2 sit in
// register 0 of the lanes kq = 0, 1, 2.
constexpr int kTcWb = 0, kTcWc = 4, kTcF = 8, kTcOne = 9, kTcXi = 10, kTcLam = 12, kTcAl = 13, kTcR = 14;
__host__ __device__ constexpr int tc_tc(int j) { return 3 + 4 * j; }
constexpr int kVConst = 27;        // per-view constants: R_c r1, R_c r2, R_c t_b + t_c (board point -> camera frame in two FMAs per
                                   // component), then R_c dR_b/dw_k [:,0:2]
constexpr int kCConst = 48;        // per-camera constants: [0,9) R_c, [9,12) t_c, [12,21) a_k (dR_c/dw_k = [a_k]x R_c), [21,24) w if the
                                   // rotation is in the small-angle branch else 0, [24] 1 / 0 for that branch, [39,47) fx fy cx cy xi lambda
                                   // beta 1/(1-alpha)^2   (camera_rotation_constants, tscm_math.h)
constexpr int kCStride = 72;       // doubles per camera record in cconst: 48 doubles, then the same 48 values as floats
constexpr int kCst = 80;           // LDS constant block: [0,27) view, [27,75) camera
constexpr int kScal = 8;           // scalars appended to H_stage
constexpr int kStStride = 16;      // doubles between the board-statistics partials of two workgroups: a 128-byte line each (written by ONE workgroup: see k_schur_gram<NV, true>)
constexpr int kCamSl = 16;         // the per-camera tile reduction runs in slices of 32 of the 512 raw entries: C * kCamSl workgroups
constexpr int kSmallBids = 36;     // camera-pair blocks of a rig of <= 8 cameras (8 + 28): their partial-tile ranges travel as kernel arguments
constexpr int kMaxCamLds = 8;      // n_pad = 16*C <= 128: reduced system solved in registers/LDS (k_solve_reduced)
constexpr int kMaxCam = 32;        // larger rigs: k_solve_reduced_big factors the system in global memory (n_pad <= 512)
constexpr int kMaxLog = 256;

// (PHASE_STAMP, TL_*, PH_ONLY, KTL*: tscm_instrument.h -- every line of instrumentation in the kernels goes through its macros)
#include "tscm_instrument.h"

// LDS hand-off inside ONE wave (64-thread workgroups): DS operations of a wave are serviced in
// issue order, so no s_barrier / vmcnt(0) drain is needed -- only the compiler must keep the
// program order of the LDS accesses.  (__syncthreads() would also drain the global prefetches.)
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Buffer addressing (SGPR descriptor + 32-bit VGPR offset + SGPR offset): the hot kernels keep no 64-bit
// per-lane addresses in registers.  Out-of-range offsets are dropped / read as zero by the hardware.
typedef int v2i __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, size_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)(bytes > 0xffffffffull ? 0xffffffffu : (unsigned)bytes), 0x00020000);
}
__device__ __forceinline__ double buf_load_f64(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0));
}
#ifndef TSCM_STORE_AUX
#define TSCM_STORE_AUX 0      // cache policy of the record stores.  sc1 (16: written through) takes 0.7 us off the Gram kernel and nothing off
                              // the iteration, and WRITE_SIZE goes from 33.6 to 59.7 MB per launch (partial lines are no longer merged in the
                              // L2); nt (2) costs the consumers more than it saves the producer: both measured, neither kept
#endif
__device__ __forceinline__ void buf_store_f64(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, double v)
{
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, v), r, (int)voff, (int)soff, TSCM_STORE_AUX);
}

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ d2 buf_load_2f64(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    return __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ void buf_store_2f64(__amdgpu_buffer_rsrc_t r, unsigned voff, double a, double b)
{
    const d2 v = { a, b };
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, v), r, (int)voff, 0, TSCM_STORE_AUX);
}

// Wave priority rule of k_eval_gram (see there; s_setprio takes an immediate, p is wave-uniform)
#ifndef TSCM_PRIO
#define TSCM_PRIO 10
#endif
// A/B switches of round 6 (experiment builds only: make variant EXTRA=-D...=0)
#ifndef TSCM_AB_GUARD
#define TSCM_AB_GUARD 1
#endif
#ifndef TSCM_AB_DONE_POLL
#define TSCM_AB_DONE_POLL 1
#endif
__device__ __forceinline__ void set_prio(int p)
{
    switch (p & 3) {
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    default: __builtin_amdgcn_s_setprio(3); break;
    }
}

// record regions (V = views of this rank)
__device__ __forceinline__ const double *rec_w(const double *rec, int slot) { return rec + (size_t)kRecW * slot; }
__device__ __forceinline__ const double *rec_e(const double *rec, int V, int slot) { return rec + (size_t)kRecW * V + (size_t)kRecE * slot; }
__device__ __forceinline__ const double *rec_g(const double *rec, int V, int slot) { return rec + (size_t)(kRecW + kRecE) * V + (size_t)kRecG * slot; }

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

enum TermReason { kNone = 0, kMaxIter, kGradTol, kMinRadius, kParamTol, kFuncTol, kInvalidSteps, kRanksDisagree };
// ctrl->fault (sticky): 1 = a device-side hand-off came late, 2 = the ranks of a communicator disagree about the LM state (decision_word)
constexpr int kFaultHandoff = 1, kFaultRanksDisagree = 2;

struct CtrlHead {
    // ---- header (polled by the host) ----
    int done, term_type, term_reason, iteration;
    int cur, lin_fail, num_successful, num_unsuccessful;
    int num_invalid, n_log, lm_iterations, fin_count;   // fin_count: arrival counter of k_reduce_control
    int fault, pad0;                    // fault: a device-side hand-off timed out (sticky; the host turns it into TSCM_E_HIP)
    double radius, decrease_factor;
    double x_cost, x_norm, gmax, gnorm;
    double model_cam, stepsq_cam;
    double se_min, se_cur, se_ref, se_cand, se_acc_ref, se_acc_cand;
    double initial_cost;
    Options opt;
    long long t_begin, t_end;           // s_memrealtime (100 MHz) in k_begin_solve / k_end_solve: device time of the solve
};

struct Ctrl : CtrlHead {
    IterLog log[kMaxLog];
};

struct DevProblem {
    int C, B, n_points, V, N, n_pad;
    int n_chunks, n_pairs, n_pchunks, n_bids;
    int rp, half;                      // LDS pitch (doubles) and rows of the Jacobian tile
    int lds_wave;                      // doubles of LDS per wave of k_eval_gram
    const double *board_xy;
    const int *view_cam, *view_board, *view_obs, *view_count;
    const double *obs_u, *obs_v;
    const int *chunk_vb, *chunk_ve, *chunk_cam, *cam_chunk_ptr;
    const int4 *chunk_desc;            // per chunk of the Gram kernels: camera, first view, end view, observation offset of the first view
    const int *bv_ptr;                 // board -> range of view SLOTS (records are stored board-major)
    const int *view_slot, *slot_cam;   // device view -> slot ; slot -> camera
    const int *slot_view, *slot_board; // slot -> device view ; slot -> board
    const int *slow_boards;            // boards seen by more than three cameras (factored by k_schur_factor, Gram by k_pair_gram)
    int n_slow;
    const int *pair_i, *pair_j;
    const int *pc_begin, *pc_end, *pc_tile, *bid_mi, *bid_mj;
    const int *bid_part_ptr;                   // per camera-pair block: contiguous range of its partial tiles in pairpart
    const int *sslot;                          // first view slot of each board, boards grouped by camera-set signature
    const int *sboard;                         // ... and the board itself (its factor record)
    const int *pair_board;                     // board of each fallback view pair
    const int *bc_begin, *bc_end, *bc_nv, *bc_tile;   // board chunks: range in sslot, views per board, tile ids [chunk*6 + t]
    const int4 *bc_desc;                       // the same per chunk in one 16-byte record: first board, end board, first slot, views per board
                                               // (device boards are numbered in signature order: a chunk's boards AND slots are contiguous)
    int n_bchunks, n_tiles;
    const unsigned char *cam_const, *cam_active;
    const unsigned char *board_const;  // [B] device board: pose block held constant (tscm_problem.board_pose_constant)
    const unsigned char *col_active;   // [n_pad] 1 = column is a free camera-side parameter
    const int *act_map;                // [n_pad] compact index -> padded column (first n_act entries)
    int n_act;
    // the compact numbering of the free camera-side columns in closed form for k_solve_reduced (<= 4 cameras): the free columns of
    // camera q are compact [cam_pre[q], cam_pre[q + 1]) = padded cam_col0[q] + 0, 1, ...  (cam_pre[q] = n_act from q = C on)
    int cam_pre[9], cam_col0[8];
    unsigned long long pair_mask;      // bit mi * 8 + mj: the camera pair shares a board (its tile of T follows by a population count)
    const int4 *solve_map;             // [kSolveMapSlots / 4][256] operand offsets of every thread of k_solve_reduced (k_solve_map)
    int cam_wg[9];                     // cam_chunk_ptr by value for rigs of <= kMaxCamLds cameras (k_reduce_control: no index load in front of the tiles)
    // frame sharding (tscm_solver_create_sharded): this rank / number of ranks; 0 / 1 on a single GPU
    int rank, world;
    // T is stored compact: one 16x16 tile per camera-pair block (mi <= mj) that ANY rank contributes to, numbered in
    // lexicographic (mi, mj) order -- the same list on every rank, so the all-reduce is over n_bids * 256 doubles
    // and every tile is rewritten in full each iteration.
    const short *bid_lut;              // [C*C] tile of block (mi, mj), mi <= mj; -1 = no board is seen by both
    int bid_part_small[kSmallBids + 1]; // bid_part_ptr by value (rigs of <= kMaxCamLds cameras: no memory round trip in front of the partial tiles)
    int g4_per;                        // k_eval_gram4<KS, true>: corners of a pass (boards of more than 56 corners: g4_plan)
    int g4s_ksv;                       // k_eval_gram4s: k-steps of a view, ceil(n_points / 4)
};

struct DevState {
    double *cam_rt[2], *intr[2], *board_rt[2];
    double *board_pc, *cam_pc;
    double *vconst;
    double *cconst[2];                 // per-camera constants of the point the records of the same index were evaluated at
    double *rec[2];
    double *campart, *campart2;
    double *H[2], *H_stage;
    double *s_b, *s_c;
    double *fac;                       // [B][kFac] e-block factors
    double *pairpart, *T;
    int *t_count;                      // arrival counter of the fused T reduction + reduced solve (k_solve_reduced<..., true>)
    int *fac_fail;                     // set by an e-block factorisation that failed (k_schur_gram / k_schur_factor); read and cleared by the reduced
                                       // solve (outside the control block: the control step may rewrite that block while the factorisations run)
    int *y_flag;                       // 2 * epoch + lin_fail once the camera step of that fused launch is written (backsub_body<.., true> waits for it)
    double *yhat;
    double *Abig;                      // compact reduced system + rhs row in 16x16 blocks, rigs of more than kMaxCamLds cameras only
    double *bs_part, *st_part;
    int n_bs_blocks, n_st_blocks;
    Ctrl *ctrl;
    CtrlHead *ctrl_snap;               // copy of the control block's head taken by k_reduce_stats: what the control step in the head of the
                                       // NEXT launch (k_schur_gram, every workgroup) reads while that launch's writer workgroup advances `ctrl`
    int *stats_count, *stats_flag;     // k_schur_gram<NV, true>: arrivals of its reduction workgroups, counted over the solve; the count the last arrival of a launch
                                       // found, in a line of its own (what the waiting workgroups poll: loads there, read-modify-writes here)
    struct CtlPub *ctl_pub;            // outcome of that step, published by the writer workgroup for the workgroups of later rounds of the grid
};
// epoch: number of control steps taken in k_schur_gram's head in this solve so far (monotonic, zeroed by k_begin_solve)
struct CtlPub { int epoch, cur, done, pad; double radius; };

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
// per-view / per-camera constants of the evaluation target in the form the hot kernel consumes.
// grid ceil((V + C)/128) x 128.
//   vconst[view][32]: r1(3) r2(3) t_b(3), then for k=0..2: R_c dR_b/dw_k[:,0] (3), R_c dR_b/dw_k[:,1] (3); 5 pad
//   cconst[cam] : R_c(9) t_c(3) dR_c/dw_k (27) fx fy cx cy xi lambda beta=alpha/(1-alpha) 1/(1-alpha)^2
// ---------------------------------------------------------------------------------------------
// the first nine per-view constants: the board point (x, y, 0) in the camera frame is x m1 + y m2 + t
__device__ __forceinline__ void view_point_constants(const double Rc[9], const double *tc, const double *bc /* r1, r2 */, const double *tb, double *o)
{
    for (int r = 0; r < 3; ++r) {
        o[r] = Rc[3 * r] * bc[0] + Rc[3 * r + 1] * bc[1] + Rc[3 * r + 2] * bc[2];
        o[3 + r] = Rc[3 * r] * bc[3] + Rc[3 * r + 1] * bc[4] + Rc[3 * r + 2] * bc[5];
        o[6 + r] = (Rc[3 * r] * tb[0] + Rc[3 * r + 1] * tb[1] + Rc[3 * r + 2] * tb[2]) + tc[r];
    }
}

// Device-side hand-offs (many producer workgroups -> the workgroup that consumes their results in the SAME launch).
// The textbook form -- plain stores, release fence, counter; counter, acquire fence, plain loads -- makes every producer
// issue a `buffer_wbl2` (the agent-scope release fence writes its XCD's L2 back).  Measured with tools/kernel_timeline.py:
// with the fences the producers of k_reduce_control ended 4.9 us (143 workgroups, config 4) and 16 us (441, config 5)
// after the kernel's first start, whatever they computed.  Here the handed-over values are written THROUGH instead
// (agent-scope stores: `global_store ... sc1`), a producer waits for their completion (`s_waitcnt vmcnt(0)`, then the
// workgroup barrier) and only then counts itself in; the consumer reads them with agent-scope loads (`sc1`: not from its
// own XCD's L2).  No L2 write-back anywhere: 4.0 / 6-8 us, the iteration 130.1 -> 127.9 us (config 4), 393.8 -> 381.2
// (config 5), same bits.  Everything else a kernel writes stays an ordinary store and reaches the next kernel through
// the kernel boundary as before.  (Counting the arrivals in two levels, sixteen workgroups per counter, was slower:
// contention on the single counter is not what the producers wait for.)
//
// What the ordering rests on.  EVERY handed-over location is written with an agent-scope atomic store and read with an
// agent-scope atomic load -- no plain access to it on either side inside the launch that hands it over -- so in the
// language's terms there is no data race; what the relaxed orders leave open is only the ORDER between the data and the
// flag.  That order is supplied by the machine, in the way the AMDGPU back-end itself implements a release on
// gfx942 / gfx950 ("buffer_wbl2 sc1; s_waitcnt vmcnt(0)" in front of the flag's store): the write-back is there for PLAIN
// stores that may still sit in the XCD's L2; an sc1 store is written through, and its vmcnt slot is returned when the write
// has reached the level all XCDs share.  `s_waitcnt vmcnt(0)` + workgroup barrier + flag is therefore the release
// sequence minus the part that has nothing to do here.  On the consumer side the sc1 loads do not hit in the L1 / the
// XCD's L2, so no `buffer_inv` is needed for THESE loads (an acquire fence would issue one per wave: 20 us for the 2,500
// waves that wait for the camera step).  This is a property of the gfx942 / gfx950 cache hierarchy, not of HIP:
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "the in-launch hand-offs (handoff_store / handoff_load) rely on gfx942 / gfx950 sc1 write-through semantics: re-derive them for this target"
#endif
// The locations handed over inside a launch -- a new read of producer-written data in a waiting workgroup MUST go
// through handoff_load, a new write of consumer-read data through handoff_store:
//   T[n_bids][256]                    t_reduce_block (T producers)        -> k_solve_nd, solver workgroup        flag: t_count
//   yhat[n_pad]                       reduced_solution_tail (solver)      -> backsub_body<.., true>              flag: y_flag
//   cam_rt[cur^1], intr[cur^1]        reduced_solution_tail (solver)      -> backsub_body<.., true> (phase B)    flag: y_flag
//   ctrl->done / fault / term_type    solver or a waiting workgroup (late hand-off) -> waiting workgroups        (atomics both sides)
//   campart2[C][512], st_part[..][3]  cam_reduce_block / board_stats_block -> k_reduce_control's last workgroup  flag: ctrl->fin_count
//   campart2, st_part, ctrl_snap      the reduction blocks riding in k_schur_gram<NV, true> -> every workgroup's control step   flag: stats_flag
//                                     (read with PLAIN loads behind the flag: single-writer lines, see k_schur_gram)
// (the solver workgroup ALSO reads cam_rt / intr of the candidate with plain loads in write_camera_record: its own
// written-through stores, program order within one workgroup, never cached in its L1 before).
// A hand-off that has not come after this long is a device fault, not a numerical event: the solver workgroup sets the
// sticky ctrl->fault together with ctrl->done (every later kernel of the stream exits at once) and the host returns
// TSCM_E_HIP.  s_memrealtime ticks: 100 MHz whatever the shader clock does.
constexpr long long kHandoffTimeoutTicks = 50 * 1000 * 1000;          // 0.5 s
__device__ __forceinline__ void handoff_store(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double handoff_load(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// the per-camera record of the evaluation target (doubles, then the same values as floats)
__device__ __forceinline__ void write_camera_record(const DevState &S, int tgt, int m, const double *cam_rt, const double *intr)
{
    double crt[3], Rc[9], a[9], wsm[3];
    for (int k = 0; k < 3; ++k) crt[k] = cam_rt[6 * m + k];
    const int small = camera_rotation_constants(crt, Rc, a, wsm);
    double *o = S.cconst[tgt] + kCStride * m;
    for (int k = 0; k < 9; ++k) o[k] = Rc[k];
    for (int k = 0; k < 3; ++k) o[9 + k] = cam_rt[6 * m + 3 + k];
    for (int k = 0; k < 9; ++k) o[12 + k] = a[k];
    for (int k = 0; k < 3; ++k) o[21 + k] = wsm[k];
    o[24] = small ? 1.0 : 0.0;
    for (int k = 25; k < 39; ++k) o[k] = 0.0;
    double I[7];
    for (int k = 0; k < 7; ++k) I[k] = intr[9 * m + k];
    for (int k = 0; k < 6; ++k) o[39 + k] = I[k];
    const double oma = 1.0 - I[6];
    o[45] = I[6] / oma;
    o[46] = 1.0 / (oma * oma);
    o[47] = 0.0;
    float *of = reinterpret_cast<float *>(o + kCConst);
    for (int k = 0; k < kCConst; ++k) of[k] = (float)o[k];
}
__device__ __forceinline__ void write_camera_record(const DevState &S, int tgt, int m) { write_camera_record(S, tgt, m, S.cam_rt[tgt], S.intr[tgt]); }

constexpr int kVStride = 48;      // doubles per view record in vconst: 27 doubles (+5 pad), then at byte 256 the same 27 values as
                                  // floats (read by the fp32-Jacobian kernel): 384 bytes
constexpr int kVFloatOff = 32;    // offset of the float copy, in doubles
constexpr int kVPrepThreads = 128;

// (the point's parameters come from explicit arrays: buffer `tgt` -- or, in the first launch of a solve, the registered start point)
__device__ __forceinline__ void view_prep_body(const DevProblem &P, const DevState &S, int tgt, int with_floats, const double *cam_rt, const double *intr, const double *board_rt)
{
    __shared__ double st[kVPrepThreads][kVFloatOff + 1];    // the 27 (+5 pad) doubles of thread t in row t (pitch 33: conflict-free both ways)
    const int t = threadIdx.x;
    const int i = blockIdx.x * kVPrepThreads + t;
    if (i < P.V) {
        // self-contained (rotations recomputed per view: cheaper than a second launch + round trip)
        const int b = P.view_board[i], m = P.view_cam[i];
        double rt[6], bc[kBoardConst], Rc[9], dRc[27];
        for (int k = 0; k < 6; ++k) rt[k] = board_rt[6 * b + k];
        board_constants(rt, bc);
        double crt[3];
        for (int k = 0; k < 3; ++k) crt[k] = cam_rt[6 * m + k];
        rotation_and_derivatives(crt, Rc, dRc);
        double *o = st[t];
        view_point_constants(Rc, cam_rt + 6 * m + 3, bc, rt + 3, o);
        for (int k = 0; k < 6; ++k) {           // six 3-vectors d -> R_c d
            const double d0 = bc[6 + 3 * k], d1 = bc[6 + 3 * k + 1], d2 = bc[6 + 3 * k + 2];
            for (int r = 0; r < 3; ++r) o[9 + 3 * k + r] = Rc[3 * r] * d0 + Rc[3 * r + 1] * d1 + Rc[3 * r + 2] * d2;
        }
        for (int k = kVConst; k < kVFloatOff; ++k) o[k] = 0.0;
    } else if (i < P.V + P.C) {
        write_camera_record(S, tgt, i - P.V, cam_rt, intr);
    }
    __syncthreads();
    // the block's records leave as one contiguous, coalesced stream: vconst[view][32]
    const int v0 = blockIdx.x * kVPrepThreads;
    const int nv = min(kVPrepThreads, P.V - v0);
    for (int e = t; e < nv * kVFloatOff; e += kVPrepThreads) {                 // compile-time divisors: shifts
        const int v = e / kVFloatOff, k = e % kVFloatOff;
        S.vconst[(size_t)kVStride * (v0 + v) + k] = st[v][k];
    }
    if (with_floats) {
        // the float half of a record (the same 27 values as floats, two per double slot) is only written for
        // the fp32-Jacobian kernel
        constexpr int kF = kVStride - kVFloatOff;
        for (int e = t; e < nv * kF; e += kVPrepThreads) {
            const int v = e / kF, k = e % kF, j = 2 * k;
            const float f0 = j < kVConst ? (float)st[v][j] : 0.f, f1 = j + 1 < kVConst ? (float)st[v][j + 1] : 0.f;
            S.vconst[(size_t)kVStride * (v0 + v) + kVFloatOff + k] = __hiloint2double(__float_as_int(f1), __float_as_int(f0));
        }
    }
}

__global__ __launch_bounds__(kVPrepThreads) void k_view_prep(DevProblem P, DevState S, int cand, int with_floats)
{
    if (S.ctrl->done) return;
    const int tgt = cand ? (S.ctrl->cur ^ 1) : S.ctrl->cur;
    view_prep_body(P, S, tgt, with_floats, S.cam_rt[tgt], S.intr[tgt], S.board_rt[tgt]);
}

// tile column -> (parity mask) bookkeeping shared by the hot kernel's epilogue and k_finalize_eval.
//   f*   = -X/k on u-rows, -Y/k on v-rows : fx is its u-half, fy its v-half
//   one* = -1 on every row                 : cx is its u-half, cy its v-half
// so with separate Gram tiles for the u-rows (GU) and the v-rows (GV) the true products are
//   <a, b> = sum over parities in mask(a) & mask(b) of G_par[tile(a)][tile(b)].
// F index (record / H layout): 0-2 w_c, 3-5 t_c, 6 fx, 7 fy, 8 cx, 9 cy, 10 xi, 11 lambda, 12 alpha, 13 r.
__host__ __device__ __forceinline__ constexpr int f_tile(int f)
{
    return f < 3 ? kTcWc + f : f < 6 ? tc_tc(f - 3) : f < 8 ? kTcF : f < 10 ? kTcOne : f == 10 ? kTcXi : f == 11 ? kTcLam : f == 12 ? kTcAl : kTcR;
}
__device__ __forceinline__ int f_mask(int f) { return (f >= 6 && f < 10) ? (1 << ((f - 6) & 1)) : 3; }

// ---------------------------------------------------------------------------------------------
// Epilogue of the Gram kernels (fp64 and fp32-Jacobian variant): one view's two accumulator tiles (u-rows, v-rows)
// -> its record, entirely in the lane's own registers.
// D layout: lane (col, kq) holds rows kq + 4*reg of tile column col (see the tile-column table above):
//   lanes kq < 3   register 0 = w_b row kq of their column              -> record row kq
//   lanes kq == 3  registers 0, 1, 2 = the t_c rows of their column     -> t_b row l = sum_j R_c[j][l] * (t_c row j),
//                  record rows 3, 4, 5 (row 3 leaves with the w_b rows in one store, rows 4 | 5 as one 16-byte store)
// The columns f* and one* store their u-row part (fx, cx) and next to it the v-row part = total - u-part (fy, cy).
// The record is column-major, so a lane's entries are adjacent: FOUR stores per view (13 in round 2, with 20
// ds_bpermute, three vector loads of R_c and ~90 integer instructions of offset arithmetic around them); lanes without
// an entry store past the end of the buffer, which the bounds check drops.  Neither the t_b x t_b block of E^T E
// (consumers derive it: tb_tb) nor its never-read upper triangle nor copies of E^T r / the diagonal are written.
// ---------------------------------------------------------------------------------------------
typedef const double __attribute__((address_space(4))) *cptr4;

// lane-constant part of the record addressing, computed once per kernel: byte offset of the lane's row-kq entry inside
// the allocation for slot 0 (region base included) and the byte stride per slot of its region
struct RecLane { unsigned off, stride, goff, gstride; };
__device__ __forceinline__ RecLane rec_lane(int lane, unsigned V)
{
    const int col = lane & 15, kq = lane >> 4;
    RecLane r;
    // the gradient column once more in the compact G region: row kq (kq < 3) / rows 3 | 4 5 (kq == 3) of the view's six
    r.goff = col == kTcR ? 8u * ((unsigned)(kRecW + kRecE) * V + (unsigned)kq) : 0xffffe000u; r.gstride = col == kTcR ? 8u * kRecG : 0u;
    if (col < 3) { r.off = 8u * ((unsigned)kRecW * V + 6u * (unsigned)col + (unsigned)kq); r.stride = 8u * kRecE; }
    else if (col == 15) { r.off = 0xffffe000u; r.stride = 0u; }
    else {
        // tile column -> F index (W column); f* and one* are the first of a (u-part, v-part) column pair
        const int f = (col & 3) == 3 ? kWcolTc + (col >> 2) : col < 8 ? col - kTcWc : col == kTcF ? 6 : col == kTcOne ? 8 : col == kTcXi ? 10 : col - 1;
        r.off = 8u * (6u * (unsigned)f + (unsigned)kq); r.stride = 8u * kRecW;
    }
    return r;
}

// t_b x t_b entry (l, lp) of one view's E^T E from the t_c columns of its W record (column-major) and the rotation of
// the camera it was evaluated with -- the SAME three operations, in the same order, the Gram kernel's epilogue used
// when it still stored the block: the values are bit-identical.
template <typename PW, typename PR>
__host__ __device__ __forceinline__ double tb_tb(PW W, PR Rc, int l, int lp)
{
    double t = Rc[3 + lp] * W[6 * (kWcolTc + 1) + 3 + l];
    t = fma(Rc[lp], W[6 * kWcolTc + 3 + l], t);
    return fma(Rc[6 + lp], W[6 * (kWcolTc + 2) + 3 + l], t);
}

__device__ __forceinline__ void store_view_record(__amdgpu_buffer_rsrc_t r_rec, int lane, const d4 &accU, const d4 &accV, cptr4 cc, unsigned slot, RecLane rl)
{
    int le = lane;
    asm volatile("" : "+v"(le));             // the lane predicates are rebuilt per view (a compare each) instead of living in SGPR pairs
    const int col = le & 15;
    const bool k3 = le >= 48, split = col == kTcF || col == kTcOne;
    constexpr unsigned BAD = 0xffffe000u;
    const unsigned off = rl.off + __umul24(slot, rl.stride);
    const double t0 = accU[0] + accV[0], t1 = accU[1] + accV[1], t2 = accU[2] + accV[2];
    // t_b rows l = 0, 1, 2 (of use in the lanes kq == 3): u+v and the u-row part
    double tbT[3], tbU[3];
#pragma unroll
    for (int l = 0; l < 3; ++l) {
        tbT[l] = fma(cc[6 + l], t2, fma(cc[l], t0, cc[3 + l] * t1));
        tbU[l] = fma(cc[6 + l], accU[2], fma(cc[l], accU[0], cc[3 + l] * accU[1]));
    }
    const double selT = k3 ? tbT[0] : t0, selU = k3 ? tbU[0] : accU[0];      // record row kq (kq < 3) / row 3 (kq == 3)
    buf_store_f64(r_rec, off, 0u, split ? selU : selT);
    buf_store_2f64(r_rec, (k3 ? off : BAD) + 8u, split ? tbU[1] : tbT[1], split ? tbU[2] : tbT[2]);
    // ... and the v-row parts of f* / one* in the next record column
    buf_store_f64(r_rec, (split ? off : BAD) + 48u, 0u, selT - selU);
    buf_store_2f64(r_rec, (split && k3 ? off : BAD) + 56u, tbT[1] - tbU[1], tbT[2] - tbU[2]);
    const unsigned og = rl.goff + __umul24(slot, rl.gstride);
    buf_store_f64(r_rec, og, 0u, selT);
    buf_store_2f64(r_rec, (k3 ? og : BAD) + 8u, tbT[1], tbT[2]);
}

// ---------------------------------------------------------------------------------------------
// THE HOT KERNEL: per-corner TSCM projection + analytic Jacobian + residual, then ALL Gram
// products of the 2 x 20 Jacobian block [E | F | r] from ONE 16x16 f64 MFMA tile per 4 rows:
//   * t_b columns are constant combinations of the t_c columns (J_tb = J_tc R_c): dropped from
//     the tile, recovered per view by a 3x3 multiply in the epilogue;
//   * (fx, fy) and (cx, cy) have disjoint row support: merged into f* and one*, separated again
//     by keeping the u-rows and the v-rows of the Jacobian in two accumulators.
//   -> 15 tile columns, v_mfma_f64_16x16x4_f64 count per view = 2*ceil(n/4) (28 for 54 corners)
//      instead of 3*ceil(2n/4) = 81 for the naive [E|F|r] padding.
// One wave per chunk of consecutive views of ONE camera, four such waves (same camera) per workgroup;
// lane = corner (coalesced SoA loads of u[], v[]); board points in LDS; the 27 per-view constants live
// one per lane in a VGPR pair (prefetched a view ahead) and are fetched with v_readlane, the 48
// per-camera constants come through the constant address space as scalar loads (SGPR operands);
// Jacobian columns are transposed through LDS (column-major, pitch 2*odd: conflict-free
// ds_read_b64) into MFMA operand layout.  The per-camera tile stays in registers for the chunk.
// All global traffic of the view loop uses buffer addressing; the record is written by 13
// unconditional stores.
// dynamic LDS: 16*rp + kCst + 2*n_points doubles (the kCst block is only used by k_eval_gram_f32).

// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// Geometry of one corner for the Gram kernels: board point (x, y, 0) -> camera frame -> Triple Sphere projection,
// residual, and the 15 Jacobian entries of the u-row and of the v-row (multi_calib.h:146-195, hand-derived: tscm_math.h).
// VC(k): the view's constants (kVConst), CC(k): the camera's (kCConst), both wave-uniform (scalar operands);
// PUT(column, u, v) receives the entries by SEMANTIC column (GCol) -- each kernel has its own tile column order.
//   * P_c = x m1 + y m2 + t as two FMAs per component (round 3; rounds 1-2: board -> world -> camera, 21 operations);
//   * camera-rotation columns: n . (a_k x Q') = a_k . (Q' x n) with Q' = P_c - t_c (camera_rotation_constants):
//     one cross product per row and three dot products (33 operations; 45 with three matrix-vector products).
// ---------------------------------------------------------------------------------------------
enum GCol { gcWb0 = 0, gcWb1, gcWb2, gcTc0, gcTc1, gcTc2, gcWc0, gcWc1, gcWc2, gcF, gcOne, gcXi, gcLam, gcAl, gcR };

template <typename FV, typename FC, typename FP>
__device__ __forceinline__ void corner_geometry(double x, double y, double ou, double ov, FV VC, FC CC, FP PUT)
{
    const double X = fma(x, VC(0), fma(y, VC(3), VC(6)));
    const double Y = fma(x, VC(1), fma(y, VC(4), VC(7)));
    const double Z = fma(x, VC(2), fma(y, VC(5), VC(8)));
    const double fx = CC(39), fy = CC(40), xi = CC(43), lam = CC(44), beta = CC(45);
    // triple sphere (multi_calib.h:170-178)
    const double rho2 = X * X + Y * Y;
    double d1, id1, d2, id2, d3, id3;
    sqrt_and_inverse(rho2 + Z * Z, d1, id1);
    const double z1 = Z + xi * d1;
    sqrt_and_inverse(rho2 + z1 * z1, d2, id2);
    const double z2 = z1 + lam * d2;
    sqrt_and_inverse(rho2 + z2 * z2, d3, id3);
    const double k = z2 + beta * d3;
    const double ik = fast_rcp(k);
    const double mx = X * ik, my = Y * ik;
    const double c1 = 1.0 + xi * Z * id1;
    const double c2 = 1.0 + lam * z1 * id2;
    const double c3 = 1.0 + beta * z2 * id3;
    const double q = beta * id3 + c3 * (lam * id2 + c2 * xi * id1);
    const double kz = c1 * c2 * c3;
    const double fxk = fx * ik, fyk = fy * ik;
    // -A = -d(u,v)/dPc  (the t_c columns)
    const double n00 = -fxk * (1.0 - X * mx * q), n01 = fxk * mx * Y * q, n02 = fxk * mx * kz;
    const double n10 = fyk * my * X * q, n11 = -fyk * (1.0 - Y * my * q), n12 = fyk * my * kz;
    PUT(gcTc0, n00, n10);
    PUT(gcTc1, n01, n11);
    PUT(gcTc2, n02, n12);
    // w_b: -A (x e_k0 + y e_k1),  e = R_c dR_b/dw_k columns
#pragma unroll
    for (int kk = 0; kk < 3; ++kk) {
        const double h0 = x * VC(9 + 6 * kk) + y * VC(12 + 6 * kk);
        const double h1 = x * VC(10 + 6 * kk) + y * VC(13 + 6 * kk);
        const double h2 = x * VC(11 + 6 * kk) + y * VC(14 + 6 * kk);
        PUT(gcWb0 + kk, n00 * h0 + n01 * h1 + n02 * h2, n10 * h0 + n11 * h1 + n12 * h2);
    }
    // w_c: -A (dR_c/dw_k P_w) = a_k . (Q' x n)
    {
        double Q0 = X - CC(9), Q1 = Y - CC(10), Q2 = Z - CC(11);
        if (CC(24) != 0.0) {                  // small-angle branch of the camera rotation (wave-uniform): Q' = Q - w x Q
            const double w0 = CC(21), w1 = CC(22), w2 = CC(23);
            const double s0 = w1 * Q2 - w2 * Q1, s1 = w2 * Q0 - w0 * Q2, s2 = w0 * Q1 - w1 * Q0;
            Q0 -= s0; Q1 -= s1; Q2 -= s2;
        }
        const double cu0 = Q1 * n02 - Q2 * n01, cu1 = Q2 * n00 - Q0 * n02, cu2 = Q0 * n01 - Q1 * n00;
        const double cv0 = Q1 * n12 - Q2 * n11, cv1 = Q2 * n10 - Q0 * n12, cv2 = Q0 * n11 - Q1 * n10;
#pragma unroll
        for (int kk = 0; kk < 3; ++kk)
            PUT(gcWc0 + kk, CC(12 + 3 * kk) * cu0 + CC(13 + 3 * kk) * cu1 + CC(14 + 3 * kk) * cu2,
                            CC(12 + 3 * kk) * cv0 + CC(13 + 3 * kk) * cv1 + CC(14 + 3 * kk) * cv2);
    }
    // f* and one*
    PUT(gcF, -mx, -my);
    PUT(gcOne, -1.0, -1.0);
    // xi, lambda, alpha: -du/dk * dk/dparam
    const double hu = fxk * mx, hv = fyk * my;
    const double kxi = c3 * c2 * d1, klam = c3 * d2, kal = d3 * CC(46);
    PUT(gcXi, hu * kxi, hv * kxi);
    PUT(gcLam, hu * klam, hv * klam);
    PUT(gcAl, hu * kal, hv * kal);
    // residual = observed - projected (multi_calib.h:192-193)
    PUT(gcR, ou - (fx * mx + CC(41)), ov - (fy * my + CC(42)));
}

#include "tscm_instrument.h"       // (second pass: the timeline buffers, behind CtrlHead and wall_clock64)
#ifndef TSCM_EXP
#define TSCM_EXP 3     // bit 0: full tiles through gram_full (0 = round 2's paired loop, for A/B runs), bit 1: first MFMA with C = 0
#endif
#ifndef TSCM_GRAM_DEPTH
#define TSCM_GRAM_DEPTH 4
#endif
// Gram contraction of one row half of a view, full tile of KS k-steps: every operand is its OWN ds_read_b64 (serviced
// in two 32-lane groups with banks mod 64: conflict-free at a pitch of 2 * odd), requested D steps ahead of the MFMA
// that consumes it.  The compiler fuses neighbouring plain loads into ds_read2_b64, which is serviced in 16-lane groups
// with banks mod 32 -- two-way conflicts at this pitch, 16 LDS cycles instead of 4 per pair: the source of
// SQ_LDS_BANK_CONFLICT in round 2's counters -- and waits for each pair right after requesting it.  Hence inline
// assembly with explicit counts: LDS operations of a wave complete in order, so after lgkmcnt(n) everything but the n
// youngest requests has arrived whatever else (scalar loads included) is in flight.
template <int OFF>
__device__ __forceinline__ double ds_read_f64(unsigned addr)
{
    double v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
template <int N>
__device__ __forceinline__ void lgkm_wait(double &v) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v) : "n"(N)); }
__device__ __forceinline__ unsigned lds_addr(const double *p)
{
    return (unsigned)(size_t)(const __attribute__((address_space(3))) double *)p;
}
template <int KS, int D, bool ZERO_C, int T = 0>
__device__ __forceinline__ void gram_steps(unsigned addr, double (&a)[KS], d4 &acc)
{
    if constexpr (T < KS) {
        if constexpr (T + D < KS) a[T + D] = ds_read_f64<32 * (T + D)>(addr);
        lgkm_wait<(KS - 1 - T < D ? KS - 1 - T : D)>(a[T]);
        if constexpr (ZERO_C && T == 0) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[T], a[T], d4{ 0.0, 0.0, 0.0, 0.0 }, 0, 0, 0);
        else acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[T], a[T], acc, 0, 0, 0);
        gram_steps<KS, D, ZERO_C, T + 1>(addr, a, acc);
    }
}
template <int KS, int D, bool ZERO_C, int T = 0>
__device__ __forceinline__ void gram_prime(unsigned addr, double (&a)[KS])
{
    if constexpr (T < D && T < KS) { a[T] = ds_read_f64<32 * T>(addr); gram_prime<KS, D, ZERO_C, T + 1>(addr, a); }
}
template <int KS, bool ZERO_C>
__device__ __forceinline__ void gram_full(const double *fp, d4 &acc)
{
    constexpr int D = TSCM_GRAM_DEPTH;
    const unsigned addr = lds_addr(fp);
    double a[KS];
    gram_prime<KS, D, ZERO_C>(addr, a);
    gram_steps<KS, D, ZERO_C>(addr, a, acc);
}

template <int RPC>   // RPC > 0: compile-time LDS pitch (HV = RPC - 2): all tile offsets become immediates
__global__ __launch_bounds__(256, 4) void k_eval_gram(DevProblem P, DevState S, int cand)
{
    // profiling aid, COMPILE TIME only (make ABLATE=n: -DTSCM_ABLATE=n; 1 = no MFMA loops, 2 = no epilogue, 4 = no
    // geometry; results invalid).  The release kernel carries neither the argument nor the branches.
#ifdef TSCM_ABLATE
    constexpr int ablate = TSCM_ABLATE;
#else
    constexpr int ablate = 0;
#endif
    // the control block is read together with the static chunk tables (one memory round trip, not two);
    // the early exit is taken right before the first view
    const int ctrl_done = S.ctrl->done, ctrl_cur = S.ctrl->cur;
    TL_ONLY(
    // profiling builds only (make EXTRA=-DTSCM_WAVE_TIMELINE): start / end time and hardware slot of every wave of the
    // launches of LM iteration 5 into g_timeline; tscm_debug_wave_timeline copies it out, tools/wave_timeline.py groups
    // the waves by SIMD
    const long long tl_t0 = wall_clock64();
    const int tl_iter = S.ctrl->iteration;
    )
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: keeps chunk/view/cnt in SGPRs
    double *lds = lds_all + (size_t)wave * P.lds_wave;     // every wave works in its own LDS region
    const int RP = RPC > 0 ? RPC : P.rp, HV = RPC > 0 ? RPC - 2 : P.half;   // pitch = 2*odd >= HV: conflict-free ds_read_b64
    double *Fl = lds;                          // [kTcols][RP]: HV rows; holds the u-rows, then the v-rows
    double *cst = Fl + max(16 * RP, 512);      // [kCst]  (column 15 of Fl stays zero; 512 = final camera-tile exchange)
    double *bxy = cst + kCst;
    const int lane = threadIdx.x & 63;
    const int chunk = blockIdx.x * 4 + wave;
    const int cam = P.chunk_cam[chunk];
    for (int i = lane; i < 2 * P.n_points; i += 64) bxy[i] = P.board_xy[i];
    const int vb = P.chunk_vb[chunk], ve = P.chunk_ve[chunk];
    const int col = lane & 15, kq = lane >> 4;
    d4 camU = { 0.0, 0.0, 0.0, 0.0 }, camV = { 0.0, 0.0, 0.0, 0.0 };
    // rows of lanes without a corner are kept at zero instead of being re-written every pass
    if (lane < HV) {
#pragma unroll
        for (int c = 0; c < 16; ++c) Fl[c * RP + lane] = 0.0;      // incl. the all-zero 16th tile column
    }
    int prev_nv = 0;                           // lanes [nv, prev_nv) hold stale rows of the previous pass
    // software prefetch: the next view's constants and first 64 observations are loaded while
    // the current view computes (one wave per SIMD-slot cannot hide HBM latency otherwise)
    double pf_u = 0.0, pf_v = 0.0;
    int warm = 0;                              // see the prefetch below
    if (ctrl_done) return;
    const int tgt = cand ? (ctrl_cur ^ 1) : ctrl_cur;
    const __amdgpu_buffer_rsrc_t r_rec = make_rsrc(S.rec[tgt], sizeof(double) * (size_t)kRec * P.V);
    // camera constants: read through the constant address space (uniform address, written by an earlier
    // kernel) -> scalar loads straight into SGPR operands, no v_readlane pair per use
    const cptr4 ccs = (cptr4)(S.cconst[tgt] + kCStride * cam);
    auto CC = [&](int k) { return ccs[k]; };                                          // camera constants (see k_view_prep)
    const RecLane rl = rec_lane(lane, (unsigned)P.V);
    const __amdgpu_buffer_rsrc_t r_vc = make_rsrc(S.vconst, sizeof(double) * (size_t)kVStride * P.V);
    const __amdgpu_buffer_rsrc_t r_u = make_rsrc(P.obs_u, sizeof(double) * (size_t)P.N), r_v = make_rsrc(P.obs_v, sizeof(double) * (size_t)P.N);
    // Per-view metadata (corner count, record slot) of a block of <= 64 views sits in lane registers and is
    // read with v_readlane; the observations of a camera's views are contiguous, so the offset is a running
    // sum.  No dependent global load -- and therefore no in-order vmcnt wait behind the previous view's
    // record stores -- is left inside the view loop.
    int off_next = vb < ve ? P.view_obs[vb] : 0;
#if TSCM_PRIO
    // Wave priority by progress.  The four waves of a SIMD share the fp64 pipe, and the arbiter serves the OLDEST wave
    // first: left alone they finish one after the other (46 / 57 / 68 / 79 us into the launch at config 4), and the last
    // ten microseconds of the kernel run on one wave per SIMD, whose dependent chains cannot fill the pipe.  A wave that
    // is further along in its chunk than its neighbours gives way: the priority falls from 3 to 0 over each half of
    // the chunk (eighths of its views, mod 4), so whoever is behind by an eighth outranks whoever is ahead, and the
    // four finish within 6 us of each other (59 / 61 / 63 / 65 us after the epilogue rewrite).  The assignment of views
    // to waves is untouched: the results are the same bits.  (TSCM_PRIO=0: off; 4, 5: the other rules of the A/B runs in
    // profiles/r03_eval_gram_ab.txt -- by phase: geometry high, MFMA low; by quarters of the chunk.)
    auto prio = [&](int ph, int view) {
        if (TSCM_PRIO == 4) set_prio(ph == 0 ? 3 : ph == 3 ? 1 : 0);
        else if (TSCM_PRIO == 5) { if (ph == 0) set_prio(3 - min(3, 4 * (view - vb) / max(1, ve - vb))); }
        else { if (ph == 0) set_prio(3 - min(3, 8 * (view - vb) / max(1, ve - vb) % 4)); }
    };
#define PRIO(ph, view) prio(ph, view)
#else
#define PRIO(ph, view)
#endif
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
        PRIO(0, view);
        const cptr4 vcs = (cptr4)(S.vconst + (size_t)kVStride * view);      // this view's 27 constants: scalar loads
        auto VC = [&](int k) { return vcs[k]; };
        d4 accU = { 0.0, 0.0, 0.0, 0.0 }, accV = { 0.0, 0.0, 0.0, 0.0 };
        for (int c0 = 0; c0 < cnt; c0 += 64) {
            const int j = c0 + lane;
            const bool valid = j < cnt;
            double *fu = Fl + lane;
            double fv[kTcols];                 // v-rows wait in registers until the u-rows have been consumed
            if (valid && !(ablate & 4)) {
                const double x = bxy[2 * j], y = bxy[2 * j + 1];
                // (RPC > 0: boards of <= 56 corners, always a single pass -- the load path below would otherwise put a
                // vmcnt(0) in front of the residual, i.e. a wait for the prefetch issued a few hundred cycles earlier)
                const bool later_pass = RPC == 0 && c0 != 0;
                const double ou = later_pass ? buf_load_f64(r_u, 8u * j, 8u * (unsigned)off) : pf_u, ov = later_pass ? buf_load_f64(r_v, 8u * j, 8u * (unsigned)off) : pf_v;
                // semantic column -> tile column of this kernel
                constexpr int tcol[15] = { kTcWb, kTcWb + 1, kTcWb + 2, tc_tc(0), tc_tc(1), tc_tc(2), kTcWc, kTcWc + 1, kTcWc + 2,
                                           kTcF, kTcOne, kTcXi, kTcLam, kTcAl, kTcR };
                corner_geometry(x, y, ou, ov, VC, CC, [&](int gc, double u, double v) { fu[tcol[gc] * RP] = u; fv[tcol[gc]] = v; });
            } else if (lane < prev_nv) {
#pragma unroll
                for (int c = 0; c < kTcols; ++c) fu[c * RP] = 0.0;
            }
            if (c0 == 0) {
                // Prefetch of the next view, issued once the current view's observations have been consumed: the
                // loads reuse the same registers (no copy that would have to wait for them), and everything
                // between here and their use at the top of the next view is four unconditional stores.
                // Always issued (the block's last view re-reads itself; lanes past the corner count read past
                // the end of the buffer, i.e. zero): unconditional loads keep the vmcnt bookkeeping exact.
                const int vn = min(view + 1, vend - 1);
                const int cn = view + 1 < vend ? __builtin_amdgcn_readlane(m_cnt, vn - vbase) : 0;
                // pull the next view's 256-byte constant record into the L2 with one tracked vector load (lanes
                // 0..3, one dword per 64-byte line): the scalar loads at the top of the next view then hit the L2
                warm = __builtin_amdgcn_raw_buffer_load_b32(r_vc, lane < 4 ? 64 * lane : (int)0xffffe000u, (int)(8u * (unsigned)kVStride * (unsigned)vn), 0);
                pf_u = buf_load_f64(r_u, lane < cn ? 8u * lane : 0xffffe000u, 8u * (unsigned)off_next);
                pf_v = buf_load_f64(r_v, lane < cn ? 8u * lane : 0xffffe000u, 8u * (unsigned)off_next);
            }
            wave_lds_fence();
            PRIO(1, view);
            const int nv = min(64, cnt - c0);
            prev_nv = nv;
            const int ksteps = (nv + 3) >> 2;
            // rows past the last corner are zero, so the loops run in
            // pairs of k-steps; operands of the next pair are fetched while the current MFMAs issue
            const double *fp = Fl + col * RP + kq;      // lane (col, kq) feeds tile column col, row 4t + kq
            const int tmax = (HV >> 2) - 2;
            constexpr int KSF = RPC > 0 ? (RPC - 2) / 4 : 1;      // k-steps of a full tile
            const bool full_tile = (TSCM_EXP & 1) && RPC > 0 && ksteps == KSF;
            if (full_tile && !(ablate & 1)) gram_full<KSF, (TSCM_EXP & 2) != 0>(fp, accU);
            else {
                double a0 = fp[0], a1 = fp[4];
                for (int t = 0; t < ksteps && !(ablate & 1); t += 2) {
                    const int tn = min(t + 2, tmax);
                    const double n0 = fp[4 * tn], n1 = fp[4 * tn + 4];
                    accU = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a0, accU, 0, 0, 0);
                    accU = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, a1, accU, 0, 0, 0);
                    a0 = n0; a1 = n1;
                }
            }
            wave_lds_fence();
            if (valid) {
#pragma unroll
                for (int c = 0; c < kTcols; ++c) fu[c * RP] = fv[c];
            }
            wave_lds_fence();
            PRIO(2, view);
            if (full_tile && !(ablate & 1)) gram_full<KSF, (TSCM_EXP & 2) != 0>(fp, accV);
            else {
                double a0 = fp[0], a1 = fp[4];
                for (int t = 0; t < ksteps && !(ablate & 1); t += 2) {
                    const int tn = min(t + 2, tmax);
                    const double n0 = fp[4 * tn], n1 = fp[4 * tn + 4];
                    accV = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a0, accV, 0, 0, 0);
                    accV = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, a1, accV, 0, 0, 0);
                    a0 = n0; a1 = n1;
                }
            }
            wave_lds_fence();
        }
        // The next view's constants are waited for HERE, a full MFMA phase after their load was issued and before
        // this view's record stores: on gfx9 loads and stores share vmcnt and may complete out of order, so any
        // wait for a load with stores in flight is a vmcnt(0) -- a wait placed right after the stores (the top
        // of the next view) would expose the whole store latency.  The geometry is done with cst by now.
        asm volatile("" :: "v"(warm));       // the warming load retires here, before this view's record stores
        PRIO(3, view);
        camU += accU; camV += accV;
        if (ablate & 2) continue;
        store_view_record(r_rec, lane, accU, accV, ccs, (unsigned)__builtin_amdgcn_readlane(m_slot, view - vbase), rl);
    }
    }   // block of <= 64 views
    TL_ONLY(
    if (lane == 0 && tl_iter == 5 && cand && chunk < kTimelineWaves) {
        g_timeline[4 * chunk] = (long long)__builtin_amdgcn_s_getreg(63492);     // HW_ID
        g_timeline[4 * chunk + 1] = (long long)__builtin_amdgcn_s_getreg(6164);  // XCC_ID
        g_timeline[4 * chunk + 2] = tl_t0;
        g_timeline[4 * chunk + 3] = wall_clock64();
    }
    )
    // the four waves of the workgroup (same camera) sum their tiles through LDS in a fixed order
    wave_lds_fence();
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) { lds[(kq + 4 * rg) * 16 + col] = camU[rg]; lds[256 + (kq + 4 * rg) * 16 + col] = camV[rg]; }
    __syncthreads();
    {
        const int t = threadIdx.x;
        const size_t st = P.lds_wave;
        double *part = S.campart + (size_t)512 * blockIdx.x;
        part[t] = (lds_all[t] + lds_all[st + t]) + (lds_all[2 * st + t] + lds_all[3 * st + t]);
        part[256 + t] = (lds_all[256 + t] + lds_all[st + 256 + t]) + (lds_all[2 * st + 256 + t] + lds_all[3 * st + 256 + t]);
    }
}

// Pass plan of the Gram kernels k_eval_gram4 / k_eval_gram_f32 (round 6: every board size): a pass holds 4 KS <= 56 rows
constexpr int kG4MaxKS = 16;                    // k-steps of a pass: at most 64 rows (one per lane)
// pass plan of a board of n corners: `passes` passes of `per` corners each (a multiple of four; the last pass takes what is left).
// A pass may hold 64 rows (16 k-steps) where the kernel's LDS -- tile + board points, four waves -- still admits four workgroups per
// CU, i.e. boards of up to 64 corners in ONE pass (8 x 8, 9 x 7: rounds 3-6a ran them as two half-empty passes of 32) and e.g. 117
// corners in two; otherwise 56 rows (14 k-steps: the tile of the 9 x 6 kernel).
struct G4Plan { int passes, per, ks; };
__host__ __device__ inline G4Plan g4_plan(int n_points)
{
    G4Plan g;
    for (int cap = 16; cap >= 14; cap -= 2) {
        g.passes = (n_points + 4 * cap - 1) / (4 * cap);
        if (g.passes < 1) g.passes = 1;
        g.ks = ((n_points + g.passes - 1) / g.passes + 3) / 4;
        if (g.ks < 1) g.ks = 1;
        g.per = 4 * g.ks;
        const long tile = g.ks * 68 > 512 ? g.ks * 68 : 512;           // (g4_tile_doubles: tscm_eval_gram4.h)
        if (g.ks <= 14 || 4 * 8 * (tile + 2L * n_points) <= 40 * 1024) break;      // four workgroups of four waves in 160 KB
    }
    return g;
}

#include "tscm_eval_f32.h"

// per-camera raw tile (GU | GV) reduction: one block per (camera, slice of 32 of the 512 raw entries).  Eight threads per
// entry take every eighth workgroup tile -- up to 32 loads per thread requested at once -- and are combined through LDS
// in a fixed order: campart2[cam][512] holds the finished sums (round 3; before: 16 groups of tiles per camera here and
// a second level in k_finalize_eval, 16 more dependent loads per thread in a kernel that is nothing but a latency chain)
__device__ void cam_reduce_block(const DevProblem &P, const DevState &S, int blk, double *sm /* 256 doubles */)
{
    const int cam = blk / kCamSl, sl = blk % kCamSl;
    int cb, ce;
    if (P.C <= kMaxCamLds) {
        cb = P.cam_wg[0]; ce = P.cam_wg[1];
#pragma unroll
        for (int q = 1; q < kMaxCamLds; ++q) { cb = cam >= q ? P.cam_wg[q] : cb; ce = cam >= q ? P.cam_wg[q + 1] : ce; }
    } else {
        cb = P.cam_chunk_ptr[cam]; ce = P.cam_chunk_ptr[cam + 1];
    }
    const int t = threadIdx.x, o = t & 31, ph = t >> 5;
    const double *src = S.campart + 32 * sl + o;
    double acc = 0.0;
    for (int base = cb + ph; base < ce; base += 256) {
        double v[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) { const int c = base + 8 * u; v[u] = src[(size_t)512 * min(c, ce - 1)]; v[u] = c < ce ? v[u] : 0.0; }
#pragma unroll
        for (int w = 16; w >= 1; w >>= 1)
#pragma unroll
            for (int u = 0; u < w; ++u) v[u] += v[u + w];
        acc += v[0];
    }
    sm[t] = acc;
    __syncthreads();
    if (t < 32)
        handoff_store(&S.campart2[(size_t)512 * cam + 32 * sl + t], ((sm[t] + sm[32 + t]) + (sm[64 + t] + sm[96 + t])) + ((sm[128 + t] + sm[160 + t]) + (sm[192 + t] + sm[224 + t])));
    __syncthreads();
}

// All-reduce over the 16 lanes of a DPP row without the LDS crossbar: a butterfly of quad_perm [1,0,3,2], quad_perm
// [2,3,0,1], row_half_mirror and row_mirror (after the first two steps every lane of a quad holds the quad's value, so
// the mirrored partner is as good as the xor partner).  A VALU move per 32-bit half and step, a few clocks of latency
// each, against ~100 ns per ds_bpermute round trip of __shfl_xor.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row16_allsum(double v)
{
    v += dpp_f64<0xB1>(v); v += dpp_f64<0x4E>(v); v += dpp_f64<0x141>(v); v += dpp_f64<0x140>(v);
    return v;
}
__device__ __forceinline__ double row16_allmax(double v)
{
    v = fmax(v, dpp_f64<0xB1>(v)); v = fmax(v, dpp_f64<0x4E>(v)); v = fmax(v, dpp_f64<0x141>(v)); v = fmax(v, dpp_f64<0x140>(v));
    return v;
}

#include "tscm_eval_gram4.h"

// deterministic block reductions (256 threads)
// Block reductions (256 threads; any multiple of 64 works): DPP butterfly inside each 16-lane row, two xor shuffles across the
// rows of a wave, one LDS exchange between the
// waves -- two barriers per call, several quantities at once (the previous LDS tree cost ten barriers per
// quantity: 1.5 us each on the single-block control paths).  Fixed order: bit-reproducible.
template <int NS>
__device__ __forceinline__ void block_reduce256(double (&sum)[NS], double &mx, double *sm)   // sm: >= 4 * (NS + 1) doubles
{
#pragma unroll
    for (int i = 0; i < NS; ++i) sum[i] = row16_allsum(sum[i]);
    mx = row16_allmax(mx);
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
#pragma unroll
        for (int i = 0; i < NS; ++i) sum[i] += __shfl_xor(sum[i], off);
        mx = fmax(mx, __shfl_xor(mx, off));
    }
    const int wave = threadIdx.x >> 6, nw = (int)blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int i = 0; i < NS; ++i) sm[wave * (NS + 1) + i] = sum[i];
        sm[wave * (NS + 1) + NS] = mx;
    }
    __syncthreads();
    if (nw == 4) {
#pragma unroll
        for (int i = 0; i < NS; ++i) sum[i] = (sm[i] + sm[(NS + 1) + i]) + (sm[2 * (NS + 1) + i] + sm[3 * (NS + 1) + i]);
        mx = fmax(fmax(sm[NS], sm[(NS + 1) + NS]), fmax(sm[2 * (NS + 1) + NS], sm[3 * (NS + 1) + NS]));
    } else {                                    // other workgroup sizes (k_solve_reduced<4, 32>): waves in order
#pragma unroll
        for (int i = 0; i < NS; ++i) { double t = 0.0; for (int w = 0; w < nw; ++w) t += sm[w * (NS + 1) + i]; sum[i] = t; }
        double t = sm[NS];
        for (int w = 1; w < nw; ++w) t = fmax(t, sm[w * (NS + 1) + NS]);
        mx = t;
    }
    __syncthreads();
}
__device__ __forceinline__ double block_sum256(double v, double *sm)
{
    double s[1] = { v }, m = 0.0;
    block_reduce256<1>(s, m, sm);
    return s[0];
}
__device__ __forceinline__ double block_max256(double v, double *sm)
{
    double s[1] = { 0.0 }, m = v;
    block_reduce256<1>(s, m, sm);
    return m;
}

// per-board gradient / norm statistics of the evaluation target (and, at iteration 0, the
// Jacobi scaling of the board columns: s = 1/(1 + ||J_col||)).  grid ceil(B/256) x 256
__device__ void board_stats_block(const DevProblem &P, const DevState &S, int cand, int init, int blk, double *sm)
{
    const int tgt = cand ? (S.ctrl->cur ^ 1) : S.ctrl->cur;
    const int b = blk * 256 + threadIdx.x;
    double gmax = 0.0, gsq = 0.0, xsq = 0.0;
    if (b < P.B) {
        const int q0 = P.bv_ptr[b], q1 = P.bv_ptr[b + 1];
        if (q1 > q0 && !P.board_const[b]) {         // constant pose blocks are not part of the reduced program
            double g[6] = { 0, 0, 0, 0, 0, 0 }, dg[6] = { 0, 0, 0, 0, 0, 0 };
            // the gradient columns of up to four views per trip, requested together (a load inside a loop of unknown
            // length is one memory round trip per view); same order of additions
            for (int qb = q0; qb < q1; qb += 4) {
                double w[4][6];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const double *G = rec_g(S.rec[tgt], P.V, min(qb + u, q1 - 1));
#pragma unroll
                    for (int i = 0; i < 6; ++i) w[u][i] = G[i];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int i = 0; i < 6; ++i) g[i] += qb + u < q1 ? w[u][i] : 0.0;
            }
            if (init) {                         // diag(E^T E) is only needed for the Jacobi scaling
                for (int q = q0; q < q1; ++q) {
                    const double *W = rec_w(S.rec[tgt], q);
                    const double *E = rec_e(S.rec[tgt], P.V, q), *Rc = S.cconst[tgt] + kCStride * P.slot_cam[q];
                    for (int i = 0; i < 3; ++i) { dg[i] += E[6 * i + i]; dg[3 + i] += tb_tb(W, Rc, i, i); }
                }
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
    double red[2] = { gsq, xsq }, m = gmax;
    block_reduce256<2>(red, m, sm);
    const double s1 = red[0], s2 = red[1];
    if (threadIdx.x == 0) { handoff_store(&S.st_part[kStStride * blk], m); handoff_store(&S.st_part[kStStride * blk + 1], s1); handoff_store(&S.st_part[kStStride * blk + 2], s2); }
}

// one launch for the two independent post-evaluation reductions:
//   blocks [0, C*kCamSl)            sums of the per-workgroup camera tiles
//   blocks [C*kCamSl, +ceil(B/256)) per-board gradient / norm statistics (+ Jacobi scaling at iteration 0)
__global__ __launch_bounds__(256) void k_reduce_stats(DevProblem P, DevState S, int cand, int init)
{
    KTL(1);
    // snapshot of the LM state for the control step in the head of the next launch (k_schur_gram, DevState::ctrl_snap):
    // taken BEFORE the early exit, so that a finished -- or faulted -- solve is seen there as well
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x < sizeof(CtrlHead) / 8)
        reinterpret_cast<unsigned long long *>(S.ctrl_snap)[threadIdx.x] = reinterpret_cast<const unsigned long long *>(S.ctrl)[threadIdx.x];
    if (S.ctrl->done) return;
    __shared__ double sm[256];
    const int nc = P.C * kCamSl;
    if ((int)blockIdx.x < nc) cam_reduce_block(P, S, blockIdx.x, sm);
    else board_stats_block(P, S, cand, init, blockIdx.x - nc, sm);
}

// H_stage = C camera tiles ([F|r]^T[F|r] from the raw u/v sums), then kScal scalars: [0] model_b [1] stepsq_b [2] xsq_b [3] gsq_b
// [4] e-block factorisation failures on this rank, then one slot per rank with that rank's board gradient max-norm
// (zero in the other ranks' slots): ONE sum all-reduce carries sums, the failure flag and the maximum.
// What the control step reads from memory that does NOT depend on the evaluation being finalised: the LM state and the
// target point's camera-side parameters, requested together with the first loads of the workgroup that runs the step.
struct ControlPre { CtrlHead c; double x[2]; bool free_param[2]; };
// what the kernel that runs the control step in its head goes on with (LDS, written by thread 0)
struct CtlOut { int cur, done; double radius, dmin, dmax; };
// `head`: where the LM state is read from -- S.ctrl where the calling workgroup is the only one that takes the step
// (k_reduce_control's last workgroup, k_control_tail, k_control), S.ctrl_snap where every workgroup of a launch takes it
// while one of them writes S.ctrl (k_schur_gram)
// ... split in two for a workgroup that has to WAIT for the evaluation's reductions first (k_schur_gram<NV, true>): what does not
// depend on them -- the parameters of both buffers, the camera flags, the back-substitution's partials (summed per thread) -- is
// requested in front of the wait, the LM state behind it
struct ControlEarly { double x0[2], x1[2]; int act[2], cst[2]; double mb, ss; };
__device__ __forceinline__ void control_early_params(const DevProblem &P, const DevState &S, ControlEarly &e)
{
    // (both parameter buffers and the camera flags are requested without waiting for `cur`: one round trip, not two)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int p = threadIdx.x + 256 * j;
        const int m = min(p >> 4, P.C - 1), a = p & 15;
        const int ia = a < 6 ? 6 * m + a : 9 * m + min(a - 6, 8);
        e.x0[j] = a < 6 ? S.cam_rt[0][ia] : S.intr[0][ia];
        e.x1[j] = a < 6 ? S.cam_rt[1][ia] : S.intr[1][ia];
        e.act[j] = P.cam_active[m]; e.cst[j] = P.cam_const[m];
    }
}
__device__ __forceinline__ void control_state(const DevProblem &P, int init, ControlPre &pre, const CtrlHead *head, const ControlEarly &e)
{
    // (the LM state through the scalar cache: wave-uniform, and when 500 workgroups take the step at once -- k_schur_gram's
    // head -- 2,000 waves x 22 vector loads of the same six cache lines queue up at one L2 channel)
    {
        static_assert(sizeof(CtrlHead) % 8 == 0, "copied in 8-byte words");
        typedef const unsigned long long __attribute__((address_space(4))) *cq4;
        const cq4 src = (cq4)(const void *)head;
        unsigned long long w[sizeof(CtrlHead) / 8];
#pragma unroll
        for (unsigned q = 0; q < sizeof(CtrlHead) / 8; ++q) w[q] = src[q];
        __builtin_memcpy(&pre.c, w, sizeof(CtrlHead));
    }
    const int tgt = init ? pre.c.cur : (pre.c.cur ^ 1);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int p = threadIdx.x + 256 * j;
        const int a = p & 15;
        const bool in = p < 16 * P.C && a < 15;
        pre.x[j] = in ? (tgt ? e.x1[j] : e.x0[j]) : 0.0;
        pre.free_param[j] = in & (e.act[j] != 0) & !((a < 6) & (e.cst[j] != 0));
    }
}
__device__ __forceinline__ void control_prefetch(const DevProblem &P, const DevState &S, int init, ControlPre &pre, const CtrlHead *head)
{
    ControlEarly e;
    control_early_params(P, S, e);
    control_state(P, init, pre, head, e);
}
// GUARD: the step behind an all-reduce of H_stage (k_control, k_schur_gram's ctl = 2) checks the ranks' decision words first
template <bool GUARD = false>
__device__ void control_step(const DevProblem &P, const DevState &S, int init, const ControlPre &pre, double *sm, const double *H, const double *sc, double *stage_copy,
                             bool writer = true, CtlOut *out = nullptr);

// raw (GU | GV) tile of one camera (G: 512 doubles in LDS) -> H layout: 14x14 [F | r]^T [F | r] in a 16x16 slot
__device__ __forceinline__ double camera_tile_entry(const double *G, int t)
{
    const int a = t >> 4, b = t & 15;
    double v = 0.0;
    if (a < 14 && b < 14) {
        const int ta = f_tile(a), tb = f_tile(b), m = f_mask(a) & f_mask(b);
        if (m & 1) v += G[ta * 16 + tb];
        if (m & 2) v += G[256 + ta * 16 + tb];
    }
    return v;
}

// Rank-divergence guard (round 6).  The sharded solver has every rank run the reduced solve and the control step REDUNDANTLY on
// the all-reduced T and H_stage; that is only right while every rank receives the same bits from every all-reduce (ring and tree
// sums do; a protocol that sums in a rank-relative order need not) and computes the same bits from them.  Nothing used to
// notice if that ever failed -- two ranks one ulp apart may take different accept / reject or termination decisions: silently
// different states behind equal numbers of collectives, or a 60 s watchdog abort.  Now every rank folds what the ranks must agree
// on -- the LM state as its last control step left it AND the replicated results of its reduced solve (model cost change and step
// norm of the camera step, the linear-solver failure flag) -- into an integer below 2^52 (exact as a double, exact through a sum
// whose other terms are zero, whatever the order) and puts it into its own slot behind the gradient-max slots of H_stage
// (k_finalize_eval); the all-reduce that carries the evaluation carries the words, and the control step behind it compares
// every slot with the word of its own state: a mismatch raises ctrl->fault = kFaultRanksDisagree on every rank in the SAME step
// (TSCM_E_PEER "ranks disagree at iteration k", communicator unusable).  No extra collective, no extra launch.
__device__ __forceinline__ double decision_word(const CtrlHead &c)
{
    unsigned long long h = 0x9E3779B97F4A7C15ull;
    auto mix = [&](unsigned long long v) { h ^= v; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 29; };
    mix(((unsigned long long)(unsigned)c.iteration << 32) | (unsigned)c.lm_iterations);
    mix(((unsigned long long)(unsigned)c.cur << 40) | ((unsigned long long)(unsigned)c.done << 32) | ((unsigned)c.term_reason << 16) | ((unsigned)c.lin_fail << 8) | (unsigned)c.num_invalid);
    mix(((unsigned long long)(unsigned)c.num_successful << 32) | (unsigned)c.num_unsuccessful);
    mix((unsigned long long)__double_as_longlong(c.radius));
    mix((unsigned long long)__double_as_longlong(c.decrease_factor));
    mix((unsigned long long)__double_as_longlong(c.x_cost));
    mix((unsigned long long)__double_as_longlong(c.model_cam));
    mix((unsigned long long)__double_as_longlong(c.stepsq_cam));
    return (double)((h >> 12) | 1ull);           // 52 bits, never zero (zero = a slot nobody wrote)
}

// the per-workgroup scalar partials of the back-substitution and of the board statistics -> the kScal + world scalars
// that follow the camera tiles in H_stage, written to `sc` (global or LDS; 256 threads; sm: block_reduce256 scratch).
// Every load is unconditional (clamped index, value masked): a load under `if (i < n)` is a branch with its own wait,
// and the eight + four of them in the ragged ends were twelve memory round trips in a row (5 us of the control
// workgroup's 10, tools/kernel_timeline.py).
// THROUGH: the board statistics were handed over inside this launch (handoff_store): read them the same way
// (the back-substitution's partials, summed per thread: written by the launch before -- no hand-off)
__device__ __forceinline__ void backsub_partials(const DevState &S, int have_backsub, double &mb, double &ss)
{
    const int t = threadIdx.x;
    mb = 0.0; ss = 0.0;
    if (have_backsub) {
        const d2 *bp = reinterpret_cast<const d2 *>(S.bs_part);
        const int n = S.n_bs_blocks;
        d2 a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = d2{ 0.0, 0.0 };
        for (int i = t; i < n; i += 8 * 256) {
            d2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = bp[min(i + 256 * u, n - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] += i + 256 * u < n ? v[u] : d2{ 0.0, 0.0 };
        }
        const d2 r = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
        mb = r[0]; ss = r[1];
    }
}
template <bool THROUGH>
__device__ __forceinline__ void reduce_scalar_partials_from(const DevProblem &P, const DevState &S, double mb, double ss, int lin_fail, double *sc, double *sm)
{
    const int t = threadIdx.x;
    double gm = 0.0, gs = 0.0, xs = 0.0;
    {
        const int n = S.n_st_blocks;
        double g4[4] = { 0, 0, 0, 0 }, s4[4] = { 0, 0, 0, 0 }, x4[4] = { 0, 0, 0, 0 };
        // THROUGH: sc1 buffer loads (what handoff_load compiles to), lanes past the end parked beyond the buffer -- read through, a
        // clamped index makes every one of a launch's 2,000 waves fetch the last line from the L2 twelve times (round 6: +0.45 us
        // on k_schur_gram<2, true>); a parked lane costs no memory request and returns zero
        const __amdgpu_buffer_rsrc_t r_st = make_rsrc(S.st_part, sizeof(double) * (size_t)kStStride * (size_t)n);
        for (int i = t; i < n; i += 4 * 256) {
            double q[4][3];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if constexpr (THROUGH) {
                    const unsigned off = i + 256 * u < n ? 8u * (unsigned)kStStride * (unsigned)(i + 256 * u) : 0xffffe000u;
                    const d2 a = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(r_st, (int)off, 0, 16));
                    q[u][0] = a[0]; q[u][1] = a[1];
                    q[u][2] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r_st, (int)off, 16, 16));
                } else {
                    const double *src = S.st_part + kStStride * (size_t)min(i + 256 * u, n - 1);
                    q[u][0] = src[0]; q[u][1] = src[1]; q[u][2] = src[2];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { const bool in = i + 256 * u < n; g4[u] = fmax(g4[u], in ? q[u][0] : 0.0); s4[u] += in ? q[u][1] : 0.0; x4[u] += in ? q[u][2] : 0.0; }
        }
        gm = fmax(fmax(g4[0], g4[1]), fmax(g4[2], g4[3])); gs = (s4[0] + s4[1]) + (s4[2] + s4[3]); xs = (x4[0] + x4[1]) + (x4[2] + x4[3]);
    }
    double red[4] = { mb, ss, gs, xs };
    block_reduce256<4>(red, gm, sm);
    mb = red[0]; ss = red[1]; gs = red[2]; xs = red[3];
    if (t == 0) {
        sc[0] = mb; sc[1] = ss; sc[2] = xs; sc[3] = gs; sc[4] = lin_fail ? 1.0 : 0.0; sc[5] = 0.0; sc[6] = 0.0; sc[7] = 0.0;
        for (int r = 0; r < P.world; ++r) sc[kScal + r] = r == P.rank ? gm : 0.0;
    }
}
template <bool THROUGH>
__device__ __forceinline__ void reduce_scalar_partials(const DevProblem &P, const DevState &S, int have_backsub, int lin_fail, double *sc, double *sm)
{
    double mb, ss;
    backsub_partials(S, have_backsub, mb, ss);
    reduce_scalar_partials_from<THROUGH>(P, S, mb, ss, lin_fail, sc, sm);
}

// camera tiles (raw u/v sums -> [F|r]^T[F|r]) into H_stage + reduction of the per-block scalar partials, for the paths
// with something between the evaluation and the control step (all-reduce: k_control follows) or without a control step
// (tscm_eval_normal_equations).  grid (C + 1) x 256, or C x 256 for the camera tiles alone.
__global__ __launch_bounds__(256) void k_finalize_eval(DevProblem P, DevState S, int have_backsub)
{
    KTL(2);
    const int done = S.ctrl->done, lin_fail = S.ctrl->lin_fail;
    // this rank's decision word (see decision_word): the state every rank must share when the control step behind the all-reduce
    // of H_stage is taken (requested with the two fields above: one round trip)
    double word = 0.0;
    if ((int)blockIdx.x == P.C && threadIdx.x == 0 && P.world > 1) word = decision_word(*S.ctrl);
    if (done) return;
    __shared__ double sm[256];
    __shared__ double G[512];
    const int t = threadIdx.x;
    if ((int)blockIdx.x < P.C) {
        const int cam = blockIdx.x;
        G[t] = S.campart2[(size_t)512 * cam + t];
        G[256 + t] = S.campart2[(size_t)512 * cam + 256 + t];
        __syncthreads();
        S.H_stage[256 * cam + t] = camera_tile_entry(G, t);
    } else {
        reduce_scalar_partials<false>(P, S, have_backsub, lin_fail, S.H_stage + 256 * P.C, sm);
        if (t == 0 && P.world > 1) for (int r = 0; r < P.world; ++r) S.H_stage[256 * P.C + kScal + P.world + r] = r == P.rank ? word : 0.0;
    }
}

// What is left of an evaluation once the camera-tile sums (campart2) and the scalar partials are complete: one batch of
// loads -- the 512 finished sums per camera, the partials, the LM state, the target point's camera parameters -- H in
// LDS, the control step on that copy.  Called by the last workgroup of k_reduce_control (writer), by k_control_tail,
// and by EVERY workgroup of k_schur_gram in its head (one of them the writer): the step is cheap, deterministic and
// needs no hand-off when everybody takes it.  Hl: 256 C + kScal + 8 doubles, Gall: 512 C, sm: 256 (LDS; C <= 8).
// THROUGH: the sums were handed over inside this launch (k_reduce_control); otherwise they come through a kernel boundary
// and plain loads let the L2s serve the 500 workgroups of k_schur_gram that all read the same 36 KB
template <bool THROUGH>
__device__ __forceinline__ void finish_evaluation(const DevProblem &P, const DevState &S, int init, int have_backsub, bool writer,
                                                  double *Hl, double *Gall, double *sm, CtlOut *out, const CtrlHead *head)
{
    const int t = threadIdx.x;
    ControlPre pre;
    control_prefetch(P, S, init, pre, head);
    double gu[kMaxCamLds], gv[kMaxCamLds];
#pragma unroll
    for (int m = 0; m < kMaxCamLds; ++m) {
        const int cam = min(m, P.C - 1);
        gu[m] = THROUGH ? handoff_load(&S.campart2[(size_t)512 * cam + t]) : S.campart2[(size_t)512 * cam + t];
        gv[m] = THROUGH ? handoff_load(&S.campart2[(size_t)512 * cam + 256 + t]) : S.campart2[(size_t)512 * cam + 256 + t];
    }
    double *scl = Hl + 256 * P.C;
    reduce_scalar_partials<THROUGH>(P, S, have_backsub, pre.c.lin_fail, scl, sm);
    KTLX(2, true);
    // (all cameras' raw tiles in LDS at once: one barrier, not two per camera)
#pragma unroll
    for (int m = 0; m < kMaxCamLds; ++m) if (m < P.C) { Gall[512 * m + t] = gu[m]; Gall[512 * m + 256 + t] = gv[m]; }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < kMaxCamLds; ++m) if (m < P.C) Hl[256 * m + t] = camera_tile_entry(Gall + 512 * m, t);
    __syncthreads();
    KTLX(3, true);
    // (no global store up to here: a barrier behind one waits for its acknowledgement, a microsecond.  The control
    // step writes H -- and H_stage, for whoever reads the staged copy -- behind its own last barrier.)
    control_step(P, S, init, pre, sm, Hl, scl, S.H_stage, writer, out);
}

// The same step for a workgroup that only needs its OUTCOME (every workgroup of k_schur_gram but the extra one that
// writes): of H only the gradient column and the cost entry of each camera enter the step -- 15 entries per camera, two
// loads per thread straight from the finished sums instead of 16 KB through LDS and two barriers.  Hl: 256 C + kScal + 8.
// ... in two parts for k_schur_gram<NV, true>: control_early in front of the wait for the riding reductions, this behind it (the same
// loads, the same arithmetic in the same order: same bits)
__device__ __forceinline__ void control_early(const DevProblem &P, const DevState &S, int have_backsub, ControlEarly &e)
{
    control_early_params(P, S, e);
    backsub_partials(S, have_backsub, e.mb, e.ss);
}
template <bool THROUGH>
__device__ __forceinline__ void control_outcome_late(const DevProblem &P, const DevState &S, int init, const ControlEarly &e, double *Hl, double *sm, CtlOut *out, const CtrlHead *head)
{
    const int t = threadIdx.x;
    ControlPre pre;
    control_state(P, init, pre, head, e);
    const int m = min(t >> 4, P.C - 1), a = t & 15;
    const int fa = min(a, 13), ta = f_tile(fa), tb = f_tile(kFR), mk = f_mask(fa) & f_mask(kFR);
    const double *pu = &S.campart2[(size_t)512 * m + ta * 16 + tb], *pv = &S.campart2[(size_t)512 * m + 256 + ta * 16 + tb];
    const double gu = THROUGH ? handoff_load(pu) : *pu, gv = THROUGH ? handoff_load(pv) : *pv;
    double *scl = Hl + 256 * P.C;
    reduce_scalar_partials_from<THROUGH>(P, S, e.mb, e.ss, pre.c.lin_fail, scl, sm);
    if (t < 16 * P.C && a < 14) Hl[256 * m + a * 16 + kFR] = ((mk & 1) ? gu : 0.0) + ((mk & 2) ? gv : 0.0);
    __syncthreads();
    control_step(P, S, init, pre, sm, Hl, scl, nullptr, /*writer=*/false, out);
}
// THROUGH: the finished sums and the board statistics were handed over inside this launch
template <bool THROUGH = false>
__device__ __forceinline__ void control_outcome(const DevProblem &P, const DevState &S, int init, int have_backsub, double *Hl, double *sm, CtlOut *out, const CtrlHead *head)
{
    const int t = threadIdx.x;
    ControlPre pre;
    control_prefetch(P, S, init, pre, head);
    // thread (camera m, a): H[m][a][kFR] for a < 14 (a = kFR = 13: the cost entry)
    const int m = min(t >> 4, P.C - 1), a = t & 15;
    const int fa = min(a, 13), ta = f_tile(fa), tb = f_tile(kFR), mk = f_mask(fa) & f_mask(kFR);
    const double *pu = &S.campart2[(size_t)512 * m + ta * 16 + tb], *pv = &S.campart2[(size_t)512 * m + 256 + ta * 16 + tb];
    const double gu = THROUGH ? handoff_load(pu) : *pu, gv = THROUGH ? handoff_load(pv) : *pv;
    double *scl = Hl + 256 * P.C;
    reduce_scalar_partials<THROUGH>(P, S, have_backsub, pre.c.lin_fail, scl, sm);
    if (t < 16 * P.C && a < 14) Hl[256 * m + a * 16 + kFR] = ((mk & 1) ? gu : 0.0) + ((mk & 2) ? gv : 0.0);
    __syncthreads();
    control_step(P, S, init, pre, sm, Hl, scl, nullptr, /*writer=*/false, out);
}

// One GPU: everything between the evaluation and the next Schur complement in ONE launch (round 3; before:
// k_reduce_stats, k_finalize_eval with a second reduction level, the control step in its last workgroup -- 6.6 + 16.4 us
// of an iteration of 131, every dependent load of these small kernels a cold round trip of 1-2 us).  The workgroups
// are k_reduce_stats' (camera-tile slices, board statistics); whichever arrives last (release -> counter -> acquire at
// agent scope, cdna guide G16) requests in ONE batch what is left -- the 512 finished sums per camera, the scalar
// partials, the LM state, the target point's camera parameters -- forms H_stage and runs the control step.
__global__ __launch_bounds__(256) void k_reduce_control(DevProblem P, DevState S, int cand, int init, int have_backsub)
{
    KTL(1);
    if (S.ctrl->done) return;
    __shared__ double sm[256];
    __shared__ int s_last;
    __shared__ double Hl[256 * kMaxCamLds + kScal + 8];
    __shared__ double Gall[512 * kMaxCamLds];
    const int t = threadIdx.x;
    const int nc = P.C * kCamSl;
    if ((int)blockIdx.x < nc) cam_reduce_block(P, S, blockIdx.x, sm);
    else board_stats_block(P, S, cand, init, blockIdx.x - nc, sm);
    // hand-off without an L2 write-back (handoff_store): the written-through results are complete, then the count
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) {
        const int old = __hip_atomic_fetch_add(&S.ctrl->fin_count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (old == (int)gridDim.x - 1) ? 1 : 0;
    }
    __syncthreads();
    if (!s_last) return;
    KTLX(0, true);
    if (t == 0) __hip_atomic_store(&S.ctrl->fin_count, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // (buffer_inv: whatever else this workgroup reads from now on is current)
    __syncthreads();
    KTLX(1, true);
    finish_evaluation<true>(P, S, init, have_backsub, /*writer=*/true, Hl, Gall, sm, nullptr, S.ctrl);
    KTLX(8, true);
    KTLX_FLUSH();
}

// ---------------------------------------------------------------------------------------------
// e-block factorisation (SchurEliminator, one 6x6 block per board): ONE LANE per board.
//   V = sum_views E^T E, Jacobi-scaled, damped with D^2 = clamp(diag)/radius, Cholesky L L^T.
// Reads only the E region of the records; writes the board's factor record (kFac doubles): everything the
// Schur-complement and back-substitution kernels need to re-derive Y = L^-1 S_b W column by column.
// grid ceil(B/256) x 256
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool chol6(const double M[21], double L[21])
{
    // packed lower: idx(i,j) = i(i+1)/2 + j; the diagonal slots hold 1 / L_jj
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double d = M[j * (j + 1) / 2 + j];
#pragma unroll
        for (int k = 0; k < j; ++k) d -= L[j * (j + 1) / 2 + k] * L[j * (j + 1) / 2 + k];
        if (!(d > 0.0)) { ok = false; d = 1.0; }
        const double inv = fast_rsqrt(d);      // 1 / L_jj: only the inverse is ever used (forward and back substitution)
        L[j * (j + 1) / 2 + j] = inv;
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

// Factor the damped, Jacobi-scaled 6x6 block of a board from M = sum_views E^T E (packed lower) and g = sum_views E^T r,
// and write the board's factor record (kFac doubles) to f (HBM or LDS).  Returns false if the block is not positive
// definite.
// A board whose pose block is constant (SetParameterBlockConstant) has no e-block: its record is all zeros, which makes
// Y = 0 (no Schur-complement contribution), z = 0 and the back-substituted step exactly 0.
__device__ __forceinline__ bool factor_core(double (&M)[21], const double (&g)[6], const double (&sb)[6], double radius, double dmin, double dmax, double *f,
                                            bool constant_block = false)
{
    if (constant_block) {
#pragma unroll
        for (int i = 0; i < kFac; ++i) f[i] = 0.0;
        return true;
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
    const bool ok = chol6(M, L);
    // forward substitution in multiply-only form: y_i = c_i w_i - sum_{k<i} m_ik y_k
    double z[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const double il = L[i * (i + 1) / 2 + i];
        const double c = sb[i] * il;
        f[kFacC + i] = c;
        f[kFacI + i] = il;
        f[kFacD + i] = D2[i];
        double w = c * g[i];
#pragma unroll
        for (int k = 0; k < i; ++k) {
            const double m = L[i * (i + 1) / 2 + k] * il;
            f[kFacM + i * (i - 1) / 2 + k] = m;
            f[kFacL + i * (i - 1) / 2 + k] = L[i * (i + 1) / 2 + k];
            w -= m * z[k];
        }
        z[i] = w;
        f[kFacZ + i] = w;
    }
    f[54] = 0.0; f[55] = 0.0;
    return ok;
}

// ... of board b with its views at slots [q0, q1), record to HBM
__device__ __forceinline__ void factor_board(const DevProblem &P, const DevState &S, int cur, double radius, double dmin, double dmax,
                                             int b, int q0, int q1)
{
    double sb[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) sb[i] = S.s_b[6 * b + i];
    double M[21], g[6] = { 0, 0, 0, 0, 0, 0 };
#pragma unroll
    for (int i = 0; i < 21; ++i) M[i] = 0.0;
    for (int q = q0; q < q1; ++q) {
        const double *E = rec_e(S.rec[cur], P.V, q), *W = rec_w(S.rec[cur], q), *Rc = S.cconst[cur] + kCStride * P.slot_cam[q];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
#pragma unroll
            for (int j = 0; j <= i; ++j) M[i * (i + 1) / 2 + j] += j < 3 ? E[6 * j + i] : tb_tb(W, Rc, i - 3, j - 3);
            g[i] += W[6 * kFR + i];
        }
    }
    if (!factor_core(M, g, sb, radius, dmin, dmax, S.fac + (size_t)kFac * b, P.board_const[b] != 0)) *S.fac_fail = 1;
}

// stand-alone factorisation of the boards seen by more than three cameras (their Gram products go through
// k_pair_gram); the others are factored inside k_schur_gram.   grid ceil(n_slow/256) x 256
__global__ __launch_bounds__(256) void k_schur_factor(DevProblem P, DevState S)
{
    if (S.ctrl->done) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P.n_slow) return;
    const int b = P.slow_boards[i];
    factor_board(P, S, S.ctrl->cur, S.ctrl->radius, S.ctrl->opt.min_lm_diagonal, S.ctrl->opt.max_lm_diagonal, b, P.bv_ptr[b], P.bv_ptr[b + 1]);
}

// e-block factorisation + Schur complement contributions  T(m_p, m_q) += Y_p^T Y_q  (views p <= q of one board,
// Y = L^-1 S_b W) in ONE launch.  Boards are grouped by their camera set ("signature"); one 4-wave workgroup per
// chunk of <= 64 boards of ONE signature, so the NV(NV+1)/2 16x16 tiles of a chunk map to fixed camera-pair blocks
// and stay in registers as MFMA accumulators.
//   phase 0a  16 lanes per board sum the E records of its views (contiguous: coalesced) into LDS
//   phase 0b  one lane per board: damped Cholesky -> the board's factor record, in LDS
//   (then the records leave for HBM -- the back-substitution needs them -- as one coalesced stream per board)
//   phase 1   a wave takes FOUR boards at a time: lane (a, kq) loads column a of the W records of board kq of the
//             group and runs the 21-FMA forward substitution with that board's multipliers (LDS).  The matrix core
//             contracts over k, and T is a sum over boards -- so the k index of v_mfma_f64_16x16x4 IS the board:
//             for each of the 6 rows r, one MFMA per tile adds sum_{4 boards} Y_p[r][i] Y_q[r][j].  No lane computes
//             anything twice and no operand has to be moved: 6 NT MFMAs and 27 NV FMAs per lane per four boards.
//             The next group's columns are requested before the MFMAs, which cover their latency.
// The four waves' tiles are summed in a fixed order through LDS.
// grid (chunks of this NV) x 256
constexpr int kChunkBoards = 64;


// the forward-substitution values of a board factored by k_schur_factor, wave-uniform through the constant address
// space (scalar loads): k_pair_gram
struct FacFwd { double m[15], c[6], z[6]; };
__device__ __forceinline__ void load_fac_fwd(const double *fac, int board, FacFwd &F)
{
    const cptr4 f = (cptr4)(fac + (size_t)kFac * board);
#pragma unroll
    for (int i = 0; i < 15; ++i) F.m[i] = f[kFacM + i];
#pragma unroll
    for (int i = 0; i < 6; ++i) { F.c[i] = f[kFacC + i]; F.z[i] = f[kFacZ + i]; }
}

// One column of Y = L^-1 S_b W from the column of W.  Column kFR (the gradient column) of EVERY view of a board is
// z = L^-1 S_b (sum over the board's views of E^T r): the reduced right-hand side reads sum_b Y_v^T z from the
// diagonal camera blocks only.
__device__ __forceinline__ void y_column(const FacFwd &F, const double (&w)[6], bool grad_col, double (&y)[6])
{
    y[0] = F.c[0] * w[0];
    y[1] = F.c[1] * w[1] - F.m[0] * y[0];
    y[2] = F.c[2] * w[2] - F.m[1] * y[0] - F.m[2] * y[1];
    y[3] = F.c[3] * w[3] - F.m[3] * y[0] - F.m[4] * y[1] - F.m[5] * y[2];
    y[4] = F.c[4] * w[4] - F.m[6] * y[0] - F.m[7] * y[1] - F.m[8] * y[2] - F.m[9] * y[3];
    y[5] = F.c[5] * w[5] - F.m[10] * y[0] - F.m[11] * y[1] - F.m[12] * y[2] - F.m[13] * y[3] - F.m[14] * y[4];
#pragma unroll
    for (int i = 0; i < 6; ++i) y[i] = grad_col ? F.z[i] : y[i];
}

// ... as the two MFMA operands of a 6-row block of ONE board (k_pair_gram): K = 6 rows as two k-steps of 4 (rows 0..3,
// then rows 4, 5 and two zero rows); lane (a, kq) supplies row kq and row 4 + kq.
__device__ __forceinline__ void y_column_operands(const FacFwd &F, const double (&w)[6], bool grad_col, int kq, double &s0, double &s1)
{
    double y[6];
    y_column(F, w, grad_col, y);
    double y0 = y[0], y1 = y[1], y2 = y[2], y3 = y[3], y4 = y[4], y5 = y[5];
    // (register values, not an indexable array: a select chain over array elements is turned into a dynamic index,
    // and the array then lives in scratch)
    asm volatile("" : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5));
    const bool lo = (kq & 1) == 0, first = kq < 2;
    const double a01 = lo ? y0 : y1, a23 = lo ? y2 : y3, a45 = lo ? y4 : y5;
    s0 = first ? a01 : a23;
    s1 = first ? a45 : 0.0;
}

// the t_b x t_c blocks (3 x 3, one per view) a board's factorisation needs to rebuild its t_b x t_b block, staged in
// LDS as r[3 * jc + l]; indexed like the W record they were taken from so that tb_tb serves both
struct RawTc {
    const double *r;
    __device__ __forceinline__ double operator[](int i) const { return r[3 * (i / 6 - kWcolTc) + (i % 6 - 3)]; }
};

// ctl = 1 (one GPU, <= 8 cameras) or 2 (communicator: the tiles are all-reduced in H_stage); this the only Schur kernel
// of the iteration: the evaluation in front of this launch
// has not been followed by its control step yet -- EVERY workgroup takes it here, in its head (finish_evaluation, LDS
// borrowed from the factor records), on the same inputs and to the same bits; workgroup 0 writes the results.  No
// launch, no hand-off and no single workgroup that the whole chip waits for: what k_reduce_control's last workgroup
// did in 10 us with 255 CUs idle happens here while nothing else could run anyway.
// With ctl the grid has one workgroup more: workgroup 0 writes the step's results (S.ctrl, H, the iteration log), publishes
// the outcome (S.ctl_pub, epoch = ctl_epoch) and does nothing else; workgroups 1 .. first_round - 1 -- those resident when the
// launch starts -- take the step themselves; the workgroups of LATER rounds of the grid (config 5 on one GPU: 1256 chunks,
// 2.5 rounds) start when a first-round workgroup has finished, long after workgroup 0, and read the published outcome:
// the step is paid once per launch, not once per round.
// RIDE (round 5; one GPU, a candidate's evaluation, a grid of ONE round): the reductions behind the evaluation -- k_reduce_stats' blocks,
// 5.2 us + a kernel boundary at config 4 in front of a kernel whose head waits for exactly their results -- ride in this launch.
// Workgroup j + 1 (j < n_stats = 16 C + ceil(B / 256)) takes reduction block j IN FRONT of its own chunk j (a grid of fewer chunks
// than blocks has workgroups that do nothing else); nothing is added to the grid, so everything is resident at once -- as extra
// workgroups the blocks pushed 132 chunks of config 4 into a second round, +5 us -- and the wait below cannot deadlock (the host
// checks max(blocks, chunks) + 1 <= resident workgroups and launches k_reduce_stats otherwise; a block that does not arrive within
// the hand-offs' time bound is a device fault and the solve is run again on separate launches: fault injection 3).  A block
// writes its results through (handoff_store, as it always did), the last block also the snapshot of the LM state, and counts
// itself in: S.stats_count, monotonic over the solve (stats_target = n_stats x the riding launches so far); the last arrival
// copies the count into S.stats_flag, a line of its own, which is what everybody polls -- 370 workgroups polling the COUNTER's line
// held the 143 read-modify-writes on it up by 4.5 us (and a workgroup keeps its slot until its atomic has returned).
// In front of the wait every workgroup requests what the control step reads that the blocks do not write (control_early) and
// then its chunk's RECORDS, from the buffer an accepted step makes current (S.ctrl->cur ^ 1: the state in front of the step;
// whoever reads it after the extra workgroup's commit, or finds the step rejected, asks again behind the step): the 34 MB stream
// while the blocks run.
// Behind the flag the step reads the snapshot (through the scalar cache, as always), two finished sums per thread and the
// statistics partials with PLAIN loads, not handoff_load (500 workgroups x 50 lines read through would queue at the memory side):
// every one of those lines is written by ONE workgroup in this launch (campart2: 256 bytes per block; st_part: a 128-byte line per
// block, kStStride; the snapshot) and by nobody else, nobody reads them in this launch before the flag (the instrumented build's
// scope reads S.ctrl instead of the snapshot for that reason), and the XCDs' L2s and the CUs' vector and scalar caches start a
// launch invalidated (what k_reduce_stats wrote has always reached the next launch's plain loads that way) -- so the first touch
// of a line from an XCD fetches what was written through, or hits the writer's own written-through copy.
#ifndef TSCM_SCHUR_OCC
#define TSCM_SCHUR_OCC 2        // workgroups per CU the NV <= 2 instantiation that serves grids of several rounds is compiled for.  3 (round 6, measured): the
                                // compiler meets 168 VGPRs with 196 bytes of scratch per lane and the kernel takes 63.7 us instead of 51.9 at config 5
#endif
// CB: boards of a chunk at most.  64, or 32 (round 6, an experiment: tscm_debug_experiment(TSCM_EXPERIMENT_SCHUR_CHUNK_32, 1)): half the W columns per thread
// (96 -> 48 VGPRs at NV = 2), 167 registers without a spill, THREE workgroups per CU.  The same arithmetic per board and per
// 4-board group; twice the workgroups and partial tiles.  Measured at config 5: 57.4 us against 52.3 -- not the default.
template <int NV, bool RIDE = false, int CB = kChunkBoards>
__global__ __launch_bounds__(256, NV <= 2 ? (RIDE ? 2 : (CB == 32 ? 3 : TSCM_SCHUR_OCC)) : 1) void k_schur_gram(DevProblem P, DevState S, int chunk0, int ctl, int first_round, int ctl_epoch, int stats_target, int n_chunks)
{
    TL_ONLY(KtlScope ktl_scope(3, ctl && !RIDE ? S.ctrl_snap : static_cast<const CtrlHead *>(S.ctrl));)     // (the snapshot: the writer workgroup advances S.ctrl while later rounds start)
    PHASE_STAMP(tsk);
    // head of the kernel: the control block and the chunk descriptor travel together (one memory round trip), every
    // other address follows from them arithmetically -- the second round trip already brings the data
    // (ctl & 4: the evaluation whose step is taken here is the solve's INITIAL one -- IterationZero: no back-substitution behind it,
    // the Jacobi scaling of the camera columns written by the extra workgroup)
    const int ctl_init = (ctl >> 2) & 1, withhold = (ctl >> 4) & 1;      // (withhold: fault injection, tscm_solver_debug_withhold_handoff(s, 3))
    ctl &= 3;
    const int n_stats = RIDE ? P.C * kCamSl + S.n_st_blocks : 0;
    const int bid = (int)blockIdx.x;
    const bool extra = ctl != 0 && bid == 0;               // the workgroup that writes the control step's results, and nothing else
    const int jblk = ctl ? bid - 1 : bid;                  // (RIDE: reduction block jblk < n_stats in front of chunk jblk < n_chunks)
    const int cblk = RIDE ? min(max(jblk, 0), n_chunks - 1) : max(jblk, 0);
    const int4 desc = P.bc_desc[chunk0 + cblk];
    constexpr int NT = NV * (NV + 1) / 2;
    // what phase 0a gathers per board: sums over its views of E^T E_wb (18) and of E^T r (6), then per view the raw
    // 3 x 3 block t_b x t_c of W (9 NV): the t_b x t_b block is built from those in phase 0b with each view's R_c
    constexpr int NE = 24 + 9 * NV, NJ = (NE + 15) / 16;
    // (one block, so that the control step in the head can borrow all of it: facl first, 16-byte aligned)
    // (round 6: the waves' partial tiles share the space of the E sums, which are dead behind phase 0b's barrier -- 53.2 KB instead of 74.8 at NV = 2: three workgroups per CU fit)
    struct __attribute__((aligned(16))) Lds {
        static_assert(CB == 64 || CB == 32, "four waves x (CB / 16) groups of four boards");
        double facl[CB][kFac];
        union { double sumE[CB][NE]; double tiles[4][NT][256]; double head_rest[256 * kMaxCamLds + kScal + 8 + 512 * kMaxCamLds + 256 - CB * kFac]; };      // (head_rest: what the control step in the head borrows beyond facl)
    };
    __shared__ Lds lds_blk;
    double (&sumE)[CB][NE] = lds_blk.sumE;
    double (&facl)[CB][kFac] = lds_blk.facl;
    constexpr int NG = CB / 16;                     // passes of 16 boards in phase 0a = groups of four boards per wave in phase 1
    double (&tiles)[4][NT][256] = lds_blk.tiles;
    const int chunk = chunk0 + cblk;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int a = lane & 15, kq = lane >> 4;
    const int c0 = desc.x, nbd = desc.y - desc.x, slot0 = desc.z;       // boards c0 .. c0 + nbd - 1 (<= kChunkBoards), views at slots slot0 + NV * i
    constexpr unsigned BAD = 0xffffe000u;
    double ev[NG][NJ];
    double w[NG][NV][6];
    // the boards of a chunk share their camera set: the rotation of view p's camera is chunk-uniform
    const double *Rcp[NV];
    auto request = [&](int cur_) {
        const __amdgpu_buffer_rsrc_t r_w = make_rsrc(S.rec[cur_], sizeof(double) * (size_t)kRec * P.V);
        // ---- requests: the pieces of the records of the boards this lane gathers (phase 0a), the W columns of the four
        //      groups of four boards its wave contracts (phase 1), the Jacobi scaling of the board it factors (phase 0b)
        {
            const int e = tid & 15;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                // entry e + 16 j of the list above: offset of its first term inside the allocation relative to the board's
                // first view, stride between the views' terms (0: a single term)
                const int idx = e + 16 * j;
                const int pv = idx < 24 ? 0 : (idx - 24) / 9, r9 = idx < 24 ? 0 : (idx - 24) % 9;
                const bool summed = idx < 24;
                const unsigned first = idx < 18 ? 8u * ((unsigned)kRecW * (unsigned)P.V + (unsigned)idx)
                                     : idx < 24 ? 8u * (unsigned)(6 * kFR + idx - 18)
                                     : 8u * (unsigned)(kRecW * pv + 6 * (kWcolTc + r9 / 3) + 3 + r9 % 3);
                const unsigned per_slot = idx < 18 ? 8u * kRecE : 8u * kRecW;
#pragma unroll
                for (int i = 0; i < NG; ++i) {
                    const int bf = 16 * i + (tid >> 4);
                    const unsigned o0 = idx < NE ? first + per_slot * (unsigned)(slot0 + NV * min(bf, nbd - 1)) : BAD;
                    double acc = 0.0;
#pragma unroll
                    for (int p = 0; p < NV; ++p) acc += buf_load_f64(r_w, (p == 0 || summed) ? o0 : BAD, per_slot * (unsigned)p);
                    ev[i][j] = acc;
                }
            }
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int bg = 4 * NG * wave + 4 * g + kq;
            const unsigned base = (a < 14 && bg < nbd) ? 8u * ((unsigned)kRecW * (unsigned)(slot0 + NV * bg) + 6u * (unsigned)a) : BAD;
#pragma unroll
            for (int p = 0; p < NV; ++p)
#pragma unroll
                for (int k = 0; k < 3; ++k) {          // column a of the view's W: six adjacent doubles
                    const d2 v = buf_load_2f64(r_w, base, 8u * (unsigned)(kRecW * p + 2 * k));
                    w[g][p][2 * k] = v[0]; w[g][p][2 * k + 1] = v[1];
                }
        }
#pragma unroll
        for (int p = 0; p < NV; ++p) Rcp[p] = S.cconst[cur_] + kCStride * P.slot_cam[slot0 + p];
    };
    int ctrl_done, cur;
    double radius, dmin, dmax;
    TL_ONLY(long long t_waited = 0, t_reduced = 0;)
    if (RIDE && !extra && jblk < n_stats) {
        // a reduction block of the evaluation in front of this launch (k_reduce_stats' body; the candidate's evaluation) before
        // the workgroup's own chunk
        double *sm = reinterpret_cast<double *>(&lds_blk);
        const int blk = jblk, nc = P.C * kCamSl;
        if (blk == n_stats - 1 && threadIdx.x < sizeof(CtrlHead) / 8)
            __hip_atomic_store(&reinterpret_cast<unsigned long long *>(S.ctrl_snap)[threadIdx.x], reinterpret_cast<const unsigned long long *>(S.ctrl)[threadIdx.x],
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (no look at ctrl->done first: a branch on a loaded value is a round trip in front of the block's own loads, and what a
        // finished solve's reductions write nobody reads)
        if (blk < nc) cam_reduce_block(P, S, blk, sm);
        else board_stats_block(P, S, /*cand=*/1, /*init=*/0, blk - nc, sm);
        PHASE_STAMP(tr1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the written-through results are complete, then the count (see "hand-offs")
        __syncthreads();
        PHASE_STAMP(tr2);
        if (threadIdx.x == 0 && !(withhold && blk == 1) && __hip_atomic_fetch_add(S.stats_count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == stats_target - 1)
            __hip_atomic_store(S.stats_flag, stats_target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // the last one: everybody's results are complete
        TL_ONLY(
        t_reduced = wall_clock64();
        if (threadIdx.x == 0 && ktl_scope.on && blk < kKtlGroups) {
            long long *o = g_phs + (size_t)kPhStamps * (2 * kKtlGroups + blk);
            o[0] = tsk; o[1] = tr1; o[2] = tr2; o[3] = t_reduced; o[4] = blk < nc;
        }
        )
        __syncthreads();                                          // (the LDS goes on to the requests' consumers)
    }
    if (RIDE && jblk >= n_chunks) return;                         // (a grid of fewer chunks than reduction blocks)
    // RIDE: the records are requested BEFORE the wait for the riding reductions and the control step, from the buffer an accepted
    // step makes current (S.ctrl->cur is the state in front of the step; whoever reads it after the extra workgroup's commit, or
    // sees the step rejected, asks again below): the 34 MB stream while the reductions run, the control step's own loads come
    // after it.  First-round workgroups only -- a later round finds the outcome published.
    int cur_spec = -1;
    ControlEarly early;
    if (RIDE && !extra && bid < first_round) {
        cur_spec = (S.ctrl->cur ^ 1) & 1;
        control_early(P, S, !ctl_init, early);          // (what the control step reads that the reductions do not write: ahead of the records)
        request(cur_spec);
    }
    if (ctl) {
        constexpr int kHl = 256 * kMaxCamLds + kScal + 8, kGall = 512 * kMaxCamLds;
        static_assert(sizeof(Lds) / sizeof(double) >= kHl + kGall + 256, "finish_evaluation's LDS (C <= 8) fits the kernel's block");
        static_assert(sizeof(Lds) / sizeof(double) >= kHl + 256, "control_outcome's LDS fits the kernel's block");
        __shared__ CtlOut s_ctl;
        double *scratch = reinterpret_cast<double *>(&lds_blk);
        // The LM state comes from the SNAPSHOT the reductions' launch took (k_reduce_stats): the extra workgroup of THIS
        // launch commits the advanced state to S.ctrl while the others may not even have started -- a workgroup that read
        // S.ctrl itself could find the step already taken and take it a second time.  Nobody writes the snapshot here.
        const CtrlHead *head = S.ctrl_snap;
        if (RIDE) {
            // the reductions ride in this launch: their results (and the snapshot) are there when all of them have counted themselves in
            // (a workgroup of a later round finds the count complete)
            __shared__ int s_late;
            if (threadIdx.x == 0) {
                const long long t_start = wall_clock64();
                int late = 0, spins = 0;
                while (__hip_atomic_load(S.stats_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < stats_target) {
                    __builtin_amdgcn_s_sleep(4);
                    if (wall_clock64() - t_start > kHandoffTimeoutTicks) { late = 1; break; }
                    // (a solve that was stopped -- a late hand-off in an EARLIER launch leaves stats_count behind its target for good --
                    // must not make every launch still enqueued behind it spin for the full bound as well)
                    // -- looked at every 256th poll only (about 30 us): polled every time by the 370 waiting workgroups, the control block's
                    // line -- which the head of every workgroup reads and the writer workgroup commits to -- cost the launch 1.1 us (round 6)
                    if (TSCM_AB_DONE_POLL && (++spins & 255) == 0 && __hip_atomic_load(&S.ctrl->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { late = 2; break; }
                }
                if (late == 1) {     // a device fault like any other late hand-off
                    __hip_atomic_store(&S.ctrl->fault, kFaultHandoff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&S.ctrl->term_type, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&S.ctrl->done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                s_late = late;
            }
            __syncthreads();
            if (s_late) return;
            // Behind the flag this workgroup reads what OTHER workgroups of this launch wrote (advisor, round 5): the finished sums and
            // the board statistics' lines are read THROUGH (handoff_load, like every other in-launch hand-off: control_outcome_late<true>,
            // finish_evaluation<true>), and the scalar cache is dropped in front of the snapshot, which control_state reads through
            // the constant address space although its last writer is a reduction block of this very launch.  (An acquire FENCE here
            // -- measured, round 6 -- costs 12.5 us per launch at config 4: it waits for the 34 MB of records requested in front of
            // the wait, which are the point of requesting them there.)
#ifndef TSCM_AB_NO_DCACHE_INV
            asm volatile("s_dcache_inv" ::: "memory");
#endif
        }
        PHASE_STAMP(tsw);
        TL_ONLY(t_waited = tsw;)
        if (head->done) return;
        if (!extra && bid >= first_round) {
            // a later round of the grid: the outcome is published (or about to be)
            if (threadIdx.x == 0) {
                const long long t_start = wall_clock64();
                bool late = false;
                while (__hip_atomic_load(&S.ctl_pub->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < ctl_epoch) {
                    __builtin_amdgcn_s_sleep(4);
                    if (wall_clock64() - t_start > kHandoffTimeoutTicks) { late = true; break; }
                }
                s_ctl.cur = __hip_atomic_load(&S.ctl_pub->cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_ctl.done = __hip_atomic_load(&S.ctl_pub->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_ctl.radius = handoff_load(&S.ctl_pub->radius);
                s_ctl.dmin = head->opt.min_lm_diagonal; s_ctl.dmax = head->opt.max_lm_diagonal;
                if (late) {          // workgroup 0 never reported: a device fault like a late hand-off of the fused solve launch
                    __hip_atomic_store(&S.ctrl->fault, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&S.ctrl->term_type, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&S.ctrl->done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    s_ctl.done = 1;
                }
            }
        } else if (ctl == 2) {
            // communicator path: H_stage holds the all-reduced tiles and scalars -- k_control's work, by every workgroup
            ControlPre pre;
            control_prefetch(P, S, 0, pre, head);
            control_step<true>(P, S, 0, pre, scratch, S.H_stage, S.H_stage + 256 * P.C, nullptr, /*writer=*/extra, &s_ctl);
        } else {
#ifdef TSCM_AB_PLAIN_RIDE_LOADS      // (A/B builds: round 5's plain loads behind the flag)
            constexpr bool TH = false;
#else
            constexpr bool TH = RIDE;
#endif
            if (extra) finish_evaluation<TH>(P, S, ctl_init, !ctl_init, true, scratch, scratch + kHl, scratch + kHl + kGall, &s_ctl, head);
            else if (RIDE && cur_spec >= 0) control_outcome_late<TH>(P, S, ctl_init, early, scratch, scratch + kHl, &s_ctl, head);
            else control_outcome<TH>(P, S, ctl_init, !ctl_init, scratch, scratch + kHl, &s_ctl, head);
        }
        if (extra) {
            // thread 0 took the serial part of the step and committed it: the outcome, written through, then the epoch
            if (threadIdx.x == 0) {
                __hip_atomic_store(&S.ctl_pub->cur, s_ctl.cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&S.ctl_pub->done, s_ctl.done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                handoff_store(&S.ctl_pub->radius, s_ctl.radius);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(&S.ctl_pub->epoch, ctl_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
        }
        __syncthreads();
        // (wave-uniform by construction -- and the compiler has to know: `cur` selects the buffer descriptors)
        ctrl_done = __builtin_amdgcn_readfirstlane(s_ctl.done); cur = __builtin_amdgcn_readfirstlane(s_ctl.cur);
        radius = s_ctl.radius; dmin = s_ctl.dmin; dmax = s_ctl.dmax;
        __syncthreads();
    } else {
        ctrl_done = S.ctrl->done; cur = S.ctrl->cur;
        radius = S.ctrl->radius; dmin = S.ctrl->opt.min_lm_diagonal; dmax = S.ctrl->opt.max_lm_diagonal;
    }
    if (ctrl_done) return;
    PHASE_STAMP(ts0);
    if constexpr (RIDE) {
        if (cur != cur_spec) request(cur);
    } else {
        // (the same requests written out where they always were: this instantiation serves the grids of several rounds -- config 5 --
        // and inlined from the lambda above it came out 2.8 us slower there)
        const double *rec = S.rec[cur];
        // ---- requests: the pieces of the records of the boards this lane gathers (phase 0a), the W columns of the four
        //      groups of four boards its wave contracts (phase 1), the Jacobi scaling of the board it factors (phase 0b)
        const __amdgpu_buffer_rsrc_t r_w = make_rsrc(rec, sizeof(double) * (size_t)kRec * P.V);
        {
            const int e = tid & 15;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                // entry e + 16 j of the list above: offset of its first term inside the allocation relative to the board's
                // first view, stride between the views' terms (0: a single term)
                const int idx = e + 16 * j;
                const int pv = idx < 24 ? 0 : (idx - 24) / 9, r9 = idx < 24 ? 0 : (idx - 24) % 9;
                const bool summed = idx < 24;
                const unsigned first = idx < 18 ? 8u * ((unsigned)kRecW * (unsigned)P.V + (unsigned)idx)
                                     : idx < 24 ? 8u * (unsigned)(6 * kFR + idx - 18)
                                     : 8u * (unsigned)(kRecW * pv + 6 * (kWcolTc + r9 / 3) + 3 + r9 % 3);
                const unsigned per_slot = idx < 18 ? 8u * kRecE : 8u * kRecW;
#pragma unroll
                for (int i = 0; i < NG; ++i) {
                    const int bf = 16 * i + (tid >> 4);
                    const unsigned o0 = idx < NE ? first + per_slot * (unsigned)(slot0 + NV * min(bf, nbd - 1)) : BAD;
                    double acc = 0.0;
#pragma unroll
                    for (int p = 0; p < NV; ++p) acc += buf_load_f64(r_w, (p == 0 || summed) ? o0 : BAD, per_slot * (unsigned)p);
                    ev[i][j] = acc;
                }
            }
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int bg = 4 * NG * wave + 4 * g + kq;
            const unsigned base = (a < 14 && bg < nbd) ? 8u * ((unsigned)kRecW * (unsigned)(slot0 + NV * bg) + 6u * (unsigned)a) : BAD;
#pragma unroll
            for (int p = 0; p < NV; ++p)
#pragma unroll
                for (int k = 0; k < 3; ++k) {          // column a of the view's W: six adjacent doubles
                    const d2 v = buf_load_2f64(r_w, base, 8u * (unsigned)(kRecW * p + 2 * k));
                    w[g][p][2 * k] = v[0]; w[g][p][2 * k + 1] = v[1];
                }
        }
#pragma unroll
        for (int p = 0; p < NV; ++p) Rcp[p] = S.cconst[cur] + kCStride * P.slot_cam[slot0 + p];
    }
    double sb[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) sb[i] = tid < nbd ? S.s_b[6 * (c0 + tid) + i] : 1.0;
    const bool board_is_const = tid < nbd && P.board_const[c0 + tid] != 0;
    // ---- phase 0a: 16 lanes per board, 16 boards per pass ------------------------------------------------------------
    {
        const int e = tid & 15, grp = tid >> 4;
#pragma unroll
        for (int i = 0; i < NG; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) if (e + 16 * j < NE) sumE[16 * i + grp][e + 16 * j] = ev[i][j];
    }
    __syncthreads();
    PHASE_STAMP(ts1);
    // ---- phase 0b: one lane per board --------------------------------------------------------------------------------
    if (tid < nbd) {
        double M[21], g[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                if (j < 3) M[i * (i + 1) / 2 + j] = sumE[tid][6 * j + i];
                else {
                    double acc = 0.0;
#pragma unroll
                    for (int p = 0; p < NV; ++p) acc += tb_tb(RawTc{ &sumE[tid][24 + 9 * p] }, Rcp[p], i - 3, j - 3);
                    M[i * (i + 1) / 2 + j] = acc;
                }
            }
            g[i] = sumE[tid][18 + i];
        }
        if (!factor_core(M, g, sb, radius, dmin, dmax, facl[tid], board_is_const)) *S.fac_fail = 1;
    }
    __syncthreads();
    PHASE_STAMP(ts2);
    // the factor records leave for HBM (the back-substitution reads them): one contiguous stream for the chunk
    for (int i = tid; i < nbd * kFac; i += 256) S.fac[(size_t)kFac * c0 + i] = (&facl[0][0])[i];
    // ---- phase 1 ---------------------------------------------------------------------------------------------------------
    d4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = d4{ 0.0, 0.0, 0.0, 0.0 };
    const bool grad_col = a == kFR;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (4 * NG * wave + 4 * g >= nbd) break;                          // wave-uniform
        const int bg = 4 * NG * wave + 4 * g + kq;
        const bool valid = bg < nbd;
        const int bl = min(bg, nbd - 1);
        FacFwd F;
#pragma unroll
        for (int i = 0; i < 15; ++i) F.m[i] = facl[bl][kFacM + i];
#pragma unroll
        for (int i = 0; i < 6; ++i) { F.c[i] = facl[bl][kFacC + i]; F.z[i] = valid ? facl[bl][kFacZ + i] : 0.0; }
        double y[NV][6];
#pragma unroll
        for (int p = 0; p < NV; ++p) y_column(F, w[g][p], grad_col, y[p]);      // lanes without a board: w = 0, z = 0 -> y = 0
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            int t = 0;
#pragma unroll
            for (int p = 0; p < NV; ++p)
#pragma unroll
                for (int q = p; q < NV; ++q, ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(y[p][r], y[q][r], acc[t], 0, 0, 0);
        }
    }
    PHASE_STAMP(ts3);
    // D layout: lane (col = a, kq) holds rows kq + 4 r of column a -> tile entry [row][col]
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) tiles[wave][t][(kq + 4 * r) * 16 + a] = acc[t][r];
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NT; ++t)
        S.pairpart[(size_t)256 * P.bc_tile[6 * chunk + t] + tid] = (tiles[0][t][tid] + tiles[1][t][tid]) + (tiles[2][t][tid] + tiles[3][t][tid]);
    TL_ONLY(
    if (threadIdx.x == 0 && ktl_scope.on && (int)blockIdx.x < kKtlGroups) {
        long long *o = g_phs + (size_t)kPhStamps * blockIdx.x;
        o[0] = tsk; o[1] = ts0; o[2] = ts1; o[3] = ts2; o[4] = ts3; o[5] = wall_clock64(); o[6] = nbd; o[7] = (bid >= first_round ? 1 : 0) | ((RIDE && t_waited ? t_waited - tsk : 0) << 1) | ((RIDE && t_reduced ? t_reduced - tsk : 0) << 32);      // (bit 0: a later round; above: ticks until the riding reductions had arrived)
    }
    )
    PH_ONLY(
    if (threadIdx.x == 0 && (cblk == 0 || cblk == 200))
        printf("schur_gram wg %d: boards %d  head %lld  E sums %lld  factor %lld  gram %lld  tiles %lld [10 ns]\n", (int)blockIdx.x, nbd, ts0 - tsk, ts1 - ts0, ts2 - ts1, ts3 - ts2, wall_clock64() - ts3);
    )
}

// Fallback for boards seen by more than three cameras: explicit list of view pairs, pre-sorted by
// camera-pair block; one 4-wave workgroup per chunk of pairs of a single block.   grid n_pchunks x 256
__global__ __launch_bounds__(256) void k_pair_gram(DevProblem P, DevState S)
{
    if (S.ctrl->done) return;
    __shared__ double red[4][256];
    const int pc = blockIdx.x;
    const int cur = S.ctrl->cur;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int a = lane & 15, kq = lane >> 4;
    d4 acc = { 0.0, 0.0, 0.0, 0.0 };
    for (int p = P.pc_begin[pc] + wave; p < P.pc_end[pc]; p += 4) {
        const double *Wi = rec_w(S.rec[cur], P.pair_i[p]), *Wj = rec_w(S.rec[cur], P.pair_j[p]);
        FacFwd F;
        load_fac_fwd(S.fac, __builtin_amdgcn_readfirstlane(P.pair_board[p]), F);     // p is wave-uniform
        double wi[6], wj[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) { wi[k] = a < 14 ? Wi[6 * a + k] : 0.0; wj[k] = a < 14 ? Wj[6 * a + k] : 0.0; }
        double i0, i1, j0, j1;
        y_column_operands(F, wi, a == kFR, kq, i0, i1);
        y_column_operands(F, wj, a == kFR, kq, j0, j1);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(i0, j0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(i1, j1, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][(kq + 4 * r) * 16 + a] = acc[r];
    __syncthreads();
    const int t = threadIdx.x;
    S.pairpart[(size_t)256 * P.pc_tile[pc] + t] = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
}

constexpr int kTEntries = 64, kTSlices = 16;
// block blk of a (n_bids * 256 / ENTRIES)-block grid of ENTRIES * kTSlices threads, partial tiles [cb, ce) of its
// camera-pair block; the summation order of an entry depends on kTSlices only, so every geometry produces the same bits
template <int ENTRIES>
__device__ __forceinline__ void t_reduce_block(const DevState &S, int bid, int part, int cb, int ce, double (*red)[ENTRIES])
{
    const int e = threadIdx.x % ENTRIES, slice = threadIdx.x / ENTRIES;
    const int entry = part * ENTRIES + e;
    const int per = (ce - cb + kTSlices - 1) / kTSlices;
    const int b0 = cb + slice * per, b1 = min(ce, b0 + per);
    // the partial tiles of one block are stored contiguously; eight loads in flight per thread, the ragged end
    // included (no dependent tail loop)
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, a4 = 0.0, a5 = 0.0, a6 = 0.0, a7 = 0.0;
    const double *src = S.pairpart + entry;
    for (int c = b0; c < b1; c += 8) {
        const double v0 = src[(size_t)256 * c];
        const double v1 = c + 1 < b1 ? src[(size_t)256 * (c + 1)] : 0.0;
        const double v2 = c + 2 < b1 ? src[(size_t)256 * (c + 2)] : 0.0;
        const double v3 = c + 3 < b1 ? src[(size_t)256 * (c + 3)] : 0.0;
        const double v4 = c + 4 < b1 ? src[(size_t)256 * (c + 4)] : 0.0;
        const double v5 = c + 5 < b1 ? src[(size_t)256 * (c + 5)] : 0.0;
        const double v6 = c + 6 < b1 ? src[(size_t)256 * (c + 6)] : 0.0;
        const double v7 = c + 7 < b1 ? src[(size_t)256 * (c + 7)] : 0.0;
        a0 += v0; a1 += v1; a2 += v2; a3 += v3; a4 += v4; a5 += v5; a6 += v6; a7 += v7;
    }
    red[slice][e] = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
    __syncthreads();
    if (slice == 0) {
        // tiles without a local partial (the pair is only seen on other ranks) are written as zeros
        double v[kTSlices];
#pragma unroll
        for (int q = 0; q < kTSlices; ++q) v[q] = red[q][e];
#pragma unroll
        for (int w = kTSlices / 2; w >= 1; w >>= 1)
#pragma unroll
            for (int q = 0; q < w; ++q) v[q] += v[q + w];
        handoff_store(&S.T[(size_t)256 * bid + entry], v[0]);      // (written through: the fused launch hands T over inside the launch)
    }
}
// grid (n_bids * 256 / kTEntries) x 1024: block (bid, part) sums kTEntries entries of the partial tiles that belong to
// one camera-pair block, the tile list split kTSlices ways across the threads of an entry (the kernel is a chain of
// memory round trips: the more of the list is in flight at once, the shorter it is)
__global__ __launch_bounds__(kTEntries * kTSlices) void k_T_reduce(DevProblem P, DevState S)
{
    if (S.ctrl->done) return;
    __shared__ double red[kTSlices][kTEntries];
    constexpr int kParts = 256 / kTEntries;
    const int bid = blockIdx.x / kParts;
    t_reduce_block<kTEntries>(S, bid, blockIdx.x % kParts, P.bid_part_ptr[bid], P.bid_part_ptr[bid + 1], red);
}

// T(i, j) for padded columns i, j of the camera side; the lower blocks are the transposed upper ones
__device__ __forceinline__ double load_T_lut(const DevProblem &P, const double *T, int i, int j)
{
    int lo = i >> 4, hi = j >> 4, a = i & 15, b = j & 15;
    if (lo > hi) { const int t = lo; lo = hi; hi = t; const int u = a; a = b; b = u; }
    const int tile = P.bid_lut[lo * P.C + hi];
    return tile >= 0 ? T[(size_t)256 * tile + a * 16 + b] : 0.0;
}

// Common end of the reduced-system solvers: yhat = S_c y (camera step = -yhat), the candidate camera parameters, and
// the camera part of the model cost change / step norm.  One thread per padded column (n_pad <= workgroup size in
// every variant).  The global operands of the tail -- the column's parameter and its row of H -- do not depend on
// the solution: tail_prefetch() issues their loads early (before the back-substitution where registers allow), so
// the tail itself waits for no memory.
struct TailOperands { double x, hg, hrow[kFA]; };
__device__ __forceinline__ void tail_prefetch(const DevProblem &P, const DevState &S, int cur, const double *H, TailOperands &o)
{
    const int i = threadIdx.x;
    o.x = 0.0; o.hg = 0.0;
#pragma unroll
    for (int b = 0; b < kFA; ++b) o.hrow[b] = 0.0;
    if (i < P.n_pad) {
        const int m = i >> 4, ai = i & 15;
        if (ai < 6) o.x = S.cam_rt[cur][6 * m + ai];
        else if (ai < 15) o.x = S.intr[cur][9 * m + (ai - 6)];
        if (ai < kFA) {
#pragma unroll
            for (int b = 0; b < kFA; ++b) o.hrow[b] = H[256 * m + ai * 16 + b];
            o.hg = H[256 * m + ai * 16 + kFR];
        }
    }
}
// yv: solution by padded column (LDS); s_sc, s_yh, s_act: LDS arrays of n_pad entries.  Every thread of the workgroup calls it.
// publish_epoch > 0: workgroups of this launch wait for the step (backsub_body<.., true>): yhat and the candidate camera
// parameters are written through, and once they are complete thread 0 sets y_flag = 2 * epoch + fail
__device__ __forceinline__ void reduced_solution_tail(const DevProblem &P, const DevState &S, int cur, int fail, const TailOperands &o,
                                                      const double *yv, const double *s_sc, double *s_yh, const unsigned char *s_act, double *sred,
                                                      int publish_epoch = 0)
{
    const int n = P.n_pad, i = threadIdx.x;
    const int m = i >> 4, ai = i & 15;
    double model = 0.0, stepsq = 0.0, yh = 0.0;
    if (i < n) {
        const bool act = s_act[i] && !fail;
        yh = act ? s_sc[i] * yv[i] : 0.0;
        handoff_store(&S.yhat[i], yh);
        s_yh[i] = yh;
        if (ai < kFA) {
            const double x = o.x;
            const double xn = x + (-yh);
            if (ai < 6) handoff_store(&S.cam_rt[cur ^ 1][6 * m + ai], xn); else handoff_store(&S.intr[cur ^ 1][9 * m + (ai - 6)], xn);
            const double d = x - xn; stepsq = d * d;
        } else if (ai < 15) {
            handoff_store(&S.intr[cur ^ 1][9 * m + (ai - 6)], o.x);   // b, c are inert
        }
    }
    if (publish_epoch > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (publish_epoch > 0 && i == 0) __hip_atomic_store(S.y_flag, 2 * publish_epoch + (fail ? 1 : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // the candidate's per-camera records for the next evaluation: the waiting workgroups are busy now, this one is not.
    // (Plain loads: this workgroup wrote the parameters itself, through its own L2, and never had them in its L1.)
    if (publish_epoch > 0 && i >= 64 && i < 64 + P.C) write_camera_record(S, cur ^ 1, i - 64);
    // model_cam = yhat^T g_c - 1/2 yhat^T H_cc yhat   (block diagonal H_cc)
    if (i < n && ai < kFA && yh != 0.0) {
        double hy = 0.0;
#pragma unroll
        for (int b = 0; b < kFA; ++b) hy += o.hrow[b] * s_yh[m * 16 + b];
        model = yh * (o.hg - 0.5 * hy);
    }
    { double red[2] = { model, stepsq }, mdummy = 0.0; block_reduce256<2>(red, mdummy, sred); model = red[0]; stepsq = red[1]; }
    if (i == 0) { S.ctrl->model_cam = model; S.ctrl->stepsq_cam = stepsq; S.ctrl->lin_fail = fail; *S.fac_fail = 0; }
}

// ---------------------------------------------------------------------------------------------
// Reduced camera system (DenseSchurComplementSolver), up to kMaxCamLds cameras: k_solve_nd (tscm_solve_nd.h).
// ---------------------------------------------------------------------------------------------
constexpr int kFusedEntries = 256 / kTSlices;        // fused T reduction: 16 entries x 16 slices = the solver's 256 threads
// epoch: 1, 2, ... = the number of fused launches of this solve so far, this one included (the host resets the
// counter to zero in front of every solve).  The arrival counter is MONOTONIC: a launch waits for epoch * producers,
// so late arrivals of a launch that was given up on can never be mistaken for this launch's.
template <int NTH, bool WAIT>
__device__ __forceinline__ void backsub_body(const DevProblem &P, const DevState &S, int with_floats, const int blk, const int nblk, const int epoch, const int t_need);

#include "tscm_solve_nd.h"
#include "tscm_solve_dense4.h"

// ---------------------------------------------------------------------------------------------
// Reduced camera system of rigs with more than kMaxCamLds cameras (up to kMaxCam: 410 free columns): the
// compact system no longer fits registers + LDS, so ONE 1024-thread workgroup runs a blocked right-looking
// Cholesky (16-column panels) on the matrix in global memory (S.Abig, L2-resident: <= 1.4 MB).  The right-hand
// side rides along as row N of the matrix, so the forward substitution w = L^{-1} b falls out of the panel
// solves and trailing updates.
//   * The CURRENT panel (diagonal block + rows below, transposed) lives in LDS: wave 0 factors the 16x16
//     diagonal block, one thread per row solves its panel row in place, and the 4x4 register tiles of the
//     rank-16 update write the columns of the NEXT panel into the other LDS buffer -- no global-memory round
//     trip sits on the panel-to-panel critical path.
//   * Every 4x4 tile of the trailing matrix is owned by the same thread for the whole factorisation (absolute
//     tile index modulo the 32 x 32 thread grid), so its read-modify-write sequence in global memory is
//     thread-private and the per-panel barriers only order LDS.
//   * Back-substitution: the rows of L a panel needs do not depend on the solution, so they are prefetched one
//     panel ahead of the 16 x 16 triangular solve.
// Same arithmetic as k_solve_reduced up to the summation order inside the updates.
// grid 1 x 1024, dynamic LDS solve_big_lds_bytes(N, n_pad).
// ---------------------------------------------------------------------------------------------
constexpr int kBigNT = 1024;
constexpr int kBigBatch = 4;        // blocks of one block row in flight per trip of the trailing update
__host__ __device__ inline size_t solve_big_lds_bytes(int N, int n_pad)
{
    return sizeof(double) * ((size_t)2 * 16 * (N + 16) + 16 * 17 + 2 * (size_t)N + 16 + 3 * (size_t)n_pad) + sizeof(int) * (size_t)N + (size_t)n_pad;
}
// Storage of the big reduced system: packed lower triangle of 16x16 blocks, each block in the register layout of the
// fp64 MFMA accumulator (lane = col + 16 * (row & 3) holds rows (row & 3) + 4 g, g = 0..3, as four consecutive doubles):
// the trailing update moves a block with ONE 32-byte load and store per lane.
__device__ __forceinline__ size_t big_block(int I, int J) { return ((size_t)(I * (I + 1) / 2 + J)) << 8; }
__device__ __forceinline__ size_t big_idx(int r, int c)
{
    return big_block(r >> 4, c >> 4) + (size_t)((((c & 15) + 16 * (r & 3)) << 2) + ((r & 15) >> 2));
}
__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
// barrier that orders LDS only (global loads stay in flight across it)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(kBigNT) void k_solve_reduced_big(DevProblem P, DevState S)
{
    extern __shared__ __attribute__((aligned(32))) double lds[];
    const int n = P.n_pad, na = P.n_act;
    const int N = (na + 15) & ~15, NP = N >> 4;     // compact columns rounded up to whole panels (identity padding)
    const int XP = N + 16;                          // pitch of the transposed panel buffers (rhs row + 15 scratch rows)
    double *pbuf = lds;               // [2][16][XP] panel k: rows k0 .. N (the rhs row) at index row - k0, transposed
    double *Ld = pbuf + 2 * 16 * XP;  // [16][17] diagonal block
    double *wv = Ld + 16 * 17;        // [N] w = L^{-1} b, then overwritten with y
    double *idg = wv + N;             // [N] 1 / L_kk
    double *yk = idg + N;             // [16] solution of the current panel
    double *yv = yk + 16;             // [n_pad] solution by padded column
    double *s_sc = yv + n;            // [n_pad]
    double *s_yh = s_sc + n;          // [n_pad]
    int *s_map = reinterpret_cast<int *>(s_yh + n);                         // [N]
    unsigned char *s_act = reinterpret_cast<unsigned char *>(s_map + N);    // [n_pad]
    __shared__ int s_fail;
    __shared__ double sred[256];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (S.ctrl->done) return;
    const int cur = S.ctrl->cur;
    const double radius = S.ctrl->radius;
    const double dmin = S.ctrl->opt.min_lm_diagonal, dmax = S.ctrl->opt.max_lm_diagonal;
    const double *H = S.H[cur];
    double *A = S.Abig;               // packed lower triangle of 16x16 blocks (big_idx), block row NP: the rhs row + scratch
    for (int i = tid; i < n; i += kBigNT) { s_sc[i] = S.s_c[i]; s_act[i] = P.col_active[i]; yv[i] = 0.0; }
    for (int i = tid; i < N; i += kBigNT) s_map[i] = i < na ? P.act_map[i] : -1;
    if (tid == 0) s_fail = S.ctrl->lin_fail | *S.fac_fail;
    __syncthreads();
    // ---- build the lower triangle and the rhs row; the first panel goes straight to LDS -------------
    // wave w owns rows w, w + 16, ...; 16 rows are in flight per trip (one memory round trip per 16 x 64 entries),
    // lanes run along the columns: T (both triangles filled in by k_T_reduce) and A are read / written row-wise
    BIG_ONLY(long long tp0 = wall_clock64(), tp_diag = 0, tp_solve = 0, tp_upd = 0;)
    for (int kb = 0; wave + 16 * kb <= N; kb += 16) {
        const int rmax = min(N, wave + 16 * (kb + 15));
        for (int c = lane; c <= rmax && c < N; c += 64) {
            const int j = s_map[c];
            const int mj = j >> 4, bj = j & 15;
            const double scj = j >= 0 ? s_sc[j] : 0.0;
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int r = wave + 16 * (kb + u);
                v[u] = (r == c) ? 1.0 : 0.0;
                if (r > N || c > r) continue;
                if (r == N) {
                    v[u] = j >= 0 ? scj * (H[256 * mj + bj * 16 + kFR] - load_T_lut(P, S.T, j, mj * 16 + kFR)) : 0.0;
                } else {
                    const int i = s_map[r];
                    if (i >= 0 && j >= 0) {
                        const int mi = i >> 4, ai = i & 15;
                        const double h = (mi == mj) ? H[256 * mi + ai * 16 + bj] : 0.0;
                        double t = s_sc[i] * scj * (h - load_T_lut(P, S.T, i, j));
                        if (i == j) t += fmin(fmax(scj * scj * h, dmin), dmax) / radius;
                        v[u] = t;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int r = wave + 16 * (kb + u);
                if (r > N || c > r) continue;
                if (c < 16) pbuf[c * XP + r] = v[u];
                else A[big_idx(r, c)] = v[u];
            }
        }
    }
    __syncthreads();
    BIG_ONLY(long long tp1 = wall_clock64();)
    // ---- factorisation -------------------------------------------------------------------------------
    for (int tk = 0; tk < NP; ++tk) {
        const int k0 = tk * 16, m0 = k0 + 16;
        double *pc = pbuf + (tk & 1) * (16 * XP);            // this panel (index row - k0)
        double *pn = pbuf + ((tk & 1) ^ 1) * (16 * XP);      // next panel (index row - m0)
        BIG_ONLY(const long long q0 = wall_clock64();)
        if (wave == 0) {
            // Cholesky of the 16x16 diagonal block in REGISTERS: lane r holds row r, the pivot and the column entries
            // travel through v_readlane (no LDS hand-off on the column-to-column dependent chain)
            const int r = lane & 15;
            double a[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) a[c] = pc[c * XP + r];          // entries right of the diagonal are never used
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                double d = readlane_f64(a[c], c);
                if (!(d > 0.0)) { d = 1.0; if (lane == 0) s_fail = 1; }
                const double isd = fast_rsqrt(d);
                const double l = a[c] * isd;                             // lane c: d / sqrt(d); lanes below: L[r][c]
                a[c] = l;
                if (lane == c) idg[k0 + c] = isd;
#pragma unroll
                for (int q = c + 1; q < 16; ++q) a[q] -= l * readlane_f64(l, q);
            }
            if (lane < 16) {
#pragma unroll
                for (int c = 0; c < 16; ++c) if (c <= r) { Ld[r * 17 + c] = a[c]; A[big_idx(k0 + r, k0 + c)] = a[c]; }
            }
        }
        lds_barrier();
        BIG_ONLY(const long long q1 = wall_clock64();)
        // panel rows m0 .. N: x = a L_kk^{-T}, one thread per row, in place in LDS (+ the final L row to global)
        for (int r = m0 + tid; r <= N; r += kBigNT) {
            double x[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) x[c] = pc[c * XP + (r - k0)];
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                double v = x[c];
#pragma unroll
                for (int q = 0; q < c; ++q) v -= x[q] * Ld[c * 17 + q];
                x[c] = v * idg[k0 + c];
            }
#pragma unroll
            for (int c = 0; c < 16; ++c) { A[big_idx(r, k0 + c)] = x[c]; pc[c * XP + (r - k0)] = x[c]; }
        }
        lds_barrier();
        BIG_ONLY(const long long q2 = wall_clock64();)
        // trailing update A_IJ -= X_I X_J^T on 16x16 blocks (four v_mfma_f64_16x16x4 each); block (I, J) belongs to
        // wave (I % 4, J % 4) for the whole factorisation; block row NP is the rhs row (rows past N: scratch).
        // kBigBatch blocks of a block row are in flight per trip; the blocks of the next panel's columns land in LDS.
        {
            const int col = lane & 15, kq = lane >> 4;
            const int Tb = tk + 1;
            for (int I = Tb + (((wave >> 2) - Tb) & 3); I <= NP; I += 4) {
                double xa[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) xa[t] = -pc[(4 * t + kq) * XP + (16 * I - k0) + col];
                for (int J0 = Tb + (((wave & 3) - Tb) & 3); J0 <= I && J0 < NP; J0 += 4 * kBigBatch) {
                    d4 acc[kBigBatch];
#pragma unroll
                    for (int u = 0; u < kBigBatch; ++u) {
                        const int J = J0 + 4 * u;
                        const int Jc = (J <= I && J < NP) ? J : J0;        // past the end: a harmless duplicate of the first block
                        acc[u] = *reinterpret_cast<const d4 *>(A + big_block(I, Jc) + 4 * lane);
                    }
#pragma unroll
                    for (int u = 0; u < kBigBatch; ++u) {
                        const int J = J0 + 4 * u;
                        if (!(J <= I && J < NP)) continue;                 // wave-uniform
                        double xb[4];
#pragma unroll
                        for (int t = 0; t < 4; ++t) xb[t] = pc[(4 * t + kq) * XP + (16 * J - k0) + col];
#pragma unroll
                        for (int t = 0; t < 4; ++t) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[t], xb[t], acc[u], 0, 0, 0);
                    }
#pragma unroll
                    for (int u = 0; u < kBigBatch; ++u) {
                        const int J = J0 + 4 * u;
                        if (!(J <= I && J < NP)) continue;
                        if (J == Tb) {
#pragma unroll
                            for (int g = 0; g < 4; ++g) pn[col * XP + 16 * (I - Tb) + kq + 4 * g] = acc[u][g];
                        } else {
                            *reinterpret_cast<d4 *>(A + big_block(I, J) + 4 * lane) = acc[u];
                        }
                    }
                }
            }
        }
        lds_barrier();
        BIG_ONLY({ const long long q3 = wall_clock64(); tp_diag += q1 - q0; tp_solve += q2 - q1; tp_upd += q3 - q2; })
    }
    __syncthreads();
    BIG_ONLY(long long tp2 = wall_clock64();)
    // ---- back-substitution L^T y = w (w = row N of the factor) -------------------------------------
    for (int i = tid; i < N; i += kBigNT) wv[i] = A[big_idx(N, i)];
    // rows of L for the first (= last) panel; thread i keeps L[k0 + c][i], wave 0 also the diagonal block
    double lrow[16], ldg[4];
    {
        const int k0 = (NP - 1) * 16;
#pragma unroll
        for (int c = 0; c < 16; ++c) lrow[c] = (NP > 0 && tid < k0) ? A[big_idx(k0 + c, tid)] : 0.0;
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int e = lane + 64 * u, r = e >> 4, c = e & 15; ldg[u] = (NP > 0 && wave == 0 && c <= r) ? A[big_idx(k0 + r, k0 + c)] : 0.0; }
    }
    __syncthreads();
    for (int tk = NP - 1; tk >= 0; --tk) {
        const int k0 = tk * 16;
        double lnext[16], dnext[4];
        {
            const int kn = k0 - 16;
#pragma unroll
            for (int c = 0; c < 16; ++c) lnext[c] = (tk > 0 && tid < kn) ? A[big_idx(kn + c, tid)] : 0.0;
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int e = lane + 64 * u, r = e >> 4, c = e & 15; dnext[u] = (tk > 0 && wave == 0 && c <= r) ? A[big_idx(kn + r, kn + c)] : 0.0; }
        }
        if (wave == 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int e = lane + 64 * u; Ld[(e >> 4) * 17 + (e & 15)] = ldg[u]; }
            wave_lds_fence();
            if (lane == 0) {
                double y[16];
#pragma unroll
                for (int c = 15; c >= 0; --c) {
                    double v = wv[k0 + c];
#pragma unroll
                    for (int q = c + 1; q < 16; ++q) v -= Ld[q * 17 + c] * y[q];
                    y[c] = v * idg[k0 + c];
                }
#pragma unroll
                for (int c = 0; c < 16; ++c) { yk[c] = y[c]; wv[k0 + c] = y[c]; }
            }
        }
        lds_barrier();
        if (tid < k0) {
            double v = wv[tid];
#pragma unroll
            for (int c = 0; c < 16; ++c) v -= lrow[c] * yk[c];
            wv[tid] = v;
        }
        lds_barrier();
#pragma unroll
        for (int c = 0; c < 16; ++c) lrow[c] = lnext[c];
#pragma unroll
        for (int u = 0; u < 4; ++u) ldg[u] = dnext[u];
    }
    TailOperands tail_ops;
    tail_prefetch(P, S, cur, H, tail_ops);
    for (int i = tid; i < na; i += kBigNT) yv[s_map[i]] = wv[i];          // back to padded columns
    __syncthreads();
    BIG_ONLY(long long tp3 = wall_clock64();)
    reduced_solution_tail(P, S, cur, s_fail, tail_ops, yv, s_sc, s_yh, s_act, sred);
    BIG_ONLY(
    if (tid == 0) printf("big solve N=%d  build %lld  factor %lld (diag %lld solve %lld update %lld)  backsub %lld  tail %lld  [10 ns ticks]\n", N,
                         tp1 - tp0, tp2 - tp1, tp_diag, tp_solve, tp_upd, tp3 - tp2, wall_clock64() - tp3);
    )
}

// Back-substitution of the board steps (SchurEliminator::BackSubstitute) AND the per-view constants of the candidate
// point (what k_view_prep computes for the initial point) in one launch.  A workgroup of NTH threads owns NTH / 8
// consecutive boards, whose views are consecutive record slots.  Small workgroups on purpose: the kernel streams the W
// region, a CU sustains ~20 GB/s of it, so the time is set by the CU with the most bytes -- thousands of small
// workgroups spread evenly, a few hundred large ones leave CUs with one or with two of them (measured: 2.3 TB/s).
//   phase A  q_b = sum_v W_v yhat[m_v]: 16 lanes per slot, lane a loads column a of the slot's W record (six adjacent doubles,
//            three 16-byte loads) straight into registers, multiplies by yhat[m_v][a] and the 16 lanes sum (DPP).  Round 5:
//            before, the records of a round were read flat into LDS and the columns taken from there -- two barriers and an LDS
//            round trip per round of 32 slots, with the next round's loads behind the first; tools/ubench_stream.hip: the column
//            pattern streams exactly as fast as the flat one (7.1 TB/s).  Everything of a round of 64 slots is requested at once
//   phase B  one lane per board: y_b = L^{-T} (z - L^{-1} S_b q_b);  delta_b = -s_b y_b;  candidate = x + delta
//            (sum_v Y_v yhat = L^{-1} S_b sum_v W_v yhat: one forward substitution per board);  the candidate rotations
//            R_c of the cameras are prepared by lanes of the second wave
//   phase C  one lane per view of these boards: board rotation columns, t_b and R_c dR_b/dw of the candidate, staged in
//            LDS and written as 256-byte records (vconst, see k_view_prep); workgroup 0 also writes the per-camera records
// grid ceil(B / (NTH / 8)) x NTH, dynamic LDS BsGeom<NTH>::kLds doubles
// NTH threads own NTH / 8 boards; a round of phase A is 32 slots = 2688 doubles of W (21 per thread at 128 threads, 10.5
// at 256: the last load of a thread is masked), the factor records are 7 doubles per thread.  Two geometries: 128
// threads / 16 boards (thousands of small workgroups: balanced on mid-size problems) and 256 threads / 32 boards for
// problems with more groups of 16 than fit the chip at once -- the serial phases B and C cost a workgroup the same ~5 us
// whatever its size, so larger workgroups halve their share per board.
constexpr int kBsTile = 64;        // views per round of phase C
template <int NTH> struct BsGeom {
    static constexpr int kBoards = NTH / 8;
    static constexpr int kPassSlots = NTH / 16;                                        // phase A: 16 lanes per slot
    static constexpr int kPasses = 4;                                                  // ... four passes per round: 24 doubles of W per thread
    static constexpr int kRoundSlots = kPasses * kPassSlots;                           // 64 slots at 256 threads, 32 at 128
    static constexpr int kLdsA = kRoundSlots * 6, kLdsB = kBoards * kFac, kLdsC = kBsTile * (kVFloatOff + 1);
    static constexpr int kLds = kLdsC > kLdsB ? (kLdsC > kLdsA ? kLdsC : kLdsA) : (kLdsB > kLdsA ? kLdsB : kLdsA);     // dynamic LDS, doubles
    static_assert(kBoards * kFac == 7 * NTH, "the factor records are 7 doubles per thread");
    static_assert(kBoards <= 64 && 64 + kMaxCam <= NTH, "lane roles of phase B: the boards in wave 0, the cameras from wave 1 on");
};

// WAIT: the workgroup runs inside the reduced solve's launch (k_solve_reduced<..., true>, one GPU).  Everything that does
// not depend on the camera step -- the first round of W records, the factor records, the boards' poses -- is requested
// at once; then thread 0 waits for the solver's flag (y_flag = 2 * epoch + lin_fail, monotonic like the T counter and
// with the same time bound) while the solver workgroup works, alone on the chip otherwise.  What the solver wrote is
// written through (handoff_store) and read behind an acquire fence.
template <int NTH, bool WAIT>
__device__ __forceinline__ void backsub_body(const DevProblem &P, const DevState &S, int with_floats, const int blk, const int nblk, const int epoch, const int t_need)
{
    constexpr int kBsBoards = BsGeom<NTH>::kBoards, kBsThreads = NTH, kPassSlots = BsGeom<NTH>::kPassSlots, kPasses = BsGeom<NTH>::kPasses, kRoundSlots = BsGeom<NTH>::kRoundSlots;
#ifndef TSCM_BS_CAM_LANE0
#define TSCM_BS_CAM_LANE0 64
#endif
    constexpr int kBsCamLane0 = TSCM_BS_CAM_LANE0;
    // head: control block and slot range in one round trip
    const int b0 = blk * kBsBoards;
    const int nbl = min(kBsBoards, P.B - b0);
    const int s0 = P.bv_ptr[b0], s1 = P.bv_ptr[b0 + nbl];                // the views of these boards: slots [s0, s1)
    const int ctrl_done = S.ctrl->done, cur = S.ctrl->cur;
    int fail = WAIT ? 0 : S.ctrl->lin_fail;
    if (ctrl_done) return;
    extern __shared__ __attribute__((aligned(16))) double dyn[];
    double (*s_qv)[6] = reinterpret_cast<double (*)[6]>(dyn);     // [kRoundSlots][6] W yhat per slot of the round   phase A
    double *s_fac = dyn;                                   // [kBsBoards][kFac]   phase B
    double *st_all = dyn;                                  // [kBsTile][kVFloatOff + 1]  phase C
    __shared__ double s_q[kBsBoards][6], s_new[kBsBoards][6];
    __shared__ double s_yh[16 * kMaxCam];                  // phase A
    // candidate R_c (9) and t_c (3), phases B and C: in the space of s_yh, which phase A is done with.  The workgroup's
    // LDS (static + dynamic) has to stay under 32 KB: config 1's 1250 workgroups are then resident at once, five per CU;
    // 768 bytes more (s_rc on its own, round 3) made it four per CU, a second round of workgroups and 22 us for 17.8
    double (*s_rc)[12] = reinterpret_cast<double (*)[12]>(s_yh);
    static_assert(12 * kMaxCam <= 16 * kMaxCam, "s_rc aliases s_yh");
    __shared__ double sm[16];
    __shared__ int s_view[kBsTile];
    const int t = threadIdx.x;
    PHASE_STAMP(ts0);
    // ---- everything whose address is known is requested now ---------------------------------------------------------
    const int my_q0 = t < nbl ? P.bv_ptr[b0 + t] : 0, my_q1 = t < nbl ? P.bv_ptr[b0 + t + 1] : 0;
    // (phase C's view / board / camera of the first tile: used at the end)
    const int c_slot = min(s0 + t, max(s1 - 1, 0));
    const int c_view = s1 > s0 ? P.slot_view[c_slot] : 0, c_board = s1 > s0 ? P.slot_board[c_slot] : 0, c_cam = s1 > s0 ? P.slot_cam[c_slot] : 0;
    // (phase B's current pose and Jacobi scaling of this lane's board)
    double xb[6], sbv[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) { xb[k] = t < nbl ? S.board_rt[cur][6 * (b0 + t) + k] : 0.0; sbv[k] = t < nbl ? S.s_b[6 * (b0 + t) + k] : 1.0; }
    // (phase B's factor records: the boards' records are contiguous, 7 doubles per thread)
    double facv[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) facv[j] = S.fac[(size_t)kFac * b0 + min(t + kBsThreads * j, nbl * kFac - 1)];
    if constexpr (!WAIT) { for (int i = t; i < P.n_pad; i += kBsThreads) s_yh[i] = S.yhat[i]; }
    if (t < kBsBoards * 6) (&s_q[0][0])[t] = 0.0;
    // the candidate's per-camera records (rotation, left-Jacobian vectors, intrinsics: write_camera_record): one camera
    // per workgroup, by the first lane of the second wave while the loads requested above are in flight -- a serial
    // chain of a few hundred operations that cost workgroup 0 4.6 us when it did all cameras after its board solves
    if constexpr (!WAIT) { if (t == 64) for (int m = blk; m < P.C; m += nblk) write_camera_record(S, cur ^ 1, m); }
    // ---- phase A -----------------------------------------------------------------------------------------------------
    {
        const int grp = t >> 4, a = t & 15;
        const __amdgpu_buffer_rsrc_t r_w = make_rsrc(S.rec[cur], sizeof(double) * (size_t)kRecW * P.V);      // (the W region only: a slot behind s1 - 1 is never addressed)
        constexpr unsigned BAD = 0xffffe000u;
        // column a of the W records of this lane's slot in each of the round's passes, and the slot's camera
        double w[kPasses][6];
        int camv[kPasses];
        auto request = [&](int rbase) {
#pragma unroll
            for (int j = 0; j < kPasses; ++j) {
                const int slot = rbase + kPassSlots * j + grp;
                const unsigned off = (a < kFA && slot < s1) ? 8u * ((unsigned)kRecW * (unsigned)slot + 6u * (unsigned)a) : BAD;
#pragma unroll
                for (int k = 0; k < 3; ++k) { const d2 v = buf_load_2f64(r_w, off, 16u * (unsigned)k); w[j][2 * k] = v[0]; w[j][2 * k + 1] = v[1]; }
                camv[j] = P.slot_cam[min(slot, max(s1 - 1, 0))];
            }
        };
        request(s0);
        if constexpr (WAIT) {
            __shared__ int s_flag;
            if (t == 0) {
                const long long t_start = wall_clock64();
                int f;
                while (((f = __hip_atomic_load(S.y_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 1) < epoch) {
                    __builtin_amdgcn_s_sleep(8);
                    if (wall_clock64() - t_start > kHandoffTimeoutTicks) { f = -1; break; }
                }
                if (f < 0) {             // the solver never reported: a device fault (see k_solve_reduced), not a failed step
                    __hip_atomic_store(&S.ctrl->fault, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&S.ctrl->term_type, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&S.ctrl->done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                s_flag = f;
            }
            __syncthreads();
            // (no acquire fence: it is a `buffer_inv` per wave, 2,500 of them at config 4, served one after the other
            // by the XCDs' L2s -- 20 us.  The few values of the solver that this workgroup reads are read through.)
            if (s_flag < 0 || __hip_atomic_load(&S.ctrl->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
            fail = s_flag & 1;
            for (int i = t; i < P.n_pad; i += kBsThreads) s_yh[i] = handoff_load(&S.yhat[i]);
            // (the candidate's per-camera records are written by the solver workgroup once it has published the step)
        }
        for (int rbase = s0; rbase < s1; rbase += kRoundSlots) {
            const int rend = min(s1, rbase + kRoundSlots);
            __syncthreads();                                            // the previous round is done with s_qv (and s_yh is there)
#pragma unroll
            for (int j = 0; j < kPasses; ++j) {
                const int sl = kPassSlots * j + grp;                    // slot of the round
                const double yh = a < kFA ? s_yh[16 * camv[j] + a] : 0.0;
                double p[6];
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    p[k] = (a < kFA ? w[j][k] : 0.0) * yh;
                    p[k] = row16_allsum(p[k]);
                }
                if (a == 0) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) s_qv[sl][k] = p[k];
                }
            }
            if (rbase + kRoundSlots < s1) request(rbase + kRoundSlots);      // (boards of more than two views: the next round's records while this one's sums are formed)
            __syncthreads();
            // per board, its views in slot order (deterministic)
            if (t < nbl) {
                for (int q = max(my_q0, rbase); q < min(my_q1, rend); ++q)
#pragma unroll
                    for (int k = 0; k < 6; ++k) s_q[t][k] += s_qv[q - rbase][k];
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 7; ++j) s_fac[t + kBsThreads * j] = facv[j];
        __syncthreads();
    }
    PHASE_STAMP(ts1);
    // ---- phase B: lanes 0 .. nbl-1 one board each; lanes 64 .. 64+C-1 the candidate camera rotations -- in the SECOND
    //      wave: as lanes of the first they ran after the board solves (a wave executes both sides of a branch) -----------
    double mb = 0.0, ss = 0.0;
    if (t < nbl) {
        const int b = b0 + t;
        if (my_q1 == my_q0 || fail) {
#pragma unroll
            for (int k = 0; k < 6; ++k) { S.board_rt[cur ^ 1][6 * b + k] = xb[k]; s_new[t][k] = xb[k]; }
        } else {
            const double *f = s_fac + kFac * t;
            double tt[6], y[6], pz[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                double v = f[kFacC + i] * s_q[t][i];
#pragma unroll
                for (int k = 0; k < i; ++k) v -= f[kFacM + i * (i - 1) / 2 + k] * pz[k];
                pz[i] = v;
                tt[i] = f[kFacZ + i] - v;
            }
#pragma unroll
            for (int i = 5; i >= 0; --i) {
                double w = tt[i];
#pragma unroll
                for (int k = i + 1; k < 6; ++k) w -= f[kFacL + k * (k - 1) / 2 + i] * y[k];
                y[i] = w * f[kFacI + i];
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                mb += 0.5 * tt[k] * tt[k] + 0.5 * f[kFacD + k] * y[k] * y[k];
                const double x = xb[k];
                const double xn = x + (-(sbv[k] * y[k]));
                const double d = x - xn;
                ss += d * d;
                S.board_rt[cur ^ 1][6 * b + k] = xn;
                s_new[t][k] = xn;
            }
        }
    } else if (t >= kBsCamLane0 && t < kBsCamLane0 + P.C) {
        const int m = t - kBsCamLane0;
        double crt[3], Rc[9], dRc[27];
        double crt6[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) crt6[k] = WAIT ? handoff_load(&S.cam_rt[cur ^ 1][6 * m + k]) : S.cam_rt[cur ^ 1][6 * m + k];
        for (int k = 0; k < 3; ++k) crt[k] = crt6[k];
        rotation_and_derivatives(crt, Rc, dRc);
        for (int k = 0; k < 9; ++k) s_rc[m][k] = Rc[k];
        for (int k = 0; k < 3; ++k) s_rc[m][9 + k] = crt6[3 + k];

    }
    {
        double red[2] = { mb, ss }, mdummy = 0.0;
        block_reduce256<2>(red, mdummy, sm);       // (contains the barriers that publish s_new / s_rc and retire s_fac)
        if (t == 0) { S.bs_part[2 * blk] = red[0]; S.bs_part[2 * blk + 1] = red[1]; }
    }
    PHASE_STAMP(ts2);
    // ---- phase C: one lane per view of these boards, tiles of kBsTile views ------------------------------------------
    for (int base = s0; base < s1; base += kBsTile) {
        const int slot = base + t;
        int view = -1;
        if (t < kBsTile && slot < s1) {
            const bool first = base == s0;
            view = first ? c_view : P.slot_view[slot];
            const int bl = (first ? c_board : P.slot_board[slot]) - b0, m = first ? c_cam : P.slot_cam[slot];
            double rt[6], bc[kBoardConst];
#pragma unroll
            for (int k = 0; k < 6; ++k) rt[k] = s_new[bl][k];
            board_constants(rt, bc);
            double *o = st_all + (size_t)t * (kVFloatOff + 1);
            view_point_constants(s_rc[m], s_rc[m] + 9, bc, rt + 3, o);
            for (int k = 0; k < 6; ++k) {           // six 3-vectors d -> R_c d
                const double d0 = bc[6 + 3 * k], d1 = bc[6 + 3 * k + 1], d2 = bc[6 + 3 * k + 2];
                for (int r = 0; r < 3; ++r) o[9 + 3 * k + r] = s_rc[m][3 * r] * d0 + s_rc[m][3 * r + 1] * d1 + s_rc[m][3 * r + 2] * d2;
            }
            for (int k = kVConst; k < kVFloatOff; ++k) o[k] = 0.0;
        }
        // the records are indexed by device view (camera-major); a workgroup's slots map to scattered views, so the
        // staging tile is drained one record per 32 consecutive lanes: 256-byte contiguous pieces
        if (t < kBsTile) s_view[t] = view;
        __syncthreads();
        const int nrec = min(kBsTile, s1 - base);
        for (int e = t; e < nrec * kVFloatOff; e += kBsThreads) {
            const int v = e / kVFloatOff, k = e % kVFloatOff;
            S.vconst[(size_t)kVStride * s_view[v] + k] = st_all[(size_t)v * (kVFloatOff + 1) + k];
        }
        if (with_floats) {
            constexpr int kF = kVStride - kVFloatOff;
            for (int e = t; e < nrec * kF; e += kBsThreads) {
                const int v = e / kF, k = e % kF, j = 2 * k;
                const double *sv = st_all + (size_t)v * (kVFloatOff + 1);
                const float f0 = j < kVConst ? (float)sv[j] : 0.f, f1 = j + 1 < kVConst ? (float)sv[j + 1] : 0.f;
                S.vconst[(size_t)kVStride * s_view[v] + kVFloatOff + k] = __hiloint2double(__float_as_int(f1), __float_as_int(f0));
            }
        }
        __syncthreads();
    }
    TL_ONLY(
    if (threadIdx.x == 0 && S.ctrl->iteration == 5 && blk < kKtlGroups) {
        long long *o = g_phs + (size_t)kPhStamps * (kKtlGroups + blk);
        o[0] = ts0; o[1] = ts0; o[2] = ts1; o[3] = ts2; o[4] = wall_clock64(); o[5] = o[4]; o[6] = nbl; o[7] = 0;
    }
    )
    PH_ONLY(
    if (threadIdx.x == 0 && (blk == 0 || blk == 200))
        printf("backsub_prep wg %d: W.yhat %lld  board solve %lld  view constants %lld [10 ns]\n", blk, ts1 - ts0, ts2 - ts1, wall_clock64() - ts2);
    )
}

template <int NTH>
__global__ __launch_bounds__(NTH) void k_backsub_prep(DevProblem P, DevState S, int with_floats)
{
    KTL(5);
    backsub_body<NTH, false>(P, S, with_floats, (int)blockIdx.x, (int)gridDim.x, 0, 0);
}

// ---------------------------------------------------------------------------------------------
// LM control (TrustRegionMinimizer + LevenbergMarquardtStrategy + TrustRegionStepEvaluator),
// one thread.  `init` = IterationZero; otherwise the tail of one loop iteration followed by
// FinalizeIterationAndCheckIfMinimizerCanContinue.
// ---------------------------------------------------------------------------------------------
// H: the (all-reduced) camera tiles, sc: the scalars behind them -- H_stage in global memory, or the LDS copy of the
// workgroup that formed them (k_reduce_control: stage_copy = H_stage, which then receives a copy as well)
// writer = false: the step is taken redundantly (k_schur_gram: every workgroup runs it in its head, on the same inputs,
// to the same bits -- no hand-off, no kernel of its own); only the writer touches global memory.  out: the new state for
// the calling workgroup.
template <bool GUARD>
__device__ void control_step(const DevProblem &P, const DevState &S, int init, const ControlPre &pre, double *sm, const double *H, const double *sc, double *stage_copy,
                             bool writer, CtlOut *out)
{
    // The LM state is read ONCE (wide loads, one memory round trip -- by control_prefetch, at the head of the kernel),
    // advanced in registers and written back once: as individual fields in global memory the ~40 dependent loads and
    // stores of this function cost about half a microsecond each on the single thread that executes it.
    Ctrl &g = *S.ctrl;
    CtrlHead c = pre.c;
    const int t = threadIdx.x;
    if (out && t == 0) { out->cur = c.cur; out->done = c.done; out->radius = c.radius; out->dmin = c.opt.min_lm_diagonal; out->dmax = c.opt.max_lm_diagonal; }
    if (c.done) return;
    const Options &o = c.opt;
    const int tgt = init ? c.cur : (c.cur ^ 1);
    // rank-divergence guard: every rank's decision word (in its slot of the all-reduced scalars) against this rank's own state
    bool disagree = false;
    if (GUARD && TSCM_AB_GUARD && P.world > 1 && t == 0) {
        const double mine = decision_word(c);
        for (int r = 0; r < P.world; ++r) disagree |= sc[kScal + P.world + r] != mine;
    }
    // camera-side norms |x - Plus(x, -g)|_inf, its 2-norm, |x|^2 and the cost, one thread per parameter
    double gmax_c = 0.0, gsq_c = 0.0, xsq_c = 0.0, cost = 0.0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {                       // (one pass up to 16 cameras, two up to kMaxCam)
        const int p = t + 256 * j;
        if (p >= 16 * P.C) break;
        const int m = p >> 4, a = p & 15;
        if (pre.free_param[j]) {
            const double x = pre.x[j];
            const double g = a < kFA ? H[256 * m + a * 16 + kFR] : 0.0;   // b, c: zero gradient
            const double d = x - (x + (-g));
            gmax_c = fmax(gmax_c, fabs(d)); gsq_c += d * d; xsq_c += x * x;
        }
        if (a == 15) cost += 0.5 * H[256 * m + kFR * 16 + kFR];
        if (init && writer && a < 15) {
            const double hii = (a < kFA) ? H[256 * m + a * 16 + a] : 0.0;
            S.s_c[p] = o.jacobi_scaling ? 1.0 / (1.0 + sqrt(hii)) : 1.0;
        }
        if (init && writer && a == 15) S.s_c[p] = 1.0;
    }
    KTLX(4, true);
    { double red[3] = { gsq_c, xsq_c, cost }; block_reduce256<3>(red, gmax_c, sm); gsq_c = red[0]; xsq_c = red[1]; cost = red[2]; }
    KTLX(5, true);
    // publish the staged (all-reduced) camera tiles as the target system's H -- behind the last barrier of this step: a
    // barrier with global stores in flight waits for their acknowledgement
    if (writer) {
#pragma unroll 8
        for (int i = t; i < 256 * P.C; i += 256) { const double h = H[i]; S.H[tgt][i] = h; if (stage_copy) stage_copy[i] = h; }
        if (stage_copy && t < kScal + P.world) stage_copy[256 * P.C + t] = sc[t];
    }
    if (t != 0) return;
    auto commit = [&]() {
        c.fin_count = 0;
        if (writer) static_cast<CtrlHead &>(g) = c;
        if (out) { out->cur = c.cur; out->done = c.done; out->radius = c.radius; }
    };
    if (GUARD && disagree) {
        // some rank holds a different LM state or computed a different camera step: stop here, on every rank in the same step
        // (each of them sees a slot that is not its own word), before another decision is taken on diverged states
        c.done = 1; c.term_type = 2; c.term_reason = kRanksDisagree; c.fault = kFaultRanksDisagree; commit(); return;
    }
    double gmax_b = 0.0;
    for (int r = 0; r < P.world; ++r) gmax_b = fmax(gmax_b, sc[kScal + r]);
    const double gmax_t = fmax(gmax_c, gmax_b);
    if (sc[4] > 0.0) c.lin_fail = 1;          // an e-block factorisation failed on some rank: every rank rejects the step
    const double gnorm_t = sqrt(gsq_c + sc[3]);
    const double xnorm_t = sqrt(xsq_c + sc[2]);
    KTLX(6, true);

    IterLog it;
    it.pad = 0;
    if (init) {
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
            if (++c.num_invalid >= o.max_invalid) { c.done = 1; c.term_type = 2; c.term_reason = kInvalidSteps; commit(); return; }
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
                c.done = 1; c.term_type = 0; c.term_reason = kParamTol; commit(); return;
            }
            if (fabs(it.cost_change) <= o.function_tolerance * c.x_cost) {
                c.done = 1; c.term_type = 0; c.term_reason = kFuncTol; commit(); return;
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
                { const double w = 2.0 * q - 1.0; c.radius = c.radius / fmax(1.0 / 3.0, 1.0 - w * w * w); }
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
    if (writer && c.n_log < kMaxLog) g.log[c.n_log] = it;
    ++c.n_log;
    if (it.iteration >= o.max_num_iterations) { c.done = 1; c.term_type = 1; c.term_reason = kMaxIter; commit(); return; }
    if (it.step_is_successful && it.gradient_max_norm <= o.gradient_tolerance) { c.done = 1; c.term_type = 0; c.term_reason = kGradTol; commit(); return; }
    if (c.radius <= o.min_radius) { c.done = 1; c.term_type = 0; c.term_reason = kMinRadius; commit(); return; }
    commit();
}

__global__ __launch_bounds__(256) void k_control(DevProblem P, DevState S, int init)
{
    __shared__ double sm[256];
    ControlPre pre;
    control_prefetch(P, S, init, pre, S.ctrl);
    control_step<true>(P, S, init, pre, sm, S.H_stage, S.H_stage + 256 * P.C, nullptr);
}

// ---------------------------------------------------------------------------------------------
// first and last launch of a solve.  What the host did with five stream operations in front of a solve (control block
// H2D, counter memset, three D2D copies of the start point, a stream synchronisation) and four synchronous copies
// behind it cost 0.25 ms per solve -- as much as two LM iterations of config 4.
// ---------------------------------------------------------------------------------------------
// the control block as the host set it up (kernel argument), the arrival counter of the fused T reduction at zero and,
// with `reset`, the registered start point in buffer 0
// The first launch of a solve also computes the constants of the initial evaluation (round 5: k_begin_solve + k_view_prep were
// 4.8 + 8.0 us in front of every solve): the grid is k_view_prep's, every thread also moves its share of the start point into buffer 0, and the
// constants are computed from where the start point IS (the registered arrays with `reset`, buffer 0 otherwise) -- nothing in
// this launch reads what another of its workgroups writes.  The control block it installs has cur = 0, done = 0.
__global__ __launch_bounds__(kVPrepThreads) void k_begin_view_prep(DevProblem P, DevState S, CtrlHead head, const double *src_cam, const double *src_intr,
                                                                   const double *src_board, double *bak_cam, double *bak_intr, double *bak_board, int with_floats)
{
    // src_*: where the start point is if not in buffer 0 already (the registered arrays with `reset`; the backup on a re-run);
    // bak_*: where a copy of the start point goes (what a re-run of this solve starts from: a late hand-off, tscm_solver.hip)
    const int i0 = blockIdx.x * kVPrepThreads + threadIdx.x, n = gridDim.x * kVPrepThreads;
    if (i0 == 0) { head.t_begin = wall_clock64(); static_cast<CtrlHead &>(*S.ctrl) = head; }
    if (i0 == 0) { *S.t_count = 0; *S.y_flag = 0; *S.fac_fail = 0; S.ctl_pub->epoch = 0; *S.stats_count = 0; *S.stats_flag = 0; }      // every solve starts with the hand-off counters of the fused launches at zero
    const double *cam = src_cam ? src_cam : S.cam_rt[0], *intr = src_intr ? src_intr : S.intr[0], *board = src_board ? src_board : S.board_rt[0];
    for (int i = i0; i < 6 * P.C; i += n) { const double v = cam[i]; if (src_cam) S.cam_rt[0][i] = v; if (bak_cam) bak_cam[i] = v; }
    for (int i = i0; i < 9 * P.C; i += n) { const double v = intr[i]; if (src_intr) S.intr[0][i] = v; if (bak_intr) bak_intr[i] = v; }
    for (int i = i0; i < 6 * P.B; i += n) { const double v = board[i]; if (src_board) S.board_rt[0][i] = v; if (bak_board) bak_board[i] = v; }
    view_prep_body(P, S, 0, with_floats, cam, intr, board);
}

// the accepted point lives in buffer `cur`: it becomes buffer 0 (what the caller downloads and the next resident solve
// starts from)
__global__ __launch_bounds__(256) void k_end_solve(DevState S, int C, int B)
{
    const int i0 = blockIdx.x * 256 + threadIdx.x, n = gridDim.x * 256;
    if (i0 == 0) S.ctrl->t_end = wall_clock64();
    if (S.ctrl->cur == 0) return;
    for (int i = i0; i < 6 * C; i += n) S.cam_rt[0][i] = S.cam_rt[1][i];
    for (int i = i0; i < 9 * C; i += n) S.intr[0][i] = S.intr[1][i];
    for (int i = i0; i < 6 * B; i += n) S.board_rt[0][i] = S.board_rt[1][i];
}

// Last launch of a one-GPU solve whose last evaluation still waits for its control step (the steps in between were taken in
// k_schur_gram's head): k_control_tail, k_end_solve and the copy of the control block to the host in ONE launch (round 5; they
// were three, 10.4 + 5.0 + 4.1 us by rocprofv3 behind every solve).  Block 0 takes and commits the step, stamps the end of the
// solve and writes the control block's head and the iteration log straight into the host's pinned copy; every other block
// derives the step's OUTCOME itself (control_outcome on the snapshot, exactly like a workgroup of k_schur_gram: same inputs,
// same bits, no hand-off) and moves its slice of the accepted point into buffer 0.
__global__ __launch_bounds__(256) void k_finish_solve(DevProblem P, DevState S, int init, int have_backsub, int C, int B, Ctrl *host_ctrl)
{
    constexpr int kHl = 256 * kMaxCamLds + kScal + 8, kGall = 512 * kMaxCamLds;
    __shared__ double sm[256];
    __shared__ double Hl[kHl];
    __shared__ CtlOut s_ctl;
    const CtrlHead *head = S.ctrl_snap;          // (taken by k_reduce_stats: block 0 rewrites S.ctrl while the others may not have started)
    if (blockIdx.x == 0) {
        __shared__ double Gall[kGall];
        finish_evaluation<false>(P, S, init, have_backsub, /*writer=*/true, Hl, Gall, sm, &s_ctl, head);
        __syncthreads();
        if (threadIdx.x == 0) S.ctrl->t_end = wall_clock64();
        __threadfence();
        __syncthreads();
        // head + the log entries written so far, 8-byte words (the host's copy is pinned, device-visible memory)
        const int n_log = min(max(__hip_atomic_load(&S.ctrl->n_log, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), 0), kMaxLog);
        const int words = (int)((sizeof(CtrlHead) + sizeof(IterLog) * (size_t)n_log) / 8);
        const unsigned long long *src = reinterpret_cast<const unsigned long long *>(S.ctrl);
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(host_ctrl);
        for (int i = threadIdx.x; i < words; i += 256) dst[i] = __hip_atomic_load(&src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    control_outcome(P, S, init, have_backsub, Hl, sm, &s_ctl, head);
    __syncthreads();
    if (__builtin_amdgcn_readfirstlane(s_ctl.cur) == 0) return;
    const int i0 = (blockIdx.x - 1) * 256 + threadIdx.x, n = (gridDim.x - 1) * 256;
    for (int i = i0; i < 6 * C; i += n) S.cam_rt[0][i] = S.cam_rt[1][i];
    for (int i = i0; i < 9 * C; i += n) S.intr[0][i] = S.intr[1][i];
    for (int i = i0; i < 6 * B; i += n) S.board_rt[0][i] = S.board_rt[1][i];
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

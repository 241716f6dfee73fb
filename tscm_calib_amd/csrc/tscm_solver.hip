// tscm_solver.hip -- host side of the MI355X-native TSCM LM solver and the C ABI (include/tscm/tscm.h).
//
// Replaces, behind the reference's own call boundary:
//   * ceres::Problem construction (TS.cpp:249-269, multi_calib.cpp:160-207)  -> tscm_solver_create
//   * ceres::Solve, DENSE_SCHUR + LM (TS.cpp:271-278, multi_calib.cpp:209-216) -> tscm_solver_solve
// The host only enqueues kernels and polls a device-resident control block: accept/reject,
// the trust-region radius and the termination tests all run on the GPU (k_control), so one LM
// iteration costs no host<->device round trip.
#include "tscm/tscm.h"
#include "tscm_kernels.h"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <numeric>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

using namespace tscm;

// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
int tscm_set_error(int code, const std::string &msg) { return fail(code, msg); }   // for tscm_rig.hip

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return fail(TSCM_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_) + " (" __FILE__ ":" + std::to_string(__LINE__) + ")"); \
    } while (0)

#define NCCL_TRY(expr)                                                                                 \
    do {                                                                                               \
        ncclResult_t r_ = (expr);                                                                      \
        if (r_ != ncclSuccess)                                                                         \
            return fail(TSCM_E_RCCL, std::string(#expr) + ": " + ncclGetErrorString(r_));             \
    } while (0)

static int g_experiment[TSCM_EXPERIMENT_COUNT] = { 0 };      // tscm_debug_experiment
static double wall() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// Exchange backends of the frame-sharded solve.  RCCL: one process per GPU (the production path).  LOCAL: all ranks
// live in ONE process on ONE device and share a stream -- the all-reduce is a kernel that sums the ranks' buffers in
// rank order (tscm_comm_create_local / tscm_solver_solve_group): it runs every line of the sharded solver on a
// one-GPU box (RCCL refuses two ranks on one device) and serves hosts that drive several shards from one thread.
struct tscm_local_group {
    int world = 0, device = 0, refs = 0;
    bool dead = false;                  // the members' states diverged (rank-divergence guard): like an IPC / RCCL communicator after a failure
    hipStream_t stream = nullptr;
    double **d_ptrs = nullptr;           // [world] device array of the members' exchange buffers (rewritten per exchange)
};
// IPC: one process per rank like RCCL, but the exchange is this library's own one-shot all-reduce over memory the ranks
// map from each other (hipIpcMemHandle: the same device, or peers over xGMI).  Every rank owns a buffer of
// 2 (parity) x world (source rank) slots of max_doubles doubles and 2 x world arrival flags; an exchange is ONE
// single-workgroup kernel per rank: store my values into my slot of every rank's buffer, release, raise my flag there;
// wait for the world flags of my own buffer; sum the slots in rank order (the bits of the LOCAL backend, on every
// rank).  4 us where an RCCL all-reduce of 16-32 KB takes 15-25 -- and the only back-end that can put several rank
// PROCESSES on one device, which is how the multi-process path runs on a one-GPU box (tools/ipc_check.py,
// `bench.py --gpus N` with fewer devices than ranks).  Exercised between processes on ONE device only (no
// multi-GPU box in reach).  Across devices it is correct by construction since round 5 -- the buffer (slots and flags) is
// fine-grained device memory (hipExtMallocWithFlags: coherent for a peer's system-scope stores and loads), peer access is
// enabled explicitly at connect, and a rank whose buffer could only be had coarse-grained refuses peers on other devices --
// but unmeasured: RCCL stays the default.  A peer that does not arrive within the bound makes the communicator unusable
// (TSCM_E_PEER from then on, like an aborted RCCL communicator).
constexpr int kIpcMaxWorld = 16;
struct IpcPeers { double *base[kIpcMaxWorld]; };
struct tscm_ipc {
    int world = 0, rank = 0;
    size_t max_doubles = 0;             // per slot
    void *own = nullptr;                // this rank's buffer (hipMalloc)
    void *mapped[kIpcMaxWorld] = {};    // the peers' buffers as mapped here (own at [rank])
    bool connected = false;
    long long count = 0;                // exchanges so far: parity and flag value of the next one
    int *d_fault = nullptr;             // raised by a kernel whose peers did not arrive within the bound
    bool fine = false;                  // the buffer is fine-grained memory (a peer on another device may use it)
    bool dead = false;                  // a peer fault or a failed solve: the ranks' exchange counters can no longer be trusted to agree
    size_t slot_doubles() const { return max_doubles; }
    size_t flags_offset() const { return 2 * (size_t)world * max_doubles; }        // in doubles (flags are 8-byte words)
    size_t total_bytes() const { return 8 * (flags_offset() + 2 * (size_t)world); }
};
struct tscm_comm {
    ncclComm_t comm = nullptr;
    tscm_local_group *group = nullptr;   // LOCAL backend
    tscm_ipc *ipc = nullptr;             // IPC backend
    int rank = 0, world = 1, device = 0;
};

// the Gram kernels share one signature; k_eval_gram4 is instantiated per k-step count of a pass (tscm_eval_gram4.h: g4_plan)
typedef void (*EvalKernel)(DevProblem, DevState, int);
static EvalKernel g4_kernel(int ks, bool multi)
{
    static const EvalKernel single[kG4MaxKS] = {
        k_eval_gram4<1, false>, k_eval_gram4<2, false>, k_eval_gram4<3, false>, k_eval_gram4<4, false>, k_eval_gram4<5, false>,
        k_eval_gram4<6, false>, k_eval_gram4<7, false>, k_eval_gram4<8, false>, k_eval_gram4<9, false>, k_eval_gram4<10, false>,
        k_eval_gram4<11, false>, k_eval_gram4<12, false>, k_eval_gram4<13, false>, k_eval_gram4<14, false>, k_eval_gram4<15, false>, k_eval_gram4<16, false> };
    // several passes: ceil(n / passes) >= 29 corners per pass, i.e. at least 8 k-steps
    static const EvalKernel passes[kG4MaxKS] = {
        nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, k_eval_gram4<8, true>, k_eval_gram4<9, true>, k_eval_gram4<10, true>,
        k_eval_gram4<11, true>, k_eval_gram4<12, true>, k_eval_gram4<13, true>, k_eval_gram4<14, true>, k_eval_gram4<15, true>, k_eval_gram4<16, true> };
    return (multi ? passes : single)[ks - 1];
}
// boards of up to 32 corners: M = g4p_views(KS) views share a pass (k_eval_gram4p)
static EvalKernel g4p_kernel(int ks)
{
    static const EvalKernel packed[8] = { k_eval_gram4p<1, 4>, k_eval_gram4p<2, 4>, k_eval_gram4p<3, 4>, k_eval_gram4p<4, 4>,
                                          k_eval_gram4p<5, 3>, k_eval_gram4p<6, 2>, k_eval_gram4p<7, 2>, k_eval_gram4p<8, 2> };
    return ks >= 1 && ks <= 8 ? packed[ks - 1] : nullptr;
}
static EvalKernel f32_kernel(int ks, bool multi)     // the fp32-Jacobian tier on the same pass plan
{
    static const EvalKernel single[kG4MaxKS] = {
        k_eval_gram_f32<1, false>, k_eval_gram_f32<2, false>, k_eval_gram_f32<3, false>, k_eval_gram_f32<4, false>, k_eval_gram_f32<5, false>,
        k_eval_gram_f32<6, false>, k_eval_gram_f32<7, false>, k_eval_gram_f32<8, false>, k_eval_gram_f32<9, false>, k_eval_gram_f32<10, false>,
        k_eval_gram_f32<11, false>, k_eval_gram_f32<12, false>, k_eval_gram_f32<13, false>, k_eval_gram_f32<14, false>, k_eval_gram_f32<15, false>, k_eval_gram_f32<16, false> };
    static const EvalKernel passes[kG4MaxKS] = {
        nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, k_eval_gram_f32<8, true>, k_eval_gram_f32<9, true>, k_eval_gram_f32<10, true>,
        k_eval_gram_f32<11, true>, k_eval_gram_f32<12, true>, k_eval_gram_f32<13, true>, k_eval_gram_f32<14, true>, k_eval_gram_f32<15, true>, k_eval_gram_f32<16, true> };
    return (multi ? passes : single)[ks - 1];
}

struct tscm_solver {
    int device = 0;
    hipStream_t stream = nullptr;
    DevProblem P{};
    DevState S{};
    std::vector<void *> allocs;
    char *arena = nullptr;              // the block dev_alloc is carving pieces from, and how much of it is taken
    size_t arena_used = 0;
    // where tscm_solver_create's wall time went, seconds (tscm_solver_create_timing): [0] runtime_init -- device selection, stream,
    // device properties: the first call of a process pays HIP's initialisation here; [1] host_layout -- view / board orders, chunk
    // tables, Schur work lists; [2] gather -- the observations into device view order on the host; [3] h2d -- device allocations and
    // uploads; [4] kernel_setup -- occupancy queries, function attributes, elimination plans, the operand map's kernel, final sync
    double create_s[5] = { 0, 0, 0, 0, 0 };
    tscm_comm *comm = nullptr;
    // host copies of the layout
    int C = 0, B = 0, V = 0, N = 0, n_points = 0, n_pad = 0;     // B, V, N: this rank's boards / views / corners
    int rank = 0, world = 1;
    int b0 = 0, B_total = 0;            // owned boards = [b0, b0 + B) of the caller's B_total
    long N_total = 0;                   // corners of the whole job (RMSE, summary)
    hipStream_t own_stream = nullptr;   // `stream` is replaced by the group's while a local group solve runs
    bool mono = false;
    std::vector<int> dev2orig;          // device view -> problem view
    std::vector<int> h_view_obs, h_view_count, h_view_cam, h_view_board, h_view_slot;   // h_view_board: DEVICE board index
    std::vector<int> board_perm;        // device board index -> board of the caller (relative to b0): see create
    // caller-owned parameter arrays (host)
    double *h_cam_rt = nullptr, *h_intr = nullptr, *h_board_rt = nullptr;
    // resident initial parameters for the benchmark
    double *d_init_cam = nullptr, *d_init_intr = nullptr, *d_init_board = nullptr;
    double *d_start_cam = nullptr, *d_start_intr = nullptr, *d_start_board = nullptr;    // start point of the solve in progress (a re-run begins there)
    bool have_init = false;
    Ctrl *h_ctrl = nullptr;             // pinned
    Ctrl *d_h_ctrl = nullptr;           // ... and its address on the device (k_finish_solve writes the control block there itself)
    size_t lds_eval = 0, lds_eval32 = 0, lds_solve = 0, lds_gram = 0, lds_bs = 0;
    int bs_threads = 128;               // geometry of k_backsub_prep: 128 threads / 16 boards or 256 / 32
    int nv_chunk0[4] = { 0, 0, 0, 0 }, nv_chunks[4] = { 0, 0, 0, 0 };      // chunk ranges of k_schur_gram<NV>
    bool fuse_reduce = true;            // this solve: k_T_reduce rides in the reduced solve's launch (tscm_options.exec_flags & TSCM_EXEC_SEPARATE_T_REDUCE clears it)
    bool fuse_backsub = true;           // this solve: k_backsub_prep rides in it too (TSCM_EXEC_SEPARATE_BACKSUB clears it)
    bool ctl_in_schur = false;          // this solve: the control step of a candidate's evaluation is taken in the head of the next k_schur_gram
    int schur_resident_ride[4] = { 0, 0, 0, 0 };   // ... of k_schur_gram<NV, true>
    int schur_resident[4] = { 0, 0, 0, 0 };   // workgroups of k_schur_gram<NV> that are resident at once (occupancy x CUs): the first round of its grid
    int schur_cb = kChunkBoards;        // boards per chunk at most: 32 where the Schur grid has several rounds anyway (k_schur_gram<NV, false, 32>: three workgroups per CU)
    int ctl_epoch = 0;                  // control steps taken in k_schur_gram's head in this solve so far
    bool stats_ride = false;            // this solve: the reductions behind a candidate's evaluation are the first workgroups of the next k_schur_gram (k_schur_gram<NV, true>)
    int stats_epoch = 0;                // launches of k_schur_gram<NV, true> in this solve so far (S.ctl_pub->stats_arrived counts their reduction workgroups)
    int eval_pending = 0;               // ... and an evaluation is waiting for it: 1 = reductions complete (one GPU), 2 = all-reduced H_stage (communicator); + 4: the solve's initial evaluation
    int t_epoch = 0;                    // fused launches of this solve so far (the hand-off counter is monotonic)
    int withhold = 0, withhold_next = 0; // this solve / the next one: fault injection (tscm_solver_debug_withhold_handoff)
    int perturb_at = 0, perturb_at_next = 0, perturb_ulps = 0;   // this solve / the next one: LM iteration at which this rank's received T is moved (tscm_solver_debug_perturb_exchange)
    int n_reruns = 0;                   // solves that were run again on separate launches after a late hand-off
    bool no_rerun = false, no_rerun_next = false;      // fault injection: the late hand-off of this / the next solve stays an error
    tscm_comm *comm_reg = nullptr;      // what tscm_solver_set_comm registered; `comm` is what the current solve uses
    int solve_variant = 0;              // 0: k_solve_reduced (up to 4 cameras: one dense block), 1: k_solve_nd (5..8 cameras, or TSCM_EXEC_GRAPH_REDUCED_ORDER), 3: k_solve_reduced_big (more than 8 cameras)
    int dense4_resident = 0;            // workgroups of k_solve_reduced<4, 16, 64, true>'s launch that are resident at once
    // k_solve_nd: [0] the nested-dissection plan of the camera-pair graph, [1] one dense block (TSCM_EXEC_DENSE_REDUCED_ORDER; also what
    // [0] is when the graph is complete); operand map and tables of each on the device; workgroups of the fused launch
    // that are resident at once (occupancy x CUs)
    NdPlan plan[2];
    const int4 *d_nd_map[2] = { nullptr, nullptr };
    const int *d_nd_tab[2] = { nullptr, nullptr }, *d_nd_bs[2] = { nullptr, nullptr };
    size_t lds_nd[2] = { 0, 0 }, lds_dense4 = 0;
    int nd_resident[2] = { 0, 0 };
    int nd = 0;                         // this solve: which of the two
    bool graph_order = false;           // this solve: k_solve_nd also for a rig of up to 4 cameras (TSCM_EXEC_GRAPH_REDUCED_ORDER / _DENSE_REDUCED_ORDER there)
    bool f32_jacobian = false;          // this solve runs k_eval_gram_f32 (tscm_options.jacobian_fp32)
    bool gram16 = false;                // this solve: TSCM_EXEC_GRAM_16X16
    bool solve_tiles = true;            // this solve: k_solve_reduced<.., MF = false> (the default; TSCM_EXEC_MFMA_REDUCED_SOLVE clears it)
    size_t lds_eval4 = 0;               // dynamic LDS of k_eval_gram4
    bool eval4s = false;                // the views of a chunk as one stream of k-steps (k_eval_gram4s): boards whose passes leave lanes idle
    size_t lds_eval4s = 0;
    EvalKernel eval4p = nullptr;        // boards of up to 32 corners: several views per pass (k_eval_gram4p); nullptr otherwise
    size_t lds_eval4p = 0;
    bool one_view_per_pass = false;     // this solve: TSCM_EXEC_ONE_VIEW_PER_PASS
    EvalKernel eval4 = nullptr, eval32 = nullptr;   // ... and its instantiation for this problem's board (g4_kernel), the fp32-Jacobian tier's (f32_kernel)
    // dominant-kernel timing
    int timing = 0;                     // 0 = off, n = bracket every n-th launch of the dominant kernel (and every n-th exchange) with HIP events
    unsigned ev_count[3] = { 0, 0, 0 }; // occurrences so far, by kind: 0 dominant kernel, 1 exchange of T, 2 exchange of H_stage
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    std::vector<int> ev_kind;
    size_t ev_used = 0;
    int t_launches[3] = { 0, 0, 0 };
    double t_ms[3] = { 0.0, 0.0, 0.0 };
};

// Device memory of a solver comes from a few large blocks (round 6: some ninety hipMalloc / hipFree pairs per solver were a
// measurable part of what one call of the drop-in costs: tscm_solver_create_timing).  Every piece starts on a 256-byte boundary
// (the alignment the kernels' 16-byte loads, the 128-byte hand-off lines and hipMalloc itself gave); a piece larger than half a
// block gets an allocation of its own.
constexpr size_t kArenaBlock = (size_t)32 << 20;
template <typename T>
static int dev_alloc(tscm_solver *s, T **p, size_t n)
{
    const size_t bytes = (std::max<size_t>(n, 1) * sizeof(T) + 255) & ~(size_t)255;
    if (bytes > kArenaBlock / 2) {
        void *q = nullptr;
        HIP_TRY(hipMalloc(&q, bytes));
        s->allocs.push_back(q);
        *p = static_cast<T *>(q);
        return 0;
    }
    if (!s->arena || s->arena_used + bytes > kArenaBlock) {
        void *q = nullptr;
        HIP_TRY(hipMalloc(&q, kArenaBlock));
        s->allocs.push_back(q);
        s->arena = static_cast<char *>(q); s->arena_used = 0;
    }
    *p = reinterpret_cast<T *>(s->arena + s->arena_used);
    s->arena_used += bytes;
    return 0;
}

template <typename T>
static int dev_upload(tscm_solver *s, const T **p, const std::vector<T> &h)
{
    T *q = nullptr;
    if (int rc = dev_alloc(s, &q, h.size())) return rc;
    if (!h.empty()) HIP_TRY(hipMemcpy(q, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    *p = q;
    return 0;
}

// ------------------------------------------------------------------------------------------------
extern "C" int tscm_abi_version(void) { return TSCM_ABI_VERSION; }
extern "C" const char *tscm_last_error(void) { return g_err.c_str(); }

extern "C" int tscm_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int tscm_device_synchronize(int device)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return fail(TSCM_E_NO_DEVICE, "no usable HIP device");
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipDeviceSynchronize());
    return 0;
}

extern "C" void tscm_default_options(tscm_options *o, int mono)
{
    o->struct_size = sizeof(tscm_options);
    o->max_num_iterations = mono ? 100 : 50;   // TS.cpp:274 ; Ceres default (multi_calib.cpp:212 is commented out)
    o->function_tolerance = 1e-6;
    o->gradient_tolerance = 1e-10;
    o->parameter_tolerance = 1e-8;
    o->initial_trust_region_radius = 1e4;
    o->max_trust_region_radius = 1e16;
    o->min_trust_region_radius = 1e-32;
    o->min_relative_decrease = 1e-3;
    o->min_lm_diagonal = 1e-6;
    o->max_lm_diagonal = 1e32;
    o->max_num_consecutive_invalid_steps = 5;
    o->jacobi_scaling = 1;
    o->check_every = 4;
    o->jacobian_fp32 = 0;
    o->exec_flags = 0;
}

// stable counting sort of `items` by key(item) in [0, n_keys): the orders tscm_solver_create needs are over small integer keys
// (camera, board, signature class), so they are O(n) passes instead of comparison sorts through lambdas (round 6: the layout of
// config 5's 160,000 views took 8.8 ms of every tscm_solve_multi call)
template <typename K>
static void counting_sort(std::vector<int> &items, int n_keys, K key)
{
    std::vector<int> start((size_t)n_keys + 1, 0), out(items.size());
    for (int x : items) start[(size_t)key(x) + 1]++;
    for (int k = 0; k < n_keys; ++k) start[k + 1] += start[k];
    for (int x : items) out[start[key(x)]++] = x;
    items.swap(out);
}

static int validate(const tscm_problem *p)
{
    if (!p) return fail(TSCM_E_INVALID, "problem is NULL");
    if (p->n_cameras < 1 || p->n_boards < 0 || p->n_points < 1 || p->n_views < 0) return fail(TSCM_E_INVALID, "negative or zero problem dimensions");
    if (p->mono && p->n_cameras != 1) return fail(TSCM_E_INVALID, "mono problem needs exactly one camera");
    if (!p->board_xy || !p->intr || (!p->board_rt && p->n_boards) || (!p->mono && !p->cam_rt)) return fail(TSCM_E_INVALID, "NULL parameter/board array");
    if (p->n_views && (!p->view_camera || !p->view_board || !p->view_offset || !p->view_count || !p->obs_u || !p->obs_v)) return fail(TSCM_E_INVALID, "NULL view/observation array");
    if (p->n_cameras > kMaxCam) return fail(TSCM_E_UNSUPPORTED, "more than 32 cameras");
    for (int v = 0; v < p->n_views; ++v) {
        if (p->view_camera[v] < 0 || p->view_camera[v] >= p->n_cameras) return fail(TSCM_E_INVALID, "view_camera out of range");
        if (p->view_board[v] < 0 || p->view_board[v] >= p->n_boards) return fail(TSCM_E_INVALID, "view_board out of range");
        if (p->view_count[v] < 0 || p->view_count[v] > p->n_points) return fail(TSCM_E_INVALID, "view_count outside [0, n_points]");
        if (p->view_offset[v] < 0) return fail(TSCM_E_INVALID, "negative view_offset");
    }
    return 0;
}

extern "C" void tscm_solver_destroy(tscm_solver *s)
{
    if (!s) return;
    (void)hipSetDevice(s->device);
    s->stream = s->own_stream;
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    for (auto &e : s->ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    for (void *q : s->allocs) (void)hipFree(q);
    if (s->h_ctrl) (void)hipHostFree(s->h_ctrl);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

// owner[b] = rank of board b: contiguous ranges balanced by corner count (shared by tscm_shard_frames and the solver)
static void shard_owner(const tscm_problem *p, int world, std::vector<int> &owner)
{
    std::vector<double> per_board(p->n_boards, 0.0);
    for (int v = 0; v < p->n_views; ++v) per_board[p->view_board[v]] += p->view_count[v];
    double total = 0.0;
    for (double x : per_board) total += x;
    owner.assign(p->n_boards, 0);
    double before = 0.0;
    for (int b = 0; b < p->n_boards; ++b) {
        const int r = total > 0.0 ? (int)(before * world / total) : 0;
        owner[b] = std::min(r, world - 1);
        before += per_board[b];
    }
}

// The device orders of a rank's views and boards (host only: no HIP call), shared by tscm_solver_create_sharded and the CPU
// test's entry point tscm_debug_layout_order: `order` = this rank's views with corners in device order (camera-major, a camera's
// views by device board), board_perm[k] = caller's board (relative to b0) that is device board k, dev_board = its inverse.
static int layout_orders(const tscm_problem *p, int b0, int b1, std::vector<int> &order, std::vector<int> &board_perm, std::vector<int> &dev_board)
{
    const int C = p->n_cameras, B = b1 - b0;
    // ---- device view order: this rank's views with corners, sorted by (camera, board) -----------
    order.clear();
    for (int v = 0; v < p->n_views; ++v) if (p->view_count[v] > 0 && p->view_board[v] >= b0 && p->view_board[v] < b1) order.push_back(v);
    // (camera, board) order: LSD -- stable by board, then stable by camera
    counting_sort(order, b1 - b0, [&](int v) { return p->view_board[v] - b0; });
    counting_sort(order, C, [&](int v) { return p->view_camera[v]; });
    for (size_t i = 1; i < order.size(); ++i)
        if (p->view_camera[order[i]] == p->view_camera[order[i - 1]] && p->view_board[order[i]] == p->view_board[order[i - 1]])
            return fail(TSCM_E_INVALID, "two views with the same (camera, board)");
    const int V = (int)order.size();
    // ---- device board order: boards grouped by camera-set signature (number of views, then the cameras), unseen boards
    // last.  The Schur kernels work on chunks of boards of ONE signature; with this numbering a chunk is a contiguous
    // range of boards AND of record slots, so its kernels derive every address from one small descriptor instead of
    // chasing per-board index tables (each dependent global load costs about a microsecond at the head of a kernel).
    dev_board.assign((size_t)B, -1);            // caller's board (relative to b0) -> device board
    {
        std::vector<int> ptr(B + 1, 0), cams(V);
        for (int i = 0; i < V; ++i) ptr[p->view_board[order[i]] - b0 + 1]++;
        for (int b = 0; b < B; ++b) ptr[b + 1] += ptr[b];
        std::vector<int> fill(B, 0);
        for (int i = 0; i < V; ++i) { const int b = p->view_board[order[i]] - b0; cams[ptr[b] + fill[b]++] = p->view_camera[order[i]]; }   // `order` is camera-major: sorted
        std::vector<int> perm(B);
        std::iota(perm.begin(), perm.end(), 0);
        auto sig_before = [&](int x, int y) {
            const int nx = ptr[x + 1] - ptr[x], ny = ptr[y + 1] - ptr[y];
            if ((nx == 0) != (ny == 0)) return ny == 0;           // boards without views go last
            if (nx != ny) return nx < ny;
            for (int k = 0; k < nx; ++k) if (cams[ptr[x] + k] != cams[ptr[y] + k]) return cams[ptr[x] + k] < cams[ptr[y] + k];
            return false;
        };
        // the distinct signatures are few (camera sets that occur): one representative board each, sorted; every board finds its
        // class through a hash of its camera list, then ONE stable counting pass -- the order a stable sort by signature gives
        {
            auto sig_hash = [&](int b) {
                unsigned long long h = 1469598103934665603ull ^ (unsigned long long)(ptr[b + 1] - ptr[b]);
                for (int k = ptr[b]; k < ptr[b + 1]; ++k) h = (h ^ (unsigned long long)(cams[k] + 1)) * 1099511628211ull;
                return h;
            };
            auto same_sig = [&](int x, int y) { return !sig_before(x, y) && !sig_before(y, x); };
            std::unordered_map<unsigned long long, std::vector<int>> by_hash;     // hash -> representatives (collisions kept apart)
            std::vector<int> reps, rep_of(B);
            for (int b = 0; b < B; ++b) {
                std::vector<int> &cand = by_hash[sig_hash(b)];
                int r = -1;
                for (int c : cand) if (same_sig(reps[c], b)) { r = c; break; }
                if (r < 0) { r = (int)reps.size(); reps.push_back(b); cand.push_back(r); }
                rep_of[b] = r;
            }
            std::vector<int> cls(reps.size());
            std::iota(cls.begin(), cls.end(), 0);
            std::sort(cls.begin(), cls.end(), [&](int x, int y) { return sig_before(reps[x], reps[y]); });
            std::vector<int> rank_of(reps.size());
            for (size_t k = 0; k < cls.size(); ++k) rank_of[cls[k]] = (int)k;
            counting_sort(perm, (int)reps.size(), [&](int b) { return rank_of[rep_of[b]]; });
        }
        board_perm = perm;
        for (int i = 0; i < B; ++i) dev_board[perm[i]] = i;
    }
    // views of one camera sorted by device board
    counting_sort(order, std::max(B, 1), [&](int v) { return dev_board[p->view_board[v] - b0]; });
    counting_sort(order, C, [&](int v) { return p->view_camera[v]; });
    return 0;
}

// Experiment switches of the LAYOUT a solver is created with (round 6: built, measured, not the default -- HISTORY A.7).  Process-wide,
// read by tscm_solver_create; an entry point and not an environment variable read inside the library (the tests and the A/B tools set
// them explicitly), not tscm_options either: nothing a production caller can set makes a solve slower.
extern "C" int tscm_debug_experiment(int which, int value)
{
    if (which < 0 || which >= TSCM_EXPERIMENT_COUNT) return fail(TSCM_E_INVALID, "unknown experiment");
    g_experiment[which] = value;
    return 0;
}

// the pass plan of the Gram kernels for a board of n_points corners (g4_plan, tscm_kernels.h): host code
extern "C" int tscm_debug_gram_plan(int n_points, int *passes, int *corners_per_pass, int *k_steps, int *views_per_pass)
{
    if (n_points < 1) return fail(TSCM_E_INVALID, "n_points must be positive");
    const G4Plan g = g4_plan(n_points);
    if (passes) *passes = g.passes;
    if (corners_per_pass) *corners_per_pass = g.per;
    if (k_steps) *k_steps = g.ks;
    if (views_per_pass) *views_per_pass = g.passes == 1 && g.ks <= 8 ? g4p_views(g.ks) : 1;       // (k_eval_gram4p: boards of up to 32 corners)
    return 0;
}

extern "C" int tscm_debug_layout_order(const tscm_problem *p, int rank, int world, int *n_views_out, int *dev2orig, int *board_perm_out, int *b0_out)
{
    if (int rc = validate(p)) return rc;
    if (world < 1 || rank < 0 || rank >= world || !n_views_out) return fail(TSCM_E_INVALID, "bad arguments");
    std::vector<int> owner;
    shard_owner(p, world, owner);
    int b0 = 0, b1 = 0;
    while (b0 < p->n_boards && owner[b0] < rank) ++b0;
    b1 = b0;
    while (b1 < p->n_boards && owner[b1] == rank) ++b1;
    std::vector<int> order, perm, dev_board;
    if (int rc = layout_orders(p, b0, b1, order, perm, dev_board)) return rc;
    *n_views_out = (int)order.size();
    if (b0_out) *b0_out = b0;
    if (dev2orig) std::copy(order.begin(), order.end(), dev2orig);
    if (board_perm_out) std::copy(perm.begin(), perm.end(), board_perm_out);
    return 0;
}

// the caller's observation arrays (device copy, as they were) -> device view order: one wave per view
__global__ __launch_bounds__(256) void k_gather_obs(const double *raw_u, const double *raw_v, const int *src_off, const int *dst_off, const int *count, int V, double *u, double *v)
{
    const int view = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (view >= V) return;
    const int so = src_off[view], d0 = dst_off[view], n = count[view];
    for (int j = lane; j < n; j += 64) { u[d0 + j] = raw_u[so + j]; v[d0 + j] = raw_v[so + j]; }
}

// Every rank is handed the WHOLE problem description (the view tables are small) and keeps the observations, records
// and pose blocks of the boards it owns.  What the ranks must agree on is derived from the whole problem, identically
// on every rank: which cameras have views at all (free columns of the reduced system), which camera pairs share a
// board (tiles of T) and the total corner count.
extern "C" int tscm_solver_create_sharded(const tscm_problem *p, int device, int rank, int world, tscm_solver **out)
{
    if (!out) return fail(TSCM_E_INVALID, "out is NULL");
    *out = nullptr;
    if (int rc = validate(p)) return rc;
    if (world < 1 || rank < 0 || rank >= world) return fail(TSCM_E_INVALID, "rank / world out of range");
    double t_mark = wall();
    double lap_s[5] = { 0, 0, 0, 0, 0 };
    auto lap = [&](int k) { const double t = wall(); lap_s[k] += t - t_mark; t_mark = t; };      // create_s: see tscm_solver
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(TSCM_E_NO_DEVICE, "no HIP device available (the TSCM solver has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(TSCM_E_NO_DEVICE, "device index out of range");
    HIP_TRY(hipSetDevice(device));

    std::unique_ptr<tscm_solver, void (*)(tscm_solver *)> sp(new tscm_solver, tscm_solver_destroy);
    tscm_solver *s = sp.get();
    s->device = device;
    s->rank = rank; s->world = world;
    // ---- frame ownership ------------------------------------------------------------------------
    std::vector<int> owner;
    shard_owner(p, world, owner);
    int b0 = 0, b1 = 0;
    {
        while (b0 < p->n_boards && owner[b0] < rank) ++b0;
        b1 = b0;
        while (b1 < p->n_boards && owner[b1] == rank) ++b1;
    }
    s->b0 = b0; s->B_total = p->n_boards;
    s->C = p->n_cameras; s->B = b1 - b0; s->n_points = p->n_points; s->mono = p->mono != 0;
    s->n_pad = 16 * s->C;
    s->h_cam_rt = p->cam_rt; s->h_intr = p->intr; s->h_board_rt = p->board_rt;
    HIP_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    s->own_stream = s->stream;
    const int C = s->C, B = s->B;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    lap(0);

    // ---- whole-problem facts (identical on every rank) -------------------------------------------
    std::vector<unsigned char> cam_const(C, 0), cam_active(C, 0);
    for (int m = 0; m < C; ++m) cam_const[m] = (p->mono || (p->cam_pose_constant && p->cam_pose_constant[m])) ? 1 : 0;
    std::vector<unsigned char> pair_present((size_t)C * C, 0);
    long N_total = 0;
    {
        // cameras per board, then every camera pair (mi <= mj) that shares a board
        std::vector<int> ptr(p->n_boards + 1, 0), cams;
        for (int v = 0; v < p->n_views; ++v) if (p->view_count[v] > 0) ptr[p->view_board[v] + 1]++;
        for (int b = 0; b < p->n_boards; ++b) ptr[b + 1] += ptr[b];
        cams.resize(ptr[p->n_boards]);
        std::vector<int> fill(p->n_boards, 0);
        for (int v = 0; v < p->n_views; ++v) {
            if (p->view_count[v] <= 0) continue;
            const int b = p->view_board[v];
            cams[ptr[b] + fill[b]++] = p->view_camera[v];
            cam_active[p->view_camera[v]] = 1;
            N_total += p->view_count[v];
        }
        for (int b = 0; b < p->n_boards; ++b)
            for (int i = ptr[b]; i < ptr[b + 1]; ++i)
                for (int j = ptr[b]; j < ptr[b + 1]; ++j) {
                    const int mi = std::min(cams[i], cams[j]), mj = std::max(cams[i], cams[j]);
                    pair_present[(size_t)mi * C + mj] = 1;
                }
    }
    s->N_total = N_total;

    // ---- device view order and device board order (layout_orders) ----------------------------------
    std::vector<int> order, dev_board;
    if (int rc0 = layout_orders(p, b0, b1, order, s->board_perm, dev_board)) return rc0;
    const int V = (int)order.size();
    s->V = V; s->dev2orig = order;
    std::vector<int> view_cam(V), view_board(V), view_obs(V), view_count(V);
    long N = 0;
    for (int i = 0; i < V; ++i) {
        const int v = order[i];
        view_cam[i] = p->view_camera[v]; view_board[i] = dev_board[p->view_board[v] - b0]; view_count[i] = p->view_count[v];   // device board index
        view_obs[i] = (int)N; N += p->view_count[v];
    }
    if (N > 0x7fffffffL) return fail(TSCM_E_UNSUPPORTED, "more than 2^31 corners");
    // the Gram kernels address observations, per-view constants and records with 32-bit buffer offsets and
    // park the stores of idle lanes at offset 0xffffe000, which must lie beyond the end of every buffer
    if ((unsigned long long)N * sizeof(double) >= 0xffffe000ull || (unsigned long long)V * kRec * sizeof(double) >= 0xffffe000ull)
        return fail(TSCM_E_UNSUPPORTED, "problem too large for 32-bit buffer offsets (more than 3.7 M views or 536 M corners on one GPU)");
    s->N = (int)N;
    s->h_view_obs = view_obs; s->h_view_count = view_count; s->h_view_cam = view_cam; s->h_view_board = view_board;
    lap(1);
    // The observations in device view order.  One GPU: the caller's arrays go to the device AS THEY ARE (one pageable copy each, no
    // host pass over them) and a kernel moves every view's corners into place (k_gather_obs) -- the host gather was 3 ms of a
    // 9 ms create at config 4 and 35 of 62 at config 5.  A rank of a sharded solver keeps only its own boards' views: it gathers
    // those on the host and uploads its share.  (Also the fallback for a caller whose view_offset table leaves the arrays sparse.)
    long raw_lo = 0, raw_hi = 0;
    for (int i = 0; i < V; ++i) {
        const long o = p->view_offset[order[i]];
        if (i == 0 || o < raw_lo) raw_lo = o;
        if (i == 0 || o + view_count[i] > raw_hi) raw_hi = o + view_count[i];
    }
    const bool device_gather = world == 1 && V > 0 && raw_hi - raw_lo <= 2 * N && (unsigned long long)(raw_hi - raw_lo) * sizeof(double) < 0xffffe000ull;
    std::vector<double> u, w;
    if (!device_gather) {
        u.resize((size_t)N); w.resize((size_t)N);
        for (int i = 0; i < V; ++i) {
            const int v = order[i];
            std::memcpy(u.data() + view_obs[i], p->obs_u + p->view_offset[v], sizeof(double) * view_count[i]);
            std::memcpy(w.data() + view_obs[i], p->obs_v + p->view_offset[v], sizeof(double) * view_count[i]);
        }
    }
    lap(2);

    // ---- chunks of views (one wave each), never straddling a camera ----------------------------
    // one round of resident waves: LDS admits floor(160 KiB / lds_eval) single-wave workgroups per CU
    // Jacobian tile geometry: HV rows (multiple of 8 covering min(64, n) corners -- the MFMA loops consume the k-steps
    // of 4 rows in PAIRS, so the tile holds an even number of them; u-rows and v-rows take turns), pitch HV + 2
    // (= 2 * odd: the 16 columns x 2 rows of a 32-lane ds_read_b64 group then hit 32 distinct bank pairs)
    const int half_rows = 8 * ((std::min(64, p->n_points) + 7) / 8);
    const int rp = half_rows + 2;      // = 2 * odd (half_rows is a multiple of 8)
    const size_t lds_eval_bytes = sizeof(double) * (std::max<size_t>((size_t)16 * rp, 512) + kCst + 2 * (size_t)p->n_points);
    // k_eval_gram runs 4 single-chunk waves per workgroup (they share only the final camera-tile sum)
    if (4 * lds_eval_bytes > 64 * 1024) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_eval_gram<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4 * lds_eval_bytes)));
    // the default Gram kernel: k_eval_gram4 instantiated for this board's pass plan (KS k-steps per pass, ceil(n / 56) passes per view)
    const G4Plan g4 = g4_plan(p->n_points);
    const EvalKernel eval4 = g4_kernel(g4.ks, g4.passes > 1);
    size_t lds_eval4 = 4 * sizeof(double) * (size_t)eval_gram4_lds_doubles(p->n_points, g4.ks);
#ifdef TSCM_G4_LDS_PAD      // occupancy experiments (tools/wave_timeline.py): fewer workgroups per CU, same kernel
    lds_eval4 += TSCM_G4_LDS_PAD;
#endif
    if (lds_eval4 > 160 * 1024) return fail(TSCM_E_UNSUPPORTED, "board with too many corners for the Gram kernel's LDS (more than about 2,000)");
    if (lds_eval4 > 64 * 1024) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(eval4), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_eval4));
    int wgs_per_cu = 0;         // resident workgroups per CU (register- and LDS-limited)
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&wgs_per_cu, reinterpret_cast<const void *>(eval4), 256, lds_eval4));
#ifdef TSCM_EVAL_WAVES          // occupancy experiments: chunk tables for this many waves per SIMD
    const int waves_per_cu = 4 * TSCM_EVAL_WAVES;
#else
    const int waves_per_cu = 4 * std::max(1, std::min(4, wgs_per_cu));
#endif
    lap(4);
    const int target_chunks = std::max(64, prop.multiProcessorCount * waves_per_cu - 4 * C);
    const int per_chunk = std::max(1, (V + target_chunks - 1) / target_chunks);
    std::vector<int> chunk_vb, chunk_ve, chunk_cam, cam_chunk_ptr(C + 1, 0);
    {
        int i = 0;
        for (int m = 0; m < C; ++m) {
            cam_chunk_ptr[m] = (int)chunk_vb.size() / 4;     // in workgroups
            int e = i;
            while (e < V && view_cam[e] == m) ++e;
            for (int b0 = i; b0 < e; b0 += per_chunk) { chunk_vb.push_back(b0); chunk_ve.push_back(std::min(e, b0 + per_chunk)); chunk_cam.push_back(m); }
            while (chunk_vb.size() % 4) { chunk_vb.push_back(e); chunk_ve.push_back(e); chunk_cam.push_back(m); }   // empty chunks: whole workgroups per camera
            i = e;
        }
        cam_chunk_ptr[C] = (int)chunk_vb.size() / 4;
    }
    // ---- board -> views (device order => increasing camera) --------------------------------------
    std::vector<int> bv_ptr(B + 1, 0), bv_idx(V);
    for (int i = 0; i < V; ++i) bv_ptr[view_board[i] + 1]++;
    for (int b = 0; b < B; ++b) bv_ptr[b + 1] += bv_ptr[b];
    {
        std::vector<int> fill(B, 0);
        for (int i = 0; i < V; ++i) { const int b = view_board[i]; bv_idx[bv_ptr[b] + fill[b]++] = i; }
    }
    // records are stored board-major: slot q of bv order <-> device view bv_idx[q]
    std::vector<int> view_slot(V), slot_cam(V), slot_view(V), slot_board(V);
    for (int q = 0; q < V; ++q) { view_slot[bv_idx[q]] = q; slot_cam[q] = view_cam[bv_idx[q]]; slot_view[q] = bv_idx[q]; slot_board[q] = view_board[bv_idx[q]]; }
    s->h_view_slot = view_slot;
    // ---- Schur-complement work lists ------------------------------------------------------------
    // camera-pair blocks ("bids") of T: every pair that shares a board on ANY rank, in lexicographic order
    std::vector<int> bid_of(C * C, -1), bid_mi, bid_mj;
    for (int mi = 0; mi < C; ++mi)
        for (int mj = mi; mj < C; ++mj)
            if (pair_present[(size_t)mi * C + mj]) { bid_of[mi * C + mj] = (int)bid_mi.size(); bid_mi.push_back(mi); bid_mj.push_back(mj); }
    auto get_bid = [&](int mi, int mj) { return bid_of[mi * C + mj]; };      // views of a board are sorted by camera: mi <= mj
    // boards grouped by camera-set signature (views of a board are already sorted by camera)
    std::vector<int> order_b;
    for (int b = 0; b < B; ++b) if (bv_ptr[b + 1] > bv_ptr[b]) order_b.push_back(b);
    auto sig_less = [&](int x, int y) {
        const int nx = bv_ptr[x + 1] - bv_ptr[x], ny = bv_ptr[y + 1] - bv_ptr[y];
        if (nx != ny) return nx < ny;
        for (int k = 0; k < nx; ++k) {
            const int cx = slot_cam[bv_ptr[x] + k], cy = slot_cam[bv_ptr[y] + k];
            if (cx != cy) return cx < cy;
        }
        return false;
    };
    // (device boards ARE numbered in signature order, unseen boards last: order_b is sorted as it stands)
    if (!std::is_sorted(order_b.begin(), order_b.end(), sig_less)) std::stable_sort(order_b.begin(), order_b.end(), sig_less);
    // Every partial tile belongs to one camera-pair block; tiles of a block are numbered contiguously
    // (two passes: count, then assign) so that k_T_reduce streams them without indirection.
    struct ChunkT { int begin, end, nv, bid[6]; };
    std::vector<ChunkT> bchunks;
    std::vector<int> sslot, sboard, pair_i, pair_j, pair_board;
    struct PChunk { int begin, end, bid; };
    std::vector<PChunk> pchunks;
    struct FbPair { int q1, q2, board; };
    std::vector<std::vector<FbPair>> fb_pairs;      // fallback pairs per bid (boards with > 3 views)
    {
        size_t fast_boards = 0;
        for (int b : order_b) if (bv_ptr[b + 1] - bv_ptr[b] <= 3) ++fast_boards;
        // chunks of 16 .. kChunkBoards boards (k_schur_gram: 4 waves x groups of 4 boards), about 512 of them on big problems
        // The Schur kernels stream the records and a CU sustains only its share of the memory system, so the CUs must
        // get equal numbers of workgroups: two per CU on big problems (a multiple of the CU count), never more than
        // kChunkBoards boards each, at least 16 (four waves of one group of four).
        const int target_bchunks = 2 * std::max(1, prop.multiProcessorCount);
        lap(1);
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&s->schur_resident[1], reinterpret_cast<const void *>(k_schur_gram<1>), 256, 0));
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&s->schur_resident[2], reinterpret_cast<const void *>(k_schur_gram<2>), 256, 0));
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&s->schur_resident[3], reinterpret_cast<const void *>(k_schur_gram<3>), 256, 0));
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&s->schur_resident_ride[1], reinterpret_cast<const void *>(k_schur_gram<1, true>), 256, 0));
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&s->schur_resident_ride[2], reinterpret_cast<const void *>(k_schur_gram<2, true>), 256, 0));
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&s->schur_resident_ride[3], reinterpret_cast<const void *>(k_schur_gram<3, true>), 256, 0));
        // Round 6, an experiment switch (tscm_debug_experiment(TSCM_EXPERIMENT_SCHUR_CHUNK_32, 1) before create): chunks of 32 boards and k_schur_gram<NV, false, 32>
        // at THREE workgroups per CU (167 registers, no spill) -- the occupancy the round-5 analysis asked for.  Measured at config 5:
        // 57.4 us against 52.3 with 64-board chunks at two per CU (twice the workgroups, each with its head, its control outcome and
        // its four barriers): not the default (HISTORY A.7).
        s->schur_cb = g_experiment[TSCM_EXPERIMENT_SCHUR_CHUNK_32] ? 32 : kChunkBoards;
        if (s->schur_cb == 32) {
            HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&s->schur_resident[1], reinterpret_cast<const void *>(k_schur_gram<1, false, 32>), 256, 0));
            HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&s->schur_resident[2], reinterpret_cast<const void *>(k_schur_gram<2, false, 32>), 256, 0));
        }
        for (int nv = 1; nv <= 3; ++nv) { s->schur_resident[nv] *= prop.multiProcessorCount; s->schur_resident_ride[nv] *= prop.multiProcessorCount; }
        lap(4);
        const int per_bchunk = std::min<int>(s->schur_cb, std::max<int>(16, (int)((fast_boards + target_bchunks - 1) / target_bchunks)));
        size_t i = 0;
        while (i < order_b.size()) {
            size_t e = i + 1;
            while (e < order_b.size() && !sig_less(order_b[i], order_b[e]) && !sig_less(order_b[e], order_b[i])) ++e;
            const int b0 = order_b[i];
            const int nv = bv_ptr[b0 + 1] - bv_ptr[b0];
            if (nv <= 3) {
                for (size_t c0 = i; c0 < e; c0 += per_bchunk) {
                    const size_t c1 = std::min(e, c0 + per_bchunk);
                    ChunkT ch{};
                    ch.begin = (int)sslot.size();
                    for (size_t k = c0; k < c1; ++k) { sslot.push_back(bv_ptr[order_b[k]]); sboard.push_back(order_b[k]); }
                    ch.end = (int)sslot.size();
                    ch.nv = nv;
                    int t = 0;
                    for (int p1 = 0; p1 < nv; ++p1)
                        for (int p2 = p1; p2 < nv; ++p2) ch.bid[t++] = get_bid(slot_cam[bv_ptr[b0] + p1], slot_cam[bv_ptr[b0] + p2]);
                    bchunks.push_back(ch);
                }
            } else {
                for (size_t k = i; k < e; ++k) {
                    const int b = order_b[k];
                    for (int q1 = bv_ptr[b]; q1 < bv_ptr[b + 1]; ++q1)
                        for (int q2 = q1; q2 < bv_ptr[b + 1]; ++q2) {
                            const int bid = get_bid(slot_cam[q1], slot_cam[q2]);
                            if ((int)fb_pairs.size() <= bid) fb_pairs.resize(bid + 1);
                            fb_pairs[bid].push_back({ q1, q2, b });
                        }
                }
            }
            i = e;
        }
        size_t n_fb = 0;
        for (auto &v : fb_pairs) n_fb += v.size();
        const int per_pchunk = std::max<int>(1, (int)((n_fb + 511) / 512));
        for (size_t bid = 0; bid < fb_pairs.size(); ++bid) {
            const int base = (int)pair_i.size();
            for (auto &pr : fb_pairs[bid]) { pair_i.push_back(pr.q1); pair_j.push_back(pr.q2); pair_board.push_back(pr.board); }
            const int end = (int)pair_i.size();
            for (int b0 = base; b0 < end; b0 += per_pchunk) pchunks.push_back({ b0, std::min(end, b0 + per_pchunk), (int)bid });
        }
    }
    const int n_bids = (int)bid_mi.size();
    std::vector<int> bid_part_ptr(n_bids + 1, 0);
    for (auto &ch : bchunks) for (int t = 0; t < ch.nv * (ch.nv + 1) / 2; ++t) bid_part_ptr[ch.bid[t] + 1]++;
    for (auto &pc : pchunks) bid_part_ptr[pc.bid + 1]++;
    for (int b = 0; b < n_bids; ++b) bid_part_ptr[b + 1] += bid_part_ptr[b];
    const int n_tiles = bid_part_ptr[n_bids];
    std::vector<int> next_tile(bid_part_ptr.begin(), bid_part_ptr.end() - 1);
    std::vector<int> bc_begin, bc_end, bc_nv, bc_tile, pc_begin, pc_end, pc_tile;
    for (auto &ch : bchunks) {
        bc_begin.push_back(ch.begin); bc_end.push_back(ch.end); bc_nv.push_back(ch.nv);
        for (int t = 0; t < 6; ++t) bc_tile.push_back(t < ch.nv * (ch.nv + 1) / 2 ? next_tile[ch.bid[t]]++ : -1);
    }
    for (auto &pc : pchunks) { pc_begin.push_back(pc.begin); pc_end.push_back(pc.end); pc_tile.push_back(next_tile[pc.bid]++); }
    const size_t n_pairs = pair_i.size();

    lap(1);
    // ---- upload --------------------------------------------------------------------------------
    DevProblem &P = s->P;
    DevState &S = s->S;
    P.C = C; P.B = B; P.n_points = p->n_points; P.V = V; P.N = (int)N; P.n_pad = s->n_pad;
    P.rank = rank; P.world = world;
    P.rp = rp; P.half = half_rows; P.lds_wave = (int)(lds_eval_bytes / sizeof(double)); P.g4_per = g4.per; P.g4s_ksv = (p->n_points + 3) / 4;
    P.n_chunks = (int)chunk_vb.size(); P.n_pairs = (int)n_pairs; P.n_pchunks = (int)pc_begin.size(); P.n_bids = n_bids;
    P.n_bchunks = (int)bc_begin.size(); P.n_tiles = n_tiles;
    std::vector<double> bxy(p->board_xy, p->board_xy + 2 * (size_t)p->n_points);
    int rc;
    if ((rc = dev_upload(s, &P.board_xy, bxy))) return rc;
    if ((rc = dev_upload(s, &P.view_cam, view_cam))) return rc;
    if ((rc = dev_upload(s, &P.view_board, view_board))) return rc;
    if ((rc = dev_upload(s, &P.view_obs, view_obs))) return rc;
    if ((rc = dev_upload(s, &P.view_count, view_count))) return rc;
    if (!device_gather) {
        if ((rc = dev_upload(s, &P.obs_u, u))) return rc;
        if ((rc = dev_upload(s, &P.obs_v, w))) return rc;
    } else {
        double *du = nullptr, *dv = nullptr, *raw = nullptr;
        int *src = nullptr;
        if ((rc = dev_alloc(s, &du, (size_t)N))) return rc;
        if ((rc = dev_alloc(s, &dv, (size_t)N))) return rc;
        const size_t n_raw = (size_t)(raw_hi - raw_lo);
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&raw), 2 * n_raw * sizeof(double)));          // (temporary: freed below)
        std::unique_ptr<double, void (*)(double *)> raw_guard(raw, [](double *q) { (void)hipFree(q); });
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&src), (size_t)V * sizeof(int)));
        std::unique_ptr<int, void (*)(int *)> src_guard(src, [](int *q) { (void)hipFree(q); });
        std::vector<int> src_off((size_t)V);
        for (int i = 0; i < V; ++i) src_off[i] = (int)(p->view_offset[order[i]] - raw_lo);
        HIP_TRY(hipMemcpy(src, src_off.data(), (size_t)V * sizeof(int), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(raw, p->obs_u + raw_lo, n_raw * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(raw + n_raw, p->obs_v + raw_lo, n_raw * sizeof(double), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_gather_obs, dim3((unsigned)((V + 3) / 4)), dim3(256), 0, 0, raw, raw + n_raw, src, P.view_obs, P.view_count, V, du, dv);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
        P.obs_u = du; P.obs_v = dv;
    }
    if ((rc = dev_upload(s, &P.chunk_vb, chunk_vb))) return rc;
    if ((rc = dev_upload(s, &P.chunk_ve, chunk_ve))) return rc;
    if ((rc = dev_upload(s, &P.chunk_cam, chunk_cam))) return rc;
    {
        // the same per chunk in ONE 16-byte record (+ the observation offset of its first view): the head of k_eval_gram4 is
        // control block + descriptor, then the data
        std::vector<int4> cd(chunk_vb.size());
        for (size_t q = 0; q < cd.size(); ++q) cd[q] = make_int4(chunk_cam[q], chunk_vb[q], chunk_ve[q], chunk_vb[q] < V ? view_obs[chunk_vb[q]] : 0);
        if ((rc = dev_upload(s, &P.chunk_desc, cd))) return rc;
    }
    if ((rc = dev_upload(s, &P.cam_chunk_ptr, cam_chunk_ptr))) return rc;
    for (int q = 0; q <= kMaxCamLds; ++q) P.cam_wg[q] = cam_chunk_ptr[std::min(q, C)];
    if ((rc = dev_upload(s, &P.bv_ptr, bv_ptr))) return rc;
    if ((rc = dev_upload(s, &P.view_slot, view_slot))) return rc;
    if ((rc = dev_upload(s, &P.slot_cam, slot_cam))) return rc;
    if ((rc = dev_upload(s, &P.slot_view, slot_view))) return rc;
    if ((rc = dev_upload(s, &P.slot_board, slot_board))) return rc;
    {
        std::vector<int> slow;
        for (int b : order_b) if (bv_ptr[b + 1] - bv_ptr[b] > 3) slow.push_back(b);
        P.n_slow = (int)slow.size();
        if ((rc = dev_upload(s, &P.slow_boards, slow))) return rc;
        // chunks are in signature order, i.e. sorted by views per board: one launch of k_schur_gram<NV> per NV present
        int max_boards = 1;
        for (size_t c = 0; c < bchunks.size(); ++c) {
            const int nv = bchunks[c].nv;
            if (s->nv_chunks[nv]++ == 0) s->nv_chunk0[nv] = (int)c;
            max_boards = std::max(max_boards, bchunks[c].end - bchunks[c].begin);
        }
        if (max_boards > s->schur_cb) return fail(TSCM_E_UNSUPPORTED, "internal error: board chunk larger than the Schur kernels' chunk size");
        s->lds_gram = 0;
        // groups of 16 boards while they all fit the chip at once (5 workgroups per CU), groups of 32 beyond that
        s->bs_threads = (B + 15) / 16 > 5 * std::max(1, prop.multiProcessorCount) * 3 / 2 ? 256 : 128;
        // ... and groups of 32 (256 threads, the reduced solve's workgroup shape) wherever the back-substitution can ride
        // in the reduced solve's launch (up to 8 cameras: k_solve_nd<.., true>)
        if (C <= kMaxCamLds && n_bids > 0) s->bs_threads = 256;
        s->lds_bs = sizeof(double) * (size_t)(s->bs_threads == 256 ? BsGeom<256>::kLds : BsGeom<128>::kLds);
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_backsub_prep<128>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)s->lds_bs));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_backsub_prep<256>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)s->lds_bs));
    }
    if ((rc = dev_upload(s, &P.pair_i, pair_i))) return rc;
    if ((rc = dev_upload(s, &P.pair_j, pair_j))) return rc;
    if ((rc = dev_upload(s, &P.pc_begin, pc_begin))) return rc;
    if ((rc = dev_upload(s, &P.pc_end, pc_end))) return rc;
    if ((rc = dev_upload(s, &P.pc_tile, pc_tile))) return rc;
    if ((rc = dev_upload(s, &P.bid_part_ptr, bid_part_ptr))) return rc;
    static_assert(kSmallBids >= kMaxCamLds * (kMaxCamLds + 1) / 2, "every camera pair of a rig the register/LDS solver takes");
    for (int b = 0; b <= kSmallBids; ++b) P.bid_part_small[b] = bid_part_ptr[std::min(b, n_bids)];
    if ((rc = dev_upload(s, &P.sslot, sslot))) return rc;
    if ((rc = dev_upload(s, &P.sboard, sboard))) return rc;
    if ((rc = dev_upload(s, &P.pair_board, pair_board))) return rc;
    if ((rc = dev_upload(s, &P.bc_begin, bc_begin))) return rc;
    if ((rc = dev_upload(s, &P.bc_end, bc_end))) return rc;
    if ((rc = dev_upload(s, &P.bc_nv, bc_nv))) return rc;
    if ((rc = dev_upload(s, &P.bc_tile, bc_tile))) return rc;
    {
        // device boards are numbered in signature order, so entry k of the sorted board list IS board k
        for (size_t k = 0; k < sboard.size(); ++k) if (sboard[k] != (int)k) return fail(TSCM_E_UNSUPPORTED, "internal error: device board order is not the signature order");
        std::vector<int4> desc(bchunks.size());
        for (size_t c = 0; c < bchunks.size(); ++c) desc[c] = make_int4(bchunks[c].begin, bchunks[c].end, bv_ptr[bchunks[c].begin], bchunks[c].nv);
        if ((rc = dev_upload(s, &P.bc_desc, desc))) return rc;
    }
    if ((rc = dev_upload(s, &P.bid_mi, bid_mi))) return rc;
    if ((rc = dev_upload(s, &P.bid_mj, bid_mj))) return rc;
    {
        std::vector<short> lut(bid_of.begin(), bid_of.end());
        if ((rc = dev_upload(s, &P.bid_lut, lut))) return rc;
    }
    std::vector<unsigned char> col_active_host;
    if ((rc = dev_upload(s, &P.cam_const, cam_const))) return rc;
    if ((rc = dev_upload(s, &P.cam_active, cam_active))) return rc;
    {
        std::vector<unsigned char> board_const((size_t)B, 0);
        if (p->board_pose_constant) for (int i = 0; i < B; ++i) board_const[i] = p->board_pose_constant[b0 + s->board_perm[i]] ? 1 : 0;
        if ((rc = dev_upload(s, &P.board_const, board_const))) return rc;
    }
    {
        std::vector<unsigned char> &col_active = col_active_host;
        col_active.assign((size_t)s->n_pad, 0);
        for (int i = 0; i < s->n_pad; ++i) { const int m = i >> 4, a = i & 15; col_active[i] = (a < kFA && cam_active[m] && !(a < 6 && cam_const[m])) ? 1 : 0; }
        if ((rc = dev_upload(s, &P.col_active, col_active))) return rc;
        // compact numbering of the free camera-side columns: the reduced system is factored without the
        // identity rows of constant / padding columns
        std::vector<int> act_map((size_t)s->n_pad, -1);
        int n_act = 0;
        for (int i = 0; i < s->n_pad; ++i) if (col_active[i]) act_map[n_act++] = i;
        P.n_act = n_act;
        // closed form of the same map for k_solve_reduced (kernel arguments, see DevProblem)
        for (int q = 0; q < 9; ++q) P.cam_pre[q] = n_act;
        for (int q = 0; q < 8; ++q) P.cam_col0[q] = 0;
        P.pair_mask = 0;
        if (C <= kMaxCamLds) {
            for (int mi = 0; mi < C; ++mi) for (int mj = mi; mj < C; ++mj) if (bid_of[mi * C + mj] >= 0) P.pair_mask |= 1ull << (mi * 8 + mj);
            int run = 0;
            for (int m = 0; m < C; ++m) {
                P.cam_pre[m] = run;
                P.cam_col0[m] = 16 * m + (cam_const[m] ? 6 : 0);
                if (cam_active[m]) run += cam_const[m] ? kFA - 6 : kFA;
            }
        }
        if ((rc = dev_upload(s, &P.act_map, act_map))) return rc;
    }

    for (int k = 0; k < 2; ++k) {
        if ((rc = dev_alloc(s, &S.cam_rt[k], 6 * (size_t)C))) return rc;
        if ((rc = dev_alloc(s, &S.intr[k], 9 * (size_t)C))) return rc;
        if ((rc = dev_alloc(s, &S.board_rt[k], 6 * (size_t)B))) return rc;
        if ((rc = dev_alloc(s, &S.rec[k], (size_t)kRec * V))) return rc;
        if ((rc = dev_alloc(s, &S.H[k], 256 * (size_t)C))) return rc;
    }
    if ((rc = dev_alloc(s, &s->d_init_cam, 6 * (size_t)C))) return rc;
    if ((rc = dev_alloc(s, &s->d_init_intr, 9 * (size_t)C))) return rc;
    if ((rc = dev_alloc(s, &s->d_init_board, 6 * (size_t)B))) return rc;
    if ((rc = dev_alloc(s, &s->d_start_cam, 6 * (size_t)C))) return rc;
    if ((rc = dev_alloc(s, &s->d_start_intr, 9 * (size_t)C))) return rc;
    if ((rc = dev_alloc(s, &s->d_start_board, 6 * (size_t)B))) return rc;
    if ((rc = dev_alloc(s, &S.board_pc, (size_t)kBoardConst * B))) return rc;
    if ((rc = dev_alloc(s, &S.cam_pc, (size_t)kCamConst * C))) return rc;
    if ((rc = dev_alloc(s, &S.vconst, (size_t)kVStride * V))) return rc;
    for (int k = 0; k < 2; ++k) if ((rc = dev_alloc(s, &S.cconst[k], (size_t)kCStride * C))) return rc;
    if ((rc = dev_alloc(s, &S.campart, 512 * (size_t)(P.n_chunks / 4)))) return rc;
    if ((rc = dev_alloc(s, &S.campart2, 512 * (size_t)C))) return rc;
    if ((rc = dev_alloc(s, &S.H_stage, 256 * (size_t)C + kScal + 2 * (size_t)world))) return rc;
    if ((rc = dev_alloc(s, &S.s_b, 6 * (size_t)B))) return rc;
    if ((rc = dev_alloc(s, &S.s_c, (size_t)s->n_pad))) return rc;
    if ((rc = dev_alloc(s, &S.fac, (size_t)kFac * B))) return rc;
    if ((rc = dev_alloc(s, &S.pairpart, 256 * (size_t)P.n_tiles))) return rc;
    if ((rc = dev_alloc(s, &S.T, 256 * (size_t)n_bids))) return rc;
    if ((rc = dev_alloc(s, &S.t_count, 1))) return rc;
    HIP_TRY(hipMemset(S.t_count, 0, sizeof(int)));
    if ((rc = dev_alloc(s, &S.y_flag, 1))) return rc;
    HIP_TRY(hipMemset(S.y_flag, 0, sizeof(int)));
    if ((rc = dev_alloc(s, &S.fac_fail, 1))) return rc;
    HIP_TRY(hipMemset(S.fac_fail, 0, sizeof(int)));
    if ((rc = dev_alloc(s, &S.yhat, (size_t)s->n_pad))) return rc;
    S.n_bs_blocks = (B + 15) / 16;                  // (upper bound for the allocation; set to the geometry's group count below)
    S.n_st_blocks = (B + 255) / 256;
    if ((rc = dev_alloc(s, &S.bs_part, 2 * (size_t)S.n_bs_blocks))) return rc;
    S.n_bs_blocks = (B + s->bs_threads / 8 - 1) / (s->bs_threads / 8);        // groups of k_backsub_prep's geometry
    if ((rc = dev_alloc(s, &S.st_part, kStStride * (size_t)S.n_st_blocks))) return rc;
    if ((rc = dev_alloc(s, &S.ctrl, 1))) return rc;
    if ((rc = dev_alloc(s, &S.ctrl_snap, 1))) return rc;
    if ((rc = dev_alloc(s, &S.ctl_pub, 1))) return rc;
    if ((rc = dev_alloc(s, &S.stats_count, 64))) return rc;      // (256 bytes each: lines of their own)
    if ((rc = dev_alloc(s, &S.stats_flag, 64))) return rc;
    HIP_TRY(hipMemset(S.stats_count, 0, 256)); HIP_TRY(hipMemset(S.stats_flag, 0, 256));
    HIP_TRY(hipMemset(S.ctl_pub, 0, sizeof(CtlPub)));
    HIP_TRY(hipMemset(S.T, 0, sizeof(double) * 256 * (size_t)n_bids));
    HIP_TRY(hipMemset(S.H_stage, 0, sizeof(double) * (256 * (size_t)C + kScal + 2 * (size_t)world)));
    HIP_TRY(hipMemset(S.campart2, 0, sizeof(double) * 512 * (size_t)C));
    HIP_TRY(hipMemset(S.ctrl, 0, sizeof(Ctrl)));
    HIP_TRY(hipMemset(S.ctrl_snap, 0, sizeof(CtrlHead)));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&s->h_ctrl), sizeof(Ctrl)));
    HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&s->d_h_ctrl), s->h_ctrl, 0));

    lap(3);
    s->lds_eval = 4 * lds_eval_bytes;
    s->lds_eval32 = sizeof(double) * (size_t)eval_f32_lds_doubles(p->n_points, g4.ks);
    s->lds_eval4 = lds_eval4; s->eval4 = eval4; s->eval32 = f32_kernel(g4.ks, g4.passes > 1);
    {
        // the stream kernel (k_eval_gram4s, tscm_eval_gram4s.h): an experiment of round 6, tscm_debug_experiment(TSCM_EXPERIMENT_GRAM_STREAM, 1) only
        const int ksv = (p->n_points + 3) / 4;
        const double fill = (double)p->n_points / (64.0 * g4.passes);
        (void)fill;
        s->eval4s = ksv >= 9 && g_experiment[TSCM_EXPERIMENT_GRAM_STREAM] != 0;          // (measured slower on every board: opt-in only)
        s->lds_eval4s = sizeof(double) * (size_t)eval_gram4s_lds_doubles(p->n_points);
        if (s->lds_eval4s > 160 * 1024) s->eval4s = false;
        if (s->eval4s && s->lds_eval4s > 64 * 1024) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_eval_gram4s), hipFuncAttributeMaxDynamicSharedMemorySize, (int)s->lds_eval4s));
    }
    if (g4.passes == 1 && g4.ks <= 8) {
        s->eval4p = g4p_kernel(g4.ks);
        s->lds_eval4p = 4 * sizeof(double) * (size_t)eval_gram4p_lds_doubles(p->n_points, g4.ks, g4p_views(g4.ks));
    }
    // reduced solve: up to 8 cameras k_solve_nd on the plan of the camera-pair graph (tscm_nd_plan.h), larger rigs in global memory
    s->solve_variant = C <= 4 ? 0 : C <= kMaxCamLds ? 1 : 3;
    if (s->solve_variant == 0) {
        // where every thread of k_solve_reduced finds its operands (camera / pair structure only): written once
        int4 *map = nullptr;
        if ((rc = dev_alloc(s, &map, (size_t)(kSolveMapSlots / 4) * 256))) return rc;
        hipLaunchKernelGGL((k_solve_map<4, 16>), dim3(1), dim3(256), 0, 0, P, map);
        HIP_TRY(hipGetLastError());
        P.solve_map = map;
        const size_t NN = 64, TT = 4, NPD = 64;
        s->lds_dense4 = sizeof(double) * (NN * (NN + 2) + 2 * (NN / TT) * (TT * TT + 2) + 2 * NN + 3 * NPD);
        int per_cu = 0;
        // (the riders share the launch with either solver workgroup: the smaller of the two occupancies)
        int per_cu_tiles = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(k_solve_reduced<4, 16, 64, true, true>), 256, std::max(s->lds_dense4, s->lds_bs)));
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_tiles, reinterpret_cast<const void *>(k_solve_reduced<4, 16, 64, true, false>), 256, std::max(s->lds_dense4, s->lds_bs)));
        per_cu = std::min(per_cu, per_cu_tiles);
        s->dense4_resident = per_cu * prop.multiProcessorCount;
    }
    if (s->solve_variant <= 1) {
        int ncols[kMaxCamLds], col0[kMaxCamLds];
        for (int m = 0; m < C; ++m) { ncols[m] = cam_active[m] ? (cam_const[m] ? kFA - 6 : kFA) : 0; col0[m] = 16 * m + (cam_const[m] ? 6 : 0); }
        // two plans: [0] along the camera-pair graph, [1] the whole system as one dense block.  A graph whose per-camera panel
        // padding does not fit the tile budget (dense but incomplete pair graphs of 8 free cameras) is solved on the dense plan;
        // only a system that fits neither is refused
        if (!nd_build_plans(C, ncols, col0, pair_present.data(), bid_of.data(), s->plan))
            return fail(TSCM_E_UNSUPPORTED, "internal error: the reduced system does not fit the register/LDS solver");
        for (int v = 0; v < 2; ++v) {
            const NdPlan &pl = s->plan[v];
            // (host replica check: the plan's columns are exactly the free columns)
            std::vector<int> cols;
            for (int pc : pl.pcol) if (pc >= 0) cols.push_back(pc);
            std::sort(cols.begin(), cols.end());
            std::vector<int> want;
            for (int i = 0; i < s->n_pad; ++i) if (col_active_host[i]) want.push_back(i);
            if (cols != want) return fail(TSCM_E_UNSUPPORTED, "internal error: elimination plan does not cover the free columns");
            const int4 *map = nullptr;
            {
                std::vector<int4> m4(pl.map.size() / 4);
                std::memcpy(m4.data(), pl.map.data(), pl.map.size() * sizeof(int));
                if ((rc = dev_upload(s, &map, m4))) return rc;
            }
            s->d_nd_map[v] = map;
            if ((rc = dev_upload(s, &s->d_nd_tab[v], pl.tab))) return rc;
            if ((rc = dev_upload(s, &s->d_nd_bs[v], pl.bs_tab))) return rc;
            s->lds_nd[v] = sizeof(double) * pl.lds_doubles;
        }
        {
            // ONE dynamic-LDS bound for the four instantiations (either plan may be launched, with or without riders), set
            // before the occupancy queries that depend on it
            const size_t lds_max = std::max(std::max(s->lds_nd[0], s->lds_nd[1]), s->lds_bs);
            if (lds_max > 64 * 1024) {
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_solve_nd<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_solve_nd<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_solve_nd<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_solve_nd<2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
            }
        }
        for (int v = 0; v < 2; ++v) {
            // workgroups of the fused launch that are resident at once: the back-substitution workgroups that ride in it WAIT
            // for the solver workgroup, so only as many are put there as fit the chip next to it (and the T producers)
            const size_t lds = std::max(s->lds_nd[v], s->lds_bs);
            int per_cu = 0;
            if (s->plan[v].tpt == 1) HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(k_solve_nd<1, true>), kNdThreads, lds));
            else HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(k_solve_nd<2, true>), kNdThreads, lds));
            s->nd_resident[v] = per_cu * prop.multiProcessorCount;
        }
        s->lds_solve = std::max(s->lds_nd[0], s->lds_nd[1]);
    }
    if (s->solve_variant == 3) {
        // rigs of 9..32 cameras: the compact system (+ rhs row) lives in global memory
        const int NN = (P.n_act + 15) & ~15;
        s->lds_solve = solve_big_lds_bytes(NN, s->n_pad);
        if ((rc = dev_alloc(s, &S.Abig, (size_t)256 * (NN / 16 + 1) * (NN / 16 + 2) / 2))) return rc;      // packed lower triangle of 16x16 blocks, incl. the rhs block row
        if (s->lds_solve > 64 * 1024) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_solve_reduced_big), hipFuncAttributeMaxDynamicSharedMemorySize, (int)s->lds_solve));
    }
    if (s->lds_eval > 160 * 1024) return fail(TSCM_E_UNSUPPORTED, "board has too many corners for the LDS board-point tile");
    HIP_TRY(hipDeviceSynchronize());
    lap(4);
    for (int k = 0; k < 5; ++k) s->create_s[k] = lap_s[k];
    *out = sp.release();
    return 0;
}

// where the wall time of this solver's tscm_solver_create went: out[0..4] = runtime_init, host_layout, gather, h2d, kernel_setup (seconds)
extern "C" int tscm_solver_create_timing(const tscm_solver *s, double out[5])
{
    if (!s || !out) return fail(TSCM_E_INVALID, "NULL argument");
    for (int k = 0; k < 5; ++k) out[k] = s->create_s[k];
    return 0;
}

extern "C" int tscm_solver_create(const tscm_problem *p, int device, tscm_solver **out)
{
    return tscm_solver_create_sharded(p, device, 0, 1, out);
}

static tscm_comm *effective_comm(const tscm_solver *s, int exec_flags)
{
    tscm_comm *c = s->comm_reg;
    return (c && (c->world > 1 || c->group || (exec_flags & TSCM_EXEC_KEEP_SINGLE_RANK_COMM))) ? c : nullptr;
}

// Fault injection for the tests of the device-side hand-off: in the NEXT solve of this solver one producer of the fused
// hand-off never reports in, and the solve must end with TSCM_E_HIP within the hand-off's time bound.  Not an option of
// a solve (tscm_options carries nothing that can make a production solve fail).
extern "C" int tscm_solver_debug_withhold_handoff(tscm_solver *s, int on)
{
    if (!s) return fail(TSCM_E_INVALID, "solver is NULL");
    s->withhold_next = on == 3 ? 2 : on ? 1 : 0;      // (3: a reduction block riding in the Schur-complement launch, not a producer of the tiles)
    s->no_rerun_next = on == 2;          // 2: ... and the solve is NOT run again on separate launches (the error path itself)
    return 0;
}

// Fault injection for the rank-divergence guard: in the NEXT solve of this solver (a rank of a communicator), the copy of the
// Schur-complement tiles T this rank RECEIVES from the all-reduce of LM iteration `iteration` (1-based) is moved by `ulps` units
// in the last place in the (fx, fx) entry of its first tile (a free column of every rig; if that is zero, the first non-zero
// entry) -- what a collective that does not hand every rank the same bits would do.
extern "C" int tscm_solver_debug_perturb_exchange(tscm_solver *s, int iteration, int ulps)
{
    if (!s) return fail(TSCM_E_INVALID, "solver is NULL");
    if (iteration < 0 || ulps < 0) return fail(TSCM_E_INVALID, "iteration and ulps must not be negative");
    s->perturb_at_next = iteration; s->perturb_ulps = ulps;
    return 0;
}
__global__ void k_debug_perturb(double *buf, size_t n, int ulps)
{
    const size_t fxfx = 6 * 16 + 6;          // F index 6 = fx (tscm_kernels.h)
    if (fxfx < n && buf[fxfx] != 0.0) { buf[fxfx] = __longlong_as_double(__double_as_longlong(buf[fxfx]) + ulps); return; }
    for (size_t i = 0; i < n; ++i)
        if (buf[i] != 0.0) { buf[i] = __longlong_as_double(__double_as_longlong(buf[i]) + ulps); return; }
}

extern "C" int tscm_solver_reruns(const tscm_solver *s)
{
    if (!s) return fail(TSCM_E_INVALID, "solver is NULL");
    return s->n_reruns;
}

extern "C" int tscm_solver_set_comm(tscm_solver *s, tscm_comm *comm)
{
    if (!s) return fail(TSCM_E_INVALID, "solver is NULL");
    if (comm && comm->device != s->device) return fail(TSCM_E_INVALID, "communicator and solver live on different devices");
    if (comm && (comm->world != s->world || comm->rank != s->rank))
        return fail(TSCM_E_INVALID, "communicator rank / world differ from the solver's shard (tscm_solver_create_sharded)");
    // a single-rank RCCL communicator is a no-op unless a solve asks for its code path (separate k_control,
    // stream-ordered all-reduces) with TSCM_EXEC_KEEP_SINGLE_RANK_COMM: effective_comm()
    s->comm_reg = comm;
    s->comm = effective_comm(s, 0);
    return 0;
}

extern "C" int tscm_solver_upload_params(tscm_solver *s, const double *cam_rt, const double *intr, const double *board_rt)
{
    if (!s || !intr || (!board_rt && s->B_total)) return fail(TSCM_E_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(s->device));
    std::vector<double> zero(6 * (size_t)s->C, 0.0);
    const double *c = (s->mono || !cam_rt) ? zero.data() : cam_rt;
    HIP_TRY(hipMemcpyAsync(s->d_init_cam, c, sizeof(double) * 6 * s->C, hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipMemcpyAsync(s->d_init_intr, intr, sizeof(double) * 9 * s->C, hipMemcpyHostToDevice, s->stream));
    // board_rt is the caller's full-length array; this rank keeps the poses of the boards it owns, in device board order
    std::vector<double> brd(6 * (size_t)s->B);
    for (int i = 0; i < s->B; ++i) std::memcpy(brd.data() + 6 * (size_t)i, board_rt + 6 * ((size_t)s->b0 + s->board_perm[i]), 6 * sizeof(double));
    if (s->B) HIP_TRY(hipMemcpyAsync(s->d_init_board, brd.data(), sizeof(double) * 6 * s->B, hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    s->have_init = true;
    return 0;
}

// one launch of the dominant kernel, optionally bracketed by HIP events on the solver's stream
// HIP-event brackets on the solver's stream: every `timing`-th occurrence of a kind (0 = the dominant kernel, 1 = the
// exchange of T, 2 = the exchange of H_stage) is timed; an event pair holds the stream for a few microseconds, so the
// sampling keeps the measurement out of the result
static int timed_pair(tscm_solver *s, int kind, hipEvent_t *e0, hipEvent_t *e1)
{
    *e0 = *e1 = nullptr;
    if (!(s->timing > 0 && (s->ev_count[kind]++ % (unsigned)s->timing) == 0)) return 0;
    if (s->ev_used == s->ev.size()) {
        hipEvent_t a, b;
        HIP_TRY(hipEventCreate(&a)); HIP_TRY(hipEventCreate(&b));
        s->ev.emplace_back(a, b);
        s->ev_kind.push_back(0);
    }
    s->ev_kind[s->ev_used] = kind;
    *e0 = s->ev[s->ev_used].first;
    *e1 = s->ev[s->ev_used].second;
    ++s->ev_used;
    return 0;
}

static int timed_begin(tscm_solver *s, int kind, hipStream_t stream, hipEvent_t *e1)
{
    hipEvent_t e0 = nullptr;
    if (int rc = timed_pair(s, kind, &e0, e1)) return rc;
    if (e0) HIP_TRY(hipEventRecord(e0, stream));
    return 0;
}

// one launch of the dominant kernel.  A timed launch carries its event pair IN the dispatch (hipExtLaunchKernelGGL:
// the events take the start and end time stamps of this kernel's packet on the solver's stream) -- two hipEventRecord
// around it are two more packets with a drain each, 8.5 us per timed launch at config 4 and 3 % of the driver's
// 20-step run
static void launch_eval_kernel(EvalKernel kernel, dim3 grid, size_t lds, tscm_solver *s, hipEvent_t e0, hipEvent_t e1, int cand)
{
    if (e0) hipExtLaunchKernelGGL(kernel, grid, dim3(256), (std::uint32_t)lds, s->stream, e0, e1, 0, s->P, s->S, cand);
    else hipLaunchKernelGGL(kernel, grid, dim3(256), lds, s->stream, s->P, s->S, cand);
}

static int launch_eval(tscm_solver *s, int cand)
{
    const DevProblem &P = s->P;
    if (P.n_chunks == 0) return 0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (int rc = timed_pair(s, 0, &e0, &e1)) return rc;
    const dim3 grid(P.n_chunks / 4);
    // 9x6 .. 7x8 boards (53..56 corners per pass) get the variant with a compile-time LDS pitch
    if (s->f32_jacobian) launch_eval_kernel(s->eval32, grid, s->lds_eval32, s, e0, e1, cand);
    else if (!s->gram16 && s->eval4p && !s->one_view_per_pass) launch_eval_kernel(s->eval4p, grid, s->lds_eval4p, s, e0, e1, cand);     // small boards: views share a pass
    else if (!s->gram16 && s->eval4s && !s->one_view_per_pass) launch_eval_kernel(k_eval_gram4s, grid, s->lds_eval4s, s, e0, e1, cand);      // the chunk's views as one stream of k-steps
    else if (!s->gram16) launch_eval_kernel(s->eval4, grid, s->lds_eval4, s, e0, e1, cand);       // every board size (round 6)
    else if (P.rp == 58) launch_eval_kernel(k_eval_gram<58>, grid, s->lds_eval, s, e0, e1, cand);
    else launch_eval_kernel(k_eval_gram<0>, grid, s->lds_eval, s, e0, e1, cand);
    return 0;
}

static int collect_timing(tscm_solver *s)
{
    for (size_t i = 0; i < s->ev_used; ++i) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, s->ev[i].first, s->ev[i].second));
        const int k = s->ev_kind[i];
        s->t_ms[k] += ms; s->t_launches[k] += 1;
    }
    s->ev_used = 0;
    return 0;
}

extern "C" int tscm_solver_kernel_time(tscm_solver *s, int enable, int *launches, double *total_ms)
{
    if (!s) return fail(TSCM_E_INVALID, "solver is NULL");
    if (launches) *launches = s->t_launches[0];
    if (total_ms) *total_ms = s->t_ms[0];
    s->t_launches[0] = 0; s->t_ms[0] = 0.0;
    s->timing = enable < 0 ? 0 : enable;
    s->ev_count[0] = s->ev_count[1] = s->ev_count[2] = 0;
    return 0;
}

extern "C" int tscm_solver_exchange_time(tscm_solver *s, int *n_T, double *ms_T, int *n_H, double *ms_H)
{
    if (!s) return fail(TSCM_E_INVALID, "solver is NULL");
    if (n_T) *n_T = s->t_launches[1];
    if (ms_T) *ms_T = s->t_ms[1];
    if (n_H) *n_H = s->t_launches[2];
    if (ms_H) *ms_H = s->t_ms[2];
    s->t_launches[1] = s->t_launches[2] = 0; s->t_ms[1] = s->t_ms[2] = 0.0;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// The LM loop over a set of shards.  `members` is ONE solver (single GPU, or one RCCL rank: the peers run the same
// loop in their own processes) or all ranks of a LOCAL group (one process, one device, one stream, lock step).
// Per iteration the ranks exchange exactly two buffers, each with a sum all-reduce:
//   T        n_bids * 256 doubles   the Schur complement tiles, after k_T_reduce
//   H_stage  256 C + kScal + 2 world  camera tiles, cost, model-cost / norm partials, failure flag, per-rank max slots, per-rank
//                                     decision words (rank-divergence guard: decision_word, tscm_kernels.h)
// ------------------------------------------------------------------------------------------------
__global__ void k_xchg_sum(double *const *bufs, int world, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double a = 0.0;
    for (int r = 0; r < world; ++r) a += bufs[r][i];       // rank order: every member receives the same bits
    for (int r = 0; r < world; ++r) bufs[r][i] = a;
}

// the IPC backend's all-reduce: one workgroup of 1024 threads per rank (n <= max_doubles)
constexpr long long kIpcTimeoutTicks = 1000000000;      // 10 s of s_memrealtime (100 MHz): the start-up skew between rank processes (code-object loads of a first launch) included
__global__ __launch_bounds__(1024) void k_ipc_allreduce(double *buf, size_t n, IpcPeers peers, int rank, int world, size_t max_doubles, long long epoch, int *fault)
{
    // (a peer that has not arrived once is gone: the exchanges still enqueued behind the failed one do not wait for it again)
    if (__hip_atomic_load(fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    const int parity = (int)(epoch & 1);
    const size_t slot = ((size_t)parity * world + rank) * max_doubles, flags = 2 * (size_t)world * max_doubles;
    for (int p = 0; p < world; ++p) {
        double *dst = peers.base[p] + slot;
        for (size_t i = threadIdx.x; i < n; i += 1024) __builtin_nontemporal_store(buf[i], dst + i);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");            // (system scope: my stores are visible wherever the flag is)
    __syncthreads();
    if ((int)threadIdx.x < world) {
        long long *f = reinterpret_cast<long long *>(peers.base[threadIdx.x] + flags) + (size_t)parity * world + rank;
        __hip_atomic_store(f, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __shared__ int s_late;
    if (threadIdx.x == 0) s_late = 0;
    __syncthreads();
    if ((int)threadIdx.x < world) {
        const long long *f = reinterpret_cast<const long long *>(peers.base[rank] + flags) + (size_t)parity * world + threadIdx.x;
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < epoch) {
            __builtin_amdgcn_s_sleep(8);
            if (wall_clock64() - t0 > kIpcTimeoutTicks) { s_late = 1; break; }
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    if (s_late) { if (threadIdx.x == 0) __hip_atomic_store(fault, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); return; }
    const double *mine = peers.base[rank] + (size_t)parity * world * max_doubles;
    for (size_t i = threadIdx.x; i < n; i += 1024) {
        double a = 0.0;
        for (int r = 0; r < world; ++r) a += __builtin_nontemporal_load(mine + (size_t)r * max_doubles + i);     // rank order: every rank receives the same bits
        buf[i] = a;
    }
}

// sum all-reduce of n doubles (in place) over the ranks of a multi-process communicator, on `stream`
static int comm_allreduce(tscm_comm *c, double *buf, size_t n, hipStream_t stream)
{
    if (c->ipc) {
        tscm_ipc *x = c->ipc;
        if (!x->connected) return fail(TSCM_E_INVALID, "tscm_comm_ipc_connect has not been called");
        if (x->dead) return fail(TSCM_E_PEER, "the IPC communicator is unusable after an earlier failure (a peer that did not arrive, or a failed solve)");
        IpcPeers peers{};
        for (int r = 0; r < x->world; ++r) peers.base[r] = static_cast<double *>(x->mapped[r]);
        for (size_t off = 0; off < n; off += x->max_doubles) {
            const size_t m = std::min(x->max_doubles, n - off);
            hipLaunchKernelGGL(k_ipc_allreduce, dim3(1), dim3(1024), 0, stream, buf + off, m, peers, x->rank, x->world, x->max_doubles, ++x->count, x->d_fault);
        }
        return 0;
    }
    NCCL_TRY(ncclAllReduce(buf, buf, n, ncclDouble, ncclSum, c->comm, stream));
    return 0;
}
// after a synchronisation: did an IPC exchange give up on a peer?
static int comm_check(tscm_comm *c)
{
    if (!c || !c->ipc) return 0;
    int f = 0;
    HIP_TRY(hipMemcpy(&f, c->ipc->d_fault, sizeof(int), hipMemcpyDeviceToHost));
    if (f) { c->ipc->dead = true; return fail(TSCM_E_PEER, "IPC exchange: a peer rank did not arrive within the bound (failed or gone); the communicator is unusable from here on"); }
    return 0;
}

struct LmRun {
    std::vector<tscm_solver *> m;
    bool separate_control() const { return m[0]->comm != nullptr; }
};

static int exchange(LmRun &run, bool t_buffer)
{
    tscm_solver *s0 = run.m[0];
    if (!s0->comm) return 0;
    const size_t n = t_buffer ? 256 * (size_t)s0->P.n_bids : 256 * (size_t)s0->P.C + kScal + 2 * (size_t)s0->P.world;
    if (n == 0) return 0;
    hipEvent_t e1 = nullptr;
    if (s0->comm->group) {
        tscm_local_group *g = s0->comm->group;      // pointer tables: [0, world) the members' T, [world, 2 world) their H_stage (run_lm)
        if (int rc = timed_begin(s0, t_buffer ? 1 : 2, g->stream, &e1)) return rc;
        hipLaunchKernelGGL(k_xchg_sum, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, g->stream, g->d_ptrs + (t_buffer ? 0 : g->world), g->world, n);
        if (e1) HIP_TRY(hipEventRecord(e1, g->stream));
        return 0;
    }
    double *buf = t_buffer ? s0->S.T : s0->S.H_stage;
    if (int rc = timed_begin(s0, t_buffer ? 1 : 2, s0->stream, &e1)) return rc;
    if (int rc = comm_allreduce(s0->comm, buf, n, s0->stream)) return rc;
    if (e1) HIP_TRY(hipEventRecord(e1, s0->stream));
    return 0;
}

// evaluation of the target point: pose constants, Gram kernel, reductions, statistics (+ all-reduce + control)
static int enqueue_eval(LmRun &run, int cand, int init, int have_backsub, bool have_view_constants = false)
{
    const bool sep = run.separate_control();
    const bool fused = !sep && run.m[0]->P.C <= kMaxCamLds;     // (rigs of more than 8 cameras: k_control as a launch of its own)
    for (tscm_solver *s : run.m) {
        const DevProblem &P = s->P;
        DevState &S = s->S;
        // the constants of a candidate point were written by k_backsub_prep; the initial point needs them here
        if (!have_backsub && !have_view_constants) hipLaunchKernelGGL(k_view_prep, dim3((P.V + P.C + kVPrepThreads - 1) / kVPrepThreads), dim3(kVPrepThreads), 0, s->stream, P, S, cand, s->f32_jacobian ? 1 : 0);
        if (int rc = launch_eval(s, cand)) return rc;
        if (fused && s->ctl_in_schur && (cand || init)) {
            // ... or the reductions alone: the next k_schur_gram takes the control step in its head (k_control_tail behind the
            // last evaluation of the solve).  Round 5: the solve's INITIAL evaluation as well (eval_pending = 1 | 4: IterationZero
            // in the head of the first Schur kernel) -- k_reduce_control's last workgroup cost every solve 18.4 us where
            // k_reduce_stats takes 5.5 and the head 4.4
            // ... and a candidate's reductions ride in that launch as well (eval_pending | 8: k_schur_gram<NV, true>)
            if (s->stats_ride && !init) { s->eval_pending = 1 | 8; continue; }
            hipLaunchKernelGGL(k_reduce_stats, dim3(P.C * kCamSl + S.n_st_blocks), dim3(256), 0, s->stream, P, S, cand, init);
            s->eval_pending = init ? 5 : 1;
            continue;
        }
        if (fused) {
            // one GPU: reductions, statistics and the control step in one launch
            hipLaunchKernelGGL(k_reduce_control, dim3(P.C * kCamSl + S.n_st_blocks), dim3(256), 0, s->stream, P, S, cand, init, have_backsub);
            continue;
        }
        hipLaunchKernelGGL(k_reduce_stats, dim3(P.C * kCamSl + S.n_st_blocks), dim3(256), 0, s->stream, P, S, cand, init);
        hipLaunchKernelGGL(k_finalize_eval, dim3(P.C + 1), dim3(256), 0, s->stream, P, S, have_backsub);
    }
    if (fused) return 0;
    if (int rc = exchange(run, /*t_buffer=*/false)) return rc;
    for (tscm_solver *s : run.m) {
        // behind the all-reduce: k_control -- or, for a candidate's evaluation, the head of the next k_schur_gram
        if (sep && s->ctl_in_schur && cand && !init) s->eval_pending = 2;
        else hipLaunchKernelGGL(k_control, dim3(1), dim3(256), 0, s->stream, s->P, s->S, init);
    }
    return 0;
}

// one GPU, up to 8 cameras: the T reduction rides in the reduced solve's launch (k_solve_nd<.., true>); with a communicator the
// all-reduce of T sits between the two
static bool fused_reduce(const tscm_solver *s) { return s->fuse_reduce && s->solve_variant <= 1 && !s->comm && s->P.n_bids > 0 && s->P.n_bids <= kSmallBids; }

static int enqueue_iteration(LmRun &run, int iteration)
{
    for (tscm_solver *s : run.m) {
        const DevProblem &P = s->P;
        DevState &S = s->S;
        const int ctl = s->eval_pending;                  // (ctl_in_schur: exactly one of the three variants below is launched)
        s->eval_pending = 0;
        if (P.n_slow) hipLaunchKernelGGL(k_schur_factor, dim3((P.n_slow + 255) / 256), dim3(256), 0, s->stream, P, S);
        const int ce = ctl ? ++s->ctl_epoch : 0;          // (ctl: 1 one GPU | 2 communicator, + 4: the initial evaluation's step, + 8: the reductions ride)
        if (ctl & 8) {
            const int ns = P.C * kCamSl + S.n_st_blocks, target = ns * ++s->stats_epoch;
            if (s->nv_chunks[1]) hipLaunchKernelGGL((k_schur_gram<1, true>), dim3(std::max(ns, s->nv_chunks[1]) + 1), dim3(256), s->lds_gram, s->stream, P, S, s->nv_chunk0[1], (ctl & 7) | (s->withhold == 2 ? 16 : 0), s->schur_resident_ride[1], ce, target, s->nv_chunks[1]);
            if (s->nv_chunks[2]) hipLaunchKernelGGL((k_schur_gram<2, true>), dim3(std::max(ns, s->nv_chunks[2]) + 1), dim3(256), s->lds_gram, s->stream, P, S, s->nv_chunk0[2], (ctl & 7) | (s->withhold == 2 ? 16 : 0), s->schur_resident_ride[2], ce, target, s->nv_chunks[2]);
            if (s->nv_chunks[3]) hipLaunchKernelGGL((k_schur_gram<3, true>), dim3(std::max(ns, s->nv_chunks[3]) + 1), dim3(256), s->lds_gram, s->stream, P, S, s->nv_chunk0[3], (ctl & 7) | (s->withhold == 2 ? 16 : 0), s->schur_resident_ride[3], ce, target, s->nv_chunks[3]);
        } else {
            if (s->nv_chunks[1] && s->schur_cb == 32) hipLaunchKernelGGL((k_schur_gram<1, false, 32>), dim3(s->nv_chunks[1] + (ctl ? 1 : 0)), dim3(256), s->lds_gram, s->stream, P, S, s->nv_chunk0[1], ctl, s->schur_resident[1], ce, 0, s->nv_chunks[1]);
            else if (s->nv_chunks[1]) hipLaunchKernelGGL(k_schur_gram<1>, dim3(s->nv_chunks[1] + (ctl ? 1 : 0)), dim3(256), s->lds_gram, s->stream, P, S, s->nv_chunk0[1], ctl, s->schur_resident[1], ce, 0, s->nv_chunks[1]);
            if (s->nv_chunks[2] && s->schur_cb == 32) hipLaunchKernelGGL((k_schur_gram<2, false, 32>), dim3(s->nv_chunks[2] + (ctl ? 1 : 0)), dim3(256), s->lds_gram, s->stream, P, S, s->nv_chunk0[2], ctl, s->schur_resident[2], ce, 0, s->nv_chunks[2]);
            else if (s->nv_chunks[2]) hipLaunchKernelGGL(k_schur_gram<2>, dim3(s->nv_chunks[2] + (ctl ? 1 : 0)), dim3(256), s->lds_gram, s->stream, P, S, s->nv_chunk0[2], ctl, s->schur_resident[2], ce, 0, s->nv_chunks[2]);
            if (s->nv_chunks[3]) hipLaunchKernelGGL(k_schur_gram<3>, dim3(s->nv_chunks[3] + (ctl ? 1 : 0)), dim3(256), s->lds_gram, s->stream, P, S, s->nv_chunk0[3], ctl, s->schur_resident[3], ce, 0, s->nv_chunks[3]);
        }
        if (P.n_pchunks) hipLaunchKernelGGL(k_pair_gram, dim3(P.n_pchunks), dim3(256), 0, s->stream, P, S);
        if (P.n_bids && !fused_reduce(s)) hipLaunchKernelGGL(k_T_reduce, dim3(P.n_bids * (256 / kTEntries)), dim3(kTEntries * kTSlices), 0, s->stream, P, S);
    }
    if (int rc = exchange(run, /*t_buffer=*/true)) return rc;
    for (tscm_solver *s : run.m)          // (tests only: tscm_solver_debug_perturb_exchange)
        if (s->perturb_at == iteration && s->comm && s->P.n_bids) hipLaunchKernelGGL(k_debug_perturb, dim3(1), dim3(1), 0, s->stream, s->S.T, 256 * (size_t)s->P.n_bids, s->perturb_ulps);
    for (tscm_solver *s : run.m) {
        const DevProblem &P = s->P;
        DevState &S = s->S;
        const int wf = s->f32_jacobian ? 1 : 0;
        if (s->solve_variant == 0 && !s->graph_order) {
            // up to 4 cameras, one dense block: the same launch shape with k_solve_reduced as the solver workgroup
            const int n_prod = fused_reduce(s) ? P.n_bids * (256 / kFusedEntries) : 0;
            const int n_bs = s->fuse_backsub && s->bs_threads == 256 && S.n_bs_blocks <= s->dense4_resident - 1 - n_prod ? S.n_bs_blocks : 0;       // all of them, or none: see below
            // (TSCM_EXEC_MFMA_REDUCED_SOLVE: the factorisation on one wave with MFMA rank-4 updates -- MF; same bits, measured slower)
            if (n_prod || n_bs) {
                if (s->solve_tiles) hipLaunchKernelGGL((k_solve_reduced<4, 16, 64, true, false>), dim3(1 + n_prod + n_bs), dim3(256), std::max(s->lds_dense4, n_bs ? s->lds_bs : (size_t)0), s->stream,
                                                       P, S, ++s->t_epoch, s->withhold == 1 ? 1 : 0, n_prod, n_bs, wf);
                else hipLaunchKernelGGL((k_solve_reduced<4, 16, 64, true, true>), dim3(1 + n_prod + n_bs), dim3(256), std::max(s->lds_dense4, n_bs ? s->lds_bs : (size_t)0), s->stream,
                                        P, S, ++s->t_epoch, s->withhold == 1 ? 1 : 0, n_prod, n_bs, wf);
            } else if (s->solve_tiles) hipLaunchKernelGGL((k_solve_reduced<4, 16, 64, false, false>), dim3(1), dim3(256), s->lds_dense4, s->stream, P, S, 0, 0, 0, 0, 0);
            else hipLaunchKernelGGL((k_solve_reduced<4, 16, 64, false, true>), dim3(1), dim3(256), s->lds_dense4, s->stream, P, S, 0, 0, 0, 0, 0);
            if (!n_bs && S.n_bs_blocks && s->bs_threads == 128) hipLaunchKernelGGL(k_backsub_prep<128>, dim3(S.n_bs_blocks), dim3(128), s->lds_bs, s->stream, P, S, wf);
            if (!n_bs && S.n_bs_blocks && s->bs_threads == 256) hipLaunchKernelGGL(k_backsub_prep<256>, dim3(S.n_bs_blocks), dim3(256), s->lds_bs, s->stream, P, S, wf);
            continue;
        }
        if (s->solve_variant <= 1) {
            // 5 to 8 cameras (and TSCM_EXEC_GRAPH_REDUCED_ORDER): T reduction (one GPU), reduced solve and -- unless TSCM_EXEC_SEPARATE_BACKSUB -- the back-substitution
            // workgroups, which wait for the camera step with their operands loaded, in ONE launch -- if ALL of them are
            // resident next to the solver workgroup and the producers (occupancy x CUs: a waiting workgroup that keeps the
            // solver off the chip would wait for ever); otherwise the back-substitution is a launch of its own.  Splitting
            // it between the two was measured and lost: at config 5 (5,000 groups, 255 of them riding) 364.3 against 356.7 us
            // per iteration -- the riders share the solver workgroup's CU and the rest needs its launch anyway
            const int v = s->nd;
            const int n_prod = fused_reduce(s) ? P.n_bids * (256 / kFusedEntries) : 0;
            const int n_bs = s->fuse_backsub && s->bs_threads == 256 && S.n_bs_blocks <= s->nd_resident[v] - 1 - n_prod ? S.n_bs_blocks : 0;
            const bool two = s->plan[v].tpt == 2;
            if (n_prod || n_bs) {
                const size_t lds = std::max(s->lds_nd[v], n_bs ? s->lds_bs : (size_t)0);
                const dim3 grid(1 + n_prod + n_bs);
                if (two) {
                    hipLaunchKernelGGL((k_solve_nd<2, true>), grid, dim3(kNdThreads), lds, s->stream, P, S, s->d_nd_map[v], s->d_nd_tab[v], s->d_nd_bs[v], s->plan[v].dims(), ++s->t_epoch, s->withhold == 1 ? 1 : 0, n_prod, n_bs, wf);
                } else {
                    hipLaunchKernelGGL((k_solve_nd<1, true>), grid, dim3(kNdThreads), lds, s->stream, P, S, s->d_nd_map[v], s->d_nd_tab[v], s->d_nd_bs[v], s->plan[v].dims(), ++s->t_epoch, s->withhold == 1 ? 1 : 0, n_prod, n_bs, wf);
                }
            } else {
                if (two) hipLaunchKernelGGL((k_solve_nd<2, false>), dim3(1), dim3(kNdThreads), s->lds_nd[v], s->stream, P, S, s->d_nd_map[v], s->d_nd_tab[v], s->d_nd_bs[v], s->plan[v].dims(), 0, 0, 0, 0, 0);
                else hipLaunchKernelGGL((k_solve_nd<1, false>), dim3(1), dim3(kNdThreads), s->lds_nd[v], s->stream, P, S, s->d_nd_map[v], s->d_nd_tab[v], s->d_nd_bs[v], s->plan[v].dims(), 0, 0, 0, 0, 0);
            }
            if (!n_bs && S.n_bs_blocks && s->bs_threads == 128) hipLaunchKernelGGL(k_backsub_prep<128>, dim3(S.n_bs_blocks), dim3(128), s->lds_bs, s->stream, P, S, wf);
            if (!n_bs && S.n_bs_blocks && s->bs_threads == 256) hipLaunchKernelGGL(k_backsub_prep<256>, dim3(S.n_bs_blocks), dim3(256), s->lds_bs, s->stream, P, S, wf);
            continue;
        }
        hipLaunchKernelGGL(k_solve_reduced_big, dim3(1), dim3(kBigNT), s->lds_solve, s->stream, P, S);
        if (S.n_bs_blocks && s->bs_threads == 128) hipLaunchKernelGGL(k_backsub_prep<128>, dim3(S.n_bs_blocks), dim3(128), s->lds_bs, s->stream, P, S, wf);
        if (S.n_bs_blocks && s->bs_threads == 256) hipLaunchKernelGGL(k_backsub_prep<256>, dim3(S.n_bs_blocks), dim3(256), s->lds_bs, s->stream, P, S, wf);
    }
    return enqueue_eval(run, /*cand=*/1, /*init=*/0, /*have_backsub=*/1);
}

static const char *reason_message(int r)
{
    switch (r) {
    case kMaxIter: return "Maximum number of iterations reached.";
    case kGradTol: return "Gradient tolerance reached.";
    case kMinRadius: return "Minimum trust region radius reached.";
    case kParamTol: return "Parameter tolerance reached.";
    case kFuncTol: return "Function tolerance reached.";
    case kInvalidSteps: return "Number of consecutive invalid steps more than Solver::Options::max_num_consecutive_invalid_steps.";
    case kRanksDisagree: return "The ranks of the communicator disagree about the state of the minimizer.";
    default: return "";
    }
}

// restores the per-solve state of the members on every exit path (also the error returns inside the loop)
struct LmRunGuard {
    LmRun &run;
    ~LmRunGuard()
    {
        for (tscm_solver *s : run.m) {
            s->f32_jacobian = false;         // the operator-level entry points are always fp64
            s->ev_used = 0;
            s->stream = s->own_stream;
        }
    }
};

static int run_lm_inner(LmRun &run, const tscm_options *opt_in, tscm_summary *sums, int reset, bool rerun, bool *late_handoff);

// Waits for the solver's stream.  With a multi-rank RCCL communicator a peer that has failed (or died) leaves this
// rank's all-reduce kernel spinning for ever -- over the intra-node transports an ncclCommAbort on the FAILING rank does
// not reach the others -- so the wait is a poll with a watchdog: no completion within kCommWatchdogSeconds aborts the
// communicator locally and fails the call with TSCM_E_RCCL.  (A whole solve of the largest supported problem is well
// under a second of device time; the bound only has to exceed the start-up skew between the rank processes.)
constexpr double kCommWatchdogSeconds = 60.0;
static int sync_stream(tscm_solver *s)
{
    tscm_comm *c = s->comm;
    // (round 6: spinning on hipStreamQuery instead of blocking here was measured -- 28-30 us of host time around a natural solve
    // against 31, nothing in the 20-step line -- and not kept)
    if (!c || c->group || !c->comm || c->world <= 1) { HIP_TRY(hipStreamSynchronize(s->stream)); return 0; }
    const double t0 = wall();
    for (;;) {
        const hipError_t q = hipStreamQuery(s->stream);
        if (q == hipSuccess) return 0;
        if (q != hipErrorNotReady) { HIP_TRY(q); }
        if (wall() - t0 > kCommWatchdogSeconds) {
            (void)ncclCommAbort(c->comm);
            c->comm = nullptr;
            return fail(TSCM_E_RCCL, "no progress on the solver stream within the watchdog interval: a peer rank has failed or is gone (communicator aborted)");
        }
        std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
}

// An RCCL rank that leaves the loop on an error aborts its communicator (unusable afterwards, like after any RCCL
// error).  That does NOT unblock its peers by itself: they leave their next all-reduce through the watchdog of
// sync_stream (TSCM_E_RCCL), or are torn down by whoever started the ranks (bench.py's launcher ends all ranks as soon
// as one exits non-zero).  A failed rank must not be re-used: start a fresh process.
static int run_lm(LmRun &run, const tscm_options *opt_in, tscm_summary *sums, int reset)
{
    bool late = false;
    int rc = run_lm_inner(run, opt_in, sums, reset, /*rerun=*/false, &late);
    if (late && !run.m[0]->comm && run.m.size() == 1 && !run.m[0]->no_rerun) {
        // A device-side hand-off of a fused launch came late (0.5 s: a debugger, a co-tenant, a context switch -- or a real
        // fault).  The solve was stopped on the device and nothing of it has left it; it is run again from its start point
        // (k_begin_view_prep kept a copy) on the launches that hand nothing over inside a launch -- same mathematics, same
        // bits as the fused ones (tests/test_gpu_parity.py).  Only if that fails too is it an error.  With a communicator the
        // ranks would have to agree on the re-run: there the late hand-off stays TSCM_E_HIP.
        tscm_options o2;
        tscm_default_options(&o2, run.m[0]->mono);
        if (opt_in) std::memcpy(&o2, opt_in, std::min(opt_in->struct_size, sizeof(tscm_options)));
        o2.struct_size = sizeof(tscm_options);
        o2.exec_flags |= TSCM_EXEC_SEPARATE_T_REDUCE | TSCM_EXEC_SEPARATE_BACKSUB | TSCM_EXEC_SEPARATE_CONTROL;
        const std::string first = g_err;
        bool late2 = false;
        rc = run_lm_inner(run, &o2, sums, reset, /*rerun=*/true, &late2);
        if (rc == 0) { g_err = "note: " + first + "; the solve was run again on separate launches and completed"; ++run.m[0]->n_reruns; }
        else g_err = "re-run after [" + first + "] failed: " + g_err;          // (tscm_solver_reruns counts completed re-runs only)
    }
    tscm_comm *c = run.m[0]->comm;
    if (rc != 0 && rc != TSCM_E_INVALID && c && c->ipc && c->world > 1) c->ipc->dead = true;     // (its exchange counter may be behind the peers' now)
    if (rc != 0 && rc != TSCM_E_INVALID && c && c->comm && c->world > 1) {
        const std::string keep = g_err;
        (void)ncclCommAbort(c->comm);
        c->comm = nullptr;
        g_err = keep + " (communicator aborted)";
    }
    return rc;
}

static int run_lm_inner(LmRun &run, const tscm_options *opt_in, tscm_summary *sums, int reset, bool rerun, bool *late_handoff)
{
    tscm_solver *s0 = run.m[0];
    for (tscm_solver *s : run.m) if (!s->have_init) return fail(TSCM_E_INVALID, "tscm_solver_upload_params has not been called");
    // the caller's struct may be SHORTER than this library's (built against an older header of ABI >= 6): read what it has, the
    // rest keeps its default.  The shortest struct the library knows ends behind exec_flags (ABI 6); a size of 0, a smaller or a
    // larger one is refused -- a struct of ABI <= 5 has max_num_iterations where struct_size is and never passes
    tscm_options opt;
    tscm_default_options(&opt, s0->mono);
    if (opt_in) {
        constexpr size_t kMinOptions = offsetof(tscm_options, exec_flags) + sizeof(int);
        if (opt_in->struct_size < kMinOptions || opt_in->struct_size > sizeof(tscm_options))
            return fail(TSCM_E_INVALID, "tscm_options.struct_size is not a size this library knows (initialise the struct with tscm_default_options; ABI 6)");
        std::memcpy(&opt, opt_in, opt_in->struct_size);
        opt.struct_size = sizeof(tscm_options);
    }
    if (opt.max_num_iterations < 0 || opt.max_num_iterations > TSCM_MAX_ITERATIONS) return fail(TSCM_E_INVALID, "max_num_iterations must be in [0, 255]");
    if (opt.exec_flags & ~TSCM_EXEC_ALL) return fail(TSCM_E_INVALID, "unknown bits in tscm_options.exec_flags (an options struct of an older ABI?)");
    HIP_TRY(hipSetDevice(s0->device));
    LmRunGuard guard{ run };
    for (tscm_solver *s : run.m) {
        s->comm = effective_comm(s, opt.exec_flags);
        s->fuse_reduce = !(opt.exec_flags & TSCM_EXEC_SEPARATE_T_REDUCE);
        s->fuse_backsub = !(opt.exec_flags & TSCM_EXEC_SEPARATE_BACKSUB);
        {
            // one GPU with <= 8 cameras (finish_evaluation's LDS fits k_schur_gram's) or a communicator; exactly one Schur kernel per iteration
            const int n_variants = (s->nv_chunks[1] ? 1 : 0) + (s->nv_chunks[2] ? 1 : 0) + (s->nv_chunks[3] ? 1 : 0);
            // (a grid of several rounds -- config 5 on one GPU: 1256 workgroups, 2.5 rounds -- pays the step in its first round
            // only: the later rounds read the outcome workgroup 0 publishes)
            s->ctl_in_schur = (s->comm || s->P.C <= kMaxCamLds) && s->P.n_slow == 0 && s->P.n_pchunks == 0 && n_variants == 1 &&
                              !(opt.exec_flags & TSCM_EXEC_SEPARATE_CONTROL);
            s->eval_pending = 0;
            s->ctl_epoch = 0;
            // (one GPU, the control step from the finished sums themselves: not behind an all-reduce)
            // -- and a grid of ONE round: every workgroup that takes a reduction block in front of its chunk is resident (they wait for
            // each other), and at config 5 (1256 chunks, 2.5 rounds) the ride costs 2.5 us where it saves 4 at config 4
            const int nv_used = s->nv_chunks[1] ? 1 : s->nv_chunks[2] ? 2 : 3;
            s->stats_ride = s->ctl_in_schur && !s->comm && s->P.C <= kMaxCamLds && !(opt.exec_flags & TSCM_EXEC_SEPARATE_STATS) &&
                            std::max(s->P.C * kCamSl + s->S.n_st_blocks, s->nv_chunks[nv_used]) + 1 <= s->schur_resident_ride[nv_used];
            s->stats_epoch = 0;
        }
        s->withhold = s->withhold_next; s->withhold_next = 0;
        s->perturb_at = s->perturb_at_next; s->perturb_at_next = 0;
        if (!rerun) { s->no_rerun = s->no_rerun_next; s->no_rerun_next = false; }
        s->gram16 = (opt.exec_flags & TSCM_EXEC_GRAM_16X16) != 0;
        s->one_view_per_pass = (opt.exec_flags & TSCM_EXEC_ONE_VIEW_PER_PASS) != 0;
        s->solve_tiles = !(opt.exec_flags & TSCM_EXEC_MFMA_REDUCED_SOLVE) || s->P.n_act > 46;      // (MF needs row 47 for the right-hand side)
        s->nd = (opt.exec_flags & TSCM_EXEC_DENSE_REDUCED_ORDER) ? 1 : 0;
        s->graph_order = (opt.exec_flags & TSCM_EXEC_GRAPH_REDUCED_ORDER) != 0 || (s->solve_variant == 0 && s->nd);
        s->t_epoch = 0;
    }
    if (s0->comm && !s0->comm->group && !s0->comm->ipc && !s0->comm->comm) return fail(TSCM_E_RCCL, "the communicator was aborted by an earlier failure");
    if (s0->comm && s0->comm->ipc && s0->comm->ipc->dead) return fail(TSCM_E_PEER, "the IPC communicator is unusable after an earlier failure (a peer that did not arrive, or a failed solve)");
    if (s0->comm && s0->comm->group && s0->comm->group->dead) return fail(TSCM_E_PEER, "the local group is unusable after its ranks disagreed about the LM state");
    if (s0->comm && s0->comm->group) {
        // a local group runs on ONE stream: lock step by stream order, no events
        tscm_local_group *g = s0->comm->group;
        if ((int)run.m.size() != g->world) return fail(TSCM_E_INVALID, "a local group solves with all of its members (tscm_solver_solve_group)");
        std::vector<double *> ptrs(2 * (size_t)g->world);
        for (int r = 0; r < g->world; ++r) { ptrs[r] = run.m[r]->S.T; ptrs[g->world + r] = run.m[r]->S.H_stage; }
        HIP_TRY(hipMemcpy(g->d_ptrs, ptrs.data(), sizeof(double *) * ptrs.size(), hipMemcpyHostToDevice));
        for (tscm_solver *s : run.m) { HIP_TRY(hipStreamSynchronize(s->own_stream)); s->stream = g->stream; }
    }
    for (size_t r = 0; r < run.m.size(); ++r) std::memset(&sums[r], 0, sizeof(tscm_summary));
    for (tscm_solver *s : run.m) {
        s->f32_jacobian = opt.jacobian_fp32 != 0;
        if (s->f32_jacobian && s->lds_eval32 > 64 * 1024)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(s->eval32), hipFuncAttributeMaxDynamicSharedMemorySize, (int)s->lds_eval32));
    }

    // control block (identical on every rank), counter of the fused T reduction, start point: one launch (k_begin_solve)
    const double t0 = wall();
    for (tscm_solver *s : run.m) {
        CtrlHead h;
        std::memset(&h, 0, sizeof(h));
        h.radius = opt.initial_trust_region_radius;
        h.decrease_factor = 2.0;
        h.opt.max_num_iterations = opt.max_num_iterations;
        h.opt.function_tolerance = opt.function_tolerance;
        h.opt.gradient_tolerance = opt.gradient_tolerance;
        h.opt.parameter_tolerance = opt.parameter_tolerance;
        h.opt.initial_radius = opt.initial_trust_region_radius;
        h.opt.max_radius = opt.max_trust_region_radius;
        h.opt.min_radius = opt.min_trust_region_radius;
        h.opt.min_relative_decrease = opt.min_relative_decrease;
        h.opt.min_lm_diagonal = opt.min_lm_diagonal;
        h.opt.max_lm_diagonal = opt.max_lm_diagonal;
        h.opt.max_invalid = opt.max_num_consecutive_invalid_steps;
        h.opt.jacobi_scaling = opt.jacobi_scaling;
        // ... and the constants of the initial evaluation (k_view_prep's work) in the same launch
        // (start point: the registered arrays with `reset`, buffer 0 otherwise, the backup of the first attempt on a re-run -- the
        // first attempt leaves that backup behind)
        const bool from_init = reset && !rerun;
        hipLaunchKernelGGL(k_begin_view_prep, dim3((s->P.V + s->P.C + kVPrepThreads - 1) / kVPrepThreads), dim3(kVPrepThreads), 0, s->stream, s->P, s->S, h,
                           rerun ? s->d_start_cam : from_init ? s->d_init_cam : nullptr, rerun ? s->d_start_intr : from_init ? s->d_init_intr : nullptr,
                           rerun ? s->d_start_board : from_init ? s->d_init_board : nullptr,
                           rerun ? nullptr : s->d_start_cam, rerun ? nullptr : s->d_start_intr, rerun ? nullptr : s->d_start_board, s->f32_jacobian ? 1 : 0);
    }

    int rc;
    if ((rc = enqueue_eval(run, /*cand=*/0, /*init=*/1, /*have_backsub=*/0, /*have_view_constants=*/true))) return rc;
    const int check_every = std::max(1, opt.check_every);
    bool done = false;
    for (int it = 1; it <= opt.max_num_iterations && !done; ++it) {
        if ((rc = enqueue_iteration(run, it))) return rc;
        if (it % check_every == 0 && it < opt.max_num_iterations) {
            // every rank takes the same decisions from the same all-reduced bits: polling one member is enough
            HIP_TRY(hipMemcpyAsync(s0->h_ctrl, s0->S.ctrl, 64, hipMemcpyDeviceToHost, s0->stream));
            if ((rc = sync_stream(s0))) return rc;
            done = s0->h_ctrl->done != 0;
        }
    }
    // The end of the solve, enqueued behind the last iteration -- ONE synchronisation for the whole solve: the last evaluation's
    // control step if the steps were taken in k_schur_gram's head, the accepted point into buffer 0, the control block and the
    // iteration log to the host.  One GPU: one launch (k_finish_solve); communicator: k_control, k_end_solve and a copy.
    static_assert(sizeof(Ctrl) == sizeof(CtrlHead) + sizeof(IterLog) * kMaxLog, "the log follows the head without padding");
    const size_t ctrl_bytes = sizeof(CtrlHead) + sizeof(IterLog) * (size_t)std::min(opt.max_num_iterations + 1, kMaxLog);
    for (tscm_solver *s : run.m) {
        const int nb = std::min(256, (6 * std::max(s->B, s->C) + 255) / 256 + 1);
        if (s->eval_pending & 1) {
            // (the reductions of the solve's last evaluation found no Schur kernel to ride in)
            if (s->eval_pending & 8) hipLaunchKernelGGL(k_reduce_stats, dim3(s->P.C * kCamSl + s->S.n_st_blocks), dim3(256), 0, s->stream, s->P, s->S, 1, 0);
            const int was_init = (s->eval_pending >> 2) & 1;
            hipLaunchKernelGGL(k_finish_solve, dim3(nb + 1), dim3(256), 0, s->stream, s->P, s->S, was_init, !was_init, s->C, s->B, s->d_h_ctrl);
            s->eval_pending = 0;
            continue;
        }
        if (s->eval_pending == 2) hipLaunchKernelGGL(k_control, dim3(1), dim3(256), 0, s->stream, s->P, s->S, 0);
        s->eval_pending = 0;
        hipLaunchKernelGGL(k_end_solve, dim3(nb), dim3(256), 0, s->stream, s->S, s->C, s->B);
        HIP_TRY(hipMemcpyAsync(s->h_ctrl, s->S.ctrl, ctrl_bytes, hipMemcpyDeviceToHost, s->stream));
    }
    if ((rc = sync_stream(s0))) return rc;
    for (tscm_solver *s : run.m) if (s->stream != s0->stream) HIP_TRY(hipStreamSynchronize(s->stream));
    if ((rc = comm_check(s0->comm))) return rc;
    const double t1 = wall();
    HIP_TRY(hipGetLastError());
    for (size_t r = 0; r < run.m.size(); ++r) {
        tscm_solver *s = run.m[r];
        Ctrl *h = s->h_ctrl;
        tscm_summary *sum = &sums[r];
        if ((rc = collect_timing(s))) return rc;
        if (h->fault == kFaultRanksDisagree) {
            // the rank-divergence guard (decision_word, tscm_kernels.h): every rank stopped in the same control step
            if (s->comm && s->comm->group) s->comm->group->dead = true;
            char msg[256];
            std::snprintf(msg, sizeof(msg), "the ranks of the communicator disagree about the LM state at iteration %d (rank %d of %d: an all-reduce that did not hand "
                          "every rank the same bits, or a replicated computation that diverged); the solve was stopped on every rank", h->iteration + 1, s->rank, s->world);
            return fail(TSCM_E_PEER, msg);
        }
        if (h->fault) { *late_handoff = true; return fail(TSCM_E_HIP, "a device-side hand-off (Schur-complement tiles -> reduced solve) did not arrive within its time bound: the solve was stopped"); }
        if (!h->done) return fail(TSCM_E_HIP, "device LM loop did not terminate");
        sum->termination_type = h->term_type;
        sum->num_iterations = std::min(h->n_log, TSCM_MAX_ITERATIONS + 1);
        sum->num_successful_steps = h->num_successful;
        sum->num_unsuccessful_steps = h->num_unsuccessful;
        sum->initial_cost = h->initial_cost;
        sum->final_cost = h->x_cost;
        sum->n_residual_blocks = (int)s->N_total;
        sum->lm_iterations = h->lm_iterations;
        for (int i = 0; i < sum->num_iterations; ++i) {
            const IterLog &l = h->log[i];
            tscm_iteration &o = sum->iterations[i];
            o.iteration = l.iteration; o.step_is_valid = l.step_is_valid; o.step_is_successful = l.step_is_successful;
            o.cost = l.cost; o.cost_change = l.cost_change; o.gradient_max_norm = l.gradient_max_norm; o.gradient_norm = l.gradient_norm;
            o.step_norm = l.step_norm; o.relative_decrease = l.relative_decrease; o.trust_region_radius = l.radius;
        }
        std::snprintf(sum->message, sizeof(sum->message), "%s", reason_message(h->term_reason));
        sum->seconds_solve = 1e-8 * (double)(h->t_end - h->t_begin);      // on the device: first to last kernel of the solve
        sum->seconds_total = t1 - t0;                                     // wall time of the call
        sum->rmse = s->N_total ? std::sqrt(2.0 * h->x_cost / (double)s->N_total) : 0.0;     // cost and N of the WHOLE job
    }
    return 0;
}

extern "C" int tscm_solver_solve_resident(tscm_solver *s, const tscm_options *opt_in, tscm_summary *sum, int reset)
{
    if (!s || !sum) return fail(TSCM_E_INVALID, "NULL argument");
    if (s->comm_reg && s->comm_reg->group && s->world > 1) return fail(TSCM_E_INVALID, "member of a local group: use tscm_solver_solve_group");
    if (s->world > 1 && !s->comm_reg) return fail(TSCM_E_INVALID, "sharded solver without a communicator (tscm_solver_set_comm)");
    LmRun run;
    run.m.push_back(s);
    return run_lm(run, opt_in, sum, reset);
}

extern "C" int tscm_solver_solve_group(tscm_solver **solvers, int n, const tscm_options *opt, tscm_summary *summaries, int reset)
{
    if (!solvers || !summaries || n < 1) return fail(TSCM_E_INVALID, "NULL argument");
    LmRun run;
    for (int r = 0; r < n; ++r) {
        tscm_solver *s = solvers[r];
        if (!s || s->world != n || s->rank != r) return fail(TSCM_E_INVALID, "solvers[r] must be shard r of n (tscm_solver_create_sharded)");
        if (n > 1 && (!s->comm_reg || !s->comm_reg->group || s->comm_reg->group != solvers[0]->comm_reg->group))
            return fail(TSCM_E_INVALID, "the solvers of a group need the communicators of ONE tscm_comm_create_local call");
        run.m.push_back(s);
    }
    return run_lm(run, opt, summaries, reset);
}

extern "C" int tscm_solver_download_params(tscm_solver *s, double *cam_rt, double *intr, double *board_rt)
{
    if (!s) return fail(TSCM_E_INVALID, "solver is NULL");
    HIP_TRY(hipSetDevice(s->device));
    if (cam_rt && !s->mono) HIP_TRY(hipMemcpy(cam_rt, s->S.cam_rt[0], sizeof(double) * 6 * s->C, hipMemcpyDeviceToHost));
    if (intr) HIP_TRY(hipMemcpy(intr, s->S.intr[0], sizeof(double) * 9 * s->C, hipMemcpyDeviceToHost));
    // only the owned boards: the other entries of the caller's array are left untouched (see tscm_solver_gather_boards)
    if (board_rt && s->B) {
        std::vector<double> brd(6 * (size_t)s->B);
        HIP_TRY(hipMemcpy(brd.data(), s->S.board_rt[0], sizeof(double) * 6 * s->B, hipMemcpyDeviceToHost));
        for (int i = 0; i < s->B; ++i) std::memcpy(board_rt + 6 * ((size_t)s->b0 + s->board_perm[i]), brd.data() + 6 * (size_t)i, 6 * sizeof(double));
    }
    return 0;
}

// After a sharded solve every rank holds the poses of its own boards.  This makes the caller's full-length array
// complete on every rank (what MultiCalib::calibrate() leaves behind: all chessboards_[i].rt_ updated): the owned
// slice in a zeroed full-length device buffer, one sum all-reduce (x + 0 is exact), one download.
extern "C" int tscm_solver_gather_boards(tscm_solver *s, double *board_rt)
{
    if (!s || (!board_rt && s->B_total)) return fail(TSCM_E_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(s->device));
    if (!s->comm_reg || s->world == 1 || s->comm_reg->group) return tscm_solver_download_params(s, nullptr, nullptr, board_rt);   // local groups share the caller's array
    if (s->B_total == 0) return 0;
    double *full = nullptr;
    const size_t n = 6 * (size_t)s->B_total;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&full), n * sizeof(double)));
    std::unique_ptr<double, void (*)(double *)> guard(full, [](double *q) { (void)hipFree(q); });
    {
        std::vector<double> mine(n, 0.0);
        if (int rc = tscm_solver_download_params(s, nullptr, nullptr, mine.data())) return rc;     // owned boards at their own positions
        HIP_TRY(hipMemcpy(full, mine.data(), n * sizeof(double), hipMemcpyHostToDevice));
    }
    if (int rc = comm_allreduce(s->comm_reg, full, n, s->stream)) return rc;
    HIP_TRY(hipMemcpyAsync(board_rt, full, n * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return comm_check(s->comm_reg);
}

extern "C" int tscm_solver_solve(tscm_solver *s, const tscm_options *opt, tscm_summary *sum)
{
    if (!s || !sum) return fail(TSCM_E_INVALID, "NULL argument");
    const double t0 = wall();
    int rc;
    if ((rc = tscm_solver_upload_params(s, s->h_cam_rt, s->h_intr, s->h_board_rt))) return rc;
    if ((rc = tscm_solver_solve_resident(s, opt, sum, 1))) return rc;
    if ((rc = tscm_solver_download_params(s, s->h_cam_rt, s->h_intr, nullptr))) return rc;
    if ((rc = tscm_solver_gather_boards(s, s->h_board_rt))) return rc;
    sum->seconds_total = wall() - t0;
    return 0;
}

static int solve_once(const tscm_problem *p, const tscm_options *opt, tscm_summary *sum)
{
    tscm_solver *s = nullptr;
    int dev = 0;
    (void)hipGetDevice(&dev);
    int rc = tscm_solver_create(p, dev, &s);
    if (rc) return rc;
    rc = tscm_solver_solve(s, opt, sum);
    tscm_solver_destroy(s);
    return rc;
}

extern "C" int tscm_solve_multi(const tscm_problem *p, const tscm_options *opt, tscm_summary *sum)
{
    if (p && p->mono) return fail(TSCM_E_INVALID, "tscm_solve_multi called with a mono problem");
    return solve_once(p, opt, sum);
}

extern "C" int tscm_solve_mono(const tscm_problem *p, const tscm_options *opt, tscm_summary *sum)
{
    if (p && !p->mono) return fail(TSCM_E_INVALID, "tscm_solve_mono called with a multi-camera problem");
    return solve_once(p, opt, sum);
}

// ------------------------------------------------------------------------------------------------
// operator level
// ------------------------------------------------------------------------------------------------
// upload the problem's current parameters into buffer 0 and compute the pose constants
static int prepare_eval(tscm_solver *s)
{
    int rc;
    if ((rc = tscm_solver_upload_params(s, s->h_cam_rt, s->h_intr, s->h_board_rt))) return rc;
    DevState &S = s->S;
    HIP_TRY(hipMemset(S.ctrl, 0, sizeof(Ctrl)));
    HIP_TRY(hipMemcpy(S.cam_rt[0], s->d_init_cam, sizeof(double) * 6 * s->C, hipMemcpyDeviceToDevice));
    HIP_TRY(hipMemcpy(S.intr[0], s->d_init_intr, sizeof(double) * 9 * s->C, hipMemcpyDeviceToDevice));
    if (s->B) HIP_TRY(hipMemcpy(S.board_rt[0], s->d_init_board, sizeof(double) * 6 * s->B, hipMemcpyDeviceToDevice));
    hipLaunchKernelGGL(k_pose_prep, dim3((s->P.B + s->P.C + 255) / 256), dim3(256), 0, s->stream, s->P, S, 0);
    hipLaunchKernelGGL(k_view_prep, dim3((s->P.V + s->P.C + kVPrepThreads - 1) / kVPrepThreads), dim3(kVPrepThreads), 0, s->stream, s->P, S, 0, 0);
    HIP_TRY(hipStreamSynchronize(s->stream));
    return 0;
}

extern "C" int tscm_eval_functor(const tscm_problem *p, int device, double *residuals, double *J_cam,
                                 double *J_board, double *J_intr, double *cost)
{
    tscm_solver *s = nullptr;
    int rc = tscm_solver_create(p, device, &s);
    if (rc) return rc;
    std::unique_ptr<tscm_solver, void (*)(tscm_solver *)> guard(s, tscm_solver_destroy);
    if ((rc = prepare_eval(s))) return rc;
    const size_t N = (size_t)s->N;
    std::vector<int> corner_view(N);
    for (int v = 0; v < s->V; ++v) for (int j = 0; j < s->h_view_count[v]; ++j) corner_view[s->h_view_obs[v] + j] = v;
    const int *d_cv = nullptr;
    double *d_res = nullptr, *d_Jc = nullptr, *d_Jb = nullptr, *d_Ji = nullptr;
    if ((rc = dev_upload(s, &d_cv, corner_view))) return rc;
    if ((rc = dev_alloc(s, &d_res, 2 * N))) return rc;
    if (J_cam && (rc = dev_alloc(s, &d_Jc, 12 * N))) return rc;
    if (J_board && (rc = dev_alloc(s, &d_Jb, 12 * N))) return rc;
    if (J_intr && (rc = dev_alloc(s, &d_Ji, 18 * N))) return rc;
    if (N) hipLaunchKernelGGL(k_eval_functor, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s->stream, s->P, s->S, d_cv, d_res, d_Jc, d_Jb, d_Ji);
    HIP_TRY(hipStreamSynchronize(s->stream));
    HIP_TRY(hipGetLastError());
    // device corner order -> problem corner order (views in problem order, empty views skipped)
    std::vector<double> h_res(2 * N), h_Jc(d_Jc ? 12 * N : 0), h_Jb(d_Jb ? 12 * N : 0), h_Ji(d_Ji ? 18 * N : 0);
    if (N) HIP_TRY(hipMemcpy(h_res.data(), d_res, sizeof(double) * 2 * N, hipMemcpyDeviceToHost));
    if (d_Jc && N) HIP_TRY(hipMemcpy(h_Jc.data(), d_Jc, sizeof(double) * 12 * N, hipMemcpyDeviceToHost));
    if (d_Jb && N) HIP_TRY(hipMemcpy(h_Jb.data(), d_Jb, sizeof(double) * 12 * N, hipMemcpyDeviceToHost));
    if (d_Ji && N) HIP_TRY(hipMemcpy(h_Ji.data(), d_Ji, sizeof(double) * 18 * N, hipMemcpyDeviceToHost));
    std::vector<long> orig_row(p->n_views, 0);
    { long k = 0; for (int v = 0; v < p->n_views; ++v) { orig_row[v] = k; k += p->view_count[v]; } }
    double c = 0.0;
    for (int dv = 0; dv < s->V; ++dv) {
        const long dst = orig_row[s->dev2orig[dv]], src = s->h_view_obs[dv];
        const int cnt = s->h_view_count[dv];
        if (residuals) std::memcpy(residuals + 2 * dst, h_res.data() + 2 * src, sizeof(double) * 2 * cnt);
        if (J_cam) std::memcpy(J_cam + 12 * dst, h_Jc.data() + 12 * src, sizeof(double) * 12 * cnt);
        if (J_board) std::memcpy(J_board + 12 * dst, h_Jb.data() + 12 * src, sizeof(double) * 12 * cnt);
        if (J_intr) std::memcpy(J_intr + 18 * dst, h_Ji.data() + 18 * src, sizeof(double) * 18 * cnt);
    }
    for (size_t k = 0; k < N; ++k) c += 0.5 * (h_res[2 * k] * h_res[2 * k] + h_res[2 * k + 1] * h_res[2 * k + 1]);     // device corner order
    if (cost) *cost = c;
    return 0;
}

extern "C" int tscm_eval_normal_equations(const tscm_problem *p, int device, double *board_gram, double *board_grad,
                                          double *view_cross, double *cam_gram, double *cam_grad, double *cost)
{
    tscm_solver *s = nullptr;
    int rc = tscm_solver_create(p, device, &s);
    if (rc) return rc;
    std::unique_ptr<tscm_solver, void (*)(tscm_solver *)> guard(s, tscm_solver_destroy);
    if ((rc = prepare_eval(s))) return rc;
    const DevProblem &P = s->P;
    DevState &S = s->S;
    if ((rc = launch_eval(s, 0))) return rc;
    hipLaunchKernelGGL(k_reduce_stats, dim3(P.C * kCamSl), dim3(256), 0, s->stream, P, S, 0, 0);   // camera blocks only
    hipLaunchKernelGGL(k_finalize_eval, dim3(P.C), dim3(256), 0, s->stream, P, S, 0);            // camera blocks only
    HIP_TRY(hipStreamSynchronize(s->stream));
    HIP_TRY(hipGetLastError());
    std::vector<double> rec((size_t)kRec * s->V), H(256 * (size_t)s->C), cc((size_t)kCStride * s->C);
    if (s->V) HIP_TRY(hipMemcpy(rec.data(), S.rec[0], sizeof(double) * rec.size(), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(H.data(), S.H_stage, sizeof(double) * H.size(), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(cc.data(), S.cconst[0], sizeof(double) * cc.size(), hipMemcpyDeviceToHost));
    if (board_gram) std::memset(board_gram, 0, sizeof(double) * 36 * (size_t)s->B);
    if (board_grad) std::memset(board_grad, 0, sizeof(double) * 6 * (size_t)s->B);
    if (view_cross) std::memset(view_cross, 0, sizeof(double) * 90 * (size_t)p->n_views);
    for (int dv = 0; dv < s->V; ++dv) {
        // the view's record: W = E^T [F | r] (14 columns x 6 rows, column-major), E = E^T E_wb (3 columns x 6 rows); the
        // t_b x t_b block of E^T E follows from the t_c columns of W and the camera rotation (tb_tb), the block above
        // the diagonal by symmetry -- what the device-side consumers do
        const double *rw = rec.data() + (size_t)kRecW * s->h_view_slot[dv];
        const double *re = rec.data() + (size_t)kRecW * s->V + (size_t)kRecE * s->h_view_slot[dv];
        const double *Rc = cc.data() + (size_t)kCStride * s->h_view_cam[dv];
        const int b = s->b0 + s->board_perm[s->h_view_board[dv]], ov = s->dev2orig[dv];
        for (int i = 0; i < 6; ++i) {
            if (board_gram)
                for (int j = 0; j < 6; ++j) {
                    const int hi = std::max(i, j), lo = std::min(i, j);
                    board_gram[36 * (size_t)b + 6 * i + j] += lo < 3 ? re[6 * lo + hi] : tb_tb(rw, Rc, hi - 3, lo - 3);
                }
            if (board_grad) board_grad[6 * (size_t)b + i] += rw[6 * kFR + i];
            if (view_cross) {
                for (int j = 0; j < 13; ++j) view_cross[90 * (size_t)ov + 15 * i + j] = rw[6 * j + i];
            }
        }
    }
    double c = 0.0;
    for (int m = 0; m < s->C; ++m) {
        const double *h = H.data() + 256 * (size_t)m;
        if (cam_gram) { std::memset(cam_gram + 225 * (size_t)m, 0, sizeof(double) * 225); for (int i = 0; i < 13; ++i) for (int j = 0; j < 13; ++j) cam_gram[225 * (size_t)m + 15 * i + j] = h[16 * i + j]; }
        if (cam_grad) { std::memset(cam_grad + 15 * (size_t)m, 0, sizeof(double) * 15); for (int i = 0; i < 13; ++i) cam_grad[15 * (size_t)m + i] = h[16 * i + kFR]; }
        c += 0.5 * h[16 * kFR + kFR];
    }
    if (cost) *cost = c;
    return 0;
}

// device buffer freed on every exit path of the small entry points below
struct DevBuf {
    double *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc(reinterpret_cast<void **>(&p), std::max<size_t>(n, 1) * sizeof(double)); }
};

extern "C" int tscm_project_points(const double *intr9, const double *points, int n, int device, double *pixels)
{
    if (!intr9 || (n > 0 && (!points || !pixels)) || n < 0) return fail(TSCM_E_INVALID, "NULL argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail(TSCM_E_NO_DEVICE, "no usable HIP device");
    HIP_TRY(hipSetDevice(device));
    if (n == 0) return 0;
    DevBuf d_i, d_p, d_o;
    HIP_TRY(d_i.alloc(9));
    HIP_TRY(d_p.alloc(3 * (size_t)n));
    HIP_TRY(d_o.alloc(2 * (size_t)n));
    HIP_TRY(hipMemcpy(d_i.p, intr9, 9 * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_p.p, points, 3 * (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_project, dim3((n + 255) / 256), dim3(256), 0, 0, d_i.p, d_p.p, n, d_o.p);
    HIP_TRY(hipMemcpy(pixels, d_o.p, 2 * (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int tscm_unproject_pixels(const double *intr9, const double *pixels, int n, int device, double *rays)
{
    if (!intr9 || (n > 0 && (!pixels || !rays)) || n < 0) return fail(TSCM_E_INVALID, "NULL argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail(TSCM_E_NO_DEVICE, "no usable HIP device");
    HIP_TRY(hipSetDevice(device));
    if (n == 0) return 0;
    DevBuf d_i, d_p, d_o;
    HIP_TRY(d_i.alloc(9));
    HIP_TRY(d_p.alloc(2 * (size_t)n));
    HIP_TRY(d_o.alloc(3 * (size_t)n));
    HIP_TRY(hipMemcpy(d_i.p, intr9, 9 * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_p.p, pixels, 2 * (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_unproject, dim3((n + 255) / 256), dim3(256), 0, 0, d_i.p, d_p.p, n, d_o.p);
    HIP_TRY(hipMemcpy(rays, d_o.p, 3 * (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int tscm_reprojection_error(const tscm_problem *p, int device, double *per_camera_mean, double *global_mean, double *rmse)
{
    tscm_solver *s = nullptr;
    int rc = tscm_solver_create(p, device, &s);
    if (rc) return rc;
    std::unique_ptr<tscm_solver, void (*)(tscm_solver *)> guard(s, tscm_solver_destroy);
    if ((rc = tscm_solver_upload_params(s, s->h_cam_rt, s->h_intr, s->h_board_rt))) return rc;
    double *d_e = nullptr, *d_q = nullptr;
    if ((rc = dev_alloc(s, &d_e, (size_t)s->V))) return rc;
    if ((rc = dev_alloc(s, &d_q, (size_t)s->V))) return rc;
    if (s->V) hipLaunchKernelGGL(k_reproj_error, dim3(s->V), dim3(64), 0, s->stream, s->P, s->d_init_cam, s->d_init_intr, s->d_init_board, d_e, d_q);
    HIP_TRY(hipStreamSynchronize(s->stream));
    HIP_TRY(hipGetLastError());
    std::vector<double> e(s->V), q(s->V);
    if (s->V) { HIP_TRY(hipMemcpy(e.data(), d_e, sizeof(double) * s->V, hipMemcpyDeviceToHost)); HIP_TRY(hipMemcpy(q.data(), d_q, sizeof(double) * s->V, hipMemcpyDeviceToHost)); }
    std::vector<double> err(s->C, 0.0);
    std::vector<long> cnt(s->C, 0);
    double sq = 0.0;
    for (int v = 0; v < s->V; ++v) { err[s->h_view_cam[v]] += e[v]; cnt[s->h_view_cam[v]] += s->h_view_count[v]; sq += q[v]; }
    double tot = 0.0; long n = 0;
    for (int m = 0; m < s->C; ++m) { tot += err[m]; n += cnt[m]; if (per_camera_mean) per_camera_mean[m] = cnt[m] ? err[m] / (double)cnt[m] : 0.0; }
    if (global_mean) *global_mean = n ? tot / (double)n : 0.0;
    if (rmse) *rmse = n ? std::sqrt(sq / (double)n) : 0.0;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// multi-GPU
// ------------------------------------------------------------------------------------------------
static_assert(sizeof(ncclUniqueId) <= TSCM_UNIQUE_ID_BYTES, "ncclUniqueId larger than the ABI buffer");

extern "C" int tscm_comm_unique_id(unsigned char id[TSCM_UNIQUE_ID_BYTES])
{
    if (!id) return fail(TSCM_E_INVALID, "id is NULL");
    ncclUniqueId u;
    NCCL_TRY(ncclGetUniqueId(&u));
    std::memset(id, 0, TSCM_UNIQUE_ID_BYTES);
    std::memcpy(id, &u, sizeof(u));
    return 0;
}

extern "C" int tscm_comm_create(const unsigned char id[TSCM_UNIQUE_ID_BYTES], int rank, int world, int device, tscm_comm **out)
{
    if (!id || !out || world < 1 || rank < 0 || rank >= world) return fail(TSCM_E_INVALID, "bad communicator arguments");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail(TSCM_E_NO_DEVICE, "no usable HIP device");
    HIP_TRY(hipSetDevice(device));
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    std::unique_ptr<tscm_comm> c(new tscm_comm);
    c->rank = rank; c->world = world; c->device = device;
    NCCL_TRY(ncclCommInitRank(&c->comm, world, u, rank));
    *out = c.release();
    return 0;
}

extern "C" int tscm_comm_create_local(int world, int device, tscm_comm **out)
{
    if (!out || world < 1) return fail(TSCM_E_INVALID, "bad communicator arguments");
    for (int r = 0; r < world; ++r) out[r] = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail(TSCM_E_NO_DEVICE, "no usable HIP device");
    HIP_TRY(hipSetDevice(device));
    std::unique_ptr<tscm_local_group> g(new tscm_local_group);
    g->world = world; g->device = device;
    HIP_TRY(hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking));
    if (hipMalloc(reinterpret_cast<void **>(&g->d_ptrs), sizeof(double *) * 2 * (size_t)world) != hipSuccess) {
        (void)hipStreamDestroy(g->stream);
        return fail(TSCM_E_NOMEM, "hipMalloc of the group's pointer table failed");
    }
    for (int r = 0; r < world; ++r) {
        tscm_comm *c = new tscm_comm;
        c->rank = r; c->world = world; c->device = device; c->group = g.get();
        out[r] = c;
    }
    g->refs = world;
    g.release();
    return 0;
}

struct IpcIdent { int pci[3]; int fine; };
static_assert(sizeof(hipIpcMemHandle_t) + sizeof(IpcIdent) <= TSCM_IPC_HANDLE_BYTES, "hipIpcMemHandle_t + identity larger than the ABI buffer");

extern "C" int tscm_comm_ipc_open(int rank, int world, int device, size_t max_doubles, tscm_comm **out, unsigned char handle[TSCM_IPC_HANDLE_BYTES])
{
    if (!out || !handle || world < 1 || world > kIpcMaxWorld || rank < 0 || rank >= world || max_doubles == 0) return fail(TSCM_E_INVALID, "bad communicator arguments (IPC: up to 16 ranks)");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail(TSCM_E_NO_DEVICE, "no usable HIP device");
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;               // (queried before anything is allocated: an early return below must not leak device memory)
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    std::unique_ptr<tscm_comm> c(new tscm_comm);
    std::unique_ptr<tscm_ipc> x(new tscm_ipc);
    c->rank = rank; c->world = world; c->device = device;
    x->rank = rank; x->world = world; x->max_doubles = (max_doubles + 1) & ~(size_t)1;
    if (hipMalloc(reinterpret_cast<void **>(&x->d_fault), sizeof(int)) != hipSuccess) return fail(TSCM_E_NOMEM, "hipMalloc failed");
    hipIpcMemHandle_t h;
    // fine-grained first (what a peer on ANOTHER device needs for its system-scope flag stores and my loads of them to meet);
    // where that cannot be had or exported, ordinary device memory -- enough for ranks that share this device, and the handle
    // says so: a peer on another device then refuses to connect
    hipError_t e = hipErrorUnknown;
    for (int attempt = 0; attempt < 2 && e != hipSuccess; ++attempt) {
        x->fine = attempt == 0;
        e = x->fine ? hipExtMallocWithFlags(&x->own, x->total_bytes(), hipDeviceMallocFinegrained) : hipMalloc(&x->own, x->total_bytes());
        if (e != hipSuccess) { x->own = nullptr; (void)hipGetLastError(); continue; }
        e = hipMemset(x->own, 0, x->total_bytes());
        if (e == hipSuccess) e = hipMemset(x->d_fault, 0, sizeof(int));
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e == hipSuccess) e = hipIpcGetMemHandle(&h, x->own);
        if (e != hipSuccess) { (void)hipFree(x->own); x->own = nullptr; (void)hipGetLastError(); }
    }
    if (e != hipSuccess) {
        (void)hipFree(x->d_fault);
        return fail(TSCM_E_HIP, std::string("IPC exchange buffer: ") + hipGetErrorString(e) + " (hipIpcGetMemHandle needs HSA_ENABLE_IPC_MODE_LEGACY=0 on hosts whose driver only supports dmabuf IPC)");
    }
    std::memset(handle, 0, TSCM_IPC_HANDLE_BYTES);
    std::memcpy(handle, &h, sizeof(h));
    {
        // behind the HIP handle: which device the buffer lives on (PCI address: ordinals differ between processes) and its kind
        IpcIdent id{};
        id.pci[0] = prop.pciDomainID; id.pci[1] = prop.pciBusID; id.pci[2] = prop.pciDeviceID; id.fine = x->fine ? 1 : 0;
        std::memcpy(handle + sizeof(h), &id, sizeof(id));
    }
    x->mapped[rank] = x->own;
    g_err = x->fine ? "note: IPC exchange buffer in fine-grained device memory" : "note: IPC exchange buffer in ordinary device memory (ranks on this device only)";
    c->ipc = x.release();
    *out = c.release();
    return 0;
}

extern "C" int tscm_comm_ipc_connect(tscm_comm *c, const unsigned char *handles)
{
    if (!c || !c->ipc || !handles) return fail(TSCM_E_INVALID, "not an IPC communicator");
    tscm_ipc *x = c->ipc;
    if (x->connected) return fail(TSCM_E_INVALID, "already connected");
    HIP_TRY(hipSetDevice(c->device));
    IpcIdent me{};
    std::memcpy(&me, handles + (size_t)TSCM_IPC_HANDLE_BYTES * x->rank + sizeof(hipIpcMemHandle_t), sizeof(me));
    for (int r = 0; r < x->world; ++r) {
        if (r == x->rank) continue;
        hipIpcMemHandle_t h;
        IpcIdent id{};
        std::memcpy(&h, handles + (size_t)TSCM_IPC_HANDLE_BYTES * r, sizeof(h));
        std::memcpy(&id, handles + (size_t)TSCM_IPC_HANDLE_BYTES * r + sizeof(h), sizeof(id));
        if (std::memcmp(id.pci, me.pci, sizeof(id.pci)) != 0) {
            // a peer on another device: both buffers fine-grained, and peer access enabled here and now (not lazily)
            if (!id.fine || !x->fine) return fail(TSCM_E_UNSUPPORTED, "IPC exchange across devices needs fine-grained exchange buffers on both ranks (one of them is ordinary device memory): put the ranks on one device or use RCCL");
            int ndev = 0, peer = -1;
            HIP_TRY(hipGetDeviceCount(&ndev));
            for (int d = 0; d < ndev && peer < 0; ++d) {
                hipDeviceProp_t prop;
                HIP_TRY(hipGetDeviceProperties(&prop, d));
                if (prop.pciDomainID == id.pci[0] && prop.pciBusID == id.pci[1] && prop.pciDeviceID == id.pci[2]) peer = d;
            }
            if (peer < 0) return fail(TSCM_E_UNSUPPORTED, "IPC exchange: a peer rank's device is not visible to this process (HIP_VISIBLE_DEVICES): no peer access");
            int can = 0;
            HIP_TRY(hipDeviceCanAccessPeer(&can, c->device, peer));
            if (!can) return fail(TSCM_E_UNSUPPORTED, "IPC exchange: no peer access between this rank's device and a peer rank's");
            const hipError_t pe = hipDeviceEnablePeerAccess(peer, 0);
            if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) { HIP_TRY(pe); }
            (void)hipGetLastError();
        }
        HIP_TRY(hipIpcOpenMemHandle(&x->mapped[r], h, hipIpcMemLazyEnablePeerAccess));
    }
    x->connected = true;
    return 0;
}

extern "C" int tscm_comm_info(const tscm_comm *c, int *rank, int *world, int *backend_ranks)
{
    if (!c) return fail(TSCM_E_INVALID, "communicator is NULL");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    if (backend_ranks) {
        int n = c->group ? c->group->world : c->ipc && c->ipc->connected ? c->ipc->world : 0;
        if (c->comm) NCCL_TRY(ncclCommCount(c->comm, &n));      // what RCCL itself reports for the communicator
        *backend_ranks = n;
    }
    return 0;
}

extern "C" void tscm_comm_destroy(tscm_comm *c)
{
    if (!c) return;
    if (c->comm) (void)ncclCommDestroy(c->comm);
    if (c->ipc) {
        (void)hipSetDevice(c->device);
        (void)hipDeviceSynchronize();
        for (int r = 0; r < c->ipc->world; ++r) if (r != c->ipc->rank && c->ipc->mapped[r]) (void)hipIpcCloseMemHandle(c->ipc->mapped[r]);
        (void)hipFree(c->ipc->own);
        (void)hipFree(c->ipc->d_fault);
        delete c->ipc;
    }
    if (c->group && --c->group->refs == 0) {
        (void)hipSetDevice(c->group->device);
        (void)hipStreamSynchronize(c->group->stream);
        (void)hipFree(c->group->d_ptrs);
        (void)hipStreamDestroy(c->group->stream);
        delete c->group;
    }
    delete c;
}

extern "C" int tscm_shard_frames(const tscm_problem *p, int world, int *owner)
{
    if (!p || !owner || world < 1) return fail(TSCM_E_INVALID, "bad arguments");
    if (int rc = validate(p)) return rc;
    std::vector<int> o;
    shard_owner(p, world, o);
    std::copy(o.begin(), o.end(), owner);
    return 0;
}

#ifdef TSCM_WAVE_TIMELINE
// profiling builds only: the per-wave timeline of the last recorded k_eval_gram launch (tools/wave_timeline.py)
extern "C" int tscm_debug_wave_timeline(long long *out, int max_waves)
{
    const int n = std::min(max_waves, tscm::kTimelineWaves);
    if (hipDeviceSynchronize() != hipSuccess) return TSCM_E_HIP;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(tscm::g_timeline), sizeof(long long) * 4 * (size_t)n) != hipSuccess) return TSCM_E_HIP;
    return n;
}
extern "C" int tscm_debug_kernel_timeline(long long *out, int max_groups)
{
    if (max_groups < tscm::kKtlGroups) return TSCM_E_INVALID;
    if (hipDeviceSynchronize() != hipSuccess) return TSCM_E_HIP;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(tscm::g_ktl), sizeof(long long) * 2 * tscm::kKtlKernels * tscm::kKtlGroups) != hipSuccess) return TSCM_E_HIP;
    return tscm::kKtlKernels;
}
extern "C" int tscm_debug_control_stamps(long long *out)
{
    if (hipDeviceSynchronize() != hipSuccess) return TSCM_E_HIP;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(tscm::g_ktlx), sizeof(long long) * 32) != hipSuccess) return TSCM_E_HIP;
    return 32;
}
extern "C" int tscm_debug_phase_stamps(long long *out, int max_groups)
{
    if (max_groups < tscm::kKtlGroups) return TSCM_E_INVALID;
    if (hipDeviceSynchronize() != hipSuccess) return TSCM_E_HIP;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(tscm::g_phs), sizeof(long long) * 3 * tscm::kPhStamps * tscm::kKtlGroups) != hipSuccess) return TSCM_E_HIP;
    return tscm::kPhStamps;
}
extern "C" int tscm_debug_wave_views(long long *out, int max_waves)
{
    const int n = std::min(max_waves, tscm::kTimelineWaves);
    if (hipDeviceSynchronize() != hipSuccess) return TSCM_E_HIP;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(tscm::g_tlv), sizeof(long long) * (4 + tscm::kTlViews) * (size_t)n) != hipSuccess) return TSCM_E_HIP;
    return 4 + tscm::kTlViews;
}
extern "C" int tscm_debug_wave_phases(long long *out, int max_waves)
{
    const int n = std::min(max_waves, tscm::kTimelineWaves);
    if (hipDeviceSynchronize() != hipSuccess) return TSCM_E_HIP;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(tscm::g_phase), sizeof(long long) * 5 * (size_t)n) != hipSuccess) return TSCM_E_HIP;
    return n;
}
#endif

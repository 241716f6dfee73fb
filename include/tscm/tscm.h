/*
 * tscm.h -- C ABI of the MI355X-native Triple-Sphere reprojection-error LM solver.
 *
 * Drop-in boundary for the hot path of imuncle/TSCM_Calib:
 * plain pointers and sizes, caller-owned parameter arrays updated in place -- the same
 * contract the reference has with Ceres, which identifies parameter blocks by pointer
 * (`intrinsic_.data()`, `rt_[i].data()`: TS.cpp:266-267, multi_calib.cpp:182-184).
 * Every entry point names the reference interface it replaces.  INTEGRATION.md shows
 * the reference-side binding (the body of MultiCalib::calibrate() /
 * TripleSphereCamera::refinement() rewritten onto these calls).
 *
 * All entry points return 0 on success or a negative TSCM_E_* code;
 * tscm_last_error() returns a thread-local description of the last failure.
 * The library never falls back to a CPU path: without a usable HIP device every
 * compute entry point fails with TSCM_E_NO_DEVICE.
 */
#ifndef TSCM_H
#define TSCM_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TSCM_ABI_VERSION 6   /* 2: tscm_problem.board_pose_constant; 3: tscm_options.exec_flags (both appended;  */
                             /*    until ABI 5 a zero-initialised options struct kept its meaning -- from ABI 6   */
                             /*    a zeroed struct has struct_size 0 and is REFUSED); 4: unknown exec_flags bits are  */
                             /*    refused, TSCM_EXEC_DENSE_REDUCED_ORDER, the fault injection of the tests is   */
                             /*    an entry point of its own (tscm_solver_debug_withhold_handoff), no option;    */
                             /* 5: tscm_comm_ipc_open / tscm_comm_ipc_connect, tscm_device_peak_fp32_mfma,       */
                             /*    TSCM_EXEC_GRAPH_REDUCED_ORDER (no struct changed);                            */
                             /* 6: tscm_options starts with struct_size (the third appended field in four rounds */
                             /*    was the moment): the library reads only as many bytes as the caller's struct  */
                             /*    has and refuses a size it does not know -- an options struct of ABI <= 5 is   */
                             /*    refused instead of misread; TSCM_E_PEER; a late device-side hand-off re-runs  */
                             /*    the solve on separate launches before it is an error; TSCM_EXEC_SEPARATE_STATS */
                             /*    (no struct changed; a library without the bit refuses it); round 6, still 6:  */
                             /*    the rank-divergence guard of the sharded solver and its test hook             */
                             /*    tscm_solver_debug_perturb_exchange (an entry point more, no struct changed)   */

enum {
    TSCM_OK = 0,
    TSCM_E_INVALID = -1,      /* bad argument / inconsistent problem description      */
    TSCM_E_NO_DEVICE = -2,    /* no HIP device, or hipSetDevice failed                */
    TSCM_E_HIP = -3,          /* a HIP runtime call failed                            */
    TSCM_E_RCCL = -4,         /* an RCCL call failed (the communicator is aborted)    */
    TSCM_E_UNSUPPORTED = -5,  /* problem shape outside what the kernels support       */
    TSCM_E_NOMEM = -6,
    TSCM_E_PEER = -7          /* the IPC exchange back-end: a peer rank did not arrive within its time bound, or its */
                              /* memory could not be mapped; any back-end: the ranks disagree about the LM state     */
                              /* (rank-divergence guard).  The communicator is unusable afterwards, like an aborted  */
                              /* RCCL communicator                                                                   */
};

/* ceres::TerminationType values the reference looks at (TS.cpp:281). */
enum { TSCM_CONVERGENCE = 0, TSCM_NO_CONVERGENCE = 1, TSCM_FAILURE = 2 };

#define TSCM_MAX_ITERATIONS 255

/*
 * Problem = what multi_calib.cpp:157-207 / TS.cpp:249-269 feed to ceres::Problem.
 *
 * A "view" is one (camera m, board i) pair with detected corners, i.e. a non-empty
 * cameras_[m].pixels()[i] with chessboards_[i].is_initial() (multi_calib.cpp:164-169);
 * corner j of the view is paired with board point worlds_[j] (:176).  Mono: a view is
 * an image with has_chessboard_[i] (TS.cpp:253), camera index 0, board index = image.
 *
 * Parameter layout is the reference's: intrinsics `fx fy cx cy xi lambda alpha b c`
 * (TS.h:105, multi_calib.h:22), poses = angle-axis (rad) then translation in board
 * units (TS.cpp:72, multi_calib.h:17).  b and c are inert (their terms are commented
 * out in both functors: TS.h:122-123, multi_calib.h:175-176) and are returned unchanged.
 */
typedef struct tscm_problem {
    int n_cameras;                 /* C                                                */
    int n_boards;                  /* B  (mono: number of images)                      */
    int n_points;                  /* board corners, e.g. 54 or 88                     */
    int n_views;
    const double *board_xy;        /* [n_points*2]  worlds_[j].x, .y (z forced to 0:   */
                                   /*   TS.h:107-109, multi_calib.h:154-156)           */
    const int *view_camera;        /* [n_views]                                        */
    const int *view_board;         /* [n_views]                                        */
    const int *view_offset;        /* [n_views] first corner of the view in obs_u/v    */
    const int *view_count;         /* [n_views] pixels[i].size()  (0..n_points)        */
    const double *obs_u;           /* [N] observed pixel x, SoA                        */
    const double *obs_v;           /* [N] observed pixel y                             */
    double *cam_rt;                /* [C*6] cameras_[m].rt_        in/out (host)       */
    double *intr;                  /* [C*9] cameras_[m].intrinsic_ in/out (host)       */
    double *board_rt;              /* [B*6] chessboards_[i].rt_ / rt_[i]  in/out (host)*/
    const unsigned char *cam_pose_constant; /* [C] SetParameterBlockConstant           */
                                   /*   (multi_calib.cpp:186: camera 0); NULL = none   */
    int mono;                      /* 1: TS.h functor -- no camera pose block at all   */
    const unsigned char *board_pose_constant; /* [B] 1 = the board's pose block is held    */
                                   /*   constant (problem.SetParameterBlockConstant on */
                                   /*   chessboards_[i].rt_ / rt_[i]); NULL = none.    */
                                   /*   The reference never does this; it is the       */
                                   /*   "intrinsics-only" form of BASELINE config 2    */
                                   /*   (all views fixed: 7 free intrinsics remain).   */
} tscm_problem;

/* ceres::Solver::Options fields the path depends on, Ceres defaults
 * (TS.cpp:271-274, multi_calib.cpp:209-212; linear solver is always DENSE_SCHUR). */
typedef struct tscm_options {
    size_t struct_size;                  /* sizeof(tscm_options) of the CALLER's header */
                                         /* (tscm_default_options sets it).  Fields     */
                                         /* behind it that the caller's struct does not */
                                         /* have take their defaults; 0 or an unknown   */
                                         /* size: TSCM_E_INVALID                        */
    int max_num_iterations;              /* 100 mono (TS.cpp:274) / 50 multi           */
    double function_tolerance;           /* 1e-6                                       */
    double gradient_tolerance;           /* 1e-10                                      */
    double parameter_tolerance;          /* 1e-8                                       */
    double initial_trust_region_radius;  /* 1e4                                        */
    double max_trust_region_radius;      /* 1e16                                       */
    double min_trust_region_radius;      /* 1e-32                                      */
    double min_relative_decrease;        /* 1e-3                                       */
    double min_lm_diagonal;              /* 1e-6                                       */
    double max_lm_diagonal;              /* 1e32                                       */
    int max_num_consecutive_invalid_steps; /* 5                                        */
    int jacobi_scaling;                  /* 1                                          */
    int check_every;                     /* host polls the device-resident LM loop     */
                                         /* every this many iterations (default 4)     */
    int jacobian_fp32;                   /* 0 (default): everything fp64, 1e-6 tier.   */
                                         /* 1: derivatives and the J^T J contraction in */
                                         /* fp32 (packed VALU + fp32 MFMA), projection, */
                                         /* residuals, cost and the whole linear solve  */
                                         /* stay fp64: north_star's 1e-3 tier          */
    int exec_flags;                      /* 0 (default).  Execution variants that do   */
                                         /* not change the mathematics (TSCM_EXEC_*)   */
} tscm_options;

/* tscm_options.exec_flags.  They select code paths for A/B runs and for tests that a one-GPU box would otherwise never
 * run; none of them changes the mathematics, all but _DENSE_ / _GRAPH_REDUCED_ORDER give the same bits.  Bits outside TSCM_EXEC_ALL are
 * refused with TSCM_E_INVALID (a caller built against an options struct without the field passes garbage here). */
enum {
    TSCM_EXEC_SEPARATE_T_REDUCE = 1,       /* keep the Schur-complement tile reduction a launch of its own instead of   */
                                           /* riding in the reduced solve's launch (what every communicator run does)   */
    TSCM_EXEC_KEEP_SINGLE_RANK_COMM = 2,   /* run the communicator code path (two all-reduces, separate control step)   */
                                           /* even when the attached communicator has one rank                          */
    TSCM_EXEC_GRAM_16X16 = 4,              /* the Gram contraction of the dominant kernel on v_mfma_f64_16x16x4 (one tile,  */
                                           /* rounds 1-3a) instead of three v_mfma_f64_4x4x4_4b per four rows (A/B runs)   */
    TSCM_EXEC_SEPARATE_BACKSUB = 8,        /* keep the back-substitution a launch of its own instead of workgroups that   */
                                           /* wait for the camera step inside the reduced solve's launch (one GPU)        */
    TSCM_EXEC_SEPARATE_CONTROL = 16,       /* take the LM control step in the reductions' launch (k_reduce_control) instead  */
                                           /* of in the head of the next Schur-complement kernel (one GPU)                    */
    TSCM_EXEC_DENSE_REDUCED_ORDER = 32,    /* k_solve_nd (rigs of 5-8 cameras) factors the reduced camera system as ONE dense  */
                                           /* block instead of along the camera-pair graph (same solution to rounding)       */
    TSCM_EXEC_GRAPH_REDUCED_ORDER = 64,    /* rigs of up to 4 cameras: k_solve_nd along the camera-pair graph instead of the  */
                                           /* dense k_solve_reduced (which is faster there: tests and A/B runs)               */
    TSCM_EXEC_SEPARATE_STATS = 128,        /* keep the reductions behind a candidate's evaluation (k_reduce_stats) a launch of  */
                                           /* their own instead of the first workgroups of the next Schur-complement launch    */
    TSCM_EXEC_MFMA_REDUCED_SOLVE = 256,    /* rigs of up to 4 cameras: the reduced camera system factored by ONE wave with rank-4  */
                                           /* updates on the matrix cores (six 16 x 16 accumulator tiles) instead of 256 threads on  */
                                           /* 4 x 4 register tiles with a barrier per panel.  Round 6: built, the SAME bits, 1 us     */
                                           /* slower at config 4 -- an experiment switch, not the default (HISTORY A.7)              */
    TSCM_EXEC_ONE_VIEW_PER_PASS = 512,     /* boards of up to 32 corners: one view per pass of the Gram kernel (k_eval_gram4) instead of  */
                                           /* several views sharing a pass (k_eval_gram4p, round 6): same bits, for tests and A/B runs     */
    TSCM_EXEC_ALL = 1023
};

/* ceres::IterationSummary subset */
typedef struct tscm_iteration {
    int iteration;
    int step_is_valid;
    int step_is_successful;
    double cost;
    double cost_change;
    double gradient_max_norm;
    double gradient_norm;
    double step_norm;
    double relative_decrease;
    double trust_region_radius;
} tscm_iteration;

/* ceres::Solver::Summary subset (what BriefReport prints: TS.cpp:280, multi_calib.cpp:218)
 * plus the error report of multi_calib.cpp:233-283. */
typedef struct tscm_summary {
    int termination_type;          /* TSCM_CONVERGENCE / NO_CONVERGENCE / FAILURE      */
    int num_iterations;            /* entries in iterations[] (iteration 0 included)   */
    int num_successful_steps;
    int num_unsuccessful_steps;
    double initial_cost;
    double final_cost;
    int n_residual_blocks;         /* corners in the program                           */
    int lm_iterations;             /* LM iterations executed (incl. a final one that   */
                                   /*   ended on a tolerance test)                     */
    tscm_iteration iterations[TSCM_MAX_ITERATIONS + 1];
    char message[128];
    double seconds_solve;          /* minimiser loop on the device: first to last kernel */
    double seconds_total;          /* wall time of the call (tscm_solve_*: incl. upload/download) */
    double rmse;                   /* sqrt(2*final_cost/N)                             */
} tscm_summary;

typedef struct tscm_solver tscm_solver;   /* opaque: device buffers, stream, layouts  */
typedef struct tscm_comm tscm_comm;       /* opaque: RCCL communicator                */

int tscm_abi_version(void);
const char *tscm_last_error(void);
int tscm_device_count(void);
/* hipSetDevice(device) + hipDeviceSynchronize(): lets a host (e.g. the multi-process benchmark) fence the
 * GPU without loading a second HIP runtime of its own. */
int tscm_device_synchronize(int device);
/* Measured fp64 ceilings of the device, every CU busy: v_mfma_f64_16x16x4_f64 and v_fma_f64 throughput in TFLOP/s
 * (a few milliseconds of micro-kernels; benchmark / roofline reporting only, not on any solver path). */
int tscm_device_peak_fp64(int device, double *mfma_tflops, double *valu_tflops);
/* ... with the two fp64 matrix instructions apart: peaks[0] = v_mfma_f64_16x16x4_f64 (the instruction of rounds 1-3a: about
 * half the datasheet rate on gfx950), peaks[1] = v_mfma_f64_4x4x4_4b_f64 (the one the dominant kernel uses since round 3:
 * the datasheet rate), peaks[2] = v_fma_f64; TFLOP/s. */
int tscm_device_peak_fp64_ex(int device, double peaks[3]);
/* ... and of v_mfma_f32_16x16x4_f32, the contraction of the fp32-Jacobian tier (tscm_options.jacobian_fp32), TFLOP/s */
int tscm_device_peak_fp32_mfma(int device, double *tflops);

void tscm_default_options(tscm_options *opt, int mono);

/* ------------------------------------------------------------------ solver handle
 * tscm_solver_create  uploads observations and builds the device layout once
 *                     (replaces the per-corner `new ReprojectionError` +
 *                     AddResidualBlock loops: TS.cpp:251-269, multi_calib.cpp:162-207).
 * tscm_solver_solve   = ceres::Solve(options, &problem, &summary)
 *                     (TS.cpp:278, multi_calib.cpp:216).  Reads the initial parameters
 *                     from the problem's cam_rt / intr / board_rt host arrays and
 *                     overwrites them with the result, like Ceres does.
 * Shapes served (anything else: TSCM_E_UNSUPPORTED, nothing is approximated): up to 32 cameras (1..8: reduced
 * system solved in registers/LDS, 9..32: blocked factorisation in global memory), any board shape whose corner
 * list fits the 160 KiB LDS tile (several thousand corners), up to 3.7 M views / 536 M corners per GPU.
 */
int tscm_solver_create(const tscm_problem *problem, int device, tscm_solver **out);
int tscm_solver_set_comm(tscm_solver *s, tscm_comm *comm);   /* frame-sharded multi-GPU, see below */
/* Where the wall time of this solver's creation went, seconds: out[0] runtime_init (device selection, stream, device properties --
 * the first call of a process pays HIP's initialisation here), [1] host_layout (view / board orders, chunk tables, Schur work
 * lists), [2] gather (the observations into device view order), [3] h2d (device allocations and uploads), [4] kernel_setup
 * (occupancy queries, function attributes, elimination plans, final synchronisation).  What one call of the reference's
 * MultiCalib::calibrate() spends before ceres::Solve: the problem build of multi_calib.cpp:157-207. */
int tscm_solver_create_timing(const tscm_solver *s, double out[5]);
/* TESTS ONLY (host code, no GPU needed): the device orders tscm_solver_create_sharded derives for rank `rank` of `world` --
 * *n_views = its views with corners, dev2orig[n_views] = the caller's view index of every device view (camera-major, a camera's
 * views by device board), board_perm[its boards] = the caller's board (relative to *b0) that becomes device board k (boards
 * grouped by camera-set signature, unseen boards last).  Outputs other than n_views may be NULL. */
int tscm_debug_layout_order(const tscm_problem *p, int rank, int world, int *n_views, int *dev2orig, int *board_perm, int *b0);
/* TESTS / A-B tools only: experiment switches of the layout the NEXT tscm_solver_create builds (process-wide; both were built, are
 * bit-identical to the default and measured slower in round 6 -- HISTORY A.7): value 0 = off (the default). */
enum {
    TSCM_EXPERIMENT_SCHUR_CHUNK_32 = 0,   /* Schur-complement chunks of 32 boards, k_schur_gram<NV, false, 32> at three workgroups per CU */
    TSCM_EXPERIMENT_GRAM_STREAM = 1,      /* k_eval_gram4s: the views of a chunk as one stream of k-steps (boards of >= 33 corners)      */
    TSCM_EXPERIMENT_COUNT = 2
};
int tscm_debug_experiment(int which, int value);
/* TESTS / bench.py (host code): the pass plan of the Gram kernels for a board of n_points corners -- passes per view, corners per
 * pass (a multiple of four), k-steps per pass (the kernels' template parameter KS <= 16) and, for boards of up to 32 corners, the
 * views that share a pass.  The reference's 11 x 8 board: 2 passes of 44 corners, KS = 11. */
int tscm_debug_gram_plan(int n_points, int *passes, int *corners_per_pass, int *k_steps, int *views_per_pass);
/* A hand-off between workgroups of one launch (evaluation's reductions -> control step; Schur-complement tiles -> reduced
 * solve -> back-substitution) that does not
 * come within its time bound (0.5 s: a debugger, a co-tenant, a context switch -- or a fault) stops the solve on the
 * device; the library then runs THAT solve again from its start point on the launches that hand nothing over inside a
 * launch (TSCM_EXEC_SEPARATE_T_REDUCE | _BACKSUB | _CONTROL, which implies _STATS: same mathematics, same bits) and returns its result;
 * tscm_last_error() carries a note, tscm_solver_reruns() counts them.  TSCM_E_HIP only if the re-run fails too -- or
 * with a communicator of several ranks, which would have to agree on it.
 * TESTS ONLY: tscm_solver_debug_withhold_handoff(s, 1): in the next solve of `s` one producer of the hand-off never
 * reports in (the solve must come back re-run, within seconds); (s, 2): ... and the re-run is forbidden: that solve
 * must end with TSCM_E_HIP within the time bound, the caller's parameters untouched, the solver usable afterwards;
 * (s, 3): like 1 for the other hand-off of an iteration -- one of the reductions behind the evaluation that ride in
 * the Schur-complement launch never counts itself in. */
int tscm_solver_debug_withhold_handoff(tscm_solver *s, int on);
int tscm_solver_reruns(const tscm_solver *s);                /* solves of `s` that were run again so far (>= 0) */
/* Rank-divergence guard (ABI 6, round 6; SURVEY 8e: "reductions must be order-deterministic ... so all replicas take the same
 * accept/reject decision" -- the reference itself is single-threaded, multi_calib.cpp:209-212).  The ranks of a communicator run
 * the reduced solve and the control step redundantly on all-reduced data; every rank folds its LM state and the replicated
 * results of its reduced solve into a 52-bit word that travels in the all-reduce of the evaluation (no extra collective, no
 * extra launch), and the control step behind it compares all ranks' words: a mismatch stops the solve on every rank in the
 * same step with TSCM_E_PEER ("ranks disagree at iteration k"); the communicator is unusable afterwards.
 * TESTS ONLY: tscm_solver_debug_perturb_exchange(s, k, ulps): in the next solve of `s` the copy of the Schur-complement tiles
 * this rank RECEIVES from the all-reduce of LM iteration k (1-based; 0: off) is moved by `ulps` units in the last place. */
int tscm_solver_debug_perturb_exchange(tscm_solver *s, int iteration, int ulps);
int tscm_solver_solve(tscm_solver *s, const tscm_options *opt, tscm_summary *summary);
/* Same as _solve but parameters start from / are left in device memory (used by the
 * benchmark to time the minimiser loop with inputs resident in HBM). reset=1 reloads
 * the parameters that were uploaded by the last tscm_solver_upload_params. */
int tscm_solver_upload_params(tscm_solver *s, const double *cam_rt, const double *intr, const double *board_rt);
int tscm_solver_solve_resident(tscm_solver *s, const tscm_options *opt, tscm_summary *summary, int reset);
int tscm_solver_download_params(tscm_solver *s, double *cam_rt, double *intr, double *board_rt);
void tscm_solver_destroy(tscm_solver *s);
/* timing of the dominant kernel (k_eval_gram) by HIP events on the solver's own stream, accumulated since the last
 * call: returns the number of timed launches and their total milliseconds, and (re)arms the timers --
 * enable = 0: off; n >= 1: bracket every n-th launch (an event pair delays the stream by a few microseconds, so a
 * benchmark samples instead of timing every launch). */
int tscm_solver_kernel_time(tscm_solver *s, int enable, int *launches, double *total_ms);
/* ... and of the two per-iteration exchanges of a sharded solve (the all-reduce of the Schur-complement tiles T and of
 * the staged camera tiles H_stage; the reference has no counterpart -- SURVEY 5: "measure it separately from compute"),
 * sampled at the same rate while tscm_solver_kernel_time has the timers armed: number of timed collectives and their
 * total milliseconds since the last call.  Zeros without a communicator. */
int tscm_solver_exchange_time(tscm_solver *s, int *n_T, double *ms_T, int *n_H, double *ms_H);

/* One-shot drop-ins ------------------------------------------------------------
 * tscm_solve_multi replaces the Ceres block of MultiCalib::calibrate()
 *                  (multi_calib.cpp:157-218); the caller then runs update_param()
 *                  (multi_calib.cpp:221-232) on its own objects.
 * tscm_solve_mono  replaces TripleSphereCamera::refinement (TS.cpp:247-282); the
 *                  reference's `return summary.termination_type == CONVERGENCE`
 *                  is `summary->termination_type == TSCM_CONVERGENCE`.             */
int tscm_solve_multi(const tscm_problem *problem, const tscm_options *opt, tscm_summary *summary);
int tscm_solve_mono(const tscm_problem *problem, const tscm_options *opt, tscm_summary *summary);

/* ------------------------------------------------------------------ operator level
 * Batched cost-functor evaluation = ceres::CostFunction::Evaluate of
 * AutoDiffCostFunction<ReprojectionError,2,6,6,9> (multi_calib.h:138-199) /
 * <...,2,9,6> (TS.h:93-134) for every corner of the problem, on the GPU, with the
 * analytic Jacobian.  Corner order = views in problem order, corners in order.
 * residuals [N*2]; J_cam [N*2*6], J_board [N*2*6], J_intr [N*2*9] row-major per corner
 * (any of the three may be NULL).  cost = 0.5*sum r^2.  Host pointers.            */
int tscm_eval_functor(const tscm_problem *problem, int device, double *residuals,
                      double *J_cam, double *J_board, double *J_intr, double *cost);

/* Schur-form normal equations at the problem's current parameters (the quantities
 * Ceres' SchurEliminator forms from the Jacobian), for parity tests.  Outputs (host,
 * any may be NULL):  board_gram [B*36] = sum E^T E ; board_grad [B*6] = E^T r ;
 * view_cross [n_views*6*15] = E^T [F_campose(6) F_intr(9)] per view ;
 * cam_gram [C*15*15] = F^T F per camera ; cam_grad [C*15] = F^T r ; cost.
 * All UNSCALED (no Jacobi scaling, no damping).                                    */
int tscm_eval_normal_equations(const tscm_problem *problem, int device, double *board_gram,
                               double *board_grad, double *view_cross, double *cam_gram,
                               double *cam_grad, double *cost);

/* ------------------------------------------------------------------ projection family
 * tscm_project_points   = TripleSphereCamera::project (TS.cpp:332-344), skew terms
 *                         included, n camera-frame points [n*3] -> pixels [n*2].
 * tscm_unproject_pixels = get_unit_sphere_coordinate (TS.h:39-57) with
 *                         transform = identity, pixels [n*2] -> unit rays [n*3].
 * tscm_reprojection_error = the report of multi_calib.cpp:233-283 / main.cpp:245-288:
 *                         per-camera mean Euclidean pixel error [C] (may be NULL),
 *                         global mean, and RMSE over all corners.                  */
int tscm_project_points(const double *intr9, const double *points, int n, int device, double *pixels);
int tscm_unproject_pixels(const double *intr9, const double *pixels, int n, int device, double *rays);
int tscm_reprojection_error(const tscm_problem *problem, int device, double *per_camera_mean,
                            double *global_mean, double *rmse);

/* ------------------------------------------------------------------ multi-GPU (frame sharding)
 * The reference has no counterpart (multi_calib.cpp:209-212 never sets num_threads); this is north_star's
 * "observations shard by image across the GPUs of one node with an RCCL all-reduce of J^T J / J^T r".
 *
 * Every rank passes the SAME whole problem to tscm_solver_create_sharded(problem, device, rank, world): the library
 * assigns contiguous, corner-balanced ranges of boards (frames) to the ranks (tscm_shard_frames) and keeps on each
 * GPU only the observations, Schur records and pose blocks of the boards that rank owns, so every board's 6x6 block
 * is rank-local; camera poses and intrinsics are replicated.  What all ranks must agree on -- which cameras have
 * views, which camera pairs share a board, the total corner count -- is derived from the whole problem.
 * Per LM iteration the ranks exchange exactly two buffers with a sum all-reduce: the Schur-complement tiles
 * (256 doubles per camera pair that shares a board) and the camera tiles + scalars (256 C + 8 + world doubles);
 * every rank then solves the reduced camera system redundantly and takes identical accept / reject decisions.
 *
 * Two exchange back-ends:
 *   RCCL   one process per GPU (the production path).  Rank 0 calls tscm_comm_unique_id and distributes the 128
 *          bytes (a socket, a file, MPI ...); every rank calls tscm_comm_create, tscm_solver_set_comm, and then
 *          tscm_solver_solve / _solve_resident like on one GPU.
 *   LOCAL  all ranks in ONE process on ONE device (tscm_comm_create_local + tscm_solver_solve_group): the shards
 *          run in lock step on one stream and the all-reduce is a kernel.  RCCL refuses two ranks on one device,
 *          so this is how the sharded solver is exercised on a single-GPU machine; results are those of the RCCL
 *          path with the same world size (same shards, same summation order: rank order).
 * tscm_solver_download_params writes only the owned boards into board_rt; tscm_solver_gather_boards completes the
 * caller's full-length array on every rank (tscm_solver_solve does both).                                        */
#define TSCM_UNIQUE_ID_BYTES 128
int tscm_solver_create_sharded(const tscm_problem *problem, int device, int rank, int world, tscm_solver **out);
int tscm_comm_unique_id(unsigned char id[TSCM_UNIQUE_ID_BYTES]);
int tscm_comm_create(const unsigned char id[TSCM_UNIQUE_ID_BYTES], int rank, int world, int device, tscm_comm **out);
int tscm_comm_create_local(int world, int device, tscm_comm **out /* [world] */);
/* IPC  one process per rank like RCCL; the exchange is the library's own one-shot all-reduce over buffers the ranks map
 *      from each other (hipIpcMemHandle).  Ranks MAY share a device (RCCL refuses that): the multi-process path on a
 *      one-GPU box.  Every rank calls tscm_comm_ipc_open (max_doubles >= 256 * max(camera-pair blocks, cameras) + 8 +
 *      2 world: api.Comm.ipc computes it; a longer buffer travels in pieces), the TSCM_IPC_HANDLE_BYTES-byte handles are
 *      all-gathered by the caller (socket, file, MPI ...),
 *      every rank calls tscm_comm_ipc_connect with all of them in rank order; then tscm_solver_set_comm as with RCCL.
 *      Exercised between processes on one device; RCCL is the production path across devices. */
#define TSCM_IPC_HANDLE_BYTES 80      /* (64 until ABI 5) the HIP handle, then the device's PCI address and the buffer's kind */
int tscm_comm_ipc_open(int rank, int world, int device, size_t max_doubles, tscm_comm **out, unsigned char handle[TSCM_IPC_HANDLE_BYTES]);
int tscm_comm_ipc_connect(tscm_comm *c, const unsigned char *handles /* [world][TSCM_IPC_HANDLE_BYTES] */);
void tscm_comm_destroy(tscm_comm *c);
/* rank / world the communicator was created with and the number of ranks the back-end itself reports
 * (ncclCommCount for RCCL, the group size for LOCAL); any output may be NULL. */
int tscm_comm_info(const tscm_comm *c, int *rank, int *world, int *backend_ranks);
/* solvers[r] = shard r of n with the r-th communicator of one tscm_comm_create_local call; summaries [n]. */
int tscm_solver_solve_group(tscm_solver **solvers, int n, const tscm_options *opt, tscm_summary *summaries, int reset);
int tscm_solver_gather_boards(tscm_solver *s, double *board_rt);
/* owner[b] = rank owning board b: contiguous ranges balanced by corner count.       */
int tscm_shard_frames(const tscm_problem *problem, int world, int *owner);


/* ------------------------------------------------------------------ rig initialisation
 * tscm_rig_init = the constructor MultiCalib::MultiCalib(cameras, worlds) (multi_calib.cpp:6-153),
 * the step right before calibrate(): camera i is chained to camera i-1 through every board both
 * see; each common board gives a pose hypothesis and the one with the smallest summed reprojection
 * error over ALL common boards and both cameras wins (:50-85 -- quadratic in the number of common
 * boards, 2.7e9 projections per camera pair at 10k views/camera: the GPU part); every board pose is
 * then chosen the same way among the cameras that see it (:90-151).  Semantics kept: Rt_to_R_t
 * builds R from float32 copies of r1, r2 and their float cross product (multi_calib.h:130-137),
 * TripleSphereCamera::ReprojectError is the SUM of pixel errors with the skew projection
 * (TS.h:58-69), strict `<` keeps the first minimum, rt_ = [cv::Rodrigues(R), t]
 * (multi_calib.h:16-18, 94-96).  The error sums are reduced in a different (tree) order than the
 * reference's sequential loop, so exact ties / 1-ulp near-ties may pick another hypothesis.
 * Adjacent cameras without a common board (undefined behaviour in the reference, :51/:86) ->
 * TSCM_E_INVALID.                                                                             */
typedef struct tscm_rig_input {
    int n_cameras, n_boards, n_points;
    const double *worlds;          /* [n_points*3] board points x, y, z                          */
    const double *intr;            /* [C*9]                                                      */
    const unsigned char *has;      /* [C*B] has_chessboard(j) of camera m                        */
    const double *Rt;              /* [C*B*9] row-major 3x3 [r1 r2 t] = TripleSphereCamera::Rt(j) */
    const double *pix_u, *pix_v;   /* [C*B*n_points] pixels()[j] (read only where has)           */
} tscm_rig_input;

typedef struct tscm_rig_result {
    double *cam_R, *cam_t, *cam_rt;        /* [C*9] row-major, [C*3], [C*6]  (host, caller-owned) */
    double *board_R, *board_t, *board_rt;  /* [B*9], [B*3], [B*6]                                 */
    unsigned char *board_initial;          /* [B] is_initial()                                     */
    int *cam_choice;                       /* [C] winning hypothesis (index among the common boards) or NULL */
    double *cam_min_error;                 /* [C] its summed error, or NULL                        */
    double seconds_hypotheses;             /* device time of the camera-chaining kernels           */
    double seconds_total;
    long long n_projections;               /* point projections evaluated on the device            */
} tscm_rig_result;

int tscm_rig_init(const tscm_rig_input *in, int device, tscm_rig_result *out);


/* ------------------------------------------------------------------ result I/O
 * The YAML main.cpp:305-319 writes through cv::FileStorage -- "cam{i}": 1x9 intrinsic_matrix_,
 * "Twc{i}": 3x4 [R | t] -- and EpipolarRectify/rectify.cpp:262-270 reads.  Host-only (no device).
 * tscm_yaml_format   renders the file into buf (NULL buf: only *needed, incl. the final NUL).
 * tscm_yaml_write    = FileStorage(path, WRITE) + the loop of main.cpp:306-318.
 * tscm_yaml_parse / tscm_yaml_read  = FileStorage(path, READ)["cam{i}"], ["Twc{i}"]: fills
 *                    intr [max_cameras*9] and Twc [max_cameras*12, row-major 3x4] (either may be
 *                    NULL) and returns the number of cameras found in *n_cameras.             */
int tscm_yaml_format(int n_cameras, const double *intr, const double *cam_R, const double *cam_t,
                     char *buf, size_t buf_size, size_t *needed);
int tscm_yaml_write(const char *path, int n_cameras, const double *intr, const double *cam_R,
                    const double *cam_t);
int tscm_yaml_parse(const char *text, int max_cameras, int *n_cameras, double *intr, double *Twc);
int tscm_yaml_read(const char *path, int max_cameras, int *n_cameras, double *intr, double *Twc);


/* ------------------------------------------------------------------ remap tables
 * The per-pixel map builders TripleSphereCamera::undistort (TS.cpp:284-306), the map part of
 * undistort_chessboard (TS.cpp:308-330, before cv::remap) and the eight 400x400 tables of
 * EpipolarRectify/rectify.cpp:86-199 are one loop:
 *     ray = R * ((j - cx)/fx, (i - cy)/fy, 1);  (u, v) = project(ray)  [TS.cpp:332-344, skew terms]
 *     mapx(i, j) = (float)(u + offset_x);  mapy(i, j) = (float)(v + offset_y)
 *   undistort:             R = I, (fx, fy, cx, cy) = the pinhole arguments
 *   undistort_chessboard:  R = Rt_[index] (3x3 [r1 r2 t]), fx = fy = 1, cx = cy = chessboard_size
 *   rectify init_remap:    R = R_cam^T * R_pair, fx = fy = cx = cy = 200, offsets = the mosaic
 *                          origin of the sampled camera (0 | 1280, 0 | 1080); its TScamera::project
 *                          returns (-1, -1) when Z <= -w2 * d1, w2 = 0.42399 (rectify.cpp:7,27):
 *                          check_w2 = 1
 * A batch of maps is one launch; map m writes rows of `out_stride` floats starting at element
 * `out_offset` of mapx / mapy (so the 400x1600 left/right tables of rectify.cpp are 4 maps each).
 * exact != 0: IEEE sqrt / divide and unfused multiply-add in the reference's operation order
 * (bit-identical floats); exact == 0: hardware reciprocal / rsqrt seeds with one correction step
 * (fp64 values within ~2 ulp, i.e. identical floats except when within 1e-15 relative of a
 * float32 rounding boundary).                                                                */
typedef struct tscm_map_desc {
    double intr[9];              /* fx fy cx cy xi lambda alpha b c of the camera that is sampled */
    double R[9];                 /* row-major 3x3                                                  */
    double fx, fy, cx, cy;       /* pinhole of the output image                                    */
    double offset_x, offset_y;
    int width, height;           /* output size: j < width, i < height                             */
    int out_stride;              /* floats per output row (>= width)                               */
    int check_w2;                /* 1: (u, v) = (-1, -1) when Z <= -w2 * d1 (rectify.cpp:27)       */
    long long out_offset;        /* first element of this map inside mapx / mapy                   */
    double w2;
} tscm_map_desc;

/* mapx / mapy: host arrays of n_elems floats (caller-owned); seconds_kernel (may be NULL) returns
 * the device time of the map kernel alone (HIP events), without the copy back to the host.     */
int tscm_build_maps(const tscm_map_desc *maps, int n_maps, int device, int exact, float *mapx,
                    float *mapy, size_t n_elems, double *seconds_kernel);


/* ------------------------------------------------------------------ mono initialisation pieces
 * tscm_estimate_focal = TripleSphereCamera::estimate_focal (TS.cpp:110-168): one circle fit
 *   (cv::SVD::solveZ of a board_w x 4 matrix) per board row of every image with a board; *focal is
 *   the mean of the accepted samples, *n_used their number (0: "focal estimation failed", fx_ = 0).
 *   pix_u / pix_v: [n_views][board_w * board_h] host arrays, count[k] = pixels[k].size() (0 = no
 *   board in image k); (cx, cy) = the principal point guess of TS.cpp:43-44.
 * tscm_poses_from_r1r2t = the loop TS.cpp:62-74: Rt_[i] (row-major 3x3 [r1 r2 t]) -> rt_[i] =
 *   [cv::Rodrigues(R), t] with R built from float32 r1, r2 and their float cross product
 *   (has[i] == 0: rt left untouched).  Host-only.
 * tscm_estimate_extrinsic = TripleSphereCamera::estimate_extrinsic (TS.cpp:170-203): per image the
 *   "look at the board" rotation (:175-187), the corners un-projected onto that plane (:188-191) and a
 *   planar PnP.  The reference calls cv::solvePnPRansac there (external OpenCV routine, randomised);
 *   this entry point runs the deterministic equivalent for an all-inlier detection -- DLT homography,
 *   pose from its columns, polar orthonormalisation, Gauss-Newton on the 6 pose parameters -- so its
 *   result is NOT comparable bit-wise with OpenCV's, only as an initial guess of the same quality.
 *   Rt: [n_views*9] row-major 3x3 [r1 r2 t] = Rt_[k]; images with count[k] == 0 or a degenerate
 *   configuration keep the caller's values; *n_estimated = number of poses written.            */
int tscm_estimate_focal(const double *pix_u, const double *pix_v, const int *count, int n_views,
                        int board_w, int board_h, double cx, double cy, int device, double *focal,
                        int *n_used);
int tscm_poses_from_r1r2t(const double *Rt, const unsigned char *has, int n, double *rt);
int tscm_estimate_extrinsic(const double *intr9, const double *pix_u, const double *pix_v, const int *count,
                            int n_views, const double *worlds, int n_points, int board_w, int device,
                            double *Rt, int *n_estimated);


/* ------------------------------------------------------------------ corner lists
 * The reference has no on-disk form of its input: corners go straight from findCorner() into
 * TripleSphereCamera::calibrate (main.cpp:40-49, 196-222).  This is the library's own, minimal text
 * format for them (fixtures, hand-over between a detector and the calibration), dense like the
 * reference's vectors: pixels()[board] per camera, empty where the board was not detected.
 *
 *   TSCM-CORNERS 1
 *   cameras <C> boards <B> cols <W> rows <H> pitch <mm> image <width> <height>
 *   view <camera> <board>            -- followed by W*H lines "<u> <v>" (%.17g: exact round trip)
 *   ...
 * Board point j = (v*pitch, u*pitch, 0), j = u*W + v (main.cpp:12-18).
 * tscm_corners_read allocates has / pix_u / pix_v (release with tscm_corners_free).  Host-only.  */
typedef struct tscm_corner_set {
    int n_cameras, n_boards, board_cols, board_rows;
    double pitch;
    int image_width, image_height;
    unsigned char *has;            /* [C*B]                                   */
    double *pix_u, *pix_v;         /* [C*B*cols*rows], zero where !has        */
} tscm_corner_set;

int tscm_corners_write(const char *path, const tscm_corner_set *set);
int tscm_corners_read(const char *path, tscm_corner_set *set);
void tscm_corners_free(tscm_corner_set *set);

/* ------------------------------------------------------------------ application of remap tables
 * tscm_remap = cv::remap(src, dst, mapx, mapy, cv::INTER_LINEAR) with the defaulted border (constant 0) as
 * TripleSphereCamera::undistort / undistort_chessboard call it (TS.cpp:304, :329) for 8-bit images of 1 or 3
 * interleaved channels; to_gray != 0 (3 channels): the result is converted like cv::cvtColor(BGR2GRAY)
 * (findCorner.cpp:9-10 on the remapped chessboard, main.cpp:71) and dst has one channel.  OpenCV's fixed-point
 * interpolation (coordinates to 1/32 pixel, 15-bit weights).  dst: map_height rows of dst_stride bytes.
 */
int tscm_remap(const unsigned char *src, int width, int height, int stride, int channels, const float *mapx, const float *mapy, int map_width,
               int map_height, int map_stride, int to_gray, int device, unsigned char *dst, int dst_stride);

/* ------------------------------------------------------------------ corner candidates (first stage)
 * tscm_detect_corners  = findCorner() up to and including its score filter (DetectCorner/findCorner.cpp:7-66:
 *                        gradient angle / weight, secondDerivCornerMetric :103-142, nonMaximumSuppression(cxy + c45,
 *                        4, 0.07, 5) :144-193, getOrientations(r = 10) :200-349, scoreCorners(radii 8, 12, 16)
 *                        :391-490, removal of candidates with score < min_score (0.01 in the reference)), plus the
 *                        quadratic sub-pixel fit of subPixelLocation (:492-541) for EVERY kept candidate (the
 *                        reference applies it to the candidates the structure recovery assigned to a board; the fit
 *                        of a candidate does not depend on that assignment).
 * Input: 8-bit grey image (the reference converts BGR with cv::cvtColor first), `stride` bytes per row; sigma as in
 * findCorner(img, sigma): even (cv::GaussianBlur needs the odd kernel size 7 sigma + 1), main.cpp:32 passes 4.
 * Output order = the order the suppression finds the maxima (columns of cells left to right, cells top to bottom).
 * The chessboard structure recovery (chessboardsFromCorners, DetectCorner/chessboard.cpp) consumes this list.
 */
typedef struct tscm_corner_candidates {
    int n;                  /* candidates kept                                                     */
    int n_maxima;           /* maxima of the corner metric before the score filter                 */
    double *x, *y;          /* [n] pixel of the maximum (integral values; x = column, y = row)     */
    double *v1, *v2;        /* [2n] the two edge directions (unit vectors; (0,0) if none found)    */
    double *score;          /* [n]                                                                 */
    double *sub;            /* [2n] sub-pixel position (x, y)                                      */
    double seconds;         /* device time of the kernels                                          */
} tscm_corner_candidates;

int tscm_detect_corners(const unsigned char *gray, int width, int height, int stride, int sigma, double min_score, int device,
                        tscm_corner_candidates *out);
/* The same for n_images images of one size in ONE pass of the kernels (a calibration run detects on every image of
 * every camera: main.cpp:24-50): out[n_images], each freed with tscm_corner_candidates_free; out[i].seconds is the
 * image's share of the batch's device time.  Results are identical to n_images single calls. */
int tscm_detect_corners_batch(const unsigned char *const *images, int n_images, int width, int height, int stride, int sigma, double min_score,
                              int device, tscm_corner_candidates *out);
void tscm_corner_candidates_free(tscm_corner_candidates *c);

/* ------------------------------------------------------------------ chessboard structure (second stage)
 * tscm_chessboards_from_corners = chessboardsFromCorners (DetectCorner/chessboard.cpp:3-103): 3x3 seeds around every
 * candidate, energy-driven growth on the four sides, overlap resolution by energy, boards turned so that
 * cols >= rows.  Host logic (sequential, a few hundred candidates), input = the lists of tscm_detect_corners
 * (x, y = pixel of the maximum; v1, v2 = [2n] edge directions).  Board q is the rows[q] x cols[q] row-major matrix of
 * candidate indices cells[offset[q] .. offset[q + 1]).  findCorner (:67-95) then reads the sub-pixel position of
 * every board member; main.cpp:33 accepts an image when exactly one board of the expected size came out.
 */
typedef struct tscm_chessboards {
    int n_boards;
    int *rows, *cols;       /* [n_boards]                         */
    int *offset;            /* [n_boards + 1] start of each board */
    int *cells;             /* candidate indices                  */
} tscm_chessboards;

int tscm_chessboards_from_corners(int n, const double *x, const double *y, const double *v1, const double *v2, tscm_chessboards *out);
void tscm_chessboards_free(tscm_chessboards *b);

#ifdef __cplusplus
}
#endif
#endif /* TSCM_H */

/*
 * tscm_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE)
 *
 * Plain-C restatement of the reference's Triple-Sphere reprojection-error
 * Levenberg-Marquardt hot path (imuncle/TSCM_Calib):
 *   - cost functors            TS.h:100-131, multi_calib.h:146-195
 *   - problem build + options  TS.cpp:247-282, multi_calib.cpp:155-218
 *   - plain projection family  TS.cpp:332-344, TS.cpp:205-245, TS.h:58-69,
 *                              multi_calib.cpp:233-283
 *   - unprojection             TS.h:39-57
 * plus the behaviour of the un-vendored dependency that holds all of the
 * solver arithmetic: Ceres Solver (find_package(Ceres), CMakeLists.txt:7;
 * version unpinned, most likely 1.14.x): Jet forward-mode autodiff,
 * AngleAxisRotatePoint, TrustRegionMinimizer + LevenbergMarquardtStrategy with
 * default options and the DENSE_SCHUR linear solver. Those parts are restated
 * from Ceres' published algorithm (ceres/jet.h, ceres/rotation.h,
 * internal/ceres/trust_region_minimizer.cc, levenberg_marquardt_strategy.cc,
 * schur_eliminator_impl.h) -- none of that source is under /root/reference.
 *
 * PARITY UNPINNED: the reference ships no tests / golden vectors for this path
 * and Ceres/Eigen/OpenCV are not installed in the build container, so this
 * oracle cannot be checked against outputs of the real reference. It is pinned
 * only by (i) 50-digit mpmath known-answer values of the cited formulas
 * (tests/golden/kat_ts.json), (ii) project/unproject round trips, (iii)
 * dual-number vs central-difference Jacobians, and (iv) SciPy least_squares
 * reaching the same optimum cost on config-1-size problems.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * call into this library.  The product (tscm_calib_amd/) never does.
 */
#ifndef TSCM_ORACLE_H
#define TSCM_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* Problem description shared by the mono and the multi-camera solve.
 * A "view" is one (camera, board/frame) pair with `count` detected corners;
 * corner j of a view observes board point j (multi_calib.cpp:169-176:
 * `for j < pixels[i].size()` pairs pixels[i][j] with worlds_[j]).           */
typedef struct {
    int n_cameras;             /* C                                            */
    int n_boards;              /* B (mono: number of views/images)             */
    int n_points;              /* corners on the board (e.g. 54 = 9x6)         */
    int n_views;               /* number of (camera, board) pairs with corners */
    const double *board_xy;    /* [n_points*2] x,y ; z is forced to 0          */
                               /*   (TS.h:107-109, multi_calib.h:154-156)      */
    const int *view_camera;    /* [n_views]                                    */
    const int *view_board;     /* [n_views]                                    */
    const int *view_offset;    /* [n_views] first corner of the view in obs_*  */
    const int *view_count;     /* [n_views] corners in the view (<= n_points)  */
    const double *obs_u;       /* [N] observed pixel x                          */
    const double *obs_v;       /* [N] observed pixel y                          */
    double *cam_rt;            /* [C*6] angle-axis + t, in/out (multi only)     */
    double *intr;              /* [C*9] fx fy cx cy xi lambda alpha b c, in/out */
    double *board_rt;          /* [B*6] angle-axis + t, in/out                  */
    const unsigned char *cam_pose_constant; /* [C] 1 = SetParameterBlockConstant
                                               (multi_calib.cpp:186); may be NULL */
    int mono;                  /* 1: TS.h functor (no camera pose), C must be 1 */
    const unsigned char *board_pose_constant; /* [B] 1 = SetParameterBlockConstant on the board's pose block; may be NULL.
                                  Not used by the reference (BASELINE config 2's "intrinsics-only" form).  Ceres removes
                                  constant blocks from the reduced program; if NO board block is left it would switch
                                  DENSE_SCHUR to DENSE_QR -- here the same normal equations go through the Cholesky path. */
} orc_problem;

typedef struct {
    int max_num_iterations;           /* 100 mono (TS.cpp:274), 50 multi (Ceres default) */
    double function_tolerance;        /* 1e-6  */
    double gradient_tolerance;        /* 1e-10 */
    double parameter_tolerance;       /* 1e-8  */
    double initial_trust_region_radius; /* 1e4 */
    double max_trust_region_radius;   /* 1e16  */
    double min_trust_region_radius;   /* 1e-32 */
    double min_relative_decrease;     /* 1e-3  */
    double min_lm_diagonal;           /* 1e-6  */
    double max_lm_diagonal;           /* 1e32  */
    int max_num_consecutive_invalid_steps; /* 5 */
    int jacobi_scaling;               /* 1     */
} orc_options;

enum { ORC_CONVERGENCE = 0, ORC_NO_CONVERGENCE = 1, ORC_FAILURE = 2 };

typedef struct {
    int iteration;
    int step_is_valid, step_is_successful;
    double cost, cost_change, gradient_max_norm, gradient_norm, step_norm;
    double relative_decrease, trust_region_radius;
} orc_iteration;

typedef struct {
    int termination_type;
    int num_iterations;            /* entries in `iterations` (incl. iteration 0) */
    int num_successful_steps, num_unsuccessful_steps;
    double initial_cost, final_cost;
    int n_residual_blocks;         /* N corners in the program */
    orc_iteration iterations[256];
    char message[128];
    double seconds_total;          /* wall time of the minimiser loop */
    double seconds_jacobian;       /* time inside residual+Jacobian evaluation */
    double seconds_linear;         /* time inside Schur eliminate/solve/back-substitute */
} orc_summary;

void orc_default_options(orc_options *o, int mono);

/* ---- camera model -------------------------------------------------------*/
/* TS.cpp:332-344 (project, with the skew terms b,c). intr = 9 doubles.      */
void orc_project(const double *intr, const double *P, double *uv);
/* TS.h:39-57 (get_unit_sphere_coordinate with transform = identity).        */
void orc_unproject(const double *intr, const double *uv, double *ray);
/* ceres::AngleAxisRotatePoint (external), double instantiation.             */
void orc_angle_axis_rotate_point(const double *aa, const double *pt, double *out);
/* cv::Rodrigues(r -> R) as used by update_param (multi_calib.h:42-57,104-108),
 * row-major 3x3. */
void orc_rodrigues(const double *aa, double *R);

/* ---- functors -----------------------------------------------------------*/
/* T = double instantiation of TS.h:100-131.  board_pt = (x,y[,ignored z]).  */
void orc_mono_residual(const double *intr, const double *rt, const double *obs,
                       const double *board_pt, double *res);
/* T = double instantiation of multi_calib.h:146-195.                        */
void orc_multi_residual(const double *cam_rt, const double *board_rt, const double *intr,
                        const double *obs, const double *board_pt, double *res);
/* T = Jet instantiation (what ceres::AutoDiffCostFunction evaluates).  Jacobians are
 * row-major [2 x block] like CostFunction::Evaluate; any pointer may be NULL. */
void orc_mono_autodiff(const double *intr, const double *rt, const double *obs,
                       const double *board_pt, double *res, double *J_intr /*2x9*/,
                       double *J_rt /*2x6*/);
void orc_multi_autodiff(const double *cam_rt, const double *board_rt, const double *intr,
                        const double *obs, const double *board_pt, double *res,
                        double *J_cam /*2x6*/, double *J_board /*2x6*/, double *J_intr /*2x9*/);

/* ---- batched evaluation of a whole problem ------------------------------*/
/* residuals [2N] interleaved (u,v per corner); J_* may be NULL; layout per corner:
 * J_cam[2*6], J_board[2*6], J_intr[2*9] row-major.  Returns cost = 0.5*sum r^2.
 * use_jets=0 evaluates the double functor (cost-only path of Ceres).           */
/* threads of the O(N) passes (default 1 = the sequential checker path; see tscm_oracle.c) */
void orc_set_num_threads(int n);
int orc_get_num_threads(void);
int orc_max_threads(void);

double orc_evaluate(const orc_problem *p, int use_jets, double *residuals,
                    double *J_cam, double *J_board, double *J_intr);

/* ---- the solve (TS.cpp:247-282 / multi_calib.cpp:155-218 + Ceres) -------*/
int orc_solve(const orc_problem *p, const orc_options *opt, orc_summary *summary);

/* ---- error reports ------------------------------------------------------*/
/* multi_calib.cpp:233-283: per-camera mean Euclidean pixel error (projection
 * with skew terms, poses through Rodrigues matrices) and the global mean.
 * per_camera may be NULL. Returns global mean. */
double orc_mean_reprojection_error(const orc_problem *p, double *per_camera);
/* sqrt(sum r^2 / N) over all corners with the double functor. */
double orc_rmse(const orc_problem *p);


/* ---- rig initialisation: MultiCalib::MultiCalib (multi_calib.cpp:6-153) -----------------------
 * The step immediately before the joint LM: chains camera i to camera i-1 through every board
 * both see, keeps the pose hypothesis with the smallest summed reprojection error
 * (TS.h:58-69: SUM of Euclidean pixel errors, projection with skew terms), then does the same
 * for every board pose.  Rt_to_R_t (multi_calib.h:130-137) builds R from FLOAT32 copies of the
 * first two columns and their float cross product; MultiCalib_camera / _chessboard convert
 * R to angle-axis with cv::Rodrigues (external, OpenCV calib3d: SVD orthonormalisation, then
 * axis from the antisymmetric part), restated here from the published algorithm.            */
typedef struct {
    int n_cameras, n_boards, n_points;
    const double *worlds;        /* [n_points*3]  board points x,y,z (z is used here: TS.h:63)  */
    const double *intr;          /* [C*9]                                                       */
    const unsigned char *has;    /* [C*B] has_chessboard                                        */
    const double *Rt;            /* [C*B*9] row-major 3x3 [r1 r2 t] of TripleSphereCamera::Rt(j) */
    const double *pix_u, *pix_v; /* [C*B*n_points] pixels()[j] (valid where has)                */
} orc_rig_input;

/* outputs: cam_R [C*9] row-major, cam_t [C*3], cam_rt [C*6]; board_R [B*9], board_t [B*3],
 * board_rt [B*6], board_initial [B]; cam_choice [C] index of the winning hypothesis (among the
 * common boards, in board order; -1 for camera 0), cam_min_error [C].
 * returns 0, or -1 where the reference has undefined behaviour (adjacent cameras share no board). */
int orc_rig_init(const orc_rig_input *in, double *cam_R, double *cam_t, double *cam_rt,
                 double *board_R, double *board_t, double *board_rt, unsigned char *board_initial,
                 int *cam_choice, double *cam_min_error);
/* multi_calib.cpp:52-78 for nj given hypotheses of camera i (Rs [nj*9], ts [nj*3]); (Rp, tp) = pose
 * of camera i-1.  The inner loop of orc_rig_init, exported for bounded CPU-baseline samples. */
void orc_rig_hypothesis_errors(const orc_rig_input *in, int i, const double *Rp, const double *tp,
                               const double *Rs, const double *ts, int nj, double *errors);
/* cv::Rodrigues(R -> rvec) restated (row-major 3x3 in, 3 out). */
void orc_rodrigues_inverse(const double *R, double *rvec);
/* Rt_to_R_t: row-major 3x3 [r1 r2 t] -> R (row-major, float32-rounded columns), t. */
void orc_Rt_to_R_t(const double *Rt, double *R, double *t);

/* ---- mono initialisation pieces (TS.cpp:62-74, 110-168) ------------------------------------*/
/* estimate_focal: pix_u/pix_v [n_views][width*height], count[k] = pixels[k].size(); returns -1 for
 * width < 4 (solveZ of a wide matrix is not restated) */
int orc_estimate_focal(const double *pix_u, const double *pix_v, const int *count, int n_views, int width, int height,
                       double cx, double cy, double *focal, int *total_num);
int orc_focal_sample(const double *pu, const double *pv, int width, double cx, double cy, double *gamma);
void orc_Rt_to_rt(const double *Rt, double *rt);
/* estimate_extrinsic (TS.cpp:170-203) with a deterministic planar PnP in place of cv::solvePnPRansac */
int orc_planar_pnp(const double *worlds, const double *xn, const double *yn, int n, double *R, double *t);
int orc_estimate_extrinsic(const double *intr, const double *pix_u, const double *pix_v, const int *count, int n_views,
                           const double *worlds, int n, int board_w, double *Rt_out);

/* ---- remap tables (TS.cpp:284-330, EpipolarRectify/rectify.cpp:86-199); same layout as tscm_map_desc */
typedef struct {
    double intr[9];
    double R[9];
    double fx, fy, cx, cy;
    double offset_x, offset_y;
    int width, height;
    int out_stride;
    int check_w2;
    long long out_offset;
    double w2;
} orc_map_desc;
void orc_build_map(const orc_map_desc *m, float *mapx, float *mapy);
/* Remap::calc_R (rectify.cpp:234-248), R row-major */
void orc_rectify_pair_rotation(const double *t1, const double *t2, double *R);

#ifdef __cplusplus
}
#endif

/* ---- corner candidates (tscm_oracle_corners.c; DetectCorner/findCorner.cpp) ---- */
void orc_corner_gradients(const unsigned char *gray, int w, int h, int stride, double *angle, double *weight);
void orc_corner_normalise(const unsigned char *gray, int w, int h, int stride, double *img);
void orc_gaussian_kernel(int sigma, double *k);
int orc_corner_metric(const double *I, int w, int h, int sigma, double *metric, double *Ixy);
int orc_corner_nms(const double *img, int width, int height, int n, double tau, int margin, int cap, double *px, double *py);
void orc_corner_orientation(const double *angle, const double *weight, int width, int height, int cu, int cv, int r, double *v);
double orc_corner_correlation_score(const double *img, const double *weight, int width, int u, int v, int r, const double *vv);
double orc_corner_score(const double *img, const double *weight, int width, int height, double px, double py, const double *vv);
void orc_subpixel_operator(double *X);
void orc_corner_subpixel(const double *Ixy, int width, const double *X, double px, double py, double *out);
int orc_detect_corners(const unsigned char *gray, int width, int height, int stride, int sigma, int cap,
                       double *px, double *py, double *v, double *score, double *sub, double *metric_out, double *ixy_out);

/* ---- chessboard structure recovery (tscm_oracle_boards.c; DetectCorner/chessboard.cpp) ---- */
int orc_chessboards_from_corners(int n, const double *px, const double *py, const double *v1, const double *v2,
                                 int max_boards, int max_cells, int *rows, int *cols, int *cells);

/* ---- cv::remap(INTER_LINEAR) + BGR2GRAY (tscm_oracle_remap.c) ---- */
void orc_remap_bilinear(const unsigned char *src, int w, int h, int stride, int channels, const float *mapx, const float *mapy, int map_w, int map_h,
                        int map_stride, int to_gray, unsigned char *dst, int dst_stride);

#endif

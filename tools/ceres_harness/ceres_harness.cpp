// ceres_harness.cpp -- the reference's Ceres solve on a fixture problem, with the REAL Ceres.
//
// What it reproduces (imuncle/TSCM_Calib), restated from the cited lines, not copied:
//   * the two cost functors, templated on the scalar type for ceres::AutoDiffCostFunction:
//       mono   TripleSphereCamera::ReprojectionError::operator()   TS.h:100-131       <2, 9, 6>   (TS.cpp:261-264)
//       multi  MultiCalib::ReprojectionError::operator()           multi_calib.h:146-195  <2, 6, 6, 9> (multi_calib.cpp:177-180)
//   * the problem build: one residual block per corner, NULL loss (TS.cpp:251-269, multi_calib.cpp:162-207),
//     cameras_[0].rt_ constant (multi_calib.cpp:186)
//   * the options: DENSE_SCHUR, minimizer_progress_to_stdout = false, max_num_iterations = 100 for the mono solve
//     (TS.cpp:271-274) and the library default for the rig (multi_calib.cpp:209-212), everything else default.
//
// usage: ceres_harness problem.bin result.json [name]
//   problem.bin  the format of examples/dropin_demo.cpp / tools/ceres_harness/export_problems.py:
//                int32 {C, B, n_points, V, N, mono}, then board_xy, view_camera, view_board, view_offset, view_count,
//                obs_u, obs_v, cam_rt, intr, board_rt, cam_pose_constant (declaration order of tscm_problem)
//   result.json  the schema of tests/golden/lm_traces.json plus the full parameter arrays and the Ceres version:
//                drop it into tests/golden/ceres_<name>.json and tests/test_ceres_golden.py compares the oracle
//                (CPU) and the HIP path (GPU) with it.
#include <ceres/ceres.h>
#include <ceres/rotation.h>
#include <ceres/version.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

namespace {

// common tail of both functors: triple-sphere projection of the camera-frame point P and the residual
// (TS.h:117-129, multi_calib.h:170-193).  intrinsic = fx fy cx cy xi lambda alpha b c; b, c are not used there.
template <typename T>
inline void ts_residual(const T *const intrinsic, const T P[3], double obs_u, double obs_v, T *residuals)
{
    const T d1 = ceres::sqrt(P[0] * P[0] + P[1] * P[1] + P[2] * P[2]);
    const T z1 = P[2] + intrinsic[4] * d1;
    const T d2 = ceres::sqrt(P[0] * P[0] + P[1] * P[1] + z1 * z1);
    const T z2 = z1 + intrinsic[5] * d2;
    const T d3 = ceres::sqrt(P[0] * P[0] + P[1] * P[1] + z2 * z2);
    const T ksai = z2 + intrinsic[6] / (T(1.0) - intrinsic[6]) * d3;
    const T u = intrinsic[0] * P[0] / ksai + intrinsic[2];
    const T v = intrinsic[1] * P[1] / ksai + intrinsic[3];
    residuals[0] = T(obs_u) - u;
    residuals[1] = T(obs_v) - v;
}

// TS.h:93-134: parameter blocks (intrinsic[9], rt[6]); the board point has z = 0 (TS.h:107-109)
struct MonoReprojectionError {
    MonoReprojectionError(double u, double v, double x, double y) : u_(u), v_(v), x_(x), y_(y) {}
    template <typename T>
    bool operator()(const T *const intrinsic, const T *const rt, T *residuals) const
    {
        const T p[3] = { T(x_), T(y_), T(0.0) };
        T P[3];
        ceres::AngleAxisRotatePoint(rt, p, P);                 // TS.h:112
        P[0] += rt[3]; P[1] += rt[4]; P[2] += rt[5];           // TS.h:113-115
        ts_residual(intrinsic, P, u_, v_, residuals);
        return true;
    }
    double u_, v_, x_, y_;
};

// multi_calib.h:138-199: parameter blocks (camera_rt[6], chessboard_rt[6], intrinsic[9])
struct MultiReprojectionError {
    MultiReprojectionError(double u, double v, double x, double y) : u_(u), v_(v), x_(x), y_(y) {}
    template <typename T>
    bool operator()(const T *const camera_rt, const T *const chessboard_rt, const T *const intrinsic, T *residuals) const
    {
        const T p[3] = { T(x_), T(y_), T(0.0) };
        T Pw[3], P[3];
        ceres::AngleAxisRotatePoint(chessboard_rt, p, Pw);     // multi_calib.h:158
        Pw[0] += chessboard_rt[3]; Pw[1] += chessboard_rt[4]; Pw[2] += chessboard_rt[5];
        ceres::AngleAxisRotatePoint(camera_rt, Pw, P);         // multi_calib.h:164
        P[0] += camera_rt[3]; P[1] += camera_rt[4]; P[2] += camera_rt[5];
        ts_residual(intrinsic, P, u_, v_, residuals);
        return true;
    }
    double u_, v_, x_, y_;
};

template <typename T>
std::vector<T> rd(FILE *f, size_t n)
{
    std::vector<T> v(n);
    if (n && fread(v.data(), sizeof(T), n, f) != n) { fprintf(stderr, "short read\n"); exit(2); }
    return v;
}

void put_flat(FILE *o, const char *key, const std::vector<double> &v)
{
    fprintf(o, " \"%s\": [", key);
    for (size_t i = 0; i < v.size(); ++i) fprintf(o, "%s%.17g", i ? ", " : "", v[i]);
    fprintf(o, "],\n");
}

void put(FILE *o, const char *key, const std::vector<double> &v, int cols)
{
    fprintf(o, " \"%s\": [", key);
    for (size_t r = 0; r * cols < v.size(); ++r) {
        fprintf(o, "%s[", r ? ", " : "");
        for (int c = 0; c < cols; ++c) fprintf(o, "%s%.17g", c ? ", " : "", v[r * cols + c]);
        fprintf(o, "]");
    }
    fprintf(o, "],\n");
}

}  // namespace

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s problem.bin result.json [name]\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    std::vector<int> h = rd<int>(f, 6);
    const int C = h[0], B = h[1], n = h[2], V = h[3], N = h[4], mono = h[5];
    std::vector<double> board_xy = rd<double>(f, 2 * (size_t)n);
    std::vector<int> view_camera = rd<int>(f, V), view_board = rd<int>(f, V), view_offset = rd<int>(f, V), view_count = rd<int>(f, V);
    std::vector<double> obs_u = rd<double>(f, N), obs_v = rd<double>(f, N);
    std::vector<double> cam_rt = rd<double>(f, 6 * (size_t)C), intr = rd<double>(f, 9 * (size_t)C), board_rt = rd<double>(f, 6 * (size_t)B);
    std::vector<unsigned char> cam_const = rd<unsigned char>(f, C);
    fclose(f);

    ceres::Problem problem;
    long blocks = 0;
    for (int v = 0; v < V; ++v) {
        const int m = view_camera[v], b = view_board[v];
        for (int j = 0; j < view_count[v]; ++j, ++blocks) {
            const double u = obs_u[view_offset[v] + j], w = obs_v[view_offset[v] + j];
            const double x = board_xy[2 * j], y = board_xy[2 * j + 1];
            if (mono) {
                ceres::CostFunction *cost = new ceres::AutoDiffCostFunction<MonoReprojectionError, 2, 9, 6>(new MonoReprojectionError(u, w, x, y));
                problem.AddResidualBlock(cost, NULL, intr.data(), board_rt.data() + 6 * b);                     // TS.cpp:265-267
            } else {
                ceres::CostFunction *cost = new ceres::AutoDiffCostFunction<MultiReprojectionError, 2, 6, 6, 9>(new MultiReprojectionError(u, w, x, y));
                problem.AddResidualBlock(cost, NULL, cam_rt.data() + 6 * m, board_rt.data() + 6 * b, intr.data() + 9 * m);   // multi_calib.cpp:181-184
            }
        }
    }
    if (!mono)
        for (int m = 0; m < C; ++m)
            if (cam_const[m] && problem.HasParameterBlock(cam_rt.data() + 6 * m)) problem.SetParameterBlockConstant(cam_rt.data() + 6 * m);   // multi_calib.cpp:186

    ceres::Solver::Options options;
    options.linear_solver_type = ceres::DENSE_SCHUR;           // TS.cpp:272, multi_calib.cpp:210
    options.minimizer_progress_to_stdout = false;              // TS.cpp:273, multi_calib.cpp:211
    if (mono) options.max_num_iterations = 100;                // TS.cpp:274 (multi_calib.cpp:212 is commented out)
    ceres::Solver::Summary summary;
    ceres::Solve(options, &problem, &summary);
    printf("%s\n", summary.BriefReport().c_str());

    FILE *o = fopen(argv[2], "w");
    if (!o) { perror(argv[2]); return 2; }
    fprintf(o, "{\n \"name\": \"%s\",\n \"ceres_version\": \"%s\",\n", argc > 3 ? argv[3] : "", CERES_VERSION_STRING);
    fprintf(o, " \"termination_type\": %d,\n \"message\": \"%s\",\n", (int)summary.termination_type, summary.message.c_str());
    fprintf(o, " \"num_iterations\": %d,\n", (int)summary.iterations.size());
    fprintf(o, " \"num_successful_steps\": %d,\n \"num_unsuccessful_steps\": %d,\n", summary.num_successful_steps, summary.num_unsuccessful_steps);
    std::vector<double> costs, radii, gmax, steps;
    std::vector<double> ok;
    for (size_t i = 0; i < summary.iterations.size(); ++i) {
        const ceres::IterationSummary &it = summary.iterations[i];
        costs.push_back(it.cost); radii.push_back(it.trust_region_radius); gmax.push_back(it.gradient_max_norm);
        steps.push_back(it.step_norm); ok.push_back(it.step_is_successful ? 1.0 : 0.0);
    }
    put_flat(o, "costs", costs);
    put_flat(o, "radii", radii);
    put_flat(o, "gradient_max_norms", gmax);
    put_flat(o, "step_norms", steps);
    put_flat(o, "step_is_successful", ok);
    put(o, "intr", intr, 9);
    put(o, "cam_rt", cam_rt, 6);
    put(o, "board_rt", board_rt, 6);
    fprintf(o, " \"initial_cost\": %.17g,\n \"final_cost\": %.17g,\n", summary.initial_cost, summary.final_cost);
    fprintf(o, " \"rmse\": %.17g,\n \"n_residual_blocks\": %ld\n}\n", blocks ? std::sqrt(2.0 * summary.final_cost / (double)blocks) : 0.0, blocks);
    fclose(o);
    return 0;
}

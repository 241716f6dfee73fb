// dropin_demo.cpp -- host C++ calling the C ABI exactly the way the reference's
// MultiCalib::calibrate() / TripleSphereCamera::refinement() would, with
// std::vector parameter blocks instead of the OpenCV-backed classes.
//
//   g++ -std=c++11 -I include examples/dropin_demo.cpp -L tscm_calib_amd/csrc -ltscm_hip
//       -Wl,-rpath,$PWD/tscm_calib_amd/csrc -o examples/dropin_demo
//   examples/dropin_demo problem.bin result.bin
//
// problem.bin : int32 {C, B, n_points, V, N, mono} then the arrays of tscm_problem in declaration
//               order (board_xy, view_camera, view_board, view_offset, view_count, obs_u, obs_v,
//               cam_rt, intr, board_rt, cam_pose_constant)
// result.bin  : cam_rt, intr, board_rt (doubles) then {termination, iterations} (int32), final cost, rmse
#include <tscm/tscm.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

template <typename T>
static std::vector<T> rd(FILE *f, size_t n)
{
    std::vector<T> v(n);
    if (n && fread(v.data(), sizeof(T), n, f) != n) { fprintf(stderr, "short read\n"); exit(2); }
    return v;
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s problem.bin result.bin\n", argv[0]); return 2; }
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

    tscm_problem P = {};
    P.n_cameras = C; P.n_boards = B; P.n_points = n; P.n_views = V;
    P.board_xy = board_xy.data();
    P.view_camera = view_camera.data(); P.view_board = view_board.data();
    P.view_offset = view_offset.data(); P.view_count = view_count.data();
    P.obs_u = obs_u.data(); P.obs_v = obs_v.data();
    P.cam_rt = mono ? nullptr : cam_rt.data(); P.intr = intr.data(); P.board_rt = board_rt.data();
    P.cam_pose_constant = cam_const.data(); P.mono = mono;

    tscm_options opt;
    tscm_default_options(&opt, mono);
    tscm_summary summary;
    const int rc = mono ? tscm_solve_mono(&P, &opt, &summary) : tscm_solve_multi(&P, &opt, &summary);
    if (rc != 0) { fprintf(stderr, "tscm error %d: %s\n", rc, tscm_last_error()); return 1; }
    printf("%s  iterations %d  initial cost %.6e  final cost %.6e  rmse %.6f px  (%.3f ms)\n", summary.message,
           summary.num_iterations - 1, summary.initial_cost, summary.final_cost, summary.rmse, 1e3 * summary.seconds_solve);

    FILE *o = fopen(argv[2], "wb");
    if (!o) { perror(argv[2]); return 2; }
    fwrite(cam_rt.data(), sizeof(double), cam_rt.size(), o);
    fwrite(intr.data(), sizeof(double), intr.size(), o);
    fwrite(board_rt.data(), sizeof(double), board_rt.size(), o);
    const int meta[2] = { summary.termination_type, summary.num_iterations };
    fwrite(meta, sizeof(int), 2, o);
    fwrite(&summary.final_cost, sizeof(double), 1, o);
    fwrite(&summary.rmse, sizeof(double), 1, o);
    fclose(o);
    return 0;
}

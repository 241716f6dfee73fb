// calibrate_from_corners.cpp -- the reference's main.cpp from the corner lists on (main.cpp:196-319), on the GPU:
//   corners.txt (TSCM-CORNERS 1, see tscm.h)  ->  per-camera calibration  ->  MultiCalib  ->  calib.yaml
//
//   g++ -std=c++11 -I include examples/calibrate_from_corners.cpp -L tscm_calib_amd/csrc -ltscm_hip
//       -Wl,-rpath,$PWD/tscm_calib_amd/csrc -o examples/calibrate_from_corners        (one command line)
//   examples/calibrate_from_corners corners.txt calib.yaml
#include <tscm/tscm_calib.hpp>

#include <cstdio>

int main(int argc, char **argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: %s corners.txt calib.yaml\n", argv[0]); return 2; }
    tscm_corner_set cs;
    if (tscm_corners_read(argv[1], &cs) != 0) { std::fprintf(stderr, "%s\n", tscm_last_error()); return 1; }
    int rc = 0;
    try {
        const int C = cs.n_cameras, B = cs.n_boards, n = cs.board_cols * cs.board_rows;
        const tscm::Size board = { cs.board_cols, cs.board_rows }, image = { cs.image_width, cs.image_height };
        std::vector<tscm::Point3d> worlds;                                   // main.cpp:12-18
        for (int u = 0; u < board.height; ++u)
            for (int v = 0; v < board.width; ++v) worlds.push_back(tscm::Point3d{ v * cs.pitch, u * cs.pitch, 0.0 });
        std::vector<tscm::TripleSphereCamera> cameras(C);
        for (int m = 0; m < C; ++m) {                                        // main.cpp:196-222
            std::vector<std::vector<tscm::Point2d> > pixels(B);
            std::vector<bool> has(B);
            for (int b = 0; b < B; ++b) {
                has[b] = cs.has[(size_t)m * B + b] != 0;
                if (!has[b]) continue;
                pixels[b].resize(n);
                for (int j = 0; j < n; ++j) pixels[b][j] = tscm::Point2d{ cs.pix_u[((size_t)m * B + b) * n + j], cs.pix_v[((size_t)m * B + b) * n + j] };
            }
            const bool ok = cameras[m].calibrate(pixels, has, worlds, image, board);
            std::printf("camera %d: %s, rmse %.4f px, fx %.3f fy %.3f cx %.3f cy %.3f xi %.4f lambda %.4f alpha %.4f\n", m, ok ? "converged" : "NOT converged",
                        cameras[m].summary.rmse, cameras[m].intrinsic_[0], cameras[m].intrinsic_[1], cameras[m].intrinsic_[2], cameras[m].intrinsic_[3],
                        cameras[m].intrinsic_[4], cameras[m].intrinsic_[5], cameras[m].intrinsic_[6]);
            if (!ok) rc = 3;
        }
        if (C > 1) {
            tscm::MultiCalib mul_calib(cameras, worlds);                     // main.cpp:233
            mul_calib.calibrate();                                           // main.cpp:234
            std::printf("%s  iterations %d  rmse %.4f px\n", mul_calib.summary.message, mul_calib.summary.num_iterations - 1, mul_calib.summary.rmse);
            for (int m = 0; m < C; ++m) std::printf("camera_%d reprojection error: %.6f\n", m, mul_calib.camera_error[m]);
            std::printf("average reproject error: %.6f\n", mul_calib.mean_error);
            mul_calib.write_yaml(argv[2]);                                   // main.cpp:305-319
        } else if (C == 1) {
            const double I3[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 }, t0[3] = { 0, 0, 0 };
            tscm::check(tscm_yaml_write(argv[2], 1, cameras[0].intrinsic_.data(), I3, t0));
        }
    } catch (const std::exception &e) {
        std::fprintf(stderr, "%s\n", e.what());
        rc = 1;
    }
    tscm_corners_free(&cs);
    return rc;
}

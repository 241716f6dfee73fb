// multicalib_demo.cpp -- the reference's main.cpp flow after corner detection (main.cpp:196-319)
// written against tscm_calib.hpp: mono-calibrated cameras -> MultiCalib(cameras, worlds) ->
// calibrate() -> YAML, with the reference's class and member names.
//
//   g++ -std=c++11 -I include examples/multicalib_demo.cpp -L tscm_calib_amd/csrc -ltscm_hip
//       -Wl,-rpath,$PWD/tscm_calib_amd/csrc -o examples/multicalib_demo      (one command line)
//   examples/multicalib_demo rig.bin result.bin calib.yaml
//
// rig.bin    : int32 {C, B, n, board_w, board_h} then doubles worlds[n*3], intr[C*9], uint8 has[C*B],
//              doubles Rt[C*B*9], pix_u[C*B*n], pix_v[C*B*n]   (= tscm_rig_input)
// result.bin : doubles cam_rt[C*6], intr[C*9], board_rt[B*6], camera_error[C], mean_error, focal0;
//              int32 {termination, iterations}; floats mapx[16], mapy[16] (corner of camera 0's undistort table)
#include <tscm/tscm_calib.hpp>

#include <cstdio>
#include <string>
#include <cstdlib>

template <typename T>
static std::vector<T> rd(FILE *f, size_t n)
{
    std::vector<T> v(n);
    if (n && fread(v.data(), sizeof(T), n, f) != n) { fprintf(stderr, "short read\n"); exit(2); }
    return v;
}

int main(int argc, char **argv)
{
    if (argc < 4) { fprintf(stderr, "usage: %s rig.bin result.bin calib.yaml\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    const std::vector<int> h = rd<int>(f, 5);
    const int C = h[0], B = h[1], n = h[2];
    const tscm::Size board = { h[3], h[4] };
    const std::vector<double> W = rd<double>(f, 3 * (size_t)n), I = rd<double>(f, 9 * (size_t)C);
    const std::vector<unsigned char> has = rd<unsigned char>(f, (size_t)C * B);
    const std::vector<double> Rt = rd<double>(f, 9 * (size_t)C * B), pu = rd<double>(f, (size_t)C * B * n), pv = rd<double>(f, (size_t)C * B * n);
    fclose(f);
    try {
        std::vector<tscm::Point3d> worlds(n);
        for (int c = 0; c < n; ++c) worlds[c] = tscm::Point3d{ W[3 * c], W[3 * c + 1], W[3 * c + 2] };
        std::vector<tscm::TripleSphereCamera> cameras(C);
        for (int m = 0; m < C; ++m) {
            tscm::TripleSphereCamera &cam = cameras[m];
            cam.intrinsic_.assign(&I[9 * (size_t)m], &I[9 * (size_t)m] + 9);
            cam.has_chessboard_.resize(B); cam.Rt_.resize(B); cam.pixels_.resize(B); cam.rt_.resize(B);
            for (int j = 0; j < B; ++j) {
                cam.has_chessboard_[j] = has[(size_t)m * B + j] != 0;
                if (!cam.has_chessboard_[j]) continue;
                for (int k = 0; k < 9; ++k) cam.Rt_[j].a[k] = Rt[9 * ((size_t)m * B + j) + k];
                cam.pixels_[j].resize(n);
                for (int c = 0; c < n; ++c) cam.pixels_[j][c] = tscm::Point2d{ pu[((size_t)m * B + j) * n + c], pv[((size_t)m * B + j) * n + c] };
            }
        }
        // a piece of the mono initialisation: the focal estimate of camera 0 from its own corners
        tscm::TripleSphereCamera probe = cameras[0];
        probe.intrinsic_[2] = 1280 / 2 - 0.5; probe.intrinsic_[3] = 1080 / 2 - 0.5;          // TS.cpp:43-44
        const double focal0 = probe.estimate_focal(probe.pixels_, board);

        // ... and the whole mono calibration of camera 0 from its corner lists alone (main.cpp:222 -> TS.cpp:30-105)
        tscm::TripleSphereCamera mono0;
        const bool mono_ok = mono0.calibrate(cameras[0].pixels_, cameras[0].has_chessboard_, worlds, tscm::Size{ 1280, 1080 }, board);
        printf("mono calibration of camera 0 from raw corners: %s, rmse %.4f px, fx %.2f (rig input %.2f)\n", mono_ok ? "converged" : "NOT converged",
               mono0.summary.rmse, mono0.intrinsic_[0], cameras[0].intrinsic_[0]);

        tscm::MultiCalib mul_calib(cameras, worlds);           // main.cpp:233
        if (argc > 4 && std::string(argv[4]) == "sharded") {
            // the multi-GPU form of the same call (one process per GPU in production; here one rank): the frames are
            // sharded over the ranks of a communicator, every rank ends with all parameters
            unsigned char id[TSCM_UNIQUE_ID_BYTES];
            tscm_comm *comm = nullptr;
            if (tscm_comm_unique_id(id) != 0 || tscm_comm_create(id, 0, 1, 0, &comm) != 0) { fprintf(stderr, "tscm_comm_create: %s\n", tscm_last_error()); return 1; }
            tscm_options o;
            tscm_default_options(&o, 0);
            o.exec_flags |= TSCM_EXEC_KEEP_SINGLE_RANK_COMM;    // run the communicator code path although it has one rank
            mul_calib.set_sharding(0, 1, comm);
            mul_calib.calibrate(&o);
            mul_calib.set_sharding(0, 1, nullptr);
            tscm_comm_destroy(comm);
        } else
        mul_calib.calibrate();                                 // main.cpp:234
        printf("%s  iterations %d  final cost %.6e\n", mul_calib.summary.message, mul_calib.summary.num_iterations - 1, mul_calib.summary.final_cost);
        for (int m = 0; m < C; ++m) printf("camera_%d reprojection error: %.6f\n", m, mul_calib.camera_error[m]);
        printf("average reproject error: %.6f\n", mul_calib.mean_error);
        mul_calib.write_yaml(argv[3]);                         // main.cpp:305-319

        // undistortion table of the calibrated camera 0 (TS.cpp:284-306), through the same class
        tscm::TripleSphereCamera cam0;
        cam0.intrinsic_ = mul_calib.cameras_[0].intrinsic_;
        std::vector<float> mapx, mapy;
        cam0.undistort(300.0, 300.0, 639.5, 539.5, tscm::Size{ 64, 48 }, mapx, mapy);

        FILE *o = fopen(argv[2], "wb");
        if (!o) { perror(argv[2]); return 2; }
        for (int m = 0; m < C; ++m) fwrite(mul_calib.cameras_[m].rt_.data(), sizeof(double), 6, o);
        for (int m = 0; m < C; ++m) fwrite(mul_calib.cameras_[m].intrinsic_.data(), sizeof(double), 9, o);
        for (int j = 0; j < B; ++j) fwrite(mul_calib.chessboards_[j].rt_.data(), sizeof(double), 6, o);
        fwrite(mul_calib.camera_error.data(), sizeof(double), C, o);
        fwrite(&mul_calib.mean_error, sizeof(double), 1, o);
        fwrite(&focal0, sizeof(double), 1, o);
        const int tail[2] = { mul_calib.summary.termination_type, mul_calib.summary.num_iterations };
        fwrite(tail, sizeof(int), 2, o);
        fwrite(mapx.data(), sizeof(float), 16, o);
        fwrite(mapy.data(), sizeof(float), 16, o);
        fclose(o);
    } catch (const std::exception &e) {
        fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}

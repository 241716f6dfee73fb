// calibrate_from_images.cpp -- the reference's main.cpp from the image files on, through the C++ mirror (no OpenCV):
//   image list  ->  monocular_calib per camera (main.cpp:8-130: findCorner, calibrate, refinement pass on the remapped
//   chessboards with the flip rule, calibrate)  ->  MultiCalib(cameras, worlds) + calibrate (main.cpp:233-234)  ->  calib.yaml
// Images are 8-bit binary PGM files (P5); the list file holds one line per camera: "<n_frames> path_0 path_1 ..." with "-"
// for a frame the camera has no image of.  The viewer and the drawing calls of main.cpp are left out.
//   usage: calibrate_from_images list.txt calib.yaml [cols rows pitch]
#include <tscm/tscm_calib.hpp>

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

namespace {

struct Image { int w = 0, h = 0; std::vector<unsigned char> pix; bool ok() const { return w > 0; } };

Image read_pgm(const std::string &path)
{
    Image im;
    std::ifstream f(path.c_str(), std::ios::binary);
    std::string magic;
    int maxval = 0, w = 0, h = 0;
    if (!(f >> magic >> w >> h >> maxval) || magic != "P5" || maxval != 255 || w < 1 || h < 1) return im;
    f.get();
    im.pix.resize((size_t)w * h);
    f.read(reinterpret_cast<char *>(im.pix.data()), (std::streamsize)im.pix.size());
    if ((size_t)f.gcount() == im.pix.size()) { im.w = w; im.h = h; }
    return im;
}

// the board of the expected size, if findCorner found exactly that (main.cpp:33)
bool board_corners(const tscm::Chessboarder_t &found, tscm::Size board, std::vector<tscm::Point2d> &out)
{
    if (found.chessboard.size() != 1 || found.chessboard[0].rows != board.height || found.chessboard[0].cols != board.width) return false;
    out.clear();
    const tscm::IndexMat &m = found.chessboard[0];
    for (int u = 0; u < m.rows; ++u)
        for (int v = 0; v < m.cols; ++v) out.push_back(found.corners.p[m.at(u, v)]);
    return true;
}

// main.cpp:8-130
void monocular_calib(const std::vector<Image> &images, double size, tscm::Size board, const std::vector<tscm::Point3d> &worlds, tscm::TripleSphereCamera &cam)
{
    const int V = (int)images.size();
    std::vector<std::vector<tscm::Point2d> > pixels(V);
    std::vector<bool> has(V, false);
    tscm::Size img_size = { 0, 0 };
    for (int i = 0; i < V; ++i) {                                              // :24-50
        if (!images[i].ok()) continue;
        img_size.width = images[i].w; img_size.height = images[i].h;
        has[i] = board_corners(tscm::findCorner(images[i].pix.data(), images[i].w, images[i].h, images[i].w, 4, cam.device()), board, pixels[i]);
    }
    cam.calibrate(pixels, has, worlds, img_size, board);                       // :57
    for (int i = 0; i < V; ++i) {                                              // :59-126 refinement pass
        if (!has[i]) continue;
        std::vector<unsigned char> chess;
        const tscm::Size cs = cam.undistort_chessboard(images[i].pix.data(), images[i].w, images[i].h, images[i].w, 1, i, board, size, chess);
        if (cs.width == 0) continue;
        std::vector<tscm::Point2d> refined;
        if (board_corners(tscm::findCorner(chess.data(), cs.width, cs.height, cs.width, 4, cam.device()), board, refined)) {
            const tscm::Mat33 &Rt = cam.Rt(i);                                 // :92-105: back through [r1 r2 t] and project()
            std::vector<tscm::Point3d> P(refined.size());
            for (size_t k = 0; k < refined.size(); ++k) {
                const double x = refined[k].x - size, y = refined[k].y - size;
                P[k] = tscm::Point3d{ Rt.a[0] * x + Rt.a[1] * y + Rt.a[2], Rt.a[3] * x + Rt.a[4] * y + Rt.a[5], Rt.a[6] * x + Rt.a[7] * y + Rt.a[8] };
            }
            pixels[i] = cam.project(P);
        }
        auto grey = [&](double x, double y) { return (int)chess[(size_t)(int)y * cs.width + (int)x]; };          // :72-89 / :107-121 flip rule
        if (grey(size / 2, size / 2) + grey(size * 3 / 2, size * 3 / 2) > grey(size * 3 / 2, size / 2) + grey(size / 2, size * 3 / 2)) {
            const std::vector<tscm::Point2d> tmp = pixels[i];
            for (size_t k = 0; k < tmp.size(); ++k) pixels[i][k] = tmp[tmp.size() - 1 - k];
        }
    }
    const bool ok = cam.calibrate(pixels, has, worlds, img_size, board);       // :127
    int n_has = 0;
    for (int i = 0; i < V; ++i) n_has += has[i] ? 1 : 0;
    std::printf("%s, boards %d of %d, rmse %.4f px, fx %.3f fy %.3f cx %.3f cy %.3f xi %.4f lambda %.4f alpha %.4f\n", ok ? "converged" : "NOT converged", n_has, V,
                cam.summary.rmse, cam.intrinsic_[0], cam.intrinsic_[1], cam.intrinsic_[2], cam.intrinsic_[3], cam.intrinsic_[4], cam.intrinsic_[5], cam.intrinsic_[6]);
}

}  // namespace

int main(int argc, char **argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: %s list.txt calib.yaml [cols rows pitch]\n", argv[0]); return 2; }
    const tscm::Size board = { argc > 5 ? std::atoi(argv[3]) : 9, argc > 5 ? std::atoi(argv[4]) : 6 };
    const double size = argc > 5 ? std::atof(argv[5]) : 45.0;
    std::ifstream list(argv[1]);
    if (!list) { std::fprintf(stderr, "cannot open %s\n", argv[1]); return 2; }
    std::vector<std::vector<Image> > images;
    std::string line;
    while (std::getline(list, line)) {
        std::istringstream is(line);
        int n = 0;
        if (!(is >> n)) continue;
        std::vector<Image> cam(n);
        for (int i = 0; i < n; ++i) { std::string path; is >> path; if (path != "-") cam[i] = read_pgm(path); }
        images.push_back(cam);
    }
    try {
        std::vector<tscm::Point3d> worlds;                                     // main.cpp:12-18
        for (int u = 0; u < board.height; ++u)
            for (int v = 0; v < board.width; ++v) worlds.push_back(tscm::Point3d{ v * size, u * size, 0.0 });
        std::vector<tscm::TripleSphereCamera> cameras(images.size());
        for (size_t m = 0; m < images.size(); ++m) { std::printf("camera %zu: ", m); monocular_calib(images[m], size, board, worlds, cameras[m]); }
        if (cameras.size() > 1) {
            tscm::MultiCalib mul_calib(cameras, worlds);                       // main.cpp:233
            mul_calib.calibrate();                                             // main.cpp:234
            std::printf("%s  iterations %d  rmse %.4f px\n", mul_calib.summary.message, mul_calib.summary.num_iterations - 1, mul_calib.summary.rmse);
            std::printf("average reproject error: %.6f\n", mul_calib.mean_error);
            mul_calib.write_yaml(argv[2]);                                     // main.cpp:305-319
        } else if (cameras.size() == 1) {
            const double I3[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 }, t0[3] = { 0, 0, 0 };
            tscm::check(tscm_yaml_write(argv[2], 1, cameras[0].intrinsic_.data(), I3, t0));
        }
    } catch (const std::exception &e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}

// find_corners_demo.cpp -- the detection step of monocular_calib (main.cpp:24-50) through the C++ mirror:
// reads a binary PGM (P5, 8 bit), runs tscm::findCorner(img, 4) and prints the board corners in board order
// when exactly one board of the expected size was found (main.cpp:33).
//   usage: find_corners_demo image.pgm [cols rows]
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "tscm/tscm_calib.hpp"

static bool read_pgm(const char *path, std::vector<unsigned char> &pix, int &w, int &h)
{
    std::ifstream f(path, std::ios::binary);
    std::string magic;
    int maxval = 0;
    if (!(f >> magic >> w >> h >> maxval) || magic != "P5" || maxval != 255 || w < 1 || h < 1) return false;
    f.get();
    pix.resize((size_t)w * h);
    f.read(reinterpret_cast<char *>(pix.data()), (std::streamsize)pix.size());
    return (size_t)f.gcount() == pix.size();
}

int main(int argc, char **argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: %s image.pgm [cols rows]\n", argv[0]); return 2; }
    const int cols = argc > 3 ? std::atoi(argv[2]) : 9, rows = argc > 3 ? std::atoi(argv[3]) : 6;
    std::vector<unsigned char> img;
    int w = 0, h = 0;
    if (!read_pgm(argv[1], img, w, h)) { std::fprintf(stderr, "cannot read %s as an 8-bit binary PGM\n", argv[1]); return 2; }
    try {
        const tscm::Chessboarder_t found = tscm::findCorner(img.data(), w, h, w, 4);
        std::printf("candidates %zu boards %zu\n", found.corners.p.size(), found.chessboard.size());
        if (found.chessboard.size() != 1 || found.chessboard[0].rows != rows || found.chessboard[0].cols != cols) {
            std::printf("no %d x %d board\n", cols, rows);
            return 1;
        }
        const tscm::IndexMat &m = found.chessboard[0];
        for (int u = 0; u < m.rows; ++u)
            for (int v = 0; v < m.cols; ++v) std::printf("%.6f %.6f\n", found.corners.p[m.at(u, v)].x, found.corners.p[m.at(u, v)].y);
    } catch (const std::exception &e) {
        std::cerr << e.what() << "\n";
        return 3;
    }
    return 0;
}

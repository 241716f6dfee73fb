// Sanitizer harness for the CPU-only part of the library (tscm_io.cpp: calibration YAML and corner-list readers).
// Built by tests/test_io_sanitizers.py with g++ -fsanitize=address,undefined; feeds the parsers valid files, then
// thousands of deterministic mutations of them (byte flips, truncations, duplicated / deleted spans, huge numbers).
// Every call must return 0 or a negative error code -- never crash, leak or trip a sanitizer.
#include "tscm/tscm.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static std::string g_err;
int tscm_set_error(int code, const std::string &msg) { g_err = msg; return code; }   // the library's error sink (tscm_solver.hip)

static unsigned long long rng_state = 88172645463325252ull;
static unsigned rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (unsigned)(rng_state >> 11); }

static std::string mutate(const std::string &src)
{
    std::string s = src;
    const int ops = 1 + rnd() % 4;
    for (int k = 0; k < ops && !s.empty(); ++k) {
        const size_t pos = rnd() % s.size();
        switch (rnd() % 7) {
        case 0: s[pos] = (char)(rnd() & 0xff); break;
        case 1: s.resize(pos); break;
        case 2: s.erase(pos, 1 + rnd() % 16); break;
        case 3: s.insert(pos, s.substr(rnd() % s.size(), 1 + rnd() % 32)); break;
        case 4: s.insert(pos, "99999999999"); break;
        case 5: s.insert(pos, "-1"); break;
        default: s.insert(pos, rnd() & 1 ? "\n" : " ]: ["); break;
        }
    }
    return s;
}

int main(int argc, char **argv)
{
    const int rounds = argc > 1 ? std::atoi(argv[1]) : 3000;
    const char *dir = argc > 2 ? argv[2] : "/tmp";
    // ---- a valid YAML, written by the library itself
    const double intr[18] = { 470.1, 470.2, 640.5, 540.25, -0.1, 0.2, 0.55, 0, 0, 480, 481, 639, 541, 0.1, -0.2, 0.6, 0, 0 };
    const double R[18] = { 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, -1, 0, 1, 0, 0, 0, 0, 1 };
    const double t[6] = { 0, 0, 0, 100.5, -20.25, 3e-7 };
    size_t need = 0;
    if (tscm_yaml_format(2, intr, R, t, nullptr, 0, &need) != 0) return 2;
    std::vector<char> buf(need);
    if (tscm_yaml_format(2, intr, R, t, buf.data(), buf.size(), nullptr) != 0) return 2;
    const std::string yaml(buf.data());
    int n = 0;
    double I[9 * 4], T[12 * 4];
    if (tscm_yaml_parse(yaml.c_str(), 4, &n, I, T) != 0 || n != 2 || std::memcmp(I, intr, sizeof intr) != 0) return 3;
    for (int i = 0; i < rounds; ++i) {
        const std::string m = mutate(yaml);
        n = -1;
        const int rc = tscm_yaml_parse(m.c_str(), (int)(rnd() % 5), &n, I, T);
        if (rc > 0) return 4;
        if (rc == 0 && (n < 0 || n > 4)) { if (n > 4) continue; return 5; }
    }
    // ---- a valid corner list
    tscm_corner_set cs;
    std::memset(&cs, 0, sizeof cs);
    cs.n_cameras = 2; cs.n_boards = 3; cs.board_cols = 4; cs.board_rows = 3; cs.pitch = 45.0; cs.image_width = 1280; cs.image_height = 1080;
    std::vector<unsigned char> has = { 1, 0, 1, 1, 1, 0 };
    std::vector<double> pu(2 * 3 * 12), pv(2 * 3 * 12);
    for (size_t i = 0; i < pu.size(); ++i) { pu[i] = 0.5 * i + 1.0 / 3.0; pv[i] = 1000.0 - 0.25 * i; }
    cs.has = has.data(); cs.pix_u = pu.data(); cs.pix_v = pv.data();
    const std::string path = std::string(dir) + "/fuzz_corners.txt";
    if (tscm_corners_write(path.c_str(), &cs) != 0) return 6;
    tscm_corner_set back;
    if (tscm_corners_read(path.c_str(), &back) != 0) return 7;
    if (back.n_cameras != 2 || back.n_boards != 3 || std::memcmp(back.has, has.data(), 6) != 0 || std::memcmp(back.pix_u, pu.data(), 12 * sizeof(double)) != 0) return 8;
    tscm_corners_free(&back);
    std::string text;
    { std::FILE *f = std::fopen(path.c_str(), "r"); char tmp[4096]; size_t k; while ((k = std::fread(tmp, 1, sizeof tmp, f)) > 0) text.append(tmp, k); std::fclose(f); }
    const std::string mpath = std::string(dir) + "/fuzz_corners_mut.txt";
    for (int i = 0; i < rounds; ++i) {
        const std::string m = mutate(text);
        { std::FILE *f = std::fopen(mpath.c_str(), "w"); std::fwrite(m.data(), 1, m.size(), f); std::fclose(f); }
        tscm_corner_set c2;
        const int rc = tscm_corners_read(mpath.c_str(), &c2);
        if (rc > 0) return 9;
        if (rc == 0) tscm_corners_free(&c2);
        else if (c2.has || c2.pix_u || c2.pix_v) return 10;      // a failed read leaves nothing to free
    }
    std::remove(path.c_str());
    std::remove(mpath.c_str());
    // ---- chessboard structure recovery (tscm_boards.cpp) on random candidate sets: jittered grids, clutter, degenerate input
    for (int i = 0; i < rounds / 10; ++i) {
        const int gw = 3 + rnd() % 8, gh = 3 + rnd() % 6, clutter = rnd() % 12;
        std::vector<double> x, y, v1, v2;
        const double ang = (rnd() % 628) / 100.0, step = 20 + rnd() % 40;
        for (int r = 0; r < gh; ++r)
            for (int c = 0; c < gw; ++c) {
                const double jx = (rnd() % 200) / 100.0 - 1.0, jy = (rnd() % 200) / 100.0 - 1.0;
                x.push_back(300 + step * (c * std::cos(ang) - r * std::sin(ang)) + jx);
                y.push_back(300 + step * (c * std::sin(ang) + r * std::cos(ang)) + jy);
                v1.push_back(std::cos(ang)); v1.push_back(std::sin(ang)); v2.push_back(-std::sin(ang)); v2.push_back(std::cos(ang));
            }
        for (int k = 0; k < clutter; ++k) {
            x.push_back(rnd() % 900); y.push_back(rnd() % 900);
            const double a = (rnd() % 628) / 100.0;
            v1.push_back(std::cos(a)); v1.push_back(std::sin(a)); v2.push_back(rnd() & 1 ? 0.0 : -std::sin(a)); v2.push_back(rnd() & 1 ? 0.0 : std::cos(a));
        }
        if (rnd() % 7 == 0) { x.resize(rnd() % 9); y.resize(x.size()); v1.resize(2 * x.size()); v2.resize(2 * x.size()); }      // fewer than 9 candidates
        if (rnd() % 9 == 0) for (size_t k = 0; k < x.size(); ++k) { x[k] = 5; y[k] = 5; }                                       // all on one spot
        tscm_chessboards cb;
        const int rc = tscm_chessboards_from_corners((int)x.size(), x.data(), y.data(), v1.data(), v2.data(), &cb);
        if (rc != 0) return 11;
        for (int q = 0; q < cb.n_boards; ++q) {
            if (cb.cols[q] < cb.rows[q] || cb.rows[q] < 3) return 12;
            for (int k = cb.offset[q]; k < cb.offset[q + 1]; ++k) if (cb.cells[k] < 0 || cb.cells[k] >= (int)x.size()) return 13;
        }
        tscm_chessboards_free(&cb);
    }
    std::printf("fuzz_io: %d rounds per parser, clean\n", rounds);
    return 0;
}

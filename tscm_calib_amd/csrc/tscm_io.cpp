// tscm_io.cpp -- result I/O of the calibration: the YAML file main.cpp:305-319 writes
// with cv::FileStorage ("cam{i}" = 1x9 intrinsics, "Twc{i}" = 3x4 [R | t]) and
// EpipolarRectify/rectify.cpp:262-270 reads back.  OpenCV is not a dependency here: the
// writer restates FileStorage's YAML emitter for CV_64F matrices (doubles as "%.16e", integral
// values as "%d.", flow sequence wrapped at column 71 with the matrix indent), pinned byte for byte
// by the reference's own example EpipolarRectify/calib.yaml (tests/golden/reference_calib.yaml).
#include "tscm/tscm.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

int tscm_set_error(int code, const std::string &msg);   // tscm_solver.hip

namespace {

constexpr int kWrapMargin = 71;      // cv::FileStorage's line width for flow sequences

std::string format_double(double v)
{
    char buf[64];
    if (std::isnan(v)) return ".Nan";
    if (std::isinf(v)) return v < 0 ? "-.Inf" : ".Inf";
    if (std::fabs(v) < 2147483647.0 && (double)std::lrint(v) == v) {
        std::snprintf(buf, sizeof buf, "%ld.", std::lrint(v));
    } else {
        std::snprintf(buf, sizeof buf, "%.16e", v);
        for (char *p = buf; *p; ++p) if (*p == ',') *p = '.';      // decimal comma locales
    }
    return buf;
}

void write_matrix(std::string &out, const std::string &name, int rows, int cols, const double *data)
{
    out += name + ": !!opencv-matrix\n";
    out += "   rows: " + std::to_string(rows) + "\n";
    out += "   cols: " + std::to_string(cols) + "\n";
    out += "   dt: d\n";
    std::string line = "   data: [";
    const int indent = 7;
    for (int i = 0; i < rows * cols; ++i) {
        const std::string tok = format_double(data[i]);
        if (i > 0) line += ',';
        if ((int)line.size() + (int)tok.size() > kWrapMargin && (int)line.size() > indent) {
            out += line + "\n";
            line.assign(indent, ' ');
        } else {
            line += ' ';
        }
        line += tok;
    }
    out += line + " ]\n";
}

struct YamlMatrix { std::string name; int rows = 0, cols = 0; std::vector<double> data; };

std::string trim(const std::string &s)
{
    size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
    return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}

bool parse_double(const std::string &tok, double &v)
{
    if (tok == ".Nan" || tok == ".NaN" || tok == ".nan") { v = std::nan(""); return true; }
    if (tok == ".Inf" || tok == "+.Inf" || tok == ".inf") { v = INFINITY; return true; }
    if (tok == "-.Inf" || tok == "-.inf") { v = -INFINITY; return true; }
    char *end = nullptr;
    v = std::strtod(tok.c_str(), &end);
    return end != tok.c_str() && *end == '\0';
}

// the subset of YAML cv::FileStorage emits for top-level CV_64F matrices
bool parse_yaml(const std::string &text, std::vector<YamlMatrix> &mats, std::string &err)
{
    std::istringstream is(text);
    std::string line;
    YamlMatrix *cur = nullptr;
    bool in_data = false;
    std::string data_text;
    auto finish_data = [&]() -> bool {
        std::string body = data_text;
        for (char &c : body) if (c == ',' || c == '[' || c == ']') c = ' ';
        std::istringstream ts(body);
        std::string tok;
        while (ts >> tok) {
            double v;
            if (!parse_double(tok, v)) { err = "matrix " + cur->name + ": bad number '" + tok + "'"; return false; }
            cur->data.push_back(v);
        }
        if (cur->rows < 0 || cur->cols < 0 || (long long)cur->data.size() != (long long)cur->rows * cur->cols) { err = "matrix " + cur->name + ": data does not match rows x cols"; return false; }
        in_data = false; data_text.clear();
        return true;
    };
    while (std::getline(is, line)) {
        if (in_data) {
            data_text += " " + line;
            if (line.find(']') != std::string::npos && !finish_data()) return false;
            continue;
        }
        const std::string t = trim(line);
        if (t.empty() || t[0] == '%' || t[0] == '#' || t == "---" || t == "...") continue;
        const size_t colon = t.find(':');
        if (colon == std::string::npos) { err = "unexpected line: " + t; return false; }
        const std::string key = trim(t.substr(0, colon)), val = trim(t.substr(colon + 1));
        const bool top = !line.empty() && line[0] != ' ' && line[0] != '\t';
        if (top) {
            if (val.find("!!opencv-matrix") == std::string::npos) { cur = nullptr; continue; }     // other node kinds: skipped
            mats.emplace_back();
            cur = &mats.back();
            cur->name = key;
            continue;
        }
        if (!cur) continue;
        if (key == "rows") cur->rows = std::atoi(val.c_str());
        else if (key == "cols") cur->cols = std::atoi(val.c_str());
        else if (key == "dt") { if (val != "d" && val != "\"d\"") { err = "matrix " + cur->name + ": only dt: d (CV_64F) is supported"; return false; } }
        else if (key == "data") {
            in_data = true; data_text = val;
            if (val.find(']') != std::string::npos && !finish_data()) return false;
        }
    }
    if (in_data) { err = "unterminated data sequence"; return false; }
    return true;
}

}  // namespace

extern "C" int tscm_yaml_format(int n_cameras, const double *intr, const double *cam_R, const double *cam_t, char *buf, size_t buf_size,
                                size_t *needed)
{
    if (n_cameras < 0 || (n_cameras > 0 && (!intr || !cam_R || !cam_t))) return tscm_set_error(TSCM_E_INVALID, "NULL argument");
    std::string out = "%YAML:1.0\n---\n";
    for (int i = 0; i < n_cameras; ++i) {
        write_matrix(out, "cam" + std::to_string(i), 1, 9, intr + 9 * i);               // main.cpp:310
        const double *R = cam_R + 9 * i, *t = cam_t + 3 * i;
        const double T[12] = { R[0], R[1], R[2], t[0], R[3], R[4], R[5], t[1], R[6], R[7], R[8], t[2] };   // main.cpp:314-316
        write_matrix(out, "Twc" + std::to_string(i), 3, 4, T);
    }
    if (needed) *needed = out.size() + 1;
    if (buf) {
        if (buf_size < out.size() + 1) return tscm_set_error(TSCM_E_INVALID, "buffer too small");
        std::memcpy(buf, out.c_str(), out.size() + 1);
    }
    return 0;
}

extern "C" int tscm_yaml_write(const char *path, int n_cameras, const double *intr, const double *cam_R, const double *cam_t)
{
    if (!path) return tscm_set_error(TSCM_E_INVALID, "NULL path");
    size_t need = 0;
    int rc = tscm_yaml_format(n_cameras, intr, cam_R, cam_t, nullptr, 0, &need);
    if (rc) return rc;
    std::vector<char> buf(need);
    rc = tscm_yaml_format(n_cameras, intr, cam_R, cam_t, buf.data(), buf.size(), nullptr);
    if (rc) return rc;
    std::ofstream f(path, std::ios::binary);
    if (!f) return tscm_set_error(TSCM_E_INVALID, std::string("cannot open ") + path + " for writing");
    f.write(buf.data(), (std::streamsize)(need - 1));
    return f.good() ? 0 : tscm_set_error(TSCM_E_INVALID, std::string("write to ") + path + " failed");
}

extern "C" int tscm_yaml_parse(const char *text, int max_cameras, int *n_cameras, double *intr, double *Twc)
{
    if (!text || !n_cameras) return tscm_set_error(TSCM_E_INVALID, "NULL argument");
    std::vector<YamlMatrix> mats;
    std::string err;
    if (!parse_yaml(text, mats, err)) return tscm_set_error(TSCM_E_INVALID, "calibration YAML: " + err);
    int n = 0;
    for (;; ++n) {     // rectify.cpp:263-270 looks the nodes up by name
        const YamlMatrix *cam = nullptr, *twc = nullptr;
        for (const YamlMatrix &m : mats) {
            if (m.name == "cam" + std::to_string(n)) cam = &m;
            if (m.name == "Twc" + std::to_string(n)) twc = &m;
        }
        if (!cam && !twc) break;
        if (!cam || !twc) return tscm_set_error(TSCM_E_INVALID, "calibration YAML: camera " + std::to_string(n) + " lacks cam or Twc");
        // (a matrix node without a data: entry parses, with an empty payload)
        if (cam->data.size() != 9 || twc->data.size() != 12 || twc->rows != 3 || twc->cols != 4) return tscm_set_error(TSCM_E_INVALID, "calibration YAML: camera " + std::to_string(n) + ": expected 1x9 and 3x4");
        if (n < max_cameras) {
            if (intr) std::memcpy(intr + 9 * n, cam->data.data(), 9 * sizeof(double));
            if (Twc) std::memcpy(Twc + 12 * n, twc->data.data(), 12 * sizeof(double));
        }
    }
    *n_cameras = n;
    if (n > max_cameras && (intr || Twc)) return tscm_set_error(TSCM_E_INVALID, "calibration YAML holds " + std::to_string(n) + " cameras, room for " + std::to_string(max_cameras));
    return 0;
}

extern "C" int tscm_yaml_read(const char *path, int max_cameras, int *n_cameras, double *intr, double *Twc)
{
    if (!path) return tscm_set_error(TSCM_E_INVALID, "NULL path");
    std::ifstream f(path, std::ios::binary);
    if (!f) return tscm_set_error(TSCM_E_INVALID, std::string("cannot open ") + path);
    std::stringstream ss;
    ss << f.rdbuf();
    return tscm_yaml_parse(ss.str().c_str(), max_cameras, n_cameras, intr, Twc);
}

// ------------------------------------------------------------------------------------------------
// corner lists (tscm.h: "TSCM-CORNERS 1")
extern "C" void tscm_corners_free(tscm_corner_set *set)
{
    if (!set) return;
    std::free(set->has); std::free(set->pix_u); std::free(set->pix_v);
    set->has = nullptr; set->pix_u = set->pix_v = nullptr;
}

extern "C" int tscm_corners_write(const char *path, const tscm_corner_set *set)
{
    if (!path || !set) return tscm_set_error(TSCM_E_INVALID, "NULL argument");
    const int C = set->n_cameras, B = set->n_boards;
    if (C < 0 || B < 0 || set->board_cols < 1 || set->board_rows < 1 || set->board_cols > 4096 || set->board_rows > 4096 || ((size_t)C * B > 0 && (!set->has || !set->pix_u || !set->pix_v)))
        return tscm_set_error(TSCM_E_INVALID, "inconsistent corner set");
    const int n = set->board_cols * set->board_rows;
    std::FILE *f = std::fopen(path, "w");
    if (!f) return tscm_set_error(TSCM_E_INVALID, std::string("cannot open ") + path + " for writing");
    std::fprintf(f, "TSCM-CORNERS 1\ncameras %d boards %d cols %d rows %d pitch %.17g image %d %d\n", C, B, set->board_cols, set->board_rows,
                 set->pitch, set->image_width, set->image_height);
    for (int m = 0; m < C; ++m)
        for (int b = 0; b < B; ++b) {
            if (!set->has[(size_t)m * B + b]) continue;
            std::fprintf(f, "view %d %d\n", m, b);
            const size_t o = ((size_t)m * B + b) * n;
            for (int j = 0; j < n; ++j) std::fprintf(f, "%.17g %.17g\n", set->pix_u[o + j], set->pix_v[o + j]);
        }
    const bool ok = std::ferror(f) == 0;
    return (std::fclose(f) == 0 && ok) ? 0 : tscm_set_error(TSCM_E_INVALID, std::string("write to ") + path + " failed");
}

extern "C" int tscm_corners_read(const char *path, tscm_corner_set *set)
{
    if (!path || !set) return tscm_set_error(TSCM_E_INVALID, "NULL argument");
    std::memset(set, 0, sizeof(*set));
    std::FILE *f = std::fopen(path, "r");
    if (!f) return tscm_set_error(TSCM_E_INVALID, std::string("cannot open ") + path);
    auto bail = [&](const std::string &msg) { std::fclose(f); tscm_corners_free(set); return tscm_set_error(TSCM_E_INVALID, "corner list " + std::string(path) + ": " + msg); };
    int version = 0;
    if (std::fscanf(f, " TSCM-CORNERS %d", &version) != 1 || version != 1) return bail("not a TSCM-CORNERS 1 file");
    if (std::fscanf(f, " cameras %d boards %d cols %d rows %d pitch %lf image %d %d", &set->n_cameras, &set->n_boards, &set->board_cols, &set->board_rows,
                    &set->pitch, &set->image_width, &set->image_height) != 7) return bail("bad header");
    const int C = set->n_cameras, B = set->n_boards;
    if (C < 0 || B < 0 || set->board_cols < 1 || set->board_rows < 1 || set->board_cols > 4096 || set->board_rows > 4096 || (long long)C * B > (1LL << 28)
        || (long long)C * B * set->board_cols * set->board_rows > (1LL << 32)) return bail("bad dimensions");
    const int n = set->board_cols * set->board_rows;
    const size_t cb = (size_t)C * B;
    set->has = static_cast<unsigned char *>(std::calloc(cb ? cb : 1, 1));
    set->pix_u = static_cast<double *>(std::calloc(cb * n ? cb * n : 1, sizeof(double)));
    set->pix_v = static_cast<double *>(std::calloc(cb * n ? cb * n : 1, sizeof(double)));
    if (!set->has || !set->pix_u || !set->pix_v) { std::fclose(f); tscm_corners_free(set); return tscm_set_error(TSCM_E_NOMEM, "out of memory"); }
    for (;;) {
        int m = 0, b = 0;
        const int got = std::fscanf(f, " view %d %d", &m, &b);
        if (got == EOF) break;
        if (got != 2) return bail("expected 'view <camera> <board>'");
        if (m < 0 || m >= C || b < 0 || b >= B) return bail("view index out of range");
        if (set->has[(size_t)m * B + b]) return bail("view " + std::to_string(m) + " " + std::to_string(b) + " listed twice");
        set->has[(size_t)m * B + b] = 1;
        const size_t o = ((size_t)m * B + b) * n;
        for (int j = 0; j < n; ++j)
            if (std::fscanf(f, " %lf %lf", &set->pix_u[o + j], &set->pix_v[o + j]) != 2) return bail("view " + std::to_string(m) + " " + std::to_string(b) + ": expected " + std::to_string(n) + " corners");
    }
    std::fclose(f);
    return 0;
}

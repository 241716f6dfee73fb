// tscm_calib.hpp -- C++11 mirror of the reference's class interface for the accelerated path, on
// top of the C ABI (tscm.h).  Same class and member names, argument meaning and post-conditions as
// imuncle/TSCM_Calib's TripleSphereCamera (TS.h) / MultiCalib (multi_calib.h), with plain structs
// where the reference uses cv::Point / cv::Mat, so that a maintainer can move a call site over by
// changing types only.  Everything heavy runs on the MI355X through libtscm_hip.so; what stays on
// the host are the 3x3 conversions the reference also does on the host (update_param:
// multi_calib.h:42-57,104-108).  Failures throw std::runtime_error(tscm_last_error()).
//
//   reference                                              here
//   TripleSphereCamera::refinement       TS.cpp:247-282    TripleSphereCamera::refinement      -> tscm_solve_mono
//   TripleSphereCamera::estimate_focal   TS.cpp:110-168    TripleSphereCamera::estimate_focal  -> tscm_estimate_focal
//   TripleSphereCamera::estimate_extrinsic TS.cpp:170-203  TripleSphereCamera::estimate_extrinsic -> tscm_estimate_extrinsic (*)
//   loop Rt_ -> rt_                      TS.cpp:62-74      TripleSphereCamera::poses_from_Rt   -> tscm_poses_from_r1r2t
//   TripleSphereCamera::calibrate        TS.cpp:30-105     TripleSphereCamera::calibrate       (the four calls above + refinement)
//   TripleSphereCamera::project          TS.cpp:332-344    TripleSphereCamera::project (batch) -> tscm_project_points
//   get_unit_sphere_coordinate           TS.h:39-57        get_unit_sphere_coordinate (batch)  -> tscm_unproject_pixels
//   TripleSphereCamera::undistort        TS.cpp:284-306    TripleSphereCamera::undistort       -> tscm_build_maps
//   undistort_chessboard (table)         TS.cpp:308-330    undistort_chessboard_maps           -> tscm_build_maps
//   MultiCalib::MultiCalib               multi_calib.cpp:6-153    MultiCalib::MultiCalib       -> tscm_rig_init
//   MultiCalib::calibrate                multi_calib.cpp:155-283  MultiCalib::calibrate        -> tscm_solve_multi, tscm_reprojection_error
//   YAML output                          main.cpp:305-319         MultiCalib::write_yaml       -> tscm_yaml_write
// (*) deterministic planar PnP in place of cv::solvePnPRansac (see tscm.h)
#ifndef TSCM_CALIB_HPP
#define TSCM_CALIB_HPP

#include <tscm/tscm.h>

#include <cmath>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace tscm {

struct Point2d { double x, y; };
struct Point3d { double x, y, z; };
struct Size { int width, height; };
struct Mat33 { double a[9]; };       // row-major, cv::Mat_<double>(3,3) order

inline void check(int rc) { if (rc != 0) throw std::runtime_error(std::string("tscm: ") + tscm_last_error()); }

// cv::Rodrigues(r -> R), the conversion update_param() does on the host (multi_calib.h:43-45,105-106)
inline Mat33 rodrigues(const double r[3])
{
    const double th2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2], th = std::sqrt(th2);
    Mat33 R = { { 1, 0, 0, 0, 1, 0, 0, 0, 1 } };
    if (th < 2.220446049250313e-16) return R;
    const double c = std::cos(th), s = std::sin(th), c1 = 1.0 - c, k[3] = { r[0] / th, r[1] / th, r[2] / th };
    const double K[9] = { 0, -k[2], k[1], k[2], 0, -k[0], -k[1], k[0], 0 };
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R.a[3 * i + j] = (i == j ? c : 0.0) + c1 * k[i] * k[j] + s * K[3 * i + j];
    return R;
}

class TripleSphereCamera {
public:
    explicit TripleSphereCamera(int device = 0) : intrinsic_(9, 0.0), device_(device) {}

    // ---- state with the reference's names (TS.h:78-92) ----
    std::vector<double> intrinsic_;                       // fx fy cx cy xi lambda alpha b c
    std::vector<std::vector<double> > rt_;                // per image: angle-axis + t
    std::vector<Mat33> Rt_;                               // per image: [r1 r2 t]
    std::vector<bool> has_chessboard_;
    std::vector<std::vector<Point2d> > pixels_;

    const std::vector<double> &intrinsic() const { return intrinsic_; }
    bool has_chessboard(int j) const { return has_chessboard_[j]; }
    std::vector<bool> has_chessboard() const { return has_chessboard_; }
    const Mat33 &Rt(int j) const { return Rt_[j]; }
    const std::vector<std::vector<Point2d> > &pixels() const { return pixels_; }
    tscm_summary summary;                                 // of the last refinement()

    // TS.cpp:247-282: joint refinement of intrinsic_ and rt_[i]; returns termination_type == CONVERGENCE
    bool refinement(const std::vector<std::vector<Point2d> > &pixels, const std::vector<Point3d> &worlds,
                    const tscm_options *options = nullptr)
    {
        const int V = (int)pixels.size(), n = (int)worlds.size();
        std::vector<double> bxy(2 * (size_t)n), u, v, rt(6 * (size_t)V, 0.0);
        std::vector<int> vc, vb, vo, vn;
        for (int j = 0; j < n; ++j) { bxy[2 * j] = worlds[j].x; bxy[2 * j + 1] = worlds[j].y; }
        for (int i = 0; i < V; ++i) {
            if (!has_chessboard_[i]) continue;                           // TS.cpp:253
            vc.push_back(0); vb.push_back(i); vo.push_back((int)u.size()); vn.push_back((int)pixels[i].size());
            for (const Point2d &p : pixels[i]) { u.push_back(p.x); v.push_back(p.y); }
            std::memcpy(&rt[6 * (size_t)i], rt_[i].data(), 6 * sizeof(double));
        }
        tscm_problem P = tscm_problem();
        P.n_cameras = 1; P.n_boards = V; P.n_points = n; P.n_views = (int)vc.size();
        P.board_xy = bxy.data(); P.view_camera = vc.data(); P.view_board = vb.data(); P.view_offset = vo.data(); P.view_count = vn.data();
        P.obs_u = u.data(); P.obs_v = v.data(); P.intr = intrinsic_.data(); P.board_rt = rt.data(); P.mono = 1;
        tscm_options o;
        if (options) o = *options; else tscm_default_options(&o, 1);
        check(tscm_solve_mono(&P, &o, &summary));
        for (int i = 0; i < V; ++i) if (has_chessboard_[i]) rt_[i].assign(&rt[6 * (size_t)i], &rt[6 * (size_t)i] + 6);
        return summary.termination_type == TSCM_CONVERGENCE;             // TS.cpp:281
    }

    // TS.cpp:110-168: fx = fy = mean circle-fit focal length; 0 when no row is usable
    double estimate_focal(const std::vector<std::vector<Point2d> > &pixels, Size chessboard_num)
    {
        const int V = (int)pixels.size(), n = chessboard_num.width * chessboard_num.height;
        std::vector<double> u((size_t)V * n, 0.0), v((size_t)V * n, 0.0);
        std::vector<int> cnt(V);
        for (int k = 0; k < V; ++k) {
            cnt[k] = (int)pixels[k].size();
            for (size_t j = 0; j < pixels[k].size() && j < (size_t)n; ++j) { u[(size_t)k * n + j] = pixels[k][j].x; v[(size_t)k * n + j] = pixels[k][j].y; }
        }
        double focal = 0.0; int used = 0;
        check(tscm_estimate_focal(u.data(), v.data(), cnt.data(), V, chessboard_num.width, chessboard_num.height,
                                  intrinsic_[2], intrinsic_[3], device_, &focal, &used));
        intrinsic_[0] = intrinsic_[1] = focal;
        return focal;
    }

    // TS.cpp:170-203 (planar PnP per image with a board; fills Rt_)
    int estimate_extrinsic(const std::vector<std::vector<Point2d> > &pixels, const std::vector<Point3d> &worlds, Size chessboard_num)
    {
        const int V = (int)pixels.size(), n = (int)worlds.size();
        std::vector<double> u((size_t)V * n, 0.0), v((size_t)V * n, 0.0), W(3 * (size_t)n), M(9 * (size_t)V, 0.0);
        std::vector<int> cnt(V);
        for (int c = 0; c < n; ++c) { W[3 * c] = worlds[c].x; W[3 * c + 1] = worlds[c].y; W[3 * c + 2] = worlds[c].z; }
        for (int k = 0; k < V; ++k) {
            cnt[k] = has_chessboard_[k] ? (int)pixels[k].size() : 0;                // TS.cpp:174
            for (size_t j = 0; j < pixels[k].size() && j < (size_t)n; ++j) { u[(size_t)k * n + j] = pixels[k][j].x; v[(size_t)k * n + j] = pixels[k][j].y; }
        }
        int done = 0;
        check(tscm_estimate_extrinsic(intrinsic_.data(), u.data(), v.data(), cnt.data(), V, W.data(), n, chessboard_num.width, device_, M.data(), &done));
        Rt_.resize(V);
        for (int k = 0; k < V; ++k) if (cnt[k]) std::memcpy(Rt_[k].a, &M[9 * (size_t)k], sizeof(Rt_[k].a));
        return done;
    }

    // TS.cpp:30-105: initial guess (unless a previous calibration succeeded), poses, refinement
    bool calibrate(const std::vector<std::vector<Point2d> > &pixels, const std::vector<bool> &has_chessboard,
                   const std::vector<Point3d> &worlds, Size img_size, Size chessboard_num)
    {
        pixels_ = pixels;
        has_chessboard_ = has_chessboard;
        Rt_.assign(pixels.size(), Mat33());
        rt_.assign(pixels.size(), std::vector<double>());
        if (!has_init_guess_) {
            intrinsic_.assign(9, 0.0);
            intrinsic_[2] = img_size.width / 2 - 0.5;                                // :43-47
            intrinsic_[3] = img_size.height / 2 - 0.5;
            intrinsic_[6] = 0.5;
            if (estimate_focal(pixels, chessboard_num) == 0.0) return false;          // :48-50
        }
        estimate_extrinsic(pixels, worlds, chessboard_num);                           // :52
        poses_from_Rt();                                                              // :62-74
        const bool status = refinement(pixels, worlds);                               // :76-78
        if (status) has_init_guess_ = true;
        for (size_t i = 0; i < Rt_.size(); ++i) {                                     // :88-102: Rt_ from the refined rt_
            if (!has_chessboard_[i]) continue;
            const Mat33 R = rodrigues(rt_[i].data());
            for (int r = 0; r < 3; ++r) { Rt_[i].a[3 * r] = R.a[3 * r]; Rt_[i].a[3 * r + 1] = R.a[3 * r + 1]; Rt_[i].a[3 * r + 2] = rt_[i][3 + r]; }
        }
        return status;
    }
    bool has_init_guess_ = false;

    // TS.cpp:62-74
    void poses_from_Rt()
    {
        const int V = (int)Rt_.size();
        std::vector<double> M(9 * (size_t)V), rt(6 * (size_t)V, 0.0);
        std::vector<unsigned char> has(V);
        for (int i = 0; i < V; ++i) { std::memcpy(&M[9 * (size_t)i], Rt_[i].a, sizeof(Rt_[i].a)); has[i] = has_chessboard_[i] ? 1 : 0; }
        check(tscm_poses_from_r1r2t(M.data(), has.data(), V, rt.data()));
        rt_.resize(V);
        for (int i = 0; i < V; ++i) if (has[i]) rt_[i].assign(&rt[6 * (size_t)i], &rt[6 * (size_t)i] + 6);
    }

    // TS.cpp:332-344 for a batch of points
    std::vector<Point2d> project(const std::vector<Point3d> &P) const
    {
        std::vector<Point2d> out(P.size());
        if (!P.empty()) check(tscm_project_points(intrinsic_.data(), &P[0].x, (int)P.size(), device_, &out[0].x));
        return out;
    }
    Point2d project(const Point3d &P) const { return project(std::vector<Point3d>(1, P))[0]; }

    // TS.h:39-57 for a batch of pixels (transform == nullptr: identity)
    std::vector<Point3d> get_unit_sphere_coordinate(const std::vector<Point2d> &pixels, const Mat33 *transform = nullptr) const
    {
        std::vector<Point3d> out(pixels.size());
        if (pixels.empty()) return out;
        check(tscm_unproject_pixels(intrinsic_.data(), &pixels[0].x, (int)pixels.size(), device_, &out[0].x));
        if (transform)
            for (Point3d &p : out) {                                     // TS.h:54-55
                const double *T = transform->a, x = p.x, y = p.y, z = p.z;
                p.x = T[0] * x + T[1] * y + T[2] * z; p.y = T[3] * x + T[4] * y + T[5] * z; p.z = T[6] * x + T[7] * y + T[8] * z;
            }
        return out;
    }

    // TS.cpp:284-306; mapx / mapy: img_size.height x img_size.width floats (CV_32FC1 layout)
    void undistort(double fx, double fy, double cx, double cy, Size img_size, std::vector<float> &mapx, std::vector<float> &mapy, bool exact = true) const
    {
        tscm_map_desc d = tscm_map_desc();
        std::memcpy(d.intr, intrinsic_.data(), sizeof(d.intr));
        d.R[0] = d.R[4] = d.R[8] = 1.0;
        d.fx = fx; d.fy = fy; d.cx = cx; d.cy = cy;
        d.width = img_size.width; d.height = img_size.height; d.out_stride = img_size.width;
        const size_t n = (size_t)img_size.width * img_size.height;
        mapx.assign(n, 0.f); mapy.assign(n, 0.f);
        check(tscm_build_maps(&d, 1, device_, exact ? 1 : 0, mapx.data(), mapy.data(), n, nullptr));
    }

    // the table of undistort_chessboard(src, index, chessboard, chessboard_size), TS.cpp:308-328
    Size undistort_chessboard_maps(int index, Size chessboard, double chessboard_size, std::vector<float> &mapx, std::vector<float> &mapy,
                                   bool exact = true) const
    {
        Size img = { (int)((chessboard.width + 1) * chessboard_size), (int)((chessboard.height + 1) * chessboard_size) };
        mapx.clear(); mapy.clear();
        if (!has_chessboard_[index]) { img.width = img.height = 0; return img; }       // TS.cpp:311-312
        tscm_map_desc d = tscm_map_desc();
        std::memcpy(d.intr, intrinsic_.data(), sizeof(d.intr));
        std::memcpy(d.R, Rt_[index].a, sizeof(d.R));
        d.fx = d.fy = 1.0; d.cx = d.cy = chessboard_size;
        d.width = img.width; d.height = img.height; d.out_stride = img.width;
        const size_t n = (size_t)img.width * img.height;
        mapx.assign(n, 0.f); mapy.assign(n, 0.f);
        check(tscm_build_maps(&d, 1, device_, exact ? 1 : 0, mapx.data(), mapy.data(), n, nullptr));
        return img;
    }

    // undistort_chessboard(src, index, chessboard, chessboard_size) (TS.cpp:308-330): table + cv::remap(INTER_LINEAR) of an
    // 8-bit image with `channels` interleaved channels (1 or 3); dst gets img.height rows of img.width * channels bytes
    // (empty when the view has no board).  to_gray: BGR input, grey output (what findCorner does next, findCorner.cpp:9-10).
    Size undistort_chessboard(const unsigned char *src, int width, int height, int stride, int channels, int index, Size chessboard, double chessboard_size,
                              std::vector<unsigned char> &dst, bool to_gray = false) const
    {
        std::vector<float> mapx, mapy;
        const Size img = undistort_chessboard_maps(index, chessboard, chessboard_size, mapx, mapy);
        dst.clear();
        if (img.width == 0) return img;
        const int out_ch = (to_gray || channels == 1) ? 1 : channels;
        dst.assign((size_t)img.width * img.height * out_ch, 0);
        check(tscm_remap(src, width, height, stride, channels, mapx.data(), mapy.data(), img.width, img.height, img.width, to_gray ? 1 : 0, device_, dst.data(),
                         img.width * out_ch));
        return img;
    }

    int device() const { return device_; }

private:
    int device_;
};

// multi_calib.h:8-83
class MultiCalib_camera {
public:
    MultiCalib_camera() : intrinsic_(9, 0.0), rt_(6, 0.0), is_initial_(false) {}
    std::vector<double> intrinsic_, rt_;
    bool is_initial() const { return is_initial_; }
    const Mat33 &R() const { return R_; }
    const double *t() const { return t_; }
    double fx() const { return intrinsic_[0]; }
    double fy() const { return intrinsic_[1]; }
    double cx() const { return intrinsic_[2]; }
    double cy() const { return intrinsic_[3]; }
    double xi() const { return intrinsic_[4]; }
    double lamda() const { return intrinsic_[5]; }
    double alpha() const { return intrinsic_[6]; }
    double b() const { return intrinsic_[7]; }
    double c() const { return intrinsic_[8]; }
    const std::vector<std::vector<Point2d> > &pixels() const { return pixel_coordinates_; }
    bool has_chessboard(int j) const { return has_chessboard_[j]; }
    void update_param() { R_ = rodrigues(rt_.data()); std::memcpy(t_, &rt_[3], sizeof(t_)); }       // multi_calib.h:42-57
private:
    friend class MultiCalib;
    std::vector<bool> has_chessboard_;
    std::vector<std::vector<Point2d> > pixel_coordinates_;
    Mat33 R_;
    double t_[3];
    bool is_initial_;
};

// multi_calib.h:85-117
class MultiCalib_chessboard {
public:
    MultiCalib_chessboard() : rt_(6, 0.0), is_initial_(false) {}
    std::vector<double> rt_;
    bool is_initial() const { return is_initial_; }
    const Mat33 &R() const { return R_; }
    const double *t() const { return t_; }
    void update_param() { R_ = rodrigues(rt_.data()); std::memcpy(t_, &rt_[3], sizeof(t_)); }       // multi_calib.h:104-108
private:
    friend class MultiCalib;
    Mat33 R_;
    double t_[3];
    bool is_initial_;
};

// multi_calib.h:119-129
class MultiCalib {
public:
    // multi_calib.cpp:6-153: rig chaining and board-pose selection on the device
    MultiCalib(const std::vector<TripleSphereCamera> &cameras, const std::vector<Point3d> &worlds, int device = 0)
        : worlds_(worlds), mean_error(0.0), device_(device)
    {
        const int C = (int)cameras.size(), B = C ? (int)cameras[0].has_chessboard_.size() : 0, n = (int)worlds.size();
        std::vector<double> W(3 * (size_t)n), I(9 * (size_t)C), Rt(9 * (size_t)C * B, 0.0), pu((size_t)C * B * n, 0.0), pv((size_t)C * B * n, 0.0);
        std::vector<unsigned char> has((size_t)C * B, 0);
        for (int c = 0; c < n; ++c) { W[3 * c] = worlds[c].x; W[3 * c + 1] = worlds[c].y; W[3 * c + 2] = worlds[c].z; }
        for (int m = 0; m < C; ++m) {
            std::memcpy(&I[9 * (size_t)m], cameras[m].intrinsic_.data(), 9 * sizeof(double));
            for (int j = 0; j < B; ++j) {
                if (!cameras[m].has_chessboard_[j]) continue;
                has[(size_t)m * B + j] = 1;
                std::memcpy(&Rt[9 * ((size_t)m * B + j)], cameras[m].Rt_[j].a, 9 * sizeof(double));
                for (int c = 0; c < n; ++c) { pu[((size_t)m * B + j) * n + c] = cameras[m].pixels_[j][c].x; pv[((size_t)m * B + j) * n + c] = cameras[m].pixels_[j][c].y; }
            }
        }
        std::vector<double> cR(9 * (size_t)C), ct(3 * (size_t)C), crt(6 * (size_t)C), bR(9 * (size_t)B), bt(3 * (size_t)B), brt(6 * (size_t)B);
        std::vector<unsigned char> init(B);
        tscm_rig_input in = { C, B, n, W.data(), I.data(), has.data(), Rt.data(), pu.data(), pv.data() };
        tscm_rig_result out = tscm_rig_result();
        out.cam_R = cR.data(); out.cam_t = ct.data(); out.cam_rt = crt.data();
        out.board_R = bR.data(); out.board_t = bt.data(); out.board_rt = brt.data(); out.board_initial = init.data();
        check(tscm_rig_init(&in, device_, &out));
        cameras_.resize(C); chessboards_.resize(B);
        for (int m = 0; m < C; ++m) {                                    // MultiCalib_camera(camera, R, t), multi_calib.h:10-37
            MultiCalib_camera &cam = cameras_[m];
            cam.intrinsic_ = cameras[m].intrinsic_;
            cam.has_chessboard_ = cameras[m].has_chessboard_;
            cam.pixel_coordinates_ = cameras[m].pixels_;
            std::memcpy(cam.R_.a, &cR[9 * (size_t)m], sizeof(cam.R_.a)); std::memcpy(cam.t_, &ct[3 * (size_t)m], sizeof(cam.t_));
            cam.rt_.assign(&crt[6 * (size_t)m], &crt[6 * (size_t)m] + 6);
            cam.is_initial_ = true;
        }
        for (int j = 0; j < B; ++j) {                                    // multi_calib.h:88-97
            MultiCalib_chessboard &cb = chessboards_[j];
            cb.is_initial_ = init[j] != 0;
            if (!cb.is_initial_) continue;
            std::memcpy(cb.R_.a, &bR[9 * (size_t)j], sizeof(cb.R_.a)); std::memcpy(cb.t_, &bt[3 * (size_t)j], sizeof(cb.t_));
            cb.rt_.assign(&brt[6 * (size_t)j], &brt[6 * (size_t)j] + 6);
        }
    }

    // Several GPUs, one process per GPU (no counterpart in the reference): rank / world of this process and the
    // communicator made from rank 0's tscm_comm_unique_id (tscm_comm_create).  calibrate() then shards the frames.
    void set_sharding(int rank, int world, tscm_comm *comm) { rank_ = rank; world_ = world; comm_ = comm; }

    // multi_calib.cpp:155-283: joint LM, write-back (update_param), reprojection-error report
    void calibrate(const tscm_options *options = nullptr)
    {
        const int C = (int)cameras_.size(), B = (int)chessboards_.size(), n = (int)worlds_.size();
        std::vector<double> bxy(2 * (size_t)n), u, v, crt(6 * (size_t)C), I(9 * (size_t)C), brt(6 * (size_t)B, 0.0);
        std::vector<int> vc, vb, vo, vn;
        std::vector<unsigned char> cc(C, 0);
        for (int j = 0; j < n; ++j) { bxy[2 * j] = worlds_[j].x; bxy[2 * j + 1] = worlds_[j].y; }
        for (int m = 0; m < C; ++m)                                      // :162-207: camera m, board i, corner j
            for (int i = 0; i < B; ++i) {
                if (!chessboards_[i].is_initial() || cameras_[m].pixel_coordinates_[i].empty()) continue;
                vc.push_back(m); vb.push_back(i); vo.push_back((int)u.size()); vn.push_back((int)cameras_[m].pixel_coordinates_[i].size());
                for (const Point2d &p : cameras_[m].pixel_coordinates_[i]) { u.push_back(p.x); v.push_back(p.y); }
            }
        for (int m = 0; m < C; ++m) { std::memcpy(&crt[6 * (size_t)m], cameras_[m].rt_.data(), 6 * sizeof(double)); std::memcpy(&I[9 * (size_t)m], cameras_[m].intrinsic_.data(), 9 * sizeof(double)); }
        for (int i = 0; i < B; ++i) if (chessboards_[i].is_initial()) std::memcpy(&brt[6 * (size_t)i], chessboards_[i].rt_.data(), 6 * sizeof(double));
        if (C) cc[0] = 1;                                                // SetParameterBlockConstant(cameras_[0].rt_), :186
        tscm_problem P = tscm_problem();
        P.n_cameras = C; P.n_boards = B; P.n_points = n; P.n_views = (int)vc.size();
        P.board_xy = bxy.data(); P.view_camera = vc.data(); P.view_board = vb.data(); P.view_offset = vo.data(); P.view_count = vn.data();
        P.obs_u = u.data(); P.obs_v = v.data(); P.cam_rt = crt.data(); P.intr = I.data(); P.board_rt = brt.data();
        P.cam_pose_constant = cc.data(); P.mono = 0;
        tscm_options o;
        if (options) o = *options; else tscm_default_options(&o, 0);
        if (comm_) {
            // frame-sharded over the ranks of set_sharding(): every rank builds this same problem, keeps the boards it owns
            // and ends with ALL parameters updated (tscm.h, "multi-GPU").  (A one-rank communicator is legal: the library
            // then solves as if there were none, unless the options carry TSCM_EXEC_KEEP_SINGLE_RANK_COMM.)
            tscm_solver *s = nullptr;
            check(tscm_solver_create_sharded(&P, device_, rank_, world_, &s));
            int rc = tscm_solver_set_comm(s, comm_);
            if (rc == 0) rc = tscm_solver_solve(s, &o, &summary);
            tscm_solver_destroy(s);
            check(rc);
        } else {
            check(tscm_solve_multi(&P, &o, &summary));
        }
        for (int m = 0; m < C; ++m) {                                    // :221-226
            cameras_[m].rt_.assign(&crt[6 * (size_t)m], &crt[6 * (size_t)m] + 6);
            cameras_[m].intrinsic_.assign(&I[9 * (size_t)m], &I[9 * (size_t)m] + 9);
            if (cameras_[m].is_initial()) cameras_[m].update_param();
        }
        for (int i = 0; i < B; ++i) {                                    // :227-232
            if (!chessboards_[i].is_initial()) continue;
            chessboards_[i].rt_.assign(&brt[6 * (size_t)i], &brt[6 * (size_t)i] + 6);
            chessboards_[i].update_param();
        }
        camera_error.assign(C, 0.0);                                     // :233-283
        double rmse = 0.0;
        check(tscm_reprojection_error(&P, device_, camera_error.data(), &mean_error, &rmse));
    }

    // main.cpp:305-319
    void write_yaml(const std::string &path) const
    {
        const int C = (int)cameras_.size();
        std::vector<double> I(9 * (size_t)C), R(9 * (size_t)C), t(3 * (size_t)C);
        for (int m = 0; m < C; ++m) {
            std::memcpy(&I[9 * (size_t)m], cameras_[m].intrinsic_.data(), 9 * sizeof(double));
            std::memcpy(&R[9 * (size_t)m], cameras_[m].R().a, 9 * sizeof(double));
            std::memcpy(&t[3 * (size_t)m], cameras_[m].t(), 3 * sizeof(double));
        }
        check(tscm_yaml_write(path.c_str(), C, I.data(), R.data(), t.data()));
    }

    std::vector<MultiCalib_camera> cameras_;
    std::vector<MultiCalib_chessboard> chessboards_;
    std::vector<Point3d> worlds_;
    int rank_ = 0, world_ = 1;                // set_sharding()
    tscm_comm *comm_ = nullptr;
    tscm_summary summary;                     // BriefReport data of the solve (:218)
    std::vector<double> camera_error;         // per-camera mean pixel error (:281)
    double mean_error;                        // "average reproject error" (:283)

private:
    int device_;
};

// ---- corner detection: findCorner(img, sigma) (DetectCorner/findCorner.cpp:7-101) -------------------------------------
// Same result structure as the reference (Corner_t / Chessboarder_t, chessboard = matrices of indices into corners.p),
// grey 8-bit image instead of cv::Mat.  Board members carry their sub-pixel position (findCorner.cpp:84-97).
struct Corner_t {
    std::vector<Point2d> p, v1, v2;
    std::vector<double> score;
};
struct IndexMat {
    int rows, cols;
    std::vector<unsigned short> data;                 // row-major, CV_16U like the reference
    unsigned short at(int r, int c) const { return data[(size_t)r * cols + c]; }
};
struct Chessboarder_t {
    Corner_t corners;
    std::vector<IndexMat> chessboard;
};

inline Chessboarder_t findCorner(const unsigned char *gray, int width, int height, int stride, int sigma, int device = 0)
{
    tscm_corner_candidates c;
    check(tscm_detect_corners(gray, width, height, stride, sigma, 0.01, device, &c));
    tscm_chessboards b;
    const int rc = tscm_chessboards_from_corners(c.n, c.x, c.y, c.v1, c.v2, &b);
    if (rc != 0) { tscm_corner_candidates_free(&c); check(rc); }
    Chessboarder_t out;
    for (int i = 0; i < c.n; ++i) {
        out.corners.p.push_back(Point2d{ c.x[i], c.y[i] });
        out.corners.v1.push_back(Point2d{ c.v1[2 * i], c.v1[2 * i + 1] });
        out.corners.v2.push_back(Point2d{ c.v2[2 * i], c.v2[2 * i + 1] });
        out.corners.score.push_back(c.score[i]);
    }
    for (int q = 0; q < b.n_boards; ++q) {
        IndexMat m;
        m.rows = b.rows[q]; m.cols = b.cols[q];
        for (int k = b.offset[q]; k < b.offset[q + 1]; ++k) {
            const int idx = b.cells[k];
            m.data.push_back((unsigned short)idx);
            out.corners.p[(size_t)idx] = Point2d{ c.sub[2 * idx], c.sub[2 * idx + 1] };
        }
        out.chessboard.push_back(m);
    }
    tscm_chessboards_free(&b);
    tscm_corner_candidates_free(&c);
    return out;
}

}  // namespace tscm
#endif

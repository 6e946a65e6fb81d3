// include/LaseCamCalCeres.h — drop-in replacement of the reference header of the same name
// (MegviiRobot/CamLaserCalibraTool, include/LaseCamCalCeres.h:11-29).
//
// Same public surface — `struct Oberserve` (sic) and the four free functions with identical
// signatures and default arguments — so main/calibr_offline.cpp:166-170 and
// main/calibr_simulation.cpp:130 compile unchanged.  The two functions on the hot path,
//     CamLaserCalClosedSolution()   (reference body: src/LaseCamCalCeres.cpp:112-203)
//     CamLaserCalibration()         (reference body: src/LaseCamCalCeres.cpp:213-383)
// and the scan front-end step next to it,
//     LineFittingCeres()            (reference body: src/LaseCamCalCeres.cpp:401-433)
// are defined here as thin host adapters: flatten std::vector<Oberserve> -> C-ABI (clc.h) ->
// hand-written HIP kernels on the MI355X.  CalibrationTool_SavePlanePoints() is out of scope
// (debug file dump) and stays declared only: it keeps coming from the reference's own
// translation unit.  For many scans at once use clc_line_fit_batched() directly.
// Link with libclc_hip.so; see INTEGRATION.md.
//
// Conventions kept from the reference: `obs` by value; `Tlc` / `Tcl` are in/out 4x4 matrices
// (initial guess in, result out; :215-219, :311-314); no return value, no exceptions by design,
// diagnostics on stdout (closed-form Tlc :202, solver summary :309, singular values / null
// space / "recover chi2" :365-381).  An error from the GPU library (e.g. no device) is
// reported on stderr and leaves the output matrix untouched.
#ifndef PROJECT_LASECAMCALCERES_H
#define PROJECT_LASECAMCALCERES_H

#include <Eigen/Core>
#include <Eigen/Geometry>

#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <mutex>
#include <thread>
#include <vector>

#include "clc.h"

struct Oberserve
{
    EIGEN_MAKE_ALIGNED_OPERATOR_NEW
    Oberserve()
    {
        tagPose_Qca = Eigen::Quaterniond(1,0,0,0);
        tagPose_tca = Eigen::Vector3d::Zero();
    }

    Eigen::Quaterniond tagPose_Qca;    // tag orientation in the camera frame at this instant
    Eigen::Vector3d tagPose_tca;
    std::vector<Eigen::Vector3d> points;           // laser points on the board in this scan
    std::vector<Eigen::Vector3d> points_on_line;   // the (two) points on the fitted scan line
};

// unchanged, CPU, provided by the reference's own src/LaseCamCalCeres.cpp
void CalibrationTool_SavePlanePoints(const std::vector<Oberserve> obs, const Eigen::Matrix4d Tcl, const std::string path);

namespace clc_adapter {

// Grow-only page-locked host buffer (clc_pinned_alloc): what the scan points are gathered into, so that the upload is a
// plain DMA.  Process-wide like the shared context; used under its mutex; never freed (see create_shared_handle).
struct PinnedBuf {
    double* p; size_t cap;
    PinnedBuf() : p(NULL), cap(0) {}
    double* ensure(size_t n)
    {
        if (n > cap) {
            if (p) clc_pinned_free(p);
            cap = n + n / 4 + 1024;
            p = (double*)clc_pinned_alloc(cap * sizeof(double));
            if (!p) cap = 0;
        }
        return p;
    }
};
inline PinnedBuf& pinned_points() { static PinnedBuf b; return b; }
inline PinnedBuf& pinned_points_on_line() { static PinnedBuf b; return b; }

struct Flat {  // pose-major CSR form of std::vector<Oberserve>; pts / ptl point into the pinned buffers above
    std::vector<double> tag_q, tag_t;
    std::vector<int64_t> pts_off, ptl_off;
    const double* pts; const double* ptl;
    Flat() : pts(NULL), ptl(NULL) {}
};

// Gathers the scans (call with the shared context locked: the pinned buffers are process-wide).
inline Flat flatten(const std::vector<Oberserve>& obs)
{
    // Eigen::Vector3d is three contiguous doubles and std::vector is contiguous: one block copy per scan, on up to four
    // threads (10^6 points element by element through operator() took ~25 ms; one thread of memcpy ~5; four ~1.5)
    static_assert(sizeof(Eigen::Vector3d) == 3 * sizeof(double), "Eigen::Vector3d is 3 contiguous doubles");
    Flat f;
    const size_t P = obs.size();
    f.tag_q.reserve(4 * P); f.tag_t.reserve(3 * P);
    f.pts_off.reserve(P + 1); f.ptl_off.reserve(P + 1);
    f.pts_off.push_back(0); f.ptl_off.push_back(0);
    size_t a = 0, b = 0;
    for (size_t i = 0; i < P; ++i) {
        const Oberserve& o = obs[i];
        f.tag_q.push_back(o.tagPose_Qca.w()); f.tag_q.push_back(o.tagPose_Qca.x());
        f.tag_q.push_back(o.tagPose_Qca.y()); f.tag_q.push_back(o.tagPose_Qca.z());
        for (int k = 0; k < 3; ++k) f.tag_t.push_back(o.tagPose_tca(k));
        a += o.points.size(); b += o.points_on_line.size();
        f.pts_off.push_back((int64_t)a);
        f.ptl_off.push_back((int64_t)b);
    }
    double* pts = pinned_points().ensure(3 * a);
    double* ptl = pinned_points_on_line().ensure(3 * b);
    if ((a > 0 && !pts) || (b > 0 && !ptl)) { std::cerr << "[clc] pinned host allocation failed" << std::endl; f.pts_off.assign(P + 1, 0); f.ptl_off.assign(P + 1, 0); return f; }
    struct Copy {
        static void run(const std::vector<Oberserve>* obs, const Flat* f, double* pts, double* ptl, size_t i0, size_t i1)
        {
            for (size_t i = i0; i < i1; ++i) {
                const Oberserve& o = (*obs)[i];
                if (!o.points.empty()) std::memcpy(pts + 3 * f->pts_off[i], &o.points[0], o.points.size() * sizeof(Eigen::Vector3d));
                if (!o.points_on_line.empty()) std::memcpy(ptl + 3 * f->ptl_off[i], &o.points_on_line[0], o.points_on_line.size() * sizeof(Eigen::Vector3d));
            }
        }
    };
    const size_t bytes = 24 * (a + b);
    const size_t T = (bytes > (8u << 20) && P >= 8) ? 4 : 1;
    if (T == 1) {
        Copy::run(&obs, &f, pts, ptl, 0, P);
    } else {
        std::vector<std::thread> th;
        for (size_t t = 1; t < T; ++t) th.push_back(std::thread(&Copy::run, &obs, &f, pts, ptl, P * t / T, P * (t + 1) / T));
        Copy::run(&obs, &f, pts, ptl, 0, P / T);
        for (size_t t = 0; t < th.size(); ++t) th[t].join();
    }
    f.pts = pts; f.ptl = ptl;
    return f;
}

// Eigen::Quaterniond(Matrix3d) on plain doubles (the conversion the reference does at :215).
inline void rot_to_quat_xyzw(const double m[9], double q[4])
{
    double t = m[0] + m[4] + m[8];
    if (t > 0.0) {
        t = std::sqrt(t + 1.0); q[3] = 0.5 * t; t = 0.5 / t;
        q[0] = (m[7] - m[5]) * t; q[1] = (m[2] - m[6]) * t; q[2] = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[4 * i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(m[4 * i] - m[4 * j] - m[4 * k] + 1.0);
        q[i] = 0.5 * t; t = 0.5 / t;
        q[3] = (m[3 * k + j] - m[3 * j + k]) * t;
        q[j] = (m[3 * j + i] + m[3 * i + j]) * t;
        q[k] = (m[3 * k + i] + m[3 * i + k]) * t;
    }
}

inline int device_index()
{
    const char* e = std::getenv("CLC_DEVICE");
    return e ? std::atoi(e) : 0;
}

// One solver context per process, created on first use and shared by every call below: creating a context
// (stream, device and pinned buffers) costs milliseconds, and main/calibr_offline.cpp:124 calls LineFittingCeres once
// per scan.  It is deliberately never destroyed — the HIP runtime may already be gone when static destructors run.
// The reference calls this API from its main thread only; the mutex makes concurrent callers queue up instead of
// sharing the context's stream.
inline clc_handle* create_shared_handle()
{
    clc_handle* h = NULL;
    if (clc_create(&h, device_index()) != CLC_OK) { std::cerr << "[clc] " << clc_last_error() << std::endl; h = NULL; }
    return h;
}
inline clc_handle* shared_handle()
{
    static clc_handle* h = create_shared_handle();
    return h;
}
inline std::mutex& shared_mutex()
{
    static std::mutex m;
    return m;
}
struct Handle {  // the shared context, locked for the duration of one API call
    std::lock_guard<std::mutex> lock;
    clc_handle* h;
    Handle() : lock(shared_mutex()), h(shared_handle()) {}
};

}  // namespace clc_adapter

// Robust line fit of one scan; Line = (m0, m1) of m0 x + m1 y + 1 = 0, in/out.
// Reference: src/LaseCamCalCeres.cpp:401-433 (Cauchy 0.05, DENSE_QR, 10 iterations).
inline void LineFittingCeres(const std::vector<Eigen::Vector3d> Points, Eigen::Vector2d & Line)
{
    std::vector<double> xy;
    xy.reserve(2 * Points.size());
    for (size_t i = 0; i < Points.size(); ++i) { xy.push_back(Points[i](0)); xy.push_back(Points[i](1)); }  // :412
    const int64_t off[2] = {0, (int64_t)Points.size()};
    double line[2] = {Line(0), Line(1)};                                                                     // :403
    clc_adapter::Handle H;
    if (!H.h) return;
    clc_options opt; clc_line_options_default(&opt);                                                         // :416,:424-425
    if (clc_line_fit_batched(H.h, &opt, xy.data(), off, 1, line, NULL) != CLC_OK) {
        std::cerr << "[clc] " << clc_last_error() << std::endl;
        return;
    }
    Line(0) = line[0];                                                                                       // :430-431
    Line(1) = line[1];
}

namespace clc_adapter {

// The observations of one calibration run resident on the GPU.  The pose-major data of std::vector<Oberserve> (tag
// poses + scan points, 24 bytes per point) crosses PCIe ONCE (clc_store_observations); the residual blocks of closed
// form, refinement and analysis pass — the sequence of main/calibr_offline.cpp:166-170 — are then selected on the device
// (clc_select_observations) instead of being re-flattened into 64-byte records and re-uploaded per call.  The two free
// functions below, which keep the reference's signatures, are one-call sessions; a caller that owns its main() can hold
// a Session across the calls:
//     clc_adapter::Session run(obs);
//     run.ClosedSolution(Tlc_initial);  Eigen::Matrix4d Tcl = Tlc_initial.inverse();  run.Calibration(Tcl, false);
class Session {
 public:
    // `obs` is read here; it is read AGAIN only if another Session (or one of the free functions) has replaced the stored scans
    // on the shared context before a later call of this one — keep it alive for as long as the Session (no copy is made: the
    // reference-size inputs are small, a C2-size one is 53 MB).
    // A temporary cannot be kept alive: `Session s(load_obs());` does not compile (the re-store would read freed memory).
    explicit Session(std::vector<Oberserve>&&) = delete;
    explicit Session(const std::vector<Oberserve>& obs) : obs_(&obs), ok_(false), generation_(-1)
    {
        Handle H;  // locked first: flatten gathers into the process-wide pinned buffers
        if (!H.h) return;
        ok_ = store(H.h);
    }

    // The outcome of the LAST call (constructor, ClosedSolution, Calibration): false when it could not run or finish (no device, a
    // store / select / solve / analysis error — the message went to std::cerr), true again after the next call that succeeds.  The
    // reference's functions return void, so a caller that wants to know asks here.  Tlc / Tcl are left unchanged by a call that
    // failed before its result (a failure of the analysis pass alone leaves the refined Tcl in place).
    bool ok() const { return ok_; }

    // Closed-form initial guess of Tlc (camera -> laser).  Reference: src/LaseCamCalCeres.cpp:112-203.
    void ClosedSolution(Eigen::Matrix4d &Tlc)
    {
        Handle H;
        if (!H.h || !mine(H.h)) { ok_ = false; return; }
        if (clc_select_observations(H.h, /*linefit=*/1, /*boundary=*/0, NULL) != CLC_OK) {  // points_on_line only, :143
            std::cerr << "[clc] " << clc_last_error() << std::endl;
            ok_ = false;
            return;
        }
        double T[16]; int unobservable = 0;
        const int rc = clc_closed_form(H.h, T, &unobservable, NULL);
        if (unobservable) {  // :173-178 — printed whatever follows: the reference goes on to solve and return a Tlc
            std::cout <<std::endl<< "~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~" << std::endl;
            std::cout << " Notice Notice Notice: system unobservable !!!!!!!" << std::endl;
            std::cout << "~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~~" << std::endl<<std::endl;
        }
        if (rc != CLC_OK) {  // only a non-finite result (the pivoted LDLT / SVD back end is defined for singular input)
            std::cerr << "[clc] " << clc_last_error() << std::endl;
            ok_ = false;
            return;
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) Tlc(i, j) = T[4 * i + j];  // :198-200
        std::cout <<"------- Closed-form solution Tlc: -------\n" << Tlc <<std::endl;      // :202
        ok_ = true;
    }

    // Nonlinear refinement of Tcl (laser -> camera).  Reference: src/LaseCamCalCeres.cpp:213-383.
    void Calibration(Eigen::Matrix4d &Tcl, bool use_linefitting_data = true, bool use_boundary_constraint = false)
    {
        Handle H;
        if (!H.h || !mine(H.h)) { ok_ = false; return; }
        int64_t n_rec = 0;
        if (clc_select_observations(H.h, use_linefitting_data, use_boundary_constraint, &n_rec) != CLC_OK) {
            std::cerr << "[clc] " << clc_last_error() << std::endl;  // incl. the reference's std::out_of_range case, :278
            ok_ = false;
            return;
        }
        double R[9], pose[7];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[3 * i + j] = Tcl(i, j);
        rot_to_quat_xyzw(R, pose + 3);                                    // :215
        pose[0] = Tcl(0, 3); pose[1] = Tcl(1, 3); pose[2] = Tcl(2, 3);    // :219
        clc_options opt; clc_options_default(&opt);                       // DENSE_QR-equivalent, 100 iterations, :302-304
        clc_summary sum;
        std::vector<clc_iteration> trace((size_t)opt.max_num_iterations + 2);  // the per-iteration table of summary.FullReport(), :309
        if (clc_solve(H.h, &opt, pose, &sum, trace.data(), (int)trace.size()) != CLC_OK) {
            std::cerr << "[clc] " << clc_last_error() << std::endl;
            ok_ = false;
            return;
        }
        static const char* term[] = {"RUNNING", "CONVERGENCE (gradient)", "CONVERGENCE (parameter)", "CONVERGENCE (function)",
                                     "CONVERGENCE (radius)", "NO_CONVERGENCE", "FAILURE"};
        {   // stands in for the iteration table of ceres::Solver::Summary::FullReport() (:309): one line per LM iteration
            const int n_it = std::min<int>(sum.num_iterations + 1, (int)trace.size());
            std::cout << "iter      cost      cost_change  |gradient|   |step|    tr_ratio  tr_radius  step\n";
            char line[160];
            for (int i = 0; i < n_it; ++i) {
                const clc_iteration& it = trace[(size_t)i];
                std::snprintf(line, sizeof(line), "%4d  %.6e  %9.2e  %9.2e  %9.2e  %9.2e  %9.2e  %s\n", it.iteration, it.cost, it.cost_change,
                              it.gradient_max_norm, it.step_norm, it.relative_decrease, it.trust_region_radius,
                              !it.step_is_valid ? "invalid" : (it.step_is_successful ? "accepted" : "rejected"));
                std::cout << line;
            }
        }
        std::cout << "Solver Summary (MI355X HIP backend)\n  Residuals            " << n_rec
                  << "\n  Initial cost         " << sum.initial_cost << "\n  Final cost           " << sum.final_cost
                  << "\n  Iterations           " << sum.num_iterations << " (successful " << sum.num_successful_steps - 1
                  << ", unsuccessful " << sum.num_unsuccessful_steps << ")\n  Evaluation passes    " << sum.num_evaluations
                  << "\n  Time (ms)            " << sum.solve_ms << "\n  Termination          "
                  << (sum.termination >= 0 && sum.termination < (int)(sizeof(term) / sizeof(term[0])) ? term[sum.termination] : "?") << std::endl;  // :309

        // write-back, :311-314 (Quaterniond::toRotationMatrix on the unit quaternion)
        {
            const double x = pose[3], y = pose[4], z = pose[5], w = pose[6];
            const double tx = 2 * x, ty = 2 * y, tz = 2 * z, twx = tx * w, twy = ty * w, twz = tz * w;
            const double txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
            Tcl(0, 0) = 1 - (tyy + tzz); Tcl(0, 1) = txy - twz;       Tcl(0, 2) = txz + twy;
            Tcl(1, 0) = txy + twz;       Tcl(1, 1) = 1 - (txx + tzz); Tcl(1, 2) = tyz - twx;
            Tcl(2, 0) = txz - twy;       Tcl(2, 1) = tyz + twx;       Tcl(2, 2) = 1 - (txx + tyy);
            Tcl(0, 3) = pose[0]; Tcl(1, 3) = pose[1]; Tcl(2, 3) = pose[2];
        }

        /// =============================  analysis code (:316-381) ==============================
        // second pass without loss and without the board-edge terms: re-selected on the device, nothing crosses PCIe
        if (use_boundary_constraint && use_linefitting_data) {
            if (clc_select_observations(H.h, use_linefitting_data, 0, NULL) != CLC_OK) {
                std::cerr << "[clc] analysis pass skipped: " << clc_last_error() << std::endl;
                ok_ = false;
                return;
            }
        }
        double Hm[36], b[6], chi, sv[6], V[36]; int n = 0;
        if (clc_information(H.h, pose, Hm, b, &chi, sv, V, &n) != CLC_OK) {
            std::cerr << "[clc] analysis pass failed: " << clc_last_error() << std::endl;
            ok_ = false;
            return;
        }
        std::cout << "----- H singular values--------:\n";
        for (int i = 0; i < 6; ++i) std::cout << sv[i] << "\n";
        if (n > 0) {
            std::cout << "====== null space basis, it's means the unobservable direction for Tcl ======" <<std::endl;
            std::cout << "       please note the unobservable direction is for Tcl, not for Tlc        " <<std::endl;
            for (int r = 0; r < 6; ++r) { for (int c = 6 - n; c < 6; ++c) std::cout << V[6 * r + c] << " "; std::cout << "\n"; }
        }
        std::cout <<"\nrecover chi2: " <<chi / 2. << std::endl;
        ok_ = true;
    }

    // Multi-hypothesis refinement (no counterpart in the reference, which refines ONE initial guess, main/calibr_offline.cpp:166-170): every
    // Tcl of starts[0 .. n_starts) is refined on the SAME observations — one ceres::Solve per start in the reference's terms, :299-309 —
    // by ONE launch (clc_solve_multistart: a workgroup per start on one copy of the data) when the problem fits a workgroup (the
    // reference's sizes do), otherwise one start after the other.  starts[] are overwritten with the refined Tcl; final_costs (optional,
    // n_starts doubles) receives each solve's final cost.  Returns the index of the lowest final cost, -1 when the call could not run.
    int CalibrationFromStarts(Eigen::Matrix4d* starts, size_t n_starts, bool use_linefitting_data = true, bool use_boundary_constraint = false,
                              double* final_costs = NULL)
    {
        Handle H;
        ok_ = false;
        if (!H.h || !starts || n_starts == 0) return -1;
        const Flat f = flatten(*obs_);
        const int P = (int)(f.pts_off.size() - 1);
        int64_t n_rec = 0;
        if (clc_flatten_observations(P, f.tag_q.data(), f.tag_t.data(), f.pts_off.data(), f.pts, f.ptl_off.data(), f.ptl,
                                     use_linefitting_data, use_boundary_constraint, NULL, &n_rec) != CLC_OK) {
            std::cerr << "[clc] " << clc_last_error() << std::endl;
            return -1;
        }
        std::vector<clc_observation> rec((size_t)n_rec);
        clc_flatten_observations(P, f.tag_q.data(), f.tag_t.data(), f.pts_off.data(), f.pts, f.ptl_off.data(), f.ptl,
                                 use_linefitting_data, use_boundary_constraint, rec.data(), &n_rec);
        const int64_t off[2] = {0, n_rec};
        std::vector<double> poses(7 * n_starts);
        for (size_t k = 0; k < n_starts; ++k) {
            double R[9];
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[3 * i + j] = starts[k](i, j);
            rot_to_quat_xyzw(R, &poses[7 * k + 3]);                                                     // :215
            for (int i = 0; i < 3; ++i) poses[7 * k + i] = starts[k](i, 3);                             // :219
        }
        std::vector<clc_summary> sums(n_starts);
        clc_options opt; clc_options_default(&opt);
        if (clc_upload_batched(H.h, rec.data(), off, 1) != CLC_OK ||
            clc_solve_multistart(H.h, &opt, n_starts, poses.data(), sums.data()) != CLC_OK) {
            std::cerr << "[clc] " << clc_last_error() << std::endl;
            return -1;
        }
        int best = 0;
        for (size_t k = 0; k < n_starts; ++k) {
            const double* q = &poses[7 * k];
            const double x = q[3], y = q[4], z = q[5], w = q[6];                                        // write-back, :311-314
            const double tx = 2 * x, ty = 2 * y, tz = 2 * z, twx = tx * w, twy = ty * w, twz = tz * w;
            const double txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
            Eigen::Matrix4d& T = starts[k];
            T(0, 0) = 1 - (tyy + tzz); T(0, 1) = txy - twz;       T(0, 2) = txz + twy;
            T(1, 0) = txy + twz;       T(1, 1) = 1 - (txx + tzz); T(1, 2) = tyz - twx;
            T(2, 0) = txz - twy;       T(2, 1) = tyz + twx;       T(2, 2) = 1 - (txx + tyy);
            T(0, 3) = q[0]; T(1, 3) = q[1]; T(2, 3) = q[2];
            T(3, 0) = 0; T(3, 1) = 0; T(3, 2) = 0; T(3, 3) = 1;
            if (final_costs) final_costs[k] = sums[k].final_cost;
            if (sums[k].final_cost < sums[(size_t)best].final_cost) best = (int)k;
        }
        ok_ = true;
        return best;
    }

 private:
    bool store(clc_handle* h)
    {
        const Flat f = flatten(*obs_);
        const int P = (int)(f.pts_off.size() - 1);
        if (clc_store_observations(h, P, f.tag_q.data(), f.tag_t.data(), f.pts_off.data(), f.pts,
                                   f.ptl_off.data(), f.ptl) != CLC_OK) {
            std::cerr << "[clc] " << clc_last_error() << std::endl;
            return false;
        }
        generation_ = clc_store_generation(h);
        return true;
    }
    // The stored scans belong to the process-wide handle: a later Session (or one of the free functions below, which are
    // one-call sessions) replaces them.  This Session then stores its caller's observations again (as calib.py's Session does),
    // rather than solve somebody else's.
    bool mine(clc_handle* h)
    {
        if (generation_ >= 0 && clc_store_generation(h) == generation_) return true;
        return store(h);
    }
    const std::vector<Oberserve>* obs_;  // the caller's observations (not owned)
    bool ok_;
    int64_t generation_;
};

}  // namespace clc_adapter

// Closed-form initial guess of Tlc (camera -> laser).  Reference: src/LaseCamCalCeres.cpp:112-203.
inline void CamLaserCalClosedSolution(const std::vector<Oberserve> obs, Eigen::Matrix4d &Tlc)
{
    clc_adapter::Session(obs).ClosedSolution(Tlc);
}

// Nonlinear refinement of Tcl (laser -> camera).  Reference: src/LaseCamCalCeres.cpp:213-383.
inline void CamLaserCalibration(const std::vector<Oberserve> obs, Eigen::Matrix4d &Tcl, bool use_linefitting_data = true, bool use_boundary_constraint = false)
{
    clc_adapter::Session(obs).Calibration(Tcl, use_linefitting_data, use_boundary_constraint);
}
#endif //PROJECT_LASECAMCALCERES_H

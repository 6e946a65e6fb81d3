// tests/dropin/sim_main.cpp — TEST ONLY.  A ROS-free port of the flow of
// main/calibr_simulation.cpp:110-143 (GenerateSimData with a fixed seed -> Tcl = I ->
// CamLaserCalibration(obs, Tcl, false) -> print Tlc) and of main/calibr_offline.cpp:166-170
// (closed form -> invert -> refine), written against the drop-in header exactly as the
// reference's nodes use it.  Prints "RESULT tlc_err R_err" for the test to parse.
#include <chrono>
#include <cstdio>
#include <random>

#include "LaseCamCalCeres.h"

static Eigen::Matrix3d rot(double a, int axis) {
    Eigen::Matrix3d R = Eigen::Matrix3d::Identity();
    const double c = std::cos(a), s = std::sin(a);
    const int i = (axis + 1) % 3, j = (axis + 2) % 3;
    R(i, i) = c; R(i, j) = -s; R(j, i) = s; R(j, j) = c;
    return R;
}

static Eigen::Matrix4d invert(const Eigen::Matrix4d& T) {
    Eigen::Matrix4d I = Eigen::Matrix4d::Identity();
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) I(i, j) = T(j, i);
    for (int i = 0; i < 3; ++i) { double s = 0; for (int k = 0; k < 3; ++k) s += T(k, i) * T(k, 3); I(i, 3) = -s; }
    return I;
}

static void GenerateSimData(std::vector<Oberserve>& obs, unsigned seed, double sigma, size_t n_poses = 50, size_t n_rays = 180) {
    Eigen::Matrix3d Rlc;  // calibr_simulation.cpp:15-20
    Rlc(0, 2) = 1; Rlc(1, 0) = -1; Rlc(2, 1) = -1;
    Eigen::Vector3d tlc(0.1, 0.2, 0.3);
    std::mt19937 generator(seed);
    std::uniform_real_distribution<double> rpy_rand(-M_PI / 6., M_PI / 6.), xy_rand(-3, 3.0), z_rand(1., 5.);
    std::normal_distribution<double> noise(0.0, 1.0);
    for (size_t i = 0; i < n_poses; i++) {
        Eigen::Matrix3d Rca = rot(rpy_rand(generator), 2) * rot(rpy_rand(generator), 1) * rot(rpy_rand(generator), 0);
        Eigen::Vector3d tca(xy_rand(generator), xy_rand(generator), z_rand(generator));
        Eigen::Matrix3d Rla = Rlc * Rca;
        Eigen::Vector3d tla = Eigen::Vector3d(Rlc * tca) + tlc;
        Eigen::Vector3d n(Rla(0, 2), Rla(1, 2), Rla(2, 2));
        const double d = -n.dot(tla);
        std::vector<Eigen::Vector3d> points;
        for (size_t j = 0; j < n_rays; j++) {
            const double theta = -M_PI_2 + j * M_PI / n_rays;
            Eigen::Vector3d ray(cos(theta), sin(theta), 0);
            double depth = -d / (ray.dot(n));
            if (std::isnan(depth) || depth < 0) continue;
            depth += sigma * noise(generator);
            Eigen::Vector3d p = depth * ray;
            if (std::fabs(p.x()) < 5 && std::fabs(p.y()) < 5) points.push_back(p);
        }
        Oberserve ob;
        ob.tagPose_Qca = Eigen::Quaterniond(Rca);
        ob.tagPose_tca = tca;
        ob.points = points;
        ob.points_on_line = points;
        obs.push_back(ob);
    }
}

static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// main/calibr_offline.cpp:166-170 at C2 size (2 000 poses x ~550 points) through clc_adapter::Session (one upload of the
// pose-major data, residual blocks selected on the device) and, for comparison, through the two free functions with the
// reference's signatures (one upload per call).
static int session_flow(size_t n_poses, size_t n_rays) {
    std::vector<Oberserve> obs;
    GenerateSimData(obs, 7u, 0.01, n_poses, n_rays);
    size_t n_pts = 0;
    for (size_t i = 0; i < obs.size(); ++i) n_pts += obs[i].points.size();
    std::cout << "obs size: " << obs.size() << ", points: " << n_pts << std::endl;
    { Eigen::Vector2d l; l(0) = 0; l(1) = 0; LineFittingCeres(obs[0].points, l); }  // HIP start-up + context, not timed
    Eigen::Matrix4d Tlc_initial = Eigen::Matrix4d::Identity(), Tcl;
    double best_session = 1e30, best_free = 1e30, t_store = 0, t_closed = 0, t_cal = 0;
    for (int rep = 0; rep < 4; ++rep) {
        const double t0 = now_ms();
        clc_adapter::Session run(obs);
        const double t1 = now_ms();
        Tlc_initial = Eigen::Matrix4d::Identity();
        run.ClosedSolution(Tlc_initial);
        const double t2 = now_ms();
        Tcl = invert(Tlc_initial);
        run.Calibration(Tcl, false);
        const double t3 = now_ms();
        if (t3 - t0 < best_session) { best_session = t3 - t0; t_store = t1 - t0; t_closed = t2 - t1; t_cal = t3 - t2; }
    }
    for (int rep = 0; rep < 3; ++rep) {
        const double t0 = now_ms();
        Eigen::Matrix4d Ti = Eigen::Matrix4d::Identity();
        CamLaserCalClosedSolution(obs, Ti);
        Eigen::Matrix4d Tf = invert(Ti);
        CamLaserCalibration(obs, Tf, false);
        best_free = std::fmin(best_free, now_ms() - t0);
    }
    {   // the same steps through the C-ABI, timed one by one (the Session of the last repetition left the scans resident)
        clc_adapter::Handle H;
        for (int rep = 0; rep < 3; ++rep) {
            double T[16], pose[7] = {0, 0, 0, 0, 0, 0, 1}; int un = 0; clc_summary sm; clc_options opt; clc_options_default(&opt);
            const double a0 = now_ms();
            clc_select_observations(H.h, 1, 0, NULL);
            const double a1 = now_ms();
            clc_closed_form(H.h, T, &un, NULL);
            const double a2 = now_ms();
            clc_select_observations(H.h, 0, 0, NULL);
            const double a3 = now_ms();
            clc_solve(H.h, &opt, pose, &sm, NULL, 0);
            const double a4 = now_ms();
            std::printf("TIMING C-ABI steps: select(linefit) %.3f ms, closed_form %.3f, select(points) %.3f, solve %.3f\n", a1 - a0, a2 - a1, a3 - a2, a4 - a3);
        }
    }
    std::printf("TIMING session flow: total %.3f ms (store %.3f, closed form %.3f, calibration + analysis %.3f); free functions (by-value obs, one upload per call): %.3f ms; points %zu\n",
                best_session, t_store, t_closed, t_cal, best_free, n_pts);
    Eigen::Matrix4d Tlc = invert(Tcl);
    const double Rgt[3][3] = {{0, 0, 1}, {-1, 0, 0}, {0, -1, 0}}, tgt[3] = {0.1, 0.2, 0.3};
    double eR = 0, et = 0;
    for (int i = 0; i < 3; ++i) { et = std::fmax(et, std::fabs(Tlc(i, 3) - tgt[i])); for (int j = 0; j < 3; ++j) eR = std::fmax(eR, std::fabs(Tlc(i, j) - Rgt[i][j])); }
    std::printf("RESULT %.3e %.3e\n", et, eR);
    return 0;
}

// Two Sessions alive at once: the second one's store replaces the scans on the shared context; the first one notices (store
// generation), stores its own observations again and solves ITS problem — and says through ok() that it could run.
static int replaced_flow() {
    std::vector<Oberserve> obs_a, obs_b;
    GenerateSimData(obs_a, 7u, 0.0, 50, 180);
    GenerateSimData(obs_b, 8u, 0.05, 30, 60);
    clc_adapter::Session a(obs_a);
    clc_adapter::Session b(obs_b);  // replaces a's scans on the device
    Eigen::Matrix4d Tlc = Eigen::Matrix4d::Identity();
    a.ClosedSolution(Tlc);
    Eigen::Matrix4d Tcl = invert(Tlc);
    a.Calibration(Tcl, false);
    Eigen::Matrix4d Tb = Eigen::Matrix4d::Identity();
    b.ClosedSolution(Tb);  // (and back again)
    Tlc = invert(Tcl);
    const double Rgt[3][3] = {{0, 0, 1}, {-1, 0, 0}, {0, -1, 0}}, tgt[3] = {0.1, 0.2, 0.3};
    double eR = 0, et = 0;
    for (int i = 0; i < 3; ++i) { et = std::fmax(et, std::fabs(Tlc(i, 3) - tgt[i])); for (int j = 0; j < 3; ++j) eR = std::fmax(eR, std::fabs(Tlc(i, j) - Rgt[i][j])); }
    std::printf("REPLACED ok_a=%d ok_b=%d %.3e %.3e\n", a.ok() ? 1 : 0, b.ok() ? 1 : 0, et, eR);
    return 0;
}

// Multi-hypothesis refinement through the drop-in header: 64 initial guesses around the identity (rotations up to ~0.5 rad about random
// axes, translations up to 0.3 m) refined on the SAME simulated observations by ONE launch (clc_adapter::Session::CalibrationFromStarts);
// the lowest final cost wins and recovers the ground truth; a start equal to the winner's start refined alone gives the same matrix.
static int multistart_flow() {
    std::vector<Oberserve> obs;
    GenerateSimData(obs, 7u, 0.0, 50, 180);
    clc_adapter::Session run(obs);
    const size_t S = 64;
    std::vector<Eigen::Matrix4d> T(S, Eigen::Matrix4d::Identity()), T0;
    unsigned int lcg = 12345u;
    for (size_t k = 1; k < S; ++k) {  // (start 0 = the identity, the simulation node's own start, :126-129)
        double v[6];
        for (int i = 0; i < 6; ++i) { lcg = lcg * 1664525u + 1013904223u; v[i] = ((double)(lcg >> 8) / 16777216.0 - 0.5); }
        const double ax = v[0], ay = v[1], az = v[2], th = std::sqrt(ax * ax + ay * ay + az * az) + 1e-12, c = std::cos(th), s = std::sin(th);
        const double ux = ax / th, uy = ay / th, uz = az / th;
        Eigen::Matrix4d& M = T[k];
        M(0, 0) = c + ux * ux * (1 - c);      M(0, 1) = ux * uy * (1 - c) - uz * s; M(0, 2) = ux * uz * (1 - c) + uy * s;
        M(1, 0) = uy * ux * (1 - c) + uz * s; M(1, 1) = c + uy * uy * (1 - c);      M(1, 2) = uy * uz * (1 - c) - ux * s;
        M(2, 0) = uz * ux * (1 - c) - uy * s; M(2, 1) = uz * uy * (1 - c) + ux * s; M(2, 2) = c + uz * uz * (1 - c);
        M(0, 3) = 0.6 * v[3]; M(1, 3) = 0.6 * v[4]; M(2, 3) = 0.6 * v[5];
    }
    T0 = T;
    std::vector<double> cost(S);
    const int best = run.CalibrationFromStarts(T.data(), S, false, false, cost.data());
    if (best < 0 || !run.ok()) { std::printf("MULTISTART failed\n"); return 1; }
    Eigen::Matrix4d one = T0[(size_t)best];
    run.CalibrationFromStarts(&one, 1, false, false, NULL);
    double same = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) same = std::fmax(same, std::fabs(one(i, j) - T[(size_t)best](i, j)));
    size_t reached = 0;
    for (size_t k = 0; k < S; ++k) reached += cost[k] <= cost[(size_t)best] + 1e-12;
    Eigen::Matrix4d Tlc = invert(T[(size_t)best]);
    const double Rgt[3][3] = {{0, 0, 1}, {-1, 0, 0}, {0, -1, 0}}, tgt[3] = {0.1, 0.2, 0.3};
    double eR = 0, et = 0;
    for (int i = 0; i < 3; ++i) { et = std::fmax(et, std::fabs(Tlc(i, 3) - tgt[i])); for (int j = 0; j < 3; ++j) eR = std::fmax(eR, std::fabs(Tlc(i, j) - Rgt[i][j])); }
    std::printf("MULTISTART best=%d cost=%.3e reached=%zu of %zu same=%.3e %.3e %.3e\n", best, cost[(size_t)best], reached, S, same, et, eR);
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && std::string(argv[1]) == "replaced") return replaced_flow();
    if (argc > 1 && std::string(argv[1]) == "multistart") return multistart_flow();
    if (argc > 1 && std::string(argv[1]) == "session")
        return session_flow(argc > 2 ? (size_t)std::atol(argv[2]) : 2000, argc > 3 ? (size_t)std::atol(argv[3]) : 900);
    const bool offline_flow = argc > 1 && std::string(argv[1]) == "offline";
    double t_fit = 0.0, t_closed = 0.0;
    int n_fit = 0;
    std::vector<Oberserve> obs;
    GenerateSimData(obs, 7u, offline_flow ? 0.01 : 0.0);
    std::cout << "obs size: " << obs.size() << std::endl;
    Eigen::Matrix4d Tlc_initial = Eigen::Matrix4d::Identity();
    if (offline_flow) {
        // calibr_offline.cpp:121-142: fit a line to every scan, keep its two end points
        std::vector<Oberserve> kept;
        for (size_t i = 0; i < obs.size(); ++i) {
            const std::vector<Eigen::Vector3d>& points = obs[i].points;
            if (points.size() < 10) continue;
            Eigen::Vector2d line;
            line(0) = 0.0; line(1) = 0.0;
            const double t0 = now_ms();
            LineFittingCeres(points, line);
            if (n_fit++ > 0) t_fit += now_ms() - t0;  // the first call pays for HIP start-up and the context
            double x_start = points.front().x(), x_end = points.back().x();
            double y_start = points.front().y(), y_end = points.back().y();
            if (std::fabs(x_end - x_start) > std::fabs(y_end - y_start)) {
                y_start = -(x_start * line(0) + 1) / line(1);
                y_end = -(x_end * line(0) + 1) / line(1);
            } else {
                x_start = -(y_start * line(1) + 1) / line(0);
                x_end = -(y_end * line(1) + 1) / line(0);
            }
            Oberserve ob = obs[i];
            ob.points_on_line.clear();
            ob.points_on_line.push_back(Eigen::Vector3d(x_start, y_start, 0));
            ob.points_on_line.push_back(Eigen::Vector3d(x_end, y_end, 0));
            kept.push_back(ob);
        }
        obs = kept;
        std::cout << "scans with a fitted line: " << obs.size() << std::endl;
        const double t0 = now_ms();
        CamLaserCalClosedSolution(obs, Tlc_initial);                    // calibr_offline.cpp:167
        t_closed = now_ms() - t0;
    }
    Eigen::Matrix4d Tcl = invert(Tlc_initial);
    const double tc0 = now_ms();
    CamLaserCalibration(obs, Tcl, false);                               // calibr_simulation.cpp:130 / calibr_offline.cpp:170
    const double t_cal = now_ms() - tc0;
    {   // the same call once more on a warm context: what a second calibration in the same process costs
        Eigen::Matrix4d Tw = invert(Tlc_initial);
        const double t0 = now_ms();
        CamLaserCalibration(obs, Tw, false);
        std::printf("TIMING calibration first %.3f ms, warm %.3f ms; closed form %.3f ms; line fit %.4f ms per scan (%d scans, after the first)\n",
                    t_cal, now_ms() - t0, t_closed, n_fit > 1 ? t_fit / (n_fit - 1) : 0.0, n_fit);
    }
    Eigen::Matrix4d Tlc = invert(Tcl);
    std::cout << "\n----- Transform from Camera to Laser Tlc is: -----\n" << std::endl << Tlc << std::endl;
    const double Rgt[3][3] = {{0, 0, 1}, {-1, 0, 0}, {0, -1, 0}}, tgt[3] = {0.1, 0.2, 0.3};
    double eR = 0, et = 0;
    for (int i = 0; i < 3; ++i) { et = std::fmax(et, std::fabs(Tlc(i, 3) - tgt[i])); for (int j = 0; j < 3; ++j) eR = std::fmax(eR, std::fabs(Tlc(i, j) - Rgt[i][j])); }
    std::printf("RESULT %.3e %.3e\n", et, eR);
    return 0;
}

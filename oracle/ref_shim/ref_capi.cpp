// ref_capi.cpp — C entry points over the REFERENCE'S OWN translation units.
//
// TEST INFRASTRUCTURE ONLY.  This file #includes the reference sources from where they lie
// (-I$(REF_ROOT)/src -I$(REF_ROOT)/include; nothing is copied into this repository) and compiles them
// against the stand-in Eigen / Ceres / sensor_msgs headers of this directory.  Everything called
// below — PointInPlaneFactor::Evaluate, PoseLocalParameterization::{Plus,ComputeJacobian},
// pi_from_ppp, CamLaserCalClosedSolution, CamLaserCalibration, LineFittingCeres, TranScanToPoints —
// is the reference's code running as written; only the linear algebra primitives and the minimiser
// behind ceres::Solve are stand-ins (mini_eigen.hpp, mini_ceres.hpp).
// Built by `make -C oracle ref` into oracle/_ref/libref.so (git-ignored).
#include <iostream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <cstring>
#include <vector>

#include "LaseCamCalCeres.cpp"               // reference: src/LaseCamCalCeres.cpp
#include "pose_local_parameterization.cpp"   // reference: src/pose_local_parameterization.cpp
#include "utilities.cpp"                     // reference: src/utilities.cpp

// The reference's simulation node, main/calibr_simulation.cpp: GenerateSimData (:8-108) and main (:110-165).  Its
// generator seeds std::default_random_engine from std::random_device (:27-28), i.e. differently on every run; here the
// device is replaced by a fixed value so that a run can be repeated, and main is renamed so that it can be called.
#include <random>
static unsigned int g_ref_seed = 1;
namespace std {
struct clc_fixed_random_device {
  unsigned int operator()() const { return g_ref_seed; }
};
}  // namespace std
#define random_device clc_fixed_random_device
#define main ref_simulation_main
#include "calibr_simulation.cpp"             // reference: main/calibr_simulation.cpp
#undef main
#undef random_device

namespace {

struct CoutSilencer {  // the reference reports on std::cout (:175-177, :202, :309, :365-381): capture it
  std::ostringstream sink;
  std::streambuf* old;
  std::string* keep;
  explicit CoutSilencer(std::string* k = nullptr) : old(std::cout.rdbuf(sink.rdbuf())), keep(k) {}
  ~CoutSilencer() {
    std::cout.rdbuf(old);
    if (keep) *keep = sink.str();
  }
};

std::vector<Oberserve> make_obs(int n_poses, const double* tag_q_wxyz, const double* tag_t, const long long* pts_off,
                                const double* pts, const long long* ptl_off, const double* ptl) {
  std::vector<Oberserve> obs((size_t)n_poses);
  for (int i = 0; i < n_poses; ++i) {
    Oberserve& o = obs[(size_t)i];
    o.tagPose_Qca = Eigen::Quaterniond(tag_q_wxyz[4 * i], tag_q_wxyz[4 * i + 1], tag_q_wxyz[4 * i + 2], tag_q_wxyz[4 * i + 3]);
    o.tagPose_tca = Eigen::Vector3d(tag_t[3 * i], tag_t[3 * i + 1], tag_t[3 * i + 2]);
    for (long long k = pts_off[i]; k < pts_off[i + 1]; ++k) o.points.push_back(Eigen::Vector3d(pts[3 * k], pts[3 * k + 1], pts[3 * k + 2]));
    for (long long k = ptl_off[i]; k < ptl_off[i + 1]; ++k) o.points_on_line.push_back(Eigen::Vector3d(ptl[3 * k], ptl[3 * k + 1], ptl[3 * k + 2]));
  }
  return obs;
}

}  // namespace

static std::string g_last_stdout_storage;

extern "C" {

// PointInPlaneFactor::Evaluate, src/LaseCamCalCeres.cpp:43-66.  jac7 may be NULL.
void ref_factor_evaluate(const double plane[4], const double point[3], double scale, const double pose[7],
                         double* residual, double* jac7) {
  PointInPlaneFactor f(Eigen::Vector4d(plane[0], plane[1], plane[2], plane[3]), Eigen::Vector3d(point[0], point[1], point[2]), scale);
  const double* params[1] = {pose};
  double* jacs[1] = {jac7};
  f.Evaluate(params, residual, jac7 ? jacs : nullptr);
}

// PoseLocalParameterization, src/pose_local_parameterization.cpp:15-40 (its members are private in the
// reference's class declaration: reach them through the public ceres::LocalParameterization interface).
void ref_pose_plus(const double x[7], const double delta[6], double out[7]) {
  PoseLocalParameterization p;
  static_cast<const ceres::LocalParameterization&>(p).Plus(x, delta, out);
}
void ref_pose_plus_jacobian(const double x[7], double jac42[42]) {
  PoseLocalParameterization p;
  static_cast<const ceres::LocalParameterization&>(p).ComputeJacobian(x, jac42);
}
int ref_pose_sizes(int* global_size, int* local_size) {
  PoseLocalParameterization p;
  *global_size = static_cast<const ceres::LocalParameterization&>(p).GlobalSize();
  *local_size = static_cast<const ceres::LocalParameterization&>(p).LocalSize();
  return 0;
}

// pi_from_ppp, src/utilities.cpp:267-272
void ref_pi_from_ppp(const double x1[3], const double x2[3], const double x3[3], double pi[4]) {
  const Eigen::Vector4d r = pi_from_ppp(Eigen::Vector3d(x1[0], x1[1], x1[2]), Eigen::Vector3d(x2[0], x2[1], x2[2]),
                                        Eigen::Vector3d(x3[0], x3[1], x3[2]));
  for (int i = 0; i < 4; ++i) pi[i] = r[i];
}

// CamLaserCalClosedSolution, src/LaseCamCalCeres.cpp:112-203.  Tlc row-major 4x4, in/out.
void ref_closed_solution(int n_poses, const double* tag_q_wxyz, const double* tag_t, const long long* pts_off,
                         const double* pts, const long long* ptl_off, const double* ptl, double Tlc[16]) {
  CoutSilencer quiet(&g_last_stdout_storage);
  Eigen::Matrix4d T;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) T(i, j) = Tlc[4 * i + j];
  CamLaserCalClosedSolution(make_obs(n_poses, tag_q_wxyz, tag_t, pts_off, pts, ptl_off, ptl), T);
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) Tlc[4 * i + j] = T(i, j);
}

// CamLaserCalibration, src/LaseCamCalCeres.cpp:213-383.  Tcl row-major 4x4, in/out.
// record[6] = {termination, iterations, successful, unsuccessful, initial_cost, final_cost} of the
// stand-in ceres::Solve; *n_blocks = residual blocks the reference's loop added.  Returns -1 when the
// reference throws std::out_of_range (empty scan in boundary mode).
int ref_calibration(int n_poses, const double* tag_q_wxyz, const double* tag_t, const long long* pts_off,
                     const double* pts, const long long* ptl_off, const double* ptl, double Tcl[16],
                     int use_linefitting_data, int use_boundary_constraint, double record[6], long long* n_blocks) {
  CoutSilencer quiet(&g_last_stdout_storage);
  Eigen::Matrix4d T;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) T(i, j) = Tcl[4 * i + j];
  try {
    CamLaserCalibration(make_obs(n_poses, tag_q_wxyz, tag_t, pts_off, pts, ptl_off, ptl), T, use_linefitting_data != 0,
                        use_boundary_constraint != 0);
  } catch (const std::out_of_range&) {
    return -1;  // obi.points.at(0) on an empty scan, :278-279
  }
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) Tcl[4 * i + j] = T(i, j);
  const ceres::SolveRecord& r = ceres::last_solve_record();
  if (record) {
    record[0] = r.termination; record[1] = r.num_iterations; record[2] = r.num_successful_steps;
    record[3] = r.num_unsuccessful_steps; record[4] = r.initial_cost; record[5] = r.final_cost;
  }
  if (n_blocks) *n_blocks = r.num_residual_blocks;
  return 0;
}

// LineFittingCeres, src/LaseCamCalCeres.cpp:401-433.  pts = n x 3, line in/out.
// GenerateSimData, main/calibr_simulation.cpp:8-108, with the random device pinned to `seed`.  Outputs for 50 poses:
// tag_q_wxyz[50*4], tag_t[50*3], counts[50] and the points (up to cap, 3 doubles each).  Returns the number of points.
long long ref_generate_sim_data(unsigned int seed, double* tag_q_wxyz, double* tag_t, long long* counts, double* pts,
                                long long cap) {
  CoutSilencer quiet;
  g_ref_seed = seed;
  std::vector<Oberserve> obs;
  GenerateSimData(obs);
  long long n = 0;
  for (size_t i = 0; i < obs.size(); ++i) {
    tag_q_wxyz[4 * i] = obs[i].tagPose_Qca.w(); tag_q_wxyz[4 * i + 1] = obs[i].tagPose_Qca.x();
    tag_q_wxyz[4 * i + 2] = obs[i].tagPose_Qca.y(); tag_q_wxyz[4 * i + 3] = obs[i].tagPose_Qca.z();
    for (int c = 0; c < 3; ++c) tag_t[3 * i + c] = obs[i].tagPose_tca[c];
    counts[i] = (long long)obs[i].points.size();
    for (size_t j = 0; j < obs[i].points.size(); ++j, ++n)
      if (n < cap) for (int c = 0; c < 3; ++c) pts[3 * n + c] = obs[i].points[j][c];
  }
  return n;
}

// The reference's whole simulation program (its main(), :110-165) with the random device pinned to `seed`; what it
// prints is available through ref_last_stdout().
int ref_simulation_program(unsigned int seed) {
  CoutSilencer quiet(&g_last_stdout_storage);
  g_ref_seed = seed;
  int argc = 1;
  char arg0[] = "simulation_lasercamcal_node";
  char* argv[] = {arg0, nullptr};
  try {
    return ref_simulation_main(argc, argv);  // early exits of the program ("Valid Calibra Data Less")
  } catch (const ros::SpinCalled&) {
    return 0;  // reached ros::spin(): the program ran to its end
  }
}

// What the last ref_calibration / ref_closed_solution call printed on std::cout (the reference reports the
// analysis pass — H singular values, null space, chi2 — only there, :365-381).  Returns the length.
long long ref_last_stdout(char* buf, long long cap) {
  const long long n = (long long)g_last_stdout_storage.size();
  if (buf && cap > 0) {
    const long long m = n < cap - 1 ? n : cap - 1;
    std::memcpy(buf, g_last_stdout_storage.data(), (size_t)m);
    buf[m] = 0;
  }
  return n;
}

void ref_line_fitting(const double* pts, long long n, double line[2], double record[6]) {
  CoutSilencer quiet;
  std::vector<Eigen::Vector3d> P;
  for (long long k = 0; k < n; ++k) P.push_back(Eigen::Vector3d(pts[3 * k], pts[3 * k + 1], pts[3 * k + 2]));
  Eigen::Vector2d L(line[0], line[1]);
  LineFittingCeres(P, L);
  line[0] = L(0);
  line[1] = L(1);
  const ceres::SolveRecord& r = ceres::last_solve_record();
  if (record) {
    record[0] = r.termination; record[1] = r.num_iterations; record[2] = r.num_successful_steps;
    record[3] = r.num_unsuccessful_steps; record[4] = r.initial_cost; record[5] = r.final_cost;
  }
}

// TranScanToPoints, src/utilities.cpp:181-215.  out = n x 3.
void ref_scan_to_points(const float* ranges, long long n, float angle_min, float angle_increment, float range_min,
                        double* out) {
  sensor_msgs::LaserScan scan;
  scan.angle_min = angle_min;
  scan.angle_increment = angle_increment;
  scan.range_min = range_min;
  scan.ranges.assign(ranges, ranges + n);
  std::vector<Eigen::Vector3d> P;
  TranScanToPoints(scan, P);
  for (long long k = 0; k < n; ++k)
    for (int c = 0; c < 3; ++c) out[3 * k + c] = P[(size_t)k][c];
}

}  // extern "C"

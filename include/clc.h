/* =====================================================================================
 * clc.h — C-ABI of the MI355X-native point-to-plane extrinsic solver ("clc" = Cam-Laser
 * Calibration).  This is the drop-in boundary: host code (the C++ adapter in
 * include/LaseCamCalCeres.h, the Python mirror in camlasercalibratool_amd/) calls these
 * entry points; everything behind them is hand-written HIP for gfx950.
 *
 * Plain pointers and sizes only; no exceptions cross this boundary; every function returns
 * an int status (CLC_OK == 0, negative on error — the reference itself has no error
 * codes).  All arithmetic is IEEE FP64.  Host pointers unless a parameter
 * says "device".  Calls on one handle are blocking and must not overlap; different
 * handles are independent (one HIP stream per handle).
 *
 * Each entry point cites the reference interface it replaces (paths relative to the
 * reference root, MegviiRobot/CamLaserCalibraTool).
 * ===================================================================================== */
#ifndef CLC_H_
#define CLC_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CLC_VERSION 210

/* status codes */
#define CLC_OK 0
#define CLC_ERR_INVALID_ARG (-1)
#define CLC_ERR_HIP (-2)        /* HIP runtime / launch failure: see clc_last_error()        */
#define CLC_ERR_NONFINITE (-3)  /* non-finite input pose or evaluation                      */
#define CLC_ERR_EMPTY_SCAN (-4) /* boundary mode on an empty `points` (reference throws
                                   std::out_of_range at src/LaseCamCalCeres.cpp:278)         */
#define CLC_ERR_NO_DATA (-5)    /* solve/eval before upload                                 */
#define CLC_ERR_LINALG (-6)     /* (reserved; round 1 returned it for a rank-deficient 9x9 closed-form
                                   system — the pivoted LDLT back end now solves those like the reference) */
#define CLC_ERR_NO_DEVICE (-7)  /* no gfx950 device / HIP runtime unavailable                */
#define CLC_ERR_COMM (-8)       /* RCCL unavailable or a collective failed: see clc_last_error() */

/* termination codes of clc_summary.termination (ceres::TerminationType + which test) */
#define CLC_RUNNING 0
#define CLC_CONVERGENCE_GRADIENT 1
#define CLC_CONVERGENCE_PARAMETER 2
#define CLC_CONVERGENCE_FUNCTION 3
#define CLC_CONVERGENCE_RADIUS 4
#define CLC_NO_CONVERGENCE 5
#define CLC_FAILURE 6

typedef struct clc_handle clc_handle;

/* One residual block, exactly the state a PointInPlaneFactor + its CauchyLoss hold
 * (src/LaseCamCalCeres.cpp:19-21,249): 8 doubles = 64 bytes, the unit of algorithmic
 * traffic.  The observation array handed to clc_upload is an array of
 * these. */
typedef struct clc_observation {
  double n[3];  /* plane normal in the camera frame (planar_.head(3))                        */
  double d;     /* plane offset (planar_[3])                                                */
  double p[3];  /* laser point in the laser frame (point_)                                  */
  double scale; /* 1/sqrt(points in this scan) (scale_, :239-240); loss a = 0.05*scale       */
} clc_observation;

/* Solver options.  Defaults (clc_options_default) = what the reference runs:
 * src/LaseCamCalCeres.cpp:302-304 (DENSE_QR, 100 iterations) + Ceres defaults + the
 * hard-coded Cauchy loss of :212,:249. */
typedef struct clc_options {
  int32_t max_num_iterations;                /* 100                                          */
  int32_t max_num_consecutive_invalid_steps; /* 5                                            */
  int32_t jacobi_scaling;                    /* 1                                            */
  int32_t use_loss;                          /* 1  (#define LOSSFUNCTION)                    */
  double loss_scale_factor;                  /* 0.05  -> CauchyLoss(0.05*scale)              */
  double initial_trust_region_radius;        /* 1e4                                          */
  double max_trust_region_radius;            /* 1e16                                         */
  double min_trust_region_radius;            /* 1e-32                                        */
  double min_relative_decrease;              /* 1e-3                                         */
  double min_lm_diagonal;                    /* 1e-6                                         */
  double max_lm_diagonal;                    /* 1e32                                         */
  double function_tolerance;                 /* 1e-6                                         */
  double gradient_tolerance;                 /* 1e-10                                        */
  double parameter_tolerance;                /* 1e-8                                         */
  /* execution knobs (no effect on results beyond reduction order) */
  int32_t launch_ahead;  /* launch-ahead depth: LM iterations the host keeps queued beyond the
                             last one the device reported done (pinned mailbox, no blocking
                             sync); 0 = library default (2)                                  */
  int32_t profile_events; /* 1: bracket every evaluation-kernel launch with HIP events on the
                             handle's stream and report them in clc_summary (the solve then
                             uses the [evaluation, controller] launch pair, not the one-launch
                             step kernel); 2: when the solve is ONE launch (see clc_solve), an
                             event pair around it -> eval_kernel_ms, eval_kernel_launches = 1;
                             otherwise controller cycle stamps only (debug)                    */
} clc_options;

/* ceres::IterationSummary subset, one per recorded iteration (iteration 0 = initial
 * evaluation).  Replaces summary.FullReport() (src/LaseCamCalCeres.cpp:309). */
typedef struct clc_iteration {
  int32_t iteration;
  int32_t step_is_valid;
  int32_t step_is_successful;
  int32_t pad_;
  double cost;
  double cost_change;
  double gradient_max_norm;
  double step_norm;
  double relative_decrease;
  double trust_region_radius;
} clc_iteration;

typedef struct clc_summary {
  int32_t termination;            /* CLC_CONVERGENCE_* / CLC_NO_CONVERGENCE / CLC_FAILURE    */
  int32_t num_iterations;         /* recorded iterations - 1                                 */
  int32_t num_successful_steps;
  int32_t num_unsuccessful_steps;
  int64_t num_evaluations;        /* fused residual+Jacobian passes over the observations    */
  double initial_cost;
  double final_cost;
  double solve_ms;                /* host wall time of the call                              */
  double eval_kernel_ms;          /* sum of evaluation-kernel durations (profile_events=1)   */
  int64_t eval_kernel_launches;   /* number of launches summed in eval_kernel_ms             */
} clc_summary;

/* ---- library ------------------------------------------------------------------------ */
int clc_version(void);
/* Last error text of the calling thread ("" if none). */
const char* clc_last_error(void);
void clc_options_default(clc_options* opt);

/* ---- handle ------------------------------------------------------------------------- */
/* Creates a solver context on HIP device `device` (one stream, scratch buffers).  It also loads the code objects of the single-problem
 * paths and pays what only the first launch / copy / allocation of a process pays (tens of ms next to the ~130-150 ms of HIP context
 * creation), so that the first call on the handle costs about what a warm one does — the reference's programs call the path once
 * per process (main/calibr_offline.cpp:166-170).  The batched kernels' code object is loaded by the first clc_upload_batched.
 * CLC_LAZY_MODULES=1 in the environment leaves all of it to the first launches.
 * CLC_ERR_NO_DEVICE without a GPU: there is no CPU fallback. */
int clc_create(clc_handle** out, int device);
void clc_destroy(clc_handle* h);
/* Run on a caller-owned hipStream_t (e.g. torch.cuda.current_stream().cuda_stream);
 * NULL restores the handle's own stream.  The switch drains the previous stream first (a finished
 * solve may still have a few no-op launches queued on it), so work on the new stream never overlaps
 * work of this handle on the old one.  Device pointers handed to this library (clc_upload_device,
 * the *_device entry points) must be ready on the handle's CURRENT stream. */
int clc_set_stream(clc_handle* h, void* hip_stream);
/* Explicit choice of the streaming paths — what the A/B tests and the profiling scripts use; a caller keeps the default (-1).
 * grid_blocks = workgroups of the evaluation / step launches (0 = library default: one 512-thread workgroup per CU).
 * flags = -1: library default = 2|16|32|128|256|512 with the size-dependent choices made per launch (non-temporal loads and deeper
 *   pipelines beyond the 256 MiB Infinity Cache) AND the on-chip solve paths of clc_solve (clc_set_auto_paths).  Any explicit value
 *   selects the streaming paths only (clc_solve = step chain / launch pair), as a sum of:
 *      1  reference shuffle reduction instead of the butterfly          2  software prefetch of the next tile (64-byte tiles)
 *      4  non-temporal loads                                           16  compact 28-byte layout where available (lossless; arrays without
 *     32  512-thread workgroups (3:2 old/young wave shares)                dense scans, and the step chain below 2e5 observations)
 *     64  compact layout: two tiles in flight per wave                128  clc_solve = ONE step_kernel launch per LM iteration (needs 32 and 16 or 256)
 *    256  row layout (scans padded to rows of 64 points: 16 B/point + a 64-byte descriptor per row, per-scan moments; 24 B/point when p.z != 0)
 *    512  row layout: equal wave shares cut at scan starts           1024  batched row kernel in 256-thread workgroups (default: one wave each)
 *   2048  batched solver: lockstep [evaluation, controller] launches 4096  batched solver: not the on-chip resident kernel (at upload: its
 *   8192  resident layout over 512 lanes even where 256 hold it            layout is not built)
 * Results change only in summation order. */
int clc_set_launch(clc_handle* h, int grid_blocks, int flags);

/* Which of the size-chosen default paths of clc_solve (flags = -1 above) may run; disable_mask is a sum of
 * 1 = not the cooperative one-launch solve (11 264 < n <= 2.6e6 observations: 256 co-resident workgroups keep the problem on
 *     chip and exchange their rows through device memory — fast on an otherwise idle GPU; a launch whose workgroups are not all
 *     resident within 0.2 ms aborts, falls back to the step chain and rests the path on the handle for 16 solves, doubling):
 *     a process that shares the GPU with long-running kernels can switch it off here and keep every other default;
 * 2 = not the single-workgroup on-chip solve (n <= 11 264 observations);
 * 8 = (at upload) not the 32-workgroup one-hop form of the cooperative solve for problems of at most 106 496 observations
 *     (every workgroup reads all 32 rows itself: one store-to-load hop per pass instead of two) — the 256-workgroup form then.
 * 0 = library default: every bit of the mask DISABLES a path.  The environment variable CLC_AUTO_PATHS_DISABLE sets the initial mask
 * of every handle. */
int clc_set_auto_paths(clc_handle* h, int disable_mask);
/* Opt-in (at upload): a problem one workgroup holds (n <= 11 264) ALSO gets the cooperative layout and clc_solve runs it on 32 co-resident
 * workgroups first — 4.6 instead of 5.3-5.9 us per pass (C1 0.132 -> 0.116 ms per solve, 10 000 observations 0.135 -> 0.107) —, with the
 * single-workgroup kernel as the fall-back when that launch times out or rests.  Not the default: upload and first solve cost ~0.2 ms more
 * (a second layout, the exchange boards), which the reference's one calibration per process never earns back; worth it from about a
 * dozen solves per upload on an otherwise idle GPU.  CLC_SMALL_ON_COOP=1 in the environment sets it on every new handle. */
int clc_set_small_on_coop(clc_handle* h, int enable);

/* Which layouts the last clc_upload* / clc_select_observations / clc_upload_batched* built on this handle, and how the
 * cooperative path is doing: a caller that sizes its problems, or shares the GPU, can see whether clc_solve and
 * clc_solve_batched run from registers + LDS in one launch or on the streaming paths (operational query; no reference
 * counterpart). */
typedef struct clc_path_info {
  int32_t single_resident;         /* 1: clc_solve = ONE single-workgroup launch (<= 512 lanes x 22 points, a lane holds one scan's points) */
  int32_t single_lanes;
  int32_t single_points_per_lane;
  int32_t coop_resident;           /* 1: the lane layout of the cooperative kernel is built: clc_solve = ONE launch of 256 workgroups */
  int32_t coop_points_per_lane;    /* <= 40 (<= 26 when the points carry z) */
  int32_t coop_points_carry_z;     /* 1: some record has p.z != 0: 24-byte slots */
  int32_t coop_resting;            /* 1: after a launch that timed out the path rests (the step chain runs) until its back-off expires */
  int32_t coop_timeouts;           /* cooperative launches that timed out on this handle */
  int32_t batched_resident;        /* 1: clc_solve_batched = ONE launch of the resident kernel */
  int32_t batched_lanes;           /* 256 / 512 lanes per problem */
  int32_t batched_points_per_lane;
  int32_t rows_layout;             /* single problem's streaming row layout: 0 none, 1 (x, y) rows, 2 rows that carry z */
  int32_t batched_rows_layout;
  int32_t coop_workgroups;         /* 256, or 32: the one-hop form for problems of at most 32 x 256 x 13 points */
  int32_t batched_points_carry_z;  /* 1: the batch has p.z != 0 somewhere and is held on chip in 24-byte slots (512 lanes x <= 22 points per problem) */
  int32_t reserved_;
  int64_t coop_solves;             /* solves that ran on the cooperative kernel */
  int64_t batched_lane_rows;       /* point rows of the batched lane layout (x lanes x 16 bytes = its size) */
  int64_t n_rows;                  /* rows of 64 points of the streaming row layouts */
  int64_t batched_n_rows;
  int64_t coop_gate_waits_expired; /* cooperative solves that took the step chain because another handle of this process held the device's
                                    * one-cooperative-launch-at-a-time gate for more than 5 ms (clc_version() >= 210; fields are only appended) */
} clc_path_info;
int clc_get_path_info(const clc_handle* h, clc_path_info* out);

/* Name (NUL-terminated, truncated to name_cap) and compute-unit count of the handle's device; either out pointer may be NULL. */
int clc_device_info(clc_handle* h, char* name, int name_cap, int* num_cus);

/* ---- problem assembly (host) --------------------------------------------------------
 * Replaces the residual-block construction loop of CamLaserCalibration,
 * src/LaseCamCalCeres.cpp:222-295 (plane per pose :227-231, scale :239-240, board-edge
 * terms :258-294 with pi_from_ppp src/utilities.cpp:267-272).  Input is the flattened
 * std::vector<Oberserve> (include/LaseCamCalCeres.h:11-24): tag_q[n_poses*4] as (w,x,y,z),
 * tag_t[n_poses*3], CSR offsets/points for `points` and `points_on_line`.
 * records == NULL only counts.  *n_records receives N. */
int clc_flatten_observations(int n_poses, const double* tag_q_wxyz, const double* tag_t,
                             const int64_t* pts_off, const double* pts,
                             const int64_t* ptl_off, const double* ptl,
                             int use_linefitting_data, int use_boundary_constraint,
                             clc_observation* records, int64_t* n_records);

/* ---- resident scans + problem assembly on the device ------------------------------------
 * The same residual-block construction (src/LaseCamCalCeres.cpp:222-295), but the pose-major data of
 * std::vector<Oberserve> — tag poses and the scan points, 24 bytes per point — is what crosses PCIe, once, and
 * stays resident; the 64-byte records of any (use_linefitting_data, use_boundary_constraint) selection are then
 * built on the device, bitwise equal to clc_flatten_observations' (every operation individually rounded, same
 * order), and handed to the upload pipeline without touching the host.  This is how the drop-in header runs
 * closed form -> calibration -> analysis pass of main/calibr_offline.cpp:166-170 on ONE upload
 * (clc_adapter::Session).  Arguments as clc_flatten_observations.  clc_select_observations replaces the
 * observation array of the handle (as clc_upload would); CLC_ERR_EMPTY_SCAN as for the host path. */
/* Page-locked host memory for arrays handed to clc_store_observations / clc_upload*: from pageable memory the runtime
 * pins and unpins the caller's pages around every copy (tens of milliseconds for 50 MB when the buffer is a fresh
 * allocation each call); from these buffers the copy is a plain DMA.  NULL on failure. */
void* clc_pinned_alloc(size_t bytes);
void clc_pinned_free(void* p);
int clc_store_observations(clc_handle* h, int n_poses, const double* tag_q_wxyz, const double* tag_t,
                           const int64_t* pts_off, const double* pts, const int64_t* ptl_off, const double* ptl);
int clc_select_observations(clc_handle* h, int use_linefitting_data, int use_boundary_constraint,
                            int64_t* n_records);
/* Number of successful clc_store_observations calls on this handle so far (-1 for NULL).  The stored scans belong to the
 * handle, not to whoever stored them: a caller that keeps "its" scans across calls on a shared handle (the Session of
 * the drop-in header) remembers this stamp and re-stores, or refuses to go on, when it has moved. */
int64_t clc_store_generation(const clc_handle* h);

/* ---- observation array --------------------------------------------------------------
 * Copies N records to the device and re-tiles them for coalesced 16-byte loads.  Stays
 * resident until the next upload/destroy.  `clc_upload_device` takes a DEVICE pointer to
 * the same AoS layout (e.g. a torch tensor's data_ptr()). */
int clc_upload(clc_handle* h, const clc_observation* records, size_t n);
int clc_upload_device(clc_handle* h, const clc_observation* records_dev, size_t n);
size_t clc_num_observations(const clc_handle* h);

/* ---- plug-in level math (element-wise parity) ---------------------------------------
 * PointInPlaneFactor::Evaluate for every uploaded record at `pose` (src/LaseCamCalCeres.cpp
 * :43-66): residuals[N] (raw, before the loss) and, if non-NULL, jacobians[N*7] row-major
 * 1x7 rows exactly as the factor writes them (7th column zero, :60). */
int clc_factor_evaluate(clc_handle* h, const double pose[7], double* residuals,
                        double* jacobians);
/* PoseLocalParameterization::Plus, batched on the device
 * (src/pose_local_parameterization.cpp:15-31): x[n*7], delta[n*6] -> x_plus_delta[n*7]. */
int clc_pose_plus(clc_handle* h, const double* x, const double* delta, double* x_plus_delta,
                  size_t n);
/* PoseLocalParameterization::ComputeJacobian: 7x6 row-major [I6;0] (cpp:33-40). */
int clc_pose_plus_jacobian(const double x[7], double jacobian[42]);

/* ---- one evaluation pass --------------------------------------------------------------
 * What Ceres' evaluator + loss corrector produce per evaluation, reduced to the normal
 * equation: cost = 1/2 sum rho, g[6] = J~^T r~, H[21] = upper triangle of J~^T J~
 * (row-major: 00 01 .. 05 11 ..).  with_loss=0 drops the Cauchy loss (analysis pass,
 * src/LaseCamCalCeres.cpp:316-362).  g/H may be NULL (cost-only pass). */
int clc_eval(clc_handle* h, const double pose[7], int with_loss, double loss_scale_factor,
             double* cost, double g[6], double H[21]);

/* ---- the solve ------------------------------------------------------------------------
 * Replaces ceres::Solve on the problem built by CamLaserCalibration
 * (src/LaseCamCalCeres.cpp:299-307): Levenberg-Marquardt trust region with Jacobi scaling
 * on the uploaded observations, SE(3) local parameterisation, Cauchy loss; runs entirely
 * on the device.  pose is in/out = [tx,ty,tz,qx,qy,qz,qw] (:219, :311-314).
 * trace (nullable) receives up to trace_cap iteration records.
 * With the library's default launch flags the whole solve is ONE launch when the problem fits on chip:
 * up to 11 264 observations in one workgroup (all p.z == 0, every lane's points from one scan), up to ~2.6e6
 * across 256 co-resident workgroups (~1.7e6 when some p.z != 0: 24-byte slots) (csrc/clc_coop.hpp; needs a
 * 256-CU device, falls back by itself otherwise or when its exchange times out — see clc_set_auto_paths);
 * beyond that — or with explicit clc_set_launch flags — one launch per LM iteration.  Same LM
 * decisions on every path; sums are taken in different orders (results agree to rounding). */
int clc_solve(clc_handle* h, const clc_options* opt, double pose[7], clc_summary* summary,
              clc_iteration* trace, int trace_cap);

/* ---- post-solve analysis ---------------------------------------------------------------
 * src/LaseCamCalCeres.cpp:316-381: un-robustified H = sum J^T J (6x6 row-major),
 * b = -sum J^T r, chi2 = sum r^2, singular values of H (descending), V (columns), and the
 * count of singular values < 1e-8 (null-space dimension).  The caller uploads the point
 * residuals only (the reference skips the board-edge terms here). */
int clc_information(clc_handle* h, const double pose[7], double H[36], double b[6],
                    double* chi2, double sv[6], double V[36], int* n_null);

/* ---- closed-form initialiser ------------------------------------------------------------
 * CamLaserCalClosedSolution, src/LaseCamCalCeres.cpp:112-203, on the uploaded
 * points_on_line records (uses n, d, p.x, p.y only, :147): device reduction of the 9x9
 * normal equation (16 bytes per point on the row layout), host 9x9 pivoted LDL^T solve (Eigen's ldlt(),
 * :181, pseudo-inverse of D) + U V^T of the 3x3 SVD (:195-196).  Tlc[16] row-major 4x4;
 * *unobservable = 1 if any singular value of A^T A < 1e-10 (:164-171) — like the reference the
 * function then still returns the Tlc the factorisations give (CLC_ERR_NONFINITE only if that is not
 * finite; *unobservable is set either way); sv9 nullable. */
int clc_closed_form(clc_handle* h, double Tlc[16], int* unobservable, double sv9[9]);

/* ---- batched independent problems ---------------------------------------------------------
 * P independent T_cl problems (own observations, own pose, own LM state), problem k owning
 * records [offsets[k], offsets[k+1]).  One workgroup per problem runs the whole LM loop on
 * the device.  poses[P*7] in/out, summaries[P]. */
int clc_upload_batched(clc_handle* h, const clc_observation* records, const int64_t* offsets,
                       size_t n_problems);
/* records_dev: DEVICE pointer to the same AoS records (ready on the handle's stream); offsets stay a host array. */
int clc_upload_batched_device(clc_handle* h, const clc_observation* records_dev, const int64_t* offsets,
                              size_t n_problems);
int clc_solve_batched(clc_handle* h, const clc_options* opt, double* poses,
                      clc_summary* summaries);
/* The handle's own page-locked, device-mapped arrays for the poses (in/out, [P*7]) and summaries ([P]) of the uploaded batch
 * (valid until the next clc_upload_batched* / clc_destroy).  Passing exactly these two pointers to clc_solve_batched solves
 * in place: the start poses are read and the results written over PCIe by the kernels themselves, and the two staging
 * copies per call (a megabyte at 8 192 problems, ~10 % of a C4-shard solve) are skipped. */
int clc_batched_host_buffers(clc_handle* h, double** poses, clc_summary** summaries);
size_t clc_num_problems(const clc_handle* h);

/* ---- scan line fitting (the step that produces points_on_line) -----------------------------
 * LineFittingCeres, src/LaseCamCalCeres.cpp:385-433, for n_scans scans at once.  Scan k owns
 * points [offsets[k], offsets[k+1]) of xy[2*M] (x, y of every scan point; z is not used, :412).
 * lines[2*n_scans] is in/out = (m0, m1) of the line m0 x + m1 y + 1 = 0: initial guess in (:403),
 * result out (:430-431).  Residual m0 x + m1 y + 1 (:391), CauchyLoss(opt->loss_scale_factor) per
 * point (:416), the same Ceres LM semantics as clc_solve.  clc_line_options_default() = Ceres
 * defaults with max_num_iterations = 10 and loss 0.05 (:416,:424-425).  summaries nullable. */
void clc_line_options_default(clc_options* opt);
int clc_line_fit_batched(clc_handle* h, const clc_options* opt, const double* xy, const int64_t* offsets,
                         size_t n_scans, double* lines, clc_summary* summaries);
/* The same with every array in DEVICE memory (ready on the handle's stream; results are complete on that stream when
 * the call returns — it synchronises): xy_dev[2*M], offsets_dev[n_scans+1] (absolute, offsets_dev[0] may be > 0),
 * lines_dev[2*n_scans] in/out, summaries_dev nullable.  Offsets and start lines are not validated. */
int clc_line_fit_batched_device(clc_handle* h, const clc_options* opt, const double* xy_dev, const int64_t* offsets_dev,
                                size_t n_scans, double* lines_dev, clc_summary* summaries_dev);

/* TranScanToPoints, src/utilities.cpp:181-215, for n_scans (<= 65535) scans at once: scan k owns rays
 * [offsets[k], offsets[k+1]) of ranges[] (float32 as in sensor_msgs/LaserScan); ray i of scan k
 * sits at angle_min[k] + i * angle_increment[k] (:192-193) and becomes (r cos, r sin, 0), or
 * (1000, 1000, 0) when r is outside [range_min[k], 30) (:201-208).  points[3 * total rays]. */
int clc_scan_to_points(clc_handle* h, const float* ranges, const int64_t* offsets, size_t n_scans,
                       const float* angle_min, const float* angle_increment, const float* range_min,
                       double* points);
/* The same with every array in DEVICE memory (any number of scans; n_rays = offsets[n_scans] - offsets[0], which
 * the caller knows; offsets_dev[0] must be 0). */
int clc_scan_to_points_device(clc_handle* h, const float* ranges_dev, const int64_t* offsets_dev, size_t n_scans,
                              size_t n_rays, const float* angle_min_dev, const float* angle_increment_dev,
                              const float* range_min_dev, double* points_dev);

/* Multi-hypothesis calibration on SHARED observations: n_starts independent LM solves (one ceres::Solve each, src/LaseCamCalCeres.cpp
 * :299-309) from n_starts start poses on the ONE problem the handle holds as a batch of one (clc_upload_batched* with n_problems = 1).
 * poses: n_starts x 7, in/out; summaries: n_starts.  Where the problem fits a workgroup (clc_path_info.batched_resident) ONE launch runs
 * every start — a workgroup per start, all of them loading the same on-chip layout: one copy of the observations in HBM instead of
 * n_starts uploaded copies — and the results are bit-identical to clc_solve_batched of n_starts copies; a larger problem runs its
 * starts one after the other on the batch's streaming path (a single problem beyond a workgroup is better served by clc_upload +
 * one clc_solve per start: the cooperative solve uses the whole GPU for each). */
int clc_solve_multistart(clc_handle* h, const clc_options* opt, size_t n_starts, double* poses_inout, clc_summary* summaries);

/* ---- multi-GPU: sharded batches + RCCL gather ------------------------------------------------
 * BASELINE.json configs[3]: independent T_cl problems shard across the GPUs of a node, one process
 * per GPU, no collective on the data path; the fixed-size result records of all ranks are gathered
 * once over xGMI.  The reference has no counterpart (single-threaded Ceres, src/LaseCamCalCeres.cpp
 * :302-304); the record is what its caller keeps of a solve: the pose written back at :311-314 and the
 * Solver::Summary fields printed at :309.
 *
 * RCCL is bound at run time (dlopen): the copy already loaded into the process (e.g. PyTorch's) is
 * reused, else librccl.so.1; CLC_RCCL_LIBRARY overrides.  Only these entry points need it. */
typedef struct clc_result_record {
  double pose[7];         /* [tx,ty,tz,qx,qy,qz,qw] */
  double final_cost;
  double initial_cost;
  double num_iterations;
  double termination;     /* CLC_CONVERGENCE_* ... as a double */
  double global_index;    /* global problem index; -1 marks a padding record */
} clc_result_record;      /* 12 doubles = 96 bytes */

#define CLC_COMM_ID_BYTES 128
typedef struct clc_comm clc_comm;
/* ncclGetUniqueId: called by ONE rank; the caller distributes the 128 bytes to the other ranks out of
 * band (torch.distributed store, MPI, a file). */
int clc_comm_unique_id(char id[CLC_COMM_ID_BYTES]);
/* ncclCommInitRank on the handle's device; collectives run on the handle's stream.  Collective call:
 * every rank of the job must enter it. */
int clc_comm_create(clc_comm** out, clc_handle* h, const char id[CLC_COMM_ID_BYTES], int rank, int world);
void clc_comm_destroy(clc_comm* c);
/* Rank and size of the communicator as RCCL itself reports them (ncclCommUserRank / ncclCommCount). */
int clc_comm_rank(const clc_comm* c);
int clc_comm_world(const clc_comm* c);
/* Path of the RCCL library the collectives are bound to at run time (the one already loaded in the process — torch's — or
 * librccl.so from the loader's path); "" before the first clc_comm_* call resolved it. */
const char* clc_comm_library(void);
/* WHERE the gathered records go (SURVEY.md §8e specifies a gather to rank 0).  root = -1 (default): every rank receives every
 * record — ncclAllGather, and every rank copies the other ranks' segments to its pinned host buffer.  root >= 0: only `root` does —
 * the collective is ncclGather to root (RCCL extension; ncclAllGather when the loaded librccl lacks it) and only root copies the other
 * segments down; every other rank returns from the gather calls below with ITS OWN segment valid (all_records / clc_comm_records())
 * and the other segments unspecified.  At N = 8 the all-ranks form costs every rank a 5.5 MB device-to-host copy per step; the rooted
 * form costs the root that copy (hidden behind the next step's kernel by the pipelined call below) and the other ranks nothing.
 * Every rank must set the same root before its next gather. */
int clc_comm_set_root(clc_comm* c, int root);
typedef struct clc_comm_info {
  int32_t struct_size;                 /* sizeof(clc_comm_info) of the library that filled it (fields are only ever appended) */
  int32_t rank, world, root;
  int32_t rooted_collective_available; /* the loaded RCCL exports ncclGather */
  int32_t copies_other_ranks_to_host;  /* this rank copies the other ranks' segments to its host (no root, or it is the root) */
  int32_t step_in_flight;              /* a pipelined step has been enqueued and not yet been returned */
  int32_t pad_;
  int64_t collectives;                 /* collectives enqueued on this communicator so far */
  int64_t rooted_collectives;          /* of them ncclGather */
  int64_t host_copies;                 /* device-to-host copy commands issued by the gather calls on this rank */
  int64_t host_copy_bytes;
  int64_t pipelined_steps;
} clc_comm_info;
int clc_comm_get_info(const clc_comm* c, clc_comm_info* out);

/* Gather (see clc_comm_set_root) of the result records the LAST clc_solve_batched on the comm's handle left in device
 * memory.  This rank owns global problem indices [first_global_index, first_global_index + P_local);
 * every rank contributes exactly cap_per_rank records (P_local <= cap_per_rank, the rest padded with
 * global_index = -1).  all_records (host, world * cap_per_rank records, rank-major; contiguous shards in
 * rank order make that the global problem order) is filled on every rank that passes one; NULL skips
 * the copy — the gathered records then stay readable in the communicator's pinned host buffer,
 * clc_comm_records(), until the next gather.  With a root set, ranks other than the root hold only their own segment.  Collective call. */
int clc_gather_results(clc_comm* c, int64_t first_global_index, size_t cap_per_rank,
                       clc_result_record* all_records);
const clc_result_record* clc_comm_records(const clc_comm* c);

/* clc_solve_batched + clc_gather_results as ONE enqueue — the step of a sharded batch (BASELINE.json configs[3]) when only the
 * gathered records are wanted.  The on-chip batched kernel writes every local problem's result record, global index
 * first_global_index + k, straight into this rank's segment of the communicator's gather buffer; the all-gather (in place) and ONE
 * copy of the gathered records to the host follow on the same stream, and the host synchronises once.  No per-problem poses or
 * summaries cross PCIe and nothing is packed or copied in between; what the two-call form returns per problem is in the records
 * (pose, costs, iterations, termination), and the local shard's totals come back in `stats` (nullable).
 * poses0: host, 7 doubles per LOCAL problem of the handle's uploaded batch (clc_upload_batched), or the handle's own pinned buffer
 * (clc_batched_host_buffers) already filled.  all_records / clc_comm_records() as in clc_gather_results.  A batch that does not run as
 * the one-launch on-chip solve falls back to the two calls (stats->fused = 0).  Collective call; same error convention as
 * clc_gather_results (a rank with a local error still takes part, with padding records, and reports afterwards). */
typedef struct clc_batch_stats {
  int64_t problems;        /* local problems solved by this call (-1 in all four counters: totals unknown for this call — the call
                            * after one that failed between its launch and its bookkeeping; the records are complete either way) */
  int64_t evaluations;     /* sum of their evaluation passes (clc_summary.num_evaluations) */
  int64_t iterations;      /* sum of their LM iterations */
  int64_t not_converged;   /* of them: terminated with CLC_NO_CONVERGENCE or CLC_FAILURE */
  int32_t fused;           /* 1: one launch + in-place all-gather + one copy; 0: the two-call fall-back ran */
  int32_t pad_;
  double kernel_ms;        /* profile_events = 1: HIP event pair around the solve launch (fused form) */
  double solve_ms;         /* host wall time of the call */
} clc_batch_stats;
int clc_solve_batched_gather(clc_comm* c, const clc_options* opt, const double* poses0, int64_t first_global_index,
                             size_t cap_per_rank, clc_result_record* all_records, clc_batch_stats* stats);

/* The same step for a STREAM of steps: the call enqueues step k (kernel + collective on the solver's stream) and returns the records
 * and totals of step k-1 (*prev_records = NULL on the first call).  The device-to-host copy of step k-1's other-rank segments runs on
 * the communicator's copy stream WHILE step k's kernel runs (step k's collective waits for it by event before it overwrites the
 * segments), and the host buffers alternate between two pinned twins: *prev_records stays valid and untouched while step k runs, until
 * the NEXT pipelined call or flush (whose step writes into that twin).  clc_gather_flush completes the last step (returns CLC_OK with *records = NULL when none is in flight).  Between a
 * pipelined call and its flush the other gather calls and clc_comm_set_root refuse.  The start poses of step k are read by its kernel:
 * `poses0` is copied into the handle's pinned buffer before the launch, the caller's array is free on return.  Requires the one-launch
 * on-chip batch (otherwise CLC_ERR_INVALID_ARG: use clc_solve_batched_gather).  Collective calls, same error convention: a LOCAL error
 * of step k-1 is returned by the call that hands back its records. */
int clc_solve_batched_gather_pipelined(clc_comm* c, const clc_options* opt, const double* poses0, int64_t first_global_index,
                                       size_t cap_per_rank, const clc_result_record** prev_records, clc_batch_stats* prev_stats);
int clc_gather_flush(clc_comm* c, const clc_result_record** records, clc_batch_stats* stats);

#ifdef __cplusplus
}
#endif
#endif /* CLC_H_ */

// tests/shim/lm_shim.cpp — TEST ONLY.  Compiles the product's LM controller
// (camlasercalibratool_amd/csrc/clc_lm.hpp, the code that runs on the GPU in lm_kernel /
// batched_lm_kernel) for the host with g++, so its control flow can be checked against the
// oracle here, where there is no GPU.  The evaluation {cost, g, H} is supplied by the test
// through a callback; nothing in the product links or calls this file.
#include "../../camlasercalibratool_amd/csrc/clc_lm.hpp"

extern "C" {

typedef void (*shim_eval_fn)(const double* pose, double* cost, double* g, double* H);

int shim_lm_solve(const clc_options* opt, double* pose, clc_summary* summary, clc_iteration* trace,
                  int trace_cap, shim_eval_fn f) {
  clc::LmState s;
  clc::LmScratch w;
  clc::lm_init(s, *opt, pose);
  int guard = 0;
  while (s.status == CLC_RUNNING && guard++ < 100000) {
    double c, g[6], H[21];
    f(s.x_eval, &c, g, H);
    clc::lm_advance(s, w, *opt, trace, trace_cap, c, g, H);
  }
  clc::lm_fill_summary(s, *summary);
  for (int i = 0; i < 7; ++i) pose[i] = s.x_out[i];
  return s.status;
}

// the same controller instantiated for the 2-parameter line fit (Euclid2Manifold)
typedef void (*shim_eval2_fn)(const double* line, double* cost, double* g, double* H3);
int shim_lm2_solve(const clc_options* opt, double* line, clc_summary* summary, clc_iteration* trace, int trace_cap,
                   shim_eval2_fn f) {
  clc::LmStateT<clc::Euclid2Manifold> s;
  clc::LmScratchT<clc::Euclid2Manifold> w;
  clc::lm_init(s, *opt, line);
  int guard = 0;
  while (s.status == CLC_RUNNING && guard++ < 100000) {
    double c, g[2], H[3];
    f(s.x_eval, &c, g, H);
    clc::lm_advance(s, w, *opt, trace, trace_cap, c, g, H);
  }
  clc::lm_fill_summary(s, *summary);
  line[0] = s.x_out[0];
  line[1] = s.x_out[1];
  return s.status;
}

void shim_pose_plus(const double* x, const double* d, double* out) { clc::pose_plus(x, d, out); }
void shim_quat_to_rot(const double* q, double* R) { clc::quat_to_rot(q, R); }

}  // extern "C"

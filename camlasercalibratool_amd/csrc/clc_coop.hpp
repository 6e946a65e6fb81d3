// clc_coop.hpp — K3c: ONE problem solved by the whole GPU in ONE launch, its scan points resident on chip.
//
// The reference's ceres::Solve (src/LaseCamCalCeres.cpp:301-307) reads the observations into the ceres::Problem once and
// iterates on them in place.  The step chain (clc_kernels.hpp K3) re-streams the problem once per LM iteration — 17.4 MB
// per pass at C2, out of the Infinity Cache — and pays a kernel boundary per iteration.  Here the problem (up to
// 256 x 256 x 40 = 2.6e6 points; C2 has 1e6) is dealt ONCE to 256 co-resident workgroups, one per CU — chunk c = records
// [c n / 256, (c + 1) n / 256) in the lane layout of clc_resident.hpp: a lane holds points of one scan, PR of them in
// registers and PL in LDS — and every LM pass runs from there:
//
//   pass        every lane: moments of its points, expansion (rows_flush), wave butterfly                -> 4 x 28 per workgroup
//   exchange    (wave 0) workgroup row (28) -> board A;  the first workgroup of each group of 32 (blockIdx % 8: the workgroups
//               the dispatcher places on one XCD) sums its group's rows -> board B;  everybody reads the 8 rows of board B
//   controller  wave 0 of EVERY workgroup runs lm_advance_wave on the same 28 totals — the same arithmetic in the same
//               order everywhere, so all 256 copies of the LM state stay bit-identical (what the step chain already relies on)
//
// The exchange is the kernel boundary's replacement.  A board word is 8 bytes: 32 payload bits | a 32-bit tag unique to
// (solve, pass); a double travels as two words, written with agent-scope stores and polled with agent-scope loads until both
// tags match — no fences, no counters, every word validates itself, and boards alternate between passes (a workgroup can
// be at most one pass ahead of the slowest reader).  Measured in isolation (scripts/probes/coop_exchange_probe.hip, MI355X):
// one level (everybody reads 256 rows) 4.9 us per round, counter + fences 12 us, two levels 2.85 us, three levels 3.4 us,
// levels inside an XCD through the L2 with buffer_inv sc0 no better (3.6-4.5 us), workgroup scope (sc0) never sees the
// rows.  So the exchange costs what the launch boundary + the next launch's read of 256 rows cost (1.2 + 1.7 us); what the
// launch saves is the streaming of the points (2.4 us per pass at C2) and the per-launch prologue: 6.9-7.2 us per pass at C2
// against 8.7-9.1 (DESIGN.md K3c has the phase table).  Also measured, negative: several polls in flight, issued a fraction of a
// round trip apart, to sample the boards more often than once per round trip — C2 kernel 0.092 ms with one poll at a time, 0.106
// with two, 0.114 with four (the younger polls still own their registers when the controller starts, and they load the fabric);
// the leaders reading their group's rows at the shared L2 through RMWs (`or 0`: an agent-scope RMW executes in the XCD's L2) — 2.60
// instead of 2.83 us per round in the probe, where eight waves share the polling, but 0.098 ms in the kernel, where wave 0 alone
// issues the 64 atomics of a poll.
//
// Summation order differs from the other layouts: results agree to rounding (1e-11 on sums), the LM decisions are the
// same.  Co-residency is what makes the polling safe: 256 workgroups on 256 CUs, one each (98 KB of LDS per workgroup; the
// host checks the device against the occupancy API); every poll is bounded by a wall-clock timeout, after which the
// workgroup raises its abort flag and leaves without publishing, the host falls back to the step chain and rests this
// path on the handle for its next 1 024 solves (doubling with every further time-out).
#pragma once
#include "clc_resident.hpp"

namespace clc {

constexpr int COOP_WGS = 256, COOP_GROUPS = 8, COOP_PER_GROUP = COOP_WGS / COOP_GROUPS, COOP_ROW_WORDS = 64;
#ifndef CLC_COOP_NW
#define CLC_COOP_NW 4
#endif
// waves per workgroup, points per lane in registers + in LDS.  Four waves = ONE wave per SIMD: a pass is ~500 instructions per wave
// of which only 22 per point, so half the lanes with twice the points each issue ~30 % fewer instructions per SIMD; the LDS part
// (98 KB) also keeps a second workgroup off the CU.
constexpr int COOP_NW = CLC_COOP_NW, COOP_NL = 64 * COOP_NW;
constexpr int COOP_PR = COOP_NW == 4 ? 16 : 0, COOP_PL = COOP_NW == 4 ? 24 : 16;
constexpr unsigned long long COOP_TIMEOUT_TICKS = 2000000ull;      // 20 ms of the 100 MHz wall clock per poll
constexpr int COOP_DONE_OK = 1, COOP_DONE_ABORT = 2;
#ifndef CLC_COOP_LEAN
#define CLC_COOP_LEAN 0
#endif
constexpr bool COOP_LEAN = CLC_COOP_LEAN != 0;  // the controller's small-footprint form (clc_controller.hpp): measured, 1.5 % slower here (registers are not short)

struct CoopBoard {
  unsigned long long a[2][COOP_WGS][COOP_ROW_WORDS];     // [pass parity][workgroup][2 x 28 words, padded]
  unsigned long long b[2][COOP_GROUPS][COOP_ROW_WORDS];  // [pass parity][group]
};

__device__ __forceinline__ void coop_put(unsigned long long* row, int e, double v, unsigned int tag) {
  const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
  __hip_atomic_store(row + 2 * e, (bits & 0xFFFFFFFF00000000ull) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(row + 2 * e + 1, (bits << 32) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long coop_get(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double coop_join(unsigned long long w0, unsigned long long w1) {
  return __longlong_as_double((long long)((w0 & 0xFFFFFFFF00000000ull) | (w1 >> 32)));
}

#ifdef CLC_STAMPS
// Debug build only (scripts/r03_coop_stamps.py): shader-clock stamps of wave 0 (and wave 7 in row 1) of workgroups 0, 7 (leaders), 8 and
// 255, per pass p < COOP_STAMP_PASSES at 12 p: pass start, pass done (wave totals in LDS), row published, (leaders) group rows gathered,
// group row published, the 8 group rows arrived, totals in LDS, controller done; inside the pass: 8 pose + plane set up, 9 points done,
// 10 padding correction + expansion done.  Slot 12 * COOP_STAMP_PASSES: kernel entry, + 1: points in.
constexpr int COOP_STAMP_PASSES = 16, COOP_STAMP_PER_PASS = 12, COOP_STAMP_SLOTS = COOP_STAMP_PER_PASS * COOP_STAMP_PASSES + 2;
__device__ long long clc_coop_stamp_buf[4][2][COOP_STAMP_SLOTS];
#define COOP_STAMP(slot)                                                                                                     \
  do {                                                                                                                       \
    if (lane == 0 && (wave == 0 || wave == COOP_NW - 1) && (wg == 0 || wg == 7 || wg == 8 || wg == 255) && (slot) < COOP_STAMP_SLOTS)  \
      clc_coop_stamp_buf[wg == 0 ? 0 : wg == 7 ? 1 : wg == 8 ? 2 : 3][wave == 0 ? 0 : 1][slot] = clock64();                  \
  } while (0)
#else
#define COOP_STAMP(slot) do {} while (0)
#endif

template <bool WITH_LOSS, bool NT>
__global__ __launch_bounds__(COOP_NL) void coop_solve_kernel(
    const double* __restrict__ xyl, const unsigned int* __restrict__ res_row, const ResLane* __restrict__ lane_desc,
    const double* __restrict__ groups, const int uni_ppl, const clc_options opt, const Pose7 pose0, clc_iteration* __restrict__ trace,
    const int trace_cap, CoopBoard* __restrict__ board, const unsigned int tag0, double* __restrict__ pose_out,
    clc_summary* __restrict__ summary_out, double* __restrict__ results, int32_t* __restrict__ host_done) {
  constexpr int NW = COOP_NW, NL = COOP_NL, PR = COOP_PR, PL = COOP_PL, NP = PR + PL;
  constexpr int CH = 6, NCH = (PL + CH - 1) / CH;
  __shared__ v2d sh_pts[PL * NL];
  __shared__ double sh_state[LM_STATE_WORDS];
  __shared__ double sh_tot[32];
  __shared__ double sh_wsum[NW][NACC];
  __shared__ double sh_park[32 + (sizeof(LmScratch) + 7) / 8];
  __shared__ int sh_abort;
  const int wg = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = wg % COOP_GROUPS;            // the XCD the dispatcher places this workgroup on (round-robin)
  const bool leader = wg < COOP_GROUPS;
  clc_iteration* const tr = wg == 0 ? trace : nullptr;
  const int tr_cap = wg == 0 ? trace_cap : 0;
  LmState& st = *reinterpret_cast<LmState*>(sh_state);
#ifdef CLC_STAMPS
  int stamp_pass = 0;
  COOP_STAMP(COOP_STAMP_PER_PASS * COOP_STAMP_PASSES);
#endif
  // uni_ppl >= 0: every chunk has this many points per lane (the usual case: chunks of equal record counts) — no offset read in
  // front of the point loads
  unsigned int row0;
  int ppl;
  if (uni_ppl >= 0) {
    row0 = (unsigned int)wg * (unsigned int)uni_ppl;
    ppl = uni_ppl;
  } else {
    row0 = res_row[wg];
    ppl = __builtin_amdgcn_readfirstlane((int)(res_row[wg + 1] - row0));
  }
  // ---- this workgroup's chunk of the problem: HBM -> registers / LDS, once (as resident_solve_kernel) ----
  // The lane descriptor is requested FIRST: loads return in order, and the plane it points to is a second, dependent round trip —
  // with the descriptor queued behind the points that trip began only when the last point had arrived.
  const ResLane dl = lane_desc[(size_t)wg * NL + tid];
  const v2d* __restrict__ src = reinterpret_cast<const v2d*>(xyl) + (size_t)row0 * NL + tid;
  const int j_last = ppl > 0 ? ppl - 1 : 0;
  // (the LDS-bound rows only when the chunk has more points per lane than the registers hold — one wave-uniform branch around the
  // whole block, unconditional loads from clamped row indices inside it: at C2, 16 points per lane, there is nothing to load)
  v2d lds_v[PL];
  const bool use_lds = ppl > PR;
  if (use_lds) {
#pragma unroll
    for (int i = 0; i < PL; ++i) {
      const int j = PR + i;
      lds_v[i] = res_load<NT>(src + (size_t)(j < j_last ? j : j_last) * NL);
    }
  }
  v2d reg[PR > 0 ? PR : 1];
#pragma unroll
  for (int j = 0; j < PR; ++j) reg[j] = res_load<NT>(src + (size_t)(j < j_last ? j : j_last) * NL);
  if (use_lds) {
#pragma unroll
    for (int i = 0; i < PL; ++i) {
      v2d v = lds_v[i];
      if (PR + i >= ppl) { v[0] = 0.0; v[1] = 0.0; }
      sh_pts[i * NL + tid] = v;
    }
  }
  const double* __restrict__ gp = groups + (size_t)dl.gid * GROUP_DOUBLES;
  const int cnt = dl.cnt;
  // plane of the lane's scan (idle lanes: zeros, scale 0 — their moments are finite and expand to nothing).  One wave per SIMD has
  // the registers to keep it: no global load in front of every pass.
  double pl_nx = 0.0, pl_ny = 0.0, pl_nz = 0.0, pl_d = 0.0, pl_s = 0.0;
  if (NW == 4 && cnt > 0) {
    const v2d a = *reinterpret_cast<const v2d*>(gp);
    const v2d b = *reinterpret_cast<const v2d*>(gp + 2);
    pl_nx = a[0]; pl_ny = a[1]; pl_nz = b[0]; pl_d = b[1]; pl_s = gp[4];
  }
  if (tid == 0) {
    lm_init(st, opt, pose0.v);
    sh_abort = 0;
  }
  const double inv_lf2 = make_uniform(1.0 / (opt.loss_scale_factor * opt.loss_scale_factor));
#ifndef CLC_COOP_GRP
#define CLC_COOP_GRP 8
#endif
  // Points per basic block = independent dependency chains: a wave alone on its SIMD needs them (C2 kernel: pairs 0.0986 ms, fours
  // 0.0942, eights 0.0909).  The running cost product is renormalised once per block: eight factors 1 + r0^2/lf^2 below 2^128 each
  // (|r0| / lf < 1.8e19) cannot overflow.
  constexpr int GRP = NW == 4 ? CLC_COOP_GRP : 2;
  const int ppl_up = (ppl + GRP - 1) / GRP * GRP;
  const int ppl_eff = ppl_up < NP ? ppl_up : NP;
  __syncthreads();
  COOP_STAMP(COOP_STAMP_PER_PASS * COOP_STAMP_PASSES + 1);
#pragma unroll
  for (int j = 0; j < PR; ++j)
    if (j >= ppl) { reg[j][0] = 0.0; reg[j][1] = 0.0; }

  // one evaluation pass at st.x_eval: the wave's 28 totals -> sh_wsum[wave]   (the pass of resident_solve_kernel)
  auto pass = [&]() {
    COOP_STAMP(COOP_STAMP_PER_PASS * stamp_pass);
    int t = tid;
    asm volatile("" : "+v"(t));  // (opaque: the pass's LDS addresses are recomputed here, not hoisted out of the loop and held — or spilled — across the controller)
    double nx = pl_nx, ny = pl_ny, nz = pl_nz, pd = pl_d, ps = pl_s;
    if (NW != 4) {  // (8 waves: two waves per SIMD share 512 VGPRs — the plane is fetched again in every pass rather than held across the controller)
      const double* g2 = gp;
      asm volatile("" : "+v"(g2));
      const v2d a = *reinterpret_cast<const v2d*>(g2);
      const v2d b = *reinterpret_cast<const v2d*>(g2 + 2);
      const double s5 = g2[4];
      const bool on = cnt > 0;
      nx = on ? a[0] : 0.0; ny = on ? a[1] : 0.0; nz = on ? b[0] : 0.0; pd = on ? b[1] : 0.0; ps = on ? s5 : 0.0;
    }
    v2d buf[2][CH];
    if (ppl > PR) {
#pragma unroll
      for (int u = 0; u < CH; ++u)
        if (u < PL) buf[0][u] = sh_pts[u * NL + t];
    }
    PoseU P;
    {
      double x[7];
#pragma unroll
      for (int i = 0; i < 7; ++i) x[i] = st.x_eval[i];
      load_pose(x, P);
    }
    RowPlane q;
    rows_plane_setup(P.R, P.t, nx, ny, nz, pd, ps, q);
#ifdef CLC_STAMPS
    asm volatile("" :: "v"(q.mx), "v"(q.c0));
    COOP_STAMP(COOP_STAMP_PER_PASS * stamp_pass + 8);
#endif
    const int npad = ppl_eff - cnt;
    const double np = (double)npad;
    RowMoments M;
    rows_moments_reset<WITH_LOSS>(M);
#pragma unroll
    for (int j0 = 0; j0 < NP; j0 += GRP) {
      if (j0 < ppl) {  // wave-uniform
#pragma unroll
        for (int j = j0; j < j0 + GRP && j < NP; ++j) {
          if (j >= PR && (j - PR) % CH == 0 && (j - PR) / CH + 1 < NCH) {  // entering an LDS chunk: request the next one
            const int c1 = (j - PR) / CH + 1;
#pragma unroll
            for (int u = 0; u < CH; ++u)
              if (c1 * CH + u < PL) buf[c1 & 1][u] = sh_pts[(c1 * CH + u) * NL + t];
          }
          const v2d v = j < PR ? reg[j < PR ? j : 0] : buf[((j - PR) / CH) & 1][(j - PR) % CH];
          rows_point<WITH_LOSS>(q, inv_lf2, v[0], v[1], M, /*renorm=*/j == j0 + GRP - 1);
        }
      }
    }
#ifdef CLC_STAMPS
    asm volatile("" :: "v"(M.S0), "v"(M.Tx), "v"(M.prod));
    COOP_STAMP(COOP_STAMP_PER_PASS * stamp_pass + 9);
#endif
    // the zero padding out again: npad points (0, 0) with r0 = c0 each
    double lp = 0.0;
    {
      const double c0 = q.c0;
      if (WITH_LOSS) {
        const double sum_p = fma(c0 * c0, inv_lf2, 1.0);
        const double w_p = rcp_ge1(sum_p);
        const double cs = npad > 0 ? np * w_p : 0.0;
        M.S0 -= cs;
        M.T0 = fma(-cs, c0, M.T0);
        int e;
        const double m = frexp_pos(sum_p, e);
        lp = npad > 0 ? np * log_mant_exp(m, e) : 0.0;
      } else {
        const double cs = npad > 0 ? np : 0.0;
        M.S0 -= cs;
        M.T0 = fma(-cs, c0, M.T0);
        M.prod = fma(-cs * c0, c0, M.prod);
      }
    }
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
    rows_flush<WITH_LOSS>(q, M, acc);
    if (WITH_LOSS) acc[27] = fma(-q.s2, lp, acc[27]);
#ifdef CLC_STAMPS
    asm volatile("" :: "v"(acc[0]), "v"(acc[20]), "v"(acc[27]));
    COOP_STAMP(COOP_STAMP_PER_PASS * stamp_pass + 10);
#endif
    wave_reduce_butterfly(acc, sh_wsum[wave], lane);
    COOP_STAMP(COOP_STAMP_PER_PASS * stamp_pass + 1);
  };

  // The 28 totals of pass `k` over all workgroups -> sh_tot, on wave 0 alone (the wave that runs the controller next: no barrier and
  // no LDS round trip between the last word's arrival and the controller); fixed order: waves, group members 0-15 + 16-31, groups
  // 0-3 + 4-7.  Lane (h = lane >> 5, e = lane & 31) polls element e of half h of the rows.  The other waves only pass the barrier.
  // false (wave 0 only): a poll timed out.
  auto exchange = [&](const int k) -> bool {
    __syncthreads();  // sh_wsum of every wave
    if (wave != 0) return true;
    const unsigned int tag = tag0 + (unsigned int)k;
    const int par = k & 1;
    const int h = lane >> 5, e = (lane & 31) < NACC ? (lane & 31) : NACC - 1;
    const bool mine = lane < NACC;
    if (mine) {
      double s = sh_wsum[0][lane];
#pragma unroll
      for (int w = 1; w < NW; ++w) s += sh_wsum[w][lane];
      coop_put(board->a[par][wg], lane, s, tag);
    }
    COOP_STAMP(COOP_STAMP_PER_PASS * stamp_pass + 2);
    bool fine = true;
    if (leader) {
      // rows of this group: workgroups grp + 8 m, m = 16 h .. 16 h + 15
      unsigned long long w0[16], w1[16];
      const unsigned long long* base = &board->a[par][grp + COOP_GROUPS * 16 * h][2 * e];
      const unsigned long long t0 = wall_clock64();
      for (;;) {
        bool ok = true;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          w0[i] = coop_get(base + (size_t)(COOP_GROUPS * i) * COOP_ROW_WORDS);
          w1[i] = coop_get(base + (size_t)(COOP_GROUPS * i) * COOP_ROW_WORDS + 1);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) ok = ok && (unsigned int)w0[i] == tag && (unsigned int)w1[i] == tag;
        if (__all(ok)) break;
        if (wall_clock64() - t0 > COOP_TIMEOUT_TICKS) { fine = false; break; }
      }
      COOP_STAMP(COOP_STAMP_PER_PASS * stamp_pass + 3);
      double s = coop_join(w0[0], w1[0]);
#pragma unroll
      for (int i = 1; i < 16; ++i) s += coop_join(w0[i], w1[i]);
      const double o = __shfl_xor(s, 32, 64);
      if (mine && fine) coop_put(board->b[par][grp], lane, s + o, tag);  // (lanes < 28 are half 0: members 0-15 first)
      COOP_STAMP(COOP_STAMP_PER_PASS * stamp_pass + 4);
    }
    if (fine) {
      // the 8 group rows: groups 4 h .. 4 h + 3
      unsigned long long w0[4], w1[4];
      const unsigned long long* base = &board->b[par][4 * h][2 * e];
      const unsigned long long t0 = wall_clock64();
      for (;;) {
        bool ok = true;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          w0[i] = coop_get(base + (size_t)i * COOP_ROW_WORDS);
          w1[i] = coop_get(base + (size_t)i * COOP_ROW_WORDS + 1);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) ok = ok && (unsigned int)w0[i] == tag && (unsigned int)w1[i] == tag;
        if (__all(ok)) break;
        __builtin_amdgcn_s_sleep(1);
        if (wall_clock64() - t0 > COOP_TIMEOUT_TICKS) { fine = false; break; }
      }
      COOP_STAMP(COOP_STAMP_PER_PASS * stamp_pass + 5);
      double s = coop_join(w0[0], w1[0]);
#pragma unroll
      for (int i = 1; i < 4; ++i) s += coop_join(w0[i], w1[i]);
      const double o = __shfl_xor(s, 32, 64);
      if (mine) sh_tot[lane] = s + o;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    COOP_STAMP(COOP_STAMP_PER_PASS * stamp_pass + 6);
    return fine;
  };

  const int cap = opt.max_num_iterations + 2;
  // wave 0: exchange + controller (its last instruction is the workgroup barrier), or — after a timeout — the abort flag and a plain
  // barrier; the other waves wait at their barrier and read the flag behind it
  pass();
  if (wave == 0) {
    int lane_c = lane;
    asm volatile("" : "+v"(lane_c));
    if (exchange(0)) lm_advance_wave<true, COOP_LEAN>(st, opt, tr, tr_cap, sh_tot, sh_park, lane_c);
    else { sh_abort = 1; __syncthreads(); }
  } else {
    exchange(0);
    __syncthreads();
  }
#ifdef CLC_STAMPS
  COOP_STAMP(7);
  stamp_pass = 1;
#endif
  for (int k = 0; k < cap && sh_abort == 0 && st.status == CLC_RUNNING; ++k) {  // (status, flag: published before the barrier)
    pass();
    if (wave == 0) {
      int lane_c = lane;
      asm volatile("" : "+v"(lane_c));
      if (exchange(k + 1)) lm_advance_wave<false, COOP_LEAN>(st, opt, tr, tr_cap, sh_tot, sh_park, lane_c);
      else { sh_abort = 1; __syncthreads(); }
    } else {
      exchange(k + 1);
      __syncthreads();
    }
#ifdef CLC_STAMPS
    COOP_STAMP(COOP_STAMP_PER_PASS * stamp_pass + 7);
    ++stamp_pass;
#endif
  }
  const bool ok = sh_abort == 0;
  if (wg == 0 && wave == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane == 0) {
      if (ok) {
        if (st.status == CLC_RUNNING) st.status = CLC_FAILURE;  // unreachable: the controller stops at the iteration cap
        batched_write_outcome(st, 0, pose_out, summary_out, results);
      }
      __hip_atomic_store(host_done, ok ? COOP_DONE_OK : COOP_DONE_ABORT, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

}  // namespace clc

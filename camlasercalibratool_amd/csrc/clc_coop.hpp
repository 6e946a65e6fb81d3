// clc_coop.hpp — K3c: ONE problem solved by the whole GPU in ONE launch, its scan points resident on chip.
//
// The reference's ceres::Solve (src/LaseCamCalCeres.cpp:301-307) reads the observations into the ceres::Problem once and
// iterates on them in place.  The step chain (clc_kernels.hpp K3) re-streams the problem once per LM iteration — 17.4 MB
// per pass at C2, out of the Infinity Cache — and pays a kernel boundary per iteration.  Here the problem (up to
// 256 x 256 x 40 = 2.6e6 points; C2 has 1e6) is dealt ONCE to 256 co-resident workgroups, one per CU — chunk c = records
// [c n / 256, (c + 1) n / 256) in the lane layout of clc_resident.hpp: a lane holds points of one scan, PR of them in
// registers and PL in LDS — and every LM pass runs from there:
//
//   pass        every lane: m = R^T n, c0 from the published rotation + translation (12 LDS reads), moments of its points,
//               expansion (rows_flush), the first two levels of the wave butterfly (28 -> 7 registers per lane)  -> LDS
//   row         wave w finishes accumulators 7 w .. 7 w + 6 over the 4 x 16 partial lanes and publishes them -> board A
//   exchange    (wave 0) the first workgroup of each group of 32 (blockIdx % 8: the workgroups the dispatcher places on one
//               XCD) sums its group's rows -> board B;  everybody reads the 8 rows of board B
//   controller  wave 0 of EVERY workgroup advances the LM state it keeps in its registers (clc_lmuni.hpp) on the same 28 totals —
//               the same arithmetic in the same order everywhere, so all 256 copies stay bit-identical (what the step chain
//               already relies on) — and publishes the next rotation + translation + status for its workgroup
//
// The exchange is the kernel boundary's replacement.  A board word is 8 bytes: 32 payload bits | a 32-bit tag unique to
// (solve, pass); a double travels as two words, written with agent-scope stores and polled with agent-scope loads until both
// tags match — no fences, no counters, every word validates itself, and boards alternate between passes (a workgroup can
// be at most one pass ahead of the slowest reader).  Measured in isolation (scripts/probes/coop_exchange_probe.hip, MI355X):
// one level (everybody reads 256 rows) 4.9 us per round, counter + fences 12 us, two levels 2.85 us, three levels 3.4 us,
// levels inside an XCD through the L2 with buffer_inv sc0 no better (3.6-4.5 us), workgroup scope (sc0) never sees the
// rows.  Also measured, negative: several polls in flight, issued a fraction of a round trip apart (C2 kernel 0.092 ms with one
// poll at a time, 0.106 with two, 0.114 with four); the leaders reading their group's rows at the shared L2 through RMWs.
//
// Round 4 (DESIGN.md K3c has the phase table before / after): the per-pass chain lost its LDS-resident controller (5 400 cycles ->
// lm_regs_*: state in registers, DPP broadcasts, what does not need the totals computed while the exchange is in flight), the
// per-lane quaternion -> rotation set-up, the analytic padding correction (a logarithm + a reciprocal per lane and pass: the
// padded slots of the last blocks are masked instead — r0 = 0: cost factor exactly 1; weight 0), and the
// [wave butterfly, barrier, wave-0 sum of four rows] tail (each wave now finishes and publishes a quarter of the row).
//
// Summation order differs from the other layouts: results agree to rounding (1e-11 on sums), the LM decisions are the
// same.  Co-residency is what makes the polling safe: 256 workgroups on 256 CUs, one each (112 KB of LDS per workgroup; the
// host checks the device against the occupancy API).  Every poll is bounded by a wall-clock timeout — 200 us for the first pass,
// which doubles as the arrival census (a workgroup that is not resident has not published its row), 1 ms afterwards; a workgroup
// that times out raises the launch's abort word on the board, which every poll loop reads, so the whole grid leaves within
// microseconds of the first timeout; the host falls back to the step chain and rests this path on the handle (16 solves, doubling
// with every further abort).
#pragma once
#include "clc_lmuni.hpp"
#include "clc_resident.hpp"

namespace clc {

constexpr int COOP_WGS = 256, COOP_GROUPS = 8, COOP_PER_GROUP = COOP_WGS / COOP_GROUPS, COOP_ROW_WORDS = 64;
// The small form: problems of few points per lane of 32 workgroups (the host decides up to how many: abi_layouts.hip, kCoopSmallMaxPpl)
// run on 32 workgroups with a ONE-hop exchange (every workgroup reads all 32 rows): a pass of ~4.5-5 us instead of ~5.5 where the
// 256-workgroup form has 1-2 points per lane.
constexpr int COOP_SMALL_WGS = 32;
// Row strides of the two boards in 8-byte words (>= COOP_ROW_WORDS: a row is 28 elements x 16 bytes = 448 bytes).  Everybody polls the
// 8 group rows at once; whether rows 512 bytes apart queue behind each other in one memory channel was measured with 1 KB and 4 KB
// strides (group rows alone, and both boards): no difference (C2 kernel 82.5-83.9 us each way) — the hops are latency, not a hot spot.
#ifndef CLC_COOP_A_STRIDE
#define CLC_COOP_A_STRIDE 64
#endif
#ifndef CLC_COOP_B_STRIDE
#define CLC_COOP_B_STRIDE 64
#endif
constexpr int COOP_A_STRIDE = CLC_COOP_A_STRIDE, COOP_B_STRIDE = CLC_COOP_B_STRIDE;
// Four waves per workgroup = ONE wave per SIMD: a pass is ~400 instructions per wave of which only 22 per point, so half the
// lanes with twice the points each issue ~30 % fewer instructions per SIMD than two waves per SIMD; the LDS part (98 KB) also
// keeps a second workgroup off the CU.  Points per lane in registers + in LDS.
constexpr int COOP_NW = 4, COOP_NL = 64 * COOP_NW;  // the point waves / lanes of a workgroup ...
constexpr int COOP_THREADS = COOP_NL + 64;           // ... + the controller wave
constexpr int COOP_STATUS_ABORT = -1;                // published instead of an LM status when the launch gives up
constexpr int COOP_PR = 16, COOP_PL = 24;
// points that carry z (p.z != 0 in some record: 24 bytes per slot instead of 16): the same registers and the same 98 KB of LDS hold
// 10 + 16 points per lane — 65 536 lanes x 26 = 1.7e6 observations
constexpr int COOP_PR_Z = 10, COOP_PL_Z = 16;
constexpr unsigned long long COOP_CENSUS_TICKS = 20000ull;    // first pass: 200 us of the 100 MHz wall clock per poll
constexpr unsigned long long COOP_TIMEOUT_TICKS = 100000ull;  // later passes: 1 ms
constexpr int COOP_DONE_OK = 1, COOP_DONE_ABORT = 2;
#ifndef CLC_COOP_POLL_SLEEP
#define CLC_COOP_POLL_SLEEP 1   // s_sleep units (64 cycles) between two polls of a board
#endif
// (round 5, after the pose / totals reads became ds_read again and lmu_pre ~400 cycles shorter: C2 kernel flat within noise, 76.0-77.7 us,
// for D1 600-1100 x D2 1200-2800; D2 3500 +1 us, D1 300 +10 us — scripts/r05_coop_sweep.sh)
#ifndef CLC_COOP_D1
#define CLC_COOP_D1 900         // shader cycles after barrier A at which a leader first looks at its group's rows
#endif
#ifndef CLC_COOP_D2
#define CLC_COOP_D2 2200        // ... at which everybody first looks at the 8 group rows
#endif

#ifndef CLC_COOP_GRP
#define CLC_COOP_GRP 8          // points per basic block of a pass (see the point loop)
#endif
#ifndef CLC_COOP_REPLICAS
#define CLC_COOP_REPLICAS 1
#endif
// Copies of the group-row board: a leader writes its group row into every copy, workgroup w reads copy w % COOP_REPLICAS.  All 256
// workgroups polling the SAME 56 cache lines made every poll queue behind ~280 requests per line, and the leaders' writes with them.
constexpr int COOP_REPLICAS = CLC_COOP_REPLICAS;
constexpr unsigned long long COOP_ABORT_CHECK_TICKS = 2000ull;  // a waiting workgroup starts reading the abort word after 20 us
struct CoopBoard {
  unsigned long long a[2][COOP_WGS][COOP_A_STRIDE];                    // [pass parity][workgroup][2 x 28 words, padded]
  unsigned long long b[2][COOP_REPLICAS][COOP_GROUPS][COOP_B_STRIDE];  // [pass parity][copy][group]
  unsigned long long ctl[8];  // [0]: tag0 of a launch that aborted (read by workgroups that have waited 20 us); [1], [2]: tuning hook
};

// One element of a row = 16 bytes {tag, high 32 bits, tag, low 32 bits} = the two self-validating 8-byte words, moved with ONE
// agent-scope (sc1) 16-byte buffer access: half the memory instructions of a poll.  (Each aligned 8-byte half is written and read
// atomically; a reader accepts an element only when both tags match.)
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
// cache-policy argument of the raw buffer builtins on gfx94x/gfx950: bit 4 = sc1 (agent scope).  (To LLVM a raw buffer load is a pure
// function of its operands: inside a polling loop it is loop-invariant and gets hoisted — the loop then spins on one sample.  The
// polling loops therefore pass their base offset through an empty asm in every iteration.)
constexpr int COOP_SC1 = 16;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t coop_rsrc(CoopBoard* board) {
  return __builtin_amdgcn_make_buffer_rsrc(board, 0, (int)sizeof(CoopBoard), 0x00020000);
}
__device__ __forceinline__ void coop_put(__amdgpu_buffer_rsrc_t rs, const unsigned int row_byte, const int e, const double v, const unsigned int tag) {
  v4u w;
  w[0] = tag; w[1] = (unsigned int)__double2hiint(v); w[2] = tag; w[3] = (unsigned int)__double2loint(v);
  __builtin_amdgcn_raw_buffer_store_b128(w, rs, row_byte + 16u * (unsigned int)e, 0, COOP_SC1);
}
__device__ __forceinline__ v4u coop_get(__amdgpu_buffer_rsrc_t rs, const unsigned int byte) {
  return __builtin_amdgcn_raw_buffer_load_b128(rs, byte, 0, COOP_SC1);
}
__device__ __forceinline__ bool coop_valid(const v4u w, const unsigned int tag) { return w[0] == tag && w[2] == tag; }
__device__ __forceinline__ double coop_value(const v4u w) { return __hiloint2double((int)w[1], (int)w[3]); }
__device__ __forceinline__ unsigned int coop_row_a(const int par, const int wg) {
  return (unsigned int)(offsetof(CoopBoard, a) + ((size_t)(par * COOP_WGS + wg) * COOP_A_STRIDE) * 8);
}
__device__ __forceinline__ unsigned int coop_row_b(const int par, const int copy, const int grp) {
  return (unsigned int)(offsetof(CoopBoard, b) + ((size_t)((par * COOP_REPLICAS + copy) * COOP_GROUPS + grp) * COOP_B_STRIDE) * 8);
}
// sleep (no issue slots, no memory traffic) until the shader clock reaches t
__device__ __forceinline__ void coop_wait_until(const long long t) {
  while (clock64() < t) __builtin_amdgcn_s_sleep(2);
}

#ifdef CLC_STAMPS
// Debug build only (scripts/stamps_coop.py): shader-clock stamps of point wave 0 and the controller wave (row 0: they stamp disjoint
// slots; point wave 3 in row 1) of workgroups 0, 7 (leaders), 8 and 255, per pass p < COOP_STAMP_PASSES at 12 p: 0 pass start, 1 pass done (partials in LDS), 2 row published, 3 (leaders) group rows
// gathered, 4 group row published, 5 the 8 group rows arrived, 6 totals in LDS, 7 controller done (behind its barrier); inside the pass:
// 8 pose + plane set up, 9 points done, 10 expansion done; 11 lmu_pre done.  Slot 12 * COOP_STAMP_PASSES: kernel entry, + 1: points in.
constexpr int COOP_STAMP_PASSES = 16, COOP_STAMP_PER_PASS = 12, COOP_STAMP_SLOTS = COOP_STAMP_PER_PASS * COOP_STAMP_PASSES + 2;
static __device__ long long clc_coop_stamp_buf[4][2][COOP_STAMP_SLOTS];
#define COOP_STAMP(slot)                                                                                                     \
  do {                                                                                                                       \
    if (lane == 0 && (wave == 0 || wave >= COOP_NW - 1) && (wg == 0 || wg == 7 || wg == 8 || wg == 255) && (slot) < COOP_STAMP_SLOTS)  \
      clc_coop_stamp_buf[wg == 0 ? 0 : wg == 7 ? 1 : wg == 8 ? 2 : 3][wave == COOP_NW - 1 ? 1 : 0][slot] = clock64();          \
  } while (0)
#else
#define COOP_STAMP(slot) do {} while (0)
#endif

// The first two levels of wave_reduce_butterfly (clc_device.hpp): 28 accumulators -> 7 registers; lane l of 16-lane row rho holds a
// partial sum (over lanes l & 15 + 16 k) of accumulator i + 7 rho in u[i].
__device__ __forceinline__ void wave_reduce_to_rows(double (&acc)[NACC], double (&u)[7]) {
  double r[14];
#pragma unroll
  for (int i = 0; i < 14; ++i) {
    double x = acc[i], y = acc[i + 14];
    swap_halves(x, y);
    r[i] = x + y;
  }
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    double x = r[i], y = r[i + 7];
    swap_rows(x, y);
    u[i] = x + y;
  }
}

template <bool WITH_LOSS, bool NT, bool WITH_Z = false, bool ONE_HOP = false>
__global__ __launch_bounds__(COOP_THREADS) void coop_solve_kernel(
    const double* __restrict__ xyl, const double* __restrict__ zl, const unsigned int* __restrict__ res_row, const ResLane* __restrict__ lane_desc,
    const double* __restrict__ groups, const int uni_ppl, const clc_options opt, const Pose7 pose0, clc_iteration* __restrict__ trace,
    const int trace_cap, CoopBoard* __restrict__ board, const unsigned int tag0, double* __restrict__ pose_out,
    clc_summary* __restrict__ summary_out, double* __restrict__ results, int32_t* __restrict__ host_done, const int n_wgs) {
  constexpr int NW = COOP_NW, NL = COOP_NL, PR = WITH_Z ? COOP_PR_Z : COOP_PR, PL = WITH_Z ? COOP_PL_Z : COOP_PL, NP = PR + PL;
  constexpr int CH = WITH_Z ? 4 : 6, NCH = (PL + CH - 1) / CH;
  static_assert(NW == 4, "the row tail maps the four 16-lane rows of a wave onto the four point waves");
  using Moments = typename std::conditional<WITH_Z, RowMoments3, RowMoments>::type;
  __shared__ v2d sh_pts[PL * NL];
  __shared__ double sh_ptz[WITH_Z ? PL * NL : 1];  // the z of the LDS-held slots (points off the lidar plane)
  __shared__ double sh_state[LM_STATE_WORDS];
  __shared__ __attribute__((aligned(16))) double sh_tot[64];  // two buffers of 32: the totals of the passes alternate (clc_lmuni.hpp)
  __shared__ double sh_red[NW * 7 * 64];  // [wave][register 0..6][lane]: the partials of wave_reduce_to_rows
  __shared__ __attribute__((aligned(16))) double sh_pub[LM_PUB_WORDS];
  const int wg = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cap = opt.max_num_iterations + 2;
#ifdef CLC_STAMPS
  int stamp_pass = 0;
  COOP_STAMP(COOP_STAMP_PER_PASS * COOP_STAMP_PASSES);
#endif

  if (wave == NW) {
    // =====================================================================================================================
    // The controller wave: no scan points, the LM state in its registers (clc_lmuni.hpp).  Per pass: [barrier A: the point waves'
    // partials are in LDS] -> (leaders: gather the group's rows, publish the group row) -> lmu_pre while the rows travel -> the 8
    // group rows -> totals -> lmu_post -> [barrier B: pose + status published].
    // =====================================================================================================================
    // ONE_HOP = false: COOP_WGS workgroups, two hops (8 groups of 32).  ONE_HOP = true: COOP_SMALL_WGS (32) workgroups — the problem is
    // small enough for them —, every one of them gathers all 32 rows itself (what a group leader does) and nobody publishes a group
    // row.  (A compile-time switch: as a run-time one it cost the 256-workgroup form 4 % at C2.)
    constexpr bool one_hop = ONE_HOP;
    constexpr int groups = ONE_HOP ? 1 : COOP_GROUPS;
    const int grp = ONE_HOP ? 0 : wg % COOP_GROUPS;  // (the XCD the dispatcher places this workgroup on: round-robin)
    const bool leader = ONE_HOP || wg < COOP_GROUPS;
    clc_iteration* const tr = wg == 0 ? trace : nullptr;
    const int tr_cap = wg == 0 ? trace_cap : 0;
    LmState& st = *reinterpret_cast<LmState*>(sh_state);
    const __amdgpu_buffer_rsrc_t rs = coop_rsrc(board);
    LmU S;
    {
      double x0[7];
#pragma unroll
      for (int i = 0; i < 7; ++i) x0[i] = pose0.v[i];
      lmu_init(S, st, opt, x0, lane);
      lmu_publish(sh_pub, x0, CLC_RUNNING, lane);
    }
    __syncthreads();  // barrier 0: the first pose is published
    bool aborted = false;

    // When to look: a poll is 17 (5) buffer loads per lane and takes ~1 us to come back, and a poll that goes out before the words it
    // looks for costs a whole round trip more — with the leaders' first poll at barrier A the pass was 1.1 us longer than with that poll
    // 1 000 cycles later.  What follows barrier A does not depend on the problem (the point waves finish and publish the row in ~950
    // cycles), so the first polls go out at fixed offsets from it (measured flat within 1 % for 500-1 000 / 2 000-4 500 cycles;
    // an adaptive offset taken from the previous pass ran away: a poll that succeeds at once says nothing about how early it could
    // have gone).
    // (one-hop form: lmu_pre runs first, ~1 100 cycles; the rows are published ~950 cycles after barrier A and take a hop to become
    // visible — first look at 1 900, measured 800...2 200: 4.6-4.9 us per pass, the minimum here)
    const long long d1 = board->ctl[1] ? (long long)board->ctl[1] : (ONE_HOP ? 1900 : CLC_COOP_D1), d2 = board->ctl[2] ? (long long)board->ctl[2] : CLC_COOP_D2;
    for (int k = 0;; ++k) {
      __syncthreads();  // barrier A
      // ---- the 28 totals of pass `k` over all workgroups -> sh_tot[1 - S.hx]; fixed order: group members 0-15 + 16-31, groups
      // 0-3 + 4-7.  Lane (h = lane >> 5, e = lane & 31) polls element e of half h of the rows. ----
      const unsigned int tag = tag0 + (unsigned int)k;
      const int par = k & 1;
      const int h = lane >> 5, e = (lane & 31) < NACC ? (lane & 31) : NACC - 1;
      const bool mine = lane < NACC;
      const unsigned long long limit = k == 0 ? COOP_CENSUS_TICKS : COOP_TIMEOUT_TICKS;
      bool fine = true;
      const long long t_a = clock64();
      // What does not need the totals (lmu_pre), while the rows travel: a follower has the whole of hop 1 for it (and looks at the group
      // rows afterwards); in the one-hop form it takes the place of the wait for the rows' publication.  (The leaders of the two-hop
      // form: behind their first look at the group rows, below.)
      if (ONE_HOP) {
        if (k > 0) lmu_pre(S, st, opt, sh_tot, tr, tr_cap, lane);
        COOP_STAMP(COOP_STAMP_PER_PASS * stamp_pass + 11);
      }
      if (leader) {
        coop_wait_until(t_a + d1);
        // rows of this group: workgroups grp + 8 m, m = 16 h .. 16 h + 15
        v4u w[16];
        const unsigned int base = coop_row_a(par, grp + groups * 16 * h) + 16u * (unsigned int)e;
        constexpr unsigned int row_step = (unsigned int)(groups * COOP_A_STRIDE * 8);
        const unsigned long long t0 = wall_clock64();
        for (;;) {
          bool ok = true;
          unsigned int bo = base;
          asm volatile("" : "+v"(bo));  // (a fresh sample every iteration)
#pragma unroll
          for (int i = 0; i < 16; ++i) w[i] = coop_get(rs, bo + (unsigned int)i * row_step);
#pragma unroll
          for (int i = 0; i < 16; ++i) ok = ok && coop_valid(w[i], tag);
          if (__all(ok)) break;
          const unsigned long long waited = wall_clock64() - t0;
          if (waited > COOP_ABORT_CHECK_TICKS) {  // (the abort word is ONE line for 256 readers: nobody looks at it while things are well)
            unsigned int co = (unsigned int)offsetof(CoopBoard, ctl);
            asm volatile("" : "+v"(co));
            if (waited > limit || coop_get(rs, co)[0] == tag0) { fine = false; break; }
          }
          __builtin_amdgcn_s_sleep(CLC_COOP_POLL_SLEEP);
        }
        COOP_STAMP(COOP_STAMP_PER_PASS * stamp_pass + 3);
        double s = coop_value(w[0]);
#pragma unroll
        for (int i = 1; i < 16; ++i) s += coop_value(w[i]);
        double s_lo = s, s_hi = s;
        swap_halves(s_lo, s_hi);  // lanes < 32: own half (members 0-15) + the other half's (16-31)
        if (one_hop) {  // the group IS the problem: these are the totals
          if (mine) sh_tot[32 * (1 - S.hx) + lane] = s_lo + s_hi;
        } else if (mine && fine) {
          int so = lane;
          asm volatile("" : "+v"(so));  // (the store's lane offset computed HERE: hoisted out of the pass loop it was spilled, and its reload sat in front of the store)
#pragma unroll
          for (int r = 0; r < COOP_REPLICAS; ++r) coop_put(rs, coop_row_b(par, (r + grp) % COOP_REPLICAS, grp), so, s_lo + s_hi, tag);
        }
        COOP_STAMP(COOP_STAMP_PER_PASS * stamp_pass + 4);
      }
      // A leader of the two-hop form has just published its group row: its first look at the 8 group rows goes out BEFORE its lmu_pre
      // and is evaluated behind it (lmu_pre is about as long as the hop: issued after it, that poll cost the leaders, and with them
      // every workgroup's next pass, ~700 cycles).
      if (!ONE_HOP) {
        if (k > 0 && !leader) lmu_pre(S, st, opt, sh_tot, tr, tr_cap, lane);
        if (!leader) COOP_STAMP(COOP_STAMP_PER_PASS * stamp_pass + 11);
      }
      if (fine && !one_hop) {
        // the 8 group rows: groups 4 h .. 4 h + 3
        coop_wait_until(t_a + d2);
        const unsigned int base = coop_row_b(par, wg % COOP_REPLICAS, 4 * h) + 16u * (unsigned int)e;
        const unsigned long long t0 = wall_clock64();
        bool pre_due = k > 0 && leader;
        // One look in flight at a time.  Measured and dropped: 2-3 looks in flight 256 cycles apart (6.8-7.1 us per pass against
        // 6.2: the extra requests slow the rows they look for), copies of the group-row board (8 or 32: the leaders' extra stores
        // cost more than the shorter queues save), a leader taking its own group row from registers with its first look issued
        // before its store (+0.15 us: that look goes out too early, see d2).
        v4u w[4];
        for (;;) {
          bool ok = true;
          unsigned int bo = base;
          asm volatile("" : "+v"(bo));  // (a fresh sample every iteration)
#pragma unroll
          for (int i = 0; i < 4; ++i) w[i] = coop_get(rs, bo + (unsigned int)(i * COOP_B_STRIDE * 8));
          if (pre_due) {  // (wave-uniform; the loads above stay in flight: nothing below touches their registers)
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            lmu_pre(S, st, opt, sh_tot, tr, tr_cap, lane);
            __builtin_amdgcn_sched_barrier(0);
            COOP_STAMP(COOP_STAMP_PER_PASS * stamp_pass + 11);
            pre_due = false;
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) ok = ok && coop_valid(w[i], tag);
          if (__all(ok)) break;
          const unsigned long long waited = wall_clock64() - t0;
          if (waited > COOP_ABORT_CHECK_TICKS) {
            unsigned int co = (unsigned int)offsetof(CoopBoard, ctl);
            asm volatile("" : "+v"(co));
            if (waited > limit || coop_get(rs, co)[0] == tag0) { fine = false; break; }
          }
          __builtin_amdgcn_s_sleep(CLC_COOP_POLL_SLEEP);
        }
        COOP_STAMP(COOP_STAMP_PER_PASS * stamp_pass + 5);
        double s = coop_value(w[0]);
#pragma unroll
        for (int i = 1; i < 4; ++i) s += coop_value(w[i]);
        double s_lo = s, s_hi = s;
        swap_halves(s_lo, s_hi);  // lanes < 32: s_lo = own half, s_hi = the other half's sum (two v_permlane32_swap: a third of a ds_bpermute shuffle)
        if (mine) sh_tot[32 * (1 - S.hx) + lane] = s_lo + s_hi;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      COOP_STAMP(COOP_STAMP_PER_PASS * stamp_pass + 6);
      if (fine) {
        lmu_post(k == 0, S, opt, sh_tot, sh_pub, tr, tr_cap, lane);
        if (S.status == CLC_RUNNING && k >= cap) {  // unreachable: the controller stops at the iteration cap
          S.status = CLC_FAILURE;
          lmu_publish_status(sh_pub, CLC_FAILURE, lane);
        }
      } else {
        // a poll timed out, or somebody else's did: the launch's abort word goes up (every poll loop reads it) and everybody leaves
        if (lane == 0) __hip_atomic_store(reinterpret_cast<unsigned int*>(&board->ctl[0]), tag0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lmu_publish_status(sh_pub, COOP_STATUS_ABORT, lane);
        aborted = true;
      }
      COOP_STAMP(COOP_STAMP_PER_PASS * stamp_pass + 7);
      __syncthreads();  // barrier B: pose + status published
      if (aborted || S.status != CLC_RUNNING) break;
#ifdef CLC_STAMPS
      ++stamp_pass;
#endif
    }
    if (wg == 0) {
      if (!aborted) {
        lmu_finish(S, st, opt, sh_tot, tr, tr_cap, lane);
        if (lane == 0) batched_write_outcome(st, 0, pose_out, summary_out, results);
      }
      if (lane == 0) __hip_atomic_store(host_done, aborted ? COOP_DONE_ABORT : COOP_DONE_OK, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;
  }

  // =======================================================================================================================
  // The four point waves.  (Raised priority: the controller wave shares SIMD 0 with point wave 0, and what it computes while the
  // rows travel — lmu_pre — must not take issue slots from that wave's share of the row: measured, without it the exchange of every
  // pass began ~1 000 cycles later.)
  // =======================================================================================================================
  __builtin_amdgcn_s_setprio(2);
  const __amdgpu_buffer_rsrc_t rs_p = coop_rsrc(board);
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
  const double* __restrict__ srcz = WITH_Z ? zl + (size_t)row0 * NL + tid : nullptr;  // (z rows: j-major like the (x, y) rows, 8 bytes per slot)
  v2d lds_v[PL];
  double lds_z[WITH_Z ? PL : 1];
  // (keyed on the slots a pass TOUCHES, not on ppl: a pass walks whole blocks of CLC_COOP_GRP slots, and where PR is not a multiple of
  // the block — the z form: 10 — the block that holds point 9 or 10 runs on into LDS slots; those must then hold zeros, not
  // whatever the last launch left there)
  const bool use_lds = (ppl + CLC_COOP_GRP - 1) / CLC_COOP_GRP * CLC_COOP_GRP > PR;
  if (use_lds) {
#pragma unroll
    for (int i = 0; i < PL; ++i) {
      const int j = PR + i;
      lds_v[i] = res_load<NT>(src + (size_t)(j < j_last ? j : j_last) * NL);
      if (WITH_Z) lds_z[i] = srcz[(size_t)(j < j_last ? j : j_last) * NL];
    }
  }
  v2d reg[PR];
  double regz[WITH_Z ? PR : 1];
#pragma unroll
  for (int j = 0; j < PR; ++j) {
    reg[j] = res_load<NT>(src + (size_t)(j < j_last ? j : j_last) * NL);
    if (WITH_Z) regz[j] = srcz[(size_t)(j < j_last ? j : j_last) * NL];
  }
  if (use_lds) {
#pragma unroll
    for (int i = 0; i < PL; ++i) {
      v2d v = lds_v[i];
      double vz = WITH_Z ? lds_z[i] : 0.0;
      if (PR + i >= ppl) { v[0] = 0.0; v[1] = 0.0; vz = 0.0; }
      sh_pts[i * NL + tid] = v;
      if (WITH_Z) sh_ptz[i * NL + tid] = vz;
    }
  }
  const double* __restrict__ gp = groups + (size_t)dl.gid * GROUP_DOUBLES;
  const int cnt = dl.cnt;
  // plane of the lane's scan (idle lanes: zeros, scale 0 — their moments are finite and expand to nothing).  One wave per SIMD has
  // the registers to keep it: no global load in front of every pass.
  double pl_nx = 0.0, pl_ny = 0.0, pl_nz = 0.0, pl_d = 0.0, pl_s2 = 0.0;
  if (cnt > 0) {
    const v2d a = *reinterpret_cast<const v2d*>(gp);
    const v2d b = *reinterpret_cast<const v2d*>(gp + 2);
    const double s = gp[4];
    pl_nx = a[0]; pl_ny = a[1]; pl_nz = b[0]; pl_d = b[1]; pl_s2 = s * s;
  }
  const double inv_lf2 = make_uniform(1.0 / (opt.loss_scale_factor * opt.loss_scale_factor));
  // Points per basic block = independent dependency chains: a wave alone on its SIMD needs them (C2 kernel: pairs 0.0986 ms, fours
  // 0.0942, eights 0.0909).  The running cost product is renormalised once per block: eight factors 1 + r0^2/lf^2 below 2^128 each
  // (|r0| / lf < 1.8e19) cannot overflow.
  constexpr int GRP = CLC_COOP_GRP;
  const int ppl_up = (ppl + GRP - 1) / GRP * GRP;
  const int ppl_eff = ppl_up < NP ? ppl_up : NP;
  // Padding: a lane processes ppl_eff slots, its last ppl_eff - cnt of them zeros.  In the blocks that some lane of the wave has
  // padding in, a padded slot is evaluated with r0 = 0 (cost factor exactly 1) and weight 0 (rows_point_masked): no moment moves.
  // (An idle lane — no scan, plane and scale zero — needs none of it.)
#ifndef CLC_COOP_PAD_ANALYTIC
#define CLC_COOP_PAD_ANALYTIC 0
#endif
  constexpr bool PAD_ANALYTIC = CLC_COOP_PAD_ANALYTIC != 0;
  static_assert(!(PAD_ANALYTIC && WITH_Z), "the analytic padding correction is written for (x, y) slots only");  // (as clc_resident.hpp: the padding's contribution taken out analytically instead of masked slots)
  const int cnt_m = cnt > 0 ? cnt : ppl_eff;
  int cmin = cnt_m;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const int o2 = __shfl_xor(cmin, off, 64);
    cmin = o2 < cmin ? o2 : cmin;
  }
  cmin = __builtin_amdgcn_readfirstlane(cmin);
  __syncthreads();  // barrier 0
  COOP_STAMP(COOP_STAMP_PER_PASS * COOP_STAMP_PASSES + 1);
#pragma unroll
  for (int j = 0; j < PR; ++j)
    if (j >= ppl) { reg[j][0] = 0.0; reg[j][1] = 0.0; if (WITH_Z) regz[j] = 0.0; }

  for (int k = 0;; ++k) {
    COOP_STAMP(COOP_STAMP_PER_PASS * stamp_pass);
    int t = tid;
    asm volatile("" : "+v"(t));  // (opaque: the pass's LDS addresses are recomputed here, not hoisted out of the loop and held)
    // ---- the pose of this pass (rotation + translation) and whether there is one: 7 broadcast 16-byte reads ----
    RowPlane q;
    {
      // (ONE batch of ds_read_b128: the status word rides in the seventh; all seven are back before the branch)
      const lds_cv2d* pb = lds_opaque(sh_pub);
      const v2d p0 = pb[0], p1 = pb[1], p2 = pb[2], p3 = pb[3], p4 = pb[4], p5 = pb[5], p6 = pb[6];
      int status = __double2loint(p6[0]);
      // (left alone the compiler reads the status word first, waits, branches and only then asks for the other six: two LDS round
      // trips at the head of every pass; the status is made to depend on all of them)
      asm volatile("" : "+v"(status) : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5));
      if (status != CLC_RUNNING) break;  // (wave-uniform: terminated, or the launch aborted)
      // R row-major = p0[0] p0[1] p1[0] | p1[1] p2[0] p2[1] | p3[0] p3[1] p4[0];  t = p4[1] p5[0] p5[1]
      q.nx = pl_nx; q.ny = pl_ny; q.nz = pl_nz; q.s2 = pl_s2;
      q.mx = fma(p3[0], pl_nz, fma(p1[1], pl_ny, p0[0] * pl_nx));
      q.my = fma(p3[1], pl_nz, fma(p2[0], pl_ny, p0[1] * pl_nx));
      q.mz = fma(p4[0], pl_nz, fma(p2[1], pl_ny, p1[0] * pl_nx));
      q.c0 = fma(p5[1], pl_nz, fma(p5[0], pl_ny, fma(p4[1], pl_nx, pl_d)));
    }
    v2d buf[2][CH];
    double bufz[2][WITH_Z ? CH : 1];
    if (use_lds) {
#pragma unroll
      for (int u = 0; u < CH; ++u)
        if (u < PL) {
          buf[0][u] = sh_pts[u * NL + t];
          if (WITH_Z) bufz[0][u] = sh_ptz[u * NL + t];
        }
    }
#ifdef CLC_STAMPS
    asm volatile("" :: "v"(q.mx), "v"(q.c0));
    COOP_STAMP(COOP_STAMP_PER_PASS * stamp_pass + 8);
#endif
    Moments M;
    lane_moments_reset<WITH_LOSS>(M);
#pragma unroll
    for (int j0 = 0; j0 < NP; j0 += GRP) {
      if (j0 < ppl) {  // wave-uniform
        if (PAD_ANALYTIC || j0 + GRP <= cmin) {  // wave-uniform: no lane of the wave has padding in this block
#pragma unroll
          for (int j = j0; j < j0 + GRP && j < NP; ++j) {
            if (j >= PR && (j - PR) % CH == 0 && (j - PR) / CH + 1 < NCH) {  // entering an LDS chunk: request the next one
              const int c1 = (j - PR) / CH + 1;
#pragma unroll
              for (int u = 0; u < CH; ++u)
                if (c1 * CH + u < PL) {
                  buf[c1 & 1][u] = sh_pts[(c1 * CH + u) * NL + t];
                  if (WITH_Z) bufz[c1 & 1][u] = sh_ptz[(c1 * CH + u) * NL + t];
                }
            }
            const v2d v = j < PR ? reg[j < PR ? j : 0] : buf[((j - PR) / CH) & 1][(j - PR) % CH];
            const double vz = !WITH_Z ? 0.0 : (j < PR ? regz[j < PR ? j : 0] : bufz[((j - PR) / CH) & 1][(j - PR) % CH]);
            lane_point<WITH_LOSS>(q, inv_lf2, v[0], v[1], vz, M, /*renorm=*/j == j0 + GRP - 1);
          }
        } else {
#pragma unroll
          for (int j = j0; j < j0 + GRP && j < NP; ++j) {
            if (j >= PR && (j - PR) % CH == 0 && (j - PR) / CH + 1 < NCH) {
              const int c1 = (j - PR) / CH + 1;
#pragma unroll
              for (int u = 0; u < CH; ++u)
                if (c1 * CH + u < PL) {
                  buf[c1 & 1][u] = sh_pts[(c1 * CH + u) * NL + t];
                  if (WITH_Z) bufz[c1 & 1][u] = sh_ptz[(c1 * CH + u) * NL + t];
                }
            }
            const v2d v = j < PR ? reg[j < PR ? j : 0] : buf[((j - PR) / CH) & 1][(j - PR) % CH];
            const double vz = !WITH_Z ? 0.0 : (j < PR ? regz[j < PR ? j : 0] : bufz[((j - PR) / CH) & 1][(j - PR) % CH]);
            lane_point_masked<WITH_LOSS>(q, j < cnt_m, inv_lf2, v[0], v[1], vz, M, /*renorm=*/j == j0 + GRP - 1);
          }
        }
      }
    }
    double lp = 0.0;
    if (PAD_ANALYTIC) {  // the zero padding out again: npad points (0, 0) with r0 = c0 each
      const int npad = ppl_eff - cnt;
      const double np = (double)npad;
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
    } else {
      lane_pad_correction<WITH_LOSS>(M, (double)(ppl_eff - cnt_m));
    }
#ifdef CLC_STAMPS
    asm volatile("" :: "v"(M.S0), "v"(M.Tx), "v"(M.prod));
    COOP_STAMP(COOP_STAMP_PER_PASS * stamp_pass + 9);
#endif
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
    lane_flush<WITH_LOSS>(q, M, acc);
    if (PAD_ANALYTIC && WITH_LOSS) acc[27] = fma(-q.s2, lp, acc[27]);
#ifdef CLC_STAMPS
    asm volatile("" :: "v"(acc[0]), "v"(acc[20]), "v"(acc[27]));
    COOP_STAMP(COOP_STAMP_PER_PASS * stamp_pass + 10);
#endif
    {
      double u[7];
      wave_reduce_to_rows(acc, u);
#pragma unroll
      for (int i = 0; i < 7; ++i) sh_red[(wave * 7 + i) * 64 + lane] = u[i];
    }
    COOP_STAMP(COOP_STAMP_PER_PASS * stamp_pass + 1);
    __syncthreads();  // barrier A
    // ---- wave w finishes accumulators 7 w .. 7 w + 6 — their 64 partials are the 16 lanes of row w of the four waves — and publishes
    // them (pass k's tag) in this workgroup's row of board A.  Fixed order: lane l of the finishing wave takes the partial of
    // (wave l >> 4, lane 16 w + (l & 15)); then halves, 16-lane rows, inside the row. ----
    {
      const unsigned int tag = tag0 + (unsigned int)k;
      const int s0 = ((lane >> 4) * 7) * 64 + 16 * wave + (lane & 15);
      double v[8];
#pragma unroll
      for (int i = 0; i < 7; ++i) v[i] = sh_red[s0 + i * 64];
      v[7] = 0.0;
      double r[4], s[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        double x = v[i], y = v[i + 4];
        swap_halves(x, y);
        r[i] = x + y;  // rows 0, 1: element i; rows 2, 3: element i + 4
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        double x = r[i], y = r[i + 2];
        swap_rows(x, y);
        s[i] = x + y;  // row rho: element i + 2 (rho & 1) + 4 (rho >> 1)
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        double w = s[i];
        w += dpp_read<0xB1>(w);   // quad_perm [1,0,3,2]
        w += dpp_read<0x4E>(w);   // quad_perm [2,3,0,1]
        w += dpp_read<0x141>(w);  // row_half_mirror
        w += dpp_read<0x140>(w);  // row_mirror
        s[i] = w;
      }
      if ((lane & 15) == 0) {
        const int rho = lane >> 4;
        const unsigned int row = coop_row_a(k & 1, wg);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int e = i + 2 * (rho & 1) + 4 * (rho >> 1);
          if (e < 7) coop_put(rs_p, row, 7 * wave + e, s[i], tag);
        }
      }
    }
    COOP_STAMP(COOP_STAMP_PER_PASS * stamp_pass + 2);
    __syncthreads();  // barrier B: the controller wave has published the next pose, or the end
#ifdef CLC_STAMPS
    ++stamp_pass;
#endif
  }
}

}  // namespace clc

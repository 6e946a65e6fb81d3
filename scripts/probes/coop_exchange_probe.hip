// Probe: what does an all-to-all exchange of one 28-double row per workgroup cost INSIDE a kernel on gfx950, 256 workgroups x 512
// threads, one per CU (all co-resident)?  Candidate replacement for the launch boundary of the step chain (boundary 1.2 us + the
// next launch's read of 256 rows 1.7 us).
//   mode 0  tagged words: a double travels as two 8-byte words {32 payload bits | 32-bit pass tag}, written with agent-scope
//           stores; every reader polls the words it needs until the tags match.  One memory round trip, no fences, no counter.
//   mode 2  two levels: the workgroups of an XCD (blockIdx % 8, checked against HW_REG_XCC_ID) exchange through their own L2
//           (workgroup-scope = sc0 accesses: the vector cache is bypassed, the L2 is shared by the XCD's 32 CUs); the XCD's first
//           workgroup sums the 32 rows and publishes the partial row with agent-scope stores; everybody polls the 8 partial rows.
//   mode 3  three levels: as mode 2, but only the 8 leaders read the 8 partial rows over the fabric; each publishes the total to its
//           own L2, where the other 31 workgroups of the XCD poll it.
//   mode 1  rows written plainly, agent-scope release fence, atomic arrival counter, readers spin on the counter, acquire, read.
// Every round depends on the previous round's totals (like an LM pass on the controller's decision).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
constexpr int G = 256, NT = 512, NACC = 28, ROW_WORDS = 64;  // 56 used
constexpr unsigned long long TIMEOUT_TICKS = 200000ull;      // 2 ms of the 100 MHz wall clock
// Workgroup scope (sc0) is NOT enough for the intra-XCD level: measured, the readers never see the rows (the CU's vector cache is a
// legal coherence point for a workgroup, so sc0 loads may hit a stale line there for ever).  Agent scope it is.
#ifndef L2_SCOPE
#define L2_SCOPE __HIP_MEMORY_SCOPE_AGENT
#endif

__device__ __forceinline__ void put_word(unsigned long long* p, unsigned long long w) {
  __hip_atomic_store(p, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long get_word(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void put_word_l2(unsigned long long* p, unsigned long long w) {
  __hip_atomic_store(p, w, __ATOMIC_RELAXED, L2_SCOPE);
}
__device__ __forceinline__ unsigned long long get_word_l2(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, L2_SCOPE);
}
// intra-XCD alternative: plain (write-through) stores, and plain loads behind an invalidate of the CU's vector cache — the L2 is
// shared by the XCD's CUs, so this is an L2 round trip instead of a fabric one
__device__ __forceinline__ void put_word_plain(unsigned long long* p, unsigned long long w) { *reinterpret_cast<volatile unsigned long long*>(p) = w; }
__device__ __forceinline__ unsigned long long get_word_plain(const unsigned long long* p) { return *reinterpret_cast<const volatile unsigned long long*>(p); }
__device__ __forceinline__ void inv_l1() { asm volatile("buffer_inv sc0" ::: "memory"); }
// read at the L2: an agent-scope RMW executes in the XCD's L2 (the compiler emits it without sc1), so `or 0` returns what the L2 holds —
// the freshest copy when the writer sits on the same XCD — without a trip to device memory
__device__ __forceinline__ unsigned long long get_word_rmw(unsigned long long* p) {
  unsigned long long z = 0;
  asm volatile("" : "+v"(z));
  return __hip_atomic_fetch_or(p, z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long hi_word(double v, unsigned int tag) {
  return ((unsigned long long)__double_as_longlong(v) & 0xFFFFFFFF00000000ull) | tag;
}
__device__ __forceinline__ unsigned long long lo_word(double v, unsigned int tag) {
  return ((unsigned long long)__double_as_longlong(v) << 32) | tag;
}
__device__ __forceinline__ double join_words(unsigned long long w0, unsigned long long w1) {
  return __longlong_as_double((long long)((w0 & 0xFFFFFFFF00000000ull) | (w1 >> 32)));
}

// modes 2 and 3.  boardA[par][wg][64] (L2 of the XCD), boardB[par][8][64] (fabric), boardC[par][8][64] (L2 of the XCD)
template <int MODE, bool INV, bool ALL1, bool RMW = false>
__global__ __launch_bounds__(NT) void exchange2_kernel(unsigned long long* boardA, unsigned long long* boardB, unsigned long long* boardC,
                                                       int rounds, unsigned int tag0, double* out, long long* ticks, int* err, int* misplaced) {
  __shared__ double red[16][32];
  __shared__ double tot[32];
  const int tid = threadIdx.x, wg = blockIdx.x, g = tid >> 5, e = tid & 31;
  unsigned int xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 0xF;
  if (tid == 0 && (int)xcc != wg % 8) atomicAdd(misplaced, 1);
  const int x = wg % 8;
  const bool leader = wg < 8;
  if (tid < 32) tot[tid] = 0.0;
  __syncthreads();
  const unsigned long long t_begin = wall_clock64();
  for (int k = 0; k < rounds; ++k) {
    const unsigned int tag = tag0 + (unsigned int)k;
    const int par = k & 1;
    if (tid < NACC) {
      const double v = (double)(wg + 1) * (double)(tid + 1) + 1e-3 * k + 1e-9 * tot[tid];
      unsigned long long* row = boardA + ((size_t)par * G + wg) * ROW_WORDS;
      if (INV) { put_word_plain(row + 2 * tid, hi_word(v, tag)); put_word_plain(row + 2 * tid + 1, lo_word(v, tag)); }
      else { put_word_l2(row + 2 * tid, hi_word(v, tag)); put_word_l2(row + 2 * tid + 1, lo_word(v, tag)); }
    }
    if (leader || ALL1) {
      // rows of this XCD: wg' = x + 8 s; thread (g, e) takes s = g and g + 16
      double s = 0.0;
      if (e < NACC) {
        unsigned long long* r0 = boardA + ((size_t)par * G + x + 8 * g) * ROW_WORDS + 2 * e;
        unsigned long long* r1 = boardA + ((size_t)par * G + x + 8 * (g + 16)) * ROW_WORDS + 2 * e;
        int tries = 0;
        unsigned long long a0, a1, b0, b1;
        const unsigned long long t0 = wall_clock64();
        bool ok;
        do {
          if (RMW && tries++ < 64) { a0 = get_word_rmw(r0); a1 = get_word_rmw(r0 + 1); b0 = get_word_rmw(r1); b1 = get_word_rmw(r1 + 1); }
          else if (INV) { inv_l1(); a0 = get_word_plain(r0); a1 = get_word_plain(r0 + 1); b0 = get_word_plain(r1); b1 = get_word_plain(r1 + 1); }
          else { a0 = get_word_l2(r0); a1 = get_word_l2(r0 + 1); b0 = get_word_l2(r1); b1 = get_word_l2(r1 + 1); }
          ok = (unsigned int)a0 == tag && (unsigned int)a1 == tag && (unsigned int)b0 == tag && (unsigned int)b1 == tag;
          if (!ok && wall_clock64() - t0 > TIMEOUT_TICKS) { *err = 1; break; }
        } while (!ok);
        s = join_words(a0, a1) + join_words(b0, b1);
      }
      red[g][e] = s;
      __syncthreads();
      if (tid < NACC) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += red[i][tid];
        if (leader) {
          unsigned long long* row = boardB + ((size_t)par * 8 + x) * ROW_WORDS;
          put_word(row + 2 * tid, hi_word(t, tag));
          put_word(row + 2 * tid + 1, lo_word(t, tag));
        }
      }
      __syncthreads();
    }
    if (MODE == 2 || leader) {
      // the 8 partial rows over the fabric: thread (xx = tid >> 5, e) for tid < 256
      double v = 0.0;
      if (tid < 256 && e < NACC) {
        const unsigned long long* r = boardB + ((size_t)par * 8 + g) * ROW_WORDS + 2 * e;
        unsigned long long a0, a1;
        const unsigned long long t0 = wall_clock64();
        bool ok;
        do {
          a0 = get_word(r); a1 = get_word(r + 1);
          ok = (unsigned int)a0 == tag && (unsigned int)a1 == tag;
          if (!ok) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > TIMEOUT_TICKS) { *err = 1; break; }
          }
        } while (!ok);
        v = join_words(a0, a1);
      }
      if (tid < 256) red[g][e] = v;
      __syncthreads();
      if (tid < NACC) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) t += red[i][tid];
        tot[tid] = t;
        if (MODE == 3) {
          unsigned long long* row = boardC + ((size_t)par * 8 + x) * ROW_WORDS;
          if (INV) { put_word_plain(row + 2 * tid, hi_word(t, tag)); put_word_plain(row + 2 * tid + 1, lo_word(t, tag)); }
          else { put_word_l2(row + 2 * tid, hi_word(t, tag)); put_word_l2(row + 2 * tid + 1, lo_word(t, tag)); }
        }
      }
      __syncthreads();
    } else {
      if (tid < NACC) {
        const unsigned long long* r = boardC + ((size_t)par * 8 + x) * ROW_WORDS + 2 * tid;
        unsigned long long a0, a1;
        const unsigned long long t0 = wall_clock64();
        bool ok;
        do {
          if (INV) { inv_l1(); a0 = get_word_plain(r); a1 = get_word_plain(r + 1); }
          else { a0 = get_word_l2(r); a1 = get_word_l2(r + 1); }
          ok = (unsigned int)a0 == tag && (unsigned int)a1 == tag;
          if (!ok) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > TIMEOUT_TICKS) { *err = 1; break; }
          }
        } while (!ok);
        tot[tid] = join_words(a0, a1);
      }
      __syncthreads();
    }
  }
  if (tid == 0) ticks[wg] = (long long)(wall_clock64() - t_begin);
  if (wg == 255 && tid < NACC) out[tid] = tot[tid];
}

template <int MODE>
__global__ __launch_bounds__(NT) void exchange_kernel(unsigned long long* board, double* plain, unsigned int* counter, int rounds,
                                                      unsigned int tag0, double* out, long long* ticks, int* err) {
  __shared__ double red[16][32];
  __shared__ double tot[32];
  const int tid = threadIdx.x, wg = blockIdx.x, g = tid >> 5, e = tid & 31;
  if (tid < 32) tot[tid] = 0.0;
  __syncthreads();
  const unsigned long long t_begin = wall_clock64();
  for (int k = 0; k < rounds; ++k) {
    const unsigned int tag = tag0 + (unsigned int)k;
    const int par = k & 1;
    // ---- this workgroup's row (depends on the previous totals) ----
    if (tid < NACC) {
      const double v = (double)(wg + 1) * (double)(tid + 1) + 1e-3 * k + 1e-9 * tot[tid];
      if (MODE == 0) {
        const unsigned long long b = (unsigned long long)__double_as_longlong(v);
        unsigned long long* row = board + ((size_t)par * G + wg) * ROW_WORDS;
        put_word(row + 2 * tid, (b & 0xFFFFFFFF00000000ull) | tag);
        put_word(row + 2 * tid + 1, (b << 32) | tag);
      } else {
        plain[((size_t)par * G + wg) * 32 + tid] = v;
      }
    }
    if (MODE == 1) {
      if (tid < 64) {  // wave 0 wrote the row
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (tid == 0) {
          __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const unsigned int want = (unsigned int)G * (unsigned int)(k + 1);
          const unsigned long long t0 = wall_clock64();
          while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > TIMEOUT_TICKS) { *err = 1; break; }
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      __syncthreads();
    }
    // ---- gather: thread (g, e) sums element e of rows g, g + 16, ... in that order ----
    double s = 0.0;
    if (e < NACC) {
      if (MODE == 0) {
        unsigned long long w0[16], w1[16];
        const unsigned long long* base = board + ((size_t)par * G + g) * ROW_WORDS + 2 * e;
        const unsigned long long t0 = wall_clock64();
        bool ok;
        do {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            w0[i] = get_word(base + (size_t)(16 * i) * ROW_WORDS);
            w1[i] = get_word(base + (size_t)(16 * i) * ROW_WORDS + 1);
          }
          ok = true;
#pragma unroll
          for (int i = 0; i < 16; ++i) ok = ok && (unsigned int)w0[i] == tag && (unsigned int)w1[i] == tag;
          if (!ok) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > TIMEOUT_TICKS) { *err = 1; break; }
          }
        } while (!ok);
#pragma unroll
        for (int i = 0; i < 16; ++i) s += __longlong_as_double((long long)((w0[i] & 0xFFFFFFFF00000000ull) | (w1[i] >> 32)));
      } else {
        const double* base = plain + ((size_t)par * G + g) * 32 + e;
        double v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = __hip_atomic_load(base + (size_t)(16 * i) * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int i = 0; i < 16; ++i) s += v[i];
      }
    }
    red[g][e] = s;
    __syncthreads();
    if (tid < NACC) {
      double t = 0.0;
#pragma unroll
      for (int i = 0; i < 16; ++i) t += red[i][tid];
      tot[tid] = t;
    }
    __syncthreads();
  }
  if (tid == 0) ticks[wg] = (long long)(wall_clock64() - t_begin);
  if (wg == 0 && tid < NACC) out[tid] = tot[tid];
}

int main() {
  unsigned long long* board; double* plain; unsigned int* counter; double* out; long long* ticks; int* err;
  (void)hipMalloc(&board, sizeof(unsigned long long) * 2 * G * ROW_WORDS);
  (void)hipMalloc(&plain, sizeof(double) * 2 * G * 32);
  (void)hipMalloc(&counter, 4); (void)hipMalloc(&out, 32 * 8); (void)hipMalloc(&ticks, G * 8); (void)hipMalloc(&err, 4);
  (void)hipMemset(board, 0, sizeof(unsigned long long) * 2 * G * ROW_WORDS);
  (void)hipMemset(err, 0, 4);
  int dev_cus = 0; (void)hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, 0);
  int occ0 = 0; (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ0, exchange_kernel<0>, NT, 0);
  printf("CUs %d, workgroups of exchange_kernel<0> per CU %d\n", dev_cus, occ0);
  if (dev_cus * occ0 < G) { printf("not all %d workgroups can be resident\n", G); return 1; }
  const int rounds = 100;
  // host model of the recurrence (same summation order)
  std::vector<double> tot(NACC, 0.0);
  for (int k = 0; k < rounds; ++k) {
    std::vector<double> nt(NACC);
    for (int e = 0; e < NACC; ++e) {
      double red[16];
      for (int g = 0; g < 16; ++g) { double s = 0.0; for (int i = 0; i < 16; ++i) s += (double)(g + 16 * i + 1) * (double)(e + 1) + 1e-3 * k + 1e-9 * tot[e]; red[g] = s; }
      double t = 0.0; for (int g = 0; g < 16; ++g) t += red[g];
      nt[e] = t;
    }
    tot = nt;
  }
  // host model of the two-level order: per XCD (s = g, g + 16 pairs, then the 16 groups), then the 8 XCDs
  std::vector<double> tot2(NACC, 0.0);
  for (int k = 0; k < rounds; ++k) {
    std::vector<double> nt(NACC);
    for (int e = 0; e < NACC; ++e) {
      double part[8];
      for (int x = 0; x < 8; ++x) {
        double t = 0.0;
        for (int g = 0; g < 16; ++g) {
          const double a = (double)(x + 8 * g + 1) * (double)(e + 1) + 1e-3 * k + 1e-9 * tot2[e];
          const double b = (double)(x + 8 * (g + 16) + 1) * (double)(e + 1) + 1e-3 * k + 1e-9 * tot2[e];
          t += a + b;
        }
        part[x] = t;
      }
      double t = 0.0; for (int x = 0; x < 8; ++x) t += part[x];
      nt[e] = t;
    }
    tot2 = nt;
  }
  unsigned long long *boardA, *boardB, *boardC; int* misplaced;
  (void)hipMalloc(&boardA, sizeof(unsigned long long) * 2 * G * ROW_WORDS); (void)hipMemset(boardA, 0, sizeof(unsigned long long) * 2 * G * ROW_WORDS);
  (void)hipMalloc(&boardB, sizeof(unsigned long long) * 2 * 8 * ROW_WORDS); (void)hipMemset(boardB, 0, sizeof(unsigned long long) * 2 * 8 * ROW_WORDS);
  (void)hipMalloc(&boardC, sizeof(unsigned long long) * 2 * 8 * ROW_WORDS); (void)hipMemset(boardC, 0, sizeof(unsigned long long) * 2 * 8 * ROW_WORDS);
  (void)hipMalloc(&misplaced, 4); (void)hipMemset(misplaced, 0, 4);
  unsigned int tag = 1;
  for (int mode = 2; mode < 9; ++mode)
    for (int rep = 0; rep < 3; ++rep) {
      hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
      (void)hipEventRecord(a, 0);
#define LAUNCH2(M, I, A) hipLaunchKernelGGL((exchange2_kernel<M, I, A>), dim3(G), dim3(NT), 0, 0, boardA, boardB, boardC, rounds, tag, out, ticks, err, misplaced)
      // 2: two levels, agent scope; 3: three levels, agent scope; 4: two levels, level 1 through the L2 (leader only); 5: three levels,
      // levels 1 and 3 through the L2; 6: two levels, level 1 through the L2 by EVERY workgroup of the XCD; 7: same with agent scope
      if (mode == 2) LAUNCH2(2, false, false);
      else if (mode == 3) LAUNCH2(3, false, false);
      else if (mode == 4) LAUNCH2(2, true, false);
      else if (mode == 5) LAUNCH2(3, true, false);
      else if (mode == 6) LAUNCH2(2, true, true);
      else if (mode == 7) LAUNCH2(2, false, true);
      else hipLaunchKernelGGL((exchange2_kernel<2, false, false, true>), dim3(G), dim3(NT), 0, 0, boardA, boardB, boardC, rounds, tag, out, ticks, err, misplaced);  // 8: level 1 read with RMWs at the L2
      (void)hipEventRecord(b, 0);
      (void)hipEventSynchronize(b);
      tag += rounds;
      float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
      double h[32]; long long ht[G]; int herr = 0, hmis = 0;
      (void)hipMemcpy(h, out, 28 * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(ht, ticks, G * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
      (void)hipMemcpy(&hmis, misplaced, 4, hipMemcpyDeviceToHost);
      double worst = 0; for (int e = 0; e < NACC; ++e) worst = fmax(worst, fabs(h[e] - tot2[e]) / fabs(tot2[e]));
      long long tmax = 0; for (int w = 0; w < G; ++w) tmax = ht[w] > tmax ? ht[w] : tmax;
      printf("mode %d rep %d: %.3f us per round (events), %.3f us per round (slowest workgroup's clock), max rel err %.2e, timeout flag %d, workgroups not on XCD blockIdx %% 8: %d\n",
             mode, rep, 1e3 * ms / rounds, tmax / 100.0 / rounds, worst, herr, hmis);
    }
  for (int mode = 0; mode < 2; ++mode)
    for (int rep = 0; rep < 4; ++rep) {
      (void)hipMemset(counter, 0, 4);
      hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
      (void)hipEventRecord(a, 0);
      if (mode == 0) hipLaunchKernelGGL(exchange_kernel<0>, dim3(G), dim3(NT), 0, 0, board, plain, counter, rounds, tag, out, ticks, err);
      else hipLaunchKernelGGL(exchange_kernel<1>, dim3(G), dim3(NT), 0, 0, board, plain, counter, rounds, tag, out, ticks, err);
      (void)hipEventRecord(b, 0);
      (void)hipEventSynchronize(b);
      tag += rounds;
      float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
      double h[32]; long long ht[G]; int herr = 0;
      (void)hipMemcpy(h, out, 28 * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(ht, ticks, G * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
      double worst = 0; for (int e = 0; e < NACC; ++e) worst = fmax(worst, fabs(h[e] - tot[e]) / fabs(tot[e]));
      long long tmax = 0; for (int w = 0; w < G; ++w) tmax = ht[w] > tmax ? ht[w] : tmax;
      printf("mode %d rep %d: %.3f us per round (events), %.3f us per round (slowest workgroup's clock), max rel err %.2e, timeout flag %d\n",
             mode, rep, 1e3 * ms / rounds, tmax / 100.0 / rounds, worst, herr);
    }
  return 0;
}

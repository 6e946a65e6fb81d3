// abi_layouts.hip — the upload paths: staged records -> 64-byte tiles, scan structure, compact / row / lane layouts; stored scans.
// (one of the translation units of the C-ABI; see clc_abi_internal.hpp)
#include "clc_abi_internal.hpp"

using namespace clc_abi;

namespace {

int retile_into(clc_handle* h, const double* d_aos, size_t n, double** d_tiles, size_t* cap_bytes) {
  const size_t n_padded = ((n + clc::TILE - 1) / clc::TILE) * clc::TILE;
  const size_t bytes = std::max<size_t>(n_padded, clc::TILE) * 8 * sizeof(double);
  if (bytes > *cap_bytes) {
    if (*d_tiles) CLC_HIP(hipFree(*d_tiles));
    *d_tiles = nullptr;
    *cap_bytes = 0;
    CLC_HIP(hipMalloc(d_tiles, bytes));
    *cap_bytes = bytes;
  }
  if (n_padded > 0) {
    const int threads = 256;
    const long long blocks = ((long long)n_padded + threads - 1) / threads;
    hipLaunchKernelGGL(clc::retile_kernel, dim3((unsigned)blocks), dim3(threads), 0, h->stream, d_aos,
                       *d_tiles, (long long)n, (long long)n_padded);
    CLC_HIP(hipGetLastError());
  }
  return CLC_OK;
}

// Upload-time re-encoding of staged AoS records, entirely on the device (O(1) host work, a few words copied back):
//   * scans = runs of records with bit-identical (n, d, scale) (and never across two problems): flags, prefix sum ->
//     scan index per record, scan starts;
//   * compact layout (clc_kernels.hpp "Compact layout"): group table + 28-byte tiles, bitwise lossless;
//   * row layout (clc_rows.hpp): every scan padded to whole rows of 64 points, (x, y) rows + one descriptor per row —
//     only when every record has p.z == 0 and the padding at most doubles the array.
// Nothing is kept when the records do not group at least 4:1 (hand-made arrays without scan structure).
template <class TIn>
int device_scan(clc_handle* h, const TIn* d_in, long long n, unsigned int minus_one, unsigned int* d_out,
                unsigned long long* d_totals /* [blocks + 1] */) {
  const long long blocks = (n + clc::SCAN_CHUNK - 1) / clc::SCAN_CHUNK;
  hipLaunchKernelGGL((clc::scan_block_totals_kernel<TIn>), dim3((unsigned)blocks), dim3(clc::SCAN_THREADS), 0, h->stream,
                     d_in, n, d_totals);
  hipLaunchKernelGGL(clc::scan_totals_kernel, dim3(1), dim3(clc::SCAN_THREADS), 0, h->stream, d_totals, blocks);
  hipLaunchKernelGGL((clc::scan_apply_kernel<TIn>), dim3((unsigned)blocks), dim3(clc::SCAN_THREADS), 0, h->stream, d_in, n,
                     d_totals, minus_one, d_out);
  CLC_HIP(hipGetLastError());
  return CLC_OK;
}


// The lane layout of the batched problems (clc_resident.hpp) from the staged records and their scan structure: plan
// (points per lane of every problem, on the device), offsets (O(P) on the host), lane descriptors + j-major point rows.
// Leaves L.ok false — and the streaming layouts in charge — when some problem does not fit a workgroup.
int build_resident(clc_handle* h, ResLayout& L, int first_try, const double* d_aos, long long n, size_t P, size_t G,
                   const long long* d_rec_off, const unsigned int* d_gid, const long long* d_starts, int max_ppl_override = 0,
                   bool with_z = false) {
  L.ok = false;
  L.with_z = false;
  L.lanes = 0;
  L.max_ppl = 0;
  L.rows = 0;
  if ((h->launch_flags & clc::FLAG_NO_RESIDENT) != 0) return CLC_OK;
  const int threads = 256;
  DevBuf<unsigned int> bppl(&h->pool), bfail(&h->pool);
  CLC_HIP(bppl.alloc(P));
  CLC_HIP(bfail.alloc(1));
  std::vector<unsigned int> ppl(P);
  int lanes = 0;
  for (int nl = first_try; nl <= (max_ppl_override > 0 ? first_try : 512) && lanes == 0; nl *= 2) {
    unsigned int failed = 0;
    CLC_HIP(hipMemsetAsync(bfail.p, 0, sizeof(unsigned int), h->stream));
    hipLaunchKernelGGL(clc::res_plan_kernel, dim3((unsigned)((P + threads - 1) / threads)), dim3(threads), 0, h->stream, d_rec_off,
                       d_gid, d_starts, (long long)P, n, (long long)G, nl, max_ppl_override > 0 ? max_ppl_override : (nl == 256 ? kResPR256 + kResPL256 : kResPR512 + kResPL512), bppl.p, bfail.p);
    CLC_HIP(hipGetLastError());
    CLC_HIP(hipMemcpyAsync(&failed, bfail.p, sizeof(unsigned int), hipMemcpyDeviceToHost, h->stream));
    CLC_HIP(hipMemcpyAsync(ppl.data(), bppl.p, sizeof(unsigned int) * P, hipMemcpyDeviceToHost, h->stream));
    CLC_HIP(hipStreamSynchronize(h->stream));
    if (!failed) lanes = nl;
  }
  if (lanes == 0) return CLC_OK;  // some problem is too long (or has too many scans) for one workgroup
  std::vector<unsigned int> row(P + 1, 0u);
  unsigned long long total = 0;
  unsigned int max_ppl = 0;
  bool uniform = true;
  for (size_t k = 0; k < P; ++k) {
    row[k] = (unsigned int)total;
    total += ppl[k];
    max_ppl = std::max(max_ppl, ppl[k]);
    uniform = uniform && ppl[k] == ppl[0];
  }
  if (total >= 0xFFFFFFF0ull) return CLC_OK;
  row[P] = (unsigned int)total;
  int rc = ensure_bytes(&L.d_row, &L.row_cap, (P + 1) * sizeof(unsigned int));
  if (rc != CLC_OK) return rc;
  rc = ensure_bytes(&L.d_desc, &L.desc_cap, P * (size_t)lanes * sizeof(clc::ResLane));
  if (rc != CLC_OK) return rc;
  // one padding row: the kernel's loads run unconditionally from clamped row indices (an empty last problem reads it)
  rc = ensure_bytes(&L.d_xy, &L.xy_cap, ((size_t)total + 1) * (size_t)lanes * 2 * sizeof(double));
  if (rc != CLC_OK) return rc;
  CLC_HIP(hipMemcpyAsync(L.d_row, row.data(), (P + 1) * sizeof(unsigned int), hipMemcpyHostToDevice, h->stream));
  CLC_HIP(hipMemsetAsync(L.d_xy + (size_t)total * (size_t)lanes * 2, 0, (size_t)lanes * 2 * sizeof(double), h->stream));
  if (with_z) {
    rc = ensure_bytes(&L.d_z, &L.z_cap, ((size_t)total + 1) * (size_t)lanes * sizeof(double));
    if (rc != CLC_OK) return rc;
    CLC_HIP(hipMemsetAsync(L.d_z + (size_t)total * (size_t)lanes, 0, (size_t)lanes * sizeof(double), h->stream));
  }
  double* d_zl = with_z ? L.d_z : nullptr;
  const unsigned int* d_row = reinterpret_cast<const unsigned int*>(L.d_row);
  clc::ResLane* d_desc = reinterpret_cast<clc::ResLane*>(L.d_desc);
  if (lanes == 256)
    hipLaunchKernelGGL((clc::res_build_kernel<256>), dim3((unsigned)P), dim3(256), 0, h->stream, d_aos, d_rec_off, d_gid, d_starts, n,
                       (long long)G, d_row, d_desc, L.d_xy, d_zl);
  else
    hipLaunchKernelGGL((clc::res_build_kernel<512>), dim3((unsigned)P), dim3(512), 0, h->stream, d_aos, d_rec_off, d_gid, d_starts, n,
                       (long long)G, d_row, d_desc, L.d_xy, d_zl);
  CLC_HIP(hipGetLastError());
  CLC_HIP(hipStreamSynchronize(h->stream));  // `row` is a host temporary
  L.lanes = lanes;
  L.max_ppl = (int)max_ppl;
  L.uni_ppl = uniform && P > 0 ? (int)ppl[0] : -1;
  L.rows = (long long)total;
  L.with_z = with_z;
  L.ok = true;
  return CLC_OK;
}

}  // namespace

namespace clc_abi {

void warm_layouts() {
  warm_kernel(reinterpret_cast<const void*>(&clc::retile_kernel));
  warm_kernel(reinterpret_cast<const void*>(&clc::scan_flag_kernel));
  warm_kernel(reinterpret_cast<const void*>(&clc::build_rows_kernel));
  warm_kernel(reinterpret_cast<const void*>(&clc::res_plan_kernel));
  warm_kernel(reinterpret_cast<const void*>(&clc::res_build_kernel<512>));
  warm_kernel(reinterpret_cast<const void*>(&clc::flatten_kernel));
}

int build_layouts(clc_handle* h, const double* d_aos, size_t n_total, const std::vector<long long>& rec_off,
                  const std::vector<long long>& tile_off, const LayoutTargets& T) {
  *T.compact_ok = false;
  *T.rows_ok = false;
  *T.n_groups = 0;
  *T.n_rows = 0;
  const size_t P = rec_off.size() - 1;
  if (n_total == 0 || P == 0) return CLC_OK;
  if (n_total >= 0xFFFFFFF0ull) return CLC_OK;  // scan indices are 32-bit; such arrays keep the 64-byte tiles
  const long long n = (long long)n_total;
  const int threads = 256;
  const long long scan_blocks = (n + clc::SCAN_CHUNK - 1) / clc::SCAN_CHUNK;
  DevBuf<unsigned char> bflag(&h->pool);
  DevBuf<unsigned int> bgid(&h->pool), bzflag(&h->pool);
  DevBuf<unsigned long long> btotals(&h->pool);
  DevBuf<long long> broff(&h->pool), btoff(&h->pool);
  CLC_HIP(bflag.alloc(n_total));
  CLC_HIP(bgid.alloc(n_total));
  CLC_HIP(bzflag.alloc(1));
  CLC_HIP(btotals.alloc((size_t)scan_blocks + 1));
  CLC_HIP(broff.alloc(P + 1));
  CLC_HIP(btoff.alloc(P + 1));
  CLC_HIP(hipMemcpyAsync(broff.p, rec_off.data(), (P + 1) * sizeof(long long), hipMemcpyHostToDevice, h->stream));
  CLC_HIP(hipMemcpyAsync(btoff.p, tile_off.data(), (P + 1) * sizeof(long long), hipMemcpyHostToDevice, h->stream));
  CLC_HIP(hipMemsetAsync(bzflag.p, 0, sizeof(unsigned int), h->stream));
  hipLaunchKernelGGL(clc::scan_flag_kernel, dim3((unsigned)((n + threads - 1) / threads)), dim3(threads), 0, h->stream, d_aos,
                     n, bflag.p, bzflag.p);
  hipLaunchKernelGGL(clc::mark_problem_starts_kernel, dim3((unsigned)((P + threads - 1) / threads)), dim3(threads), 0,
                     h->stream, broff.p, (long long)P, n, bflag.p);
  CLC_HIP(hipGetLastError());
  int rc = device_scan<unsigned char>(h, bflag.p, n, 1u, bgid.p, btotals.p);
  if (rc != CLC_OK) return rc;
  unsigned int last_gid = 0, any_z = 0;
  CLC_HIP(hipMemcpyAsync(&last_gid, bgid.p + (n - 1), sizeof(unsigned int), hipMemcpyDeviceToHost, h->stream));
  CLC_HIP(hipMemcpyAsync(&any_z, bzflag.p, sizeof(unsigned int), hipMemcpyDeviceToHost, h->stream));
  CLC_HIP(hipStreamSynchronize(h->stream));
  const size_t G = (size_t)last_gid + 1;
  // fewer than 4 points per scan on average: the streaming layouts do not pay (the step chain keeps the 64-byte tiles) — the on-chip
  // layouts, where a lane carries its own plane anyway, are still built
  const bool sparse = G * 4 > n_total;
  if (sparse && T.res == nullptr && T.coop == nullptr) return CLC_OK;
  DevBuf<long long> bstarts(&h->pool);
  CLC_HIP(bstarts.alloc(G + 1));
  hipLaunchKernelGGL(clc::scan_starts_kernel, dim3((unsigned)((n + threads - 1) / threads)), dim3(threads), 0, h->stream,
                     bflag.p, bgid.p, n, (long long)G, bstarts.p);
  CLC_HIP(hipGetLastError());
  // ---- compact layout ----
  const size_t total_tiles = (size_t)tile_off[P];
  rc = ensure_bytes(T.d_ct, T.ct_cap, std::max<size_t>(total_tiles, 1) * clc::CTILE_DOUBLES * sizeof(double));
  if (rc != CLC_OK) return rc;
  rc = ensure_bytes(T.d_gr, T.gr_cap, G * clc::GROUP_DOUBLES * sizeof(double));
  if (rc != CLC_OK) return rc;
  hipLaunchKernelGGL(clc::build_groups_dev_kernel, dim3((unsigned)((G + threads - 1) / threads)), dim3(threads), 0, h->stream,
                     d_aos, bstarts.p, (long long)G, *T.d_gr);
  {
    long long max_padded = 0;
    for (size_t k = 0; k < P; ++k) max_padded = std::max(max_padded, (tile_off[k + 1] - tile_off[k]) * clc::TILE);
    const unsigned ydim = (unsigned)std::min<long long>(4096, std::max<long long>(1, (max_padded + threads - 1) / threads));
    hipLaunchKernelGGL(clc::build_ctiles_kernel, dim3((unsigned)P, ydim), dim3(threads), 0, h->stream, d_aos, bgid.p, broff.p,
                       btoff.p, *T.d_ct);
  }
  CLC_HIP(hipGetLastError());
  // ---- row layout ----
  bool rows_ok = false;
  long long R = 0;
  DevBuf<unsigned int> brows(&h->pool), brbeg(&h->pool);
  DevBuf<unsigned long long> btot2(&h->pool);
  const size_t row_doubles = any_z ? clc::ROW_DOUBLES_Z : clc::ROW_DOUBLES;
  if (T.rows_z) *T.rows_z = any_z != 0;
  {
    const long long gblocks = ((long long)G + clc::SCAN_CHUNK - 1) / clc::SCAN_CHUNK;
    CLC_HIP(brows.alloc(G));
    CLC_HIP(brbeg.alloc(G + 1));
    CLC_HIP(btot2.alloc((size_t)gblocks + 1));
    hipLaunchKernelGGL(clc::scan_rows_kernel, dim3((unsigned)((G + threads - 1) / threads)), dim3(threads), 0, h->stream,
                       bstarts.p, (long long)G, brows.p);
    CLC_HIP(hipMemsetAsync(brbeg.p, 0, sizeof(unsigned int), h->stream));
    rc = device_scan<unsigned int>(h, brows.p, (long long)G, 0u, brbeg.p + 1, btot2.p);
    if (rc != CLC_OK) return rc;
    unsigned int total_rows = 0;
    CLC_HIP(hipMemcpyAsync(&total_rows, brbeg.p + G, sizeof(unsigned int), hipMemcpyDeviceToHost, h->stream));
    CLC_HIP(hipStreamSynchronize(h->stream));
    R = (long long)total_rows;
    rows_ok = !sparse && R > 0 && (size_t)R * clc::ROW <= 3 * n_total + 64 * P;  // rows at least a third full on average
  }
  if (rows_ok) {
    // one padding row each: the streaming loop's prologue loads run unconditionally from clamped row indices
    rc = ensure_bytes(T.d_rxy, T.rxy_cap, ((size_t)R + 1) * row_doubles * sizeof(double));
    if (rc != CLC_OK) return rc;
    rc = ensure_bytes(T.d_rdesc, T.rdesc_cap, ((size_t)R + 1) * sizeof(clc::RowDesc) + clc::wave_split_bytes(R));  // + the wave split table
    if (rc != CLC_OK) return rc;
    CLC_HIP(hipMemsetAsync(*T.d_rxy + (size_t)R * row_doubles, 0, row_doubles * sizeof(double), h->stream));
    CLC_HIP(hipMemsetAsync(reinterpret_cast<char*>(*T.d_rdesc) + (size_t)R * sizeof(clc::RowDesc), 0, sizeof(clc::RowDesc), h->stream));
    const long long slots = R * clc::ROW;
    hipLaunchKernelGGL(clc::build_rows_kernel, dim3((unsigned)((slots + threads - 1) / threads)), dim3(threads), 0, h->stream,
                       d_aos, bstarts.p, brbeg.p, (long long)G, R, (int)row_doubles, *T.d_rxy, reinterpret_cast<clc::RowDesc*>(*T.d_rdesc));
    if (T.d_prob_row)
      hipLaunchKernelGGL(clc::problem_rows_kernel, dim3((unsigned)((P + 1 + threads - 1) / threads)), dim3(threads), 0,
                         h->stream, broff.p, bgid.p, brbeg.p, (long long)P, n, R, *T.d_prob_row);
    CLC_HIP(hipGetLastError());
  }
  if (T.res != nullptr) {
    T.res->ok = false;
    if (!any_z) {
      // batches: 256 lanes (two problems per CU) unless flag 8192; a single problem: 512 lanes (it has its CU to itself)
      const int first_try = (T.d_prob_row == nullptr || (h->launch_flags & clc::FLAG_RESIDENT_WG512) != 0) ? 512 : 256;
      rc = build_resident(h, *T.res, first_try, d_aos, n, P, G, broff.p, bgid.p, bstarts.p);
      if (rc != CLC_OK) return rc;
    } else if (T.d_prob_row != nullptr) {
      // a BATCH whose points carry z: the 512-lane form with 24-byte slots (one problem per CU; resident_solve_kernel<.., WITH_Z>) — the
      // batch is still read from HBM once per solve.  (A single problem with z: the cooperative kernel's z form, below.)
      rc = build_resident(h, *T.res, 512, d_aos, n, P, G, broff.p, bgid.p, bstarts.p, kResPRz + kResPLz, /*with_z=*/true);
      if (rc != CLC_OK) return rc;
    }
  }
  if (T.coop != nullptr) {
    T.coop->ok = false;
    // (records with p.z != 0: the cooperative kernel's WITH_Z form holds 24-byte slots)
    // (clc_set_small_on_coop: the cooperative layout also for a problem one workgroup holds — clc_solve then prefers it)
    const bool small_on_coop = h->small_on_coop;
    if (P == 1 && n > 0 && h->num_cus >= clc::COOP_WGS && (small_on_coop || !(T.res != nullptr && T.res->ok))) {
      // the one problem in chunks of equal record counts, one per workgroup (a chunk may begin and end inside a scan: res_scan_extent):
      // COOP_WGS of them, or COOP_SMALL_WGS where that leaves a lane at most kCoopSmallMaxPpl points (the one-hop form of the kernel)
      const int cap_ppl = any_z ? clc::COOP_PR_Z + clc::COOP_PL_Z : clc::COOP_PR + clc::COOP_PL;
      // Up to this many points per lane of 32 workgroups the one-hop form runs (a host-side choice between two launch forms of the same
      // kernel).  Round 5 sweep (scripts/r05_small_form.py, us per pass, 32 / 256 workgroups): 1e5 observations 5.00 / 5.61, 1.3e5
      // 5.61 / 5.54, 1.6e5 5.55 / 5.51, 2e5 5.94 / 5.53 — a lane's extra points cost 0.065 us each, the second hop ~1 us: 13 points
      // per lane (106 496 observations) is where the one-hop form stops paying (round 4: 10).
      constexpr int kCoopSmallMaxPpl = 13;
      int small_cap = kCoopSmallMaxPpl;
#ifdef CLC_TEST_HOOKS
      if (const char* e = std::getenv("CLC_COOP_SMALL_MAX_PPL")) small_cap = std::max(1, std::atoi(e));  // (tuning hook, hooks build only: scripts/r05_small_form.py)
#endif
      const int small_ppl = std::min(small_cap, cap_ppl);
      const bool small_ok = n <= (long long)clc::COOP_SMALL_WGS * clc::COOP_NL * small_ppl && (h->auto_disable & 8) == 0;
      for (int attempt = small_ok ? 0 : 1; attempt < 2 && !T.coop->ok; ++attempt) {
        const int wgs = attempt == 0 ? clc::COOP_SMALL_WGS : clc::COOP_WGS;
        std::vector<long long> chunk((size_t)wgs + 1);
        for (int c = 0; c <= wgs; ++c) chunk[c] = (long long)((__int128)n * c / wgs);
        DevBuf<long long> bchunk(&h->pool);
        CLC_HIP(bchunk.alloc(chunk.size()));
        CLC_HIP(hipMemcpyAsync(bchunk.p, chunk.data(), chunk.size() * sizeof(long long), hipMemcpyHostToDevice, h->stream));
        CLC_HIP(hipStreamSynchronize(h->stream));
        rc = build_resident(h, *T.coop, clc::COOP_NL, d_aos, n, (size_t)wgs, G, bchunk.p, bgid.p, bstarts.p,
                            attempt == 0 ? small_ppl : cap_ppl, any_z != 0);
        if (rc != CLC_OK) return rc;
        T.coop->wgs = wgs;
      }  // (the small form's layout did not fit — scans that leave half-filled lanes: the 256-workgroup form is tried next)
    }
  }
  CLC_HIP(hipStreamSynchronize(h->stream));  // the temporaries above are freed on return
  *T.n_groups = (long long)G;
  *T.compact_ok = !sparse;
  *T.n_rows = R;
  *T.rows_ok = rows_ok;
  return CLC_OK;
}


// ---------------------------------------------------------------------------------------------------------------------------
// Small single problems — the reference's own sizes (main/calibr_simulation.cpp: 50 poses x ~114 points; main/calibr_offline.cpp:
// O(10^2) poses): at most 512 x 22 records, p.z == 0, so that ONE workgroup holds the problem.  The generic pipeline above finds the
// scan structure on the device and reads it back three times (scan count, row count, points per lane) plus two synchronisations for
// host temporaries: ~0.2 ms of waiting for ~20 us of kernels.  Here the HOST knows the structure — from the stored scans' offsets and tag
// poses (clc_select_observations) or from the records themselves (clc_upload) —, plans the layouts (scan starts, rows per scan, points
// per lane: the arithmetic of scan_rows_kernel / res_plan_kernel) and enqueues ONE copy of the tables + the same build kernels, without
// a single read-back or synchronisation.  Same kernels, same tables: the layouts are bit for bit the generic pipeline's
// (tests/test_gpu_edge_cases.py::test_small_problems_planned_on_the_host_...).
struct SmallFlatten {  // clc_select_observations: the records are built on the device first (flatten_kernel)
  int n_poses;
  bool linefit, boundary;
  const std::vector<long long>* rec_off;  // [n_poses + 1]
};

constexpr size_t kSmallMaxRecords = (size_t)512 * (kResPR512 + kResPL512);

static size_t align8(size_t b) { return (b + 7) & ~(size_t)7; }

// *used = false: not a case for this path (nothing was enqueued) — the caller runs the generic pipeline.
int small_fast_upload(clc_handle* h, const std::vector<long long>& starts, long long n, bool any_z, const clc_observation* host_records,
                      const SmallFlatten* fj, bool* used) {
  *used = false;
  const size_t G = starts.size() - 1;
  if (!h->fast_small || n <= 0 || (size_t)n > kSmallMaxRecords || any_z || h->small_on_coop || G == 0 || G > 512) return CLC_OK;
  if ((h->launch_flags & clc::FLAG_NO_RESIDENT) != 0) return CLC_OK;
  // ---- the plan (host): rows per scan (scan_rows_kernel), points per lane of the 512-lane layout (res_plan_kernel) ----
  const bool sparse = G * 4 > (size_t)n;
  std::vector<unsigned int> brbeg(G + 1, 0u);
  for (size_t g = 0; g < G; ++g) brbeg[g + 1] = brbeg[g] + (unsigned int)((starts[g + 1] - starts[g] + clc::ROW - 1) / clc::ROW);
  const long long R = (long long)brbeg[G];
  const bool rows_ok = !sparse && R > 0 && (size_t)R * clc::ROW <= 3 * (size_t)n + 64;
  int ppl = 0;
  for (long long c = std::max<long long>(1, (n + 511) / 512); c <= kResPR512 + kResPL512 && ppl == 0; ++c) {
    long long lanes = 0;
    for (size_t g = 0; g < G && lanes <= 512; ++g) lanes += (starts[g + 1] - starts[g] + c - 1) / c;
    if (lanes <= 512) ppl = (int)c;
  }
  if (ppl == 0) return CLC_OK;  // scans that leave too many half-filled lanes: the generic path (cooperative layout)
  // ---- staging: [flatten's record offsets][rec_off, tile_off of the one problem][starts][gid][row_begin][res_row][records] ----
  const size_t P1 = fj ? (size_t)fj->n_poses + 1 : 0;
  const size_t o_foff = 0, o_off = align8(o_foff + P1 * 8), o_starts = o_off + 4 * 8, o_gid = o_starts + (G + 1) * 8,
               o_brbeg = align8(o_gid + (size_t)n * 4), o_row = align8(o_brbeg + (G + 1) * 4), o_rec = align8(o_row + 8),
               total = o_rec + (host_records ? (size_t)n * sizeof(clc_observation) : 0);
  CLC_HIP(hipSetDevice(h->device));
  if (total > h->stage_cap) {
    CLC_HIP(hipStreamSynchronize(h->stream));
    if (h->h_stage) CLC_HIP(hipHostFree(h->h_stage));
    if (h->d_stage) CLC_HIP(hipFree(h->d_stage));
    h->h_stage = h->d_stage = nullptr; h->stage_cap = 0; h->stage_busy = false;
    const size_t cap = std::max<size_t>(total + total / 2, (size_t)256 << 10);
    CLC_HIP(hipHostMalloc(reinterpret_cast<void**>(&h->h_stage), cap, hipHostMallocDefault));
    CLC_HIP(hipMalloc(reinterpret_cast<void**>(&h->d_stage), cap));
    h->stage_cap = cap;
  }
  if (!h->ev_stage) CLC_HIP(hipEventCreateWithFlags(&h->ev_stage, hipEventDisableTiming));
  if (h->stage_busy) CLC_HIP(hipEventSynchronize(h->ev_stage));  // the previous copy out of the pinned block (long done in practice)
  {
    char* b = h->h_stage;
    if (fj) std::memcpy(b + o_foff, fj->rec_off->data(), P1 * 8);
    long long* off = reinterpret_cast<long long*>(b + o_off);
    off[0] = 0; off[1] = n; off[2] = 0; off[3] = (n + clc::TILE - 1) / clc::TILE;
    std::memcpy(b + o_starts, starts.data(), (G + 1) * 8);
    unsigned int* gid = reinterpret_cast<unsigned int*>(b + o_gid);
    for (size_t g = 0; g < G; ++g)
      for (long long k = starts[g]; k < starts[g + 1]; ++k) gid[k] = (unsigned int)g;
    std::memcpy(b + o_brbeg, brbeg.data(), (G + 1) * 4);
    unsigned int* row = reinterpret_cast<unsigned int*>(b + o_row);
    row[0] = 0u; row[1] = (unsigned int)ppl;
    if (host_records) std::memcpy(b + o_rec, host_records, (size_t)n * sizeof(clc_observation));
  }
  // from here on things are enqueued: an error leaves the handle without observations rather than with half of them
  h->compact_ok = false; h->rows_ok = false; h->sres.ok = false; h->cres.ok = false; h->split_grid = -1; h->selection_key = -1;
  h->n_obs = 0;
  CLC_HIP(hipMemcpyAsync(h->d_stage, h->h_stage, total, hipMemcpyHostToDevice, h->stream));
  CLC_HIP(hipEventRecord(h->ev_stage, h->stream));
  h->stage_busy = true;
  const long long* d_off = reinterpret_cast<const long long*>(h->d_stage + o_off);
  const long long* d_starts = reinterpret_cast<const long long*>(h->d_stage + o_starts);
  const unsigned int* d_gid = reinterpret_cast<const unsigned int*>(h->d_stage + o_gid);
  const unsigned int* d_brbeg = reinterpret_cast<const unsigned int*>(h->d_stage + o_brbeg);
  const unsigned int* d_row = reinterpret_cast<const unsigned int*>(h->d_stage + o_row);
  const double* d_aos = reinterpret_cast<const double*>(h->d_stage + o_rec);
  const int threads = 256;
  int rc = CLC_OK;
  if (fj) {  // the selection's records, built on the device from the resident scans (bitwise clc_flatten_observations')
    rc = ensure_bytes(&h->d_small_aos, &h->small_aos_cap, (size_t)n * sizeof(clc_observation));
    if (rc != CLC_OK) return rc;
    long long* d_soff = reinterpret_cast<long long*>(h->d_soff);
    hipLaunchKernelGGL(clc::flatten_kernel, dim3((unsigned)fj->n_poses), dim3(clc::BLOCK), 0, h->stream, fj->n_poses, h->d_sq, h->d_st, d_soff,
                       h->d_spts, d_soff + P1, h->d_sptl, fj->linefit ? 1 : 0, fj->boundary ? 1 : 0,
                       reinterpret_cast<const long long*>(h->d_stage + o_foff), h->d_small_aos);
    CLC_HIP(hipGetLastError());
    d_aos = h->d_small_aos;
  }
  rc = retile_into(h, d_aos, (size_t)n, &h->d_tiles, &h->tiles_cap_bytes);
  if (rc != CLC_OK) return rc;
  // compact layout: group table (what the on-chip kernels take a lane's plane from) + 28-byte tiles
  const size_t tiles = (size_t)((n + clc::TILE - 1) / clc::TILE);
  rc = ensure_bytes(&h->d_ctiles, &h->ctiles_cap_bytes, std::max<size_t>(tiles, 1) * clc::CTILE_DOUBLES * sizeof(double));
  if (rc == CLC_OK) rc = ensure_bytes(&h->d_groups, &h->groups_cap_bytes, G * clc::GROUP_DOUBLES * sizeof(double));
  if (rc != CLC_OK) return rc;
  hipLaunchKernelGGL(clc::build_groups_dev_kernel, dim3((unsigned)((G + threads - 1) / threads)), dim3(threads), 0, h->stream, d_aos, d_starts,
                     (long long)G, h->d_groups);
  {
    const long long max_padded = (long long)tiles * clc::TILE;
    const unsigned ydim = (unsigned)std::min<long long>(4096, std::max<long long>(1, (max_padded + threads - 1) / threads));
    hipLaunchKernelGGL(clc::build_ctiles_kernel, dim3(1u, ydim), dim3(threads), 0, h->stream, d_aos, d_gid, d_off, d_off + 2, h->d_ctiles);
  }
  CLC_HIP(hipGetLastError());
  if (rows_ok) {  // row layout (one padding row each: see build_layouts)
    rc = ensure_bytes(&h->d_rxy, &h->rxy_cap_bytes, ((size_t)R + 1) * clc::ROW_DOUBLES * sizeof(double));
    if (rc == CLC_OK) rc = ensure_bytes(&h->d_rdesc, &h->rdesc_cap_bytes, ((size_t)R + 1) * sizeof(clc::RowDesc) + clc::wave_split_bytes(R));
    if (rc != CLC_OK) return rc;
    CLC_HIP(hipMemsetAsync(h->d_rxy + (size_t)R * clc::ROW_DOUBLES, 0, clc::ROW_DOUBLES * sizeof(double), h->stream));
    CLC_HIP(hipMemsetAsync(reinterpret_cast<char*>(h->d_rdesc) + (size_t)R * sizeof(clc::RowDesc), 0, sizeof(clc::RowDesc), h->stream));
    const long long slots = R * clc::ROW;
    hipLaunchKernelGGL(clc::build_rows_kernel, dim3((unsigned)((slots + threads - 1) / threads)), dim3(threads), 0, h->stream, d_aos, d_starts,
                       d_brbeg, (long long)G, R, (int)clc::ROW_DOUBLES, h->d_rxy, reinterpret_cast<clc::RowDesc*>(h->d_rdesc));
    CLC_HIP(hipGetLastError());
  }
  {  // the 512-lane layout of the single-workgroup solve (build_resident for one problem, first_try = 512)
    ResLayout& L = h->sres;
    rc = ensure_bytes(&L.d_row, &L.row_cap, 2 * sizeof(unsigned int));
    if (rc == CLC_OK) rc = ensure_bytes(&L.d_desc, &L.desc_cap, (size_t)512 * sizeof(clc::ResLane));
    if (rc == CLC_OK) rc = ensure_bytes(&L.d_xy, &L.xy_cap, ((size_t)ppl + 1) * 512 * 2 * sizeof(double));
    if (rc != CLC_OK) return rc;
    CLC_HIP(hipMemcpyAsync(L.d_row, d_row, 2 * sizeof(unsigned int), hipMemcpyDeviceToDevice, h->stream));
    CLC_HIP(hipMemsetAsync(L.d_xy + (size_t)ppl * 512 * 2, 0, (size_t)512 * 2 * sizeof(double), h->stream));
    hipLaunchKernelGGL((clc::res_build_kernel<512>), dim3(1), dim3(512), 0, h->stream, d_aos, d_off, d_gid, d_starts, n, (long long)G, d_row,
                       reinterpret_cast<clc::ResLane*>(L.d_desc), L.d_xy, (double*)nullptr);
    CLC_HIP(hipGetLastError());
    L.lanes = 512; L.max_ppl = ppl; L.uni_ppl = ppl; L.rows = ppl; L.with_z = false; L.ok = true;
  }
  h->n_obs = (size_t)n;
  h->n_groups = (long long)G;
  h->compact_ok = !sparse;
  h->n_rows = R;
  h->rows_ok = rows_ok;
  h->rows_z = false;
  ++h->fast_small_uploads;
  *used = true;
  return CLC_OK;
}

// The scan structure of a selection of the stored scans, on the host: record k starts a scan when its (n, d, scale) differ bitwise from
// record k - 1's (scan_flag_kernel's rule) — the records of one pose share the tag plane and the scale, its two board-edge records
// carry the edge planes.  false: not known on the host (stored scans too large: clc_store_observations keeps no host copy).
bool plan_selection_scans(const clc_handle* h, bool linefit, bool boundary, const std::vector<long long>& rec_off, std::vector<long long>* starts,
                          bool* any_z) {
  if (!h->store_small) return false;
  const int P = h->store_poses;
  const std::vector<long long>& off = linefit ? h->s_ptl_off : h->s_pts_off;
  starts->clear();
  unsigned long long prev[5] = {0, 0, 0, 0, 0};
  bool have = false;
  auto record = [&](long long k, const double* n3, double d, double scale) {
    unsigned long long key[5];
    std::memcpy(key, n3, 24); std::memcpy(key + 3, &d, 8); std::memcpy(key + 4, &scale, 8);
    if (!have || std::memcmp(key, prev, sizeof(key)) != 0) starts->push_back(k);
    std::memcpy(prev, key, sizeof(key));
    have = true;
  };
  for (int i = 0; i < P; ++i) {
    const long long cnt = off[(size_t)i + 1] - off[(size_t)i];
    const bool edges = boundary && linefit;
    if (cnt <= 0 && !edges) continue;
    clc::host::PosePlanes pp;
    clc::host::pose_planes(h->s_tag_q.data(), h->s_tag_t.data(), i, edges, pp);
    const double scale = 1.0 / std::sqrt((double)cnt);  // :239-240
    long long k = rec_off[(size_t)i];
    if (cnt > 0) record(k, pp.n, pp.d, scale);
    k += cnt;
    if (edges) {
      record(k, pp.pi1, pp.pi1[3], scale);
      record(k + 1, pp.pi2, pp.pi2[3], scale);
    }
  }
  starts->push_back(rec_off[(size_t)P]);
  *any_z = (linefit ? h->s_any_z_ptl : h->s_any_z_pts) || (boundary && linefit && h->s_any_z_ends);
  return true;
}

// ... and of an array of records on the host (clc_upload).
void plan_record_scans(const clc_observation* rec, long long n, std::vector<long long>* starts, bool* any_z) {
  starts->clear();
  bool z = false;
  for (long long k = 0; k < n; ++k) {
    const clc_observation& a = rec[k];
    bool nw = k == 0;
    if (!nw) {
      const clc_observation& b = rec[k - 1];
      nw = std::memcmp(a.n, b.n, 32) != 0 || std::memcmp(&a.scale, &b.scale, 8) != 0;  // n[3], d are contiguous: 32 bytes
    }
    if (nw) starts->push_back(k);
    z = z || a.p[2] != 0.0;
  }
  starts->push_back(n);
  *any_z = z;
}

}  // namespace clc_abi

extern "C" {

int clc_flatten_observations(int n_poses, const double* tag_q_wxyz, const double* tag_t,
                             const int64_t* pts_off, const double* pts, const int64_t* ptl_off,
                             const double* ptl, int use_linefitting_data,
                             int use_boundary_constraint, clc_observation* records,
                             int64_t* n_records) {
  if (n_poses < 0 || !n_records || (n_poses > 0 && (!tag_q_wxyz || !tag_t || !pts_off || !ptl_off)))
    return fail(CLC_ERR_INVALID_ARG, "clc_flatten_observations: bad argument");
  int rc = clc::host::flatten(n_poses, tag_q_wxyz, tag_t, pts_off, pts, ptl_off, ptl,
                              use_linefitting_data != 0, use_boundary_constraint != 0, records, n_records);
  if (rc == CLC_ERR_EMPTY_SCAN)
    return fail(rc, "clc_flatten_observations: boundary constraint on an empty scan (reference: std::out_of_range at LaseCamCalCeres.cpp:278)");
  return rc;
}

int clc_upload_device(clc_handle* h, const clc_observation* records_dev, size_t n) {
  if (!h || (n > 0 && !records_dev)) return fail(CLC_ERR_INVALID_ARG, "clc_upload_device: bad argument");
  CLC_HIP(hipSetDevice(h->device));
  h->compact_ok = false;
  h->rows_ok = false;
  h->sres.ok = false;
  h->cres.ok = false;
  h->split_grid = -1;
  h->selection_key = -1;
  int rc = retile_into(h, reinterpret_cast<const double*>(records_dev), n, &h->d_tiles, &h->tiles_cap_bytes);
  if (rc != CLC_OK) return rc;
  CLC_HIP(hipStreamSynchronize(h->stream));
  h->n_obs = n;
  const std::vector<long long> rec_off = {0, (long long)n};
  const std::vector<long long> tile_off = {0, (long long)((n + clc::TILE - 1) / clc::TILE)};
  const LayoutTargets T = {&h->d_ctiles, &h->ctiles_cap_bytes, &h->d_groups, &h->groups_cap_bytes, &h->n_groups, &h->compact_ok,
                           &h->d_rxy, &h->rxy_cap_bytes, &h->d_rdesc, &h->rdesc_cap_bytes, &h->n_rows, &h->rows_ok, nullptr,
                           // a problem one workgroup can hold (<= 512 lanes x 22 points) also gets the lane layout: clc_solve then runs
                           // its whole LM loop in ONE single-workgroup launch from registers + LDS (solve_resident_single)
                           n <= (size_t)512 * (kResPR512 + kResPL512) ? &h->sres : nullptr, &h->rows_z,
                           // what one workgroup cannot hold (more points; or more than 512 scans, or scans whose lengths leave
                           // too many half-filled lanes: a lane holds points of ONE scan), up to 65 536 lanes x 40 points, is
                           // dealt to 256 workgroups: the cooperative solve (clc_coop.hpp)
                           n <= (size_t)clc::COOP_WGS * clc::COOP_NL * (clc::COOP_PR + clc::COOP_PL) ? &h->cres : nullptr};
  return build_layouts(h, reinterpret_cast<const double*>(records_dev), n, rec_off, tile_off, T);
}

int clc_upload(clc_handle* h, const clc_observation* records, size_t n) {
  if (!h || (n > 0 && !records)) return fail(CLC_ERR_INVALID_ARG, "clc_upload: bad argument");
  CLC_HIP(hipSetDevice(h->device));
  if (n > 0 && n <= kSmallMaxRecords && h->fast_small) {  // a problem one workgroup holds: planned on the host, no read-backs
    std::vector<long long> starts;
    bool any_z = false, used = false;
    plan_record_scans(records, (long long)n, &starts, &any_z);
    const int rc = small_fast_upload(h, starts, (long long)n, any_z, records, nullptr, &used);
    if (rc != CLC_OK || used) return rc;
  }
  DevBuf<double> aos(&h->pool);
  if (n > 0) {
    CLC_HIP(aos.alloc(n * 8));
    CLC_HIP(hipMemcpy(aos.p, records, n * sizeof(clc_observation), hipMemcpyHostToDevice));
  }
  return clc_upload_device(h, reinterpret_cast<const clc_observation*>(aos.p), n);
}

}  // extern "C"

// ---- resident scans + device-side problem assembly ---------------------------------------------------------------
namespace {

int grow(double** p, size_t* cap, size_t bytes) { return ensure_bytes(p, cap, std::max<size_t>(bytes, 8)); }

// per-pose record offsets of a selection (host, O(poses)); CLC_ERR_EMPTY_SCAN mirrors the reference's .at(0) throw

int selection_offsets(const clc_handle* h, bool linefit, bool boundary, std::vector<long long>* rec_off) {
  const int P = h->store_poses;
  rec_off->assign((size_t)P + 1, 0);
  const std::vector<long long>& off = linefit ? h->s_ptl_off : h->s_pts_off;
  for (int i = 0; i < P; ++i) {
    long long c = off[(size_t)i + 1] - off[(size_t)i];
    if (boundary && linefit) {
      if (h->s_pts_off[(size_t)i + 1] - h->s_pts_off[(size_t)i] <= 0) return CLC_ERR_EMPTY_SCAN;
      c += 2;
    }
    (*rec_off)[(size_t)i + 1] = (*rec_off)[(size_t)i] + c;
  }
  return CLC_OK;
}


}  // namespace

namespace clc_abi {
// builds the records of the selection on the device into *aos (allocated here)
int flatten_on_device(clc_handle* h, bool linefit, bool boundary, DevBuf<double>* aos, long long* n_out) {
  if (h->store_poses < 0) return fail(CLC_ERR_NO_DATA, "clc_select_observations: no scans stored (clc_store_observations)");
  std::vector<long long> rec_off;
  const int rc = selection_offsets(h, linefit, boundary, &rec_off);
  if (rc == CLC_ERR_EMPTY_SCAN)
    return fail(rc, "clc_select_observations: boundary constraint on an empty scan (reference: std::out_of_range at LaseCamCalCeres.cpp:278)");
  const int P = h->store_poses;
  const long long N = rec_off[(size_t)P];
  *n_out = N;
  CLC_HIP(aos->alloc((size_t)std::max<long long>(N, 1) * 8));
  if (N == 0 || P == 0) return CLC_OK;
  long long* d_off = reinterpret_cast<long long*>(h->d_soff);
  CLC_HIP(hipMemcpyAsync(d_off + 2 * ((size_t)P + 1), rec_off.data(), sizeof(long long) * ((size_t)P + 1), hipMemcpyHostToDevice,
                         h->stream));
  hipLaunchKernelGGL(clc::flatten_kernel, dim3((unsigned)P), dim3(clc::BLOCK), 0, h->stream, P, h->d_sq, h->d_st, d_off,
                     h->d_spts, d_off + ((size_t)P + 1), h->d_sptl, linefit ? 1 : 0, boundary ? 1 : 0,
                     d_off + 2 * ((size_t)P + 1), aos->p);
  CLC_HIP(hipGetLastError());
  CLC_HIP(hipStreamSynchronize(h->stream));  // rec_off (host vector) must outlive the copy
  return CLC_OK;
}


}  // namespace clc_abi

extern "C" {

int clc_store_observations(clc_handle* h, int n_poses, const double* tag_q_wxyz, const double* tag_t,
                           const int64_t* pts_off, const double* pts, const int64_t* ptl_off, const double* ptl) {
  if (!h || n_poses < 0 || (n_poses > 0 && (!tag_q_wxyz || !tag_t || !pts_off || !ptl_off)))
    return fail(CLC_ERR_INVALID_ARG, "clc_store_observations: bad argument");
  const size_t P = (size_t)n_poses;
  h->store_poses = -1;
  h->s_pts_off.assign(P + 1, 0);
  h->s_ptl_off.assign(P + 1, 0);
  for (size_t i = 0; i <= P && P > 0; ++i) {
    h->s_pts_off[i] = pts_off[i] - pts_off[0];
    h->s_ptl_off[i] = ptl_off[i] - ptl_off[0];
    if (i > 0 && (h->s_pts_off[i] < h->s_pts_off[i - 1] || h->s_ptl_off[i] < h->s_ptl_off[i - 1]))
      return fail(CLC_ERR_INVALID_ARG, "clc_store_observations: offsets not monotone");
  }
  const size_t M = (size_t)h->s_pts_off[P], ML = (size_t)h->s_ptl_off[P];
  if ((M > 0 && !pts) || (ML > 0 && !ptl)) return fail(CLC_ERR_INVALID_ARG, "clc_store_observations: NULL points");
  CLC_HIP(hipSetDevice(h->device));
  int rc = grow(&h->d_sq, &h->sq_cap, P * 4 * sizeof(double));
  if (rc == CLC_OK) rc = grow(&h->d_st, &h->st_cap, P * 3 * sizeof(double));
  if (rc == CLC_OK) rc = grow(&h->d_spts, &h->spts_cap, M * 3 * sizeof(double));
  if (rc == CLC_OK) rc = grow(&h->d_sptl, &h->sptl_cap, ML * 3 * sizeof(double));
  if (rc == CLC_OK) rc = grow(&h->d_soff, &h->soff_cap, 3 * (P + 1) * sizeof(long long));
  if (rc != CLC_OK) return rc;
  if (P > 0) {
    CLC_HIP(hipMemcpyAsync(h->d_sq, tag_q_wxyz, P * 4 * sizeof(double), hipMemcpyHostToDevice, h->stream));
    CLC_HIP(hipMemcpyAsync(h->d_st, tag_t, P * 3 * sizeof(double), hipMemcpyHostToDevice, h->stream));
    long long* d_off = reinterpret_cast<long long*>(h->d_soff);
    CLC_HIP(hipMemcpyAsync(d_off, h->s_pts_off.data(), (P + 1) * sizeof(long long), hipMemcpyHostToDevice, h->stream));
    CLC_HIP(hipMemcpyAsync(d_off + (P + 1), h->s_ptl_off.data(), (P + 1) * sizeof(long long), hipMemcpyHostToDevice, h->stream));
  }
  if (M > 0) CLC_HIP(hipMemcpyAsync(h->d_spts, pts + 3 * pts_off[0], M * 3 * sizeof(double), hipMemcpyHostToDevice, h->stream));
  if (ML > 0) CLC_HIP(hipMemcpyAsync(h->d_sptl, ptl + 3 * ptl_off[0], ML * 3 * sizeof(double), hipMemcpyHostToDevice, h->stream));
  // reference-size scans: the host keeps what it needs to plan a selection's layouts itself (tag poses, whether any z is non-zero)
  h->store_small = P > 0 && P <= 4096 && M <= 16384 && ML <= 16384;
  if (h->store_small) {
    h->s_tag_q.assign(tag_q_wxyz, tag_q_wxyz + 4 * P);
    h->s_tag_t.assign(tag_t, tag_t + 3 * P);
    auto any_z = [](const double* p, size_t count) { for (size_t k = 0; k < count; ++k) if (p[3 * k + 2] != 0.0) return true; return false; };
    h->s_any_z_pts = M > 0 && any_z(pts + 3 * pts_off[0], M);
    h->s_any_z_ptl = ML > 0 && any_z(ptl + 3 * ptl_off[0], ML);
    h->s_any_z_ends = false;
    for (size_t i = 0; i < P; ++i)
      if (pts_off[i + 1] > pts_off[i] && (pts[3 * pts_off[i] + 2] != 0.0 || pts[3 * (pts_off[i + 1] - 1) + 2] != 0.0)) h->s_any_z_ends = true;
  }
  // points_on_line bit-identical to points (the simulation node's input): the selections (linefit, no boundary) and (no linefit) are
  // then the same records and clc_select_observations builds them once.  Checked for reference-size inputs only (a memcmp of <= 384 KB).
  h->store_lines_equal_points = M == ML && M <= 16384 && h->s_pts_off == h->s_ptl_off &&
                                (M == 0 || std::memcmp(pts + 3 * pts_off[0], ptl + 3 * ptl_off[0], M * 3 * sizeof(double)) == 0);
  CLC_HIP(hipStreamSynchronize(h->stream));  // the caller's arrays may go away
  h->store_poses = n_poses;
  h->store_generation++;
  return CLC_OK;
}

int64_t clc_store_generation(const clc_handle* h) { return h ? h->store_generation : -1; }

int clc_select_observations(clc_handle* h, int use_linefitting_data, int use_boundary_constraint, int64_t* n_records) {
  if (!h) return fail(CLC_ERR_INVALID_ARG, "clc_select_observations: NULL handle");
  CLC_HIP(hipSetDevice(h->device));
  // The selection the handle's observation array already IS (same stored scans, same residual blocks): nothing to build.  The sequence
  // of main/calibr_offline.cpp:166-170 selects points_on_line for the closed form and again for the refinement; the simulation node
  // stores points_on_line == points (main/calibr_simulation.cpp:102), so its two selections are the same records as well.
  bool linefit = use_linefitting_data != 0;
  const bool boundary = use_boundary_constraint != 0 && linefit;  // (the board-edge terms exist with points_on_line only, :258)
  if (linefit && !boundary && h->store_lines_equal_points) linefit = false;
  const long long key = h->store_poses < 0 ? -1 : h->store_generation * 4 + (linefit ? 2 : 0) + (boundary ? 1 : 0);
  const long long cfg = ((long long)h->launch_flags << 8) | ((long long)h->auto_disable << 1) | (h->small_on_coop ? 1 : 0);  // what an upload depends on
  if (key >= 0 && key == h->selection_key && cfg == h->selection_cfg) {
    if (n_records) *n_records = (int64_t)h->n_obs;
    return CLC_OK;
  }
  if (h->store_small && h->fast_small && h->store_poses > 0) {  // reference-size scans: planned on the host, enqueued without a read-back
    std::vector<long long> rec_off, starts;
    const int orc = selection_offsets(h, linefit, boundary, &rec_off);
    bool any_z = false, used = false;
    if (orc == CLC_OK && rec_off.back() > 0 && (size_t)rec_off.back() <= kSmallMaxRecords &&
        plan_selection_scans(h, linefit, boundary, rec_off, &starts, &any_z)) {
      const SmallFlatten fj = {h->store_poses, linefit, boundary, &rec_off};
      const int frc = small_fast_upload(h, starts, rec_off.back(), any_z, nullptr, &fj, &used);
      if (frc != CLC_OK) return frc;
      if (used) {
        if (n_records) *n_records = (int64_t)rec_off.back();
        h->selection_key = key; h->selection_cfg = cfg;
        return CLC_OK;
      }
    }
  }
  DevBuf<double> aos(&h->pool);
  long long N = 0;
  int rc = flatten_on_device(h, linefit, boundary, &aos, &N);
  if (rc != CLC_OK) return rc;
  if (n_records) *n_records = (int64_t)N;
  rc = clc_upload_device(h, reinterpret_cast<const clc_observation*>(aos.p), (size_t)N);
  if (rc == CLC_OK) { h->selection_key = key; h->selection_cfg = cfg; }  // (clc_upload_device — any upload — forgets the previous selection)
  return rc;
}


}  // extern "C"

// ---- batched ---------------------------------------------------------------------------
namespace {
// records: host pointer (on_device = false: staged through a temporary device buffer) or device pointer
int upload_batched_impl(clc_handle* h, const clc_observation* records, bool on_device, const int64_t* offsets,
                        size_t n_problems) {
  if (!h || !offsets || (n_problems > 0 && !records))
    return fail(CLC_ERR_INVALID_ARG, "clc_upload_batched: bad argument");
  CLC_HIP(hipSetDevice(h->device));
  {  // the batched unit's code object: loaded here, once per device and process, before the first batched solve (see clc_create)
    static std::atomic<bool> warm[64];
    std::atomic<bool>& w = warm[h->device >= 0 && h->device < 64 ? h->device : 0];
    if (!w.exchange(true) && !std::getenv("CLC_LAZY_MODULES")) warm_batched();
  }
  const size_t P = n_problems;
  std::vector<long long> tile_off(P + 1, 0), nobs(P, 0);
  long long max_tiles = 0;
  for (size_t k = 0; k < P; ++k) {
    const int64_t n = offsets[k + 1] - offsets[k];
    if (n < 0) return fail(CLC_ERR_INVALID_ARG, "clc_upload_batched: offsets not monotone");
    nobs[k] = n;
    tile_off[k + 1] = tile_off[k] + (n + clc::TILE - 1) / clc::TILE;
    max_tiles = std::max<long long>(max_tiles, tile_off[k + 1] - tile_off[k]);
  }
  h->batch_max_tiles = max_tiles;
  const size_t total_tiles = (size_t)tile_off[P];
  h->batch_total_tiles = total_tiles;
  const size_t bytes = std::max<size_t>(total_tiles, 1) * clc::TILE_DOUBLES * sizeof(double);
  if (bytes > h->btiles_cap_bytes) {
    if (h->d_btiles) CLC_HIP(hipFree(h->d_btiles));
    h->d_btiles = nullptr; h->btiles_cap_bytes = 0;
    CLC_HIP(hipMalloc(&h->d_btiles, bytes));
    h->btiles_cap_bytes = bytes;
  }
  if (P > h->problems_cap) {
    void* olds[] = {h->d_tile_off, h->d_nobs, h->d_states, h->d_results, h->d_prob_row};
    for (void* p : olds) if (p) CLC_HIP(hipFree(p));
    if (h->h_poses) CLC_HIP(hipHostFree(h->h_poses));
    if (h->h_summaries) CLC_HIP(hipHostFree(h->h_summaries));
    h->h_poses = nullptr; h->h_summaries = nullptr;
    h->d_tile_off = nullptr; h->d_nobs = nullptr; h->d_poses = nullptr; h->d_summaries = nullptr;
    h->d_states = nullptr; h->d_results = nullptr; h->results_valid = 0; h->d_prob_row = nullptr;
    h->problems_cap = 0;
    CLC_HIP(hipMalloc(&h->d_tile_off, sizeof(long long) * (P + 1)));
    CLC_HIP(hipMalloc(&h->d_nobs, sizeof(long long) * P));
    CLC_HIP(hipHostMalloc(&h->h_poses, sizeof(double) * 7 * P, hipHostMallocMapped));
    CLC_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&h->d_poses), h->h_poses, 0));
    CLC_HIP(hipHostMalloc(&h->h_summaries, sizeof(clc_summary) * P, hipHostMallocMapped));
    CLC_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&h->d_summaries), h->h_summaries, 0));
    CLC_HIP(hipMalloc(&h->d_states, sizeof(clc::LmState) * P));
    CLC_HIP(hipMalloc(&h->d_results, sizeof(clc_result_record) * P));
    CLC_HIP(hipMalloc(&h->d_prob_row, sizeof(long long) * (P + 1)));
    h->problems_cap = P;
  }
  if (P == 0) { h->n_problems = 0; return CLC_OK; }
  CLC_HIP(hipMemcpy(h->d_tile_off, tile_off.data(), sizeof(long long) * (P + 1), hipMemcpyHostToDevice));
  CLC_HIP(hipMemcpy(h->d_nobs, nobs.data(), sizeof(long long) * P, hipMemcpyHostToDevice));
  // stage the AoS records, then re-tile every problem into its own whole tiles
  const size_t n_total = (size_t)(offsets[P] - offsets[0]);
  DevBuf<double> baos(&h->pool);
  DevBuf<long long> boff(&h->pool);
  double* d_aos = nullptr;
  if (n_total > 0 && on_device) {
    d_aos = const_cast<double*>(reinterpret_cast<const double*>(records + offsets[0]));
  } else if (n_total > 0) {
    CLC_HIP(baos.alloc(n_total * 8));
    CLC_HIP(hipMemcpy(baos.p, records + offsets[0], n_total * sizeof(clc_observation), hipMemcpyHostToDevice));
    d_aos = baos.p;
  }
  std::vector<long long> rel(P + 1);
  for (size_t k = 0; k <= P; ++k) rel[k] = offsets[k] - offsets[0];
  CLC_HIP(boff.alloc(P + 1));
  long long* d_off = boff.p;
  CLC_HIP(hipMemcpy(d_off, rel.data(), sizeof(long long) * (P + 1), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(clc::retile_batched_kernel, dim3((unsigned)P), dim3(256), 0, h->stream, d_aos, d_off,
                     h->d_tile_off, h->d_btiles);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  int crc = CLC_OK;
  h->bcompact_ok = false;
  h->brows_ok = false;
  h->bres.ok = false;
  h->results_valid = 0;
  if (e == hipSuccess && d_aos) {
    const LayoutTargets T = {&h->d_bctiles, &h->bctiles_cap_bytes, &h->d_bgroups, &h->bgroups_cap_bytes, &h->bn_groups,
                             &h->bcompact_ok, &h->d_brxy, &h->brxy_cap_bytes, &h->d_brdesc, &h->brdesc_cap_bytes, &h->bn_rows,
                             &h->brows_ok, &h->d_prob_row, &h->bres, &h->brows_z};
    crc = build_layouts(h, d_aos, n_total, rel, tile_off, T);
  }
  if (e != hipSuccess) return fail(CLC_ERR_HIP, "clc_upload_batched: retile", e);
  if (crc != CLC_OK) return crc;
  h->batch_max_rows = 0;
  if (h->brows_ok) {  // O(P) words back: the whole-solve kernel is chosen on the real longest problem, not an estimate
    std::vector<long long> pr(P + 1);
    CLC_HIP(hipMemcpy(pr.data(), h->d_prob_row, sizeof(long long) * (P + 1), hipMemcpyDeviceToHost));
    for (size_t k = 0; k < P; ++k) h->batch_max_rows = std::max(h->batch_max_rows, pr[k + 1] - pr[k]);
  }
  h->n_problems = P;
  return CLC_OK;
}
}  // namespace

int clc_upload_batched(clc_handle* h, const clc_observation* records, const int64_t* offsets, size_t n_problems) {
  return upload_batched_impl(h, records, false, offsets, n_problems);
}

int clc_upload_batched_device(clc_handle* h, const clc_observation* records_dev, const int64_t* offsets, size_t n_problems) {
  return upload_batched_impl(h, records_dev, true, offsets, n_problems);
}

size_t clc_num_problems(const clc_handle* h) { return h ? h->n_problems : 0; }


// rtc_pairs.hip -- all-pairs |A_i ∩ A_j| over sorted distinct sketches.
//
// Produces the integers compute_minhash_mst / compute_kssd_mst get from their inverted index
// (src/MST.cpp:1408-1435, :428-487 in the reference tree) for every (i, j) of a tile, i.e. the
// dense form of modifyMST's pair loop (src/MST.cpp:851-866).
//
//   algo 1  pair_merge_kernel : one lane per pair, two-pointer merge straight from global memory.
//           Handles any sketch size / width; used as fallback and as on-device cross-check.
//   algo 2  (rtc_pairs_tiled.hip) LDS mask-table tiles.
#include <algorithm>
#include <vector>

#include "rtc_internal.h"

int rtc_pair_common_tiled(rtc_ctx* ctx, const void* d_hashes, int width, const uint64_t* d_start,
                          const uint32_t* d_len, uint32_t n, uint32_t row0, uint32_t row1,
                          uint32_t col0, uint32_t col1, uint32_t* d_common, uint64_t ld,
                          int lower_only, int* handled);

int rtc_pair_edges_tiled(rtc_ctx* ctx, const void* d_hashes, int width, const uint64_t* d_start,
                         const uint32_t* d_len, uint32_t n, uint32_t row0, uint32_t row1, uint32_t col0,
                         uint32_t col1, int lower_only, int radio, rtc_cedge* d_edges, uint64_t cap,
                         uint64_t* d_count, int* handled);

namespace {

template <typename T>
__global__ __launch_bounds__(256) void pair_merge_kernel(const T* __restrict__ hashes,
                                                         const uint64_t* __restrict__ start,
                                                         const uint32_t* __restrict__ len,
                                                         uint32_t row0, uint32_t row1, uint32_t col0,
                                                         uint32_t col1, uint32_t* __restrict__ out,
                                                         uint64_t ld, int lower_only) {
  const uint32_t col = col0 + blockIdx.x * 64 + (threadIdx.x & 63);
  const uint32_t row = row0 + blockIdx.y * 4 + (threadIdx.x >> 6);
  if (row >= row1 || col >= col1) return;
  if (lower_only && col >= row) return;
  const T* a = hashes + start[row];
  const T* b = hashes + start[col];
  const uint32_t na = len[row], nb = len[col];
  uint32_t i = 0, j = 0, c = 0;
  if (na && nb) {
    T va = a[0], vb = b[0];
    while (true) {
      if (va < vb) { if (++i >= na) break; va = a[i]; }
      else if (vb < va) { if (++j >= nb) break; vb = b[j]; }
      else { c++; ++i; ++j; if (i >= na || j >= nb) break; va = a[i]; vb = b[j]; }
    }
  }
  out[(uint64_t)(row - row0) * ld + (col - col0)] = c;
}

// Mash's pairwise estimator, i.e. what Sketch::MinHash::jaccard()/distance() return to the dense loop
// modifyMST (src/MST.cpp:851-866) -- RabbitSketch is absent from the reference tree, this restates the
// published Mash algorithm ([U]): merge the two ascending lists, stop after
// `sketch_size` elements of the UNION, common = shared elements among them, denom = union elements seen.
template <typename T>
__global__ __launch_bounds__(256) void pair_mash_kernel(const T* __restrict__ hashes, const uint64_t* __restrict__ start,
                                                        const uint32_t* __restrict__ len, uint32_t sketch_size,
                                                        uint32_t row0, uint32_t row1, uint32_t col0, uint32_t col1,
                                                        uint32_t* __restrict__ common_out, uint32_t* __restrict__ denom_out,
                                                        uint64_t ld) {
  const uint32_t col = col0 + blockIdx.x * 64 + (threadIdx.x & 63);
  const uint32_t row = row0 + blockIdx.y * 4 + (threadIdx.x >> 6);
  if (row >= row1 || col >= col1) return;
  const T* a = hashes + start[row];
  const T* b = hashes + start[col];
  const uint32_t na = len[row], nb = len[col];
  uint32_t i = 0, j = 0, c = 0, d = 0;
  while (d < sketch_size && i < na && j < nb) {
    const T va = a[i], vb = b[j];
    if (va < vb) i++;
    else if (vb < va) j++;
    else { c++; i++; j++; }
    d++;
  }
  if (d < sketch_size) {  // one list exhausted: the rest of the other one still belongs to the union
    const uint32_t rest = (i < na ? na - i : 0) + (j < nb ? nb - j : 0);
    d += rest < sketch_size - d ? rest : sketch_size - d;
  }
  const uint64_t o = (uint64_t)(row - row0) * ld + (col - col0);
  common_out[o] = c;
  denom_out[o] = d;
}

}  // namespace

extern "C" int rtc_pair_mash_dev(rtc_ctx* ctx, const void* d_hashes, int width, const uint64_t* d_start,
                                 const uint32_t* d_len, uint32_t n, uint32_t sketch_size, uint32_t row0, uint32_t row1,
                                 uint32_t col0, uint32_t col1, uint32_t* d_common, uint32_t* d_denom, uint64_t ld) {
  if (!ctx || !d_start || !d_len || !d_common || !d_denom) return RTC_ERR_ARG;
  if (width != 4 && width != 8) return rtc_fail(ctx, RTC_ERR_ARG, "width must be 4 or 8");
  if (row1 > n || col1 > n || row0 > row1 || col0 > col1) return rtc_fail(ctx, RTC_ERR_ARG, "tile outside [0,n)");
  if (ld < (uint64_t)(col1 - col0)) return rtc_fail(ctx, RTC_ERR_ARG, "ld smaller than tile width");
  if (row0 == row1 || col0 == col1) return RTC_OK;
  if (!d_hashes) return RTC_ERR_ARG;
  RTC_HIP(ctx, hipSetDevice(ctx->device));
  dim3 grid((col1 - col0 + 63) / 64, (row1 - row0 + 3) / 4);
  if (grid.y > 65535) return rtc_fail(ctx, RTC_ERR_UNSUPPORTED, "more than 262140 rows per call");
  if (width == 8)
    hipLaunchKernelGGL(pair_mash_kernel<uint64_t>, grid, dim3(256), 0, ctx->stream, (const uint64_t*)d_hashes, d_start, d_len,
                       sketch_size, row0, row1, col0, col1, d_common, d_denom, ld);
  else
    hipLaunchKernelGGL(pair_mash_kernel<uint32_t>, grid, dim3(256), 0, ctx->stream, (const uint32_t*)d_hashes, d_start, d_len,
                       sketch_size, row0, row1, col0, col1, d_common, d_denom, ld);
  RTC_CHECK_LAUNCH(ctx);
  return RTC_OK;
}

extern "C" int rtc_pair_common_dev(rtc_ctx* ctx, const void* d_hashes, int width,
                                   const uint64_t* d_start, const uint32_t* d_len, uint32_t n,
                                   uint32_t row0, uint32_t row1, uint32_t col0, uint32_t col1,
                                   uint32_t* d_common, uint64_t ld, int lower_only, int algo) {
  if (!ctx || !d_start || !d_len || !d_common) return RTC_ERR_ARG;
  if (width != 4 && width != 8) return rtc_fail(ctx, RTC_ERR_ARG, "width must be 4 or 8");
  if (row1 > n || col1 > n || row0 > row1 || col0 > col1) return rtc_fail(ctx, RTC_ERR_ARG, "tile outside [0,n)");
  if (ld < (uint64_t)(col1 - col0)) return rtc_fail(ctx, RTC_ERR_ARG, "ld smaller than tile width");
  if (algo < 0 || algo > 2) return rtc_fail(ctx, RTC_ERR_ARG, "unknown algo %d", algo);
  if (row0 == row1 || col0 == col1) return RTC_OK;
  if (!d_hashes) return RTC_ERR_ARG;
  RTC_HIP(ctx, hipSetDevice(ctx->device));
  if (algo == 0 || algo == 2) {
    int handled = 0;
    int st = rtc_pair_common_tiled(ctx, d_hashes, width, d_start, d_len, n, row0, row1, col0, col1,
                                   d_common, ld, lower_only, &handled);
    if (st != RTC_OK) return st;
    if (handled) return RTC_OK;
    if (algo == 2) return rtc_fail(ctx, RTC_ERR_UNSUPPORTED, "tiled pair kernel cannot take this input");
  }
  dim3 grid((col1 - col0 + 63) / 64, (row1 - row0 + 3) / 4);
  if (grid.y > 65535) return rtc_fail(ctx, RTC_ERR_UNSUPPORTED, "more than 262140 rows per call in merge path");
  if (width == 8)
    hipLaunchKernelGGL(pair_merge_kernel<uint64_t>, grid, dim3(256), 0, ctx->stream,
                       (const uint64_t*)d_hashes, d_start, d_len, row0, row1, col0, col1, d_common, ld, lower_only);
  else
    hipLaunchKernelGGL(pair_merge_kernel<uint32_t>, grid, dim3(256), 0, ctx->stream,
                       (const uint32_t*)d_hashes, d_start, d_len, row0, row1, col0, col1, d_common, ld, lower_only);
  RTC_CHECK_LAUNCH(ctx);
  return RTC_OK;
}

// Candidate edges of a tile.  Sparse inputs (few co-occurrences for the size of the tile): the inverted join
// (rtc_pairs_join.hip).  Otherwise the tiled path: survivors are emitted by the pair kernel itself.  Inputs
// the tiled scheme cannot take go through the per-pair merge kernel into a dense scratch matrix
// (row chunks of <= 1 GiB) that rtc_extract_edges_dev filters.
namespace { __global__ void add_count_kernel(unsigned long long* count, unsigned long long add) { *count += add; } }

extern "C" int rtc_pair_last_path(const rtc_ctx* ctx) { return ctx ? ctx->pair_last_path : 0; }
extern "C" int rtc_diag_counters(const rtc_ctx* ctx, uint64_t out[8]) {
  if (!ctx || !out) return RTC_ERR_ARG;
  for (int i = 0; i < 8; i++) out[i] = ctx->diag[i];
  return RTC_OK;
}

extern "C" int rtc_pair_edges_dev(rtc_ctx* ctx, const void* d_hashes, int width, const uint64_t* d_start,
                                  const uint32_t* d_len, uint32_t n, uint32_t row0, uint32_t row1, uint32_t col0,
                                  uint32_t col1, int radio, rtc_cedge* d_edges, uint64_t cap, uint64_t* d_count) {
  if (!ctx || !d_start || !d_len || !d_count || (cap && !d_edges)) return RTC_ERR_ARG;
  if (width != 4 && width != 8) return rtc_fail(ctx, RTC_ERR_ARG, "width must be 4 or 8");
  if (row1 > n || col1 > n || row0 > row1 || col0 > col1) return rtc_fail(ctx, RTC_ERR_ARG, "tile outside [0,n)");
  if (row0 == row1 || col0 == col1) return RTC_OK;
  if (!d_hashes) return RTC_ERR_ARG;
  RTC_HIP(ctx, hipSetDevice(ctx->device));
  int handled = 0;
  if (!ctx->opt.pair_force_merge) {
    RTC_TRY(rtc_pair_edges_join(ctx, d_hashes, width, d_start, d_len, n, row0, row1, col0, col1, radio, d_edges, cap, d_count,
                                1.0, &handled));
    if (handled) { ctx->pair_last_path = 3; ctx->diag[0]++; return RTC_OK; }
    // The join has just refused the set as dense from its sample and expects more candidate edges than the caller's list
    // holds: say so through the count (the protocol of an overflowing list: "count past the capacity, grow, call again") BEFORE
    // the tiled kernel walks the whole tile for a list that cannot take its output.  The second call finds the refusal
    // remembered and runs the kernel, whatever the list holds by then.
    if (const uint64_t hint = ctx->join_dense.edges_hint; hint > cap && hint <= ((uint64_t)64 << 20)) {
      ctx->join_dense.edges_hint = 0;
      hipLaunchKernelGGL(add_count_kernel, dim3(1), dim3(1), 0, ctx->stream, (unsigned long long*)d_count, (unsigned long long)hint);
      RTC_CHECK_LAUNCH(ctx);
      ctx->pair_last_path = 2;
      ctx->diag[6]++;
      return RTC_OK;
    }
    RTC_TRY(rtc_pair_edges_tiled(ctx, d_hashes, width, d_start, d_len, n, row0, row1, col0, col1, 1, radio, d_edges, cap,
                                 d_count, &handled));
  }
  if (handled) { ctx->pair_last_path = 2; ctx->diag[1]++; return RTC_OK; }
  ctx->pair_last_path = 1;
  ctx->diag[2]++;
  const uint64_t ld = col1 - col0;
  uint32_t rows_per = (uint32_t)std::max<uint64_t>(4, std::min<uint64_t>(row1 - row0, ((uint64_t)1 << 30) / (ld * 4)));
  rows_per = std::min<uint32_t>(rows_per, 262140);
  uint32_t* d_common = nullptr;
  RTC_TRY(rtc_ws(ctx, 2, (size_t)rows_per * ld * 4, (void**)&d_common));
  for (uint32_t r0 = row0; r0 < row1; r0 += rows_per) {
    const uint32_t r1 = std::min(row1, r0 + rows_per);
    RTC_TRY(rtc_pair_common_dev(ctx, d_hashes, width, d_start, d_len, n, r0, r1, col0, col1, d_common, ld, 1, 1));
    RTC_TRY(rtc_extract_edges_dev(ctx, d_common, ld, r0, r1, col0, col1, d_len, radio < 0 ? 0x7fffffff : radio, d_edges, cap,
                                  d_count));
  }
  return RTC_OK;
}

// ---- rtc_warmup: an empty launch per sketch unit, then a toy clustering that touches every translation unit of the pair / MST / greedy phases ----
extern "C" int rtc_warmup(int device) {
  rtc_ctx* ctx = nullptr;
  RTC_TRY(rtc_ctx_create(device, &ctx));
  ctx->quiet = 1;
  int st = RTC_OK;
  void *d_h = nullptr, *d_s = nullptr, *d_l = nullptr, *d_e = nullptr, *d_c = nullptr;
  // the sketch units first: the command lines' first batch reaches them a few milliseconds from now
  (void)rtc_touch_unpack(ctx); (void)rtc_touch_sketch_minhash(ctx); (void)rtc_touch_sketch_kssd(ctx); (void)rtc_touch_sketch_minhash_packed(ctx);
  do {
    const uint32_t n = 12, s = 24;  // twelve sketches of 24 hashes, neighbours share half of them
    std::vector<uint64_t> h((size_t)n * s), start(n);
    std::vector<uint32_t> len(n, s), cfg(n, s);
    for (uint32_t g = 0; g < n; g++) {
      start[g] = (uint64_t)g * s;
      for (uint32_t e = 0; e < s; e++) h[(size_t)g * s + e] = ((uint64_t)(g / 2) * s + e + (g & 1) * (s / 2) + 1) * 0x9E3779B97F4A7C15ULL >> 12;
      std::sort(h.begin() + (size_t)g * s, h.begin() + (size_t)(g + 1) * s);
    }
#define W_HIP(call) if ((call) != hipSuccess) { st = rtc_fail(ctx, RTC_ERR_HIP, "rtc_warmup: %s", #call); break; }
    W_HIP(hipMalloc(&d_h, h.size() * 8)); W_HIP(hipMalloc(&d_s, n * 8)); W_HIP(hipMalloc(&d_l, n * 4));
    W_HIP(hipMalloc(&d_e, 4096 * sizeof(rtc_cedge))); W_HIP(hipMalloc(&d_c, 8));
    W_HIP(hipMemcpy(d_h, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    W_HIP(hipMemcpy(d_s, start.data(), n * 8, hipMemcpyHostToDevice));
    W_HIP(hipMemcpy(d_l, len.data(), n * 4, hipMemcpyHostToDevice));
    W_HIP(hipMemset(d_c, 0, 8));
#undef W_HIP
    int handled = 0;
    st = rtc_pair_edges_join(ctx, d_h, 8, (const uint64_t*)d_s, (const uint32_t*)d_l, n, 1, n, 0, n - 1, -1, (rtc_cedge*)d_e, 4096,
                             (uint64_t*)d_c, -1.0, &handled);                                 // join + rocPRIM
    if (st != RTC_OK) break;
    std::vector<rtc_edge> mst(n);
    uint64_t m = 0;
    st = rtc_mst(ctx, d_h, 8, (const uint64_t*)d_s, (const uint32_t*)d_l, n, 21, 0, 0.05, mst.data(), &m);  // tiled kernel, Boruvka, forest sort
    if (st != RTC_OK) break;
    std::vector<int32_t> rep(n);
    uint32_t ncl = 0;
    st = rtc_greedy(ctx, d_h, 8, (const uint64_t*)d_s, (const uint32_t*)d_l, n, cfg.data(), 21, 0, 0, 0.05, rep.data(), &ncl);
  } while (false);
  (void)hipFree(d_h); (void)hipFree(d_s); (void)hipFree(d_l); (void)hipFree(d_e); (void)hipFree(d_c);
  rtc_ctx_destroy(ctx);
  return st;
}

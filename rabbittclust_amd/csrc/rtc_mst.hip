// rtc_mst.hip -- candidate-edge extraction and minimum spanning forest on the GPU.
//
// Replaces the edge/Kruskal part of compute_minhash_mst / compute_kssd_mst
// (src/MST.cpp:1466-1547, :1715-1723 in the reference tree):
//   * extract_edges_kernel applies the reference's pair filters (j < i, common > 0, size-ratio
//     "radio" test, :1468-1487) to the dense common matrix and compacts survivors with a wave
//     ballot + one global atomic per wave;
//   * Boruvka rounds pick, per current component, the minimum outgoing edge.  Weights are never
//     compared as log() results on the device: the key is the IEEE-754 bit pattern of the
//     similarity double J = common/denom (correctly rounded division, monotone in the exact
//     rational), so the order is exact and identical on every rank; ties break on (i,j).
//     Distances are evaluated on the host with the reference's expression order.
#include <math.h>

#include <algorithm>
#include <numeric>

#include "rtc_internal.h"

namespace {

constexpr uint64_t KEY_NONE = 0x7FFFFFFFFFFFFFFFULL;  // fits int64 for all-reduce(MIN)

// similarity key: smaller key == more similar == smaller distance
__device__ __forceinline__ uint64_t weight_key(uint32_t common, uint32_t sa, uint32_t sb, int is_containment) {
  double denom = is_containment ? (double)(sa < sb ? sa : sb) : (double)((uint64_t)sa + sb - common);
  double J = (double)common / denom;  // in (0,1]; IEEE division, correctly rounded
  return 0x4000000000000000ULL - (uint64_t)__double_as_longlong(J);
}

__global__ __launch_bounds__(256) void extract_edges_kernel(const uint32_t* __restrict__ common, uint64_t ld,
                                                            uint32_t row0, uint32_t row1, uint32_t col0,
                                                            uint32_t col1, const uint32_t* __restrict__ len,
                                                            int radio, rtc_cedge* __restrict__ edges,
                                                            uint64_t cap, unsigned long long* __restrict__ count) {
  // each lane covers 4 consecutive columns (one 16-byte load when aligned); a block covers 1024
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t cbase = col0 + blockIdx.x * 1024;
  const uint32_t c4 = cbase + threadIdx.x * 4;
  const bool vec_ok = (ld & 3) == 0 && (((uintptr_t)common) & 15) == 0;
  for (uint32_t row = row0 + blockIdx.y; row < row1; row += gridDim.y) {
    if (cbase >= row) continue;  // whole block on/above the diagonal (uniform)
    const uint32_t* rp = common + (uint64_t)(row - row0) * ld;
    uint32_t v[4] = {0, 0, 0, 0};
    const uint32_t off = c4 - col0;
    if (c4 + 3 < col1 && vec_ok) {
      const uint4 q = *reinterpret_cast<const uint4*>(rp + off);
      v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; j++) if (c4 + j < col1) v[j] = rp[off + j];
    }
    const uint32_t s0 = len[row];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const uint32_t col = c4 + j;
      bool keep = false;
      if (col < col1 && col < row && v[j] > 0 && s0 > 0) {
        const uint32_t s1 = len[col];
        if (s1 > 0) {
          const uint32_t mn = s0 < s1 ? s0 : s1, mx = s0 > s1 ? s0 : s1;
          keep = !((uint64_t)mx > (uint64_t)(int64_t)radio * (uint64_t)mn);  // src/MST.cpp:1484
        }
      }
      const uint64_t bal = __ballot(keep);
      if (bal) {
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(count, (unsigned long long)__popcll(bal));
        base = __shfl(base, 0);
        const uint64_t idx = base + (uint64_t)__popcll(bal & ((1ULL << lane) - 1ULL));
        if (keep && idx < cap) edges[idx] = rtc_cedge{row, col, v[j]};
      }
    }
  }
}

__global__ __launch_bounds__(256) void boruvka_minweight_kernel(const rtc_cedge* __restrict__ edges, uint64_t m,
                                                                const uint32_t* __restrict__ len, int is_containment,
                                                                const uint32_t* __restrict__ comp,
                                                                unsigned long long* __restrict__ wkey) {
  for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < m; e += (uint64_t)gridDim.x * blockDim.x) {
    const rtc_cedge ed = edges[e];
    const uint32_t ci = comp[ed.i], cj = comp[ed.j];
    if (ci == cj) continue;
    const uint64_t key = weight_key(ed.common, len[ed.i], len[ed.j], is_containment);
    if (key < wkey[ci]) atomicMin(&wkey[ci], (unsigned long long)key);
    if (key < wkey[cj]) atomicMin(&wkey[cj], (unsigned long long)key);
  }
}

__global__ __launch_bounds__(256) void boruvka_minedge_kernel(const rtc_cedge* __restrict__ edges, uint64_t m,
                                                              const uint32_t* __restrict__ len, int is_containment,
                                                              const uint32_t* __restrict__ comp,
                                                              const unsigned long long* __restrict__ wkey,
                                                              unsigned long long* __restrict__ ekey) {
  for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < m; e += (uint64_t)gridDim.x * blockDim.x) {
    const rtc_cedge ed = edges[e];
    const uint32_t ci = comp[ed.i], cj = comp[ed.j];
    if (ci == cj) continue;
    const uint64_t key = weight_key(ed.common, len[ed.i], len[ed.j], is_containment);
    const unsigned long long id = ((unsigned long long)ed.i << 32) | ed.j;
    if (key == wkey[ci] && id < ekey[ci]) atomicMin(&ekey[ci], id);
    if (key == wkey[cj] && id < ekey[cj]) atomicMin(&ekey[cj], id);
  }
}

__global__ __launch_bounds__(256) void boruvka_fetch_kernel(const rtc_cedge* __restrict__ edges, uint64_t m,
                                                            const uint32_t* __restrict__ comp,
                                                            const unsigned long long* __restrict__ ekey,
                                                            uint32_t* __restrict__ ecommon) {
  for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < m; e += (uint64_t)gridDim.x * blockDim.x) {
    const rtc_cedge ed = edges[e];
    const uint32_t ci = comp[ed.i], cj = comp[ed.j];
    if (ci == cj) continue;
    const unsigned long long id = ((unsigned long long)ed.i << 32) | ed.j;
    if (ekey[ci] == id) ecommon[ci] = ed.common;
    if (ekey[cj] == id) ecommon[cj] = ed.common;
  }
}

__global__ void fill_u64_kernel(unsigned long long* p, uint64_t n, unsigned long long v) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) p[i] = v;
}

inline uint32_t grid_for(uint64_t work, int num_cu) {
  uint64_t b = (work + 255) / 256;
  uint64_t mx = (uint64_t)num_cu * 8;
  return (uint32_t)std::max<uint64_t>(1, std::min(b, mx));
}

struct HostUF {
  std::vector<uint32_t> p;
  explicit HostUF(uint32_t n) : p(n) { std::iota(p.begin(), p.end(), 0u); }
  uint32_t find(uint32_t x) {
    uint32_t r = x;
    while (p[r] != r) r = p[r];
    while (p[x] != r) { uint32_t nx = p[x]; p[x] = r; x = nx; }
    return r;
  }
};

}  // namespace

// src/MST.cpp:1295,1489-1515 with the reference's operation order (host libm log)
static double host_mst_distance(int common, int size0, int size1, int kmer_size, int is_containment) {
  const double inv_kmer_size = 1.0 / kmer_size;
  if (!is_containment) {
    int denom = size0 + size1 - common;
    double jaccard = denom == 0 ? 0.0 : (double)common / denom;
    if (jaccard == 1.0) return 0.0;
    if (jaccard == 0.0) return 1.0;
    double ratio = (2.0 * jaccard) / (1.0 + jaccard);
    return -inv_kmer_size * log(ratio);
  }
  int denom = size0 < size1 ? size0 : size1;
  double containment = denom == 0 ? 0.0 : (double)common / denom;
  if (containment == 1.0) return 0.0;
  if (containment == 0.0) return 1.0;
  return -inv_kmer_size * log(containment);
}

extern "C" {

int rtc_extract_edges_dev(rtc_ctx* ctx, const uint32_t* d_common, uint64_t ld, uint32_t row0, uint32_t row1,
                          uint32_t col0, uint32_t col1, const uint32_t* d_len, int radio, rtc_cedge* d_edges,
                          uint64_t cap, uint64_t* d_count) {
  if (!ctx || !d_common || !d_len || !d_count || (cap && !d_edges)) return RTC_ERR_ARG;
  if (row0 >= row1 || col0 >= col1) return RTC_OK;
  RTC_HIP(ctx, hipSetDevice(ctx->device));
  dim3 grid((col1 - col0 + 1023) / 1024, std::min<uint32_t>(row1 - row0, 4096));
  hipLaunchKernelGGL(extract_edges_kernel, grid, dim3(256), 0, ctx->stream, d_common, ld, row0, row1, col0, col1,
                     d_len, radio, d_edges, cap, (unsigned long long*)d_count);
  RTC_CHECK_LAUNCH(ctx);
  return RTC_OK;
}

int rtc_boruvka_minweight_dev(rtc_ctx* ctx, const rtc_cedge* d_edges, uint64_t m, const uint32_t* d_len,
                              int is_containment, const uint32_t* d_comp, uint32_t n, uint64_t* d_wkey) {
  if (!ctx || !d_len || !d_comp || !d_wkey || (m && !d_edges)) return RTC_ERR_ARG;
  RTC_HIP(ctx, hipSetDevice(ctx->device));
  hipLaunchKernelGGL(fill_u64_kernel, dim3(grid_for(n, ctx->num_cu)), dim3(256), 0, ctx->stream,
                     (unsigned long long*)d_wkey, (uint64_t)n, (unsigned long long)KEY_NONE);
  RTC_CHECK_LAUNCH(ctx);
  if (m) {
    hipLaunchKernelGGL(boruvka_minweight_kernel, dim3(grid_for(m, ctx->num_cu)), dim3(256), 0, ctx->stream, d_edges, m,
                       d_len, is_containment, d_comp, (unsigned long long*)d_wkey);
    RTC_CHECK_LAUNCH(ctx);
  }
  return RTC_OK;
}

int rtc_boruvka_minedge_dev(rtc_ctx* ctx, const rtc_cedge* d_edges, uint64_t m, const uint32_t* d_len,
                            int is_containment, const uint32_t* d_comp, uint32_t n, const uint64_t* d_wkey,
                            uint64_t* d_ekey) {
  if (!ctx || !d_len || !d_comp || !d_wkey || !d_ekey || (m && !d_edges)) return RTC_ERR_ARG;
  RTC_HIP(ctx, hipSetDevice(ctx->device));
  hipLaunchKernelGGL(fill_u64_kernel, dim3(grid_for(n, ctx->num_cu)), dim3(256), 0, ctx->stream,
                     (unsigned long long*)d_ekey, (uint64_t)n, (unsigned long long)KEY_NONE);
  RTC_CHECK_LAUNCH(ctx);
  if (m) {
    hipLaunchKernelGGL(boruvka_minedge_kernel, dim3(grid_for(m, ctx->num_cu)), dim3(256), 0, ctx->stream, d_edges, m,
                       d_len, is_containment, d_comp, (const unsigned long long*)d_wkey, (unsigned long long*)d_ekey);
    RTC_CHECK_LAUNCH(ctx);
  }
  return RTC_OK;
}

int rtc_boruvka_fetch_dev(rtc_ctx* ctx, const rtc_cedge* d_edges, uint64_t m, const uint32_t* d_comp, uint32_t n,
                          const uint64_t* d_ekey, uint32_t* d_ecommon) {
  if (!ctx || !d_comp || !d_ekey || !d_ecommon || (m && !d_edges)) return RTC_ERR_ARG;
  RTC_HIP(ctx, hipSetDevice(ctx->device));
  RTC_HIP(ctx, hipMemsetAsync(d_ecommon, 0, (size_t)n * 4, ctx->stream));
  if (m) {
    hipLaunchKernelGGL(boruvka_fetch_kernel, dim3(grid_for(m, ctx->num_cu)), dim3(256), 0, ctx->stream, d_edges, m,
                       d_comp, (const unsigned long long*)d_ekey, d_ecommon);
    RTC_CHECK_LAUNCH(ctx);
  }
  return RTC_OK;
}

int rtc_edges_to_mst_host(const rtc_cedge* h_sel, uint64_t m, const uint32_t* h_len, int kmer_size,
                          int is_containment, rtc_edge* h_out) {
  if ((m && (!h_sel || !h_out)) || !h_len) return RTC_ERR_ARG;
  for (uint64_t e = 0; e < m; e++) {
    h_out[e].preNode = (int32_t)h_sel[e].i;
    h_out[e].sufNode = (int32_t)h_sel[e].j;
    h_out[e].dist = host_mst_distance((int)h_sel[e].common, (int)h_len[h_sel[e].i], (int)h_len[h_sel[e].j],
                                      kmer_size, is_containment);
  }
  std::sort(h_out, h_out + m, [](const rtc_edge& a, const rtc_edge& b) {
    if (a.dist != b.dist) return a.dist < b.dist;
    if (a.preNode != b.preNode) return a.preNode < b.preNode;
    return a.sufNode < b.sufNode;
  });
  return RTC_OK;
}

int rtc_boruvka_merge_host(uint32_t n, const uint64_t* h_ekey, const uint32_t* h_ecommon, uint32_t* h_comp,
                           rtc_cedge* h_sel, uint64_t* h_n_sel, uint64_t* h_added) {
  if (!h_ekey || !h_ecommon || !h_comp || !h_sel || !h_n_sel || !h_added) return RTC_ERR_ARG;
  HostUF uf(n);
  for (uint32_t v = 0; v < n; v++) uf.p[v] = h_comp[v];  // labels are root vertex ids
  uint64_t added = 0, ns = *h_n_sel;
  for (uint32_t c = 0; c < n; c++) {
    const uint64_t id = h_ekey[c];
    if (id == KEY_NONE) continue;
    const uint32_t i = (uint32_t)(id >> 32), j = (uint32_t)id;
    if (i >= n || j >= n) return RTC_ERR_ARG;
    const uint32_t a = uf.find(i), b = uf.find(j);
    if (a == b) continue;  // the partner component picked the same edge
    uf.p[a < b ? b : a] = a < b ? a : b;
    h_sel[ns++] = rtc_cedge{i, j, h_ecommon[c]};
    added++;
  }
  for (uint32_t v = 0; v < n; v++) h_comp[v] = uf.find(v);
  *h_n_sel = ns;
  *h_added = added;
  return RTC_OK;
}

int rtc_mst(rtc_ctx* ctx, const void* d_hashes, int width, const uint64_t* d_start, const uint32_t* d_len,
            uint32_t n, int kmer_size, int is_containment, double threshold, rtc_edge* h_edges_out,
            uint64_t* h_n_edges) {
  if (!ctx || !h_n_edges || (n && (!d_start || !d_len || !h_edges_out))) return RTC_ERR_ARG;
  *h_n_edges = 0;
  if (n < 2) return RTC_OK;
  RTC_HIP(ctx, hipSetDevice(ctx->device));
  const int radio = (int)(2.0 * exp(threshold * (kmer_size - 1)) - 1.0);  // src/MST.cpp:26-37,1292

  // ---- all-pairs in row chunks -> compacted candidate edges ----
  const uint64_t budget = 2ull << 30;  // bytes of dense common matrix resident at a time
  uint32_t rows_per = (uint32_t)std::max<uint64_t>(64, std::min<uint64_t>(n, budget / ((uint64_t)n * 4)));
  rows_per = ((rows_per + 63) / 64) * 64;
  uint32_t* d_common = nullptr;
  RTC_TRY(rtc_ws(ctx, 2, (size_t)rows_per * n * 4, (void**)&d_common));
  uint64_t cap = std::max<uint64_t>(1u << 20, (uint64_t)n * 16);
  rtc_cedge* d_edges = nullptr;
  unsigned long long* d_count = nullptr;
  RTC_HIP(ctx, hipMalloc(&d_edges, cap * sizeof(rtc_cedge)));
  hipError_t e0 = hipMalloc(&d_count, 8);
  if (e0 != hipSuccess) { (void)hipFree(d_edges); return rtc_fail(ctx, RTC_ERR_NOMEM, "hipMalloc count"); }
  int st = RTC_OK;
  uint64_t m = 0;
  auto cleanup = [&]() { (void)hipFree(d_edges); (void)hipFree(d_count); };
#define MST_TRY(x) do { st = (x); if (st != RTC_OK) { cleanup(); return st; } } while (0)
#define MST_HIP(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { cleanup(); return rtc_fail(ctx, RTC_ERR_HIP, "%s -> %s", #x, hipGetErrorString(e__)); } } while (0)
  MST_HIP(hipMemsetAsync(d_count, 0, 8, ctx->stream));
  for (uint32_t r0 = 1; r0 < n; r0 += rows_per) {
    const uint32_t r1 = std::min<uint32_t>(n, r0 + rows_per);
    const uint32_t c1 = r1 - 1;  // columns < max row
    MST_TRY(rtc_pair_common_dev(ctx, d_hashes, width, d_start, d_len, n, r0, r1, 0, c1, d_common, n, 1, 0));
    while (true) {
      MST_TRY(rtc_extract_edges_dev(ctx, d_common, n, r0, r1, 0, c1, d_len, radio, d_edges, cap, (uint64_t*)d_count));
      unsigned long long cnt = 0;
      MST_HIP(hipMemcpyAsync(&cnt, d_count, 8, hipMemcpyDeviceToHost, ctx->stream));
      MST_HIP(hipStreamSynchronize(ctx->stream));
      if (cnt <= cap) { m = cnt; break; }
      // grow and redo this chunk's extraction from the previous fill level
      uint64_t ncap = std::max<uint64_t>(cnt + cnt / 2, cap * 2);
      rtc_cedge* nd = nullptr;
      MST_HIP(hipMalloc(&nd, ncap * sizeof(rtc_cedge)));
      MST_HIP(hipMemcpyAsync(nd, d_edges, m * sizeof(rtc_cedge), hipMemcpyDeviceToDevice, ctx->stream));
      MST_HIP(hipStreamSynchronize(ctx->stream));
      (void)hipFree(d_edges);
      d_edges = nd; cap = ncap;
      unsigned long long mm = m;
      MST_HIP(hipMemcpyAsync(d_count, &mm, 8, hipMemcpyHostToDevice, ctx->stream));
      MST_HIP(hipStreamSynchronize(ctx->stream));
    }
  }

  // ---- Boruvka rounds ----
  std::vector<uint32_t> h_len(n), h_comp(n);
  MST_HIP(hipMemcpyAsync(h_len.data(), d_len, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
  void* ws3 = nullptr;
  MST_TRY(rtc_ws(ctx, 3, (size_t)n * (8 + 8 + 4 + 4) + 64, &ws3));
  uint64_t* d_wkey = (uint64_t*)ws3;
  uint64_t* d_ekey = d_wkey + n;
  uint32_t* d_ecommon = (uint32_t*)(d_ekey + n);
  uint32_t* d_comp = d_ecommon + n;
  std::vector<uint64_t> h_ekey(n);
  std::vector<uint32_t> h_ecommon(n);
  std::vector<rtc_cedge> sel(n);
  uint64_t nsel = 0;
  std::iota(h_comp.begin(), h_comp.end(), 0u);
  for (int round = 0; round < 64 && m > 0; round++) {
    MST_HIP(hipMemcpyAsync(d_comp, h_comp.data(), (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    MST_TRY(rtc_boruvka_minweight_dev(ctx, d_edges, m, d_len, is_containment, d_comp, n, d_wkey));
    MST_TRY(rtc_boruvka_minedge_dev(ctx, d_edges, m, d_len, is_containment, d_comp, n, d_wkey, d_ekey));
    MST_TRY(rtc_boruvka_fetch_dev(ctx, d_edges, m, d_comp, n, d_ekey, d_ecommon));
    MST_HIP(hipMemcpyAsync(h_ekey.data(), d_ekey, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    MST_HIP(hipMemcpyAsync(h_ecommon.data(), d_ecommon, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    MST_HIP(hipStreamSynchronize(ctx->stream));
    uint64_t added = 0;
    MST_TRY(rtc_boruvka_merge_host(n, h_ekey.data(), h_ecommon.data(), h_comp.data(), sel.data(), &nsel, &added));
    if (!added) break;
  }
  MST_HIP(hipStreamSynchronize(ctx->stream));
  cleanup();
#undef MST_TRY
#undef MST_HIP
  RTC_TRY(rtc_edges_to_mst_host(sel.data(), nsel, h_len.data(), kmer_size, is_containment, h_edges_out));
  *h_n_edges = nsel;
  return RTC_OK;
}

}  // extern "C"

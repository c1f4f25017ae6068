// rtc_sort.hip -- the selected forest in the reference's output order, sorted on the device.
//
// The reference sorts its MST edges by (dist, preNode, sufNode) on the host (src/MST.cpp:1721: std::sort before the
// final Kruskal; the forest is written in that order).  dist is a monotone function of the similarity double
// J = common / denom, so the order by (weight key of J, i, j) is the same order: two stable radix sorts of <= n - 1
// records (rocPRIM) replace a host std::sort that was the larger half of the MST phase.  The host still evaluates
// the distances with its own libm (rtc_edges_to_mst_host) and checks the order it receives -- anything out of
// order there (never seen) simply falls back to its own sort.
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "rtc_internal.h"

namespace {

__host__ __device__ __forceinline__ uint64_t sort_weight_denom(uint32_t common, uint32_t sa, uint32_t sb, int wmode) {
  if ((wmode & 3) == 1) return sa < sb ? sa : sb;
  const uint64_t u = (uint64_t)sa + sb - common;
  if ((wmode & 3) == 2) { const uint64_t s = (uint32_t)wmode >> 2; return u < s ? u : s; }
  return u;
}

__global__ __launch_bounds__(256) void forest_ids_kernel(const rtc_cedge* __restrict__ sel, uint32_t ns, uint64_t* __restrict__ id,
                                                         uint32_t* __restrict__ idx) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= ns) return;
  id[e] = ((uint64_t)sel[e].i << 32) | sel[e].j;
  idx[e] = e;
}
// smaller key = more similar = smaller distance (the key of rtc_mst.hip's weight_key)
__global__ __launch_bounds__(256) void forest_weights_kernel(const rtc_cedge* __restrict__ sel, const uint32_t* __restrict__ idx,
                                                             uint32_t ns, const uint32_t* __restrict__ len, int wmode,
                                                             uint64_t* __restrict__ w) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ns) return;
  const rtc_cedge ed = sel[idx[t]];
  const uint64_t d = sort_weight_denom(ed.common, len[ed.i], len[ed.j], wmode);
  const double J = d ? (double)ed.common / (double)d : 0.0;
  w[t] = 0x4000000000000000ULL - (uint64_t)__double_as_longlong(J);
}
__global__ __launch_bounds__(256) void forest_gather_kernel(const rtc_cedge* __restrict__ sel, const uint32_t* __restrict__ idx,
                                                            uint32_t ns, rtc_cedge* __restrict__ out) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < ns) out[t] = sel[idx[t]];
}

inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

// d_sel[0 .. ns) -> the same records ordered by (weight key, i, j).  Scratch: slot 5 of the context.
int rtc_sort_forest_device(rtc_ctx* ctx, rtc_cedge* d_sel, uint64_t ns64, const uint32_t* d_len, int wmode) {
  if (ns64 < 2 || ns64 >= (1ull << 31)) return RTC_OK;
  const uint32_t ns = (uint32_t)ns64;
  hipStream_t s = ctx->stream;
  size_t tb = 0;
  RTC_HIP(ctx, rocprim::radix_sort_pairs(nullptr, tb, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const uint32_t*)nullptr,
                                         (uint32_t*)nullptr, (size_t)ns, 0u, 64u, s));
  const size_t b_k = up256((size_t)ns * 8), b_i = up256((size_t)ns * 4), b_e = up256((size_t)ns * sizeof(rtc_cedge));
  void* ws = nullptr;
  RTC_TRY(rtc_ws(ctx, 5, 2 * b_k + 2 * b_i + b_e + up256(tb) + 256, &ws));
  uint64_t* k0 = (uint64_t*)ws;
  uint64_t* k1 = (uint64_t*)((char*)ws + b_k);
  uint32_t* i0 = (uint32_t*)((char*)ws + 2 * b_k);
  uint32_t* i1 = (uint32_t*)((char*)ws + 2 * b_k + b_i);
  rtc_cedge* out = (rtc_cedge*)((char*)ws + 2 * b_k + 2 * b_i);
  void* tmp = (char*)ws + 2 * b_k + 2 * b_i + b_e;
  const dim3 g((ns + 255) / 256), b(256);
  hipLaunchKernelGGL(forest_ids_kernel, g, b, 0, s, (const rtc_cedge*)d_sel, ns, k0, i0);
  RTC_CHECK_LAUNCH(ctx);
  RTC_HIP(ctx, rocprim::radix_sort_pairs(tmp, tb, (const uint64_t*)k0, k1, (const uint32_t*)i0, i1, (size_t)ns, 0u, 64u, s));
  hipLaunchKernelGGL(forest_weights_kernel, g, b, 0, s, (const rtc_cedge*)d_sel, (const uint32_t*)i1, ns, d_len, wmode, k0);
  RTC_CHECK_LAUNCH(ctx);
  RTC_HIP(ctx, rocprim::radix_sort_pairs(tmp, tb, (const uint64_t*)k0, k1, (const uint32_t*)i1, i0, (size_t)ns, 0u, 64u, s));
  hipLaunchKernelGGL(forest_gather_kernel, g, b, 0, s, (const rtc_cedge*)d_sel, (const uint32_t*)i0, ns, out);
  RTC_CHECK_LAUNCH(ctx);
  RTC_HIP(ctx, hipMemcpyAsync(d_sel, out, (size_t)ns * sizeof(rtc_cedge), hipMemcpyDeviceToDevice, s));
  return RTC_OK;
}

// rtc_internal.h -- shared host-side plumbing for the HIP translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/rtclust.h"

struct rtc_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  int num_cu = 256;
  int lds_per_wg = 65536;
  std::string err;
  // growable device scratch (segment tables, partial sketches, partition tables ...)
  void* ws[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t ws_bytes[6] = {0, 0, 0, 0, 0, 0};
  // pinned host staging for small synchronous read-backs
  void* pinned = nullptr;
  size_t pinned_bytes = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

int rtc_fail(rtc_ctx* ctx, int code, const char* fmt, ...);
// returns device scratch slot `slot` grown to at least `bytes`
int rtc_ws(rtc_ctx* ctx, int slot, size_t bytes, void** out);
int rtc_pinned(rtc_ctx* ctx, size_t bytes, void** out);

#define RTC_HIP(ctx, call)                                                                  \
  do {                                                                                      \
    hipError_t e__ = (call);                                                                \
    if (e__ != hipSuccess)                                                                  \
      return rtc_fail((ctx), RTC_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #call,      \
                      hipGetErrorString(e__));                                              \
  } while (0)

#define RTC_TRY(call)                  \
  do {                                 \
    int s__ = (call);                  \
    if (s__ != RTC_OK) return s__;     \
  } while (0)

// RTC_DEBUG_SYNC=1 in the environment synchronises after every launch and names it on stderr, so
// an asynchronous device fault can be pinned on a kernel
#define RTC_CHECK_LAUNCH(ctx)                                                        \
  do {                                                                               \
    RTC_HIP(ctx, hipGetLastError());                                                 \
    if (rtc_debug_sync()) {                                                          \
      fprintf(stderr, "[rtc] %s:%d launched, syncing\n", __FILE__, __LINE__);        \
      RTC_HIP(ctx, hipStreamSynchronize(ctx->stream));                               \
      fprintf(stderr, "[rtc] %s:%d ok\n", __FILE__, __LINE__);                       \
    }                                                                                \
  } while (0)
static inline bool rtc_debug_sync() {
  static const bool on = getenv("RTC_DEBUG_SYNC") != nullptr;
  return on;
}

// ---- device helpers shared by kernels ------------------------------------------------------
__device__ __forceinline__ uint64_t rtc_rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
__device__ __forceinline__ uint32_t rtc_lane() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

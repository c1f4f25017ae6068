// rtc_ctx.hip -- context, error reporting, device memory helpers, event timer.
#include <stdarg.h>
#include <string.h>

#include <time.h>

#include "rtc_internal.h"

static thread_local std::string g_err;  // failures that happen without a context

int rtc_fail(rtc_ctx* ctx, int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf;
  else g_err = buf;
  return code;
}

int rtc_ws(rtc_ctx* ctx, int slot, size_t bytes, void** out) {
  if (slot < 0 || slot >= 6) return rtc_fail(ctx, RTC_ERR_ARG, "bad scratch slot %d", slot);
  if (bytes > ctx->ws_bytes[slot]) {
    RTC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->ws[slot]) RTC_HIP(ctx, hipFree(ctx->ws[slot]));
    ctx->ws[slot] = nullptr;
    ctx->ws_bytes[slot] = 0;
    size_t want = bytes + bytes / 4 + 4096;
    ctx->free_hbm_at = -1.0;
    hipError_t e = hipMalloc(&ctx->ws[slot], want);
    if (e != hipSuccess) return rtc_fail(ctx, RTC_ERR_NOMEM, "hipMalloc(%zu) scratch: %s", want, hipGetErrorString(e));
    ctx->ws_bytes[slot] = want;
  }
  *out = ctx->ws[slot];
  return RTC_OK;
}

uint64_t rtc_free_hbm(rtc_ctx* ctx) {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  const double now = (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
  if (ctx->free_hbm_at < 0 || now - ctx->free_hbm_at > 0.1) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = (size_t)8 << 30; }
    ctx->free_hbm_cached = free_b;
    ctx->free_hbm_at = now;
  }
  return ctx->free_hbm_cached;
}

int rtc_pinned(rtc_ctx* ctx, size_t bytes, void** out) {
  if (bytes > ctx->pinned_bytes) {
    if (ctx->pinned) RTC_HIP(ctx, hipHostFree(ctx->pinned));
    ctx->pinned = nullptr;
    ctx->pinned_bytes = 0;
    size_t want = bytes + 4096;
    RTC_HIP(ctx, hipHostMalloc(&ctx->pinned, want, hipHostMallocDefault));
    ctx->pinned_bytes = want;
  }
  *out = ctx->pinned;
  return RTC_OK;
}

extern "C" {

const char* rtc_version(void) { return "rabbittclust_amd 0.1 (gfx950)"; }

int rtc_device_count(void) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess) return 0;
  return ndev;
}

}  // extern "C"

void rtc_options_from_env(rtc_options* o) {
  *o = rtc_options();
  auto num = [](const char* name, long long dflt) -> long long { const char* e = getenv(name); return e ? atoll(e) : dflt; };
  auto flag = [](const char* name) -> int { const char* e = getenv(name); return e != nullptr && strcmp(e, "0") != 0 && *e != 0; };  // set, and not to "0" / ""
  o->verbose = flag("RTC_VERBOSE");
  o->pair_join = (int)num("RTC_PAIR_JOIN", 1);
  o->join_semi = (int)num("RTC_JOIN_SEMI", 1);
  o->join_fullsort = flag("RTC_JOIN_FULLSORT");
  o->join_debug = flag("RTC_JOIN_DEBUG");
  o->pair_force_merge = flag("RTC_PAIR_FORCE_MERGE");
  o->pair_ktarget = (uint32_t)std::max<long long>(0, num("RTC_PAIR_KTARGET", 0));
  if (const char* e = getenv("RTC_PAIR_TCOLS_BUDGET")) o->pair_tcols_budget = std::max<uint64_t>(strtoull(e, nullptr, 10), 1);
  if (const char* e = getenv("RTC_EDGE_BUDGET")) o->edge_budget = std::max<uint64_t>(strtoull(e, nullptr, 10), 1024);
  if (const char* e = getenv("RTC_GREEDY_GLOBAL_PAIRS")) { o->has_greedy_global_pairs = true; o->greedy_global_pairs = strtoull(e, nullptr, 10); }
  o->kssd_cuckoo = flag("RTC_KSSD_CUCKOO");
  o->kssd_nofast = (int)num("RTC_KSSD_NOFAST", 0);
  o->sketch_packed = flag("RTC_SKETCH_PACKED");
  o->sketch_no_packed = flag("RTC_SKETCH_NO_PACKED");
  o->sketch_rounds = (int)std::max<long long>(0, num("RTC_SKETCH_ROUNDS", 0));
  if (getenv("RTC_SKETCH_ROUNDS") && o->sketch_rounds < 1) o->sketch_rounds = 1;
  o->sketch_t0_factor = getenv("RTC_SKETCH_T0_FACTOR") ? (int)std::max<long long>(0, num("RTC_SKETCH_T0_FACTOR", 0)) : -1;
  o->comm_force_rccl = flag("RTC_COMM_FORCE_RCCL");
  if (const char* e = getenv("RTC_COMM_TIMEOUT_S")) o->comm_timeout_s = atof(e);
}

extern "C" {

int rtc_ctx_reload_options(rtc_ctx* ctx) {
  if (!ctx) return RTC_ERR_ARG;
  rtc_options_from_env(&ctx->opt);
  return RTC_OK;
}

int rtc_ctx_create(int device, rtc_ctx** out) {
  if (!out) return RTC_ERR_ARG;
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return rtc_fail(nullptr, RTC_ERR_HIP, "hipGetDeviceCount -> %s (%d devices)", hipGetErrorString(e), ndev);
  if (device < 0 || device >= ndev) return rtc_fail(nullptr, RTC_ERR_ARG, "device %d of %d", device, ndev);
  e = hipSetDevice(device);
  if (e != hipSuccess) return rtc_fail(nullptr, RTC_ERR_HIP, "hipSetDevice(%d) -> %s", device, hipGetErrorString(e));
  rtc_ctx* ctx = new rtc_ctx();
  ctx->device = device;
  rtc_options_from_env(&ctx->opt);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
    ctx->num_cu = prop.multiProcessorCount;
    ctx->lds_per_wg = (int)prop.sharedMemPerBlock;
  }
  if (hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) {
    delete ctx;
    return rtc_fail(nullptr, RTC_ERR_HIP, "hipEventCreate failed");
  }
  *out = ctx;
  return RTC_OK;
}

void rtc_ctx_destroy(rtc_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (int i = 0; i < 6; i++)
    if (ctx->ws[i]) (void)hipFree(ctx->ws[i]);
  if (ctx->edge_cache) (void)hipFree(ctx->edge_cache);
  if (ctx->edge_cache_count) (void)hipFree(ctx->edge_cache_count);
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
  if (ctx->mst_pinned) (void)hipHostFree(ctx->mst_pinned);
  if (ctx->sticky) (void)hipHostFree(ctx->sticky);
  if (ctx->kssd.d_index) (void)hipFree(ctx->kssd.d_index);
  if (ctx->kssd.d_table) (void)hipFree(ctx->kssd.d_table);
  if (ctx->kssd.d_bucket) (void)hipFree(ctx->kssd.d_bucket);
  if (ctx->kssd.d_bloom) (void)hipFree(ctx->kssd.d_bloom);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->pk0) (void)hipEventDestroy(ctx->pk0);
  if (ctx->pk1) (void)hipEventDestroy(ctx->pk1);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  if (ctx->owned_stream) (void)hipStreamDestroy(ctx->owned_stream);
  delete ctx;
}

int rtc_ctx_own_stream(rtc_ctx* ctx) {
  if (!ctx) return RTC_ERR_ARG;
  if (ctx->owned_stream) return RTC_OK;
  RTC_HIP(ctx, hipSetDevice(ctx->device));
  RTC_HIP(ctx, hipStreamCreateWithFlags(&ctx->owned_stream, hipStreamNonBlocking));
  ctx->stream = ctx->owned_stream;
  return RTC_OK;
}

int rtc_ctx_set_stream(rtc_ctx* ctx, void* hip_stream) {
  if (!ctx) return RTC_ERR_ARG;
  ctx->stream = (hipStream_t)hip_stream;
  return RTC_OK;
}

int rtc_ctx_sync(rtc_ctx* ctx) {
  if (!ctx) return RTC_ERR_ARG;
  RTC_HIP(ctx, hipSetDevice(ctx->device));  // the current device is per host thread; a NULL stream means ITS default stream
  RTC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return rtc_sticky_error(ctx);  // what an asynchronous argument check found meanwhile
}

const char* rtc_last_error(const rtc_ctx* ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

int rtc_device_info(rtc_ctx* ctx, int out[3]) {
  if (!ctx || !out) return RTC_ERR_ARG;
  out[0] = ctx->num_cu;
  out[1] = ctx->lds_per_wg;
  out[2] = 64;
  return RTC_OK;
}

int rtc_dev_alloc(rtc_ctx* ctx, size_t bytes, void** d_ptr) {
  if (ctx) ctx->free_hbm_at = -1.0;
  if (!ctx || !d_ptr) return RTC_ERR_ARG;
  *d_ptr = nullptr;
  if (bytes == 0) bytes = 16;
  RTC_HIP(ctx, hipSetDevice(ctx->device));  // the current device is per host thread
  hipError_t e = hipMalloc(d_ptr, bytes);
  if (e != hipSuccess) return rtc_fail(ctx, RTC_ERR_NOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
  return RTC_OK;
}

int rtc_dev_free(rtc_ctx* ctx, void* d_ptr) {
  if (ctx) ctx->free_hbm_at = -1.0;
  if (!ctx) return RTC_ERR_ARG;
  if (d_ptr) RTC_HIP(ctx, hipFree(d_ptr));
  return RTC_OK;
}

int rtc_dev_mem_info(rtc_ctx* ctx, size_t* free_bytes, size_t* total_bytes) {
  if (!ctx || !free_bytes || !total_bytes) return RTC_ERR_ARG;
  RTC_HIP(ctx, hipSetDevice(ctx->device));
  RTC_HIP(ctx, hipMemGetInfo(free_bytes, total_bytes));
  return RTC_OK;
}

int rtc_host_alloc(rtc_ctx* ctx, size_t bytes, void** h_ptr) {
  if (!ctx || !h_ptr) return RTC_ERR_ARG;
  *h_ptr = nullptr;
  if (bytes == 0) bytes = 16;
  RTC_HIP(ctx, hipSetDevice(ctx->device));
  hipError_t e = hipHostMalloc(h_ptr, bytes, hipHostMallocDefault);
  if (e != hipSuccess) return rtc_fail(ctx, RTC_ERR_NOMEM, "hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e));
  return RTC_OK;
}

int rtc_host_free(rtc_ctx* ctx, void* h_ptr) {
  if (!ctx) return RTC_ERR_ARG;
  if (h_ptr) RTC_HIP(ctx, hipHostFree(h_ptr));
  return RTC_OK;
}

int rtc_copy_h2d(rtc_ctx* ctx, void* d_dst, const void* h_src, size_t bytes) {
  if (!ctx || (bytes && (!d_dst || !h_src))) return RTC_ERR_ARG;
  if (!bytes) return RTC_OK;
  RTC_HIP(ctx, hipSetDevice(ctx->device));
  RTC_HIP(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
  RTC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return RTC_OK;
}

int rtc_copy_d2h(rtc_ctx* ctx, void* h_dst, const void* d_src, size_t bytes) {
  if (!ctx || (bytes && (!h_dst || !d_src))) return RTC_ERR_ARG;
  if (!bytes) return RTC_OK;
  RTC_HIP(ctx, hipSetDevice(ctx->device));
  RTC_HIP(ctx, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  RTC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return RTC_OK;
}

int rtc_memset_dev(rtc_ctx* ctx, void* d_ptr, int value, size_t bytes) {
  if (!ctx || (bytes && !d_ptr)) return RTC_ERR_ARG;
  if (!bytes) return RTC_OK;
  RTC_HIP(ctx, hipSetDevice(ctx->device));
  RTC_HIP(ctx, hipMemsetAsync(d_ptr, value, bytes, ctx->stream));
  return RTC_OK;
}

int rtc_timer_start(rtc_ctx* ctx) {
  if (!ctx) return RTC_ERR_ARG;
  RTC_HIP(ctx, hipSetDevice(ctx->device));
  RTC_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  return RTC_OK;
}

int rtc_timer_stop(rtc_ctx* ctx, float* ms_out) {
  if (!ctx || !ms_out) return RTC_ERR_ARG;
  RTC_HIP(ctx, hipSetDevice(ctx->device));
  RTC_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  RTC_HIP(ctx, hipEventSynchronize(ctx->ev1));
  RTC_HIP(ctx, hipEventElapsedTime(ms_out, ctx->ev0, ctx->ev1));
  return RTC_OK;
}

}  // extern "C"

// measurement hook (not part of include/rtclust.h): the HIP runtime's own teardown of a device, timed by the command lines'
// RTC_EXIT_PROBE to tell what a process leaves to the kernel at _exit
extern "C" int rtc_debug_device_reset(int device) {
  if (hipSetDevice(device) != hipSuccess) return RTC_ERR_HIP;
  return hipDeviceReset() == hipSuccess ? RTC_OK : RTC_ERR_HIP;
}

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

// The library's switches (README: environment).  Read from the environment ONCE, when the context is created -- no call path
// asks the environment again; rtc_ctx_reload_options reads them anew for a context that is already there (tests, tuning runs).
struct rtc_options {
  int verbose = 0;                 // RTC_VERBOSE
  int pair_join = 1;               // RTC_PAIR_JOIN: 0 never, 1 by the cost rule, 2 wherever its scratch fits
  int join_semi = 1;               // RTC_JOIN_SEMI: 0 never, 1 by rule, 2 always
  int join_fullsort = 0;           // RTC_JOIN_FULLSORT
  int join_debug = 0;              // RTC_JOIN_DEBUG
  int pair_force_merge = 0;        // RTC_PAIR_FORCE_MERGE
  uint32_t pair_ktarget = 0;       // RTC_PAIR_KTARGET (0: the kernel's default)
  uint64_t pair_tcols_budget = 0;  // RTC_PAIR_TCOLS_BUDGET (0: default)
  uint64_t edge_budget = 0;        // RTC_EDGE_BUDGET (0: default)
  bool has_greedy_global_pairs = false;
  uint64_t greedy_global_pairs = 0;  // RTC_GREEDY_GLOBAL_PAIRS
  int kssd_cuckoo = 0;             // RTC_KSSD_CUCKOO
  int kssd_nofast = 0;             // RTC_KSSD_NOFAST
  int sketch_packed = 0, sketch_no_packed = 0;  // RTC_SKETCH_PACKED / RTC_SKETCH_NO_PACKED (LDS table layout of the MinHash kernels)
  int sketch_rounds = 0;           // RTC_SKETCH_ROUNDS (0: planned)
  int sketch_t0_factor = -1;       // RTC_SKETCH_T0_FACTOR (-1: by rule)
  int comm_force_rccl = 0;         // RTC_COMM_FORCE_RCCL
  double comm_timeout_s = 120.0;   // RTC_COMM_TIMEOUT_S (<= 0: forever)
};
void rtc_options_from_env(rtc_options* o);

struct rtc_ctx {
  int device = 0;
  rtc_options opt;
  hipStream_t stream = nullptr;
  hipStream_t owned_stream = nullptr;  // rtc_ctx_own_stream
  int num_cu = 256;
  int lds_per_wg = 65536;
  std::string err;
  // growable device scratch (segment tables, partial sketches, partition tables ...)
  void* ws[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t ws_bytes[6] = {0, 0, 0, 0, 0, 0};
  // pinned host staging for small synchronous read-backs
  void* pinned = nullptr;
  size_t pinned_bytes = 0;
  // a word of page-locked host memory the device can write: asynchronous argument checks (rtc_check_runs_async) raise it, the
  // next packed sketch call and rtc_ctx_sync report it
  uint32_t* sticky = nullptr;
  uint64_t free_hbm_cached = 0;   // rtc_free_hbm
  double free_hbm_at = -1.0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // HIP events around the last pair_tiled_kernel launch, on the stream it was launched on (rtc_pair_last_kernel_ms)
  hipEvent_t pk0 = nullptr, pk1 = nullptr;
  int pk_valid = 0;
  // tiled pair kernel: the last plan (slice offsets + transposed column copy in scratch slots 1 / 4).
  // Reused only while pair_plan_hold is set by a caller that guarantees unchanged sketches between
  // launches (the row-chunk loop of the dense candidate-edge path).
  int quiet = 0;           // rtc_warmup's context: no RTC_VERBOSE lines
  int pair_last_path = 0;  // rtc_pair_last_path
  // rtc_diag_counters: [0] pair tiles the join took, [1] tiled-kernel tiles, [2] merge-kernel tiles, [3] candidate lists contracted
  // to their forest, [4] greedy runs replayed from ONE global join, [5] greedy query blocks of the block loop, [6] estimates handed
  // back instead of a launch (rtc_pair_edges_dev's overflow protocol)
  uint64_t diag[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // the inverted join's last refusal for density: the input it counted
  // keyed on the sketch buffer, its generation (every sketch / gather call on this context bumps sketch_gen) and the tile
  struct { const void* hashes = nullptr; uint64_t gen = 0; uint32_t n = 0, row0 = 0, row1 = 0, col0 = 0, col1 = 0; uint64_t K = 0, maxkey = 0; uint64_t edges_hint = 0; /* set by a refusal from the sample, for this call only: candidate edges to expect */ } join_dense;
  uint64_t sketch_gen = 1;
  // the candidate edge list of the last clustering call (rtc_edge_list_free keeps it, rtc_candidate_edges_device takes it): a
  // command line clusters once, a service or a benchmark loop many times over sets of one size -- the second call then starts with
  // a list that held the first one's edges, no allocation and no overflow redo
  void* mst_pinned = nullptr;      // rtc_mst_bufs: page-locked sketch sizes + forest of a whole-MST call
  size_t mst_pinned_bytes = 0;
  void* edge_cache = nullptr;
  uint64_t edge_cache_cap = 0;
  void* edge_cache_count = nullptr;
  int pair_plan_hold = 0, pair_plan_valid = 0;
  uint32_t pair_plan_tc1_hint = 0;
  struct {
    const void *hashes, *start, *len;
    uint32_t n, tc0, tc1;
    int width, P, npl;
    const uint32_t* d_so;
    const uint64_t* d_tbase;
    const void* d_tcols;
    const void* d_tcols_lo;
  } pair_plan = {};
  // KSSD filter tables of this context (cuckoo index or full table), keyed by (half_subk, drlevel, checksum)
  struct {
    int half_subk = -1, drlevel = -1;
    uint64_t checksum = 0;
    void* d_index = nullptr;
    int ck1 = 13, ck2 = 13;
    int32_t* d_table = nullptr;
    void* d_bucket = nullptr;  // bucket index (64 KiB of patterns) + ranks (64 KiB)
    void* d_bloom = nullptr;   // blocked Bloom filter of the kept middle 12-mers and their reverse complements (64 KiB)
    int bvar = -1;             // which bucket bits the index uses, -1: none
  } kssd;
};

int rtc_fail(rtc_ctx* ctx, int code, const char* fmt, ...);
// returns device scratch slot `slot` grown to at least `bytes`
int rtc_ws(rtc_ctx* ctx, int slot, size_t bytes, void** out);
int rtc_pinned(rtc_ctx* ctx, size_t bytes, void** out);
// free HBM of the context's device, for the scratch budgets of the pair phase: hipMemGetInfo walks the driver's tables, so the
// answer is kept for 100 ms or until this context allocates or frees (the budgets are halves of the free memory, not margins)
// One empty launch per sketch translation unit: the HIP runtime maps a unit's device code at its first launch (~10 ms each);
// rtc_warmup does that beside the command lines' first PCIe copies instead of in front of their first sketch.
int rtc_touch_sketch_minhash(rtc_ctx* ctx);
int rtc_touch_sketch_kssd(rtc_ctx* ctx);
int rtc_touch_sketch_minhash_packed(rtc_ctx* ctx);
int rtc_touch_unpack(rtc_ctx* ctx);
// The run list's contract -- ascending by start, disjoint, inside the batch -- checked on the device in one pass, without a host
// round trip: a violation raises the context's sticky flag (rtc_unpack.hip).  rtc_sticky_error: RTC_ERR_ARG once if it is up.
int rtc_check_runs_async(rtc_ctx* ctx, const uint64_t* d_runs, uint64_t n_runs, uint64_t n_bases);
int rtc_sticky_error(rtc_ctx* ctx);
uint64_t rtc_free_hbm(rtc_ctx* ctx);

// ---- internal C++ interfaces shared by the translation units ----------------------------------
// all-reduce of a per-round key array across the ranks of a multi-GPU run (rtc_comm.hip);
// dtype 0 = int64, 1 = uint32, 2 = uint64; op 0 = MIN, 1 = MAX
struct rtc_reduce_hook {
  int (*all_reduce)(void* self, void* d_buf, size_t count, int dtype, int op);
  void* self;
};
struct rtc_edge_list {  // device-resident candidate edges
  rtc_cedge* d_edges = nullptr;
  uint64_t cap = 0, m = 0;
  unsigned long long* d_count = nullptr;
  int contractions = 0;
};
// on_new (optional): called with every batch of freshly produced candidate edges (device pointer, count)
// before the list may be contracted -- the --dense histograms see every candidate pair exactly once
struct rtc_edge_observer {
  int (*on_new)(void* self, const rtc_cedge* d_new, uint64_t count);
  void* self;
};
int rtc_candidate_edges_device(rtc_ctx* ctx, const void* d_hashes, int width, const uint64_t* d_start,
                               const uint32_t* d_len, uint32_t n, uint32_t row0, uint32_t row1, int kmer_size,
                               int is_containment, double threshold, uint32_t s_fixed, rtc_edge_list* el,
                               const rtc_edge_observer* obs = nullptr);
// (ctx != NULL: a list of up to 1 GiB stays with the context for its next clustering call instead of going back to the driver)
void rtc_edge_list_free(rtc_edge_list* el, rtc_ctx* ctx = nullptr);
// rtc_pairs_join.hip: candidate edges of a lower-triangle tile by the inverted join; *handled = 0 when it declines
int rtc_pair_edges_join(rtc_ctx* ctx, const void* d_hashes, int width, const uint64_t* d_start, const uint32_t* d_len,
                        uint32_t n, uint32_t row0, uint32_t row1, uint32_t col0, uint32_t col1, int radio,
                        rtc_cedge* d_edges, uint64_t cap, uint64_t* d_count, double tiled_scale, int* handled);
// sorted: leave the forest in the reference's output order (weight, i, j) -- rtc_sort.hip; contractions pass false.
// max_len: the longest sketch when the caller knows it (sizes vary): the edge id of a round then carries the count too
int rtc_msf_device(rtc_ctx* ctx, const rtc_cedge* d_edges, uint64_t m, const uint32_t* d_len, uint32_t n,
                   int is_containment, uint32_t s_fixed, const rtc_reduce_hook* hook, rtc_cedge* d_sel,
                   uint64_t* n_sel_out, int* rounds_out, bool sorted = true, uint32_t max_len = 0);
int rtc_sort_forest_device(rtc_ctx* ctx, rtc_cedge* d_sel, uint64_t ns, const uint32_t* d_len, int wmode);
uint32_t rtc_fixed_size_of(const uint32_t* h_len, uint32_t n);
size_t rtc_msf_scratch_bytes(uint32_t n);
struct rtc_mst_bufs_t { uint32_t* h_len; rtc_cedge* h_sel; rtc_cedge* d_sel; };  // page-locked sizes + forest, device forest list
int rtc_mst_bufs(rtc_ctx* ctx, uint32_t n, rtc_mst_bufs_t* b);
int rtc_edges_to_mst_host_fixed(const rtc_cedge* h_sel, uint64_t m, const uint32_t* h_len, int kmer_size, int is_containment,
                                uint32_t s_fixed, rtc_edge* h_out);

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

// rtc_comm.hip -- the multi-GPU side of the path behind the C ABI: RCCL collectives over xGMI and
// the sharded clust-mst step built on them.
//
// The reference is a single shared-memory process (OpenMP over 8-row blocks of the pair space,
// src/MST.cpp:1382; over files, src/SketchInfo.cpp:878).  Here every GPU ("rank") sketches its own
// block of genomes, the sketches are gathered once into the canonical order (genome g of rank r at
// row r*n_local + g) and the strict lower triangle of the N x N pair space is cut into contiguous
// row ranges of equal cost.  Two kinds of exchange exist and nothing else:
//   * gather of sketch rows: W grouped ncclBroadcast calls, each in place on the owner's rows of the
//     global buffer (an all-gather with arbitrary row ranges, so the first part of every rank's
//     sketches travels on a side stream while the second part is still being sketched);
//   * per Boruvka round one ncclAllReduce(MIN) over a u64[n] key array (fixed sketch sizes), or
//     MIN, MIN, MAX over three small arrays (variable sizes).  The union step runs on every rank's
//     device on the identical reduced arrays.
// One rtc_comm belongs to one rtc_ctx (one GPU).  Ranks may be processes (bench.py under
// torch.distributed.run: rtc_comm_init_rank with an id the caller distributed) or host threads of
// one process (clust-mst: rtc_comm_init_all).  When two contexts of one process sit on the SAME
// device -- RCCL refuses duplicate GPUs -- an in-process exchange (host barrier + device copies)
// stands in, which is how the protocol is tested on a one-GPU box.
#include <math.h>
#include <rccl/rccl.h>   // types and enums only: the library itself is loaded on first use (below)
#include <dlfcn.h>
#include <mutex>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <thread>
#include <memory>
#include <mutex>
#include <numeric>
#include <vector>

#include "rtc_internal.h"

namespace {

// librccl.so is 570 MB of code objects for every architecture; a single-GPU run (the common command line) never needs
// it, so it is not a link-time dependency: the ten entry points used here are resolved on the first communicator call.
struct Rccl {
  void* h = nullptr;
  const char* err = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
std::once_flag g_rccl_once;

// Where RCCL is looked for, in this order: $RTC_RCCL_LIB (a file), a copy this process already holds (a host that
// imported torch: its wheel ships librccl.so beside its own libamdhip64), the directory of the HIP runtime this
// library itself resolved against, $ROCM_PATH/lib, /opt/rocm/lib, and last the loader's own search by soname.
// RTLD_LOCAL: only the function pointers below are used, nothing of RCCL goes into the global namespace.
std::string g_rccl_err, g_rccl_path;
void* rccl_open() {
  auto try_open = [](const std::string& path, int extra) -> void* {
    if (path.empty()) return nullptr;
    void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL | extra);
    if (h) g_rccl_path = path;
    else if (!extra) { const char* e = dlerror(); g_rccl_err += "  " + path + ": " + (e ? e : "?") + "\n"; }
    return h;
  };
  if (const char* e = getenv("RTC_RCCL_LIB")) {
    if (void* h = try_open(e, 0)) return h;
    return nullptr;  // an explicit choice that does not load is an error, not a hint
  }
  for (const char* name : {"librccl.so.1", "librccl.so"})
    if (void* h = try_open(name, RTLD_NOLOAD)) return h;
  std::vector<std::string> dirs;
  Dl_info info;
  if (dladdr((const void*)&hipGetDeviceCount, &info) && info.dli_fname) {
    std::string f = info.dli_fname;
    const size_t slash = f.rfind('/');
    if (slash != std::string::npos) dirs.push_back(f.substr(0, slash));
  }
  if (const char* e = getenv("ROCM_PATH")) dirs.push_back(std::string(e) + "/lib");
  dirs.push_back("/opt/rocm/lib");
  for (const std::string& d : dirs)
    for (const char* name : {"librccl.so.1", "librccl.so"})
      if (void* h = try_open(d + "/" + name, 0)) return h;
  for (const char* name : {"librccl.so.1", "librccl.so"})
    if (void* h = try_open(name, 0)) return h;
  return nullptr;
}

const Rccl* rccl() {  // nullptr when the library or one of its symbols is missing (g_rccl.err says which)
  std::call_once(g_rccl_once, [] {
    g_rccl.h = rccl_open();
    if (!g_rccl.h) { g_rccl_err = "librccl not found; tried\n" + g_rccl_err; g_rccl.err = g_rccl_err.c_str(); return; }
#define RTC_SYM(field, sym)                                                      \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(g_rccl.h, sym)); \
    if (!g_rccl.field && !g_rccl.err) g_rccl.err = sym " missing from librccl"
    RTC_SYM(GetUniqueId, "ncclGetUniqueId");
    RTC_SYM(CommInitRank, "ncclCommInitRank");
    RTC_SYM(CommInitAll, "ncclCommInitAll");
    RTC_SYM(CommDestroy, "ncclCommDestroy");
    RTC_SYM(CommAbort, "ncclCommAbort");
    RTC_SYM(AllReduce, "ncclAllReduce");
    RTC_SYM(Broadcast, "ncclBroadcast");
    RTC_SYM(GroupStart, "ncclGroupStart");
    RTC_SYM(GroupEnd, "ncclGroupEnd");
    RTC_SYM(GetErrorString, "ncclGetErrorString");
#undef RTC_SYM
    if (getenv("RTC_VERBOSE")) fprintf(stderr, "[comm]  RCCL from %s\n", g_rccl_path.c_str());
  });
  return g_rccl.err ? nullptr : &g_rccl;
}
#define RTC_NEED_RCCL(ctx)                                                                             \
  const Rccl* nc__ = rccl();                                                                           \
  if (!nc__) return rtc_fail((ctx), RTC_ERR_UNSUPPORTED, "RCCL is not available: %s", g_rccl.err)

// Watchdog: no collective may wait longer than this for its peers (RTC_COMM_TIMEOUT_S, default 120 s; <= 0: forever).
// A rank that never arrives -- it failed before the call, took another branch, died -- otherwise leaves the others inside
// the collective until somebody kills the job.
double comm_timeout_s(const rtc_ctx* ctx) {  // the context's option, read from the environment when the context was created
  const double v = ctx->opt.comm_timeout_s;
  return v > 0 ? v : 1e30;
}

struct LocalGroup {  // in-process exchange for contexts sharing a device
  std::mutex m;
  std::condition_variable cv;
  int n = 0, arrived = 0;
  uint64_t gen = 0;
  bool broken = false;  // a barrier timed out: every rank of the group fails from now on
  std::vector<const void*> ptr;
  bool barrier(double t) {  // t: the calling rank's deadline in seconds; false: a peer did not arrive in time (or the group broke earlier)
    std::unique_lock<std::mutex> lk(m);
    if (broken) return false;
    const uint64_t g = gen;
    if (++arrived == n) { arrived = 0; gen++; cv.notify_all(); return true; }
    const bool ok = t >= 1e29 ? (cv.wait(lk, [&] { return gen != g || broken; }), true)
                              : cv.wait_for(lk, std::chrono::duration<double>(t), [&] { return gen != g || broken; });
    if (!ok || broken) { broken = true; cv.notify_all(); return false; }
    return true;
  }
};

}  // namespace

struct rtc_comm {
  rtc_ctx* ctx = nullptr;
  int rank = 0, size = 1;
  ncclComm_t nccl = nullptr;
  std::shared_ptr<LocalGroup> local;
  hipStream_t side = nullptr;      // gathers overlap compute on the context stream
  hipEvent_t ev_ready = nullptr, ev_done = nullptr, ev_watch = nullptr;
  hipEvent_t ev_front = nullptr, ev_front_side = nullptr;  // in front of the collective(s) being watched (context / side stream)
  hipEvent_t ev_t[3] = {nullptr, nullptr, nullptr};        // rtc_mst_sharded's phase timing
  bool side_front_pending = false;
  bool side_busy = false;
  bool broken = false;             // a collective timed out and the communicator was aborted
  uint32_t kssd_need = 0;          // rtc_sketch_kssd_packed_sharded: longest sketch of this rank's batches so far
  int kssd_width = 0;              // ... and the tuple width they came out with
};

#define RTC_NCCL(ctx, call)                                                                     \
  do {                                                                                          \
    ncclResult_t r__ = (call);                                                                  \
    if (r__ != ncclSuccess)                                                                     \
      return rtc_fail((ctx), RTC_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #call, nc__->GetErrorString(r__)); \
  } while (0)

namespace {

template <typename T>
__global__ void local_reduce_kernel(const void* const* __restrict__ srcs, int nsrc, size_t count, int op, T* __restrict__ dst) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
    T v = ((const T*)srcs[0])[i];
    for (int s = 1; s < nsrc; s++) {
      const T w = ((const T*)srcs[s])[i];
      v = op == 0 ? (w < v ? w : v) : (w > v ? w : v);
    }
    dst[i] = v;
  }
}

int comm_broken(rtc_comm* c, const char* what) {
  c->broken = true;
  return rtc_fail(c->ctx, RTC_ERR_COMM, "%s: rank %d of %d waited %.0f s for its peers (RTC_COMM_TIMEOUT_S); the communicator is "
                  "aborted", what, c->rank, c->size, comm_timeout_s(c->ctx));
}
#define RTC_LOCAL_BARRIER(c, what) do { if (!(c)->local->barrier(comm_timeout_s((c)->ctx))) return comm_broken((c), (what)); } while (0)

// RCCL collectives are enqueued, not awaited: the host polls an event behind the collective on its stream, and when the
// deadline passes aborts the communicator (ncclCommAbort: the only way out of a collective whose peers never come).
// The deadline counts the wait for the PEERS only: the clock starts when the stream has reached the collective (the event
// comm_front recorded in front of it has completed) -- kernels queued ahead of it on the stream, e.g. a long pair phase on
// a rank that arrives late for a good reason, are not time spent waiting for anybody.
int comm_front(rtc_comm* c, hipStream_t stream) {
  if (!c->nccl) return RTC_OK;
  if (stream == c->side) {
    if (c->side_front_pending) return RTC_OK;  // the first gather of a batch marks the front; rtc_comm_wait watches the batch
    c->side_front_pending = true;
    RTC_HIP(c->ctx, hipEventRecord(c->ev_front_side, stream));
  } else {
    RTC_HIP(c->ctx, hipEventRecord(c->ev_front, stream));
  }
  return RTC_OK;
}
int comm_watch(rtc_comm* c, hipStream_t stream, const char* what) {
  rtc_ctx* ctx = c->ctx;
  if (!c->nccl) return RTC_OK;
  RTC_HIP(ctx, hipEventRecord(c->ev_watch, stream));
  const hipEvent_t front = stream == c->side ? c->ev_front_side : c->ev_front;
  if (stream == c->side) c->side_front_pending = false;
  const double limit = comm_timeout_s(ctx);
  auto t0 = std::chrono::steady_clock::now();
  bool reached = false;  // the stream has arrived at the collective: the clock runs
  int spins = 0;
  for (;;) {
    const hipError_t e = hipEventQuery(c->ev_watch);
    if (e == hipSuccess) return RTC_OK;
    if (e != hipErrorNotReady) return rtc_fail(ctx, RTC_ERR_HIP, "%s: hipEventQuery -> %s", what, hipGetErrorString(e));
    if (!reached) {
      const hipError_t f = hipEventQuery(front);
      if (f == hipErrorNotReady) {
        // the stream has not arrived yet: work queued ahead of the collective is running.  That wait is not charged to the
        // peers, but it is bounded too (ten limits): a kernel that never ends, or a side stream stuck behind an earlier
        // collective nobody watched, must not keep the host here forever.
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 10.0 * limit) {
          if (const Rccl* nc = rccl()) (void)nc->CommAbort(c->nccl);
          c->nccl = nullptr;
          return comm_broken(c, what);
        }
        if (++spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
        continue;
      }
      reached = true;  // (an event never recorded reads as complete: the clock then starts at once, as before)
      t0 = std::chrono::steady_clock::now();
    }
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) {
      if (const Rccl* nc = rccl()) (void)nc->CommAbort(c->nccl);
      c->nccl = nullptr;
      return comm_broken(c, what);
    }
    if (++spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));  // the first ~ms busy: collectives are short
  }
}
#define RTC_COMM_ALIVE(c) do { if ((c)->broken) return rtc_fail((c)->ctx, RTC_ERR_COMM, "the communicator was aborted after a collective timed out"); } while (0)

int comm_finish_init(rtc_comm* c) {
  rtc_ctx* ctx = c->ctx;
  RTC_HIP(ctx, hipSetDevice(ctx->device));
  RTC_HIP(ctx, hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
  RTC_HIP(ctx, hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming));
  RTC_HIP(ctx, hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming));
  RTC_HIP(ctx, hipEventCreateWithFlags(&c->ev_watch, hipEventDisableTiming));
  RTC_HIP(ctx, hipEventCreateWithFlags(&c->ev_front, hipEventDisableTiming));
  RTC_HIP(ctx, hipEventCreateWithFlags(&c->ev_front_side, hipEventDisableTiming));
  return RTC_OK;
}

// dtype 0 = int64, 1 = uint32, 2 = uint64; op 0 = MIN, 1 = MAX; on `stream`
int comm_all_reduce_on(rtc_comm* c, void* d_buf, size_t count, int dtype, int op, hipStream_t stream) {
  rtc_ctx* ctx = c->ctx;
  RTC_COMM_ALIVE(c);
  if ((c->size == 1 && !c->nccl) || count == 0) return RTC_OK;
  if (c->nccl) {
    RTC_NEED_RCCL(ctx);
    RTC_TRY(comm_front(c, stream));
    RTC_NCCL(ctx, nc__->AllReduce(d_buf, d_buf, count, dtype == 0 ? ncclInt64 : dtype == 1 ? ncclUint32 : ncclUint64, op == 0 ? ncclMin : ncclMax, c->nccl, stream));
    return comm_watch(c, stream, "all-reduce");
  }
  LocalGroup& g = *c->local;
  const size_t esz = dtype == 1 ? 4 : 8;
  void* ws = nullptr;
  RTC_TRY(rtc_ws(ctx, 5, count * esz + (size_t)c->size * 8 + 64, &ws));
  void* tmp = ws;
  const void** d_ptrs = (const void**)((char*)ws + ((count * esz + 63) & ~(size_t)63));
  RTC_HIP(ctx, hipStreamSynchronize(stream));
  g.ptr[c->rank] = d_buf;
  RTC_LOCAL_BARRIER(c, "all-reduce");
  std::vector<const void*> ptrs(g.ptr.begin(), g.ptr.begin() + c->size);
  RTC_HIP(ctx, hipMemcpyAsync((void*)d_ptrs, ptrs.data(), (size_t)c->size * 8, hipMemcpyHostToDevice, stream));
  const uint32_t grid = (uint32_t)std::max<size_t>(1, std::min<size_t>((count + 255) / 256, 2048));
  if (dtype == 0) hipLaunchKernelGGL(local_reduce_kernel<long long>, dim3(grid), dim3(256), 0, stream, (const void* const*)d_ptrs, c->size, count, op, (long long*)tmp);
  else if (dtype == 2) hipLaunchKernelGGL(local_reduce_kernel<unsigned long long>, dim3(grid), dim3(256), 0, stream, (const void* const*)d_ptrs, c->size, count, op, (unsigned long long*)tmp);
  else hipLaunchKernelGGL(local_reduce_kernel<uint32_t>, dim3(grid), dim3(256), 0, stream, (const void* const*)d_ptrs, c->size, count, op, (uint32_t*)tmp);
  RTC_HIP(ctx, hipGetLastError());
  RTC_HIP(ctx, hipStreamSynchronize(stream));
  RTC_LOCAL_BARRIER(c, "all-reduce");  // everybody has read everybody's input
  RTC_HIP(ctx, hipMemcpyAsync(d_buf, tmp, count * esz, hipMemcpyDeviceToDevice, stream));
  RTC_HIP(ctx, hipStreamSynchronize(stream));
  RTC_LOCAL_BARRIER(c, "all-reduce");
  return RTC_OK;
}

// rows [a, b) of every rank's block of the canonical global buffer travel to all ranks, in place
int comm_gather_rows_on(rtc_comm* c, void* d_global, size_t row_bytes, uint32_t n_local, uint32_t a, uint32_t b, hipStream_t stream) {
  rtc_ctx* ctx = c->ctx;
  RTC_COMM_ALIVE(c);
  if ((c->size == 1 && !c->nccl) || b <= a || row_bytes == 0) return RTC_OK;
  const size_t bytes = (size_t)(b - a) * row_bytes;
  if (c->nccl) {
    RTC_NEED_RCCL(ctx);
    RTC_TRY(comm_front(c, stream));
    RTC_NCCL(ctx, nc__->GroupStart());
    for (int r = 0; r < c->size; r++) {
      char* p = (char*)d_global + ((size_t)r * n_local + a) * row_bytes;
      ncclResult_t st = nc__->Broadcast(p, p, bytes, ncclInt8, r, c->nccl, stream);
      if (st != ncclSuccess) {
        (void)nc__->GroupEnd();
        c->side_front_pending = false;
        return rtc_fail(ctx, RTC_ERR_HIP, "ncclBroadcast -> %s", nc__->GetErrorString(st));
      }
    }
    if (ncclResult_t st = nc__->GroupEnd(); st != ncclSuccess) {
      c->side_front_pending = false;
      return rtc_fail(ctx, RTC_ERR_HIP, "ncclGroupEnd -> %s", nc__->GetErrorString(st));
    }
    return stream == c->side ? RTC_OK : comm_watch(c, stream, "gather");  // side stream: rtc_comm_wait watches it
  }
  LocalGroup& g = *c->local;
  RTC_HIP(ctx, hipStreamSynchronize(stream));
  g.ptr[c->rank] = d_global;
  RTC_LOCAL_BARRIER(c, "gather");
  for (int r = 0; r < c->size; r++) {
    if (r == c->rank) continue;
    const size_t off = ((size_t)r * n_local + a) * row_bytes;
    RTC_HIP(ctx, hipMemcpyAsync((char*)d_global + off, (const char*)g.ptr[r] + off, bytes, hipMemcpyDefault, stream));
  }
  RTC_HIP(ctx, hipStreamSynchronize(stream));
  RTC_LOCAL_BARRIER(c, "gather");
  return RTC_OK;
}

int hook_all_reduce(void* self, void* d_buf, size_t count, int dtype, int op) {
  rtc_comm* c = (rtc_comm*)self;
  return comm_all_reduce_on(c, d_buf, count, dtype, op, c->ctx->stream);
}

}  // namespace

extern "C" {

int rtc_comm_unique_id(void* id_out) {
  if (!id_out) return RTC_ERR_ARG;
  static_assert(sizeof(ncclUniqueId) == RTC_COMM_ID_BYTES, "rtclust.h: RTC_COMM_ID_BYTES");
  RTC_NEED_RCCL(nullptr);
  ncclUniqueId id;
  ncclResult_t r = nc__->GetUniqueId(&id);
  if (r != ncclSuccess) return rtc_fail(nullptr, RTC_ERR_HIP, "ncclGetUniqueId -> %s", nc__->GetErrorString(r));
  memcpy(id_out, &id, sizeof id);
  return RTC_OK;
}

int rtc_comm_init_rank(rtc_ctx* ctx, int nranks, int rank, const void* id, rtc_comm** out) {
  if (!ctx || !out || nranks < 1 || rank < 0 || rank >= nranks || (nranks > 1 && !id)) return RTC_ERR_ARG;
  *out = nullptr;
  RTC_HIP(ctx, hipSetDevice(ctx->device));
  std::unique_ptr<rtc_comm> c(new rtc_comm());
  c->ctx = ctx; c->rank = rank; c->size = nranks;
  if (nranks > 1 || (id && ctx->opt.comm_force_rccl)) {  // the env switch drives the RCCL calls on one GPU (tests)
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    RTC_NEED_RCCL(ctx);
    RTC_NCCL(ctx, nc__->CommInitRank(&c->nccl, nranks, uid, rank));
  }
  RTC_TRY(comm_finish_init(c.get()));
  *out = c.release();
  return RTC_OK;
}

int rtc_comm_init_all(rtc_ctx** ctxs, int n, rtc_comm** comms_out) {
  if (!ctxs || !comms_out || n < 1) return RTC_ERR_ARG;
  for (int i = 0; i < n; i++) { if (!ctxs[i]) return RTC_ERR_ARG; comms_out[i] = nullptr; }
  bool distinct = true;
  for (int i = 0; i < n; i++) for (int j = 0; j < i; j++) if (ctxs[i]->device == ctxs[j]->device) distinct = false;
  std::vector<std::unique_ptr<rtc_comm>> cs;
  for (int i = 0; i < n; i++) { cs.emplace_back(new rtc_comm()); cs[i]->ctx = ctxs[i]; cs[i]->rank = i; cs[i]->size = n; }
  if ((n > 1 && distinct) || (n == 1 && ctxs[0]->opt.comm_force_rccl)) {  // the env switch: RCCL on one GPU (tests)
    std::vector<int> devs(n);
    std::vector<ncclComm_t> nc(n);
    for (int i = 0; i < n; i++) devs[i] = ctxs[i]->device;
    RTC_NEED_RCCL(ctxs[0]);
    RTC_NCCL(ctxs[0], nc__->CommInitAll(nc.data(), n, devs.data()));
    for (int i = 0; i < n; i++) cs[i]->nccl = nc[i];
  } else if (n > 1) {  // shared device(s): RCCL rejects duplicate GPUs
    auto g = std::make_shared<LocalGroup>();
    g->n = n; g->ptr.assign(n, nullptr);
    for (int i = 0; i < n; i++) cs[i]->local = g;
  }
  for (int i = 0; i < n; i++) RTC_TRY(comm_finish_init(cs[i].get()));
  for (int i = 0; i < n; i++) comms_out[i] = cs[i].release();
  return RTC_OK;
}

void rtc_comm_destroy(rtc_comm* c) {
  if (!c) return;
  (void)hipSetDevice(c->ctx->device);
  if (c->side) { (void)hipStreamSynchronize(c->side); (void)hipStreamDestroy(c->side); }
  if (c->ev_ready) (void)hipEventDestroy(c->ev_ready);
  if (c->ev_done) (void)hipEventDestroy(c->ev_done);
  if (c->ev_watch) (void)hipEventDestroy(c->ev_watch);
  if (c->ev_front) (void)hipEventDestroy(c->ev_front);
  if (c->ev_front_side) (void)hipEventDestroy(c->ev_front_side);
  for (hipEvent_t e : c->ev_t) if (e) (void)hipEventDestroy(e);
  if (c->nccl && rccl()) (void)rccl()->CommDestroy(c->nccl);
  delete c;
}

int rtc_comm_rank(const rtc_comm* c) { return c ? c->rank : -1; }
int rtc_comm_size(const rtc_comm* c) { return c ? c->size : 0; }
const char* rtc_comm_backend(const rtc_comm* c) { return !c ? "" : (c->nccl ? "rccl" : (c->local ? "in-process" : "single")); }

int rtc_comm_all_reduce(rtc_comm* c, void* d_buf, size_t count, int dtype, int op) {
  if (!c || (count && !d_buf) || dtype < 0 || dtype > 2 || op < 0 || op > 1) return RTC_ERR_ARG;
  RTC_HIP(c->ctx, hipSetDevice(c->ctx->device));
  return comm_all_reduce_on(c, d_buf, count, dtype, op, c->ctx->stream);
}

int rtc_comm_all_reduce_host(rtc_comm* c, int64_t* h_vals, size_t count, int op) {
  if (!c || !h_vals || count == 0 || count > 64 || op < 0 || op > 1) return RTC_ERR_ARG;
  RTC_COMM_ALIVE(c);
  if (c->size == 1 && !c->nccl) return RTC_OK;
  rtc_ctx* ctx = c->ctx;
  RTC_HIP(ctx, hipSetDevice(ctx->device));
  void* ws = nullptr;
  RTC_TRY(rtc_ws(ctx, 0, 4096, &ws));
  RTC_HIP(ctx, hipMemcpyAsync(ws, h_vals, count * 8, hipMemcpyHostToDevice, ctx->stream));
  RTC_TRY(comm_all_reduce_on(c, ws, count, 0, op, ctx->stream));
  RTC_HIP(ctx, hipMemcpyAsync(h_vals, ws, count * 8, hipMemcpyDeviceToHost, ctx->stream));
  RTC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return RTC_OK;
}

int rtc_comm_gather_rows(rtc_comm* c, void* d_global, size_t row_bytes, uint32_t n_local, uint32_t a, uint32_t b, int async) {
  if (!c || !d_global || a > b || b > n_local) return RTC_ERR_ARG;
  rtc_ctx* ctx = c->ctx;
  RTC_HIP(ctx, hipSetDevice(ctx->device));
  ctx->sketch_gen++;  // other ranks' rows arrive in the buffer
  if (!async) return comm_gather_rows_on(c, d_global, row_bytes, n_local, a, b, ctx->stream);
  // side stream: starts when the work enqueued so far on the context stream (the sketch kernel of
  // these rows) is done; rtc_comm_wait joins it back
  RTC_HIP(ctx, hipEventRecord(c->ev_ready, ctx->stream));
  RTC_HIP(ctx, hipStreamWaitEvent(c->side, c->ev_ready, 0));
  RTC_TRY(comm_gather_rows_on(c, d_global, row_bytes, n_local, a, b, c->side));
  c->side_busy = true;
  return RTC_OK;
}

// d_buf[0..bytes) of rank `root` replaces every other rank's d_buf (context stream)
int rtc_comm_broadcast(rtc_comm* c, void* d_buf, size_t bytes, int root) {
  if (!c || (bytes && !d_buf) || root < 0 || root >= c->size) return RTC_ERR_ARG;
  rtc_ctx* ctx = c->ctx;
  RTC_COMM_ALIVE(c);
  if ((c->size == 1 && !c->nccl) || bytes == 0) return RTC_OK;
  RTC_HIP(ctx, hipSetDevice(ctx->device));
  if (c->nccl) {
    RTC_NEED_RCCL(ctx);
    RTC_TRY(comm_front(c, ctx->stream));
    RTC_NCCL(ctx, nc__->Broadcast(d_buf, d_buf, bytes, ncclInt8, root, c->nccl, ctx->stream));
    return comm_watch(c, ctx->stream, "broadcast");
  }
  LocalGroup& g = *c->local;
  RTC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  g.ptr[c->rank] = d_buf;
  RTC_LOCAL_BARRIER(c, "broadcast");
  if (c->rank != root) RTC_HIP(ctx, hipMemcpyAsync(d_buf, g.ptr[root], bytes, hipMemcpyDefault, ctx->stream));
  RTC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  RTC_LOCAL_BARRIER(c, "broadcast");
  return RTC_OK;
}

int rtc_comm_wait(rtc_comm* c) {
  if (!c) return RTC_ERR_ARG;
  RTC_COMM_ALIVE(c);
  if (!c->side_busy) return RTC_OK;
  rtc_ctx* ctx = c->ctx;
  RTC_HIP(ctx, hipSetDevice(ctx->device));
  RTC_HIP(ctx, hipEventRecord(c->ev_done, c->side));
  RTC_HIP(ctx, hipStreamWaitEvent(ctx->stream, c->ev_done, 0));
  c->side_busy = false;
  return comm_watch(c, c->side, "gather (side stream)");  // the host waits here for the gathers, with the deadline
}

// Contiguous row ranges of the strict lower triangle with equal cost; row i costs (i + fixed_cols)
// (fixed_cols: per-row-block table build that does not depend on the row length, ~1.84 columns per
// sketch hash on MI355X).  h_bounds[world + 1].
int rtc_triangle_rows(uint32_t n, int world, double fixed_cols, uint32_t* h_bounds) {
  if (world < 1 || !h_bounds) return RTC_ERR_ARG;
  const double c = fixed_cols, area = (double)n * n / 2.0 + c * n;
  h_bounds[0] = 0;
  for (int r = 1; r <= world; r++) {
    double v = -c + sqrt(c * c + 2.0 * area * r / world);
    long b = lround(v);
    b = std::min<long>(std::max<long>(b, h_bounds[r - 1]), n);
    h_bounds[r] = (uint32_t)b;
  }
  h_bounds[world] = n;
  return RTC_OK;
}

// Multi-GPU MinHash sketch phase: this rank's genomes -> its block of the canonical global buffers
// (d_out_global: size*n_local*stride u64, d_cnt_global: size*n_local u32), in two parts so that the
// gather of the first part (side stream) runs beside the sketch kernel of the second.  Every rank
// passes the same n_local / stride.  Returns with the gathers enqueued; the context stream already
// waits for them (no host synchronisation here).
int rtc_sketch_minhash_sharded(rtc_ctx* ctx, rtc_comm* c, const uint8_t* d_seq, const uint64_t* h_off, uint32_t n_local,
                               int k, uint32_t seed, const uint32_t* h_sizes, uint32_t size, uint64_t* d_out_global,
                               uint32_t stride, uint32_t* d_cnt_global) {
  if (!ctx || !c || c->ctx != ctx || !h_off || (n_local && (!d_seq || !d_out_global || !d_cnt_global))) return RTC_ERR_ARG;
  if (n_local == 0) return RTC_OK;
  {  // the canonical order needs identical block shapes
    int64_t v[4] = {(int64_t)n_local, -(int64_t)n_local, (int64_t)stride, -(int64_t)stride};
    RTC_TRY(rtc_comm_all_reduce_host(c, v, 4, 1));
    if (v[0] != (int64_t)n_local || -v[1] != (int64_t)n_local || v[2] != (int64_t)stride || -v[3] != (int64_t)stride)
      return rtc_fail(ctx, RTC_ERR_ARG, "ranks disagree on genomes per rank (%u here, %lld..%lld) or stride (%u here, %lld..%lld)",
                      n_local, (long long)-v[1], (long long)v[0], stride, (long long)-v[3], (long long)v[2]);
  }
  const uint32_t slots = 3u * (uint32_t)ctx->num_cu;
  uint32_t split = n_local;
  if (c->size > 1 && n_local >= 8) split = (n_local >= 2 * slots) ? std::max(slots, (uint32_t)(0.8 * n_local) / slots * slots) : (3 * n_local) / 4;
  uint64_t* my_out = d_out_global + (size_t)c->rank * n_local * stride;
  uint32_t* my_cnt = d_cnt_global + (size_t)c->rank * n_local;
  const uint32_t parts[3] = {0, split, n_local};
  for (int p = 0; p < 2; p++) {
    const uint32_t a = parts[p], b = parts[p + 1];
    if (b <= a) continue;
    RTC_TRY(rtc_sketch_minhash_dev(ctx, d_seq, h_off + a, b - a, k, seed, h_sizes ? h_sizes + a : nullptr, size,
                                   my_out + (size_t)a * stride, stride, my_cnt + a));
    RTC_TRY(rtc_comm_gather_rows(c, d_out_global, (size_t)stride * 8, n_local, a, b, 1));
    RTC_TRY(rtc_comm_gather_rows(c, d_cnt_global, 4, n_local, a, b, 1));
  }
  return rtc_comm_wait(c);
}

// The same phase for a rank whose genomes are resident as batches in the 2-bit staging format (the command lines' form;
// north_star's "packed sequence"): one call per batch, in the order of the rank's rows.  Batch rows [row_first,
// row_first + n_batch) of this rank's block are sketched straight from the packed bases and their gather is started on the
// side stream behind the sketch kernel, so it travels beside the NEXT batch's kernel; the last batch (last != 0) is cut in
// two parts like the character form, and the call returns with the context stream waiting for every gather.  Every rank
// passes the same sequence of (row_first, n_batch).
namespace {
int sharded_shape_check(rtc_ctx* ctx, rtc_comm* c, uint32_t n_local, uint32_t stride, uint32_t n_batch) {
  int64_t v[6] = {(int64_t)n_local, -(int64_t)n_local, (int64_t)stride, -(int64_t)stride, (int64_t)n_batch, -(int64_t)n_batch};
  RTC_TRY(rtc_comm_all_reduce_host(c, v, 6, 1));
  if (v[0] != (int64_t)n_local || -v[1] != (int64_t)n_local || v[2] != (int64_t)stride || -v[3] != (int64_t)stride ||
      v[4] != (int64_t)n_batch || -v[5] != (int64_t)n_batch)
    return rtc_fail(ctx, RTC_ERR_ARG, "ranks disagree on genomes per rank (%u here, %lld..%lld), stride (%u here, %lld..%lld) or first batch "
                    "(%u here, %lld..%lld)", n_local, (long long)-v[1], (long long)v[0], stride, (long long)-v[3], (long long)v[2],
                    n_batch, (long long)-v[5], (long long)v[4]);
  return RTC_OK;
}
// the two parts of a rank's last batch: the first a whole number of full rounds of the sketch kernel (one genome per
// workgroup slot) so that the extra launch boundary costs no idle tail
uint32_t last_batch_split(const rtc_ctx* ctx, const rtc_comm* c, uint32_t n_batch) {
  if (c->size <= 1 || n_batch < 8) return n_batch;
  const uint32_t slots = 3u * (uint32_t)ctx->num_cu;
  return n_batch >= 2 * slots ? std::max(slots, (uint32_t)(0.8 * n_batch) / slots * slots) : (3 * n_batch) / 4;
}
}  // namespace

int rtc_sketch_minhash_packed_sharded(rtc_ctx* ctx, rtc_comm* c, const uint8_t* d_packed, uint64_t n_bases, const uint64_t* d_runs,
                                      uint64_t n_runs, const uint64_t* h_off, uint32_t n_batch, uint32_t row_first, uint32_t n_local,
                                      int last, int k, uint32_t seed, const uint32_t* h_sizes, uint32_t size, uint64_t* d_out_global,
                                      uint32_t stride, uint32_t* d_cnt_global) {
  if (!ctx || !c || c->ctx != ctx || !h_off || (n_batch && (!d_packed || !d_out_global || !d_cnt_global))) return RTC_ERR_ARG;
  if ((uint64_t)row_first + n_batch > n_local) return rtc_fail(ctx, RTC_ERR_ARG, "batch rows [%u, %u) outside the rank's %u rows", row_first, row_first + n_batch, n_local);
  if (row_first == 0) RTC_TRY(sharded_shape_check(ctx, c, n_local, stride, n_batch));  // the canonical order needs identical block shapes
  uint64_t* my_out = d_out_global + ((size_t)c->rank * n_local + row_first) * stride;
  uint32_t* my_cnt = d_cnt_global + (size_t)c->rank * n_local + row_first;
  const uint32_t parts[3] = {0, last ? last_batch_split(ctx, c, n_batch) : n_batch, n_batch};
  for (int p = 0; p < 2; p++) {
    const uint32_t a = parts[p], b = parts[p + 1];
    if (b <= a) continue;
    RTC_TRY(rtc_sketch_minhash_packed_dev(ctx, d_packed, n_bases, d_runs, n_runs, h_off + a, b - a, k, seed, h_sizes ? h_sizes + a : nullptr,
                                          size, my_out + (size_t)a * stride, stride, my_cnt + a));
    RTC_TRY(rtc_comm_gather_rows(c, d_out_global, (size_t)stride * 8, n_local, row_first + a, row_first + b, 1));
    RTC_TRY(rtc_comm_gather_rows(c, d_cnt_global, 4, n_local, row_first + a, row_first + b, 1));
  }
  return last ? rtc_comm_wait(c) : RTC_OK;
}

// --fast: sketchFileWithKssd (src/SketchInfo.cpp:994-1252) over the rank's packed batches, same protocol.  KSSD sketches vary
// in length: the rows travel at the caller's `stride` (a tight one saves link time: 1.25 x the expected count is ample for
// genomes of one length).  A batch whose longest sketch exceeds it is still gathered -- the collectives of all ranks stay
// matched -- and the call with last != 0 returns RTC_ERR_OVERFLOW on EVERY rank with *h_need = the longest sketch any rank
// produced; the caller then repeats the phase with wider rows.  *width_out: 4 or 8, identical on all ranks.
int rtc_sketch_kssd_packed_sharded(rtc_ctx* ctx, rtc_comm* c, const uint8_t* d_packed, uint64_t n_bases, const uint64_t* d_runs,
                                   uint64_t n_runs, const uint64_t* h_off, uint32_t n_batch, uint32_t row_first, uint32_t n_local,
                                   int last, int kmer_size, int drlevel, const int32_t* h_shuffled_dim, void* d_out_global,
                                   uint32_t stride, uint32_t* d_cnt_global, int* width_out, uint32_t* h_need) {
  if (!ctx || !c || c->ctx != ctx || !h_off || !width_out || (n_batch && (!d_packed || !d_out_global || !d_cnt_global))) return RTC_ERR_ARG;
  if ((uint64_t)row_first + n_batch > n_local) return rtc_fail(ctx, RTC_ERR_ARG, "batch rows [%u, %u) outside the rank's %u rows", row_first, row_first + n_batch, n_local);
  if (row_first == 0) {
    RTC_TRY(sharded_shape_check(ctx, c, n_local, stride, n_batch));
    c->kssd_need = 0;
    c->kssd_width = 0;
  }
  const int width = ((kmer_size + 1) / 2 - drlevel > 8) ? 8 : 4;  // src/SketchInfo.cpp:1021
  char* my_out = (char*)d_out_global + ((size_t)c->rank * n_local + row_first) * stride * width;
  uint32_t* my_cnt = d_cnt_global + (size_t)c->rank * n_local + row_first;
  const uint32_t parts[3] = {0, last ? last_batch_split(ctx, c, n_batch) : n_batch, n_batch};
  int failed = RTC_OK;
  for (int p = 0; p < 2; p++) {
    const uint32_t a = parts[p], b = parts[p + 1];
    if (b <= a) continue;
    uint32_t need = 0;
    int w = width;
    const int st = rtc_sketch_kssd_packed_dev(ctx, d_packed, n_bases, d_runs, n_runs, h_off + a, b - a, kmer_size, drlevel, h_shuffled_dim,
                                              my_out + (size_t)a * stride * width, stride, my_cnt + a, &w, &need);
    if (st == RTC_OK || st == RTC_ERR_OVERFLOW) { c->kssd_need = std::max(c->kssd_need, need); c->kssd_width = w; }
    else if (failed == RTC_OK) failed = st;  // (an argument error: every rank sees it; a device fault: nothing to save)
    if (failed != RTC_OK) break;
    RTC_TRY(rtc_comm_gather_rows(c, d_out_global, (size_t)stride * width, n_local, row_first + a, row_first + b, 1));
    RTC_TRY(rtc_comm_gather_rows(c, d_cnt_global, 4, n_local, row_first + a, row_first + b, 1));
  }
  *width_out = width;
  if (failed != RTC_OK) return failed;
  if (!last) return RTC_OK;
  RTC_TRY(rtc_comm_wait(c));
  int64_t v[3] = {(int64_t)c->kssd_need, (int64_t)c->kssd_width, -(int64_t)c->kssd_width};
  RTC_TRY(rtc_comm_all_reduce_host(c, v, 3, 1));
  if (h_need) *h_need = (uint32_t)v[0];
  if (v[1] != -v[2]) return rtc_fail(ctx, RTC_ERR_ARG, "ranks disagree on the KSSD tuple width (%lld / %lld)", (long long)v[1], (long long)-v[2]);
  if (v[0] > (int64_t)stride) return rtc_fail(ctx, RTC_ERR_OVERFLOW, "a genome produced %lld KSSD tuples on some rank, the rows hold %u", (long long)v[0], stride);
  return RTC_OK;
}

// compute_minhash_mst / compute_kssd_mst (src/MST.cpp:1290-1737, :216-807) across the ranks of `c`:
// this rank evaluates its row range of the pair space, the Boruvka rounds all-reduce their key
// arrays, every rank returns the identical forest (identical to rtc_mst on one GPU: the total order
// on (weight, i, j) makes the minimum spanning forest unique).  Sketches: the complete, canonical set.
int rtc_mst_sharded(rtc_ctx* ctx, rtc_comm* c, const void* d_hashes, int width, const uint64_t* d_start,
                    const uint32_t* d_len, uint32_t n, int kmer_size, int is_containment, double threshold,
                    rtc_edge* h_edges_out, uint64_t* h_n_edges, rtc_shard_stats* stats) {
  if (!ctx || !c || c->ctx != ctx || !h_n_edges || (n && (!d_start || !d_len || !h_edges_out))) return RTC_ERR_ARG;
  *h_n_edges = 0;
  if (stats) memset(stats, 0, sizeof *stats);
  if (n < 2) return RTC_OK;
  RTC_HIP(ctx, hipSetDevice(ctx->device));
  RTC_TRY(rtc_comm_wait(c));
  rtc_mst_bufs_t B{};
  RTC_TRY(rtc_mst_bufs(ctx, n, &B));
  uint32_t* const h_len = B.h_len;
  RTC_HIP(ctx, hipMemcpyAsync(h_len, d_len, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
  RTC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const uint32_t s_fixed = rtc_fixed_size_of(h_len, n);
  double mean = 0;
  for (uint32_t g = 0; g < n; g++) mean += h_len[g];
  mean /= n;
  std::vector<uint32_t> bounds(c->size + 1);
  RTC_TRY(rtc_triangle_rows(n, c->size, c->size > 1 ? 1.84 * mean : 0.0, bounds.data()));
  const uint32_t row0 = bounds[c->rank], row1 = bounds[c->rank + 1];

  // A failure that only one rank sees (its candidate list does not fit, an allocation fails) must not leave the
  // others waiting inside the next collective: the ranks agree on the status (one small all-reduce(MAX)) before the
  // Boruvka rounds start and return the error together.  (A HIP fault inside the rounds is not recoverable either way.)
  auto agree = [&](int local) -> int {
    if (c->size == 1 && !c->nccl) return local;
    int64_t v = local;
    const int r = rtc_comm_all_reduce_host(c, &v, 1, 1);
    if (r != RTC_OK) return r;
    if (local == RTC_OK && v != RTC_OK) return rtc_fail(ctx, (int)v, "another rank of the sharded MST step failed (status %d)", (int)v);
    return local != RTC_OK ? local : (int)v;
  };
  int st = RTC_OK;
  if (!c->ev_t[0]) {  // the phase events live with the communicator
    for (int i = 0; i < 3 && st == RTC_OK; i++)
      if (hipEventCreate(&c->ev_t[i]) != hipSuccess) st = rtc_fail(ctx, RTC_ERR_HIP, "hipEventCreate failed");
  }
  hipEvent_t e0 = c->ev_t[0], e1 = c->ev_t[1], e2 = c->ev_t[2];
  rtc_edge_list el{};
  rtc_cedge* const d_sel = B.d_sel;
  const bool verbose = ctx->opt.verbose && !ctx->quiet;
  auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double tv0 = now();
  if (st == RTC_OK) {
    (void)hipEventRecord(e0, ctx->stream);
    st = rtc_candidate_edges_device(ctx, d_hashes, width, d_start, d_len, n, row0, row1, kmer_size, is_containment,
                                    threshold, s_fixed, &el);
    (void)hipEventRecord(e1, ctx->stream);
    // (the count read-backs above have synchronised the stream: a packed batch whose run list broke its contract is known by now)
    if (st == RTC_OK) st = rtc_sticky_error(ctx);
  }
  const double tv1 = now();
  uint64_t nsel = 0;
  int rounds = 0;
  st = agree(st);
  const rtc_reduce_hook hook{hook_all_reduce, c};
  if (st == RTC_OK) st = rtc_msf_device(ctx, el.d_edges, el.m, d_len, n, is_containment, s_fixed, (c->size > 1 || c->nccl) ? &hook : nullptr, d_sel, &nsel, &rounds,
                                        true, s_fixed ? 0u : *std::max_element(h_len, h_len + n));
  const double tv2 = now();
  if (st == RTC_OK && nsel) {
    hipError_t e = hipMemcpyAsync(B.h_sel, d_sel, nsel * sizeof(rtc_cedge), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) st = rtc_fail(ctx, RTC_ERR_HIP, "forest read-back -> %s", hipGetErrorString(e));
  }
  if (st == RTC_OK) { (void)hipEventRecord(e2, ctx->stream); (void)hipEventSynchronize(e2); }
  if (stats && st == RTC_OK) {
    (void)hipEventElapsedTime(&stats->pair_ms, e0, e1);
    (void)hipEventElapsedTime(&stats->mst_ms, e1, e2);
    stats->row0 = row0; stats->row1 = row1; stats->cand_edges = el.m; stats->rounds = (uint32_t)rounds;
    stats->s_fixed = s_fixed; stats->contractions = (uint32_t)el.contractions;
  }
  const uint64_t m_edges = el.m;
  rtc_edge_list_free(&el, ctx);
  if (st != RTC_OK) return st;
  const double tv3 = now();
  RTC_TRY(rtc_edges_to_mst_host_fixed(B.h_sel, nsel, h_len, kmer_size, is_containment, s_fixed, h_edges_out));
  *h_n_edges = nsel;
  if (verbose)
    fprintf(stderr, "[mst]   rank %d of %d, %u sketches, rows [%u, %u): %llu candidate edges in %.4fs, forest (%d rounds, %llu edges) %.4fs, read-back "
            "%.4fs, host distances %.4fs\n", c->rank, c->size, n, row0, row1, (unsigned long long)m_edges, tv1 - tv0, rounds,
            (unsigned long long)nsel, tv2 - tv1, tv3 - tv2, now() - tv3);
  return RTC_OK;
}

}  // extern "C"

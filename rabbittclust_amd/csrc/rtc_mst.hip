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
//     When every sketch has the same size s (the fixed-size MinHash configs) the distance is
//     monotone in `common` alone and one u64 key  (s - common) | i | j  carries weight, edge and
//     count: one atomicMin pass and -- across GPUs -- ONE all-reduce(MIN) per round.
//   * The union step runs on the device too: every component hooks onto the component its minimum
//     edge leads to (the strict total order on keys leaves only 2-cycles, broken towards the smaller
//     root), the hooked forest is flattened by following the successor chains, and the chosen edges
//     are appended to the forest list.  A round is launch -> [all-reduce] -> launch, the host reads
//     back one counter.
//     Distances are evaluated on the host with the reference's expression order.
#include <math.h>
#include <time.h>

#include <sched.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <numeric>
#include <thread>
#include <vector>

#include "rtc_internal.h"

namespace {

constexpr uint64_t KEY_NONE = 0x7FFFFFFFFFFFFFFFULL;  // above every key (fused keys use 63 bits, edge ids need n < 2^31); also a valid int64

// Weight modes (the `is_containment` argument of the entry points): 0 = set Jaccard common / |A u B| (the index
// path, src/MST.cpp:1489-1503), 1 = containment common / min(|A|, |B|) (:1504-1515, and containDistance() of the
// dense loop), 2 | s << 2 = Mash's union-truncated estimator of the dense loop (MinHash::jaccard(), modifyMST
// src/MST.cpp:851-866): common among the first s union elements over  min(s, |A| + |B| - common)  -- the union
// elements Mash's merge has seen when it stops (s, or all of them when both lists run out first).
__host__ __device__ __forceinline__ uint64_t weight_denom(uint32_t common, uint32_t sa, uint32_t sb, int wmode) {
  if ((wmode & 3) == 1) return sa < sb ? sa : sb;
  const uint64_t u = (uint64_t)sa + sb - common;
  if ((wmode & 3) == 2) { const uint64_t s = (uint32_t)wmode >> 2; return u < s ? u : s; }
  return u;
}
// similarity key: smaller key == more similar == smaller distance
__device__ __forceinline__ uint64_t weight_key(uint32_t common, uint32_t sa, uint32_t sb, int wmode) {
  const uint64_t d = weight_denom(common, sa, sb, wmode);
  const double J = d ? (double)common / (double)d : 0.0;  // in [0,1]; IEEE division, correctly rounded
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

// The running minimum as the atomic unit has it: a load that goes to the L2, not to the CU's own cache.  The plain load it
// replaces could stay stale for a whole launch, and every edge of a large component then went through to the atomic unit,
// where atomics on one address queue up (~10 ns each: the third Boruvka round of the headline took 213 us of its 0.7 ms).
__device__ __forceinline__ unsigned long long peek_min(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// key[c] = min(key[c], k) for the lanes with `on`, called by whole waves.  Lanes that name the same component pool their
// values first (the edge list comes out of the pair phase column by column, so a wave's 64 edges touch a handful of
// components from the second round on): one atomic per component for the first four of a wave, lane by lane beyond.
__device__ __forceinline__ void wave_min_update(unsigned long long* __restrict__ key, uint32_t c, unsigned long long k, bool on) {
  const uint32_t lane = threadIdx.x & 63;
  // A lane whose key cannot lower its component's running minimum drops out before anything is pooled: the minimum settles after
  // the first few edges of a component, and from then on nearly every edge fails this one look (11.6 M edges of 100 000 genomes,
  // most of them chance collisions of unrelated genomes with one common hash: 200 -> 70 us a round).  The value read can only be
  // above the final minimum, so nothing that could still win is dropped.
  if (on) on = k < peek_min(&key[c]);
  uint64_t todo = __ballot(on);
  for (int it = 0; it < 4 && todo; it++) {  // (wave-uniform)
    const int lead = __builtin_ctzll(todo);
    const uint32_t c0 = (uint32_t)__shfl((int)c, lead);
    const bool mine = ((todo >> lane) & 1ULL) && c == c0;
    unsigned long long kk = mine ? k : ~0ULL;
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long t = __shfl_xor(kk, o); kk = t < kk ? t : kk; }
    if ((int)lane == lead && kk < peek_min(&key[c0])) atomicMin(&key[c0], kk);
    todo &= ~__ballot(mine);
  }
  if (((todo >> lane) & 1ULL) && k < peek_min(&key[c])) atomicMin(&key[c], k);
}

__global__ __launch_bounds__(256) void boruvka_minweight_kernel(const rtc_cedge* __restrict__ edges, uint64_t m,
                                                                const uint32_t* __restrict__ len, int is_containment,
                                                                const uint32_t* __restrict__ comp,
                                                                unsigned long long* __restrict__ wkey, const uint32_t* __restrict__ go = nullptr) {
  if (go && !*go) return;
  for (uint64_t e0 = (uint64_t)blockIdx.x * blockDim.x; e0 < m; e0 += (uint64_t)gridDim.x * blockDim.x) {  // whole waves stay together
    const uint64_t e = e0 + threadIdx.x;
    bool on = e < m;
    uint32_t ci = 0, cj = 0;
    unsigned long long key = 0;
    if (on) {
      const rtc_cedge ed = edges[e];
      ci = comp[ed.i]; cj = comp[ed.j];
      on = ci != cj;
      if (on) key = weight_key(ed.common, len[ed.i], len[ed.j], is_containment);
    }
    wave_min_update(wkey, ci, key, on);
    wave_min_update(wkey, cj, key, on);
  }
}

__global__ __launch_bounds__(256) void boruvka_minedge_kernel(const rtc_cedge* __restrict__ edges, uint64_t m,
                                                              const uint32_t* __restrict__ len, int is_containment,
                                                              const uint32_t* __restrict__ comp,
                                                              const unsigned long long* __restrict__ wkey,
                                                              unsigned long long* __restrict__ ekey, const uint32_t* __restrict__ go = nullptr,
                                                              int idx_bits = 0, int cbits = 0) {
  if (go && !*go) return;
  for (uint64_t e0 = (uint64_t)blockIdx.x * blockDim.x; e0 < m; e0 += (uint64_t)gridDim.x * blockDim.x) {  // whole waves stay together
    const uint64_t e = e0 + threadIdx.x;
    bool oi = false, oj = false;
    uint32_t ci = 0, cj = 0;
    unsigned long long id = 0;
    if (e < m) {
      const rtc_cedge ed = edges[e];
      ci = comp[ed.i]; cj = comp[ed.j];
      if (ci != cj) {
        const uint64_t key = weight_key(ed.common, len[ed.i], len[ed.j], is_containment);
        // cbits != 0: the id carries the count too, (i << B | j) << cbits | common -- the same order on (i, j), no third pass
        id = cbits ? (((((unsigned long long)ed.i << idx_bits) | ed.j) << cbits) | ed.common) : (((unsigned long long)ed.i << 32) | ed.j);
        oi = key == wkey[ci];
        oj = key == wkey[cj];
      }
    }
    wave_min_update(ekey, ci, id, oi);
    wave_min_update(ekey, cj, id, oj);
  }
}

__global__ __launch_bounds__(256) void boruvka_fetch_kernel(const rtc_cedge* __restrict__ edges, uint64_t m,
                                                            const uint32_t* __restrict__ comp,
                                                            const unsigned long long* __restrict__ ekey,
                                                            uint32_t* __restrict__ ecommon, const uint32_t* __restrict__ go = nullptr) {
  if (go && !*go) return;
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

// fixed-size mode: key = (s - common) << 2B | i << B | j   (B = index bits; smaller key = closer pair)
__global__ __launch_bounds__(256) void boruvka_minkey_kernel(const rtc_cedge* __restrict__ edges, uint64_t m,
                                                             const uint32_t* __restrict__ comp, uint32_t s_fixed,
                                                             int idx_bits, unsigned long long* __restrict__ key,
                                                             const uint32_t* __restrict__ go = nullptr) {
  if (go && !*go) return;
  for (uint64_t e0 = (uint64_t)blockIdx.x * blockDim.x; e0 < m; e0 += (uint64_t)gridDim.x * blockDim.x) {  // whole waves stay together
    const uint64_t e = e0 + threadIdx.x;
    bool on = e < m;
    uint32_t ci = 0, cj = 0;
    unsigned long long k = 0;
    if (on) {
      const rtc_cedge ed = edges[e];
      ci = comp[ed.i]; cj = comp[ed.j];
      on = ci != cj;
      k = ((unsigned long long)(s_fixed - ed.common) << (2 * idx_bits)) | ((unsigned long long)ed.i << idx_bits) | (unsigned long long)ed.j;
    }
    wave_min_update(key, ci, k, on);
    wave_min_update(key, cj, k, on);
  }
}

struct RoundKeys {
  const unsigned long long* key;  // fixed mode: fused key; variable mode: edge id (i << 32 | j)
  const uint32_t* ecommon;        // variable mode only
  uint32_t s_fixed;               // 0: variable mode
  int idx_bits;
  int cbits;                      // variable mode: != 0 when the edge id carries the count (key = (i << idx_bits | j) << cbits | common)
};

__device__ __forceinline__ bool round_edge(const RoundKeys& K, uint32_t c, uint32_t& i, uint32_t& j, uint32_t& common) {
  const unsigned long long k = K.key[c];
  if (k == KEY_NONE) return false;
  if (K.s_fixed) {
    const unsigned long long mask = (1ULL << K.idx_bits) - 1ULL;
    j = (uint32_t)(k & mask);
    i = (uint32_t)((k >> K.idx_bits) & mask);
    common = K.s_fixed - (uint32_t)(k >> (2 * K.idx_bits));
  } else if (K.cbits) {
    const unsigned long long mask = (1ULL << K.idx_bits) - 1ULL;
    common = (uint32_t)(k & ((1ULL << K.cbits) - 1ULL));
    j = (uint32_t)((k >> K.cbits) & mask);
    i = (uint32_t)(k >> (K.cbits + K.idx_bits));
  } else {
    i = (uint32_t)(k >> 32);
    j = (uint32_t)k;
    common = K.ecommon[c];
  }
  return true;
}

// Every root v with a minimum edge hooks onto the component at the edge's other end.  Two roots that
// picked each other picked the same edge (strict total order on keys): the smaller one stays a root
// and records the edge, the larger one hooks without recording it.
__global__ __launch_bounds__(256) void boruvka_hook_kernel(RoundKeys K, const uint32_t* __restrict__ comp, uint32_t n,
                                                           uint32_t* __restrict__ succ, rtc_cedge* __restrict__ sel,
                                                           unsigned long long* __restrict__ nsel,
                                                           uint32_t* __restrict__ added, const uint32_t* __restrict__ go) {
  if (go && !*go) return;  // the round before added nothing: the forest is complete (rounds are enqueued ahead of the host's look)
  // the forest counter is ONE address: a block reserves its edges with one atomic (a wave each cost 75 us of queueing at the
  // atomic unit in the first rounds of 200 000 vertices -- 3 125 waves, two atomics each)
  __shared__ uint32_t s_cnt[4];
  __shared__ unsigned long long s_base;
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint32_t v0 = blockIdx.x * blockDim.x; v0 < n; v0 += gridDim.x * blockDim.x) {  // whole blocks stay together
    const uint32_t v = v0 + threadIdx.x;
    uint32_t s = v, i = 0, j = 0, cm = 0;
    bool append = false;
    if (v < n && comp[v] == v && round_edge(K, v, i, j, cm)) {
      const uint32_t ci = comp[i], cj = comp[j];
      const uint32_t d = ci == v ? cj : ci;
      uint32_t i2, j2, cm2;
      bool mutual = false;
      if (round_edge(K, d, i2, j2, cm2)) {
        const uint32_t di = comp[i2], dj = comp[j2];
        mutual = (di == d ? dj : di) == v;
      }
      if (mutual && v < d) { s = v; append = true; }
      else { s = d; append = !mutual; }
    }
    if (v < n) succ[v] = s;
    const uint64_t bal = __ballot(append);
    if (lane == 0) s_cnt[wave] = (uint32_t)__popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) {
      const uint32_t tot = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
      s_base = tot ? atomicAdd(nsel, (unsigned long long)tot) : 0ull;
      if (tot) atomicAdd(added, tot);
    }
    __syncthreads();
    if (append) {
      uint32_t before = 0;
      for (uint32_t w = 0; w < wave; w++) before += s_cnt[w];
      sel[s_base + before + (uint64_t)__popcll(bal & ((1ULL << lane) - 1ULL))] = rtc_cedge{i, j, cm};
    }
    __syncthreads();
  }
}

// comp[v] <- root of comp[v] in the hooked forest (successor chains end in a root with succ[r] == r)
__global__ __launch_bounds__(256) void boruvka_relabel_kernel(uint32_t* __restrict__ comp, const uint32_t* __restrict__ succ,
                                                              uint32_t n) {
  for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < n; v += gridDim.x * blockDim.x) {
    uint32_t r = comp[v];
    while (true) { const uint32_t nx = succ[r]; if (nx == r) break; r = nx; }
    comp[v] = r;
  }
}

// one-GPU rounds (rtc_msf_device without an all-reduce hook): the per-vertex passes are folded together.
// begin: comp = identity, keys empty.  relabel_reset: the relabel pass of round r, the key reset of round r + 1 and the
// zeroing of round r + 1's "edges added" counter (two counters alternate, the host still reads round r's).
__global__ __launch_bounds__(256) void boruvka_begin_kernel(uint32_t* __restrict__ comp, unsigned long long* __restrict__ wkey,
                                                            unsigned long long* __restrict__ ekey, uint32_t n) {
  for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < n; v += gridDim.x * blockDim.x) {
    comp[v] = v;
    wkey[v] = KEY_NONE;
    if (ekey) ekey[v] = KEY_NONE;
  }
}
__global__ __launch_bounds__(256) void boruvka_relabel_reset_kernel(uint32_t* __restrict__ comp, const uint32_t* __restrict__ succ,
                                                                    uint32_t n, unsigned long long* __restrict__ wkey,
                                                                    unsigned long long* __restrict__ ekey,
                                                                    const uint32_t* __restrict__ go) {
  if (go && !*go) return;
  for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < n; v += gridDim.x * blockDim.x) {
    uint32_t r = comp[v];
    while (true) { const uint32_t nx = succ[r]; if (nx == r) break; r = nx; }
    comp[v] = r;
    wkey[v] = KEY_NONE;
    if (ekey) ekey[v] = KEY_NONE;
  }
}

__global__ void iota_u32_kernel(uint32_t* p, uint32_t n) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = i;
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

// src/MST.cpp:1295,1489-1515 with the reference's operation order (host libm log); mode 2: Mash's
// MinHash::distance() (-ln(2j/(1+j))/k, 1 when j = 0, never above 1)
static double host_mst_distance(int common, int size0, int size1, int kmer_size, int wmode) {
  const double inv_kmer_size = 1.0 / kmer_size;
  if ((wmode & 3) == 2) {
    const uint64_t d = weight_denom((uint32_t)common, (uint32_t)size0, (uint32_t)size1, wmode);
    const double j = d ? (double)common / (double)d : 0.0;
    if (j == 0.0) return 1.0;
    if (j == 1.0) return 0.0;
    const double dist = -log(2.0 * j / (1.0 + j)) / kmer_size;
    return dist > 1.0 ? 1.0 : dist;
  }
  if (!wmode) {
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

// the union step of a round (hook + relabel + the counter read back): rtc_boruvka_union_dev and the multi-GPU rounds
static int boruvka_union_round(rtc_ctx* ctx, uint32_t n, const RoundKeys& K, uint32_t* d_comp, uint32_t* d_succ, rtc_cedge* d_sel,
                               uint64_t* d_nsel, uint32_t* h_added) {
  uint32_t* d_added = (uint32_t*)(d_nsel + 1);
  RTC_HIP(ctx, hipMemsetAsync(d_added, 0, 4, ctx->stream));
  hipLaunchKernelGGL(boruvka_hook_kernel, dim3(grid_for(n, ctx->num_cu)), dim3(256), 0, ctx->stream, K, (const uint32_t*)d_comp, n, d_succ,
                     d_sel, (unsigned long long*)d_nsel, d_added, (const uint32_t*)nullptr);
  RTC_CHECK_LAUNCH(ctx);
  hipLaunchKernelGGL(boruvka_relabel_kernel, dim3(grid_for(n, ctx->num_cu)), dim3(256), 0, ctx->stream, d_comp, (const uint32_t*)d_succ, n);
  RTC_CHECK_LAUNCH(ctx);
  void* hp = nullptr;
  RTC_TRY(rtc_pinned(ctx, 64, &hp));
  RTC_HIP(ctx, hipMemcpyAsync(hp, d_added, 4, hipMemcpyDeviceToHost, ctx->stream));
  RTC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *h_added = *(const uint32_t*)hp;
  return RTC_OK;
}

// scratch slot 3 of a forest call: two u64 and three u32 per vertex, the forest counter and the rounds' counters
size_t rtc_msf_scratch_bytes(uint32_t n) { return (((size_t)n * (8 + 8 + 4 + 4 + 4) + 64 + 8 + 4 * 64 + 64) + 255) & ~(size_t)255; }

// ---- host threads of the distance loop: a pool that lives with the process ----
// Spawning eight threads per call cost more than the 200 000 logarithms they shared (2.2 ms of a 4.7 ms MST phase); the pool's
// workers sleep on a condition variable between calls.  Size: the cores this process may really use (affinity mask and cgroup-v2
// CPU quota -- the GPU boxes show 256 CPUs under a 16-CPU quota), at most 16, the caller being one of them.
namespace {
unsigned host_usable_cores() {
  unsigned n = std::max(1u, std::thread::hardware_concurrency());
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min<unsigned>(n, (unsigned)std::max(1, CPU_COUNT(&set)));
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char q[32] = {0};
    unsigned long long period = 0;
    if (fscanf(f, "%31s %llu", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
      const unsigned long long quota = strtoull(q, nullptr, 10);
      if (quota > 0) n = std::min<unsigned>(n, (unsigned)std::max<unsigned long long>(1, (quota + period - 1) / period));
    }
    fclose(f);
  }
  return n;
}
class HostPool {
 public:
  static HostPool& get() { static HostPool p; return p; }
  unsigned size() const { return (unsigned)th_.size() + 1; }
  // fn(part, parts) on every member of the pool (the caller is part 0); returns when all are done.  One caller at a time.
  void run(const std::function<void(unsigned, unsigned)>& fn) {
    std::lock_guard<std::mutex> one(call_);
    const unsigned parts = size();
    {
      std::lock_guard<std::mutex> lk(m_);
      fn_ = &fn; pending_ = parts - 1; gen_++;
    }
    cv_.notify_all();
    fn(0, parts);
    std::unique_lock<std::mutex> lk(m_);
    done_.wait(lk, [&] { return pending_ == 0; });
    fn_ = nullptr;
  }
 private:
  HostPool() {
    const unsigned n = std::min(16u, host_usable_cores());
    for (unsigned t = 1; t < n; t++) th_.emplace_back([this, t] { work(t); });
  }
  ~HostPool() {
    { std::lock_guard<std::mutex> lk(m_); stop_ = true; }
    cv_.notify_all();
    for (auto& t : th_) t.join();
  }
  void work(unsigned part) {
    uint64_t seen = 0;
    for (;;) {
      const std::function<void(unsigned, unsigned)>* fn;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
        if (stop_) return;
        seen = gen_;
        fn = fn_;
      }
      (*fn)(part, size());
      { std::lock_guard<std::mutex> lk(m_); if (--pending_ == 0) done_.notify_one(); }
    }
  }
  std::vector<std::thread> th_;
  std::mutex m_, call_;
  std::condition_variable cv_, done_;
  const std::function<void(unsigned, unsigned)>* fn_ = nullptr;
  unsigned pending_ = 0;
  uint64_t gen_ = 0;
  bool stop_ = false;
};
}  // namespace

// rtc_edges_to_mst_host with the caller's knowledge that every sketch holds s_fixed hashes (0: sizes vary): the distance is then a
// function of `common` alone and comes from a table of s_fixed + 1 host-libm values -- the same doubles, one log() per distinct
// count instead of one per edge.
int rtc_edges_to_mst_host_fixed(const rtc_cedge* h_sel, uint64_t m, const uint32_t* h_len, int kmer_size, int is_containment,
                                uint32_t s_fixed, rtc_edge* h_out) {
  if ((m && (!h_sel || !h_out)) || !h_len) return RTC_ERR_ARG;
  std::vector<double> table;
  if (s_fixed && s_fixed <= (1u << 20) && m > 64) {
    table.resize((size_t)s_fixed + 1);
    for (uint32_t c = 0; c <= s_fixed; c++) table[c] = host_mst_distance((int)c, (int)s_fixed, (int)s_fixed, kmer_size, is_containment);
  }
  auto fill = [&](uint64_t e0, uint64_t e1) {
    for (uint64_t e = e0; e < e1; e++) {
      h_out[e].preNode = (int32_t)h_sel[e].i;
      h_out[e].sufNode = (int32_t)h_sel[e].j;
      h_out[e].dist = !table.empty() && h_sel[e].common <= s_fixed
                          ? table[h_sel[e].common]
                          : host_mst_distance((int)h_sel[e].common, (int)h_len[h_sel[e].i], (int)h_len[h_sel[e].j], kmer_size, is_containment);
    }
  };
  auto less = [](const rtc_edge& a, const rtc_edge& b) {
    if (a.dist != b.dist) return a.dist < b.dist;
    if (a.preNode != b.preNode) return a.preNode < b.preNode;
    return a.sufNode < b.sufNode;
  };
  // the distances are the host libm's (the reference's); beyond a few thousand edges the loop is shared by the pool's threads,
  // each of which also checks the order of its own stretch (forests that come from the device are already in the reference's
  // output order, rtc_sort.hip: a check instead of a sort)
  const uint64_t per = table.empty() ? 4096 : 32768;
  std::atomic<int> unsorted{0};
  if (m <= per) {
    fill(0, m);
    if (!std::is_sorted(h_out, h_out + m, less)) unsorted = 1;
  } else {
    HostPool::get().run([&](unsigned part, unsigned parts) {
      const uint64_t e0 = m * part / parts, e1 = m * (part + 1) / parts;
      fill(e0, e1);
      if (!std::is_sorted(h_out + e0, h_out + e1, less)) unsorted = 1;
    });
    const unsigned parts = HostPool::get().size();
    for (unsigned p = 1; p < parts && !unsorted; p++) {  // the seams
      const uint64_t e = m * p / parts;
      if (e > 0 && e < m && less(h_out[e], h_out[e - 1])) unsorted = 1;
    }
  }
  if (unsorted) std::sort(h_out, h_out + m, less);
  return RTC_OK;
}

// Page-locked staging and device scratch of a whole-MST call, from the context's own pools (no hipMalloc / hipFree and no
// pageable copies per call -- together 0.8 ms of the 4.2 ms a 200 000-genome forest took): the sketch sizes and the forest on the
// host side, the forest list on the device side (behind rtc_msf_device's arrays in scratch slot 3).
int rtc_mst_bufs(rtc_ctx* ctx, uint32_t n, rtc_mst_bufs_t* b) {
  const size_t b_len = ((size_t)n * 4 + 255) & ~(size_t)255;
  const size_t want = b_len + (size_t)n * sizeof(rtc_cedge) + 256;  // a block of its own: the shared staging block (rtc_pinned) is rewritten by the pair phase's planners
  if (want > ctx->mst_pinned_bytes) {
    if (ctx->mst_pinned) RTC_HIP(ctx, hipHostFree(ctx->mst_pinned));
    ctx->mst_pinned = nullptr; ctx->mst_pinned_bytes = 0;
    RTC_HIP(ctx, hipHostMalloc(&ctx->mst_pinned, want + want / 4, hipHostMallocDefault));
    ctx->mst_pinned_bytes = want + want / 4;
  }
  b->h_len = (uint32_t*)ctx->mst_pinned;
  b->h_sel = (rtc_cedge*)((char*)ctx->mst_pinned + b_len);
  void* ws3 = nullptr;
  RTC_TRY(rtc_ws(ctx, 3, rtc_msf_scratch_bytes(n) + (size_t)n * sizeof(rtc_cedge) + 256, &ws3));
  b->d_sel = (rtc_cedge*)((char*)ws3 + rtc_msf_scratch_bytes(n));
  return RTC_OK;
}

// ---- minimum spanning forest of a device-resident candidate list (shared by rtc_mst and the
// multi-GPU step; `hook` all-reduces the per-round key arrays across ranks, NULL on one GPU) ----
int rtc_msf_device(rtc_ctx* ctx, const rtc_cedge* d_edges, uint64_t m, const uint32_t* d_len, uint32_t n,
                   int is_containment, uint32_t s_fixed, const rtc_reduce_hook* hook, rtc_cedge* d_sel,
                   uint64_t* n_sel_out, int* rounds_out, bool sorted, uint32_t max_len) {
  *n_sel_out = 0;
  if (rounds_out) *rounds_out = 0;
  if (n < 2) return RTC_OK;
  if (n >= (1u << 31)) return rtc_fail(ctx, RTC_ERR_UNSUPPORTED, "%u sketches: edge ids (i << 32 | j) must stay below the empty key 2^63 - 1", n);
  int idx_bits = rtc_boruvka_key_bits(n, s_fixed);
  if (s_fixed && !idx_bits) s_fixed = 0;
  // Sketches of different sizes (--fast, containment): the weight needs its own 64 bits, but the edge id that breaks ties can
  // carry the count when the caller knows the longest sketch: two vertex indices + the count in 63 bits (200 000 sketches of up
  // to 1 000 000 hashes: 18 + 18 + 20).  Then a round is two passes over the edges instead of three -- and two all-reduces
  // across GPUs instead of three, which at eight GPUs are the larger part of a round.
  int cbits = 0;
  if (!s_fixed && max_len) {
    int b = 1;
    while (b < 32 && (1ull << b) < (uint64_t)n) b++;
    int w = 1;
    while (w < 32 && (1ull << w) <= (uint64_t)max_len) w++;
    if (2 * b + w <= 63) { idx_bits = b; cbits = w; }
  }
  void* ws3 = nullptr;
  RTC_TRY(rtc_ws(ctx, 3, rtc_msf_scratch_bytes(n), &ws3));
  uint64_t* d_wkey = (uint64_t*)ws3;
  uint64_t* d_ekey = d_wkey + n;
  uint32_t* d_ecommon = (uint32_t*)(d_ekey + n);
  uint32_t* d_comp = d_ecommon + n;
  uint32_t* d_succ = d_comp + n;
  uint64_t* d_nsel = (uint64_t*)(((uintptr_t)(d_succ + n) + 63) & ~(uintptr_t)63);
  if (!hook) {  // one GPU: three (fixed sizes) or five (variable) kernels per round, the host looks in every third round
    unsigned long long* wkey = (unsigned long long*)d_wkey;
    unsigned long long* ekey = s_fixed ? nullptr : (unsigned long long*)d_ekey;
    // d_added[r] = forest edges round r added.  Rounds are enqueued in groups of GROUP without a host round trip between them:
    // every kernel of round r > 0 first looks at d_added[r - 1] and returns at once when it is 0 (the forest was complete), so
    // the rounds enqueued past the end cost a few empty launches instead of a synchronisation per round (20 us each, a quarter
    // of a round at 200 000 vertices).
    uint32_t* d_added = (uint32_t*)(d_nsel + 1);
    constexpr int MAX_ROUNDS = 64, GROUP = 3;
    const dim3 gn(grid_for(n, ctx->num_cu)), gm(grid_for(std::max<uint64_t>(m, 1), ctx->num_cu)), blk(256);
    RTC_HIP(ctx, hipMemsetAsync(d_nsel, 0, 8 + 4 * MAX_ROUNDS, ctx->stream));
    hipLaunchKernelGGL(boruvka_begin_kernel, gn, blk, 0, ctx->stream, d_comp, wkey, ekey, n);
    RTC_CHECK_LAUNCH(ctx);
    void* hp = nullptr;
    RTC_TRY(rtc_pinned(ctx, 8 + 4 * MAX_ROUNDS, &hp));
    const RoundKeys K{s_fixed ? wkey : ekey, d_ecommon, s_fixed, idx_bits, cbits};
    int rounds = 0;
    for (int round = 0; round < MAX_ROUNDS && !rounds;) {
      const int first = round;
      for (int g = 0; g < (first == 0 ? 2 : GROUP) && round < MAX_ROUNDS; g++, round++) {
        const uint32_t* go = round ? d_added + round - 1 : nullptr;
        if (m) {
          if (s_fixed) {
            hipLaunchKernelGGL(boruvka_minkey_kernel, gm, blk, 0, ctx->stream, d_edges, m, (const uint32_t*)d_comp, s_fixed, idx_bits, wkey, go);
          } else {
            hipLaunchKernelGGL(boruvka_minweight_kernel, gm, blk, 0, ctx->stream, d_edges, m, d_len, is_containment, (const uint32_t*)d_comp, wkey, go);
            hipLaunchKernelGGL(boruvka_minedge_kernel, gm, blk, 0, ctx->stream, d_edges, m, d_len, is_containment, (const uint32_t*)d_comp,
                               (const unsigned long long*)wkey, ekey, go, idx_bits, cbits);
            if (!cbits)
              hipLaunchKernelGGL(boruvka_fetch_kernel, gm, blk, 0, ctx->stream, d_edges, m, (const uint32_t*)d_comp,
                                 (const unsigned long long*)ekey, d_ecommon, go);
          }
          RTC_CHECK_LAUNCH(ctx);
        }
        hipLaunchKernelGGL(boruvka_hook_kernel, gn, blk, 0, ctx->stream, K, (const uint32_t*)d_comp, n, d_succ, d_sel,
                           (unsigned long long*)d_nsel, d_added + round, go);
        RTC_CHECK_LAUNCH(ctx);
        hipLaunchKernelGGL(boruvka_relabel_reset_kernel, gn, blk, 0, ctx->stream, d_comp, (const uint32_t*)d_succ, n, wkey, ekey, go);
        RTC_CHECK_LAUNCH(ctx);
      }
      RTC_HIP(ctx, hipMemcpyAsync(hp, d_nsel, 8 + 4 * (size_t)round, hipMemcpyDeviceToHost, ctx->stream));
      RTC_HIP(ctx, hipStreamSynchronize(ctx->stream));
      *n_sel_out = ((const unsigned long long*)hp)[0];
      const uint32_t* added = (const uint32_t*)((const char*)hp + 8);
      for (int r = first; r < round; r++)
        if (!added[r]) { rounds = r + 1; break; }  // round r found no edge that leaves a component: it was the last one that ran
    }
    if (!rounds) rounds = MAX_ROUNDS;
    if (rounds_out) *rounds_out = rounds;
    if (sorted) RTC_TRY(rtc_sort_forest_device(ctx, d_sel, *n_sel_out, d_len, is_containment));
    return RTC_OK;
  }
  RTC_TRY(rtc_boruvka_init_dev(ctx, n, d_comp, d_nsel));
  int rounds = 0;
  for (int round = 0; round < 64; round++) {
    if (s_fixed) {
      RTC_TRY(rtc_boruvka_minkey_dev(ctx, d_edges, m, d_comp, n, s_fixed, d_wkey));
      if (hook) RTC_TRY(hook->all_reduce(hook->self, d_wkey, n, 2, 0));
    } else if (cbits) {  // the edge id carries the count: MIN(weight), MIN(id) -- no third exchange
      RTC_TRY(rtc_boruvka_minweight_dev(ctx, d_edges, m, d_len, is_containment, d_comp, n, d_wkey));
      if (hook) RTC_TRY(hook->all_reduce(hook->self, d_wkey, n, 2, 0));
      hipLaunchKernelGGL(fill_u64_kernel, dim3(grid_for(n, ctx->num_cu)), dim3(256), 0, ctx->stream, (unsigned long long*)d_ekey, (uint64_t)n,
                         (unsigned long long)KEY_NONE);
      if (m)
        hipLaunchKernelGGL(boruvka_minedge_kernel, dim3(grid_for(m, ctx->num_cu)), dim3(256), 0, ctx->stream, d_edges, m, d_len, is_containment,
                           (const uint32_t*)d_comp, (const unsigned long long*)d_wkey, (unsigned long long*)d_ekey, (const uint32_t*)nullptr,
                           idx_bits, cbits);
      RTC_CHECK_LAUNCH(ctx);
      if (hook) RTC_TRY(hook->all_reduce(hook->self, d_ekey, n, 2, 0));
    } else {
      RTC_TRY(rtc_boruvka_minweight_dev(ctx, d_edges, m, d_len, is_containment, d_comp, n, d_wkey));
      if (hook) RTC_TRY(hook->all_reduce(hook->self, d_wkey, n, 2, 0));
      RTC_TRY(rtc_boruvka_minedge_dev(ctx, d_edges, m, d_len, is_containment, d_comp, n, d_wkey, d_ekey));
      if (hook) RTC_TRY(hook->all_reduce(hook->self, d_ekey, n, 2, 0));
      RTC_TRY(rtc_boruvka_fetch_dev(ctx, d_edges, m, d_comp, n, d_ekey, d_ecommon));
      if (hook) RTC_TRY(hook->all_reduce(hook->self, d_ecommon, n, 1, 1));
    }
    uint32_t added = 0;
    RTC_TRY(boruvka_union_round(ctx, n, RoundKeys{(const unsigned long long*)(s_fixed ? d_wkey : d_ekey), d_ecommon, s_fixed, idx_bits, cbits},
                                d_comp, d_succ, d_sel, d_nsel, &added));
    rounds++;
    if (!added) break;
  }
  unsigned long long ns = 0;
  RTC_HIP(ctx, hipMemcpyAsync(&ns, d_nsel, 8, hipMemcpyDeviceToHost, ctx->stream));
  RTC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *n_sel_out = ns;
  if (rounds_out) *rounds_out = rounds;
  if (sorted) RTC_TRY(rtc_sort_forest_device(ctx, d_sel, ns, d_len, is_containment));
  return RTC_OK;
}

// ---- candidate edges of rows [row0, row1) (columns j < i), device list with growth / contraction ----
// The list is produced optimistically in one launch; if it would exceed the edge budget (dense
// inputs: same-species collections) the rows are walked in chunks and the list is contracted to
// its own minimum spanning forest whenever it passes half the budget -- the reference bounds its
// memory the same way (sort + Kruskal per row block, src/MST.cpp:1543-1546); the forest of a union is the
// forest of the parts' forests.
int rtc_candidate_edges_device(rtc_ctx* ctx, const void* d_hashes, int width, const uint64_t* d_start,
                               const uint32_t* d_len, uint32_t n, uint32_t row0, uint32_t row1, int kmer_size,
                               int is_containment, double threshold, uint32_t s_fixed, rtc_edge_list* el,
                               const rtc_edge_observer* obs) {
  const int radio = (int)(2.0 * exp(threshold * (kmer_size - 1)) - 1.0);  // src/MST.cpp:26-37,1292
  uint64_t budget = (uint64_t)256 << 20;  // edges (3 GiB)
  if (ctx->opt.edge_budget) budget = ctx->opt.edge_budget;  // tests of the dense path
  budget = std::max<uint64_t>(budget, 66ull * n + 1024);  // a 64-row block on top of a contracted list always fits
  el->m = 0;
  if (row1 <= std::max<uint32_t>(row0, 1)) return RTC_OK;
  row0 = std::max<uint32_t>(row0, 1);
  if (!el->d_edges && !el->d_count && ctx->edge_cache) {  // the list the context kept from its last clustering call
    el->d_edges = (rtc_cedge*)ctx->edge_cache; el->cap = ctx->edge_cache_cap; el->d_count = (unsigned long long*)ctx->edge_cache_count;
    ctx->edge_cache = nullptr; ctx->edge_cache_cap = 0; ctx->edge_cache_count = nullptr;
  }
  if (!el->d_count) RTC_HIP(ctx, hipMalloc((void**)&el->d_count, 64));
  auto ensure = [&](uint64_t want) -> int {
    if (want <= el->cap) return RTC_OK;
    rtc_cedge* nd = nullptr;
    hipError_t e = hipMalloc((void**)&nd, want * sizeof(rtc_cedge));
    if (e != hipSuccess) return rtc_fail(ctx, RTC_ERR_NOMEM, "candidate edge list of %llu edges (%.1f GB): %s",
                                         (unsigned long long)want, want * 12e-9, hipGetErrorString(e));
    if (el->m) RTC_HIP(ctx, hipMemcpyAsync(nd, el->d_edges, el->m * sizeof(rtc_cedge), hipMemcpyDeviceToDevice, ctx->stream));
    RTC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (el->d_edges) (void)hipFree(el->d_edges);
    el->d_edges = nd; el->cap = want;
    ctx->free_hbm_at = -1.0;  // (allocated and freed behind rtc_free_hbm's back: the pair phase sizes its scratch from it)
    return RTC_OK;
  };
  // the first list: 160 edges a genome (BASELINE configs[2]: 116 at 100 000 genomes of families of ten, 190 MB), a one-shot run's only launch then fits
  RTC_TRY(ensure(std::min<uint64_t>(budget, std::max<uint64_t>((uint64_t)1 << 20, (uint64_t)n * 160))));
  auto run_rows = [&](uint32_t r0, uint32_t r1, unsigned long long* cnt_out) -> int {
    unsigned long long mm = el->m;
    RTC_HIP(ctx, hipMemcpyAsync(el->d_count, &mm, 8, hipMemcpyHostToDevice, ctx->stream));
    RTC_TRY(rtc_pair_edges_dev(ctx, d_hashes, width, d_start, d_len, n, r0, r1, 0, r1 - 1, radio, el->d_edges, el->cap,
                               (uint64_t*)el->d_count));
    RTC_HIP(ctx, hipMemcpyAsync(cnt_out, el->d_count, 8, hipMemcpyDeviceToHost, ctx->stream));
    RTC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RTC_OK;
  };
  unsigned long long cnt = 0;
  RTC_TRY(run_rows(row0, row1, &cnt));
  // A count past the capacity is either exact (the kernel ran and dropped what did not fit) or the join's ESTIMATE from its
  // density sample (nothing ran, rtc_pair_edges_dev): grow to it and launch again -- the repeated call always runs to the end
  // and counts exactly, and when the estimate was short that exact count can be past the grown list too (or past the
  // budget: the chunked path below).  Two rounds settle every case; the bound only keeps a broken counter from looping.
  for (int redo = 0; cnt > el->cap && cnt <= budget; redo++) {
    if (redo == 3) return rtc_fail(ctx, RTC_ERR_HIP, "candidate edge count %llu still past the list (%llu) after three launches", cnt, (unsigned long long)el->cap);
    RTC_TRY(ensure(std::min<uint64_t>(budget, cnt + cnt / 16)));
    RTC_TRY(run_rows(row0, row1, &cnt));
  }
  if (cnt <= el->cap) {
    el->m = cnt;
    if (obs && cnt) RTC_TRY(obs->on_new(obs->self, el->d_edges, cnt));
    return RTC_OK;
  }
  // ---- dense input: row chunks + contraction ----
  RTC_TRY(ensure(budget));
  rtc_cedge* d_sel = nullptr;
  RTC_HIP(ctx, hipMalloc((void**)&d_sel, (size_t)n * sizeof(rtc_cedge)));
  int st = RTC_OK;
  ctx->pair_plan_hold = 1;  // the sketches do not change between the chunk launches: build the plan once
  ctx->pair_plan_valid = 0;
  ctx->pair_plan_tc1_hint = row1 - 1;
  uint32_t r0 = row0;
  uint32_t rows_per = (uint32_t)std::max<uint64_t>(64, (budget / 4) / std::max<uint32_t>(row1, 1) / 64 * 64);
  auto contract = [&]() -> int {
    uint64_t ns = 0;
    RTC_TRY(rtc_msf_device(ctx, el->d_edges, el->m, d_len, n, is_containment, s_fixed, nullptr, d_sel, &ns, nullptr, false));
    RTC_HIP(ctx, hipMemcpyAsync(el->d_edges, d_sel, ns * sizeof(rtc_cedge), hipMemcpyDeviceToDevice, ctx->stream));
    el->m = ns;
    el->contractions++;
    ctx->diag[3]++;
    return RTC_OK;
  };
  while (r0 < row1 && st == RTC_OK) {
    const uint32_t r1 = std::min<uint32_t>(row1, r0 + rows_per);
    st = run_rows(r0, r1, &cnt);
    if (st != RTC_OK) break;
    if (cnt > el->cap) {  // the chunk does not fit on top of the list: contract the list, then shrink the chunk
      if (el->m >= n) { st = contract(); continue; }
      if (rows_per > 64) { rows_per = std::max<uint32_t>(64, rows_per / 2 / 64 * 64); continue; }
      st = rtc_fail(ctx, RTC_ERR_NOMEM, "edge budget %llu too small for a 64-row block", (unsigned long long)budget);
      break;
    }
    if (obs && cnt > el->m) { st = obs->on_new(obs->self, el->d_edges + el->m, cnt - el->m); if (st != RTC_OK) break; }
    el->m = cnt;
    r0 = r1;
    if (el->m > budget / 2 && r0 < row1) st = contract();
  }
  ctx->pair_plan_hold = 0;
  ctx->pair_plan_valid = 0;
  (void)hipStreamSynchronize(ctx->stream);
  (void)hipFree(d_sel);
  return st;
}

void rtc_edge_list_free(rtc_edge_list* el, rtc_ctx* ctx) {
  if (ctx && el->d_edges && el->d_count && el->cap * sizeof(rtc_cedge) <= ((size_t)1 << 30)) {
    if (ctx->edge_cache) (void)hipFree(ctx->edge_cache);
    if (ctx->edge_cache_count) (void)hipFree(ctx->edge_cache_count);
    ctx->edge_cache = el->d_edges; ctx->edge_cache_cap = el->cap; ctx->edge_cache_count = el->d_count;
    *el = rtc_edge_list{};
    return;
  }
  if (el->d_edges) (void)hipFree(el->d_edges);
  if (el->d_count) (void)hipFree(el->d_count);
  if (ctx) ctx->free_hbm_at = -1.0;
  *el = rtc_edge_list{};
}

// all sketches the same size (and the fused key fits)?  0 otherwise
uint32_t rtc_fixed_size_of(const uint32_t* h_len, uint32_t n) {
  if (!n || !h_len[0]) return 0;
  for (uint32_t i = 1; i < n; i++) if (h_len[i] != h_len[0]) return 0;
  return rtc_boruvka_key_bits(n, h_len[0]) ? h_len[0] : 0;
}

namespace {
__global__ __launch_bounds__(256) void minmax_u32_kernel(const uint32_t* __restrict__ a, uint32_t n, uint32_t* __restrict__ mm) {
  uint32_t lo = 0xffffffffu, hi = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) { lo = min(lo, a[i]); hi = max(hi, a[i]); }
  for (int o = 32; o > 0; o >>= 1) { lo = min(lo, (uint32_t)__shfl_xor((int)lo, o)); hi = max(hi, (uint32_t)__shfl_xor((int)hi, o)); }
  if ((threadIdx.x & 63) == 0) { atomicMin(&mm[0], lo); atomicMax(&mm[1], hi); }
}
}  // namespace

namespace {
// every pair (row, col < row) of a dense tile becomes an edge (the dense loop has no filters): edge of (row, col) at
// base + row (row - 1) / 2 - row0 (row0 - 1) / 2 + col
__global__ __launch_bounds__(256) void all_pairs_edges_kernel(const uint32_t* __restrict__ common, uint64_t ld, uint32_t row0,
                                                              uint32_t row1, rtc_cedge* __restrict__ edges) {
  const uint64_t first = (uint64_t)row0 * (row0 - (row0 ? 1 : 0)) / 2;
  for (uint32_t row = row0 + blockIdx.y; row < row1; row += gridDim.y) {
    const uint64_t o = (uint64_t)row * (row - (row ? 1 : 0)) / 2 - first;
    for (uint32_t col = blockIdx.x * blockDim.x + threadIdx.x; col < row; col += gridDim.x * blockDim.x)
      edges[o + col] = rtc_cedge{row, col, common[(uint64_t)(row - row0) * ld + col]};
  }
}
}  // namespace

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

int rtc_boruvka_key_bits(uint32_t n, uint32_t s_fixed) {
  if (!s_fixed || n < 2) return 0;
  int b = 1;
  while (b < 32 && (1ull << b) < (uint64_t)n) b++;
  int w = 1;
  while (w < 32 && (1ull << w) <= (uint64_t)s_fixed) w++;
  return (w + 2 * b <= 63) ? b : 0;
}

int rtc_boruvka_init_dev(rtc_ctx* ctx, uint32_t n, uint32_t* d_comp, uint64_t* d_nsel) {
  if (!ctx || !d_comp || !d_nsel) return RTC_ERR_ARG;
  RTC_HIP(ctx, hipSetDevice(ctx->device));
  if (n) {
    hipLaunchKernelGGL(iota_u32_kernel, dim3(grid_for(n, ctx->num_cu)), dim3(256), 0, ctx->stream, d_comp, n);
    RTC_CHECK_LAUNCH(ctx);
  }
  RTC_HIP(ctx, hipMemsetAsync(d_nsel, 0, 16, ctx->stream));  // [0] = forest size, [1] = edges added by the last round
  return RTC_OK;
}

int rtc_boruvka_minkey_dev(rtc_ctx* ctx, const rtc_cedge* d_edges, uint64_t m, const uint32_t* d_comp, uint32_t n,
                           uint32_t s_fixed, uint64_t* d_key) {
  if (!ctx || !d_comp || !d_key || (m && !d_edges)) return RTC_ERR_ARG;
  const int idx_bits = rtc_boruvka_key_bits(n, s_fixed);
  if (!idx_bits) return rtc_fail(ctx, RTC_ERR_ARG, "fused Boruvka key does not fit: n=%u s=%u", n, s_fixed);
  RTC_HIP(ctx, hipSetDevice(ctx->device));
  hipLaunchKernelGGL(fill_u64_kernel, dim3(grid_for(n, ctx->num_cu)), dim3(256), 0, ctx->stream,
                     (unsigned long long*)d_key, (uint64_t)n, (unsigned long long)KEY_NONE);
  RTC_CHECK_LAUNCH(ctx);
  if (m) {
    hipLaunchKernelGGL(boruvka_minkey_kernel, dim3(grid_for(m, ctx->num_cu)), dim3(256), 0, ctx->stream, d_edges, m, d_comp,
                       s_fixed, idx_bits, (unsigned long long*)d_key);
    RTC_CHECK_LAUNCH(ctx);
  }
  return RTC_OK;
}

int rtc_boruvka_union_dev(rtc_ctx* ctx, uint32_t n, uint32_t s_fixed, const uint64_t* d_key, const uint32_t* d_ecommon,
                          uint32_t* d_comp, uint32_t* d_succ, rtc_cedge* d_sel, uint64_t* d_nsel, uint32_t* h_added) {
  if (!ctx || !d_key || !d_comp || !d_succ || !d_sel || !d_nsel || !h_added || (!s_fixed && !d_ecommon)) return RTC_ERR_ARG;
  *h_added = 0;
  if (!n) return RTC_OK;
  RTC_HIP(ctx, hipSetDevice(ctx->device));
  RoundKeys K{(const unsigned long long*)d_key, d_ecommon, s_fixed, rtc_boruvka_key_bits(n, s_fixed), 0};
  if (s_fixed && !K.idx_bits) return rtc_fail(ctx, RTC_ERR_ARG, "fused Boruvka key does not fit: n=%u s=%u", n, s_fixed);
  return boruvka_union_round(ctx, n, K, d_comp, d_succ, d_sel, d_nsel, h_added);
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
  return rtc_edges_to_mst_host_fixed(h_sel, m, h_len, kmer_size, is_containment, 0, h_out);
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
  return rtc_mst_append(ctx, d_hashes, width, d_start, d_len, n, 0, kmer_size, is_containment, threshold, h_edges_out, h_n_edges);
}

int rtc_mst_append(rtc_ctx* ctx, const void* d_hashes, int width, const uint64_t* d_start, const uint32_t* d_len,
                   uint32_t n, uint32_t start_index, int kmer_size, int is_containment, double threshold,
                   rtc_edge* h_edges_out, uint64_t* h_n_edges) {
  return rtc_mst_dense(ctx, d_hashes, width, d_start, d_len, n, start_index, kmer_size, is_containment, threshold, h_edges_out,
                       h_n_edges, 0, nullptr, nullptr);
}

namespace {
// --dense accumulation (src/MST.cpp:1517-1530): every candidate pair adds one to the start bucket
// t0 = lower_bound(radius, dist) of both genomes and to its ANI bin.  Distances are the HOST doubles
// (same libm expression as the edge weights), so the buckets are the reference's bit for bit; with a
// common sketch size they come from a table indexed by `common`.
struct DenseAcc {
  rtc_ctx* ctx; const uint32_t* h_len; uint32_t n; int k, cont, span; uint32_t s_fixed;
  int32_t* dense; uint64_t* ani;
  std::vector<double> radius; std::vector<int16_t> t0_of, ani_of; std::vector<rtc_cedge> buf;
  void classify(uint32_t common, uint32_t a, uint32_t b, int& t0, int& an) const {
    const double d = host_mst_distance((int)common, (int)a, (int)b, k, cont);
    t0 = (int)(std::lower_bound(radius.begin(), radius.end(), d) - radius.begin());
    an = (int)((1.0 - d) * 100.0);
    if (an >= 101) an = 100;
    if (an < 0) an = 0;  // distances above 1 (no clamp in compute_minhash_mst); the reference would index out of range
  }
  static int on_new(void* self, const rtc_cedge* d_new, uint64_t count) {
    DenseAcc* A = (DenseAcc*)self;
    const uint64_t slab = 1u << 22;
    A->buf.resize((size_t)std::min<uint64_t>(slab, count));
    for (uint64_t p = 0; p < count; p += slab) {
      const uint64_t c = std::min<uint64_t>(slab, count - p);
      hipError_t e = hipMemcpyAsync(A->buf.data(), d_new + p, c * sizeof(rtc_cedge), hipMemcpyDeviceToHost, A->ctx->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(A->ctx->stream);
      if (e != hipSuccess) return rtc_fail(A->ctx, RTC_ERR_HIP, "dense: edge read-back -> %s", hipGetErrorString(e));
      for (uint64_t q = 0; q < c; q++) {
        const rtc_cedge& ed = A->buf[q];
        int t0, an;
        if (A->s_fixed) { t0 = A->t0_of[ed.common]; an = A->ani_of[ed.common]; }
        else A->classify(ed.common, A->h_len[ed.i], A->h_len[ed.j], t0, an);
        if (t0 < A->span) { A->dense[(size_t)t0 * A->n + ed.i]++; A->dense[(size_t)t0 * A->n + ed.j]++; }
        A->ani[an]++;
      }
    }
    return RTC_OK;
  }
};
}  // namespace

int rtc_mst_dense(rtc_ctx* ctx, const void* d_hashes, int width, const uint64_t* d_start, const uint32_t* d_len,
                  uint32_t n, uint32_t start_index, int kmer_size, int is_containment, double threshold,
                  rtc_edge* h_edges_out, uint64_t* h_n_edges, int dense_span, int32_t* h_dense, uint64_t* h_ani) {
  if (!ctx || !h_n_edges || (n && (!d_start || !d_len || !h_edges_out))) return RTC_ERR_ARG;
  if (dense_span < 0 || (dense_span > 0 && (!h_dense || !h_ani))) return RTC_ERR_ARG;
  *h_n_edges = 0;
  if (dense_span) { memset(h_dense, 0, (size_t)dense_span * n * sizeof(int32_t)); memset(h_ani, 0, 101 * sizeof(uint64_t)); }
  if (n < 2 || start_index >= n) return RTC_OK;
  RTC_HIP(ctx, hipSetDevice(ctx->device));
  rtc_mst_bufs_t B{};
  RTC_TRY(rtc_mst_bufs(ctx, n, &B));
  uint32_t* const h_len = B.h_len;
  RTC_HIP(ctx, hipMemcpyAsync(h_len, d_len, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
  RTC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const uint32_t s_fixed = rtc_fixed_size_of(h_len, n);

  const bool verbose = ctx->opt.verbose && !ctx->quiet;
  auto now = []() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; };
  const double tv0 = now();
  rtc_edge_list el{};
  rtc_cedge* const d_sel = B.d_sel;
  DenseAcc acc{ctx, h_len, n, kmer_size, is_containment, dense_span, s_fixed, h_dense, h_ani, {}, {}, {}, {}};
  rtc_edge_observer obs{DenseAcc::on_new, &acc};
  if (dense_span) {
    const double step = 1.0 / dense_span;                              // src/MST.cpp:1333-1340
    for (int i = 0; i < dense_span; i++) acc.radius.push_back(step * (double)i);
    if (s_fixed) {
      acc.t0_of.resize(s_fixed + 1); acc.ani_of.resize(s_fixed + 1);
      for (uint32_t c = 0; c <= s_fixed; c++) { int t0, an; acc.classify(c, s_fixed, s_fixed, t0, an); acc.t0_of[c] = (int16_t)t0; acc.ani_of[c] = (int16_t)an; }
    }
  }
  int st = rtc_candidate_edges_device(ctx, d_hashes, width, d_start, d_len, n, std::max<uint32_t>(start_index, 1), n, kmer_size,
                                      is_containment, threshold, s_fixed, &el, dense_span ? &obs : nullptr);
  if (st == RTC_OK && dense_span) {  // start-bucket counts -> cumulative density counts (:1703-1713)
    for (uint32_t g = 0; g < n; g++) {
      int32_t a = 0;
      for (int t = 0; t < dense_span; t++) { a += h_dense[(size_t)t * n + g]; h_dense[(size_t)t * n + g] = a; }
    }
  }
  uint64_t nsel = 0;
  const double tv1 = now();
  int rounds = 0;
  if (st == RTC_OK) st = rtc_msf_device(ctx, el.d_edges, el.m, d_len, n, is_containment, s_fixed, nullptr, d_sel, &nsel, &rounds, true,
                                        s_fixed ? 0u : *std::max_element(h_len, h_len + n));
  const double tv2 = now();
  if (st == RTC_OK && nsel) {
    hipError_t e = hipMemcpyAsync(B.h_sel, d_sel, nsel * sizeof(rtc_cedge), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) st = rtc_fail(ctx, RTC_ERR_HIP, "forest read-back -> %s", hipGetErrorString(e));
  }
  const uint64_t m_edges = el.m;
  rtc_edge_list_free(&el, ctx);
  if (st != RTC_OK) return st;
  RTC_TRY(rtc_edges_to_mst_host_fixed(B.h_sel, nsel, h_len, kmer_size, is_containment, s_fixed, h_edges_out));
  *h_n_edges = nsel;
  if (verbose)
    fprintf(stderr, "[mst]   %u sketches: %llu candidate edges in %.4fs, forest (%d rounds, %llu edges) in %.4fs, finish %.4fs\n", n,
            (unsigned long long)m_edges, tv1 - tv0, rounds, (unsigned long long)nsel, tv2 - tv1, now() - tv2);
  return RTC_OK;
}

// modifyMST (src/MST.cpp:809-1018): the dense loop.  EVERY pair i < j with j >= start_index is an edge -- no
// "shares a hash" / size-ratio filters -- weighted by MinHash::distance() (Mash's union-truncated estimator) or, for
// containment sketches, containDistance(); the result is a spanning TREE (pairs without a common hash weigh 1).
// Rows are evaluated in chunks into a dense count matrix (rtc_pair_mash_dev / rtc_pair_common_dev), every pair of the
// chunk is appended to the edge list, and the list is contracted to its own forest whenever the next chunk would
// not fit (the reference does the same per 8-row block: sort + kruskalAlgorithm, :905-908).
int rtc_mst_mash(rtc_ctx* ctx, const void* d_hashes, int width, const uint64_t* d_start, const uint32_t* d_len, uint32_t n,
                 uint32_t start_index, int kmer_size, int is_containment, uint32_t sketch_size, rtc_edge* h_edges_out,
                 uint64_t* h_n_edges, int dense_span, int32_t* h_dense, uint64_t* h_ani) {
  if (!ctx || !h_n_edges || (n && (!d_start || !d_len || !h_edges_out))) return RTC_ERR_ARG;
  if (dense_span < 0 || (dense_span > 0 && (!h_dense || !h_ani))) return RTC_ERR_ARG;
  if (!is_containment && (sketch_size == 0 || sketch_size >= (1u << 29))) return rtc_fail(ctx, RTC_ERR_ARG, "sketch_size %u", sketch_size);
  *h_n_edges = 0;
  if (dense_span) { memset(h_dense, 0, (size_t)dense_span * n * sizeof(int32_t)); memset(h_ani, 0, 101 * sizeof(uint64_t)); }
  if (n < 2 || start_index >= n) return RTC_OK;
  RTC_HIP(ctx, hipSetDevice(ctx->device));
  const int wmode = is_containment ? 1 : (2 | (int)(sketch_size << 2));
  std::vector<uint32_t> h_len(n);
  RTC_HIP(ctx, hipMemcpyAsync(h_len.data(), d_len, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
  RTC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const uint32_t s_fixed = rtc_fixed_size_of(h_len.data(), n);  // equal sizes: the weight is monotone in `common` in every mode

  uint64_t budget = (uint64_t)256 << 20;
  if (ctx->opt.edge_budget) budget = ctx->opt.edge_budget;
  const uint32_t rbeg = std::max<uint32_t>(start_index, 1);
  uint32_t rows_per = (uint32_t)std::min<uint64_t>(262140, std::max<uint64_t>(4, (((uint64_t)1 << 26) / n)));
  rows_per = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(rows_per, std::max<uint64_t>(budget / 2 / n, 1)));
  auto pairs_below = [](uint64_t r) { return r * (r - (r ? 1 : 0)) / 2; };
  const uint64_t total = pairs_below(n) - pairs_below(rbeg);
  const uint64_t cap = std::min<uint64_t>(total, std::max<uint64_t>(budget, (uint64_t)n + (uint64_t)rows_per * n));
  uint32_t *d_common = nullptr, *d_denom = nullptr;
  void* ws2 = nullptr;
  RTC_TRY(rtc_ws(ctx, 2, (size_t)rows_per * n * 4 * (is_containment ? 1 : 2) + 64, &ws2));
  d_common = (uint32_t*)ws2;
  d_denom = d_common + (size_t)rows_per * n;
  rtc_cedge *d_edges = nullptr, *d_sel = nullptr;
  if (hipMalloc((void**)&d_edges, cap * sizeof(rtc_cedge)) != hipSuccess) return rtc_fail(ctx, RTC_ERR_NOMEM, "edge list of %llu pairs", (unsigned long long)cap);
  if (hipMalloc((void**)&d_sel, (size_t)n * sizeof(rtc_cedge)) != hipSuccess) { (void)hipFree(d_edges); return rtc_fail(ctx, RTC_ERR_NOMEM, "hipMalloc forest list"); }

  DenseAcc acc{ctx, h_len.data(), n, kmer_size, wmode, dense_span, 0, h_dense, h_ani, {}, {}, {}, {}};
  if (dense_span) {
    const double step = 1.0 / dense_span;                              // :819-823
    for (int i = 0; i < dense_span; i++) acc.radius.push_back(step * (double)i);
  }
  int st = RTC_OK;
  uint64_t m = 0, nsel = 0;
  for (uint32_t r0 = rbeg; r0 < n && st == RTC_OK; r0 += rows_per) {
    const uint32_t r1 = std::min<uint32_t>(n, r0 + rows_per);
    const uint64_t chunk = pairs_below(r1) - pairs_below(r0);
    if (is_containment) st = rtc_pair_common_dev(ctx, d_hashes, width, d_start, d_len, n, r0, r1, 0, r1 - 1, d_common, n, 1, 0);
    else st = rtc_pair_mash_dev(ctx, d_hashes, width, d_start, d_len, n, sketch_size, r0, r1, 0, r1 - 1, d_common, d_denom, n);
    if (st != RTC_OK) break;
    if (m + chunk > cap) {  // contract what is there to its forest (at most n - 1 edges)
      st = rtc_msf_device(ctx, d_edges, m, d_len, n, wmode, s_fixed, nullptr, d_sel, &nsel, nullptr, false);
      if (st != RTC_OK) break;
      if (hipMemcpyAsync(d_edges, d_sel, nsel * sizeof(rtc_cedge), hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) { st = rtc_fail(ctx, RTC_ERR_HIP, "forest copy"); break; }
      m = nsel;
      ctx->diag[3]++;
    }
    hipLaunchKernelGGL(all_pairs_edges_kernel, dim3(std::max<uint32_t>(1, std::min<uint32_t>((r1 + 255) / 256, 64)), std::min<uint32_t>(r1 - r0, 4096)),
                       dim3(256), 0, ctx->stream, d_common, (uint64_t)n, r0, r1, d_edges + m);
    if (hipGetLastError() != hipSuccess) { st = rtc_fail(ctx, RTC_ERR_HIP, "all_pairs_edges_kernel launch"); break; }
    if (dense_span) { st = DenseAcc::on_new(&acc, d_edges + m, chunk); if (st != RTC_OK) break; }
    m += chunk;
  }
  if (st == RTC_OK && dense_span) {  // start-bucket counts -> cumulative density counts (:985-994)
    for (uint32_t g = 0; g < n; g++) {
      int32_t a = 0;
      for (int t = 0; t < dense_span; t++) { a += h_dense[(size_t)t * n + g]; h_dense[(size_t)t * n + g] = a; }
    }
  }
  std::vector<rtc_cedge> sel;
  if (st == RTC_OK) st = rtc_msf_device(ctx, d_edges, m, d_len, n, wmode, s_fixed, nullptr, d_sel, &nsel, nullptr);
  if (st == RTC_OK && nsel) {
    sel.resize(nsel);
    hipError_t e = hipMemcpyAsync(sel.data(), d_sel, nsel * sizeof(rtc_cedge), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) st = rtc_fail(ctx, RTC_ERR_HIP, "forest read-back -> %s", hipGetErrorString(e));
  }
  (void)hipStreamSynchronize(ctx->stream);
  (void)hipFree(d_edges); (void)hipFree(d_sel);
  if (st != RTC_OK) return st;
  RTC_TRY(rtc_edges_to_mst_host(sel.data(), nsel, h_len.data(), kmer_size, wmode, h_edges_out));
  for (uint64_t e = 0; e < nsel; e++) std::swap(h_edges_out[e].preNode, h_edges_out[e].sufNode);  // EdgeInfo{i, j}, i < j (:891)
  std::sort(h_edges_out, h_edges_out + nsel, [](const rtc_edge& a, const rtc_edge& b) {
    if (a.dist != b.dist) return a.dist < b.dist;
    if (a.preNode != b.preNode) return a.preNode < b.preNode;
    return a.sufNode < b.sufNode;
  });
  *h_n_edges = nsel;
  return RTC_OK;
}

// The forest of a device-resident candidate list on ONE GPU: all Boruvka rounds behind one call (the per-round
// primitives above remain for hosts that bring their own collectives).  d_sel: n entries; *h_n_sel edges are written.
int rtc_msf_dev(rtc_ctx* ctx, const rtc_cedge* d_edges, uint64_t m, const uint32_t* d_len, uint32_t n, int is_containment,
                rtc_cedge* d_sel, uint64_t* h_n_sel, int* h_rounds) {
  if (!ctx || !d_len || !d_sel || !h_n_sel || (m && !d_edges)) return RTC_ERR_ARG;
  *h_n_sel = 0;
  if (h_rounds) *h_rounds = 0;
  if (n < 2) return RTC_OK;
  RTC_HIP(ctx, hipSetDevice(ctx->device));
  // equal sketch sizes (and a key that fits): one fused key per component and round
  void* ws = nullptr;
  RTC_TRY(rtc_ws(ctx, 5, 256, &ws));
  uint32_t* d_mm = (uint32_t*)ws;
  RTC_HIP(ctx, hipMemsetAsync(d_mm, 0xff, 4, ctx->stream));
  RTC_HIP(ctx, hipMemsetAsync(d_mm + 1, 0, 4, ctx->stream));
  hipLaunchKernelGGL(minmax_u32_kernel, dim3(std::min<uint32_t>((n + 255) / 256, 256)), dim3(256), 0, ctx->stream, d_len, n, d_mm);
  RTC_CHECK_LAUNCH(ctx);
  uint32_t mm[2] = {0, 0};
  RTC_HIP(ctx, hipMemcpyAsync(mm, d_mm, 8, hipMemcpyDeviceToHost, ctx->stream));
  RTC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const uint32_t s_fixed = (mm[0] == mm[1] && mm[0] > 0 && rtc_boruvka_key_bits(n, mm[0])) ? mm[0] : 0;
  return rtc_msf_device(ctx, d_edges, m, d_len, n, is_containment, s_fixed, nullptr, d_sel, h_n_sel, h_rounds);
}

}  // extern "C"

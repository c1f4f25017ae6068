// rtc_pairs_join.hip -- candidate edges of a lower-triangle tile by an inverted join on the device.
//
// The reference finds |A_i ∩ A_j| through an inverted index hash -> genomes (src/MST.cpp:1408-1435,
// :428-487): only pairs that share a hash are ever touched.  This is the same computation laid out for a
// GPU: the index is a sort.
//   1. flatten   every hash of the genomes [g0, g1) with its genome id            (one streaming pass)
//   2. sort      (hash, genome) by hash -- stable, so genomes ascend inside a posting list   (radix sort)
//   3. count     element a of a posting list pairs with the later elements of the list whose genome is a row of the
//                tile: its partners, a contiguous part of the sorted list; a descriptor (first partner, partners) per
//                element that has any, kept in its column genome's part of a flat array
//   4. columns   a wave per column genome counts the column's partner lists in an LDS table row -> count:
//                the counts are |A_row ∩ A_col|; the reference's candidate filters (src/MST.cpp:1468-1487) ->
//                (i, j, common) triples
// (until round 5 steps 4-7 wrote one code (row << bits | col) per co-occurrence, sorted the codes and run-length-encoded them)
// Results are the integers the tiled kernel (rtc_pairs_tiled.hip) produces for the same tile, pair for pair;
// the cost is O(hashes + co-occurrences) instead of O(rows * cols * s / 64).  The tiled kernel stays the
// general path: it is taken when the co-occurrence count (estimated from a sample before the sort, known exactly after
// step 3) would make the join the slower of the two (many near-identical genomes: posting lists of thousands), when the
// scratch would not fit, and by callers that hold a tiled plan across launches.  The sort, the offset scan and the
// reduction are rocPRIM device primitives; the kernels around them are below.
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cmath>

#include "rtc_internal.h"

namespace {

struct U32ToU64 {
  __host__ __device__ uint64_t operator()(uint32_t v) const { return (uint64_t)v; }
};

// largest hash of the genomes [g0, g0 + ng): sketches ascend, so it is the largest last element
template <typename T>
__global__ __launch_bounds__(256) void join_maxkey_kernel(const T* __restrict__ hashes, const uint64_t* __restrict__ start,
                                                          const uint32_t* __restrict__ len, uint32_t g0, uint32_t ng,
                                                          unsigned long long* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long m = 0;
  if (i < ng) { const uint32_t L = len[g0 + i]; if (L) m = (unsigned long long)hashes[start[g0 + i] + L - 1]; }
  for (int o = 32; o > 0; o >>= 1) { const unsigned long long t = __shfl_xor(m, o); m = t > m ? t : m; }
  // one atomic per workgroup (atomics on one address queue up)
  __shared__ unsigned long long s_m[4];
  if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; w++) m = s_m[w] > m ? s_m[w] : m;
    if (m) atomicMax(out, m);
  }
}

// A look at the density of the input before anything is sorted: 1/64 of the hash space (by a multiplicative hash of the value)
// is counted into a direct-address table, Sum c (c - 1) / 2 over its cells x 64 estimates the co-occurrences of the whole set.
// Sets with posting lists of thousands (families of near-identical genomes) show up here for 40 us instead of after a flat
// copy, a radix sort of every hash and a count (0.6 ms + 300 MB of scratch) -- the first launch on such a set used to pay that.
constexpr int SAMPLE_SHIFT = 6, SAMPLE_CELL_BITS = 20;
template <typename T>
__global__ __launch_bounds__(256) void join_sample_kernel(const T* __restrict__ hashes, const uint64_t* __restrict__ start,
                                                          const uint32_t* __restrict__ len, uint32_t g0, uint32_t* __restrict__ table) {
  const uint32_t g = g0 + blockIdx.x;
  const uint32_t L = len[g];
  const T* src = hashes + start[g];
  for (uint32_t e = threadIdx.x; e < L; e += blockDim.x) {
    const uint64_t x = (uint64_t)src[e] * 0x9E3779B97F4A7C15ull;
    if ((x >> (64 - SAMPLE_SHIFT)) == 0) atomicAdd(table + (uint32_t)((x >> (64 - SAMPLE_SHIFT - SAMPLE_CELL_BITS)) & ((1u << SAMPLE_CELL_BITS) - 1u)), 1u);
  }
}
// out[0] = Sum c (c - 1) / 2, out[1] = Sum c^2, out[2] = Sum c over the cells (Sum c^2 / Sum c: the list length a sampled hash sees)
__global__ __launch_bounds__(256) void join_sample_sum_kernel(const uint32_t* __restrict__ table, unsigned long long* __restrict__ out) {
  unsigned long long acc = 0, sq = 0, su = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < (1u << SAMPLE_CELL_BITS); i += gridDim.x * blockDim.x) {
    const unsigned long long c = table[i];
    acc += c * (c - 1) / 2;
    sq += c * c;
    su += c;
  }
  for (int o = 32; o > 0; o >>= 1) { acc += __shfl_xor(acc, o); sq += __shfl_xor(sq, o); su += __shfl_xor(su, o); }
  // one set of atomics per workgroup: the three sums share a cache line, and atomics on one line queue up (~13 ns each; a set per
  // wave of 256 workgroups made this kernel 43 us for a 4 MB read)
  __shared__ unsigned long long s_r[4][3];
  if ((threadIdx.x & 63) == 0) { s_r[threadIdx.x >> 6][0] = acc; s_r[threadIdx.x >> 6][1] = sq; s_r[threadIdx.x >> 6][2] = su; }
  __syncthreads();
  if (threadIdx.x < 3) {
    const unsigned long long v = s_r[0][threadIdx.x] + s_r[1][threadIdx.x] + s_r[2][threadIdx.x] + s_r[3][threadIdx.x];
    if (v) atomicAdd(out + threadIdx.x, v);
  }
}

// keys[off[g - g0] + e] = element e of sketch g, vals[...] = g; one workgroup per genome
template <typename T>
__global__ __launch_bounds__(256) void join_flatten_kernel(const T* __restrict__ hashes, const uint64_t* __restrict__ start,
                                                           const uint32_t* __restrict__ len, const uint64_t* __restrict__ off,
                                                           uint32_t g0, T* __restrict__ keys, uint32_t* __restrict__ vals) {
  const uint32_t g = g0 + blockIdx.x;
  const uint32_t L = len[g];
  const T* src = hashes + start[g];
  const uint64_t o = off[blockIdx.x];
  for (uint32_t e = threadIdx.x; e < L; e += blockDim.x) { keys[o + e] = src[e]; vals[o + e] = g; }
}

// ---- semi-join in front of the sort (row shards of a multi-GPU run: few rows, many columns) -------------------
// Only hashes that occur in a ROW genome can make a (row, col) pair.  A blocked Bloom filter of the rows' hashes (two
// bits in one 64-bit word per hash, ~8 bits of filter per hash: ~6 % false positives, harmless) lets the flat copy
// keep just the column hashes that may: the rank holding the last eighth of the rows of 100 000 sketches sorts
// 1.3e7 instead of 1e8 records.  Order inside a genome is kept (the stable sort needs genomes ascending in a list).
template <typename T>
__device__ __forceinline__ void bloom_slot(T key, int wshift, uint32_t& word, unsigned long long& mask) {
  const uint64_t h = (uint64_t)key * 0x9E3779B97F4A7C15ULL;
  word = (uint32_t)(h >> wshift);
  mask = (1ULL << (h & 63)) | (1ULL << ((h >> 6) & 63));
}
template <typename T>
__global__ __launch_bounds__(256) void join_bloom_build_kernel(const T* __restrict__ hashes, const uint64_t* __restrict__ start,
                                                               const uint32_t* __restrict__ len, uint32_t row0, int wshift,
                                                               unsigned long long* __restrict__ bloom) {
  const uint32_t g = row0 + blockIdx.x;
  const uint32_t L = len[g];
  const T* src = hashes + start[g];
  for (uint32_t e = threadIdx.x; e < L; e += blockDim.x) {
    uint32_t w; unsigned long long m;
    bloom_slot<T>(src[e], wshift, w, m);
    atomicOr(&bloom[w], m);
  }
}
// FILL = false: kept[g - g0] = hashes of genome g that pass, and a byte per hash (at the genome's place among ALL hashes,
// off_all) that says whether it did; FILL = true: write the ones marked (and g) at off[g - g0], in order -- from the bytes, not
// from the filter again: the probe is a random 8-byte read per hash (0.8 ms per pass over the 10^8 column hashes of a row shard of
// BASELINE configs[4], the largest part of a rank's pair phase at eight GPUs until round 6).
template <typename T, bool FILL>
__global__ __launch_bounds__(256) void join_semi_kernel(const T* __restrict__ hashes, const uint64_t* __restrict__ start,
                                                        const uint32_t* __restrict__ len, uint32_t g0, uint32_t row0, int wshift,
                                                        const unsigned long long* __restrict__ bloom, uint32_t* __restrict__ kept,
                                                        const uint64_t* __restrict__ off, T* __restrict__ keys,
                                                        uint32_t* __restrict__ vals, const uint64_t* __restrict__ off_all,
                                                        uint8_t* __restrict__ passed) {
  __shared__ uint32_t wtot[4];
  const uint32_t g = g0 + blockIdx.x;
  const uint32_t L = len[g];
  const T* src = hashes + start[g];
  const bool is_row = g >= row0;  // rows are later rows' columns too, and their own hashes all pass by construction
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint64_t base = FILL ? off[blockIdx.x] : 0;
  uint8_t* mine = passed + off_all[blockIdx.x];
  uint32_t total = 0;
  for (uint32_t e0 = 0; e0 < L; e0 += 256) {
    const uint32_t e = e0 + threadIdx.x;
    bool keep = false;
    T key = 0;
    if (e < L) {
      key = src[e];
      keep = is_row;
      if (!is_row) {
        if (FILL) keep = mine[e] != 0;
        else {
          uint32_t w; unsigned long long m;
          bloom_slot<T>(key, wshift, w, m);
          keep = (bloom[w] & m) == m;
        }
      }
      if (!FILL) mine[e] = keep ? 1 : 0;
    }
    const uint64_t bal = __ballot(keep);
    if (lane == 0) wtot[wave] = (uint32_t)__popcll(bal);
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) { const uint32_t t = wtot[w]; if ((uint32_t)w < wave) before += t; all += t; }
    if (FILL && keep) {
      const uint64_t o = base + total + before + (uint32_t)__popcll(bal & ((1ULL << lane) - 1ULL));
      keys[o] = key; vals[o] = g;
    }
    total += all;
    __syncthreads();
  }
  if (!FILL && threadIdx.x == 0) kept[blockIdx.x] = total;
}

// 64-bit hashes are sorted on their 32 most significant bits that vary only (half the radix passes; bottom-s MinHash
// values are small, the bits above the largest hash are skipped as well).  Distinct hashes that agree in those bits
// end up in one run in input order; such a run is out of order somewhere, which is what this kernel
// looks for (a few per million hashes).  An inversion is noted in one of the FIX_SLOTS places of its workgroup's own row of
// `slots` -- no shared counter: until round 6 every wave that saw one took its place in ONE list with an atomic, and atomics on one
// address queue up (~7.5 ns each: 2.2 ms for the 291 000 inversions of K = 10^8 on 32 bits, which is why such sets were sorted on
// 40 bits, a fifth radix pass of 0.78 ms).  Only a workgroup with more inversions than its row holds (one in a thousand at that
// density) spills into the list: fix[0] = entries of the list, fix[1] = "could not repair", fix[2..] = positions.
constexpr uint32_t FIX_CAP = 1u << 20;
constexpr uint32_t FIX_RUN_MAX = 2048;
constexpr uint32_t FIX_SLOTS = 4, FIX_NONE = 0xffffffffu;
// past this many expected inversions another radix pass is the cheaper way (the repair itself is a lane per inversion)
constexpr uint32_t FIX_EXPECT_MAX = 1u << 20;
__global__ __launch_bounds__(256) void join_inversions_kernel(const uint64_t* __restrict__ ks, uint32_t K, int sh, uint32_t* __restrict__ fix,
                                                              uint32_t* __restrict__ slots) {
  __shared__ uint32_t s_n;
  __shared__ uint32_t s_at[FIX_SLOTS];
  if (threadIdx.x == 0) s_n = 0;
  if (threadIdx.x < FIX_SLOTS) s_at[threadIdx.x] = FIX_NONE;
  __syncthreads();
  const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
  bool inv = false;
  if (a + 1 < K) {
    const uint64_t x = ks[a], y = ks[a + 1];
    if ((x >> sh) > (y >> sh)) fix[1] = 1;  // not sorted on the bits asked for: never seen, never trusted
    inv = x > y;
  }
  if (inv) {
    const uint32_t i = atomicAdd(&s_n, 1u);  // (LDS)
    if (i < FIX_SLOTS) s_at[i] = a;
    else {
      const uint32_t j = atomicAdd(&fix[0], 1u);
      if (j < FIX_CAP) fix[2 + j] = a;
    }
  }
  __syncthreads();
  if (threadIdx.x < FIX_SLOTS) slots[(size_t)blockIdx.x * FIX_SLOTS + threadIdx.x] = s_at[threadIdx.x];
}

// entry t of the inversions: the workgroups' rows first, the spill list behind them
__device__ __forceinline__ uint32_t* fix_entry(uint32_t t, uint32_t nslots, uint32_t* slots, uint32_t* fix) {
  if (t < nslots) return slots + t;
  const uint32_t j = t - nslots;
  return j < min(fix[0], FIX_CAP) ? fix + 2 + j : nullptr;
}

// One lane per inversion.  First (read-only) pass: is this the first inversion of its run?  The others are struck
// out (bit 31).  Second pass: the remaining lanes each sort their run by the full hash with a stable insertion
// sort (two or three posting lists interleaved, rarely more than a few dozen elements); the runs are disjoint.
__global__ __launch_bounds__(64) void join_repair_owner_kernel(const uint64_t* __restrict__ ks, int sh, uint32_t* __restrict__ fix,
                                                               uint32_t* __restrict__ slots, uint32_t nslots) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (fix[0] > FIX_CAP) { if (i == 0) fix[1] = 1; return; }
  uint32_t* ent = fix_entry(i, nslots, slots, fix);
  if (!ent || *ent == FIX_NONE) return;
  const uint32_t a = *ent;
  const uint64_t pre = ks[a] >> sh;
  uint32_t s = a;
  while (s > 0 && (ks[s - 1] >> sh) == pre) {
    s--;
    if (a - s > FIX_RUN_MAX) { fix[1] = 1; return; }
    if (ks[s] > ks[s + 1]) { *ent = a | 0x80000000u; return; }  // an earlier inversion of the same run
  }
}
__global__ __launch_bounds__(64) void join_repair_kernel(uint64_t* __restrict__ ks, uint32_t* __restrict__ vs, uint32_t K, int sh,
                                                         uint32_t* __restrict__ fix, uint32_t* __restrict__ slots, uint32_t nslots) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (fix[0] > FIX_CAP || fix[1]) return;
  uint32_t* ent = fix_entry(i, nslots, slots, fix);
  if (!ent) return;
  const uint32_t a = *ent;
  if (a & 0x80000000u) return;  // (struck out, or FIX_NONE)
  const uint64_t pre = ks[a] >> sh;
  uint32_t s = a;
  while (s > 0 && (ks[s - 1] >> sh) == pre) s--;
  uint32_t e = a + 1;
  while (e < K && (ks[e] >> sh) == pre) {
    e++;
    if (e - s > FIX_RUN_MAX) { fix[1] = 1; return; }
  }
  for (uint32_t x = s + 1; x < e; x++) {
    const uint64_t kx = ks[x];
    const uint32_t vx = vs[x];
    uint32_t y = x;
    while (y > s && ks[y - 1] > kx) { ks[y] = ks[y - 1]; vs[y] = vs[y - 1]; y--; }
    ks[y] = kx; vs[y] = vx;
  }
}

// ---- the column-centric second half (round 5) ---------------------------------------------------------------------------
// Element a of the sorted list: [a + 1, ge) are the later members of its posting list (genomes ascending); those that are
// rows of the tile are a contiguous part [lo, lo + cnt) of it -- the element's PARTNERS.  An element whose own genome g is a
// column of the tile and that has partners leaves the descriptor (lo, cnt) in its column's part of `desc` (column g owns
// places off[g - g0] .. off[g - g0 + 1]: as many as the genome has hashes; the order inside is whatever the atomics made
// it) and adds itself to the column's counter = elements << 40 | partners (a 64-byte line per column: atomics on one line
// queue up behind each other).  A wave then counts one column's partners (a few hundred to a few thousand row ids, a dozen
// distinct) in an LDS table of its own, reading the partner lists where the sort left them, and appends the column's edges:
// no second sort, no run-length encode, no copy of the co-occurrences, no K-sized scan.
constexpr int CC_ESHIFT = 40;
constexpr unsigned long long CC_PMASK = (1ULL << CC_ESHIFT) - 1ULL;
constexpr int CC_CSTRIDE = 1;  // counters, in 8-byte words, from one column to the next (a 64-byte line each was measured: the count kernel 145 -> 234 us)
template <typename T>
__global__ __launch_bounds__(256) void join_count_kernel(const T* __restrict__ ks, const uint32_t* __restrict__ vs, uint32_t K,
                                                         uint32_t row0, uint32_t col0, uint32_t col1, uint32_t g0,
                                                         const uint64_t* __restrict__ off, unsigned long long* __restrict__ colcnt,
                                                         uint2* __restrict__ desc) {
  const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;  // (whole waves stay: the vote below)
  const uint32_t lane = threadIdx.x & 63;
  const bool in = a < K;
  const T key = in ? ks[a] : (T)0;
  const bool same_next = in && a + 1 < K && ks[a + 1] == key;
  // Where the posting lists end inside this wave: no loads for a list that does (nearly all of them: lists are a few to a few
  // dozen elements, a wave holds 64).  One that runs past the wave's last element is followed from there.
  const uint64_t ends = __ballot(!same_next);
  if (!same_next) return;
  const uint32_t g = vs[a];
  if (g < col0 || g >= col1) return;
  uint32_t ge;  // one past the list's last element
  const uint64_t m = ends >> lane;
  if (m) ge = a + (uint32_t)__builtin_ctzll(m) + 1;
  else {
    // gallop from the first element behind the wave (it holds the key: the wave's last lane said "same"): largest ge with
    // ks[ge - 1] == key
    uint32_t step = 1, known = a - lane + 64;
    while (known + step < K && ks[known + step] == key) { known += step; step <<= 1; }
    uint32_t hi_ex = min(K, known + step);  // ks[hi_ex] != key (or K)
    uint32_t l = known + 1, h = hi_ex;
    while (l < h) { const uint32_t mid = (l + h) >> 1; if (ks[mid] == key) l = mid + 1; else h = mid; }
    ge = l;
  }
  // genomes ascend in (a, ge) and are all above g and below row1 (the flat copy ends there): only a column below row0 - 1 has
  // to look for its first partner that is a row
  uint32_t lo = a + 1;
  if (row0 > g + 1) {
    uint32_t q = ge;
    while (lo < q) { const uint32_t mid = (lo + q) >> 1; if (vs[mid] < row0) lo = mid + 1; else q = mid; }
  }
  const uint32_t cnt = ge - lo;
  if (cnt) {
    const unsigned long long was = atomicAdd(colcnt + (size_t)(g - g0) * CC_CSTRIDE, (1ULL << CC_ESHIFT) | (unsigned long long)cnt);
    desc[off[g - g0] + (was >> CC_ESHIFT)] = make_uint2(lo, cnt);
  }
}
struct ColPartners {  // the partners of column i, for the sum over the columns
  const unsigned long long* colcnt;
  __device__ unsigned long long operator()(uint32_t i) const { return colcnt[(size_t)i * CC_CSTRIDE] & CC_PMASK; }
};

constexpr uint32_t CC_EMPTY = 0xffffffffu;
__device__ __forceinline__ uint32_t cc_slot(uint32_t r, uint32_t mask) { return ((r * 0x9E3779B1u) >> 12) & mask; }
__device__ __forceinline__ bool cc_keep(uint32_t s0, uint32_t s1, int radio) {  // the reference's size filter, src/MST.cpp:1484
  const uint32_t mn = s0 < s1 ? s0 : s1, mx = s0 > s1 ? s0 : s1;
  return (uint64_t)mx <= (uint64_t)(uint32_t)radio * (uint64_t)mn;
}
// One wave per column (cpw of them in turn when there are more columns than the chip has wave slots).  256 descriptors per
// step, CC_DPL = four per lane: the lists of up to CC_SHORT partners are laid end to end as addresses in the wave's `src` (a
// prefix sum over the lanes says where; CC_SRC addresses at a time), then read back 1 024 at a time -- sixteen independent
// gathers per lane in flight (pair phase of the headline with 4 / 8 / 16 of them, from 64 / 128 / 256 descriptors a step:
// 0.98-1.00 / 0.97 / 0.95-0.96 ms; the kernel waits on memory, not on the table) -- into the wave's table (row -> count);
// a longer list is walked by the whole wave.  A row that is in the table already costs one
// plain read and one add; a first sight takes the compare-and-swap and notes its slot, so that reading the table out (and
// clearing it) walks the distinct partners, not the slots: through the size filter into a staging list.  Staged edges are
// appended with ONE global atomic per workgroup (atomics on the one list counter are what the chip serialises: ~13 ns each).
// A column with more than CC_LIGHT_MAX distinct partners goes on the heavy list (join_colcount_heavy_kernel).
constexpr int CC_WAVES = 4, CC_SLOTS = 1024, CC_STAGE = 64, CC_DPL = 4, CC_DEPTH = 16, CC_SHORT = 64, CC_SRC = 1024, CC_LIGHT_MAX = 640;
static_assert(CC_DEPTH % 4 == 0 && CC_LIGHT_MAX + 256 + 64 <= CC_SLOTS, "the table's margin");
__global__ __launch_bounds__(64 * CC_WAVES) void join_colcount_kernel(const uint32_t* __restrict__ vs, const uint2* __restrict__ desc,
                                                                     const uint64_t* __restrict__ off, const unsigned long long* __restrict__ colcnt,
                                                                     uint32_t g0, uint32_t c_lo, uint32_t c_hi, uint32_t cpw,
                                                                     const uint32_t* __restrict__ len, int radio, rtc_cedge* __restrict__ edges,
                                                                     unsigned long long cap, unsigned long long* __restrict__ count,
                                                                     uint32_t* __restrict__ heavy, uint32_t* __restrict__ heavy_n) {
  __shared__ uint32_t s_key[CC_WAVES][CC_SLOTS], s_cnt[CC_WAVES][CC_SLOTS], s_src[CC_WAVES][CC_SRC];
  __shared__ uint16_t s_seen[CC_WAVES][CC_SLOTS];
  __shared__ rtc_cedge s_stage[CC_WAVES][CC_STAGE];
  __shared__ uint32_t s_nst[CC_WAVES];
  __shared__ unsigned long long s_base;
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint32_t* key = s_key[wave];
  uint32_t* cnt = s_cnt[wave];
  uint32_t* src = s_src[wave];
  uint16_t* seen = s_seen[wave];
  rtc_cedge* stage = s_stage[wave];
  const uint32_t first = min(c_hi, c_lo + (blockIdx.x * CC_WAVES + wave) * cpw);  // (c_hi: a wave without columns, there for the barriers)
  for (uint32_t i = lane; i < (uint32_t)CC_SLOTS; i += 64) { key[i] = CC_EMPTY; cnt[i] = 0; }
  uint32_t nst = 0;  // staged edges (wave-uniform)
  auto flush = [&]() {
    if (!nst) return;
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(count, (unsigned long long)nst);
    base = ((unsigned long long)(uint32_t)__shfl((int)(base >> 32), 0) << 32) | (uint32_t)__shfl((int)(uint32_t)base, 0);
    for (uint32_t i = lane; i < nst; i += 64) if (base + i < cap) edges[base + i] = stage[i];
    nst = 0;
  };
  for (uint32_t c = first; c < first + cpw && c < c_hi; c++) {
    const unsigned long long cc = colcnt[(size_t)(c - g0) * CC_CSTRIDE];
    const uint32_t ne = (uint32_t)(cc >> CC_ESHIFT);
    if (!ne) continue;  // (wave-uniform)
    const unsigned long long partners = cc & CC_PMASK;
    const uint2* d = desc + off[c - g0];
    const uint32_t s1 = len[c];
    uint32_t slots = 64;
    while (slots < (uint32_t)CC_SLOTS && (unsigned long long)slots < 2 * partners) slots <<= 1;
    const uint32_t mask = slots - 1;
    uint32_t distinct = 0;
    auto insert = [&](uint32_t r) {  // (a row id is never the marker: ids stay below 2^31)
      bool fresh = false;
      uint32_t at = 0;
      if (r != CC_EMPTY) {
        at = cc_slot(r, mask);
        for (;;) {  // ends: at most CC_LIGHT_MAX + 256 of the 1 024 slots are ever taken
          uint32_t o = key[at];
          if (o == CC_EMPTY) { o = atomicCAS(&key[at], CC_EMPTY, r); fresh = o == CC_EMPTY; }
          if (o == r || fresh) break;
          at = (at + 1) & mask;
        }
        atomicAdd(&cnt[at], 1u);
      }
      const uint64_t fm = __ballot(fresh);
      if (fresh) seen[distinct + (uint32_t)__popcll(fm & ((1ULL << lane) - 1ULL))] = (uint16_t)at;
      distinct += (uint32_t)__popcll(fm);
    };
    uint2 dn[CC_DPL];
#pragma unroll
    for (int j = 0; j < CC_DPL; j++) dn[j] = 64 * j + lane < ne ? d[64 * j + lane] : make_uint2(0, 0);
    for (uint32_t e0 = 0; e0 < ne && distinct <= (uint32_t)CC_LIGHT_MAX; e0 += 64 * CC_DPL) {
      uint2 de[CC_DPL];
      uint32_t ns[CC_DPL], pos[CC_DPL];
      uint32_t mine = 0;
#pragma unroll
      for (int j = 0; j < CC_DPL; j++) {
        de[j] = dn[j];
        dn[j] = e0 + 64 * (CC_DPL + j) + lane < ne ? d[e0 + 64 * (CC_DPL + j) + lane] : make_uint2(0, 0);  // the next step's descriptors are on their way
        ns[j] = de[j].y <= (uint32_t)CC_SHORT ? de[j].y : 0;  // the short lists end to end
        mine += ns[j];
      }
      uint32_t incl = mine;  // inclusive prefix sum over the lanes
#pragma unroll
      for (int sft = 1; sft < 64; sft <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, sft); if ((int)lane >= sft) incl += o; }
      const uint32_t total = (uint32_t)__shfl((int)incl, 63);
      pos[0] = incl - mine;
#pragma unroll
      for (int j = 1; j < CC_DPL; j++) pos[j] = pos[j - 1] + ns[j - 1];
      for (uint32_t w0 = 0; w0 < total && distinct <= (uint32_t)CC_LIGHT_MAX; w0 += (uint32_t)CC_SRC) {
#pragma unroll
        for (int j = 0; j < CC_DPL; j++) {
          const uint32_t t0 = pos[j] < w0 ? w0 - pos[j] : 0u;
          const uint32_t t1 = min(ns[j], w0 + (uint32_t)CC_SRC > pos[j] ? w0 + (uint32_t)CC_SRC - pos[j] : 0u);
          for (uint32_t t = t0; t < t1; t++) src[pos[j] + t - w0] = de[j].x + t;
        }
        const uint32_t wn = min((uint32_t)CC_SRC, total - w0);
        for (uint32_t i0 = 0; i0 < wn && distinct <= (uint32_t)CC_LIGHT_MAX; i0 += 64 * CC_DEPTH) {
          uint32_t r[CC_DEPTH];
#pragma unroll
          for (int k = 0; k < CC_DEPTH; k++) { const uint32_t i = i0 + 64 * k + lane; r[k] = i < wn ? vs[src[i]] : CC_EMPTY; }
#pragma unroll
          for (int k0 = 0; k0 < CC_DEPTH; k0 += 4) {  // (the table's margin is 256 first sights between two looks)
            if (distinct <= (uint32_t)CC_LIGHT_MAX) {
#pragma unroll
              for (int k = k0; k < k0 + 4; k++) insert(r[k]);
            }
          }
        }
      }
      // the longer lists, by the whole wave
#pragma unroll
      for (int j = 0; j < CC_DPL; j++) {
        uint64_t big = __ballot(de[j].y > (uint32_t)CC_SHORT);
        while (big && distinct <= (uint32_t)CC_LIGHT_MAX) {  // (wave-uniform)
          const int sl = __builtin_ctzll(big);
          big &= big - 1ULL;
          const uint32_t lo = (uint32_t)__shfl((int)de[j].x, sl), n = (uint32_t)__shfl((int)de[j].y, sl);
          for (uint32_t i = 0; i < n && distinct <= (uint32_t)CC_LIGHT_MAX; i += 64) insert(i + lane < n ? vs[lo + i + lane] : CC_EMPTY);
        }
      }
    }
    // a table of `slots` takes partners <= slots / 2 whatever they are; only the full-size one can run over
    const bool over = distinct > (uint32_t)CC_LIGHT_MAX;
    if (over && lane == 0) heavy[atomicAdd(heavy_n, 1u)] = c;  // (at most one entry per column: the list has ng places)
    for (uint32_t i0 = 0; i0 < distinct; i0 += 64) {  // (wave-uniform trip count)
      bool keep = false;
      uint32_t rr = 0, common = 0;
      if (i0 + lane < distinct) {
        const uint32_t sl = seen[i0 + lane];
        rr = key[sl];
        common = cnt[sl];
        key[sl] = CC_EMPTY;
        cnt[sl] = 0;
        keep = !over && (radio < 0 || cc_keep(len[rr], s1, radio));
      }
      const uint64_t m = __ballot(keep);
      if (m) {
        if (nst + (uint32_t)__popcll(m) > (uint32_t)CC_STAGE) flush();
        if (keep) stage[nst + (uint32_t)__popcll(m & ((1ULL << lane) - 1ULL))] = rtc_cedge{rr, c, common};
        nst += (uint32_t)__popcll(m);
      }
    }
  }
  // what is still staged: one global atomic for the workgroup
  if (lane == 0) s_nst[wave] = nst;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t tot = 0;
    for (int w = 0; w < CC_WAVES; w++) tot += s_nst[w];
    s_base = tot ? atomicAdd(count, (unsigned long long)tot) : 0ull;
  }
  __syncthreads();
  unsigned long long base = s_base;
  for (uint32_t w = 0; w < wave; w++) base += s_nst[w];
  for (uint32_t i = lane; i < nst; i += 64) if (base + i < cap) edges[base + i] = stage[i];
}
// The heavy list: one 256-lane workgroup per column over the 64 KB a workgroup may declare.  First as ONE hash table row ->
// count of 8 192 slots for the whole workgroup (a column of a very large set has hundreds to thousands of chance partners
// spread over all the row ids): one walk over the column's partner lists, good for up to ~6 000 distinct partners (each
// wave stops at a quarter of that, so the probes always end).  Beyond that a counter per ROW id, CC_HW ids at a time: the
// lists are walked once per occupied id range (a first walk marks which ranges hold any) -- any number of distinct
// partners, nothing hashed.  Short lists by a lane each, longer ones by a wave.
constexpr uint32_t CC_HW = 16384 - 64, CC_HSLOTS = 8192, CC_HWAVE_MAX = 1536;
__global__ __launch_bounds__(256) void join_colcount_heavy_kernel(const uint32_t* __restrict__ vs, const uint2* __restrict__ desc,
                                                                  const uint64_t* __restrict__ off, const unsigned long long* __restrict__ colcnt,
                                                                  uint32_t g0, uint32_t row0, uint32_t row1, const uint32_t* __restrict__ len,
                                                                  int radio, rtc_cedge* __restrict__ edges, unsigned long long cap,
                                                                  unsigned long long* __restrict__ count, const uint32_t* __restrict__ heavy,
                                                                  const uint32_t* __restrict__ heavy_n) {
  __shared__ uint32_t lds[16384];
  uint32_t* hkey = lds;              // hash table form: keys and counts
  uint32_t* hcnt = lds + CC_HSLOTS;
  uint32_t* cnt = lds;               // id-range form: CC_HW counters and the occupancy bits
  uint32_t* s_occ = lds + CC_HW;     // bit p of word w: id range 32 w + p holds a partner (ranges past 2 048 share the last bit)
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t nh = *heavy_n;
  // f(id) for every partner id of the column; step() by the whole wave after every id per lane
  auto walk = [&](const uint2* d, uint32_t ne, auto&& f, auto&& step) {
    for (uint32_t e0 = 0; e0 < ne; e0 += 256) {  // (uniform trip count)
      const uint32_t e = e0 + threadIdx.x;
      const uint2 de = e < ne ? d[e] : make_uint2(0, 0);
      const uint32_t ns = de.y <= (uint32_t)CC_SHORT ? de.y : 0;
      uint32_t nmax = ns;  // the wave's longest short list
      for (int o = 32; o > 0; o >>= 1) nmax = max(nmax, (uint32_t)__shfl_xor((int)nmax, o));
      for (uint32_t t = 0; t < nmax; t++) {  // (wave-uniform)
        if (t < ns) f(vs[de.x + t]);
        step();
      }
      uint64_t big = __ballot(de.y > (uint32_t)CC_SHORT);
      while (big) {  // (wave-uniform)
        const int sl = __builtin_ctzll(big);
        big &= big - 1ULL;
        const uint32_t lo = (uint32_t)__shfl((int)de.x, sl), n = (uint32_t)__shfl((int)de.y, sl);
        for (uint32_t i0 = 0; i0 < n; i0 += 64) {  // (wave-uniform)
          if (i0 + lane < n) f(vs[lo + i0 + lane]);
          step();
        }
      }
    }
  };
  for (uint32_t h = blockIdx.x; h < nh; h += gridDim.x) {
    const uint32_t c = heavy[h];
    const uint32_t ne = (uint32_t)(colcnt[(size_t)(c - g0) * CC_CSTRIDE] >> CC_ESHIFT);
    const uint2* d = desc + off[c - g0];
    const uint32_t s1 = len[c];
    // ---- the hash table form ----
    for (uint32_t i = threadIdx.x; i < CC_HSLOTS; i += 256) { hkey[i] = CC_EMPTY; hcnt[i] = 0; }
    __syncthreads();
    uint32_t wave_fresh = 0, lane_fresh = 0;  // first sights by this wave (uniform) / by this lane since the last step
    walk(d, ne, [&](uint32_t id) {
      if (wave_fresh > CC_HWAVE_MAX) return;  // given up: the id-range form below starts over
      uint32_t at = cc_slot(id, CC_HSLOTS - 1);
      bool fresh = false;
      for (;;) {  // ends: the four waves together take at most 4 (CC_HWAVE_MAX + 64) of the 8 192 slots
        uint32_t o = hkey[at];
        if (o == CC_EMPTY) { o = atomicCAS(&hkey[at], CC_EMPTY, id); fresh = o == CC_EMPTY; }
        if (o == id || fresh) break;
        at = (at + 1) & (CC_HSLOTS - 1);
      }
      atomicAdd(&hcnt[at], 1u);
      lane_fresh += fresh;
    }, [&]() {  // after every id per lane: the wave's first sights so far
      wave_fresh += (uint32_t)__popcll(__ballot(lane_fresh != 0));
      lane_fresh = 0;
    });
    if (!__syncthreads_or(wave_fresh > CC_HWAVE_MAX)) {
      for (uint32_t i0 = 0; i0 < CC_HSLOTS; i0 += 256) {  // (uniform trip count)
        const uint32_t r = hkey[i0 + threadIdx.x];
        const bool keep = r != CC_EMPTY && (radio < 0 || cc_keep(len[r], s1, radio));
        const uint64_t m = __ballot(keep);
        if (m) {
          const int lead = __builtin_ctzll(m);
          unsigned long long at = 0;
          if ((int)lane == lead) at = atomicAdd(count, (unsigned long long)__popcll(m));
          at = ((unsigned long long)(uint32_t)__shfl((int)(at >> 32), lead) << 32) | (uint32_t)__shfl((int)(uint32_t)at, lead);
          if (keep) {
            const unsigned long long idx = at + __popcll(m & ((1ULL << lane) - 1ULL));
            if (idx < cap) edges[idx] = rtc_cedge{r, c, hcnt[i0 + threadIdx.x]};
          }
        }
      }
      __syncthreads();
      continue;
    }
    // ---- the id-range form ----
    const uint32_t rlo = max(c + 1, row0);  // a partner's id is above its column's
    const uint32_t npass = (row1 - rlo + CC_HW - 1) / CC_HW;
    if (threadIdx.x < 64) s_occ[threadIdx.x] = 0;
    __syncthreads();
    walk(d, ne, [&](uint32_t id) {
      const uint32_t p = min((id - rlo) / CC_HW, 2047u);
      atomicOr(&s_occ[p >> 5], 1u << (p & 31));
    }, [] {});
    __syncthreads();
    for (uint32_t p = 0; p < npass; p++) {
      const uint32_t pb = min(p, 2047u);
      if (!((s_occ[pb >> 5] >> (pb & 31)) & 1u)) continue;  // (uniform)
      const uint32_t base = rlo + p * CC_HW;
      for (uint32_t i = threadIdx.x; i < CC_HW; i += 256) cnt[i] = 0;
      __syncthreads();
      walk(d, ne, [&](uint32_t id) {
        const uint32_t dd = id - base;  // (below base: wraps past CC_HW)
        if (dd < CC_HW) atomicAdd(&cnt[dd], 1u);
      }, [] {});
      __syncthreads();
      for (uint32_t i0 = 0; i0 < CC_HW; i0 += 256) {  // (uniform trip count)
        const uint32_t i = i0 + threadIdx.x;
        const uint32_t common = i < CC_HW ? cnt[i] : 0;
        const uint32_t r = base + i;
        const bool keep = common != 0 && (radio < 0 || cc_keep(len[r], s1, radio));
        const uint64_t m = __ballot(keep);
        if (m) {
          const int lead = __builtin_ctzll(m);
          unsigned long long at = 0;
          if ((int)lane == lead) at = atomicAdd(count, (unsigned long long)__popcll(m));
          at = ((unsigned long long)(uint32_t)__shfl((int)(at >> 32), lead) << 32) | (uint32_t)__shfl((int)(uint32_t)at, lead);
          if (keep) {
            const unsigned long long idx = at + __popcll(m & ((1ULL << lane) - 1ULL));
            if (idx < cap) edges[idx] = rtc_cedge{r, c, common};
          }
        }
      }
      __syncthreads();
    }
    __syncthreads();
  }
}

inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }
// co-occurrences per second through the column kernel, the cost rule's figure: families of 40 .. 1 000 among 10 000 sketches
// (4.7e7 .. 1.1e9 co-occurrences) leave 0.85e11 .. 1.36e11 after the K-proportional part (tools/ab_join_families.sh,
// profiles/r05_join_column_tail_ab.txt)
constexpr double JOIN_E_RATE = 1.0e11;


template <typename T>
int join_impl(rtc_ctx* ctx, const T* d_hashes, const uint64_t* d_start, const uint32_t* d_len, uint32_t n, uint32_t row0,
              uint32_t row1, uint32_t col0, uint32_t col1, int radio, rtc_cedge* d_edges, uint64_t cap, uint64_t* d_count,
              double tiled_scale, int* handled) {
  *handled = 0;
  ctx->join_dense.edges_hint = 0;
  const int mode = tiled_scale < 0 ? 2 : ctx->opt.pair_join;  // 0: off, 1: where the cost rule says so, 2: wherever it can run (tests)  // tiled_scale < 0: rtc_warmup wants this path whatever the input
  if (mode <= 0 || ctx->pair_plan_hold) return RTC_OK;
  // only pairs (row, col) with col < row exist: genomes outside [g0, g1) take no part
  const uint32_t g0 = std::min(col0, row0), g1 = row1;
  if (g1 <= g0 + 1) return RTC_OK;
  const uint32_t ng = g1 - g0;
  hipStream_t s = ctx->stream;

  // ---- 1. offsets of the flat copy ----
  void* ws0 = nullptr;
  size_t tb_scan = 0;
  {
    auto it = rocprim::make_transform_iterator(d_len + g0, U32ToU64());
    RTC_HIP(ctx, rocprim::inclusive_scan(nullptr, tb_scan, it, (uint64_t*)nullptr, (size_t)ng, rocprim::plus<uint64_t>(), s));
  }
  const bool sampled = mode == 1;  // the cost rule is in charge: let it see the density before anything is sorted
  const size_t b_off = up256((size_t)(ng + 5) * 8), b_fix = up256((size_t)(FIX_CAP + 2) * 4), b_tab = sampled ? ((size_t)4 << SAMPLE_CELL_BITS) : 0;
  RTC_TRY(rtc_ws(ctx, 0, b_off + b_fix + up256(tb_scan) + b_tab + 256, &ws0));
  uint64_t* d_off = (uint64_t*)ws0;
  uint32_t* d_fix = (uint32_t*)((char*)ws0 + b_off);
  {
    auto it = rocprim::make_transform_iterator(d_len + g0, U32ToU64());
    RTC_HIP(ctx, hipMemsetAsync(d_off, 0, 8, s));
    RTC_HIP(ctx, hipMemsetAsync(d_off + ng + 1, 0, 32, s));  // [ng + 1]: the largest hash, [ng + 2 .. ng + 4]: the sample's sums
    void* tmp = (char*)ws0 + b_off + b_fix;
    RTC_HIP(ctx, rocprim::inclusive_scan(tmp, tb_scan, it, d_off + 1, (size_t)ng, rocprim::plus<uint64_t>(), s));
    hipLaunchKernelGGL(join_maxkey_kernel<T>, dim3((ng + 255) / 256), dim3(256), 0, s, d_hashes, d_start, d_len, g0, ng,
                       (unsigned long long*)(d_off + ng + 1));
    RTC_CHECK_LAUNCH(ctx);
    if (sampled) {
      uint32_t* d_tab = (uint32_t*)((char*)ws0 + b_off + b_fix + up256(tb_scan));
      RTC_HIP(ctx, hipMemsetAsync(d_tab, 0, b_tab, s));
      hipLaunchKernelGGL(join_sample_kernel<T>, dim3(ng), dim3(256), 0, s, d_hashes, d_start, d_len, g0, d_tab);
      RTC_CHECK_LAUNCH(ctx);
      hipLaunchKernelGGL(join_sample_sum_kernel, dim3(128), dim3(256), 0, s, (const uint32_t*)d_tab, (unsigned long long*)(d_off + ng + 2));
      RTC_CHECK_LAUNCH(ctx);
    }
  }
  void* hpin = nullptr;
  RTC_TRY(rtc_pinned(ctx, 64, &hpin));
  RTC_HIP(ctx, hipMemcpyAsync(hpin, d_off + ng, 16, hipMemcpyDeviceToHost, s));
  RTC_HIP(ctx, hipMemcpyAsync((char*)hpin + 24, d_off + ng + 2, 24, hipMemcpyDeviceToHost, s));
  RTC_HIP(ctx, hipMemcpyAsync((char*)hpin + 16, d_off + (std::max(row0, g0) - g0), 8, hipMemcpyDeviceToHost, s));
  RTC_HIP(ctx, hipStreamSynchronize(s));
  uint64_t K64 = *(const uint64_t*)hpin;
  const uint64_t K_all = K64;  // before the semi-join: what the tiled kernel would walk
  const uint64_t maxkey = ((const uint64_t*)hpin)[1];
  const uint64_t K_rows = K64 - ((const uint64_t*)hpin)[2];  // hashes of the row genomes [row0, row1)
  const double E_sample = sampled ? (double)((const uint64_t*)hpin)[3] * (double)(1u << SAMPLE_SHIFT) : 0.0;  // co-occurrences among all of [g0, g1), estimated
  if (K64 < 2) return RTC_OK;
  // the same input (same buffer and sketch generation, same counts, same largest hash, same tile) was found too dense
  // for the join a moment ago (repeated launches over one sketch set): straight to the tiled kernel, no second look
  auto& jd = ctx->join_dense;
  const bool seen_dense = jd.hashes == (const void*)d_hashes && jd.gen == ctx->sketch_gen && jd.K == K_all && jd.maxkey == maxkey &&
                          jd.n == n && jd.row0 == row0 && jd.row1 == row1 && jd.col0 == col0 && jd.col1 == col1;
  if (seen_dense && mode == 1) return RTC_OK;
  auto note_dense = [&]() { jd.hashes = d_hashes; jd.gen = ctx->sketch_gen; jd.n = n; jd.row0 = row0; jd.row1 = row1; jd.col0 = col0; jd.col1 = col1; jd.K = K_all; jd.maxkey = maxkey; };

  // ---- semi-join: columns keep only the hashes some row has (0: never, 1: when the rows hold less than a quarter of
  // the hashes, 2: whenever there is a column that is not a row) ----
  const int semi_mode = ctx->opt.join_semi;
  const bool semi = row0 > g0 && K_rows > 0 && (semi_mode >= 2 || (semi_mode == 1 && K_rows * 4 < K64 && K64 >= (1u << 22)));
  uint32_t* d_kept = nullptr;
  uint64_t* d_off_all = nullptr;  // semi-join: the places of the genomes among ALL hashes, and a "passed the filter" byte per hash
  uint8_t* d_passed = nullptr;
  unsigned long long* d_bloom = nullptr;
  int wshift = 0;
  if (semi) {
    uint64_t words = 1024;
    while (words * 8 < K_rows) words <<= 1;  // ~8 filter bits per hash
    wshift = 64;
    for (uint64_t w = words; w > 1; w >>= 1) wshift--;
    void* ws2 = nullptr;
    const size_t b_bloom = up256(words * 8), b_kept = up256((size_t)ng * 4), b_offall = up256((size_t)(ng + 1) * 8), b_passed = up256((size_t)K_all + 64);
    size_t tb_sk = 0;
    {
      auto it = rocprim::make_transform_iterator((const uint32_t*)nullptr, U32ToU64());
      RTC_HIP(ctx, rocprim::inclusive_scan(nullptr, tb_sk, it, (uint64_t*)nullptr, (size_t)ng, rocprim::plus<uint64_t>(), s));
    }
    {
      const int st = rtc_ws(ctx, 2, b_bloom + b_kept + up256(tb_sk) + b_offall + b_passed + 256, &ws2);
      if (st == RTC_ERR_NOMEM) return RTC_OK;
      if (st != RTC_OK) return st;
    }
    d_bloom = (unsigned long long*)ws2;
    d_kept = (uint32_t*)((char*)ws2 + b_bloom);
    void* tmpk = (char*)ws2 + b_bloom + b_kept;
    d_off_all = (uint64_t*)((char*)ws2 + b_bloom + b_kept + up256(tb_sk));
    d_passed = (uint8_t*)((char*)d_off_all + b_offall);
    RTC_HIP(ctx, hipMemcpyAsync(d_off_all, d_off, (size_t)(ng + 1) * 8, hipMemcpyDeviceToDevice, s));  // (d_off becomes the offsets of the kept hashes below)
    RTC_HIP(ctx, hipMemsetAsync(d_bloom, 0, words * 8, s));
    hipLaunchKernelGGL(join_bloom_build_kernel<T>, dim3(row1 - row0), dim3(256), 0, s, d_hashes, d_start, d_len, row0, wshift, d_bloom);
    RTC_CHECK_LAUNCH(ctx);
    hipLaunchKernelGGL((join_semi_kernel<T, false>), dim3(ng), dim3(256), 0, s, d_hashes, d_start, d_len, g0, row0, wshift,
                       (const unsigned long long*)d_bloom, d_kept, (const uint64_t*)nullptr, (T*)nullptr, (uint32_t*)nullptr,
                       (const uint64_t*)d_off_all, d_passed);
    RTC_CHECK_LAUNCH(ctx);
    auto it = rocprim::make_transform_iterator((const uint32_t*)d_kept, U32ToU64());
    RTC_HIP(ctx, rocprim::inclusive_scan(tmpk, tb_sk, it, d_off + 1, (size_t)ng, rocprim::plus<uint64_t>(), s));  // d_off[0] stays 0
    RTC_HIP(ctx, hipMemcpyAsync(hpin, d_off + ng, 8, hipMemcpyDeviceToHost, s));
    RTC_HIP(ctx, hipStreamSynchronize(s));
    K64 = *(const uint64_t*)hpin;
    if (ctx->opt.join_debug) fprintf(stderr, "[join] semi-join: %llu row hashes, %llu of the column + row hashes kept\n",
                                          (unsigned long long)K_rows, (unsigned long long)K64);
    if (K64 < 2) { *handled = 1; return RTC_OK; }
  }
  if (K64 >= (1ull << 31)) return RTC_OK;
  const uint32_t K = (uint32_t)K64;

  // ---- cost rule, first half: the sort alone against the tiled kernel's probes ----
  // tiled: every column of a 1024-column block probes the table of every 64-row block below the diagonal with all
  // of its hashes, ~4.0e11 probes/s (round 4's kernel, planning included); join: ~1.1e10 (u64) / 2.8e10 (u32) sorted keys/s
  // (tools/ubench/sort_rates.hip; the count and the column kernel's per-list work are in that figure), 1.0e11 co-occurrences/s through
  // the column kernel (the sort + encode + emit of rounds 3-4 ran at 1.5e10), measured on MI355X.
  const double avg = (double)K_all / ng;
  const double rows = (double)(row1 - row0);
  const double cols_mean = std::max(1.0, 0.5 * ((double)std::min(col1, row0) + (double)std::min(col1, row1 - 1)) - (double)col0);
  const double t_tiled = tiled_scale * (rows / 64.0 + 1.0) * cols_mean * avg / 4.0e11;
  const double t_sort = (double)K / (sizeof(T) == 8 ? 1.1e10 : 2.8e10);
  if (mode == 1 && t_sort > 0.7 * t_tiled) return RTC_OK;
  if (sampled) {
    // the sample's verdict, with a margin of 1.1 for its noise (what it lets through is still counted exactly below):
    // the tile holds rows x cols_mean of the ng (ng - 1) / 2 pairs the sample looked at
    const double frac = std::min(1.0, rows * cols_mean / (0.5 * (double)ng * (double)(ng - 1)));
    if (t_sort + E_sample * frac / JOIN_E_RATE > 1.1 * t_tiled) {
      // How many candidate edges such a set is likely to yield, for the caller's list: genomes that share a hash come in
      // groups about as large as the posting list a sampled hash sees (Sum c^2 / Sum c), and a group of g yields g (g - 1) / 2
      // pairs.  1.5 x that, never more than the tile holds; a list that turns out too short is grown and the launch redone as ever.
      const double sq = (double)((const uint64_t*)hpin)[4], su = (double)((const uint64_t*)hpin)[5];
      const double g = su > 0 ? sq / su : 1.0;
      const double pairs_tile = rows * cols_mean;
      jd.edges_hint = (uint64_t)std::min(pairs_tile, 1.5 * frac * 0.5 * (double)ng * std::max(0.0, g - 1.0)) + 1024;
      note_dense();
      return RTC_OK;
    }
  }

  const uint64_t avail = rtc_free_hbm(ctx) + ctx->ws_bytes[1];

  // ---- 2. flat copy + stable sort by hash ----
  size_t tb_sort = 0;
  RTC_HIP(ctx, rocprim::radix_sort_pairs(nullptr, tb_sort, (const T*)nullptr, (T*)nullptr, (const uint32_t*)nullptr,
                                         (uint32_t*)nullptr, (size_t)K, 0u, (unsigned)(8 * sizeof(T)), s));
  size_t tb_red = 0;
  {
    auto it = rocprim::make_transform_iterator(rocprim::make_counting_iterator<uint32_t>(0), ColPartners{nullptr});
    RTC_HIP(ctx, rocprim::reduce(nullptr, tb_red, it, (unsigned long long*)nullptr, 0ull, (size_t)ng, rocprim::plus<unsigned long long>(), s));
  }
  const size_t b_keys = up256((size_t)K * sizeof(T)), b_vals = up256((size_t)(K + 1) * 4);
  const size_t b_tmp1 = up256(std::max(std::max(tb_sort, tb_red), ((size_t)K / 256 + 2) * FIX_SLOTS * 4));  // (+ the half-key sort's rows of inversions)
  // keys0 | vals0 | keys1 | vals1 | temp | colcnt, total | heavy.  The descriptors (8 bytes per element) lie over
  // keys0 | vals0, which the sort has read by then.
  const size_t b_col = up256(((size_t)ng * CC_CSTRIDE + 8) * 8), b_heavy = up256((size_t)(ng + 4) * 4);
  const size_t need1 = 2 * b_keys + 2 * b_vals + b_tmp1 + b_col + b_heavy + 256;
  if (need1 > avail / 2) return RTC_OK;
  ctx->pair_plan_valid = 0;  // scratch slot 1 is the tiled plan's
  void* ws1 = nullptr;
  {
    const int st = rtc_ws(ctx, 1, need1, &ws1);
    if (st == RTC_ERR_NOMEM) return RTC_OK;
    if (st != RTC_OK) return st;
  }
  T* keys0 = (T*)ws1;
  uint32_t* vals0 = (uint32_t*)((char*)ws1 + b_keys);
  T* keys1 = (T*)((char*)ws1 + b_keys + b_vals);
  uint32_t* vals1 = (uint32_t*)((char*)ws1 + 2 * b_keys + b_vals);
  void* tmp1 = (char*)ws1 + 2 * b_keys + 2 * b_vals;
  uint2* d_desc = (uint2*)ws1;
  static_assert(sizeof(uint2) <= sizeof(T) + 4, "the descriptors fit the unsorted copy");
  // ---- 3. partners per element (descriptors by column), the co-occurrence count ----
  uint64_t* d_colcnt = (uint64_t*)((char*)ws1 + 2 * b_keys + 2 * b_vals + b_tmp1);  // a line per column genome: elements << 40 | partners
  uint64_t* d_total = d_colcnt + (size_t)ng * CC_CSTRIDE;                            // all partners
  uint32_t* d_heavy = (uint32_t*)((char*)d_colcnt + b_col);  // [0]: entries, [4..): columns for the heavy kernel
  uint64_t E = 0;
  // radix passes only over the bits that vary: [0, end_bit) holds every hash.  u64: first on the 32 bits below
  // end_bit + repair of the rare mixed runs; when the repair gives up (a collision inside a very long posting
  // list), once more on all of them.  rocPRIM 4.2 mis-sorts ranges [b > 0, 64) below ~1M keys (its merge-sort
  // path, tools/ubench/sort_check.hip): those inputs take the full range.
  unsigned end_bit = 1;
  while (end_bit < 8 * sizeof(T) && (maxkey >> end_bit)) end_bit++;
  // distinct hashes that agree in the b sorted bits: about K^2 / 2^(b + 3) inversions (measured 3 535 at K = 10^7, b = 32);
  // b grows by a radix pass (8 bits) while they would cost more in the repair list's atomics than the pass does
  unsigned sort_bits = 32;
  while (sort_bits < 64 && (double)K * (double)K / std::ldexp(1.0, (int)sort_bits + 3) > (double)FIX_EXPECT_MAX) sort_bits += 8;
  const unsigned half_bit = end_bit > sort_bits ? end_bit - sort_bits : 0;
  for (int attempt = 0; attempt < 2; attempt++) {
    const bool halfsort = sizeof(T) == 8 && attempt == 0 && half_bit > 0 && (end_bit < 64 || K >= (1u << 22)) &&
                          !ctx->opt.join_fullsort;
    // (the flat copy anew for the second attempt: the first one's descriptors lie over it)
    if (semi)
      hipLaunchKernelGGL((join_semi_kernel<T, true>), dim3(ng), dim3(256), 0, s, d_hashes, d_start, d_len, g0, row0, wshift,
                         (const unsigned long long*)d_bloom, (uint32_t*)nullptr, (const uint64_t*)d_off, keys0, vals0,
                         (const uint64_t*)d_off_all, d_passed);
    else
      hipLaunchKernelGGL(join_flatten_kernel<T>, dim3(ng), dim3(256), 0, s, d_hashes, d_start, d_len, d_off, g0, keys0, vals0);
    RTC_CHECK_LAUNCH(ctx);
    RTC_HIP(ctx, rocprim::radix_sort_pairs(tmp1, tb_sort, (const T*)keys0, keys1, (const uint32_t*)vals0, vals1, (size_t)K,
                                           halfsort ? half_bit : 0u, end_bit, s));
    if constexpr (sizeof(T) == 8) {
      if (halfsort) {
        // (the workgroups' rows of inversions lie in the sort's temporary storage: the sort is done, the reduction below comes later)
        uint32_t* d_slots = (uint32_t*)tmp1;
        const uint32_t nwg = (K + 255) / 256, nslots = nwg * FIX_SLOTS;
        RTC_HIP(ctx, hipMemsetAsync(d_fix, 0, 8, s));
        hipLaunchKernelGGL(join_inversions_kernel, dim3(nwg), dim3(256), 0, s, (const uint64_t*)keys1, K, (int)half_bit, d_fix, d_slots);
        RTC_CHECK_LAUNCH(ctx);
        const uint32_t nent = nslots + FIX_CAP;
        hipLaunchKernelGGL(join_repair_owner_kernel, dim3((nent + 63) / 64), dim3(64), 0, s, (const uint64_t*)keys1, (int)half_bit, d_fix, d_slots, nslots);
        RTC_CHECK_LAUNCH(ctx);
        hipLaunchKernelGGL(join_repair_kernel, dim3((nent + 63) / 64), dim3(64), 0, s, (uint64_t*)keys1, vals1, K, (int)half_bit, d_fix, d_slots, nslots);
        RTC_CHECK_LAUNCH(ctx);
      }
    }
    RTC_HIP(ctx, hipMemsetAsync(d_colcnt, 0, (size_t)ng * CC_CSTRIDE * 8, s));
    hipLaunchKernelGGL(join_count_kernel<T>, dim3((K + 255) / 256), dim3(256), 0, s, (const T*)keys1, (const uint32_t*)vals1, K,
                       row0, col0, col1, g0, (const uint64_t*)d_off, (unsigned long long*)d_colcnt, d_desc);
    RTC_CHECK_LAUNCH(ctx);
    {
      auto it = rocprim::make_transform_iterator(rocprim::make_counting_iterator<uint32_t>(0), ColPartners{(const unsigned long long*)d_colcnt});
      RTC_HIP(ctx, rocprim::reduce(tmp1, tb_red, it, (unsigned long long*)d_total, 0ull, (size_t)ng, rocprim::plus<unsigned long long>(), s));
    }
    RTC_HIP(ctx, hipMemcpyAsync(hpin, d_total, 8, hipMemcpyDeviceToHost, s));
    if (halfsort) RTC_HIP(ctx, hipMemcpyAsync((char*)hpin + 8, d_fix, 8, hipMemcpyDeviceToHost, s));
    RTC_HIP(ctx, hipStreamSynchronize(s));
    E = *(const uint64_t*)hpin;
    if (ctx->opt.join_debug) fprintf(stderr, "[join] K=%u attempt=%d halfsort=%d E=%llu (sample: %.3g) inversions=%u giveup=%u t_sort=%.3g t_tiled=%.3g ms\n", K, attempt, (int)halfsort,
                                          (unsigned long long)E, E_sample, halfsort ? ((const uint32_t*)hpin)[2] : 0u, halfsort ? ((const uint32_t*)hpin)[3] : 0u, t_sort * 1e3, t_tiled * 1e3);
    if (halfsort && ((const uint32_t*)hpin)[3]) {  // not repaired: sort on all bits -- unless the (approximate) count
      // of the unrepaired lists already says the input is dense: then the tiled kernel runs, without a second sort
      if (E >= (1ull << 31) || (mode == 1 && t_sort + (double)E / JOIN_E_RATE > t_tiled)) { note_dense(); return RTC_OK; }  // (a second sort is still to come)
      continue;
    }
    break;
  }
  if (E == 0) { *handled = 1; return RTC_OK; }  // no two genomes of the tile share a hash: no candidates
  if (E >= (1ull << 31)) { note_dense(); return RTC_OK; }
  // ---- cost rule, second half: the sort and the count are paid for; what is still to come against the tiled kernel ----
  if (mode == 1 && (double)E / JOIN_E_RATE > t_tiled) { note_dense(); return RTC_OK; }

  // ---- 4. the column-centric tail ----
  RTC_HIP(ctx, hipMemsetAsync(d_heavy, 0, 4, s));
  const uint32_t c_lo = std::max(col0, g0), c_hi = std::min(col1, g1);
  // a column per wave; more (up to 8) when the columns outnumber the chip's wave slots (1 / 2 / 4 per wave at 10 000 columns:
  // pair phase 0.99-1.00 / 1.02-1.03 / 1.03-1.05 ms on one box, profiles/r05_join_column_tail_ab.txt)
  const uint32_t cpw = std::min<uint32_t>(8, std::max<uint32_t>(1, (c_hi - c_lo) / (256 * 16 * 2)));
  const uint32_t nwaves = (c_hi - c_lo + cpw - 1) / cpw;
  hipLaunchKernelGGL(join_colcount_kernel, dim3((nwaves + CC_WAVES - 1) / CC_WAVES), dim3(64 * CC_WAVES), 0, s, (const uint32_t*)vals1,
                     (const uint2*)d_desc, (const uint64_t*)d_off, (const unsigned long long*)d_colcnt, g0, c_lo, c_hi, cpw, d_len, radio, d_edges,
                     (unsigned long long)cap, (unsigned long long*)d_count, d_heavy + 4, d_heavy);
  RTC_CHECK_LAUNCH(ctx);
  hipLaunchKernelGGL(join_colcount_heavy_kernel, dim3(512), dim3(256), 0, s, (const uint32_t*)vals1, (const uint2*)d_desc, (const uint64_t*)d_off,
                     (const unsigned long long*)d_colcnt, g0, row0, row1, d_len, radio, d_edges, (unsigned long long)cap, (unsigned long long*)d_count,
                     (const uint32_t*)(d_heavy + 4), (const uint32_t*)d_heavy);
  RTC_CHECK_LAUNCH(ctx);
  *handled = 1;
  return RTC_OK;
}

}  // namespace

// Candidate edges (i, j, common) of rows [row0, row1) x cols [col0, col1), j < i, appended at *d_count like the
// tiled kernel does.  *handled = 0: the caller runs the tiled kernel (nothing was appended).  tiled_scale: what
// fraction of the tile the caller's alternative would really walk (1 for the MST flows; the greedy loop only
// measures queries against representatives).
int rtc_pair_edges_join(rtc_ctx* ctx, const void* d_hashes, int width, const uint64_t* d_start, const uint32_t* d_len,
                        uint32_t n, uint32_t row0, uint32_t row1, uint32_t col0, uint32_t col1, int radio,
                        rtc_cedge* d_edges, uint64_t cap, uint64_t* d_count, double tiled_scale, int* handled) {
  *handled = 0;
  if (n < 2) return RTC_OK;
  if (width == 8)
    return join_impl<uint64_t>(ctx, (const uint64_t*)d_hashes, d_start, d_len, n, row0, row1, col0, col1, radio, d_edges, cap,
                               d_count, tiled_scale, handled);
  return join_impl<uint32_t>(ctx, (const uint32_t*)d_hashes, d_start, d_len, n, row0, row1, col0, col1, radio, d_edges, cap,
                             d_count, tiled_scale, handled);
}

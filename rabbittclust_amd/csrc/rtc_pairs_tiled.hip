// rtc_pairs_tiled.hip -- all-pairs |A_i ∩ A_j| with an LDS-resident row-block inverted table.
//
// Dense (every pair is evaluated, no data-dependent skipping of pairs) but work-efficient form of
// the reference's inverted-index intersection (src/MST.cpp:1408-1435): instead of one merge per
// pair, a workgroup owns a block of 64 rows x 1024 columns and, for each hash-range partition p,
//   1. builds in LDS an open-addressing table  hash -> 64-bit mask of the rows containing it
//      (bucketised, 4 keys per 32-byte bucket; ds_cmpst_b64 / ds_or_b64), from the rows' sorted
//      slices that fall in partition p;
//   2. every lane (= one column sketch) streams its own slice of partition p and probes the table
//      with two ds_read_b128; a hit returns the mask of ALL 64 rows containing that hash;
//   3. masks are accumulated in per-lane bit-sliced counters (plane k holds bit k of the 64 row
//      counters), i.e. one probe serves 64 pairs and the add is a wave-uniform ripple of
//      AND/XOR on 64-bit registers.
// The column slices are first copied into a partition-major, element-major, column-minor layout
// (tcols[base[p] + e*n + c]) so that the probe loop's loads are coalesced across lanes (= columns).
// After the last partition each lane unpacks its 64 counters and writes them (coalesced across
// lanes) to common[row][col].  Partition boundaries are data quantiles computed from a sample, so
// the table load stays ~25 %; blocks whose slices would overflow the table are split into row
// sub-blocks inside the kernel, and inputs the scheme cannot take fall back to the merge kernel.
//
// Two output forms of the same kernel: EMIT_DENSE stores the counts into common[row][col];
// EMIT_EDGES applies the reference's pair filters (src/MST.cpp:1468-1487: j < i, common > 0, size
// ratio "radio") to the 64 counters a lane holds and appends the survivors (i, j, common) straight
// to the candidate-edge list -- one global atomic per workgroup reserves the range, no dense matrix
// is written or re-read.
#include <algorithm>
#include <vector>

#include "rtc_internal.h"

namespace {

constexpr int TW = 1024;                // lanes per workgroup = columns per block (512: two per CU, measured 60 % slower)
constexpr int ROWS = 64;                // rows per block = mask width
constexpr int SLOTS = 8 * TW;           // table slots
constexpr int BUCKET = 4;               // keys per bucket
constexpr int NB = SLOTS / BUCKET;      // buckets
constexpr int LOG2NB = TW == 1024 ? 11 : 10;
constexpr uint32_t KCAP_HARD = SLOTS / 16 * 11;  // max keys per table build (~69 % load)
constexpr uint32_t KTARGET = SLOTS / 4;          // planned mean keys per table (25 % load)
constexpr int RPW = ROWS / (TW / 64);   // rows a wave builds at once
constexpr int LPR = 64 / RPW;           // lanes per row in the build
constexpr int MAXP = 512;
constexpr int DEPTH = 2;                // trips (of four keys) of a column's slice requested ahead of the probes (3 and 4 measured slower)

template <typename T> struct KeyTraits;
template <> struct KeyTraits<uint64_t> {
  static constexpr uint64_t EMPTY = ~0ULL;
  __device__ static __forceinline__ uint32_t bucket(uint64_t k) {
    return (((uint32_t)k ^ (uint32_t)(k >> 32)) * 0x9E3779B1u) >> (32 - LOG2NB);
  }
};
template <> struct KeyTraits<uint32_t> {
  static constexpr uint32_t EMPTY = ~0u;
  __device__ static __forceinline__ uint32_t bucket(uint32_t k) { return (k * 0x9E3779B1u) >> (32 - LOG2NB); }
};

struct TileShared {
  uint32_t rlo[ROWS], rhi[ROWS];
  uint64_t rstart[ROWS];
  unsigned long long special;  // rows containing the EMPTY sentinel value itself
  uint32_t nsub;
  uint32_t sub_end[ROWS + 1];
  // edge emission
  uint32_t rlen[ROWS];
  uint32_t wave_tot[TW / 64];
  unsigned long long gbase;
};

enum { EMIT_DENSE = 0, EMIT_EDGES = 1 };
struct EdgeSink {                 // EMIT_EDGES only
  const uint32_t* len;            // sketch lengths (radio test)
  rtc_cedge* edges;
  unsigned long long cap;
  unsigned long long* count;
  int radio;                      // < 0: no size-ratio test
};

__device__ __forceinline__ unsigned long long lds_cas(unsigned long long* p, unsigned long long cmp, unsigned long long v) {
  return atomicCAS(p, cmp, v);
}
__device__ __forceinline__ uint32_t lds_cas(uint32_t* p, uint32_t cmp, uint32_t v) { return atomicCAS(p, cmp, v); }

// Beside the keys, a bucket keeps four 15-bit fingerprints (low key bits) in 8 bytes and, in bit 31 of the
// second word, an "overflowed" flag set by an insert that had to move past the full bucket: the probe's
// straight line reads only those 8 bytes (one ds_read_b64) and goes to the keys when a fingerprint matches
// or the bucket overflowed.
template <typename T> __device__ __forceinline__ uint32_t fingerprint(T key) { return (uint32_t)key & 0x7fffu; }

template <typename T>
__device__ __forceinline__ void table_insert(T* keys, unsigned long long* masks, uint32_t* fps, TileShared* sh, T key, int r) {
  const unsigned long long bit = 1ULL << r;
  if (key == KeyTraits<T>::EMPTY) { atomicOr(&sh->special, bit); return; }
  uint32_t b = KeyTraits<T>::bucket(key);
  while (true) {
#pragma unroll
    for (int j = 0; j < BUCKET; j++) {
      const uint32_t slot = b * BUCKET + j;
      T old;
      if constexpr (sizeof(T) == 8) old = (T)lds_cas((unsigned long long*)&keys[slot], (unsigned long long)KeyTraits<T>::EMPTY, (unsigned long long)key);
      else old = (T)lds_cas((uint32_t*)&keys[slot], (uint32_t)KeyTraits<T>::EMPTY, (uint32_t)key);
      if (old == KeyTraits<T>::EMPTY || old == key) {
        atomicOr(&masks[slot], bit);
        if (old == KeyTraits<T>::EMPTY) atomicOr(&fps[b * 2 + (j >> 1)], fingerprint(key) << (16 * (j & 1)));
        return;
      }
    }
    atomicOr(&fps[b * 2 + 1], 0x80000000u);  // overflowed
    b = (b + 1) & (NB - 1);
  }
}

__device__ __forceinline__ unsigned long long table_lookup(const uint64_t* keys, const unsigned long long* masks,
                                                           const TileShared* sh, uint64_t key) {
  if (key == ~0ULL) return sh->special;
  uint32_t b = KeyTraits<uint64_t>::bucket(key);
  while (true) {
    const ulonglong2 k01 = *reinterpret_cast<const ulonglong2*>(&keys[b * BUCKET]);
    const ulonglong2 k23 = *reinterpret_cast<const ulonglong2*>(&keys[b * BUCKET + 2]);
    int hit = -1;
    if (k01.x == key) hit = 0;
    if (k01.y == key) hit = 1;
    if (k23.x == key) hit = 2;
    if (k23.y == key) hit = 3;
    if (hit >= 0) return masks[b * BUCKET + hit];
    if (k01.x == ~0ULL || k01.y == ~0ULL || k23.x == ~0ULL || k23.y == ~0ULL) return 0ULL;
    b = (b + 1) & (NB - 1);
  }
}

__device__ __forceinline__ unsigned long long table_lookup(const uint32_t* keys, const unsigned long long* masks,
                                                           const TileShared* sh, uint32_t key) {
  if (key == ~0u) return sh->special;
  uint32_t b = KeyTraits<uint32_t>::bucket(key);
  while (true) {
    const uint4 k = *reinterpret_cast<const uint4*>(&keys[b * BUCKET]);
    int hit = -1;
    if (k.x == key) hit = 0;
    if (k.y == key) hit = 1;
    if (k.z == key) hit = 2;
    if (k.w == key) hit = 3;
    if (hit >= 0) return masks[b * BUCKET + hit];
    if (k.x == ~0u || k.y == ~0u || k.z == ~0u || k.w == ~0u) return 0ULL;
    b = (b + 1) & (NB - 1);
  }
}

// so: [(P+1)][n] slice offsets (so[p][g] = lower_bound(sketch g, bound[p])), so[0]=0, so[P]=len.
// tcols covers the column range [tc0, tc0 + tnc): element e of column c's slice in partition p sits at
// tcols[tbase[p] + e*tnc + (c - tc0)].
template <typename T, int NPL, int EMIT>
__global__ __launch_bounds__(TW, TW == 1024 ? 1 : 2) void pair_tiled_kernel(const T* __restrict__ hashes,
                                                        const uint64_t* __restrict__ start,
                                                        const T* __restrict__ tcols,          // transposed column slices
                                                        const uint64_t* __restrict__ tbase,   // [P] element offsets
                                                        const uint32_t* __restrict__ so, int P, uint32_t n,
                                                        uint32_t tc0, uint32_t tnc,
                                                        uint32_t row0, uint32_t row1, uint32_t col0,
                                                        uint32_t col1, uint32_t* __restrict__ out, uint64_t ld,
                                                        int lower_only, EdgeSink sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* keys = reinterpret_cast<T*>(smem);
  unsigned long long* masks = reinterpret_cast<unsigned long long*>(smem + (size_t)SLOTS * sizeof(T));
  uint32_t* fps = reinterpret_cast<uint32_t*>(smem + (size_t)SLOTS * (sizeof(T) + 8));  // [NB][2]
  TileShared* sh = reinterpret_cast<TileShared*>(smem + (size_t)SLOTS * (sizeof(T) + 8 + 2));

  const int tid = threadIdx.x;
  // Row blocks vary fastest in the grid: the workgroups running at the same time (round-robin over
  // the XCDs) then probe the same 1024-column slab, which each XCD's L2 fetches once instead of
  // once per row block.
  const uint32_t rb0 = row0 + blockIdx.x * ROWS;
  const uint32_t nrows = min((uint32_t)ROWS, row1 - rb0);
  const uint32_t cb0 = col0 + blockIdx.y * TW;
  if (lower_only && cb0 + 1 > rb0 + nrows - 1) return;  // no (row, col) with col < row in this block
  const uint32_t c = cb0 + tid;
  const bool col_active = c < col1;
  const uint32_t wave = tid >> 6, lane = tid & 63;

  unsigned long long planes[NPL];
#pragma unroll
  for (int k = 0; k < NPL; k++) planes[k] = 0ULL;

  uint32_t clo = col_active ? so[c] : 0;  // so[0][c]
  if (tid < ROWS) sh->rstart[tid] = tid < (int)nrows ? start[rb0 + tid] : 0;
  // Every global load a partition needs is requested one partition (or one phase) ahead: the column's next
  // slice end, the rows' slice bounds (lanes 0..63), the first four probe keys (before the table is built).
  const bool rowlane = tid < (int)nrows;
  uint32_t chi_n = col_active ? so[(size_t)n + c] : 0;
  uint32_t rlo_n = rowlane ? so[rb0 + tid] : 0, rhi_n = rowlane ? so[(size_t)n + rb0 + tid] : 0;

  for (int p = 0; p < P; p++) {
    const uint32_t chi = chi_n;
    const T* tp = tcols + tbase[p] + (c - tc0);
    T nq[DEPTH][4];  // rows past the slice's end hold other data (the copy is padded by 4 * DEPTH rows): masked by `rem`
#pragma unroll
    for (int d = 0; d < DEPTH; d++)
#pragma unroll
      for (int j = 0; j < 4; j++) nq[d][j] = col_active ? tp[(size_t)(4 * d + j) * tnc] : (T)0;
    if (p + 1 < P) chi_n = col_active ? so[(size_t)(p + 2) * n + c] : 0;
    __syncthreads();  // previous partition's probes are done (table and rlo/rhi reusable)
    if (tid < ROWS) { sh->rlo[tid] = rlo_n; sh->rhi[tid] = rhi_n; }
    if (p + 1 < P) { rlo_n = rhi_n; rhi_n = rowlane ? so[(size_t)(p + 2) * n + rb0 + tid] : 0; }
    __syncthreads();
    if (wave == 0) {  // split the row block only if its keys would overflow one table (rare)
      uint32_t sz = (lane < nrows) ? sh->rhi[lane] - sh->rlo[lane] : 0;
      uint32_t tot = sz;
      for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
      if (tot <= KCAP_HARD) {
        if (lane == 0) { sh->sub_end[0] = nrows; sh->nsub = 1; }
      } else if (lane == 0) {
        uint32_t ns = 0, acc = 0;
        for (uint32_t r = 0; r < nrows; r++) {
          const uint32_t szr = sh->rhi[r] - sh->rlo[r];
          if (acc + szr > KCAP_HARD && acc > 0) { sh->sub_end[ns++] = r; acc = 0; }
          acc += szr;
        }
        sh->sub_end[ns++] = nrows;
        sh->nsub = ns;
      }
    }
    __syncthreads();
    const uint32_t nsub = sh->nsub;
    uint32_t ra = 0;
    for (uint32_t sb = 0; sb < nsub; sb++) {
      const uint32_t rbnd = sh->sub_end[sb];
      // ---- clear ----
      {
        uint4* kq = reinterpret_cast<uint4*>(keys);
        for (int i = tid; i < (int)(SLOTS * sizeof(T) / 16); i += TW) kq[i] = make_uint4(~0u, ~0u, ~0u, ~0u);
        uint4* mq = reinterpret_cast<uint4*>(masks);  // masks and fingerprints are adjacent
        for (int i = tid; i < SLOTS * (8 + 2) / 16; i += TW) mq[i] = make_uint4(0u, 0u, 0u, 0u);
        if (tid == 0) sh->special = 0ULL;
      }
      __syncthreads();
      // ---- build: LPR lanes per row, all (<= 64) rows at once, four loads in flight per lane ----
      {
        const uint32_t r = ra + wave * RPW + lane / LPR;
        if (r < rbnd) {
          const T* rp = hashes + sh->rstart[r];
          const uint32_t hi = sh->rhi[r];
          for (uint32_t e = sh->rlo[r] + (lane % LPR); e < hi; e += 4 * LPR) {
            T kq[4];
#pragma unroll
            for (int j = 0; j < 4; j++) kq[j] = (e + LPR * j < hi) ? rp[e + LPR * j] : (T)0;
#pragma unroll
            for (int j = 0; j < 4; j++) if (e + LPR * j < hi) table_insert<T>(keys, masks, fps, sh, kq[j], (int)r);
          }
        }
      }
      __syncthreads();
      // ---- probe: this lane's column slice, read coalesced from the transposed copy ----
      // Four keys per trip.  The next trip's global loads are issued before this trip's table work;
      // the four home buckets are read back to back (no data-dependent loop in the common case),
      // the row masks are fetched only by waves in which some lane hit, and the rare key whose home
      // bucket is full without a match (or that equals the EMPTY marker) takes the looping lookup.
      {
        const uint32_t mylen = chi - clo;
        if (sb > 0) {
#pragma unroll
          for (int d = 0; d < DEPTH; d++)
#pragma unroll
            for (int j = 0; j < 4; j++) nq[d][j] = col_active ? tp[(size_t)(4 * d + j) * tnc] : (T)0;
        }
        for (uint32_t e = 0; e < mylen; e += 4) {
          T bq[4];  // DEPTH trips of keys are in flight: the walk is short, the memory far
#pragma unroll
          for (int j = 0; j < 4; j++) {
            bq[j] = nq[0][j];
#pragma unroll
            for (int d = 0; d + 1 < DEPTH; d++) nq[d][j] = nq[d + 1][j];
          }
#pragma unroll
          for (int j = 0; j < 4; j++) nq[DEPTH - 1][j] = (e + 4 * DEPTH + j < mylen) ? tp[(size_t)(e + 4 * DEPTH + j) * tnc] : (T)0;
          // Fingerprint words of the four home buckets, read back to back; per key four 16-bit compares and
          // the overflow bit, all as wave masks.  A lane whose fingerprint matches (its key may be in the
          // home bucket) or whose bucket overflowed (it may sit further on) takes the looping lookup on the
          // keys below -- a few waves of the column block whose family lies in this row block, and the ~0.4 %
          // of the buckets that overflow; the others never leave this straight line.
          const uint32_t rem = mylen - e;
          uint64_t todo[4];
          uint2 fw[4];
#pragma unroll
          for (int j = 0; j < 4; j++) fw[j] = *reinterpret_cast<const uint2*>(&fps[KeyTraits<T>::bucket(bq[j]) * 2]);
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const uint32_t q = fingerprint(bq[j]);
            const uint64_t m = __ballot((fw[j].x & 0xffffu) == q) | __ballot((fw[j].x >> 16) == q) |
                               __ballot((fw[j].y & 0xffffu) == q) | __ballot(((fw[j].y >> 16) & 0x7fffu) == q);
            todo[j] = (m | __ballot((int32_t)fw[j].y < 0)) & __ballot(rem > (uint32_t)j);
          }
          if (p == P - 1) {  // the EMPTY marker as a key (only the largest value of the last partition): sh->special
#pragma unroll
            for (int j = 0; j < 4; j++) todo[j] |= __ballot(bq[j] == KeyTraits<T>::EMPTY && rem > (uint32_t)j);
          }
          if (!(todo[0] | todo[1] | todo[2] | todo[3])) continue;
#pragma unroll
          for (int j = 0; j < 4; j++) {
            if (!todo[j]) continue;  // wave-uniform
            unsigned long long carry = 0ULL;
            if (__builtin_amdgcn_inverse_ballot_w64(todo[j])) carry = table_lookup(keys, masks, sh, bq[j]);
#pragma unroll
            for (int k = 0; k < NPL; k++) {
              if (!__ballot(carry != 0ULL)) break;  // wave-uniform
              const unsigned long long t = planes[k] & carry;
              planes[k] ^= carry;
              carry = t;
            }
          }
        }
      }
      if (sb + 1 < nsub) __syncthreads();
      ra = rbnd;
    }
    clo = chi;
  }

  if constexpr (EMIT == EMIT_DENSE) {
    // ---- unpack the 64 bit-sliced counters of this lane's column ----
    for (uint32_t r = 0; r < nrows; r++) {
      uint32_t cnt = 0;
#pragma unroll
      for (int k = 0; k < NPL; k++) cnt |= (uint32_t)((planes[k] >> r) & 1ULL) << k;
      const uint32_t row = rb0 + r;
      if (col_active && (!lower_only || c < row)) out[(uint64_t)(row - row0) * ld + (c - col0)] = cnt;
    }
  } else {
    // ---- candidate edges: filter the 64 counters as bit masks, reserve a range, write survivors ----
    // bit r of `keep`: pair (row rb0 + r, column c) becomes an EdgeInfo in the reference
    // (src/MST.cpp:1468-1487): common > 0 (which implies both sketches non-empty), j < i, and
    // max(|A|,|B|) <= radio * min(|A|,|B|).
    if (tid < ROWS) sh->rlen[tid] = tid < (int)nrows ? sink.len[rb0 + tid] : 0;
    unsigned long long keep = 0ULL;
#pragma unroll
    for (int k = 0; k < NPL; k++) keep |= planes[k];
    if (nrows < (uint32_t)ROWS) keep &= (1ULL << nrows) - 1ULL;
    if (!col_active) keep = 0ULL;
    if (lower_only && c >= rb0) {  // rows rb0 + r > c  <=>  r > c - rb0
      const uint32_t d = c - rb0 + 1;
      keep = d >= 64 ? 0ULL : (keep & (~0ULL << d));
    }
    __syncthreads();  // rlen visible
    if (sink.radio >= 0 && keep) {
      const uint32_t s1 = sink.len[c];
      unsigned long long m = keep;
      while (m) {
        const int r = __builtin_ctzll(m);
        m &= m - 1ULL;
        const uint32_t s0 = sh->rlen[r];
        const uint32_t mn = s0 < s1 ? s0 : s1, mx = s0 > s1 ? s0 : s1;
        if ((uint64_t)mx > (uint64_t)(uint32_t)sink.radio * (uint64_t)mn) keep &= ~(1ULL << r);  // src/MST.cpp:1484
      }
    }
    const uint32_t mine = (uint32_t)__popcll(keep);
    // inclusive prefix sum within the wave
    uint32_t incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t v = (uint32_t)__shfl_up((int)incl, o);
      if (lane >= (uint32_t)o) incl += v;
    }
    if (lane == 63) sh->wave_tot[wave] = incl;
    __syncthreads();
    uint32_t wbase = 0, total = 0;
#pragma unroll
    for (int w = 0; w < TW / 64; w++) {
      const uint32_t wt = sh->wave_tot[w];
      if ((uint32_t)w < wave) wbase += wt;
      total += wt;
    }
    if (total == 0) return;  // workgroup-uniform
    if (tid == 0) sh->gbase = atomicAdd(sink.count, (unsigned long long)total);
    __syncthreads();
    unsigned long long idx = sh->gbase + wbase + (incl - mine);
    while (keep) {
      const int r = __builtin_ctzll(keep);
      keep &= keep - 1ULL;
      uint32_t cnt = 0;
#pragma unroll
      for (int k = 0; k < NPL; k++) cnt |= (uint32_t)((planes[k] >> r) & 1ULL) << k;
      if (idx < sink.cap) sink.edges[idx] = rtc_cedge{rb0 + (uint32_t)r, c, cnt};
      idx++;
    }
  }
}

// so[p][g] = lower_bound(sketch g, bound[p]) for p = 0..P (so[0] = 0, so[P] = len): one lane per (p, g)
template <typename T>
__global__ __launch_bounds__(256) void slice_offsets_kernel(const T* __restrict__ hashes, const uint64_t* __restrict__ start,
                                                            const uint32_t* __restrict__ len, const T* __restrict__ bounds, int P,
                                                            uint32_t n, uint32_t* __restrict__ so) {
  const uint32_t p = blockIdx.y;  // 0 .. P
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  const uint32_t L = len[g];
  uint32_t r;
  if (p == 0) r = 0;
  else if (p == (uint32_t)P) r = L;
  else {
    const T* a = hashes + start[g];
    const T b = bounds[p];
    uint32_t lo = 0, hi = L;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (a[mid] < b) lo = mid + 1; else hi = mid; }
    r = lo;
  }
  so[(size_t)p * n + g] = r;
}

// slice lengths: pmax[p] = maximum over the columns [c0, c1) (sizes the transposed copy), amax[0] = maximum over every
// sketch and partition (a single slice must fit one table build); one lane per (p, g), coalesced over g
__global__ __launch_bounds__(256) void slice_max_kernel(const uint32_t* __restrict__ so, uint32_t n, uint32_t c0, uint32_t c1,
                                                        uint32_t* __restrict__ amax, uint32_t* __restrict__ pmax) {
  const uint32_t p = blockIdx.y;
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t d = 0, dc = 0;
  if (g < n) {
    d = so[(size_t)(p + 1) * n + g] - so[(size_t)p * n + g];
    dc = (g >= c0 && g < c1) ? d : 0;
  }
  for (int o = 32; o > 0; o >>= 1) {
    d = max(d, (uint32_t)__shfl_xor((int)d, o));
    dc = max(dc, (uint32_t)__shfl_xor((int)dc, o));
  }
  if ((threadIdx.x & 63) == 0) {
    if (dc) atomicMax(&pmax[p], dc);
    if (d) atomicMax(amax, d);
  }
}

// tcols[tbase[p] + e*tnc + (c - tc0)] = element e of column c's slice in partition p
template <typename T>
__global__ void transpose_slices_kernel(const T* __restrict__ hashes, const uint64_t* __restrict__ start,
                                        const uint32_t* __restrict__ so, const uint64_t* __restrict__ tbase, int P,
                                        uint32_t n, uint32_t tc0, uint32_t tnc, T* __restrict__ tcols) {
  const uint32_t ci = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t p = blockIdx.y;
  if (ci >= tnc) return;
  const uint32_t c = tc0 + ci;
  const uint32_t lo = so[(size_t)p * n + c], hi = so[(size_t)(p + 1) * n + c];
  const T* src = hashes + start[c] + lo;
  T* dst = tcols + tbase[p] + ci;
  for (uint32_t e = 0; e < hi - lo; e++) dst[(size_t)e * tnc] = src[e];
}

// planning inputs in one launch: sum / max of the sketch lengths and SAMPLE_PER evenly spaced hashes from each of
// up to SAMPLE_SK evenly spaced sketches (quantiles -> partition bounds).  Block i < ns samples sketch i; the sample
// buffer is strided (SAMPLE_PER per sketch, nsamples[i] of them valid) and compacted on the host.
constexpr int SAMPLE_SK = 64;
constexpr int SAMPLE_PER = 64;
struct PlanStats { unsigned long long total; uint32_t lmax; uint32_t pad; };

template <typename T>
__global__ __launch_bounds__(256) void plan_stats_kernel(const T* __restrict__ hashes, const uint64_t* __restrict__ start,
                                                         const uint32_t* __restrict__ len, uint32_t n,
                                                         PlanStats* __restrict__ stats, T* __restrict__ samples,
                                                         uint32_t* __restrict__ nsamples) {
  unsigned long long sum = 0; uint32_t mx = 0;
  for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < n; g += gridDim.x * blockDim.x) { sum += len[g]; mx = max(mx, len[g]); }
  for (int o = 32; o > 0; o >>= 1) {
    sum += __shfl_xor((unsigned long long)sum, o);
    mx = max(mx, (uint32_t)__shfl_xor((int)mx, o));
  }
  if ((threadIdx.x & 63) == 0) { if (sum) atomicAdd(&stats->total, sum); if (mx) atomicMax(&stats->lmax, mx); }
  const uint32_t ns = min(n, (uint32_t)SAMPLE_SK);
  if (blockIdx.x < ns) {
    const uint32_t g = (uint32_t)((uint64_t)blockIdx.x * n / ns);
    const uint32_t L = len[g];
    const uint32_t take = min(L, (uint32_t)SAMPLE_PER);
    for (uint32_t t = threadIdx.x; t < take; t += blockDim.x)
      samples[blockIdx.x * SAMPLE_PER + t] = hashes[start[g] + (uint64_t)t * L / take];
    if (threadIdx.x == 0) nsamples[blockIdx.x] = take;
  }
}

// Everything a launch needs besides the tile itself: partition bounds -> slice offsets of every
// sketch, and the transposed copy of the column range [tc0, tc0 + tnc).  Lives in the context's
// scratch slots 1 (offsets) and 4 (transposed copy) until the next plan is built.
struct PairPlan {
  int P = 0, npl = 0;
  uint32_t n = 0, tc0 = 0, tnc = 0;
  const uint32_t* d_so = nullptr;
  const uint64_t* d_tbase = nullptr;
  const void* d_tcols = nullptr;
};

template <typename T, int NPL, int EMIT>
int launch_tiled(rtc_ctx* ctx, const T* d_hashes, const uint64_t* d_start, const PairPlan& pl, uint32_t row0,
                 uint32_t row1, uint32_t col0, uint32_t col1, uint32_t* d_common, uint64_t ld, int lower_only,
                 const EdgeSink& sink) {
  const size_t lds = (size_t)SLOTS * (sizeof(T) + 8 + 2) + sizeof(TileShared);
  auto kern = pair_tiled_kernel<T, NPL, EMIT>;
  RTC_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  dim3 grid((row1 - row0 + ROWS - 1) / ROWS, (col1 - col0 + TW - 1) / TW);
  hipLaunchKernelGGL(kern, grid, dim3(TW), lds, ctx->stream, d_hashes, d_start, (const T*)pl.d_tcols, pl.d_tbase, pl.d_so,
                     pl.P, pl.n, pl.tc0, pl.tnc, row0, row1, col0, col1, d_common, ld, lower_only, sink);
  RTC_CHECK_LAUNCH(ctx);
  return RTC_OK;
}

template <typename T, int EMIT>
int run_plan(rtc_ctx* ctx, const T* d_hashes, const uint64_t* d_start, const PairPlan& pl, uint32_t row0, uint32_t row1,
             uint32_t col0, uint32_t col1, uint32_t* d_common, uint64_t ld, int lower_only, const EdgeSink& sink) {
#define LT(NPLV) launch_tiled<T, NPLV, EMIT>(ctx, d_hashes, d_start, pl, row0, row1, col0, col1, d_common, ld, lower_only, sink)
  if (pl.npl <= 10) return LT(10);
  if (pl.npl <= 13) return LT(13);
  if (pl.npl <= 16) return LT(16);
  return LT(20);
#undef LT
}

// Builds the plan for columns [tc0, tc1).  *ok = 0 (and RTC_OK) when the tiled scheme cannot or should
// not take the input: counters too wide, a slice that fits no table, or a transposed copy beyond the
// memory budget (one huge sketch among many small ones inflates every partition's maximum) -- the
// caller then runs the per-pair merge kernel, which takes anything.
template <typename T>
int build_plan(rtc_ctx* ctx, const T* d_hashes, const uint64_t* d_start, const uint32_t* d_len, uint32_t n,
               uint32_t tc0, uint32_t tc1, PairPlan* pl, int* ok) {
  *ok = 0;
  if (n == 0 || tc1 <= tc0) return RTC_OK;
  // ---- planning inputs: one kernel, one small read-back ----
  void* wsp = nullptr;
  const size_t bsamp = (size_t)SAMPLE_SK * SAMPLE_PER * sizeof(T);
  const size_t bhead = sizeof(PlanStats) + (size_t)SAMPLE_SK * 4;  // stats + per-sketch sample counts (a multiple of 8)
  RTC_TRY(rtc_ws(ctx, 0, bhead + bsamp + 64, &wsp));
  PlanStats* d_stats = (PlanStats*)wsp;
  uint32_t* d_ns = (uint32_t*)((char*)wsp + sizeof(PlanStats));
  T* d_samples = (T*)((char*)wsp + bhead);
  RTC_HIP(ctx, hipMemsetAsync(wsp, 0, bhead, ctx->stream));
  hipLaunchKernelGGL(plan_stats_kernel<T>, dim3(std::max<uint32_t>(std::min<uint32_t>((n + 255) / 256, 256), std::min<uint32_t>(n, SAMPLE_SK))), dim3(256), 0,
                     ctx->stream, d_hashes, d_start, d_len, n, d_stats, d_samples, d_ns);
  RTC_CHECK_LAUNCH(ctx);
  void* hpin = nullptr;
  RTC_TRY(rtc_pinned(ctx, bhead + bsamp + 64 + (size_t)(MAXP + 1) * 8, &hpin));
  RTC_HIP(ctx, hipMemcpyAsync(hpin, wsp, bhead + bsamp, hipMemcpyDeviceToHost, ctx->stream));
  RTC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const PlanStats hst = *(const PlanStats*)hpin;
  std::vector<T> sample;
  {
    const uint32_t* h_ns = (const uint32_t*)((const char*)hpin + sizeof(PlanStats));
    const T* h_s = (const T*)((const char*)hpin + bhead);
    for (int i = 0; i < SAMPLE_SK; i++) sample.insert(sample.end(), h_s + (size_t)i * SAMPLE_PER, h_s + (size_t)i * SAMPLE_PER + std::min<uint32_t>(h_ns[i], SAMPLE_PER));
  }
  const uint32_t nsamp = (uint32_t)sample.size();
  const uint64_t tot = hst.total;
  const uint32_t lmax = hst.lmax;
  if (tot == 0 || lmax >= (1u << 20) || nsamp == 0) return RTC_OK;  // nothing to gain / counters too wide: merge path
  const double avg = (double)tot / n;
  uint32_t ktarget = KTARGET;
  if (const char* e = getenv("RTC_PAIR_KTARGET")) { const int v = atoi(e); if (v >= 256 && v <= (int)KCAP_HARD) ktarget = (uint32_t)v; }  // tuning experiments
  int P = 1;
  while (P < MAXP && (double)ROWS * avg / P > ktarget) P <<= 1;
  std::sort(sample.begin(), sample.end());
  T* h_bounds_pin = (T*)((char*)hpin + bhead + bsamp + 64 - ((bhead + bsamp + 64) % 8));
  const uint32_t tnc = tc1 - tc0;

  // Memory budget of the transposed copy: sum_p(pmax[p]) * tnc elements.  With even lengths it is
  // ~1.8x the hashes themselves; it is allowed 16x (or 1 GiB) and never more than half of the free HBM.
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = (size_t)8 << 30;
  uint64_t budget = std::max<uint64_t>((uint64_t)1 << 30, 16ull * tot * sizeof(T));
  budget = std::min<uint64_t>(budget, (uint64_t)(free_b + ctx->ws_bytes[4]) / 2);
  if (const char* e = getenv("RTC_PAIR_TCOLS_BUDGET")) budget = strtoull(e, nullptr, 10);  // tests of the fallback

  for (int attempt = 0; attempt < 3; attempt++) {
    std::vector<T> bounds(P + 1);
    bounds[0] = 0;
    for (int p = 1; p < P; p++) bounds[p] = sample[(size_t)((uint64_t)p * sample.size() / P)];
    bounds[P] = (T)~(T)0;
    void* ws = nullptr;
    const size_t bso = (size_t)(P + 1) * n * 4;
    const size_t bb = (size_t)(P + 1) * sizeof(T);
    RTC_TRY(rtc_ws(ctx, 1, bso + bb + 64 + (size_t)(P + 1) * 4, &ws));
    uint32_t* d_so = (uint32_t*)ws;
    T* d_bounds = (T*)((char*)ws + bso);
    uint32_t* d_max = (uint32_t*)((char*)ws + bso + bb + (8 - bb % 8) % 8);  // [0] = max slice of any sketch, [1..P] = per-partition max over the columns
    uint32_t* d_pmax = d_max + 1;
    memcpy(h_bounds_pin, bounds.data(), bb);
    RTC_HIP(ctx, hipMemcpyAsync(d_bounds, h_bounds_pin, bb, hipMemcpyHostToDevice, ctx->stream));
    RTC_HIP(ctx, hipMemsetAsync(d_max, 0, (size_t)(P + 1) * 4, ctx->stream));
    hipLaunchKernelGGL(slice_offsets_kernel<T>, dim3((n + 255) / 256, (uint32_t)P + 1), dim3(256), 0, ctx->stream, d_hashes, d_start, d_len,
                       d_bounds, P, n, d_so);
    RTC_CHECK_LAUNCH(ctx);
    hipLaunchKernelGGL(slice_max_kernel, dim3((n + 255) / 256, (uint32_t)P), dim3(256), 0, ctx->stream, d_so, n, tc0, tc1, d_max, d_pmax);
    RTC_CHECK_LAUNCH(ctx);
    std::vector<uint32_t> h_maxes(P + 1);
    RTC_HIP(ctx, hipMemcpyAsync(h_maxes.data(), d_max, (size_t)(P + 1) * 4, hipMemcpyDeviceToHost, ctx->stream));
    RTC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const uint32_t* h_pmax = h_maxes.data() + 1;
    if (h_maxes[0] > KCAP_HARD) {  // some single slice does not fit a table: refine the partition
      if (P >= MAXP) return RTC_OK;
      P = std::min(MAXP, P * 4);
      continue;
    }
    // ---- partition-major transposed copy of the column slices ----
    std::vector<uint64_t> tbase(P + 1, 0);
    for (int p = 0; p < P; p++) tbase[p + 1] = tbase[p] + (uint64_t)h_pmax[p] * tnc;
    if (tbase[P] * sizeof(T) > budget) return RTC_OK;  // merge path (skewed lengths)
    void* ws4 = nullptr;
    const size_t btb = (size_t)P * 8;
    {
      const int st = rtc_ws(ctx, 4, (tbase[P] + 4ull * DEPTH * tnc) * sizeof(T) + btb + 256, &ws4);  // + 4 * DEPTH rows: unconditional first probe loads
      if (st == RTC_ERR_NOMEM) return RTC_OK;  // the merge kernel needs no scratch
      if (st != RTC_OK) return st;
    }
    uint64_t* d_tbase = (uint64_t*)ws4;
    T* d_tcols = (T*)((char*)ws4 + ((btb + 255) / 256) * 256);
    memcpy(h_bounds_pin, tbase.data(), btb);  // the bounds upload completed before the read-back above
    RTC_HIP(ctx, hipMemcpyAsync(d_tbase, h_bounds_pin, btb, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(transpose_slices_kernel<T>, dim3((tnc + 255) / 256, (uint32_t)P), dim3(256), 0, ctx->stream, d_hashes,
                       d_start, d_so, d_tbase, P, n, tc0, tnc, d_tcols);
    RTC_CHECK_LAUNCH(ctx);
    int npl = 1;
    while ((1u << npl) <= lmax) npl++;
    pl->P = P; pl->npl = npl; pl->n = n; pl->tc0 = tc0; pl->tnc = tnc;
    pl->d_so = d_so; pl->d_tbase = d_tbase; pl->d_tcols = d_tcols;
    *ok = 1;
    return RTC_OK;
  }
  return RTC_OK;
}

template <typename T>
int tiled_impl(rtc_ctx* ctx, const T* d_hashes, const uint64_t* d_start, const uint32_t* d_len, uint32_t n,
               uint32_t row0, uint32_t row1, uint32_t col0, uint32_t col1, uint32_t* d_common, uint64_t ld,
               int lower_only, const EdgeSink* sink, int* handled) {
  *handled = 0;
  PairPlan pl;
  int ok = 0;
  auto& pc = ctx->pair_plan;
  if (ctx->pair_plan_hold && ctx->pair_plan_valid && pc.hashes == (const void*)d_hashes && pc.start == (const void*)d_start &&
      pc.len == (const void*)d_len && pc.n == n && pc.width == (int)sizeof(T) && pc.tc0 <= col0 && col1 <= pc.tc1) {
    pl.P = pc.P; pl.npl = pc.npl; pl.n = n; pl.tc0 = pc.tc0; pl.tnc = pc.tc1 - pc.tc0;
    pl.d_so = pc.d_so; pl.d_tbase = pc.d_tbase; pl.d_tcols = pc.d_tcols;
    ok = 1;
  } else {
    const uint32_t tc1 = ctx->pair_plan_hold ? std::min(n, std::max(col1, ctx->pair_plan_tc1_hint)) : col1;
    RTC_TRY(build_plan<T>(ctx, d_hashes, d_start, d_len, n, col0, tc1, &pl, &ok));
    ctx->pair_plan_valid = 0;
    if (ok && ctx->pair_plan_hold) {
      pc.hashes = d_hashes; pc.start = d_start; pc.len = d_len; pc.n = n; pc.tc0 = col0; pc.tc1 = tc1;
      pc.width = (int)sizeof(T); pc.P = pl.P; pc.npl = pl.npl; pc.d_so = pl.d_so; pc.d_tbase = pl.d_tbase; pc.d_tcols = pl.d_tcols;
      ctx->pair_plan_valid = 1;
    }
  }
  if (!ok) return RTC_OK;
  if (sink) RTC_TRY((run_plan<T, EMIT_EDGES>(ctx, d_hashes, d_start, pl, row0, row1, col0, col1, nullptr, 0, lower_only, *sink)));
  else RTC_TRY((run_plan<T, EMIT_DENSE>(ctx, d_hashes, d_start, pl, row0, row1, col0, col1, d_common, ld, lower_only, EdgeSink{})));
  *handled = 1;
  return RTC_OK;
}

}  // namespace

int rtc_pair_common_tiled(rtc_ctx* ctx, const void* d_hashes, int width, const uint64_t* d_start,
                          const uint32_t* d_len, uint32_t n, uint32_t row0, uint32_t row1, uint32_t col0,
                          uint32_t col1, uint32_t* d_common, uint64_t ld, int lower_only, int* handled) {
  *handled = 0;
  if (n == 0) return RTC_OK;
  if (width == 8)
    return tiled_impl<uint64_t>(ctx, (const uint64_t*)d_hashes, d_start, d_len, n, row0, row1, col0, col1, d_common,
                                ld, lower_only, nullptr, handled);
  return tiled_impl<uint32_t>(ctx, (const uint32_t*)d_hashes, d_start, d_len, n, row0, row1, col0, col1, d_common, ld,
                              lower_only, nullptr, handled);
}

// Candidate edges of the tile straight from the tiled kernel (no dense matrix).  radio < 0: no size test.
int rtc_pair_edges_tiled(rtc_ctx* ctx, const void* d_hashes, int width, const uint64_t* d_start,
                         const uint32_t* d_len, uint32_t n, uint32_t row0, uint32_t row1, uint32_t col0,
                         uint32_t col1, int lower_only, int radio, rtc_cedge* d_edges, uint64_t cap,
                         uint64_t* d_count, int* handled) {
  *handled = 0;
  if (n == 0) return RTC_OK;
  const EdgeSink sink{d_len, d_edges, (unsigned long long)cap, (unsigned long long*)d_count, radio};
  if (width == 8)
    return tiled_impl<uint64_t>(ctx, (const uint64_t*)d_hashes, d_start, d_len, n, row0, row1, col0, col1, nullptr, 0,
                                lower_only, &sink, handled);
  return tiled_impl<uint32_t>(ctx, (const uint32_t*)d_hashes, d_start, d_len, n, row0, row1, col0, col1, nullptr, 0,
                              lower_only, &sink, handled);
}

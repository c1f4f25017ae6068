// rtc_pairs_tiled.hip -- all-pairs |A_i n A_j| with an LDS-resident row-block inverted table.
//
// Dense (every pair is evaluated, no data-dependent skipping of pairs) but work-efficient form of
// the reference's inverted-index intersection (src/MST.cpp:1408-1435): instead of one merge per
// pair, a workgroup owns a block of 64 rows x 1024 columns and, for each hash-range partition p,
//   1. builds in LDS a table  hash -> 64-bit mask of the rows containing it  from the rows' sorted slices that fall
//      in partition p: 4-byte slots (fingerprint | entry index | overflow flag | generation) in buckets of four,
//      entries {key image, row mask} in an array of their own (ds_cmpst_b32 claims a slot, ds_or_b64 adds a row);
//      a slot of another generation is free, so a new table costs no clearing;
//   2. every lane (= one column sketch) streams its own slice of partition p as 32-bit DIGESTS, four per 16-byte load,
//      and reads each key's home bucket with one ds_read_b128; one wave vote per trip of four keys decides whether any
//      lane may have a hit, and only then are entries read and key images compared (a hit returns the mask of ALL 64
//      rows containing that hash);
//   3. masks are accumulated in per-lane bit-sliced counters (plane k holds bit k of the 64 row counters), i.e. one
//      probe serves 64 pairs; the (up to) four masks of a trip go through a 4:2 carry-save compressor, then a
//      wave-uniform ripple of AND/XOR on 64-bit registers.
// The column slices are first copied into a partition-major, group-major, column-minor layout of digests
// (tcols[base[p] + t*n + c] = elements 4t..4t+3 of column c's slice; 64-bit keys: a second copy with the low halves of
// the key images, read only to confirm a hit) so that the probe loop's loads are coalesced across lanes (= columns).
// After the last partition each lane unpacks its 64 counters and writes them (coalesced across
// lanes) to common[row][col].  Partition boundaries are data quantiles computed from a sample, so
// the table holds about half a key per bucket; blocks whose slices would overflow the table are split into row
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

constexpr int TW = 1024;                // lanes per workgroup = columns per block
constexpr int ROWS = 64;                // rows per block = mask width
constexpr int LOG2NB = 12;
constexpr int NB = 1 << LOG2NB;         // buckets of four 4-byte slots (64 KiB)
constexpr int BUCKET = 4;
constexpr uint32_t ECAP = 5120;         // entries {key, row mask} of 16 bytes (80 KiB): one per (row, element) of a table build
constexpr uint32_t KCAP_HARD = ECAP;    // max keys per table build
constexpr uint32_t KTARGET = 4096;      // planned mean keys per table: a key per bucket (2 048 until round 6: half the partitions -- half the
                                        // table builds, barriers and slice tails -- now pay more than the fuller buckets cost: kernel 1.53 -> 1.39 ms at
                                        // 10 000 u64 sketches, 2.88 -> 2.60 ms at 25 000 u32, sparse shapes -2 %, same box, RTC_PAIR_KTARGET A/B)
constexpr int RPW = ROWS / (TW / 64);   // rows a wave builds at once
constexpr int LPR = 64 / RPW;           // lanes per row in the build
constexpr int MAXP = 512;
constexpr int DEPTH = 3;                // trips (four digests = one 16-byte load per lane) of a column's slice requested ahead of the probes

// A slot names an entry: fingerprint [0,14) | entry index [14,27) | "this bucket overflowed" (slot 0 only) bit 27 |
// generation [28,32).  A slot whose generation is not the current table's is free: a new table costs no clearing
// (the 64 KiB are wiped once per 15 builds, when the 4-bit generation wraps).
constexpr uint32_t FP_MASK = 0x3fffu, IDX_SHIFT = 14, IDX_MASK = 0x1fffu, FLAG_BIT = 1u << 27, GEN_SHIFT = 28;
constexpr uint32_t MATCH_MASK = (0xfu << GEN_SHIFT) | FP_MASK;  // (slot ^ (gen | fp)) & MATCH_MASK == 0: live, same fingerprint
constexpr uint32_t FLAG_TEST = (0xfu << GEN_SHIFT) | FLAG_BIT;  // (slot0 ^ (gen | FLAG_BIT)) & FLAG_TEST == 0: live and overflowed

// A key travels as its image under a bijection (multiplication by an odd constant): equal images, equal keys.  The
// 32-bit digest that picks bucket and fingerprint is the image of a 32-bit key, the HIGH half of the image of a
// 64-bit key; the low half of the latter is only looked at to confirm a hit.
template <typename T> struct KeyTraits;
template <> struct KeyTraits<uint64_t> {
  __device__ static __forceinline__ uint64_t image(uint64_t k) { return k * 0x9E3779B97F4A7C15ULL; }
  __device__ static __forceinline__ uint32_t digest(uint64_t m) { return (uint32_t)(m >> 32); }
};
template <> struct KeyTraits<uint32_t> {
  __device__ static __forceinline__ uint32_t image(uint32_t k) { return k * 0x9E3779B1u; }
  __device__ static __forceinline__ uint32_t digest(uint32_t m) { return m; }
};
__device__ __forceinline__ uint32_t mix_bucket(uint32_t h) { return h >> (32 - LOG2NB); }
__device__ __forceinline__ uint32_t mix_fp(uint32_t h) { return (h >> 4) & FP_MASK; }

// What a probe carries is the 32-bit digest (the transposed column copy holds digests, 4 bytes per hash whatever the
// key width: the probe loop is bound by the bytes it streams).  An entry holds the key's whole image; a lane whose
// 64-bit key finds its fingerprint reads the low half of its image from a second transposed copy to compare.
template <typename T> struct alignas(16) Entry { T key; unsigned long long mask; };

struct RowMeta {                 // one partition's view of the row block (double-buffered: written one partition ahead)
  uint32_t rlo[ROWS], rhi[ROWS];
  uint32_t ebase[ROWS];          // first entry of the row's slice inside its sub-block's table
  uint32_t nsub;
  uint32_t sub_end[ROWS + 1];
};
struct TileShared {
  uint64_t rstart[ROWS];
  RowMeta meta[2];
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

// Row r's element with entry index `idx` goes into the table.  The entry is written first, then a slot is claimed
// with one compare-and-swap; a key another row has brought already only gets this row's bit.
template <typename T>
__device__ __forceinline__ void table_insert(uint32_t* slots, Entry<T>* ent, uint32_t gen28, T key, int r, uint32_t idx) {
  const unsigned long long bit = 1ULL << r;
  const T ekey = KeyTraits<T>::image(key);
  const uint32_t h = KeyTraits<T>::digest(ekey);
  uint32_t b = mix_bucket(h);
  const uint32_t live = gen28 | mix_fp(h);
  const uint32_t want = live | (idx << IDX_SHIFT);
  ent[idx].key = ekey;
  ent[idx].mask = bit;
  __atomic_signal_fence(__ATOMIC_SEQ_CST);  // the entry before the slot that names it (a wave's LDS operations execute in order)
  while (true) {
    const uint4 sv = *reinterpret_cast<const uint4*>(&slots[b * BUCKET]);
    const uint32_t s4[4] = {sv.x, sv.y, sv.z, sv.w};
#pragma unroll
    for (int j = 0; j < BUCKET; j++) {
      uint32_t s = s4[j];
      if ((s >> GEN_SHIFT) != (gen28 >> GEN_SHIFT)) {  // free as far as this lane has seen
        const uint32_t old = atomicCAS(&slots[b * BUCKET + j], s, want);
        if (old == s) return;
        s = old;                                       // taken in the meantime (by a live entry): look at it
      }
      if (((s ^ live) & MATCH_MASK) == 0) {
        const uint32_t e2 = (s >> IDX_SHIFT) & IDX_MASK;
        if (ent[e2].key == ekey) { atomicOr(&ent[e2].mask, bit); return; }
      }
    }
    atomicOr(&slots[b * BUCKET], FLAG_BIT);  // full without this key: probes and inserts go on to the next bucket
    b = (b + 1) & (NB - 1);
  }
}

// Row mask of the key with image `key` and digest h, 0 if no row of the table has it.
// The general walk: every slot of the home bucket, then the buckets an overflow has led to.
template <typename T>
__device__ __forceinline__ unsigned long long table_lookup(const uint32_t* slots, const Entry<T>* ent, uint32_t gen28, T key,
                                                           uint32_t h, uint4 sv) {
  uint32_t b = mix_bucket(h);
  const uint32_t live = gen28 | mix_fp(h);
  while (true) {
    const uint32_t s4[4] = {sv.x, sv.y, sv.z, sv.w};
#pragma unroll
    for (int j = 0; j < BUCKET; j++) {
      if (((s4[j] ^ live) & MATCH_MASK) == 0) {
        const Entry<T> e = ent[(s4[j] >> IDX_SHIFT) & IDX_MASK];
        if (e.key == key) return e.mask;
      }
    }
    if (((sv.x ^ (gen28 | FLAG_BIT)) & FLAG_TEST) != 0) return 0ULL;
    b = (b + 1) & (NB - 1);
    sv = *reinterpret_cast<const uint4*>(&slots[b * BUCKET]);
  }
}

// Row block meta data of one partition from the rows' slice bounds (lanes 0..63 of one wave hold one row each):
// entry bases by a wave prefix sum; the block is split into sub-blocks only when its keys do not fit one table.
__device__ __forceinline__ void write_meta(RowMeta* m, uint32_t lane, uint32_t nrows, uint32_t rlo, uint32_t rhi) {
  const uint32_t sz = lane < nrows ? rhi - rlo : 0;
  uint32_t incl = sz;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t v = (uint32_t)__shfl_up((int)incl, o);
    if (lane >= (uint32_t)o) incl += v;
  }
  const uint32_t tot = (uint32_t)__shfl((int)incl, 63);
  m->rlo[lane] = rlo;
  m->rhi[lane] = rhi;
  if (tot <= KCAP_HARD) {
    m->ebase[lane] = incl - sz;
    if (lane == 0) { m->sub_end[0] = nrows; m->nsub = 1; }
  } else {
    // rare: serial split by lane 0 (the sizes come back through the LDS)
    m->ebase[lane] = sz;
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
      uint32_t ns = 0, acc = 0;
      for (uint32_t r = 0; r < nrows; r++) {
        const uint32_t szr = m->ebase[r];
        if (acc + szr > KCAP_HARD && acc > 0) { m->sub_end[ns++] = r; acc = 0; }
        m->ebase[r] = acc;
        acc += szr;
      }
      m->sub_end[ns++] = nrows;
      m->nsub = ns;
    }
  }
}

// so: [(P+1)][n] slice offsets (so[p][g] = lower_bound(sketch g, bound[p])), so[0]=0, so[P]=len.
// tcols covers the column range [tc0, tc0 + tnc): the digests of elements 4t .. 4t+3 of column c's slice in
// partition p are the 16 bytes tcols[tbase[p] + t*tnc + (c - tc0)] (zeros past the slice's end).
template <typename T, int NPL, int EMIT>
__global__ __launch_bounds__(TW, 1) void pair_tiled_kernel(const T* __restrict__ hashes,
                                                        const uint64_t* __restrict__ start,
                                                        const uint4* __restrict__ tcols,      // transposed column slices (digests)
                                                        const uint4* __restrict__ tcols_lo,   // 64-bit keys: low halves of the images, same layout
                                                        const uint64_t* __restrict__ tbase,   // [P] offsets in 16-byte groups
                                                        const uint32_t* __restrict__ so, int P, uint32_t n,
                                                        uint32_t tc0, uint32_t tnc,
                                                        uint32_t row0, uint32_t row1, uint32_t col0,
                                                        uint32_t col1, uint32_t* __restrict__ out, uint64_t ld,
                                                        int lower_only, EdgeSink sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t* slots = reinterpret_cast<uint32_t*>(smem);
  Entry<T>* ent = reinterpret_cast<Entry<T>*>(smem + (size_t)NB * BUCKET * 4);
  TileShared* sh = reinterpret_cast<TileShared*>(smem + (size_t)NB * BUCKET * 4 + (size_t)ECAP * sizeof(Entry<T>));

  const int tid = threadIdx.x;
  // Row blocks vary fastest in the grid: the workgroups running at the same time (round-robin over
  // the XCDs) then probe the same 1024-column slab, which each XCD's L2 fetches once instead of
  // once per row block.
  const uint32_t rb0 = row0 + blockIdx.x * ROWS;
  const uint32_t nrows = min((uint32_t)ROWS, row1 - rb0);
  // the column blocks run from the last to the first: the blocks near the diagonal -- whose columns are related to their rows
  // (families sit side by side) and resolve hits on most trips -- start first, and the launch's last, partly idle round of
  // workgroups is made of the cheap ones far below the diagonal
  const uint32_t cb0 = col0 + (gridDim.y - 1 - blockIdx.y) * TW;
  if (lower_only && cb0 + 1 > rb0 + nrows - 1) return;  // no (row, col) with col < row in this block
  const uint32_t c = cb0 + tid;
  // (lower_only: a column at or beyond the block's last row pairs with none of its rows -- in a block on the diagonal
  // that is half of the lanes, whole waves of them: they take no part in the probes)
  const bool col_active = c < col1 && (!lower_only || c + 1 < rb0 + nrows);
  const uint32_t wave = tid >> 6, lane = tid & 63;

  unsigned long long planes[NPL];
#pragma unroll
  for (int k = 0; k < NPL; k++) planes[k] = 0ULL;

  uint32_t clo = col_active ? so[c] : 0;  // so[0][c]
  if (tid < ROWS) sh->rstart[tid] = tid < (int)nrows ? start[rb0 + tid] : 0;
  // Every global load a partition needs is requested one partition (or one phase) ahead: the column's next
  // slice end, the rows' slice bounds (wave 0), the first probe keys (before the table is built).
  const bool rowlane = tid < (int)nrows;
  uint32_t chi_n = col_active ? so[(size_t)n + c] : 0;
  uint32_t rlo_n = 0, rhi_n = 0;
  if (wave == 0) {
    rlo_n = rowlane ? so[rb0 + tid] : 0;
    rhi_n = rowlane ? so[(size_t)n + rb0 + tid] : 0;
    write_meta(&sh->meta[0], lane, nrows, rlo_n, rhi_n);
    rlo_n = rhi_n;
    if (P > 1) rhi_n = rowlane ? so[(size_t)2 * n + rb0 + tid] : 0;
  }
  uint32_t gen = 15;  // the first build wipes the slots

  for (int p = 0; p < P; p++) {
    const uint32_t chi = chi_n;
    const uint4* tp = tcols + tbase[p] + (c - tc0);
    uint4 nq[DEPTH];  // groups past the slice's end hold other data (the copy is padded by DEPTH groups): masked by `rem`
#pragma unroll
    for (int d = 0; d < DEPTH; d++) nq[d] = col_active ? tp[(size_t)d * tnc] : make_uint4(0u, 0u, 0u, 0u);
    if (p + 1 < P) chi_n = col_active ? so[(size_t)(p + 2) * n + c] : 0;
    __syncthreads();  // previous partition's probes are done (table reusable); this partition's meta data is visible
    const RowMeta* mt = &sh->meta[p & 1];
    const uint32_t nsub = mt->nsub;
    uint32_t ra = 0;
    for (uint32_t sb = 0; sb < nsub; sb++) {
      const uint32_t rbnd = mt->sub_end[sb];
      if (++gen == 16) {  // generation wrap (and the very first build): wipe the slots
        uint4* sq = reinterpret_cast<uint4*>(slots);
        for (int i = tid; i < NB * BUCKET / 4; i += TW) sq[i] = make_uint4(0u, 0u, 0u, 0u);
        gen = 1;
        __syncthreads();
      }
      const uint32_t gen28 = gen << GEN_SHIFT;
      // ---- build: LPR lanes per row, all (<= 64) rows at once, four loads in flight per lane ----
      {
        const uint32_t r = ra + wave * RPW + lane / LPR;
        if (r < rbnd) {
          const T* rp = hashes + sh->rstart[r];
          const uint32_t lo = mt->rlo[r], hi = mt->rhi[r], eb = mt->ebase[r];
          for (uint32_t e = lo + (lane % LPR); e < hi; e += 4 * LPR) {
            T kq[4];
#pragma unroll
            for (int j = 0; j < 4; j++) kq[j] = (e + LPR * j < hi) ? rp[e + LPR * j] : (T)0;
#pragma unroll
            for (int j = 0; j < 4; j++)
              if (e + LPR * j < hi) table_insert<T>(slots, ent, gen28, kq[j], (int)r, eb + (e + LPR * j - lo));
          }
        }
      }
      __syncthreads();
      // next partition's row meta data, one partition ahead (wave 0, before its own probes)
      if (sb == 0 && wave == 0 && p + 1 < P) {
        write_meta(&sh->meta[(p + 1) & 1], lane, nrows, rlo_n, rhi_n);
        rlo_n = rhi_n;
        if (p + 2 < P) rhi_n = rowlane ? so[(size_t)(p + 3) * n + rb0 + tid] : 0;
      }
      // ---- probe: this lane's column slice, read coalesced from the transposed copy ----
      // Four digests (one 16-byte load) per trip, the next trips' loads in flight.  Straight line per key: the home
      // bucket's four slots in one ds_read_b128, "live and my fingerprint" for each slot and "live and overflowed"
      // for the bucket as values that are zero when true, their minimum; ONE wave vote per trip on the minimum of the
      // four keys.  Only a trip in which some lane may have a hit goes on to the entries -- all four keys side by
      // side: the entry of each key's matching slot (and, for 64-bit keys, the lane's own key from its sketch) in
      // one round trip; what that leaves open (a second slot with the fingerprint, an overflowed bucket) takes the
      // general walk.
      {
        const uint32_t mylen = chi - clo;
        const uint4* tp_lo = sizeof(T) == 8 ? tcols_lo + tbase[p] + (c - tc0) : nullptr;  // 64-bit keys only
        if (sb > 0) {
#pragma unroll
          for (int d = 0; d < DEPTH; d++) nq[d] = col_active ? tp[(size_t)d * tnc] : make_uint4(0u, 0u, 0u, 0u);
        }
        const uint32_t flagq = gen28 | FLAG_BIT;
        for (uint32_t e = 0; e < mylen; e += 4) {
          const uint4 bq = nq[0];
#pragma unroll
          for (int d = 0; d + 1 < DEPTH; d++) nq[d] = nq[d + 1];
          nq[DEPTH - 1] = (e + 4 * DEPTH < mylen) ? tp[(size_t)(e / 4 + DEPTH) * tnc] : make_uint4(0u, 0u, 0u, 0u);
          const uint32_t hq[4] = {bq.x, bq.y, bq.z, bq.w};
          uint32_t mn[4], cs[4];
          uint4 sv[4];
#pragma unroll
          for (int j = 0; j < 4; j++) sv[j] = *reinterpret_cast<const uint4*>(&slots[mix_bucket(hq[j]) * BUCKET]);
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const uint32_t live = gen28 | mix_fp(hq[j]);
            const uint32_t d0 = (sv[j].x ^ live) & MATCH_MASK, d1 = (sv[j].y ^ live) & MATCH_MASK;
            const uint32_t d2 = (sv[j].z ^ live) & MATCH_MASK, d3 = (sv[j].w ^ live) & MATCH_MASK;
            const uint32_t fl = (sv[j].x ^ flagq) & FLAG_TEST;
            cs[j] = min(min(d0, d1), min(d2, d3));
            mn[j] = min(cs[j], fl);
          }
          if (!__ballot(min(min(mn[0], mn[1]), min(mn[2], mn[3])) == 0u)) continue;
          const uint32_t rem = mylen - e;
          unsigned long long hm[4];
          T myk[4];
          Entry<T> en[4];
          bool open[4];
          uint4 lo4 = make_uint4(0u, 0u, 0u, 0u);
          if constexpr (sizeof(T) == 8) lo4 = tp_lo[(size_t)(e / 4) * tnc];
          const uint32_t lq[4] = {lo4.x, lo4.y, lo4.z, lo4.w};
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const uint32_t live = gen28 | mix_fp(hq[j]);
            const bool cand = cs[j] == 0u && rem > (uint32_t)j;
            uint32_t sl = sv[j].x;  // the first slot with the fingerprint
            if (((sv[j].x ^ live) & MATCH_MASK) != 0) sl = sv[j].y;
            if (((sv[j].x ^ live) & MATCH_MASK) != 0 && ((sv[j].y ^ live) & MATCH_MASK) != 0) sl = sv[j].z;
            if (((sv[j].x ^ live) & MATCH_MASK) != 0 && ((sv[j].y ^ live) & MATCH_MASK) != 0 && ((sv[j].z ^ live) & MATCH_MASK) != 0) sl = sv[j].w;
            en[j] = ent[cand ? ((sl >> IDX_SHIFT) & IDX_MASK) : 0u];
            if constexpr (sizeof(T) == 8) myk[j] = ((T)hq[j] << 32) | lq[j];
            else myk[j] = (T)hq[j];
            open[j] = mn[j] == 0u && rem > (uint32_t)j;
          }
          bool any_open = false;
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const bool hit = open[j] && cs[j] == 0u && en[j].key == myk[j];
            hm[j] = hit ? en[j].mask : 0ULL;
            open[j] = open[j] && !hit;
            any_open = any_open || open[j];
          }
          if (__ballot(any_open)) {  // rare: fingerprint without the key, or an overflowed home bucket
#pragma unroll
            for (int j = 0; j < 4; j++)
              if (open[j]) hm[j] = table_lookup<T>(slots, ent, gen28, myk[j], hq[j], sv[j]);
          }
          // the (up to) four row masks into the bit-sliced counters: a 4:2 compressor on planes 0 and 1, then a
          // ripple from plane 2 that stops when no lane of the wave carries any more
          {
            const unsigned long long s1 = planes[0] ^ hm[0] ^ hm[1];
            const unsigned long long c1 = (planes[0] & hm[0]) | (hm[1] & (planes[0] ^ hm[0]));
            planes[0] = s1 ^ hm[2] ^ hm[3];
            const unsigned long long c2 = (s1 & hm[2]) | (hm[3] & (s1 ^ hm[2]));
            unsigned long long carry = (planes[1] & c1) | (c2 & (planes[1] ^ c1));
            planes[1] = planes[1] ^ c1 ^ c2;
#pragma unroll
            for (int k = 2; k < NPL; k++) {
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
// sketch and partition (a single slice must fit one table build); one workgroup per partition, coalesced over g, two
// atomics per workgroup (one per wave on a single address used to be most of this kernel's time)
__global__ __launch_bounds__(1024) void slice_max_kernel(const uint32_t* __restrict__ so, uint32_t n, uint32_t c0, uint32_t c1,
                                                         uint32_t* __restrict__ amax, uint32_t* __restrict__ pmax) {
  __shared__ uint32_t sd[16], sdc[16];
  const uint32_t p = blockIdx.x;
  uint32_t d = 0, dc = 0;
  for (uint32_t g = threadIdx.x; g < n; g += blockDim.x) {
    const uint32_t v = so[(size_t)(p + 1) * n + g] - so[(size_t)p * n + g];
    d = max(d, v);
    if (g >= c0 && g < c1) dc = max(dc, v);
  }
  for (int o = 32; o > 0; o >>= 1) {
    d = max(d, (uint32_t)__shfl_xor((int)d, o));
    dc = max(dc, (uint32_t)__shfl_xor((int)dc, o));
  }
  if ((threadIdx.x & 63) == 0) { sd[threadIdx.x >> 6] = d; sdc[threadIdx.x >> 6] = dc; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (uint32_t w = 1; w < blockDim.x / 64; w++) { d = max(d, sd[w]); dc = max(dc, sdc[w]); }
    pmax[p] = dc;
    if (d) atomicMax(amax, d);
  }
}

// tcols[tbase[p] + t*tnc + (c - tc0)] = digests of elements 4t .. 4t+3 of column c's slice in partition p (zeros past its
// end); 64-bit keys: tcols_lo, same layout, the low halves of their images
template <typename T>
__global__ void transpose_slices_kernel(const T* __restrict__ hashes, const uint64_t* __restrict__ start,
                                        const uint32_t* __restrict__ so, const uint64_t* __restrict__ tbase, int P,
                                        uint32_t n, uint32_t tc0, uint32_t tnc, uint4* __restrict__ tcols,
                                        uint4* __restrict__ tcols_lo) {
  const uint32_t ci = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t p = blockIdx.y;
  if (ci >= tnc) return;
  const uint32_t c = tc0 + ci;
  const uint32_t lo = so[(size_t)p * n + c], hi = so[(size_t)(p + 1) * n + c];
  const T* src = hashes + start[c] + lo;
  const uint32_t len = hi - lo;
  for (uint32_t e = 0; e < len; e += 4) {
    T m[4];
#pragma unroll
    for (int j = 0; j < 4; j++) m[j] = e + j < len ? KeyTraits<T>::image(src[e + j]) : (T)0;
    const size_t o = tbase[p] + (size_t)(e / 4) * tnc + ci;
    tcols[o] = make_uint4(KeyTraits<T>::digest(m[0]), KeyTraits<T>::digest(m[1]), KeyTraits<T>::digest(m[2]), KeyTraits<T>::digest(m[3]));
    if constexpr (sizeof(T) == 8) tcols_lo[o] = make_uint4((uint32_t)m[0], (uint32_t)m[1], (uint32_t)m[2], (uint32_t)m[3]);
  }
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
  const void* d_tcols_lo = nullptr;  // 64-bit keys only
};

template <typename T, int NPL, int EMIT>
int launch_tiled(rtc_ctx* ctx, const T* d_hashes, const uint64_t* d_start, const PairPlan& pl, uint32_t row0,
                 uint32_t row1, uint32_t col0, uint32_t col1, uint32_t* d_common, uint64_t ld, int lower_only,
                 const EdgeSink& sink) {
  const size_t lds = (size_t)NB * BUCKET * 4 + (size_t)ECAP * sizeof(Entry<T>) + sizeof(TileShared);
  auto kern = pair_tiled_kernel<T, NPL, EMIT>;
  RTC_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  dim3 grid((row1 - row0 + ROWS - 1) / ROWS, (col1 - col0 + TW - 1) / TW);
  if (!ctx->pk0) { RTC_HIP(ctx, hipEventCreate(&ctx->pk0)); RTC_HIP(ctx, hipEventCreate(&ctx->pk1)); }
  RTC_HIP(ctx, hipEventRecord(ctx->pk0, ctx->stream));
  hipLaunchKernelGGL(kern, grid, dim3(TW), lds, ctx->stream, d_hashes, d_start, (const uint4*)pl.d_tcols, (const uint4*)pl.d_tcols_lo, pl.d_tbase, pl.d_so,
                     pl.P, pl.n, pl.tc0, pl.tnc, row0, row1, col0, col1, d_common, ld, lower_only, sink);
  RTC_CHECK_LAUNCH(ctx);
  RTC_HIP(ctx, hipEventRecord(ctx->pk1, ctx->stream));
  ctx->pk_valid = 1;
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
  if (ctx->opt.pair_ktarget >= 256 && ctx->opt.pair_ktarget <= KCAP_HARD) ktarget = ctx->opt.pair_ktarget;  // tuning experiments
  int P = (int)std::min<double>((double)MAXP, std::max(1.0, std::ceil((double)ROWS * avg / ktarget)));  // (any count: the bounds are sample quantiles)
  std::sort(sample.begin(), sample.end());
  T* h_bounds_pin = (T*)((char*)hpin + bhead + bsamp + 64 - ((bhead + bsamp + 64) % 8));
  const uint32_t tnc = tc1 - tc0;

  // Memory budget of the transposed copy: sum_p(pmax[p]) * tnc elements.  With even lengths it is
  // ~1.8x the hashes themselves; it is allowed 16x (or 1 GiB) and never more than half of the free HBM.
  const uint64_t free_b = rtc_free_hbm(ctx);
  uint64_t budget = std::max<uint64_t>((uint64_t)1 << 30, 16ull * tot * sizeof(T));
  budget = std::min<uint64_t>(budget, (uint64_t)(free_b + ctx->ws_bytes[4]) / 2);
  if (ctx->opt.pair_tcols_budget) budget = ctx->opt.pair_tcols_budget;  // tests of the fallback

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
    hipLaunchKernelGGL(slice_max_kernel, dim3((uint32_t)P), dim3(1024), 0, ctx->stream, d_so, n, tc0, tc1, d_max, d_pmax);
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
    for (int p = 0; p < P; p++) tbase[p + 1] = tbase[p] + (uint64_t)((h_pmax[p] + 3) / 4) * tnc;  // 16-byte groups of four digests
    const uint64_t copies = sizeof(T) == 8 ? 2 : 1;  // 64-bit keys: digests + low halves
    if (tbase[P] * 16 * copies > budget) return RTC_OK;  // merge path (skewed lengths)
    void* ws4 = nullptr;
    const size_t btb = (size_t)P * 8;
    {
      const int st = rtc_ws(ctx, 4, (tbase[P] + (uint64_t)DEPTH * tnc) * 16 * copies + btb + 512, &ws4);  // + DEPTH groups: unconditional first probe loads
      if (st == RTC_ERR_NOMEM) return RTC_OK;  // the merge kernel needs no scratch
      if (st != RTC_OK) return st;
    }
    uint64_t* d_tbase = (uint64_t*)ws4;
    uint4* d_tcols = (uint4*)((char*)ws4 + ((btb + 255) / 256) * 256);
    uint4* d_tcols_lo = d_tcols + (tbase[P] + (uint64_t)DEPTH * tnc);
    memcpy(h_bounds_pin, tbase.data(), btb);  // the bounds upload completed before the read-back above
    RTC_HIP(ctx, hipMemcpyAsync(d_tbase, h_bounds_pin, btb, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(transpose_slices_kernel<T>, dim3((tnc + 255) / 256, (uint32_t)P), dim3(256), 0, ctx->stream, d_hashes,
                       d_start, d_so, d_tbase, P, n, tc0, tnc, d_tcols, d_tcols_lo);
    RTC_CHECK_LAUNCH(ctx);
    int npl = 1;
    while ((1u << npl) <= lmax) npl++;
    pl->P = P; pl->npl = npl; pl->n = n; pl->tc0 = tc0; pl->tnc = tnc;
    pl->d_so = d_so; pl->d_tbase = d_tbase; pl->d_tcols = d_tcols; pl->d_tcols_lo = d_tcols_lo;
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
    pl.d_so = pc.d_so; pl.d_tbase = pc.d_tbase; pl.d_tcols = pc.d_tcols; pl.d_tcols_lo = pc.d_tcols_lo;
    ok = 1;
  } else {
    const uint32_t tc1 = ctx->pair_plan_hold ? std::min(n, std::max(col1, ctx->pair_plan_tc1_hint)) : col1;
    RTC_TRY(build_plan<T>(ctx, d_hashes, d_start, d_len, n, col0, tc1, &pl, &ok));
    ctx->pair_plan_valid = 0;
    if (ok && ctx->pair_plan_hold) {
      pc.hashes = d_hashes; pc.start = d_start; pc.len = d_len; pc.n = n; pc.tc0 = col0; pc.tc1 = tc1;
      pc.width = (int)sizeof(T); pc.P = pl.P; pc.npl = pl.npl; pc.d_so = pl.d_so; pc.d_tbase = pl.d_tbase; pc.d_tcols = pl.d_tcols; pc.d_tcols_lo = pl.d_tcols_lo;
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

// Duration of the last pair_tiled_kernel launch of this context (HIP events on its launch stream); waits for it.
extern "C" int rtc_pair_last_kernel_ms(rtc_ctx* ctx, float* ms_out) {
  if (!ctx || !ms_out) return RTC_ERR_ARG;
  *ms_out = 0.f;
  if (!ctx->pk_valid) return rtc_fail(ctx, RTC_ERR_ARG, "no tiled pair kernel has run on this context");
  RTC_HIP(ctx, hipSetDevice(ctx->device));
  RTC_HIP(ctx, hipEventSynchronize(ctx->pk1));
  RTC_HIP(ctx, hipEventElapsedTime(ms_out, ctx->pk0, ctx->pk1));
  return RTC_OK;
}

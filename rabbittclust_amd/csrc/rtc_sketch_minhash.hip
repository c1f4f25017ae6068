// rtc_sketch_minhash.hip -- bottom-s MinHash sketching on gfx950.
//
// Replaces Sketch::MinHash::{update,storeMinHashes} driven from src/SketchInfo.cpp:918-942,969
// (reference tree paths).  One workgroup (256 lanes = 4 wave64) walks one *segment* of a genome:
//   * 16-byte coalesced loads stage a 15 KiB tile of ASCII bases into LDS;
//   * every lane owns 60 consecutive k-mer end positions of the tile (lane stride 15 dwords ->
//     conflict-free ds_read_b32) and rolls the 2-bit forward / reverse-complement words;
//   * the canonical word is expanded to ASCII with v_perm_b32 and hashed with MurmurHash3_x64_128
//     (runtime k, seed) -- integer-ALU bound: ten 64x64 multiplies per k-mer;
//   * hashes below the running threshold T are appended to an LDS candidate buffer with a
//     wave ballot + one LDS atomic per wave; when the buffer fills, an in-LDS bitonic sort +
//     dedup keeps the s smallest distinct values and lowers T.
// Large genomes / small batches are split into several segments whose partial sketches are
// merged by merge_partials_kernel (bottom-s is a mergeable summary).
#include <algorithm>

#include "rtc_internal.h"

namespace {

constexpr int WG = 256;
constexpr int RUN_DW = 15;                            // dwords of owned bases per lane per tile
constexpr int WARM_DW = 8;                            // 32 warm-up bases (k-1 <= 31)
constexpr int TILE_BASES = WG * RUN_DW * 4;           // 15360
constexpr int TILE_DW = WG * RUN_DW + WARM_DW;        // 3848 dwords in LDS
constexpr int STEP_APPENDS = WG * 4;                  // worst-case appends per dword iteration
constexpr uint64_t SENT = ~0ULL;

struct Segment {
  uint64_t g_begin, g_end;  // genome byte range in d_seq
  uint64_t s_begin, s_end;  // k-mer END positions owned by this segment (absolute)
  uint64_t out_off;         // element offset into out buffer
  uint32_t cnt_slot;        // index into cnt buffer
  uint32_t sketch_size;
};

struct Ctrl {
  uint64_t T;
  uint32_t count;
  uint32_t overflow;
  uint32_t saw_max;
  uint32_t scan_base;
  uint32_t wave_tot[4];
};

// ---- MurmurHash3_x64_128, first output word, for a k-byte key held in w[0..7] (zero padded) ----
__device__ __forceinline__ uint64_t fmix64(uint64_t x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return x;
}

__device__ __forceinline__ uint64_t murmur3_h1(const uint32_t (&w)[8], int k, uint32_t seed) {
  const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
  uint64_t h1 = seed, h2 = seed;
  uint64_t t0, t1;  // tail words
  if (k >= 16) {    // wave-uniform
    uint64_t k1 = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
    uint64_t k2 = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
    k1 *= c1; k1 = rtc_rotl64(k1, 31); k1 *= c2; h1 ^= k1;
    h1 = rtc_rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
    k2 *= c2; k2 = rtc_rotl64(k2, 33); k2 *= c1; h2 ^= k2;
    h2 = rtc_rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
    t0 = (uint64_t)w[4] | ((uint64_t)w[5] << 32);
    t1 = (uint64_t)w[6] | ((uint64_t)w[7] << 32);
    if (k == 32) {
      uint64_t k1b = t0, k2b = t1;
      k1b *= c1; k1b = rtc_rotl64(k1b, 31); k1b *= c2; h1 ^= k1b;
      h1 = rtc_rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
      k2b *= c2; k2b = rtc_rotl64(k2b, 33); k2b *= c1; h2 ^= k2b;
      h2 = rtc_rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
      t0 = t1 = 0;
    }
  } else {
    t0 = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
    t1 = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
  }
  const int tail = k & 15;
  if (tail > 8) { uint64_t k2 = t1; k2 *= c2; k2 = rtc_rotl64(k2, 33); k2 *= c1; h2 ^= k2; }
  if (tail > 0) { uint64_t k1 = t0; k1 *= c1; k1 = rtc_rotl64(k1, 31); k1 *= c2; h1 ^= k1; }
  h1 ^= (uint64_t)k; h2 ^= (uint64_t)k;
  h1 += h2; h2 += h1;
  h1 = fmix64(h1); h2 = fmix64(h2);
  h1 += h2;
  return h1;
}

// expand the canonical 2-bit word (first base in the top bits after left alignment) into the
// ASCII bytes MurmurHash3 consumes: 4 codes -> 4 selector bytes -> v_perm_b32 over "ACGT".
__device__ __forceinline__ uint32_t codes_to_ascii(uint32_t e) {
  uint32_t t = ((e << 24) | (e << 14) | (e << 4) | (e >> 6)) & 0x03030303u;
  return __builtin_amdgcn_perm(0u, 0x54474341u, t);
}

struct KParams {
  int k;
  uint32_t seed;
  uint32_t use64;
  int lshift;          // 64 - 2k
  int rc_shift;        // 2k - 2
  uint64_t kmask;      // low 2k bits
  uint32_t bmask[8];   // byte masks of the 8 key dwords
};

__device__ __forceinline__ uint64_t kmer_hash(uint64_t canon, const KParams& P) {
  const uint64_t x = canon << P.lshift;
  const uint32_t hi = (uint32_t)(x >> 32), lo = (uint32_t)x;
  uint32_t w[8];
  w[0] = codes_to_ascii(hi >> 24) & P.bmask[0];
  w[1] = codes_to_ascii((hi >> 16) & 0xff) & P.bmask[1];
  w[2] = codes_to_ascii((hi >> 8) & 0xff) & P.bmask[2];
  w[3] = codes_to_ascii(hi & 0xff) & P.bmask[3];
  w[4] = codes_to_ascii(lo >> 24) & P.bmask[4];
  w[5] = codes_to_ascii((lo >> 16) & 0xff) & P.bmask[5];
  w[6] = codes_to_ascii((lo >> 8) & 0xff) & P.bmask[6];
  w[7] = codes_to_ascii(lo & 0xff) & P.bmask[7];
  uint64_t h = murmur3_h1(w, P.k, P.seed);
  return P.use64 ? h : (h & 0xffffffffULL);
}

__device__ __forceinline__ KParams make_kparams(int k, uint32_t seed) {
  KParams P;
  P.k = k; P.seed = seed; P.use64 = k > 16 ? 1u : 0u;
  P.lshift = 64 - 2 * k; P.rc_shift = 2 * k - 2;
  P.kmask = k == 32 ? ~0ULL : ((1ULL << (2 * k)) - 1);
#pragma unroll
  for (int d = 0; d < 8; d++) {
    int nb = k - 4 * d;
    P.bmask[d] = nb >= 4 ? 0xffffffffu : (nb <= 0 ? 0u : ((1u << (8 * nb)) - 1u));
  }
  return P;
}

// ---- block-wide merge: sort buf[0..CAP), drop duplicates, keep the `s` smallest ----------------
__device__ void bitonic_sort_lds(uint64_t* buf, int cap) {
  const int t = threadIdx.x;
  for (int k = 2; k <= cap; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = t; i < (cap >> 1); i += WG) {
        const int a = ((i & ~(j - 1)) << 1) | (i & (j - 1));
        const int b = a | j;
        const bool up = (a & k) == 0;
        const uint64_t va = buf[a], vb = buf[b];
        if ((va > vb) == up) { buf[a] = vb; buf[b] = va; }
      }
      __syncthreads();
    }
  }
}

// On entry: buf[0..ctrl->count) holds candidates (unsorted, duplicates allowed), all threads
// arrive.  On exit: buf[0..count) ascending distinct, count <= s, ctrl->T updated.
__device__ void merge_block(uint64_t* buf, Ctrl* ctrl, int cap, uint32_t s) {
  const int t = threadIdx.x;
  __syncthreads();
  const uint32_t n = ctrl->count < (uint32_t)cap ? ctrl->count : (uint32_t)cap;
  for (int i = n + t; i < cap; i += WG) buf[i] = SENT;
  if (t == 0) ctrl->scan_base = 0;
  __syncthreads();
  bitonic_sort_lds(buf, cap);
  // streaming compaction in rounds of WG elements (dest <= src always)
  const uint32_t lane = t & 63, wave = t >> 6;
  for (int r = 0; r < cap; r += WG) {
    const int idx = r + t;
    const uint64_t v = buf[idx];
    const bool keep = v != SENT && (idx == 0 || v != buf[idx - 1]);
    const uint64_t bal = __ballot(keep);
    const uint32_t before = __popcll(bal & ((1ULL << lane) - 1ULL));
    if (lane == 0) ctrl->wave_tot[wave] = (uint32_t)__popcll(bal);
    __syncthreads();  // all reads of this round done; wave totals visible
    const uint32_t sb = ctrl->scan_base;
    const uint32_t w0 = ctrl->wave_tot[0], w1 = ctrl->wave_tot[1], w2 = ctrl->wave_tot[2], w3 = ctrl->wave_tot[3];
    const uint32_t base = sb + (wave > 0 ? w0 : 0) + (wave > 1 ? w1 : 0) + (wave > 2 ? w2 : 0);
    const uint32_t dest = base + before;
    if (keep && dest < s) buf[dest] = v;
    __syncthreads();  // writes done; wave_tot / scan_base may be rewritten
    if (t == 0) ctrl->scan_base = sb + w0 + w1 + w2 + w3;  // read again only after the next barrier
  }
  __syncthreads();
  if (t == 0) {
    uint32_t c = ctrl->scan_base < s ? ctrl->scan_base : s;
    ctrl->count = c;
    ctrl->T = (c == s && s > 0) ? buf[s - 1] : SENT;
    ctrl->overflow = 0;
  }
  __syncthreads();
}

// ---- the sketch kernel ---------------------------------------------------------------------------
__global__ __launch_bounds__(WG) void sketch_minhash_kernel(const uint8_t* __restrict__ seq,
                                                            const Segment* __restrict__ segs,
                                                            int k, uint32_t seed, int cap,
                                                            uint64_t* __restrict__ out,
                                                            uint32_t* __restrict__ cnt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t* buf = reinterpret_cast<uint64_t*>(smem);
  uint32_t* tile = reinterpret_cast<uint32_t*>(smem + (size_t)cap * 8);
  Ctrl* ctrl = reinterpret_cast<Ctrl*>(smem + (size_t)cap * 8 + (size_t)TILE_DW * 4);

  const Segment sg = segs[blockIdx.x];
  const KParams P = make_kparams(k, seed);
  const int t = threadIdx.x;
  const uint32_t lane = t & 63;
  const uint32_t s = sg.sketch_size;

  if (t == 0) { ctrl->T = SENT; ctrl->count = 0; ctrl->overflow = 0; ctrl->saw_max = 0; ctrl->scan_base = 0; }
  __syncthreads();

  uint64_t T = SENT;
  bool safe_mode = true;
  const uint32_t room = (uint32_t)cap - s;  // >= 2048 by construction

  for (uint64_t T0 = sg.s_begin & ~15ULL; T0 < sg.s_end && s > 0; T0 += TILE_BASES) {
    // ---- stage the tile: positions [T0-32, T0+TILE_BASES) ----
    for (int c = t; c < TILE_DW / 4; c += WG) {
      const int64_t q = (int64_t)T0 - 32 + 16 * (int64_t)c;
      uint4 v;
      if (q >= (int64_t)sg.g_begin && q + 16 <= (int64_t)sg.g_end) {
        v = *reinterpret_cast<const uint4*>(seq + q);
      } else {
        uint32_t ww[4];
#pragma unroll
        for (int d = 0; d < 4; d++) {
          uint32_t x = 0;
#pragma unroll
          for (int b = 0; b < 4; b++) {
            const int64_t p = q + 4 * d + b;
            const uint32_t ch = (p >= (int64_t)sg.g_begin && p < (int64_t)sg.g_end) ? seq[p] : (uint32_t)'N';
            x |= ch << (8 * b);
          }
          ww[d] = x;
        }
        v = make_uint4(ww[0], ww[1], ww[2], ww[3]);
      }
      *reinterpret_cast<uint4*>(tile + 4 * c) = v;
    }
    __syncthreads();

    const uint32_t count_at_tile_start = ctrl->count;
    // owned positions of this lane relative to T0: [60t, 60t+60); hash window limits
    const int64_t lo64 = (int64_t)sg.s_begin - (int64_t)T0;
    const int64_t hi64 = (int64_t)sg.s_end - (int64_t)T0;
    const int rel_lo = lo64 < 0 ? 0 : (int)lo64;
    const int rel_hi = hi64 > TILE_BASES ? TILE_BASES : (int)hi64;

    bool redo;
    do {
      redo = false;
      uint64_t fwd = 0, rc = 0;
      int run = 0;
      int d = 0;
      int d_stop = safe_mode ? WARM_DW : (WARM_DW + RUN_DW);
      while (true) {
        for (; d < d_stop; d++) {
          const uint32_t wv = tile[t * RUN_DW + d];
          const bool hashing = d >= WARM_DW;
          const int rel0 = 60 * t + 4 * (d - WARM_DW);
#pragma unroll
          for (int b = 0; b < 4; b++) {
            const uint32_t c = (wv >> (8 * b)) & 0xffu;
            const uint32_t code2 = (c >> 1) & 3u;
            const uint32_t code = code2 ^ (code2 >> 1);
            const bool valid = ((c & 0xC0u) == 0x40u) && ((0x0010008Au >> (c & 31u)) & 1u);
            fwd = ((fwd << 2) | code) & P.kmask;
            rc = (rc >> 2) | ((uint64_t)(code ^ 3u) << P.rc_shift);
            run = valid ? run + 1 : 0;
            if (hashing) {
              const int rel = rel0 + b;
              const bool ok = run >= P.k && rel >= rel_lo && rel < rel_hi;
              const uint64_t canon = fwd < rc ? fwd : rc;
              const uint64_t h = kmer_hash(canon, P);
              const bool pass = ok && h < T;
              if (ok && h == SENT) ctrl->saw_max = 1;
              const uint64_t bal = __ballot(pass);
              if (bal) {  // wave-uniform
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&ctrl->count, (uint32_t)__popcll(bal));
                base = __shfl(base, 0);
                const uint32_t idx = base + (uint32_t)__popcll(bal & ((1ULL << lane) - 1ULL));
                if (pass) {
                  if (idx < (uint32_t)cap) buf[idx] = h;
                  else ctrl->overflow = 1;
                }
              }
            }
          }
        }
        if (d >= WARM_DW + RUN_DW) break;
        // safe mode: bound the next chunk so the buffer cannot overflow
        __syncthreads();
        uint32_t cn = ctrl->count;
        if ((uint32_t)cap - cn < (uint32_t)STEP_APPENDS) {
          merge_block(buf, ctrl, cap, s);
          T = ctrl->T;
          cn = ctrl->count;
        }
        int chunk = (int)(((uint32_t)cap - cn) / (uint32_t)STEP_APPENDS);
        d_stop = d + chunk;
        if (d_stop > WARM_DW + RUN_DW) d_stop = WARM_DW + RUN_DW;
        __syncthreads();  // everyone has read count before anyone appends again
      }
      __syncthreads();
      if (ctrl->overflow) {
        // optimistic pass lost candidates: fold what we have, then redo this tile safely
        // (count may exceed cap: clamp happens inside merge_block)
        merge_block(buf, ctrl, cap, s);
        T = ctrl->T;
        safe_mode = true;
        redo = true;
      }
    } while (redo);

    // ---- end of tile: decide about merging and the next tile's mode ----
    const uint32_t cn = ctrl->count;
    const uint32_t appended = cn - (count_at_tile_start < cn ? count_at_tile_start : cn);
    const bool need_merge = cn > s + room / 2;
    safe_mode = appended > room / 4;
    __syncthreads();  // all reads of ctrl->count / tile done before merge or next staging
    if (need_merge) {
      merge_block(buf, ctrl, cap, s);
      T = ctrl->T;
    }
  }

  // ---- final fold and write-out ----
  merge_block(buf, ctrl, cap, s);
  uint32_t n = ctrl->count;
  uint64_t* o = out + sg.out_off;
  for (uint32_t i = t; i < n; i += WG) o[i] = buf[i];
  if (t == 0) {
    if (ctrl->saw_max && n < s) { o[n] = SENT; n++; }
    cnt[sg.cnt_slot] = n;
  }
}

// ---- merge of per-segment partial sketches (one workgroup per multi-segment genome) ---------------
struct MergeJob {
  uint64_t part_off;   // element offset of first partial in partial buffer
  uint32_t part_cnt0;  // index of first partial's count
  uint32_t nparts;
  uint64_t out_off;
  uint32_t cnt_slot;
  uint32_t sketch_size;
  uint32_t stride;
  uint32_t pad;
};

__global__ __launch_bounds__(WG) void merge_partials_kernel(const MergeJob* __restrict__ jobs,
                                                            const uint64_t* __restrict__ parts,
                                                            const uint32_t* __restrict__ pcnt, int cap,
                                                            uint64_t* __restrict__ out,
                                                            uint32_t* __restrict__ cnt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t* buf = reinterpret_cast<uint64_t*>(smem);
  Ctrl* ctrl = reinterpret_cast<Ctrl*>(smem + (size_t)cap * 8);
  const MergeJob jb = jobs[blockIdx.x];
  const int t = threadIdx.x;
  const uint32_t s = jb.sketch_size;
  if (t == 0) { ctrl->T = SENT; ctrl->count = 0; ctrl->overflow = 0; ctrl->saw_max = 0; ctrl->scan_base = 0; }
  __syncthreads();
  for (uint32_t p = 0; p < jb.nparts; p++) {
    const uint32_t pc = pcnt[jb.part_cnt0 + p];
    const uint64_t* src = parts + jb.part_off + (uint64_t)p * jb.stride;
    const uint32_t base = ctrl->count;
    __syncthreads();
    for (uint32_t i = t; i < pc; i += WG) {
      uint64_t v = src[i];
      if (v == SENT) ctrl->saw_max = 1;
      buf[base + i] = v;  // SENT entries are dropped by the merge
    }
    __syncthreads();
    if (t == 0) ctrl->count = base + pc;
    merge_block(buf, ctrl, cap, s);
  }
  uint32_t n = ctrl->count;
  uint64_t* o = out + jb.out_off;
  for (uint32_t i = t; i < n; i += WG) o[i] = buf[i];
  if (t == 0) {
    if (ctrl->saw_max && n < s) { o[n] = SENT; n++; }
    cnt[jb.cnt_slot] = n;
  }
}

inline int pow2ceil(int x) { int p = 1; while (p < x) p <<= 1; return p; }

}  // namespace

extern "C" int rtc_sketch_minhash_dev(rtc_ctx* ctx, const uint8_t* d_seq, const uint64_t* h_off,
                                      uint32_t n, int k, uint32_t seed, const uint32_t* h_sizes,
                                      uint32_t size, uint64_t* d_out, uint32_t stride,
                                      uint32_t* d_cnt) {
  if (!ctx || !h_off || (n && (!d_seq || !d_out || !d_cnt))) return RTC_ERR_ARG;
  if (k < 1 || k > 32) return rtc_fail(ctx, RTC_ERR_ARG, "k=%d outside 1..32", k);
  if (n == 0) return RTC_OK;
  if (((uintptr_t)d_seq & 15) != 0) return rtc_fail(ctx, RTC_ERR_ARG, "d_seq must be 16-byte aligned");
  RTC_HIP(ctx, hipSetDevice(ctx->device));

  uint32_t smax = 0;
  for (uint32_t g = 0; g < n; g++) {
    uint32_t s = h_sizes ? h_sizes[g] : size;
    if (s > stride) return rtc_fail(ctx, RTC_ERR_ARG, "sketch size %u of genome %u exceeds stride %u", s, g, stride);
    smax = std::max(smax, s);
  }
  const int cap = pow2ceil((int)std::max<uint32_t>(2 * smax, smax + 2048));
  const size_t lds = (size_t)cap * 8 + (size_t)TILE_DW * 4 + sizeof(Ctrl);
  if (lds > (size_t)160 * 1024)
    return rtc_fail(ctx, RTC_ERR_UNSUPPORTED, "sketch size %u needs %zu B of LDS (> 160 KiB)", smax, lds);

  // ---- plan segments ----
  uint64_t total = 0;
  for (uint32_t g = 0; g < n; g++) {
    if (h_off[g + 1] < h_off[g]) return rtc_fail(ctx, RTC_ERR_ARG, "offsets not monotone at genome %u", g);
    total += h_off[g + 1] - h_off[g];
  }
  const uint64_t target_segs = (uint64_t)ctx->num_cu * 12;
  uint64_t seg_len = total / target_segs;
  const uint64_t min_seg = 8ull * TILE_BASES;
  if (seg_len < min_seg) seg_len = min_seg;
  std::vector<Segment> segs;
  std::vector<MergeJob> jobs;
  segs.reserve(n + 1024);
  uint64_t part_elems = 0;
  uint32_t part_slots = 0;
  for (uint32_t g = 0; g < n; g++) {
    const uint64_t b = h_off[g], e = h_off[g + 1], len = e - b;
    const uint32_t s = h_sizes ? h_sizes[g] : size;
    uint64_t ns = (len + seg_len / 2) / seg_len;
    if (ns < 1) ns = 1;
    if (ns > 4096) ns = 4096;
    if (ns == 1) {
      segs.push_back(Segment{b, e, b, e, (uint64_t)g * stride, g, s});
    } else {
      MergeJob jb{part_elems, part_slots, (uint32_t)ns, (uint64_t)g * stride, g, s, stride, 0};
      for (uint64_t i = 0; i < ns; i++) {
        uint64_t sb = b + len * i / ns, se = b + len * (i + 1) / ns;
        // partial sketches live in scratch: offsets are relative to the partial buffer and
        // flagged by cnt_slot >= n (resolved below)
        segs.push_back(Segment{b, e, sb, se, part_elems, n + part_slots, s});
        part_elems += stride;
        part_slots++;
      }
      jobs.push_back(jb);
    }
  }
  // Partial outputs and final outputs use different base pointers: launch the kernel twice over
  // disjoint segment lists (direct-to-output first, partials second) to keep the kernel simple.
  std::vector<Segment> direct, partial;
  for (const Segment& sgm : segs) {
    if (sgm.cnt_slot < n) direct.push_back(sgm);
    else { Segment p = sgm; p.cnt_slot -= n; partial.push_back(p); }
  }
  const size_t bseg = (direct.size() + partial.size()) * sizeof(Segment);
  const size_t bjobs = jobs.size() * sizeof(MergeJob);
  void* ws0 = nullptr;
  RTC_TRY(rtc_ws(ctx, 0, bseg + bjobs + 64, &ws0));
  Segment* d_direct = (Segment*)ws0;
  Segment* d_partial = d_direct + direct.size();
  MergeJob* d_jobs = (MergeJob*)((char*)ws0 + bseg);
  void* hp = nullptr;
  RTC_TRY(rtc_pinned(ctx, bseg + bjobs + 64, &hp));
  RTC_HIP(ctx, hipStreamSynchronize(ctx->stream));  // pinned staging may still be in flight
  memcpy(hp, direct.data(), direct.size() * sizeof(Segment));
  memcpy((char*)hp + direct.size() * sizeof(Segment), partial.data(), partial.size() * sizeof(Segment));
  memcpy((char*)hp + bseg, jobs.data(), bjobs);
  RTC_HIP(ctx, hipMemcpyAsync(ws0, hp, bseg + bjobs, hipMemcpyHostToDevice, ctx->stream));

  uint64_t* d_parts = nullptr;
  uint32_t* d_pcnt = nullptr;
  if (!partial.empty()) {
    void* ws1 = nullptr;
    RTC_TRY(rtc_ws(ctx, 1, part_elems * 8 + (size_t)part_slots * 4 + 64, &ws1));
    d_parts = (uint64_t*)ws1;
    d_pcnt = (uint32_t*)((char*)ws1 + part_elems * 8);
  }

  RTC_HIP(ctx, hipFuncSetAttribute((const void*)sketch_minhash_kernel,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  if (!direct.empty()) {
    hipLaunchKernelGGL(sketch_minhash_kernel, dim3((uint32_t)direct.size()), dim3(WG), lds, ctx->stream,
                       d_seq, d_direct, k, seed, cap, d_out, d_cnt);
    RTC_CHECK_LAUNCH(ctx);
  }
  if (!partial.empty()) {
    hipLaunchKernelGGL(sketch_minhash_kernel, dim3((uint32_t)partial.size()), dim3(WG), lds, ctx->stream,
                       d_seq, d_partial, k, seed, cap, d_parts, d_pcnt);
    RTC_CHECK_LAUNCH(ctx);
    const size_t lds_m = (size_t)cap * 8 + sizeof(Ctrl);
    RTC_HIP(ctx, hipFuncSetAttribute((const void*)merge_partials_kernel,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_m));
    hipLaunchKernelGGL(merge_partials_kernel, dim3((uint32_t)jobs.size()), dim3(WG), lds_m, ctx->stream,
                       d_jobs, d_parts, d_pcnt, cap, d_out, d_cnt);
    RTC_CHECK_LAUNCH(ctx);
  }
  return RTC_OK;
}

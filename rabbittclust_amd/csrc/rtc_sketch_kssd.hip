// rtc_sketch_kssd.hip -- KSSD (--fast) sketching on gfx950.
//
// Replaces the per-file body of sketchFileWithKssd (src/SketchInfo.cpp:994-1252 in the reference
// tree): rolling 2-bit k-mer and reverse complement (:1126-1135), canonical minimum (:1141),
// shuffled-dimension filter (:1142-1149), dimension-reduced tuple (:1150-1152), set + ascending
// sort (:1154-1157, :1180-1193).
//
// The reference probes a phmap of the kept (dim_id -> rank) pairs per k-mer.  Here the kept set
// (dim_end of 2^(4*half_subk) ids; 4096 of 16 Mi at the default drlevel 3) is compiled on the host:
//  * 24-bit dim_id, K = 18..28 (every default configuration): sketch_kssd_bloom_kernel.  The steady state looks
//    at the FORWARD strand only, in two stages over the kept middle 12-mers and their reverse complements: a 2^18-bit
//    map in LDS of the 18 bits the four k-mers of a dword share (one read per dword), then, 64 surviving dwords at a
//    time, a blocked Bloom filter; it queues the positions of possible hits, and those are finished exactly (both
//    strands, canonical minimum, an exact bucket index of the kept ids: 8192 buckets of four 16-bit patterns + ranks,
//    in global memory).
//  * two-table cuckoo index in LDS (other k-mer lengths, 28-bit dim_id): two ds_read_b32 per k-mer.
//  * the full table in HBM when more than 8192 ids are kept (drlevel <= 2) or nothing else can be built.
// Survivors are appended to the genome's output row with wave-aggregated atomics; each row is then sorted and
// deduplicated -- by one wave in registers up to 1 024 tuples (kssd_sort_unique_wave_kernel), in LDS
// (kssd_sort_unique_kernel) or by a global merge sort (kssd_big_*) beyond.
#include <algorithm>
#include <vector>

#include "rtc_internal.h"

namespace {

constexpr int WG = 512;
// Lane geometry per tile (template parameters of the kernel): RUN_DW dwords of owned k-mer end positions
// behind WARM_DW warm-up dwords that only roll the windows (>= K-1 bases), together whole 16-byte loads.
//   K <= 25:  18 + 6 dwords (72 owned + 24 warm-up bases, six loads)
//   else:     19 + 9 dwords (76 + 36, seven loads)
// Lane runs of 72-76 B keep the live 128-B lines inside an XCD's L2 (see rtc_sketch_minhash.hip).
constexpr int TILE_BASES_MAX = WG * 19 * 4;
constexpr int MAX_LDS_KEEP = 8192;
constexpr int BUCKET_BYTES = 65536;  // bucket index: 8192 buckets x four 16-bit patterns

struct KSegment {
  uint64_t g_begin, g_end;
  uint64_t s_begin, s_end;
  uint32_t genome;
  uint32_t pad;
};

struct KssdParams {
  int K;             // even k-mer length (2*half_k)
  int drlevel;
  int use64;
  int rev_add_move;  // 4*half_k - 2
  int dim_shift;     // 2*half_outctx_len
  int und1_shl;      // 2K - 4*half_outctx_len
  int dimbits;       // 4*half_subk (24 or 28)
  int ck1, ck2;      // log2 slots of cuckoo table 1 (low bits of dim_id) / table 2 (high bits)
  int dim_end;
  int lshift;        // 64 - 2K: top-aligned windows
  uint32_t dimmask;  // low `dimbits` bits
  uint32_t m1key, m2key;  // entry bits compared with dim_id in table 1 / table 2
  int nofast;        // prefilter kernel: RTC_KSSD_NOFAST=1 sends every chunk through the guarded loads (tests)
  uint64_t tupmask, domask, undomask0, undomask1;
};

__device__ __forceinline__ uint4 load_bases16(const uint8_t* __restrict__ seq, int64_t q, uint64_t g_begin,
                                              uint64_t g_end) {
  if (q >= (int64_t)g_begin && q + 16 <= (int64_t)g_end) return *reinterpret_cast<const uint4*>(seq + q);
  uint32_t ww[4];
#pragma unroll
  for (int d = 0; d < 4; d++) {
    uint32_t x = 0;
#pragma unroll
    for (int b = 0; b < 4; b++) {
      const int64_t p = q + 4 * d + b;
      const uint32_t ch = (p >= (int64_t)g_begin && p < (int64_t)g_end) ? seq[p] : (uint32_t)'N';
      x |= ch << (8 * b);
    }
    ww[d] = x;
  }
  return make_uint4(ww[0], ww[1], ww[2], ww[3]);
}

enum { IDX_HBM = 0, IDX_CUCKOO = 1 };

struct KssdTables {
  const uint32_t* l_t1;         // cuckoo table 1 (LDS)
  const uint32_t* l_t2;         // cuckoo table 2 (LDS)
  const int32_t* g_table;       // full shuffle table (HBM path)
};

#define RTC_LDS __attribute__((address_space(3)))
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint64_t reduced_tuple(const KssdParams& P, uint64_t u, uint32_t rank) {   // :1150-1152
  return (((u & P.undomask0) | ((u & P.undomask1) << P.und1_shl)) >> (P.drlevel * 4)) | (uint64_t)rank;
}

// wave-aggregated append of the lanes in `bal` to the genome's output row
__device__ __forceinline__ void append_tuples(uint64_t bal, bool keep, uint64_t dr, uint32_t lane, void* orow,
                                              uint32_t* ocnt, uint32_t stride, int use64) {
  uint32_t base = 0;
  if (lane == 0) base = atomicAdd(ocnt, (uint32_t)__popcll(bal));
  base = __shfl(base, 0);
  const uint32_t idx = base + (uint32_t)__popcll(bal & ((1ULL << lane) - 1ULL));
  if (keep && idx < stride) {
    if (use64) reinterpret_cast<uint64_t*>(orow)[idx] = dr;
    else reinterpret_cast<uint32_t*>(orow)[idx] = (uint32_t)dr;
  }
}

// The general walk over the 16-byte groups grp0.. of one lane's tile window: guarded loads, any
// character, any segment / genome edge.  Each lane walks 96 (K <= 25: 24 warm-up + 72 owned k-mer end
// positions) or 112 (36 + 76) consecutive bases per tile, read straight from global memory as 16-byte
// loads; four bases are decoded at once (SWAR) while the wave holds only valid bases, one by one otherwise.
template <int IDX, int RUN_DW, int WARM_DW>
__device__ __forceinline__ void generic_groups(const uint8_t* __restrict__ seq, const KSegment& sg, const KssdParams& P,
                                               const KssdTables& TB, uint64_t T0, int rel_lo, int rel_hi, int t,
                                               uint32_t lane, void* orow, uint32_t* ocnt, uint32_t stride, int grp0,
                                               uint64_t tuple, uint64_t rvs, int run, bool clean) {
  constexpr int OWN = RUN_DW * 4;
  constexpr int TILE_BASES = WG * RUN_DW * 4;
  const bool interior = rel_lo == 0 && rel_hi == TILE_BASES;  // every position of the tile is owned
  const int64_t p0 = (int64_t)T0 + OWN * t - 4 * WARM_DW;
  const bool fastroll = P.K <= 28;  // 2K+8 bits fit the 64-bit extended window
  const uint32_t m1mask = (1u << P.ck1) - 1u;
  uint4 nxt = load_bases16(seq, p0 + 16 * grp0, sg.g_begin, sg.g_end);
  for (int grp = grp0; grp < (WARM_DW + RUN_DW) / 4; grp++) {
    const uint4 cur = nxt;
    if (grp + 1 < (WARM_DW + RUN_DW) / 4) nxt = load_bases16(seq, p0 + 16 * (grp + 1), sg.g_begin, sg.g_end);
    const uint32_t wv4[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
    for (int qd = 0; qd < 4; qd++) {
      const int d = grp * 4 + qd;
      const uint32_t wv = wv4[qd];
      const bool emitting = d >= WARM_DW;  // wave-uniform
      const int rel0 = OWN * t + 4 * (d - WARM_DW);
      uint64_t uni[4] = {0, 0, 0, 0};
      bool ok[4] = {false, false, false, false};
      const uint32_t up = wv & 0xDFDFDFDFu;
      const uint32_t codes4 = ((wv >> 1) ^ (wv >> 2)) & 0x03030303u;  // BaseMap, src/SketchInfo.cpp:1007-1017
      const bool allvalid = __builtin_amdgcn_perm(0u, 0x54474341u, codes4) == up;
      const bool fast = fastroll && __all(allvalid);  // wave-uniform
      clean = clean && fast;
      if (fast) {
        const uint32_t pack = __builtin_amdgcn_udot4(codes4, 0x01041040u, 0u, false);  // c0<<6|c1<<4|c2<<2|c3
        const uint32_t rp = __builtin_amdgcn_udot4(codes4, 0x40100401u, 0u, false) ^ 0xffu;
        const uint64_t F = (tuple << 8) | pack;
        const uint64_t R = rvs | ((uint64_t)rp << (2 * P.K));
        if (emitting) {
          // scalar ownership test for the steady state (tile interior to the segment, only valid
          // bases in this wave since the tile began => run = 4d >= 4*WARM_DW >= K-1 and every position owned)
          const bool allok = interior && clean;
          // the four windows top-aligned (tuple :1134 / rvs :1135 four times): bits below the window are
          // not cleaned -- they cannot change which of two different k-mers is smaller (:1141), and of
          // two equal ones either will do
#pragma unroll
          for (int b = 0; b < 4; b++) {
            const uint64_t f = F << (P.lshift - 6 + 2 * b);
            const uint64_t r = R << (P.lshift - 2 - 2 * b);
            uni[b] = f < r ? f : r;
          }
          if (allok) {
#pragma unroll
            for (int b = 0; b < 4; b++) ok[b] = true;
          } else {
#pragma unroll
            for (int b = 0; b < 4; b++) {
              const int rel = rel0 + b;
              ok[b] = run + b + 1 >= P.K && rel >= rel_lo && rel < rel_hi;           // :1139
            }
          }
        }
        tuple = F;      // bits above the window are masked where windows are cut / by the per-base path
        rvs = R >> 8;   // R < 2^(2K+8) by construction
        run += 4;
      } else {
#pragma unroll
        for (int b = 0; b < 4; b++) {
          const uint32_t c = (wv >> (8 * b)) & 0xffu;
          const uint32_t code = ((c >> 1) ^ (c >> 2)) & 3u;
          const bool valid = ((c & 0xC0u) == 0x40u) && ((0x0010008Au >> (c & 31u)) & 1u);
          tuple = ((tuple << 2) | code) & P.tupmask;                               // :1134
          rvs = (rvs >> 2) + ((uint64_t)(code ^ 3u) << P.rev_add_move);            // :1135
          run = valid ? run + 1 : 0;                                               // base counter :1136,1161
          const int rel = rel0 + b;
          ok[b] = run >= P.K && rel >= rel_lo && rel < rel_hi;
          uni[b] = (tuple < rvs ? tuple : rvs) << P.lshift;
        }
      }
      if (!emitting) continue;
      uint32_t rank[4];
      bool keep[4];
      // dim_id (:1142) sits at bit lshift + dim_shift of a top-aligned window: shifting the bits above it
      // out leaves it at the top of a 32-bit word, from where the table-2 slot (its high ck2 bits) and
      // the value itself are one 32-bit shift each -- no masks
      const int drop = 64 - (P.lshift + P.dim_shift) - P.dimbits;
#pragma unroll
      for (int b = 0; b < 4; b++) {
        const uint32_t xh = (uint32_t)((uni[b] << drop) >> 32);
        const uint32_t dim_id = xh >> (32 - P.dimbits);
        rank[b] = 0;
        if (IDX == IDX_CUCKOO) {
          const uint32_t e1 = TB.l_t1[dim_id & m1mask];
          const uint32_t e2 = TB.l_t2[xh >> (32 - P.ck2)];
          const bool m1 = ((e1 ^ dim_id) & P.m1key) == 0u;
          const bool m2 = ((e2 ^ dim_id) & P.m2key) == 0u;
          rank[b] = m1 ? (e1 & 0xfffu) : (e2 >> 20);
          keep[b] = ok[b] && (m1 || m2);
        } else {
          keep[b] = false;
          if (ok[b]) {
            const int32_t sd = TB.g_table[dim_id];
            keep[b] = sd >= 0 && sd < P.dim_end;                                   // :1054
            rank[b] = (uint32_t)sd;
          }
        }
      }
      if (!__any(keep[0] | keep[1] | keep[2] | keep[3])) continue;
#pragma unroll
      for (int b = 0; b < 4; b++) {
        const uint64_t bal = __ballot(keep[b]);
        if (bal) {  // wave-uniform
          const uint64_t u = uni[b] >> P.lshift;  // the exact 2K-bit tuple
          append_tuples(bal, keep[b], reduced_tuple(P, u, rank[b]), lane, orow, ocnt, stride, P.use64);
        }
      }
    }
  }
}

template <int IDX, int RUN_DW, int WARM_DW>
__global__ __launch_bounds__(WG) void sketch_kssd_kernel(const uint8_t* __restrict__ seq,
                                                         const KSegment* __restrict__ segs, KssdParams P,
                                                         const uint32_t* __restrict__ g_t1,    // cuckoo table 1 [1 << ck1]
                                                         const uint32_t* __restrict__ g_t2,    // cuckoo table 2 [1 << ck2]
                                                         const int32_t* __restrict__ g_table,  // full table (HBM path)
                                                         void* __restrict__ out, uint32_t stride,
                                                         uint32_t* __restrict__ cnt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t* l_t1 = reinterpret_cast<uint32_t*>(smem);
  uint32_t* l_t2 = l_t1 + (1u << P.ck1);

  static_assert((RUN_DW + WARM_DW) % 4 == 0, "a lane's window must be whole 16-byte loads");
  constexpr int TILE_BASES = WG * RUN_DW * 4;
  const KSegment sg = segs[blockIdx.x];
  const int t = threadIdx.x;
  const uint32_t lane = t & 63;
  if (IDX == IDX_CUCKOO) {
    for (int i = t; i < (1 << P.ck1); i += WG) l_t1[i] = g_t1[i];
    for (int i = t; i < (1 << P.ck2); i += WG) l_t2[i] = g_t2[i];
    __syncthreads();
  }
  const KssdTables TB{l_t1, l_t2, g_table};
  void* orow = reinterpret_cast<unsigned char*>(out) + (uint64_t)sg.genome * stride * (P.use64 ? 8 : 4);
  uint32_t* ocnt = cnt + sg.genome;

  for (uint64_t T0 = sg.s_begin & ~15ULL; T0 < sg.s_end; T0 += TILE_BASES) {
    const int64_t lo64 = (int64_t)sg.s_begin - (int64_t)T0;
    const int64_t hi64 = (int64_t)sg.s_end - (int64_t)T0;
    const int rel_lo = lo64 < 0 ? 0 : (int)lo64;
    const int rel_hi = hi64 > TILE_BASES ? TILE_BASES : (int)hi64;
    generic_groups<IDX, RUN_DW, WARM_DW>(seq, sg, P, TB, T0, rel_lo, rel_hi, t, lane, orow, ocnt, stride, 0, 0ULL, 0ULL,
                                            0, true);
  }
}

constexpr int WGB = 1024;  // lanes per workgroup of the prefilter kernel: 2 workgroups per CU (64 KiB of filters + 16 KiB of queues each) = 8 waves per SIMD
constexpr int WGB_WAVES_EU = 8;
typedef uint16_t bq_t;     // exact-drain queue entry: dword index relative to the queue's base

// ---- forward-strand prefilter (the default --fast configuration, K = 18..28, 24-bit dim_id) --------------
// A k-mer is kept when the dim_id of its CANONICAL form is one of the dim_end kept dimensions (:1141-1149).
// dim_id is the k-mer's middle 12 bases, and the middle of the reverse complement is the reverse complement of
// the middle: whichever strand is canonical, the FORWARD k-mer's middle 12-mer lies in S2 = S u rc(S) (8192
// 12-mers of 16 Mi).  So the steady state never forms the reverse strand, never compares strands and never checks
// characters.  It is a conservative candidate generator in two stages:
//   * a wave walks a contiguous stretch of its segment 1 KiB at a time, lane l holding bases 16 l .. 16 l + 15 of the
//     chunk (ONE coalesced 16-byte load per lane: every 128-byte line is requested once);
//   * the lane packs its 16 bases into one register (four SWAR decodes + v_dot4), takes the 16 (32) bases in
//     front of them from its neighbour lane(s) with a DPP wave shift (lane 0: carried over from the previous
//     chunk in SGPRs) -- no warm-up bases at all -- and cuts the 32-bit word E that holds the four middle 12-mers of a
//     dword's k-mers with one v_alignbit;
//   * stage 1, once per DWORD: the four middle 12-mers of a dword's k-mers share nine bases (18 bits of E).  A
//     2^18-bit map in LDS (32 KiB) says whether those 18 bits occur in ANY member of S2 at one of the four
//     alignments (11.7 % of the map is set): one ds_read_b32 and six VALU instructions for four k-mers.  Until round
//     4 every k-mer probed a blocked Bloom filter -- one random ds_read_b64 each, and the kernel sat on the LDS
//     (SQ_LDS_IDX_ACTIVE 81 % of the cycles, 69 % of them bank conflicts: 32 random bank pairs per 32-lane group);
//   * the dwords that pass (one lane in eight) go, as (E, position), to a per-wave LDS queue; stage 2 takes 64 of
//     them at a time -- every lane busy -- and probes a blocked Bloom filter of S2 (3584 blocks of 64 bits, block
//     and bits from a multiplicative hash of the whole 12-mer, so that a 12-mer that shares 18 bits with a member
//     is no likelier to pass than any other) for the four k-mers of each.
// Nothing is lost: a valid k-mer's bases decode exactly, and whatever else decodes to a hit (characters outside
// ACGT, bases beyond a genome's end, false positives: together 0.5 % of the dwords, 0.2 % are true members) is
// dropped later.  Survivors put the dword's POSITION into a second per-wave queue; bloom_drain re-reads those K + 3
// bases from memory (L2 / Infinity Cache) and does the reference's arithmetic exactly: characters, both strands,
// canonical minimum, dimension lookup (the exact bucket index, read from global memory here), reduced tuple, append.
constexpr int CORE_BYTES = 32768;                  // stage 1: 2^18 bits
constexpr uint32_t BLOOM2_BLOCKS = 3584;           // stage 2: 3584 blocks x 8 B (28 KiB: what two workgroups per CU leave)
constexpr int BLOOM2_BYTES = BLOOM2_BLOCKS * 8;
constexpr int BLOOM_BYTES = CORE_BYTES + BLOOM2_BYTES;   // what the host uploads: [map | filter]
constexpr uint32_t BLOOM2_MUL = 0x9E3779B1u;
constexpr int Q1_CAP = 128;                        // stage-1 survivors per wave, (E, position) pairs of 8 bytes: < 64 left over + what a
                                                   // chunk adds; a chunk that would not fit (more than a quarter of its dwords pass the map:
                                                   // 11.7 % do on random sequence) is walked dword by dword with the queues served in between
constexpr int Q1_BYTES = (WGB / 64) * Q1_CAP * 8;
constexpr int BQ_CAP = 128;                        // stage-2 survivors per wave: < 64 left over + at most 64 per stage-2 batch
constexpr int BQ_BYTES = (WGB / 64) * BQ_CAP * (int)sizeof(bq_t);
static_assert(BLOOM_BYTES + Q1_BYTES + BQ_BYTES <= 81920, "two workgroups per CU");
constexpr int BQ_SPAN = 255;                       // chunks a wave may walk on one queue base: 255 * 256 + 255 dwords < 2^16
constexpr int CHUNK = 1024;                        // bases a wave takes per step (64 lanes x 16)
typedef bq_t RTC_LDS* lds_u32_ptr;
typedef u32x2 RTC_LDS* lds_q1_ptr;

struct BloomSeg {
  uint64_t g_begin, g_end, s_begin, s_end, base;   // base: queue entries are positions relative to it
};

__device__ __forceinline__ uint32_t exact_rank_global(int var, uint32_t dim_id, const uint32_t* __restrict__ g_bk,
                                                      const uint16_t* __restrict__ g_rank) {
  const uint32_t addr = var ? ((dim_id >> 8) & 0xfff8u) : (dim_id & 0xfff8u);
  const uint32_t q = var ? (dim_id & 0xffffu) : ((dim_id & 0xffu) | ((dim_id >> 8) & 0xff00u));
  const uint2 e = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned char*>(g_bk) + addr);
  const uint32_t pat[4] = {e.x & 0xffffu, e.x >> 16, e.y & 0xffffu, e.y >> 16};
  uint32_t found = 0xffffffffu;
#pragma unroll
  for (int sl = 0; sl < 4; sl++)
    if (pat[sl] == q) found = g_rank[(addr >> 1) + sl] & 0xfffu;
  return found;
}

// the k-mer `u` (canonical, exact 2K bits) of a lane with ok set: dimension lookup, reduced tuple, append
__device__ __forceinline__ void bloom_emit(bool ok, uint64_t u, const KssdParams& P, const uint32_t* __restrict__ g_bk,
                                           const uint16_t* __restrict__ g_rank, int var, uint32_t lane, void* orow,
                                           uint32_t* ocnt, uint32_t stride) {
  const uint32_t dim_id = (uint32_t)(u >> P.dim_shift) & 0xffffffu;              // :1142
  uint32_t rank = 0xffffffffu;
  if (ok) rank = exact_rank_global(var, dim_id, g_bk, g_rank);
  const bool keep = rank != 0xffffffffu;
  const uint64_t bal = __ballot(keep);
  if (bal) append_tuples(bal, keep, reduced_tuple(P, u, rank), lane, orow, ocnt, stride, P.use64);
}

// n <= 64 queued dwords wq[first .. first + n), one per lane.  The K + 3 bases that end the dword's four k-mers are read again:
// when they are all ACGTacgt (one vote for the batch) both strands of the whole stretch come out of seven SWAR
// decodes and the four k-mers are cut out of them; otherwise the batch is walked base by base exactly as the
// reference does (:1126-1161) -- any character, genome edges.
template <int K>
__device__ __forceinline__ void bloom_drain(const uint8_t* __restrict__ seq, const BloomSeg& sg, const KssdParams& P,
                                            const uint32_t* __restrict__ g_bk, const uint16_t* __restrict__ g_rank, int var,
                                            lds_u32_ptr wq, uint32_t first, uint32_t n, uint32_t lane, void* orow,
                                            uint32_t* ocnt, uint32_t stride) {
  constexpr int NB = K + 3;             // bases walked: the first k-mer's first .. the last k-mer's last
  constexpr int NQ = (NB + 3) / 4;      // dwords that hold them once aligned; PAD bases follow the last k-mer
  constexpr int PAD = 4 * NQ - NB;
  constexpr int NDW = NQ + 1;           // dwords that cover them at any alignment
  {
    const bool have = lane < n;
    const int64_t q0 = (int64_t)sg.base + (have ? 4 * (int64_t)wq[first + lane] : 0);  // first base of the dword
    const int64_t b0 = q0 - (K - 1);                             // first base of its first k-mer
    const int64_t a0 = b0 & ~(int64_t)3;
    uint32_t w[NDW + 1];
    if (have && a0 >= (int64_t)sg.g_begin && a0 + 4 * NDW <= (int64_t)sg.g_end) {
#pragma unroll
      for (int j = 0; j < NDW; j++) w[j] = *reinterpret_cast<const uint32_t*>(seq + a0 + 4 * j);
    } else {
#pragma unroll 1
      for (int j = 0; j < NDW; j++) {
        uint32_t x = 0;
#pragma unroll 1
        for (int b = 3; b >= 0; b--) {
          const int64_t pp = a0 + 4 * j + b;
          const uint32_t ch = (have && pp >= (int64_t)sg.g_begin && pp < (int64_t)sg.g_end) ? seq[pp] : (uint32_t)'N';
          x = (x << 8) | ch;
        }
        w[j] = x;
      }
    }
    w[NDW] = 0;
    const uint32_t sh = (uint32_t)(b0 - a0) * 8u;  // bytes to drop in front: 0, 8, 16, 24 bits
#pragma unroll
    for (int j = 0; j < NQ; j++) w[j] = __builtin_amdgcn_alignbit(w[j + 1], w[j], sh);  // w[] now starts at b0
    // ownership of the four k-mer end positions q0 .. q0 + 3
    bool own[4];
#pragma unroll
    for (int b = 0; b < 4; b++) own[b] = have && q0 + b >= (int64_t)sg.s_begin && q0 + b < (int64_t)sg.s_end;
    uint32_t codes[NQ], bad = 0;
#pragma unroll
    for (int j = 0; j < NQ; j++) {
      codes[j] = ((w[j] >> 1) ^ (w[j] >> 2)) & 0x03030303u;                       // BaseMap, src/SketchInfo.cpp:1007-1017
      bad = __builtin_amdgcn_bitop3_b32(bad, __builtin_amdgcn_perm(0u, 0x54474341u, codes[j]), w[j], 0xF6);  // bad | (perm ^ w)
    }
    if (!__ballot(have && (bad & 0xDFDFDFDFu) != 0u)) {
      // every base of every lane is one of ACGTacgt: F = the 4 NQ bases, first on top; R = their reverse
      // complement (v_dot4 per dword gives the forward byte and the reverse-complement byte)
      uint64_t F = 0, R = 0;
#pragma unroll
      for (int j = 0; j < NQ; j++) {
        const uint32_t pack = __builtin_amdgcn_udot4(codes[j], 0x01041040u, 0u, false);          // c0<<6|c1<<4|c2<<2|c3
        const uint32_t rp = __builtin_amdgcn_udot4(codes[j], 0x40100401u, 0u, false) ^ 0xffu;   // complements, reversed
        F |= (uint64_t)pack << (8 * (NQ - 1 - j));
        R |= (uint64_t)rp << (8 * j);
      }
#pragma unroll
      for (int b = 0; b < 4; b++) {
        const uint64_t tuple = (F >> (2 * (PAD + 3 - b))) & P.tupmask;           // bases b .. b + K - 1 (:1134)
        const uint64_t rvs = (R >> (2 * b)) & P.tupmask;                         // their reverse complement (:1135)
        bloom_emit(own[b], tuple < rvs ? tuple : rvs, P, g_bk, g_rank, var, lane, orow, ocnt, stride);   // :1141
      }
    } else {
      uint64_t tuple = 0, rvs = 0;
      int run = 0;
#pragma unroll 1
      for (int c = 0; c < NB; c++) {
        const uint32_t ch = w[0] & 0xffu;
        // the stream moves down one byte (NQ dwords; rolled: this is the rare path)
#pragma unroll
        for (int j = 0; j < NQ; j++) w[j] = __builtin_amdgcn_alignbit(j + 1 < NQ ? w[j + 1] : 0u, w[j], 8);
        const uint32_t code = ((ch >> 1) ^ (ch >> 2)) & 3u;
        const bool valid = ((ch & 0xC0u) == 0x40u) && ((0x0010008Au >> (ch & 31u)) & 1u);
        tuple = ((tuple << 2) | code) & P.tupmask;                               // :1134
        rvs = (rvs >> 2) + ((uint64_t)(code ^ 3u) << P.rev_add_move);            // :1135
        run = valid ? run + 1 : 0;                                               // base counter :1136,1161
        if (c < K - 1) continue;
        const int64_t pos = b0 + c;
        const bool ok = have && run >= K && pos >= (int64_t)sg.s_begin && pos < (int64_t)sg.s_end;   // :1139 + ownership
        bloom_emit(ok, tuple < rvs ? tuple : rvs, P, g_bk, g_rank, var, lane, orow, ocnt, stride);
      }
    }
  }
}

// 16 bases at q, 'N' for every position outside the genome: chunks at a genome's ends only (one instance, rolled)
__device__ __noinline__ uint4 load_bases16_edge(const uint8_t* __restrict__ seq, int64_t q, uint64_t g_begin, uint64_t g_end) {
  uint32_t ww[4];
#pragma unroll 1
  for (int d = 0; d < 4; d++) {
    uint32_t x = 0;
#pragma unroll 1
    for (int b = 3; b >= 0; b--) {
      const int64_t p = q + 4 * d + b;
      x = (x << 8) | ((p >= (int64_t)g_begin && p < (int64_t)g_end) ? (uint32_t)seq[p] : (uint32_t)'N');
    }
    ww[d] = x;
  }
  return make_uint4(ww[0], ww[1], ww[2], ww[3]);
}

// lane l receives lane l - 1's value, lane 0 keeps `first` (DPP wave_shr:1)
__device__ __forceinline__ uint32_t from_lane_below(uint32_t v, uint32_t first) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)first, (int)v, 0x138, 0xf, 0xf, false);
}

// stage 2: n <= 64 queued dwords q[0 .. n), one per lane: the four k-mers of each against the Bloom filter of S2 (the
// four blocks requested together); the owned survivors' positions go to the exact-drain queue.  Returns the new length
// of that queue (at most + 64).
template <int K>
__device__ __forceinline__ uint32_t bloom_stage2(lds_q1_ptr q, uint32_t n, uint32_t lane, lds_u32_ptr wq, uint32_t qn,
                                                 const BloomSeg& bs) {
  constexpr int DS = K - 12;
  constexpr bool NARROW = DS >= 8 && DS <= 10;
  constexpr int FO0 = NARROW ? DS - 2 : 6;
  const bool have = lane < n;
  u32x2 ent = {0u, 0u};
  if (have) ent = q[lane];
  const uint32_t E = ent.x;
  uint32_t h[4];
  u32x2 blk[4];
#pragma unroll
  for (int b = 0; b < 4; b++) {
    h[b] = ((E >> (FO0 - 2 * b)) & 0xffffffu) * BLOOM2_MUL;
    blk[b] = *(const RTC_LDS u32x2*)(uintptr_t)((uint32_t)CORE_BYTES + (__umulhi(h[b], BLOOM2_BLOCKS) << 3));
  }
  uint32_t acc = 0;
#pragma unroll
  for (int b = 0; b < 4; b++)  // four bits per member
    acc |= (blk[b].x >> ((h[b] >> 8) & 31u)) & (blk[b].x >> ((h[b] >> 3) & 31u)) & (blk[b].y >> ((h[b] >> 13) & 31u)) & (blk[b].y >> (h[b] & 31u));
  // the dword's k-mers end at pos .. pos + 3: queued only when one of them is owned by this segment
  const int64_t pos = (int64_t)bs.base + 4 * (int64_t)ent.y;
  const bool mine = have && (acc & 1u) != 0u && pos + 3 >= (int64_t)bs.s_begin && pos < (int64_t)bs.s_end;
  const uint64_t bal = __ballot(mine);
  if (mine) wq[qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u))] = (bq_t)ent.y;
  return qn + (uint32_t)__popcll(bal);
}

template <int K>
__global__ __launch_bounds__(WGB, WGB_WAVES_EU) void sketch_kssd_bloom_kernel(const uint8_t* __restrict__ seq,
                                                               const KSegment* __restrict__ segs, KssdParams P,
                                                               const uint32_t* __restrict__ g_bloom,  // [2^18-bit map | 3584 x 8 B]
                                                               const uint32_t* __restrict__ g_bk,     // exact index, patterns
                                                               const uint16_t* __restrict__ g_rank,   // exact index, ranks
                                                               int var, void* __restrict__ out, uint32_t stride,
                                                               uint32_t* __restrict__ cnt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  static_assert(K % 2 == 0 && K >= 18 && K <= 28, "24-bit dim_id in the middle of at most 28 bases");
  constexpr int DS = K - 12;                       // dim_shift: bit of a window where its middle 12-mer starts
  // With X = (bases before the lane's 16 : the lane's 16 bases), the word E of dword qd holds the four middle
  // 12-mers of the k-mers that end in it.  K = 20, 22: E = the 16 bases in front of the dword (fields at bits
  // [DS - 2 - 2b, DS + 22 - 2b)); otherwise E = the 32 bits from DS up of the window that ends with the dword
  // (fields at [6 - 2b, 30 - 2b)), which reaches into the second neighbour's bases.  The 18 bits [FO0, FO0 + 18)
  // of E belong to all four fields.
  constexpr bool NARROW = DS >= 8 && DS <= 10;
  constexpr int FO0 = NARROW ? DS - 2 : 6;
  constexpr int AHEAD = 3;                         // chunks requested ahead of the one being walked (= the unroll of the chunk loop; 2 .. 6 measured at eight waves per SIMD: 3 is best by 2-3 %)
  const KSegment sg = segs[blockIdx.x];
  const int t = threadIdx.x;
  const uint32_t lane = t & 63;
  {
    uint4* l4 = reinterpret_cast<uint4*>(smem);
    const uint4* g4 = reinterpret_cast<const uint4*>(g_bloom);
    for (int i = t; i < BLOOM_BYTES / 16; i += WGB) l4[i] = g4[i];
    __syncthreads();
  }
  if ((uint32_t)(uintptr_t)(RTC_LDS unsigned char*)smem != 0u) __builtin_trap();  // the filters are addressed absolutely
  void* orow = reinterpret_cast<unsigned char*>(out) + (uint64_t)sg.genome * stride * (P.use64 ? 8 : 4);
  uint32_t* ocnt = cnt + sg.genome;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  // this wave's queues: stage-1 survivors as (E, position) pairs, stage-2 survivors as positions
  const lds_q1_ptr q1 = (lds_q1_ptr)(uintptr_t)(BLOOM_BYTES + wv * Q1_CAP * 8);
  const lds_u32_ptr wq = (lds_u32_ptr)(uintptr_t)(BLOOM_BYTES + Q1_BYTES + wv * BQ_CAP * (int)sizeof(bq_t));
  uint32_t q1n = 0, qn = 0;                                                        // wave-uniform
  BloomSeg bs{sg.g_begin, sg.g_end, sg.s_begin, sg.s_end, 0};  // base: set whenever both queues are empty
  int cq = 0;                                                 // the chunk the queues' base points at

  // this wave's chunks [c, c1) of the segment's 1 KiB-aligned span (a segment is a genome or a piece of one: < 2^31 chunks)
  const int64_t A0 = (int64_t)(sg.s_begin & ~(uint64_t)(CHUNK - 1));
  const int64_t NC = ((int64_t)sg.s_end - A0 + CHUNK - 1) / CHUNK;
  int c = (int)(NC * wv / (WGB / 64));
  const int c1 = (int)(NC * (wv + 1) / (WGB / 64));
  // chunks [in_lo, in_hi) lie inside the genome: one 16-byte load per lane; the others (a genome's first and last) byte by byte
  const int64_t lo64 = ((int64_t)sg.g_begin - A0 + CHUNK - 1) >> 10, hi64 = ((int64_t)sg.g_end - A0) >> 10;  // CHUNK = 2^10
  const int in_lo = P.nofast ? 0x7fffffff : (int)(lo64 < -1 ? -1 : lo64), in_hi = (int)(hi64 > 0x7fffffff ? 0x7fffffff : hi64);
  const uint8_t* lane_seq = seq + A0 + 16 * (int64_t)lane;

  auto fetch = [&](int ci) -> uint4 {  // lane's 16 bases of chunk ci ('N' outside the genome)
    const bool inside = ci >= in_lo && ci < in_hi;  // wave-uniform
    return inside ? *reinterpret_cast<const uint4*>(lane_seq + (int64_t)ci * CHUNK)
                  : load_bases16_edge(seq, A0 + (int64_t)ci * CHUNK + 16 * (int64_t)lane, sg.g_begin, sg.g_end);
  };
  // 16 bases -> 32 bits, first base on top, in the FILTERS' alphabet: bits 1 and 2 of the character as they are (A 0, C 1,
  // T 2, G 3 -- a Gray code of BaseMap's A 0, C 1, G 2, T 3, :1007-1017), which saves the xor that turns them into
  // BaseMap's codes: one v_and and one v_dot4 per dword (the dot product of 2 x code with the weights 64, 16, 4, 1 is twice
  // the packed byte).  The host builds the map and the Bloom filter over the same alphabet (gray24); the exact drain reads
  // the characters again and works in the reference's encoding.
  auto pack16 = [&](const uint4 d) -> uint32_t {
    const uint32_t w[4] = {d.x, d.y, d.z, d.w};
    uint32_t pk[4];
#pragma unroll
    for (int qd = 0; qd < 4; qd++) pk[qd] = __builtin_amdgcn_udot4(w[qd] & 0x06060606u, 0x01041040u, 0u, false);
    return (pk[0] << 23) | (pk[1] << 15) | (pk[2] << 7) | (pk[3] >> 1);
  };
  auto word_of = [&](int qd, uint32_t W, uint32_t Wp, uint32_t Wpp) -> uint32_t {  // E of the lane's dword qd (0..3, a constant)
    if (NARROW) return qd ? __builtin_amdgcn_alignbit(Wp, W, 32 - 8 * qd) : Wp;
    const int sft = 24 - 8 * qd + DS;  // bits of (Wpp : Wp : W) below E
    return sft == 0 ? W : sft < 32 ? __builtin_amdgcn_alignbit(Wp, W, sft) : sft == 32 ? Wp : __builtin_amdgcn_alignbit(Wpp, Wp, sft - 32);
  };
  auto map_word = [&](uint32_t E) -> uint32_t {  // stage 1: the 18 bits all four fields share, E[FO0 .. FO0 + 18), against the map
    return *(const RTC_LDS uint32_t*)(uintptr_t)((E >> (FO0 + 3)) & 0x7ffcu);
  };

  uint32_t carry1 = 0, carry2 = 0;  // the packed bases of lanes 63 / 62 of the previous chunk (SGPRs)
  bool primed = false;
  int careful = -1;                 // >= 0: chunk c did not fit the stage-1 queue; its dword `careful` is next, one per round
  for (;;) {
    // The one place where the queues are served (and where the pipeline (re)starts): a full batch of the exact-drain
    // queue first (64 entries from its end: every lane busy, a drain costs the same for 3 entries as for 64), then a full
    // batch of stage-1 survivors through stage 2; at the end of the wave's stretch, and when the queues' base is too far
    // behind, also what is left of either.
    const bool flush = c >= c1 || c - cq >= BQ_SPAN;
    for (;;) {
      if (qn >= 64 || (flush && qn && q1n == 0)) {
        const uint32_t n = qn < 64 ? qn : 64;
        bloom_drain<K>(seq, bs, P, g_bk, g_rank, var, wq, qn - n, n, lane, orow, ocnt, stride);
        qn -= n;
      } else if (q1n >= 64 || (flush && q1n)) {
        const uint32_t n = q1n < 64 ? q1n : 64;
        q1n -= n;
        qn = bloom_stage2<K>(q1 + q1n, n, lane, wq, qn, bs);
      } else {
        break;
      }
    }
    if (c >= c1) break;
    if (qn == 0 && q1n == 0) { cq = c; bs.base = (uint64_t)(A0 + (int64_t)c * CHUNK); }  // empty queues: their base moves up to here
    if (careful >= 0) {
      // One dword of chunk c per round (q1n, qn < 64 here; a dword adds at most 64): the chunk and the one in front
      // of it are read again.  Rare by construction -- long stretches whose 9-mers all occur in kept 12-mers.
      const uint32_t Wb = pack16(fetch(c - 1));
      const uint32_t W = pack16(fetch(c));
      const uint32_t Wp = from_lane_below(W, __builtin_amdgcn_readlane(Wb, 63));
      const uint32_t Wpp = from_lane_below(Wp, __builtin_amdgcn_readlane(Wb, 62));
      const uint32_t e4[4] = {word_of(0, W, Wp, Wpp), word_of(1, W, Wp, Wpp), word_of(2, W, Wp, Wpp), word_of(3, W, Wp, Wpp)};
      const uint32_t E = careful == 0 ? e4[0] : careful == 1 ? e4[1] : careful == 2 ? e4[2] : e4[3];
      const bool pass = ((map_word(E) >> ((E >> FO0) & 31u)) & 1u) != 0u;
      const uint64_t bal = __ballot(pass);
      if (pass) {
        const uint32_t at = q1n + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
        u32x2 ent;
        ent.x = E;
        ent.y = ((uint32_t)(c - cq) << 8) + 4u * lane + (uint32_t)careful;
        q1[at] = ent;
      }
      q1n += (uint32_t)__popcll(bal);
      if (++careful == 4) {
        careful = -1;
        carry1 = __builtin_amdgcn_readlane(W, 63);
        carry2 = __builtin_amdgcn_readlane(W, 62);
        primed = true;
        c++;
      }
      continue;
    }
    if (!primed) {  // the bases in front of the first chunk
      const uint32_t Wb = pack16(fetch(c - 1));
      carry1 = __builtin_amdgcn_readlane(Wb, 63);
      carry2 = __builtin_amdgcn_readlane(Wb, 62);
      primed = true;
    }
    uint4 D[AHEAD];
#pragma unroll
    for (int j = 0; j < AHEAD; j++) D[j] = c + j < c1 ? fetch(c + j) : make_uint4(0u, 0u, 0u, 0u);
    bool stop = false;
    while (!stop) {
#pragma unroll
      for (int j = 0; j < AHEAD; j++) {
        if (c >= c1 || qn >= 64u || c - cq >= BQ_SPAN) { stop = true; break; }  // done, or a full exact-drain batch waits
        const uint32_t W = pack16(D[j]);
        if (c + AHEAD < c1) {
          __builtin_amdgcn_sched_barrier(0);  // the request stays here (hoisted, its registers would pile up)
          D[j] = fetch(c + AHEAD);
        }
        const uint32_t Wp = from_lane_below(W, carry1);
        uint32_t Wpp = 0;
        if (!NARROW) Wpp = from_lane_below(Wp, carry2);
        const uint32_t rel0 = ((uint32_t)(c - cq) << 8) + 4u * lane;  // queue entry of the lane's first dword: dwords from the base
        // stage 1 for the lane's four dwords side by side: the four words E, their four map reads in flight together,
        // four votes; the survivors of all four go to the queue with ONE update of its length
        uint32_t E[4], mw[4];
        uint64_t bal[4];
        uint32_t pv[4];
#pragma unroll
        for (int qd = 0; qd < 4; qd++) { E[qd] = word_of(qd, W, Wp, Wpp); mw[qd] = map_word(E[qd]); }
        uint32_t total = 0;
#pragma unroll
        for (int qd = 0; qd < 4; qd++) {
          pv[qd] = __builtin_amdgcn_ubfe(mw[qd], E[qd] >> FO0, 1u);  // the map's bit, 0 or 1 (the offset's low five bits count)
          bal[qd] = __ballot(pv[qd] != 0u);
          total += (uint32_t)__popcll(bal[qd]);
        }
        if (q1n + total >= (uint32_t)Q1_CAP) { careful = 0; stop = true; break; }  // (the carries still describe chunk c - 1)
        carry1 = __builtin_amdgcn_readlane(W, 63);
        if (!NARROW) carry2 = __builtin_amdgcn_readlane(W, 62);
#pragma unroll
        for (int qd = 0; qd < 4; qd++) {
          if (pv[qd] != 0u) {
            const uint32_t at = q1n + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal[qd] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal[qd], 0u));
            u32x2 ent;
            ent.x = E[qd];
            ent.y = rel0 + qd;
            q1[at] = ent;
          }
          q1n += (uint32_t)__popcll(bal[qd]);
        }
        if (q1n >= 64) {  // wave-uniform: one full batch for stage 2 from the end of the queue (q1n < 128, qn < 64 here)
          q1n -= 64;
          qn = bloom_stage2<K>(q1 + q1n, 64, lane, wq, qn, bs);
        }
        c++;
      }
    }
  }
}

// ---- the same prefilter over the command lines' 2-bit staging format (rtc_unpack.hip: base i at bits 2 (i & 3) of byte
// i >> 2, BaseMap codes; everything that is not ACGT -- N, IUPAC codes, record separators, the gaps between genomes --
// listed as runs (start, length), ascending and disjoint) ------------------------------------------------------------------
// The batch is sketched as it crossed PCIe: 0.25 B per base read once, no ASCII copy in HBM, no character decode.  A lane
// holds 64 bases (one 16-byte load: a wave takes 4 096 bases per step); its four words are turned into the filters'
// alphabet and order with five instructions each (v_bfrev puts the first base on top, the Gray code of a code b is
// b ^ (b >> 1)), the bases in front of a word come from the lane's own previous word or, for its first, from the
// neighbour lane (DPP).  Stage 1, the queues and stage 2 are those of the ASCII kernel.  The exact drain takes the
// K + 3 codes of a queued dword from the packed stream and asks the run list whether a k-mer's window touches a run
// (the last run that starts at or before the k-mer's end must have ended before its first base): a k-mer is valid in
// the reference (:1136-1139, :1160-1164) exactly when none of its K characters is outside ACGT.
struct PackedBatch {
  const uint8_t* bytes;      // packed bases
  uint64_t n_bases;          // bases the buffer holds (a multiple of 64)
  const uint64_t* runs;      // (start, length) pairs
  const uint2* seg_runs;     // per segment: the runs [x, y) that can touch a k-mer the segment owns (packed_seg_runs_kernel)
};

// per segment, the first run that ends behind s_begin - (K - 1) and the first that starts at or behind s_end: the exact
// drain looks a window up among these only (none for most segments of a finished genome)
__global__ __launch_bounds__(256) void packed_seg_runs_kernel(const KSegment* __restrict__ segs, uint32_t nseg, const uint64_t* __restrict__ runs,
                                                              uint32_t n_runs, int K, uint2* __restrict__ seg_runs) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nseg) return;
  const int64_t first = (int64_t)segs[s].s_begin - (K - 1), end = (int64_t)segs[s].s_end;
  uint32_t lo = 0, hi = n_runs;  // runs that end at or before `first` (ends ascend with the starts: the runs are disjoint)
  while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((int64_t)(runs[2 * (uint64_t)mid] + runs[2 * (uint64_t)mid + 1]) <= first) lo = mid + 1; else hi = mid; }
  const uint32_t x = lo;
  hi = n_runs;                   // runs that start before `end`
  while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((int64_t)runs[2 * (uint64_t)mid] < end) lo = mid + 1; else hi = mid; }
  seg_runs[s] = make_uint2(x, lo);
}

// is [first, last] free of the runs [rlo, rhi)?  (ascending by start, disjoint; the runs in front of rlo end at or before first)
__device__ __forceinline__ bool window_valid(const uint64_t* __restrict__ runs, uint32_t rlo, uint32_t rhi, int64_t first, int64_t last) {
  uint32_t lo = rlo, hi = rhi;  // -> the number of runs that start at or before `last`
  while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((int64_t)runs[2 * (uint64_t)mid] <= last) lo = mid + 1; else hi = mid; }
  if (lo == rlo) return true;
  return (int64_t)(runs[2 * (uint64_t)(lo - 1)] + runs[2 * (uint64_t)(lo - 1) + 1]) <= first;
}

// n <= 64 queued dwords wq[first .. first + n), one per lane.  The K + 3 codes that end the dword's four k-mers (at most 62
// bits) are taken from the stream again at their own alignment; the forward strand of all of them is the stream bit-reversed
// with the two bits of every base put back in order, the reverse complement is the stream inverted as it lies (the reference
// appends complements from the top, :1135: a k-mer's first base ends up lowest).  What the characters were is the run list's
// business: a lane whose whole stretch is clear of runs (nearly all) asks once, the others once per k-mer.
template <int K>
__device__ __forceinline__ void packed_drain(const PackedBatch& B, uint2 sr, const BloomSeg& sg, const KssdParams& P,
                                             const uint32_t* __restrict__ g_bk, const uint16_t* __restrict__ g_rank, int var,
                                             lds_u32_ptr wq, uint32_t first, uint32_t n, uint32_t lane, void* orow,
                                             uint32_t* ocnt, uint32_t stride) {
  constexpr int NB = K + 3;  // bases walked: the first k-mer's first .. the last k-mer's last
  const bool have = lane < n;
  const int64_t q0 = (int64_t)sg.base + (have ? 4 * (int64_t)wq[first + lane] : 0);  // first base of the dword
  const int64_t b0 = q0 - (K - 1);                                                    // first base of its first k-mer
  const int64_t s0 = b0 < 0 ? 0 : b0;
  uint32_t w[3] = {0u, 0u, 0u};
  if (have) {
    const uint64_t nbytes = B.n_bases / 4;
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const uint64_t byte = ((uint64_t)(s0 >> 4) + j) * 4;
      if (byte + 4 <= nbytes) w[j] = *reinterpret_cast<const uint32_t*>(B.bytes + byte);
    }
  }
  const uint32_t sh = 2u * (uint32_t)(s0 & 15);  // bits in front of base s0 in w[0]
  uint64_t X = ((uint64_t)__builtin_amdgcn_alignbit(w[2], w[1], sh) << 32) | __builtin_amdgcn_alignbit(w[1], w[0], sh);  // base s0 + t at bits 2 t
  if (b0 < 0) X = b0 > -4 ? X << (2 * (uint32_t)(-b0)) : 0;  // (k-mers that begin in front of the batch are never owned)
  const uint64_t Y = __brevll(X);
  const uint64_t F = ((Y & 0x5555555555555555ull) << 1) | ((Y >> 1) & 0x5555555555555555ull);  // base t at bits [62 - 2 t, 64 - 2 t)
  const uint64_t R = ~X;
  const bool clean = have && window_valid(B.runs, sr.x, sr.y, b0, b0 + NB - 1);
#pragma unroll
  for (int b = 0; b < 4; b++) {
    const int64_t pos = q0 + b;  // the k-mer of bases b .. b + K - 1 of the stretch ends here
    bool ok = have && pos >= (int64_t)sg.s_begin && pos < (int64_t)sg.s_end && pos - (K - 1) >= (int64_t)sg.g_begin && pos < (int64_t)sg.g_end;
    if (ok && !clean) ok = window_valid(B.runs, sr.x, sr.y, pos - (K - 1), pos);                  // :1136-1139, :1160-1164
    const uint64_t tuple = (F >> (64 - 2 * (b + K))) & P.tupmask;                                 // :1134
    const uint64_t rvs = (R >> (2 * b)) & P.tupmask;                                              // :1135
    bloom_emit(ok, tuple < rvs ? tuple : rvs, P, g_bk, g_rank, var, lane, orow, ocnt, stride);   // :1141
  }
}

// 16 packed bases (first base in the low bits, BaseMap codes) -> first base on top, Gray codes: the filters' word
__device__ __forceinline__ uint32_t packed_word(uint32_t x) {
  const uint32_t y = __brev(x);  // pairs in order, the two bits of a pair swapped: (lo, hi)
  return ((y & 0x55555555u) << 1) | (((y >> 1) ^ y) & 0x55555555u);  // (hi, lo ^ hi)
}

constexpr int PCHUNK = 4096;   // bases a wave takes per step of the packed kernel (64 lanes x 64)
constexpr int PQ_SPAN = 63;    // chunks on one queue base: 63 * 1024 + 1023 dwords < 2^16

template <int K>
__global__ __launch_bounds__(WGB, WGB_WAVES_EU) void sketch_kssd_packed_kernel(PackedBatch B, const KSegment* __restrict__ segs, KssdParams P,
                                                               const uint32_t* __restrict__ g_bloom, const uint32_t* __restrict__ g_bk,
                                                               const uint16_t* __restrict__ g_rank, int var, void* __restrict__ out,
                                                               uint32_t stride, uint32_t* __restrict__ cnt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  static_assert(K % 2 == 0 && K >= 18 && K <= 28, "24-bit dim_id in the middle of at most 28 bases");
  constexpr int DS = K - 12;
  constexpr bool NARROW = DS >= 8 && DS <= 10;
  constexpr int FO0 = NARROW ? DS - 2 : 6;
  constexpr int AHEAD = 2;
  const KSegment sg = segs[blockIdx.x];
  const int t = threadIdx.x;
  const uint32_t lane = t & 63;
  {
    uint4* l4 = reinterpret_cast<uint4*>(smem);
    const uint4* g4 = reinterpret_cast<const uint4*>(g_bloom);
    for (int i = t; i < BLOOM_BYTES / 16; i += WGB) l4[i] = g4[i];
    __syncthreads();
  }
  if ((uint32_t)(uintptr_t)(RTC_LDS unsigned char*)smem != 0u) __builtin_trap();  // the filters are addressed absolutely
  void* orow = reinterpret_cast<unsigned char*>(out) + (uint64_t)sg.genome * stride * (P.use64 ? 8 : 4);
  uint32_t* ocnt = cnt + sg.genome;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const lds_q1_ptr q1 = (lds_q1_ptr)(uintptr_t)(BLOOM_BYTES + wv * Q1_CAP * 8);
  const lds_u32_ptr wq = (lds_u32_ptr)(uintptr_t)(BLOOM_BYTES + Q1_BYTES + wv * BQ_CAP * (int)sizeof(bq_t));
  uint32_t q1n = 0, qn = 0;
  BloomSeg bs{sg.g_begin, sg.g_end, sg.s_begin, sg.s_end, 0};
  const uint2 srange = B.seg_runs[blockIdx.x];
  int cq = 0;

  const int64_t A0 = (int64_t)(sg.s_begin & ~(uint64_t)(PCHUNK - 1));
  const int64_t NC = ((int64_t)sg.s_end - A0 + PCHUNK - 1) / PCHUNK;
  int c = (int)(NC * wv / (WGB / 64));
  const int c1 = (int)(NC * (wv + 1) / (WGB / 64));
  const int64_t nbytes = (int64_t)(B.n_bases / 4);

  auto fetch = [&](int ci) -> uint4 {  // the lane's 64 bases of chunk ci (zeros outside the buffer: candidates at most, dropped later)
    const int64_t byte = (A0 + (int64_t)ci * PCHUNK) / 4 + 16 * (int64_t)lane;
    if (byte < 0 || byte + 16 > nbytes) return make_uint4(0u, 0u, 0u, 0u);
    return *reinterpret_cast<const uint4*>(B.bytes + byte);
  };
  // the word E of dword qd of the lane's word j: G = the lane's four words, P3 / P2 = the last two words of the lane below
  auto word_of = [&](int j, int qd, const uint32_t* G, uint32_t P3, uint32_t P2) -> uint32_t {
    const uint32_t W = G[j], Wp = j >= 1 ? G[j - 1] : P3, Wpp = j >= 2 ? G[j - 2] : (j == 1 ? P3 : P2);
    if (NARROW) return qd ? __builtin_amdgcn_alignbit(Wp, W, 32 - 8 * qd) : Wp;
    const int sft = 24 - 8 * qd + DS;
    return sft == 0 ? W : sft < 32 ? __builtin_amdgcn_alignbit(Wp, W, sft) : sft == 32 ? Wp : __builtin_amdgcn_alignbit(Wpp, Wp, sft - 32);
  };
  auto map_word = [&](uint32_t E) -> uint32_t { return *(const RTC_LDS uint32_t*)(uintptr_t)((E >> (FO0 + 3)) & 0x7ffcu); };

  uint32_t carry3 = 0, carry2 = 0;  // words 3 and 2 of lane 63 of the previous chunk
  bool primed = false;
  int careful = -1;                 // >= 0: dword `careful` (0 .. 15 = 4 j + qd) of chunk c is next, one per round
  for (;;) {
    const bool flush = c >= c1 || c - cq >= PQ_SPAN;
    for (;;) {
      if (qn >= 64 || (flush && qn && q1n == 0)) {
        const uint32_t n = qn < 64 ? qn : 64;
        packed_drain<K>(B, srange, bs, P, g_bk, g_rank, var, wq, qn - n, n, lane, orow, ocnt, stride);
        qn -= n;
      } else if (q1n >= 64 || (flush && q1n)) {
        const uint32_t n = q1n < 64 ? q1n : 64;
        q1n -= n;
        qn = bloom_stage2<K>(q1 + q1n, n, lane, wq, qn, bs);
      } else {
        break;
      }
    }
    if (c >= c1) break;
    if (qn == 0 && q1n == 0) { cq = c; bs.base = (uint64_t)(A0 + (int64_t)c * PCHUNK); }
    if (careful >= 0) {
      const uint4 db = fetch(c - 1), dc = fetch(c);
      const uint32_t G[4] = {packed_word(dc.x), packed_word(dc.y), packed_word(dc.z), packed_word(dc.w)};
      const uint32_t b3 = packed_word(db.w), b2 = packed_word(db.z);
      const uint32_t P3 = from_lane_below(G[3], __builtin_amdgcn_readlane(b3, 63)), P2 = from_lane_below(G[2], __builtin_amdgcn_readlane(b2, 63));
      uint32_t E = 0;
#pragma unroll
      for (int i = 0; i < 16; i++) if (careful == i) E = word_of(i >> 2, i & 3, G, P3, P2);
      const bool pass = ((map_word(E) >> ((E >> FO0) & 31u)) & 1u) != 0u;
      const uint64_t bal = __ballot(pass);
      if (pass) {
        u32x2 ent;
        ent.x = E;
        ent.y = ((uint32_t)(c - cq) << 10) + 16u * lane + (uint32_t)careful;
        q1[q1n + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u))] = ent;
      }
      q1n += (uint32_t)__popcll(bal);
      if (++careful == 16) {
        careful = -1;
        carry3 = __builtin_amdgcn_readlane(G[3], 63);
        carry2 = __builtin_amdgcn_readlane(G[2], 63);
        primed = true;
        c++;
      }
      continue;
    }
    if (!primed) {
      const uint4 db = fetch(c - 1);
      carry3 = __builtin_amdgcn_readlane(packed_word(db.w), 63);
      carry2 = __builtin_amdgcn_readlane(packed_word(db.z), 63);
      primed = true;
    }
    uint4 D[AHEAD];
#pragma unroll
    for (int j = 0; j < AHEAD; j++) D[j] = c + j < c1 ? fetch(c + j) : make_uint4(0u, 0u, 0u, 0u);
    bool stop = false;
    while (!stop) {
#pragma unroll
      for (int u = 0; u < AHEAD; u++) {
        if (c >= c1 || qn >= 64u || c - cq >= PQ_SPAN) { stop = true; break; }
        const uint32_t G[4] = {packed_word(D[u].x), packed_word(D[u].y), packed_word(D[u].z), packed_word(D[u].w)};
        if (c + AHEAD < c1) {
          __builtin_amdgcn_sched_barrier(0);
          D[u] = fetch(c + AHEAD);
        }
        const uint32_t P3 = from_lane_below(G[3], carry3);
        uint32_t P2 = 0;
        if (!NARROW) P2 = from_lane_below(G[2], carry2);
        const uint32_t relc = ((uint32_t)(c - cq) << 10) + 16u * lane;
        bool spill = false;  // wave-uniform: a word's survivors would not fit the stage-1 queue
#pragma unroll
        for (int j = 0; j < 4; j++) {
          uint32_t E[4], mw[4], pv[4];
          uint64_t bal[4];
#pragma unroll
          for (int qd = 0; qd < 4; qd++) { E[qd] = word_of(j, qd, G, P3, P2); mw[qd] = map_word(E[qd]); }
          uint32_t total = 0;
#pragma unroll
          for (int qd = 0; qd < 4; qd++) {
            pv[qd] = __builtin_amdgcn_ubfe(mw[qd], E[qd] >> FO0, 1u);
            bal[qd] = __ballot(pv[qd] != 0u);
            total += (uint32_t)__popcll(bal[qd]);
          }
          if (q1n + total >= (uint32_t)Q1_CAP) { careful = 4 * j; spill = true; break; }  // dwords 0 .. 4 j - 1 of the chunk are queued
#pragma unroll
          for (int qd = 0; qd < 4; qd++) {
            if (pv[qd] != 0u) {
              u32x2 ent;
              ent.x = E[qd];
              ent.y = relc + 4 * j + qd;
              q1[q1n + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal[qd] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal[qd], 0u))] = ent;
            }
            q1n += (uint32_t)__popcll(bal[qd]);
          }
          if (q1n >= 64) {
            if (qn >= 64u) { careful = 4 * j + 4; spill = true; break; }  // the exact-drain queue is served first: the rest of the chunk one dword per round
            q1n -= 64;
            qn = bloom_stage2<K>(q1 + q1n, 64, lane, wq, qn, bs);
          }
        }
        if (spill) {
          if (careful >= 16) {  // the chunk was complete after all
            careful = -1;
            carry3 = __builtin_amdgcn_readlane(G[3], 63);
            carry2 = __builtin_amdgcn_readlane(G[2], 63);
            c++;
          }
          stop = true;
          break;
        }
        carry3 = __builtin_amdgcn_readlane(G[3], 63);
        if (!NARROW) carry2 = __builtin_amdgcn_readlane(G[2], 63);
        c++;
      }
    }
  }
}

// One WAVE per genome for rows of at most 64 * NPER tuples (a 2 Mbp genome at drlevel 3 yields ~490, a 5 Mbp one ~1 200):
// element e = 64 r + lane sits in register r of its lane, the bitonic network exchanges across lanes with a wave shuffle
// while the distance is below 64 and between registers of one lane above -- no LDS, no workgroup barrier (the LDS
// kernel below pays 45 of them for 512 tuples; it keeps the longer rows).  Dedup and compaction run register by register
// (ascending e), one ballot each.
template <typename OutT, int NPER>
__device__ __forceinline__ void wave_sort_unique(OutT* __restrict__ row, uint32_t m, uint32_t lane, uint32_t* __restrict__ cnt_out) {
  constexpr int N = 64 * NPER;
  OutT v[NPER];
#pragma unroll
  for (int r = 0; r < NPER; r++) { const uint32_t e = 64u * r + lane; v[r] = e < m ? row[e] : (OutT)~(OutT)0; }
#pragma unroll
  for (int k = 2; k <= N; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j >= 64) {  // partner in another register of this lane
#pragma unroll
        for (int r = 0; r < NPER; r++) {
          const int pr = r ^ (j >> 6);
          if (pr > r) {
            const bool up = ((64 * r) & k) == 0;  // k >= 128 here: the direction is the register's
            const OutT a = v[r], b = v[pr];
            const bool sw = (a > b) == up;
            v[r] = sw ? b : a;
            v[pr] = sw ? a : b;
          }
        }
      } else {        // partner in lane ^ j, same register
#pragma unroll
        for (int r = 0; r < NPER; r++) {
          OutT o;
          if constexpr (sizeof(OutT) == 8) {
            const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v[r], j), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v[r] >> 32), j);
            o = ((OutT)hi << 32) | lo;
          } else {
            o = (OutT)__shfl_xor((int)v[r], j);
          }
          const uint32_t e = 64u * r + lane;
          const bool up = (e & (uint32_t)k) == 0, lower = (lane & (uint32_t)j) == 0;
          const OutT mn = v[r] < o ? v[r] : o, mx = v[r] < o ? o : v[r];
          v[r] = (lower == up) ? mn : mx;
        }
      }
    }
  }
  // the first m sorted entries are exactly the real ones (padding is the maximum value): distinct values to the row's front
  uint32_t written = 0;  // wave-uniform
  OutT last = 0;         // the value in front of register r's lane 0: lane 63 of register r - 1
#pragma unroll
  for (int r = 0; r < NPER; r++) {
    OutT prev;
    if constexpr (sizeof(OutT) == 8) {
      const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)v[r], 1), hi = (uint32_t)__shfl_up((int)(uint32_t)(v[r] >> 32), 1);
      prev = ((OutT)hi << 32) | lo;
    } else {
      prev = (OutT)__shfl_up((int)v[r], 1);
    }
    if (lane == 0) prev = last;
    const uint32_t e = 64u * r + lane;
    const bool keep = e < m && (e == 0 || v[r] != prev);
    const uint64_t bal = __ballot(keep);
    // every read of the row happened before the sort: writing in place is safe
    if (keep) row[written + (uint32_t)__popcll(bal & ((1ULL << lane) - 1ULL))] = v[r];
    written += (uint32_t)__popcll(bal);
    if constexpr (sizeof(OutT) == 8) {
      last = ((OutT)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v[r] >> 32), 63) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v[r], 63);
    } else {
      last = (OutT)__builtin_amdgcn_readlane((int)v[r], 63);
    }
  }
  if (lane == 0) *cnt_out = written;
}

constexpr int WAVE_SORT_MAX = 1024;  // tuples a wave sorts in registers (16 per lane)
template <typename OutT>
__global__ __launch_bounds__(256) void kssd_sort_unique_wave_kernel(OutT* __restrict__ out, uint32_t stride, uint32_t* __restrict__ cnt, uint32_t n) {
  const uint32_t g = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= n) return;
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t m = min(cnt[g], stride);
  if (m == 0 || m > (uint32_t)WAVE_SORT_MAX) return;
  OutT* row = out + (uint64_t)g * stride;
  if (m <= 256) wave_sort_unique<OutT, 4>(row, m, lane, cnt + g);
  else if (m <= 512) wave_sort_unique<OutT, 8>(row, m, lane, cnt + g);
  else wave_sort_unique<OutT, 16>(row, m, lane, cnt + g);
}

// one workgroup per genome: sort + dedup the appended tuples in LDS (hashArr sort :1185,:1192)
template <typename OutT>
__global__ __launch_bounds__(WG) void kssd_sort_unique_kernel(OutT* __restrict__ out, uint32_t stride,
                                                              uint32_t* __restrict__ cnt, int cap, int wave_max) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  OutT* buf = reinterpret_cast<OutT*>(smem);
  uint32_t* wave_tot = reinterpret_cast<uint32_t*>(smem + (size_t)cap * sizeof(OutT));  // [WG/64]
  uint32_t& scan_base = wave_tot[WG / 64];
  const uint32_t g = blockIdx.x;
  const int t = threadIdx.x;
  const uint32_t m = min(cnt[g], stride);
  if (m <= (uint32_t)wave_max || m > (uint32_t)cap) return;  // short rows: kssd_sort_unique_wave_kernel; rows beyond one LDS buffer: kssd_big_* below
  OutT* row = out + (uint64_t)g * stride;
  int n2 = 64;  // bitonic network over the next power of two (a 2 Mbp genome at drlevel 3: ~490 tuples -> 512)
  while (n2 < (int)m) n2 <<= 1;
  for (int i = t; i < n2; i += WG) buf[i] = i < (int)m ? row[i] : (OutT)~(OutT)0;
  if (t == 0) scan_base = 0;
  __syncthreads();
  for (int k = 2; k <= n2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = t; i < (n2 >> 1); i += WG) {
        const int a = ((i & ~(j - 1)) << 1) | (i & (j - 1));
        const int b = a | j;
        const bool up = (a & k) == 0;
        const OutT va = buf[a], vb = buf[b];
        if ((va > vb) == up) { buf[a] = vb; buf[b] = va; }
      }
      __syncthreads();
    }
  }
  // the first m sorted entries are exactly the real ones (padding is the maximum value)
  const uint32_t lane = t & 63, wave = t >> 6;
  for (int r = 0; r < n2; r += WG) {
    const int idx = r + t;
    const OutT v = buf[idx];
    const bool keep = idx < (int)m && (idx == 0 || v != buf[idx - 1]);
    const uint64_t bal = __ballot(keep);
    if (lane == 0) wave_tot[wave] = (uint32_t)__popcll(bal);
    __syncthreads();
    const uint32_t sb = scan_base;
    uint32_t base = sb, total = 0;
#pragma unroll
    for (int w = 0; w < WG / 64; w++) { const uint32_t wt = wave_tot[w]; if ((uint32_t)w < wave) base += wt; total += wt; }
    if (keep) row[base + (uint32_t)__popcll(bal & ((1ULL << lane) - 1ULL))] = v;
    __syncthreads();
    if (t == 0) scan_base = sb + total;
  }
  __syncthreads();
  if (t == 0) cnt[g] = scan_base;
}

// ---- rows with more tuples than one LDS buffer: global merge sort ------------------------------
// (a 100 Mbp genome at drlevel 3 yields ~24 000 tuples, a 3 Gbp one ~730 000.)  Chunks of `cap`
// tuples are sorted in LDS, then log2(#chunks) merge passes ping-pong between the output row and a
// scratch row: every workgroup produces one 2048-element tile of a merged run pair, found with two
// merge-path searches, and the final pass is followed by a streaming dedup back into the row.
struct BigRow {
  uint32_t genome, count;
  uint64_t tmp_off;          // element offset of this row in the scratch buffer
  uint32_t chunk0, tile0;    // first chunk / tile index of this row in the flattened grids
};
constexpr int BIG_TILE = 2048;

__device__ __forceinline__ int find_row(const BigRow* rows, int nrows, uint32_t idx, bool by_tile) {
  int lo = 0, hi = nrows - 1;
  while (lo < hi) {  // last row whose first chunk/tile index is <= idx
    const int mid = (lo + hi + 1) >> 1;
    if ((by_tile ? rows[mid].tile0 : rows[mid].chunk0) <= idx) lo = mid; else hi = mid - 1;
  }
  return lo;
}

template <typename OutT>
__device__ __forceinline__ void bitonic_lds(OutT* buf, int n2) {
  const int t = threadIdx.x;
  for (int k = 2; k <= n2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = t; i < (n2 >> 1); i += WG) {
        const int a = ((i & ~(j - 1)) << 1) | (i & (j - 1));
        const int b = a | j;
        const bool up = (a & k) == 0;
        const OutT va = buf[a], vb = buf[b];
        if ((va > vb) == up) { buf[a] = vb; buf[b] = va; }
      }
      __syncthreads();
    }
  }
}

template <typename OutT>
__global__ __launch_bounds__(WG) void kssd_big_chunk_sort_kernel(OutT* __restrict__ out, uint32_t stride,
                                                                 const BigRow* __restrict__ rows, int nrows, int cap) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  OutT* buf = reinterpret_cast<OutT*>(smem);
  const int r = find_row(rows, nrows, blockIdx.x, false);
  const BigRow br = rows[r];
  const uint32_t c0 = (blockIdx.x - br.chunk0) * (uint32_t)cap;
  const uint32_t len = min((uint32_t)cap, br.count - c0);
  OutT* row = out + (uint64_t)br.genome * stride + c0;
  const int t = threadIdx.x;
  for (int i = t; i < cap; i += WG) buf[i] = i < (int)len ? row[i] : (OutT)~(OutT)0;
  __syncthreads();
  bitonic_lds(buf, cap);
  for (int i = t; i < (int)len; i += WG) row[i] = buf[i];  // real entries sort before the padding (ties are equal values)
}

// merged position d of runs A[0..la) and B[0..lb): how many of the first d outputs come from A (ties: A first)
template <typename OutT>
__device__ __forceinline__ uint32_t merge_path(const OutT* A, uint32_t la, const OutT* B, uint32_t lb, uint32_t d) {
  uint32_t lo = d > lb ? d - lb : 0, hi = d < la ? d : la;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (A[mid] <= B[d - 1 - mid]) lo = mid + 1; else hi = mid;
  }
  return lo;
}

template <typename OutT>
__global__ __launch_bounds__(WG) void kssd_big_merge_pass_kernel(OutT* out, uint32_t stride, OutT* tmp,
                                                                 const BigRow* __restrict__ rows, int nrows,
                                                                 uint32_t w, int src_is_tmp) {
  __shared__ OutT buf[BIG_TILE];
  __shared__ uint32_t cut[2];
  const int r = find_row(rows, nrows, blockIdx.x, true);
  const BigRow br = rows[r];
  OutT* rowp = out + (uint64_t)br.genome * stride;
  OutT* tmpp = tmp + br.tmp_off;
  const OutT* src = src_is_tmp ? tmpp : rowp;
  OutT* dst = src_is_tmp ? rowp : tmpp;
  const uint32_t c = br.count;
  const uint32_t o0 = (blockIdx.x - br.tile0) * (uint32_t)BIG_TILE;
  const uint32_t o1 = min(c, o0 + (uint32_t)BIG_TILE);
  const uint32_t ps = o0 / (2 * w) * (2 * w);        // w is a multiple of BIG_TILE: a tile never straddles run pairs
  const uint32_t am = min(c, ps + w), bm = min(c, ps + 2 * w);
  const OutT* A = src + ps; const uint32_t la = am - ps;
  const OutT* B = src + am; const uint32_t lb = bm - am;
  const int t = threadIdx.x;
  if (t < 2) cut[t] = merge_path(A, la, B, lb, (t ? o1 : o0) - ps);
  __syncthreads();
  const uint32_t a0 = cut[0], a1 = cut[1];
  const uint32_t b0 = (o0 - ps) - a0, b1 = (o1 - ps) - a1;
  const uint32_t na = a1 - a0, nb = b1 - b0;  // na + nb == o1 - o0
  for (uint32_t i = t; i < (uint32_t)BIG_TILE; i += WG)
    buf[i] = i < na ? A[a0 + i] : (i < na + nb ? B[b0 + (i - na)] : (OutT)~(OutT)0);
  __syncthreads();
  bitonic_lds(buf, BIG_TILE);
  for (uint32_t i = t; i < o1 - o0; i += WG) dst[o0 + i] = buf[i];
}

// sorted src (row itself or scratch) -> distinct values at the front of the row; cnt[genome] = #distinct
template <typename OutT>
__global__ __launch_bounds__(WG) void kssd_big_dedup_kernel(OutT* out, uint32_t stride, const OutT* tmp,
                                                            const BigRow* __restrict__ rows, uint32_t* __restrict__ cnt,
                                                            int src_is_tmp) {
  __shared__ uint32_t wave_tot[WG / 64];
  __shared__ OutT carry[2];
  const BigRow br = rows[blockIdx.x];
  OutT* row = out + (uint64_t)br.genome * stride;
  const OutT* src = src_is_tmp ? tmp + br.tmp_off : row;
  const int t = threadIdx.x;
  const uint32_t lane = t & 63, wave = t >> 6;
  uint32_t written = 0;  // identical in every thread
  for (uint32_t r0 = 0; r0 < br.count; r0 += WG) {
    const uint32_t idx = r0 + t;
    const bool in = idx < br.count;
    const OutT v = in ? src[idx] : (OutT)0;
    // the predecessor of a round's first element was saved before the previous round's writes
    const OutT prev = (in && idx > 0) ? (t == 0 ? carry[(r0 / WG) & 1] : src[idx - 1]) : (OutT)0;
    const bool keep = in && (idx == 0 || v != prev);
    const uint64_t bal = __ballot(keep);
    if (lane == 0) wave_tot[wave] = (uint32_t)__popcll(bal);
    if (t == WG - 1) carry[((r0 / WG) + 1) & 1] = v;
    __syncthreads();  // every read of this round is done before any write (dst <= src positions)
    uint32_t base = written, total = 0;
#pragma unroll
    for (int wv = 0; wv < WG / 64; wv++) { const uint32_t wt = wave_tot[wv]; if ((uint32_t)wv < wave) base += wt; total += wt; }
    if (keep) row[base + (uint32_t)__popcll(bal & ((1ULL << lane) - 1ULL))] = v;
    written += total;
    __syncthreads();
  }
  if (t == 0) cnt[br.genome] = written;
}

__global__ void max_u32_kernel(const uint32_t* __restrict__ a, uint32_t n, uint32_t* __restrict__ out_max) {
  uint32_t v = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) v = max(v, a[i]);
  for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, o));
  if ((threadIdx.x & 63) == 0 && v) atomicMax(out_max, v);
}

uint64_t table_checksum(const int32_t* t, size_t n) {
  uint64_t h = 1469598103934665603ULL;
  const size_t step = n > 4096 ? n / 4096 : 1;
  for (size_t i = 0; i < n; i += step) { h ^= (uint32_t)t[i]; h *= 1099511628211ULL; }
  return h ^ n;
}

}  // namespace

// rows with more than `cap` appended tuples: chunk sort + merge passes + dedup (see kssd_big_* kernels)
static int kssd_sort_big_rows(rtc_ctx* ctx, void* d_out, uint32_t stride, uint32_t* d_cnt, uint32_t n, int use64, int cap) {
  std::vector<uint32_t> h_cnt(n);
  RTC_HIP(ctx, hipMemcpyAsync(h_cnt.data(), d_cnt, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
  RTC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  std::vector<BigRow> rows;
  uint64_t tmp_elems = 0;
  uint32_t chunks = 0, tiles = 0, cmax = 0;
  for (uint32_t g = 0; g < n; g++) {
    if (h_cnt[g] <= (uint32_t)cap) continue;
    rows.push_back(BigRow{g, h_cnt[g], tmp_elems, chunks, tiles});
    tmp_elems += h_cnt[g];
    chunks += (h_cnt[g] + cap - 1) / cap;
    tiles += (h_cnt[g] + BIG_TILE - 1) / BIG_TILE;
    cmax = std::max(cmax, h_cnt[g]);
  }
  if (rows.empty()) return RTC_OK;
  const int w8 = use64 ? 8 : 4;
  void *ws1 = nullptr, *ws2 = nullptr, *hp = nullptr;
  RTC_TRY(rtc_ws(ctx, 1, tmp_elems * w8 + 64, &ws1));
  RTC_TRY(rtc_ws(ctx, 2, rows.size() * sizeof(BigRow) + 64, &ws2));
  RTC_TRY(rtc_pinned(ctx, rows.size() * sizeof(BigRow) + 64, &hp));
  memcpy(hp, rows.data(), rows.size() * sizeof(BigRow));
  RTC_HIP(ctx, hipMemcpyAsync(ws2, hp, rows.size() * sizeof(BigRow), hipMemcpyHostToDevice, ctx->stream));
  const BigRow* d_rows = (const BigRow*)ws2;
  const int nrows = (int)rows.size();
  const size_t lds_c = (size_t)cap * w8;
#define BIG_PIPELINE(OT)                                                                                              \
  do {                                                                                                                \
    auto ks = kssd_big_chunk_sort_kernel<OT>;                                                                         \
    RTC_HIP(ctx, hipFuncSetAttribute((const void*)ks, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c));       \
    hipLaunchKernelGGL(ks, dim3(chunks), dim3(WG), lds_c, ctx->stream, (OT*)d_out, stride, d_rows, nrows, cap);       \
    RTC_CHECK_LAUNCH(ctx);                                                                                            \
    int src_is_tmp = 0;                                                                                               \
    for (uint64_t w = (uint64_t)cap; w < cmax; w <<= 1) {                                                             \
      hipLaunchKernelGGL(kssd_big_merge_pass_kernel<OT>, dim3(tiles), dim3(WG), 0, ctx->stream, (OT*)d_out, stride,   \
                         (OT*)ws1, d_rows, nrows, (uint32_t)w, src_is_tmp);                                           \
      RTC_CHECK_LAUNCH(ctx);                                                                                          \
      src_is_tmp ^= 1;                                                                                                \
    }                                                                                                                 \
    hipLaunchKernelGGL(kssd_big_dedup_kernel<OT>, dim3(nrows), dim3(WG), 0, ctx->stream, (OT*)d_out, stride,          \
                       (const OT*)ws1, d_rows, d_cnt, src_is_tmp);                                                    \
    RTC_CHECK_LAUNCH(ctx);                                                                                            \
  } while (0)
  if (use64) BIG_PIPELINE(uint64_t); else BIG_PIPELINE(uint32_t);
#undef BIG_PIPELINE
  return RTC_OK;
}

// the batch in the staging format (rtc_sketch_kssd_packed_dev); d_seq then holds the packed bases
struct PackedArgs { uint64_t n_bases; const uint64_t* d_runs; uint64_t n_runs; };

static int sketch_kssd_impl(rtc_ctx* ctx, const uint8_t* d_seq, const PackedArgs* pk, const uint64_t* h_off, uint32_t n,
                            int kmer_size, int drlevel, const int32_t* h_shuffled_dim, void* d_out,
                            uint32_t stride, uint32_t* d_cnt, int* width_out, uint32_t* h_need) {
  if (!ctx || !h_off || !h_shuffled_dim || !width_out || (n && (!d_seq || !d_out || !d_cnt))) return RTC_ERR_ARG;
  if (kmer_size < 2 || kmer_size > 32) return rtc_fail(ctx, RTC_ERR_ARG, "kmer_size=%d outside 2..32", kmer_size);
  if (drlevel < 0 || drlevel > 8) return rtc_fail(ctx, RTC_ERR_ARG, "drlevel=%d outside 0..8", drlevel);
  RTC_HIP(ctx, hipSetDevice(ctx->device));
  ctx->sketch_gen++;  // sketches on this context change: memos keyed on a sketch buffer are stale
  // src/SketchInfo.cpp:1019-1048
  const int half_k = (kmer_size + 1) / 2;
  const int K = half_k * 2;
  const int use64 = half_k - drlevel > 8 ? 1 : 0;
  const int half_subk = 6 - drlevel >= 2 ? 6 : drlevel + 2;
  *width_out = use64 ? 8 : 4;
  if (half_subk > 7) return rtc_fail(ctx, RTC_ERR_UNSUPPORTED, "drlevel %d needs a 2^%d-entry shuffle table (the reference's int arithmetic overflows there too)", drlevel, 4 * half_subk);
  if (half_k < half_subk) return rtc_fail(ctx, RTC_ERR_UNSUPPORTED, "kmer_size %d too small for half_subk %d", kmer_size, half_subk);
  const int dim_size = 1 << (4 * half_subk);
  const int dim_end = 1 << (4 * (half_subk - drlevel));
  const int comp_bittl = 64 - 4 * half_k;
  const int half_outctx_len = half_k - half_subk;
  KssdParams P;
  P.K = K; P.drlevel = drlevel; P.use64 = use64;
  P.rev_add_move = 4 * half_k - 2;
  P.tupmask = 0xffffffffffffffffULL >> comp_bittl;
  P.domask = (P.tupmask >> (4 * half_outctx_len)) << (2 * half_outctx_len);
  const uint64_t undomask = (P.tupmask ^ P.domask) & P.tupmask;
  P.undomask1 = undomask & (P.tupmask >> ((half_k + half_subk) * 2));
  P.undomask0 = undomask ^ P.undomask1;
  P.dim_shift = 2 * half_outctx_len;
  P.und1_shl = K * 2 - half_outctx_len * 4;
  P.dimbits = 4 * half_subk;
  P.dim_end = dim_end;
  if (n == 0) return RTC_OK;
  if (((uintptr_t)d_seq & 15) != 0) return rtc_fail(ctx, RTC_ERR_ARG, "the sequence buffer must be 16-byte aligned");
  if (pk && ((pk->n_bases & 63) || pk->n_runs >> 32 || (pk->n_runs && !pk->d_runs)))
    return rtc_fail(ctx, RTC_ERR_ARG, "packed batch: n_bases must be a multiple of 64, fewer than 2^32 runs");
  RTC_HIP(ctx, hipSetDevice(ctx->device));
  if (pk) {  // the run list's contract, checked on the device beside the sketching; reported behind the capacity read-back below
    RTC_TRY(rtc_sticky_error(ctx));
    RTC_TRY(rtc_check_runs_async(ctx, pk->d_runs, pk->n_runs, pk->n_bases));
  }

  // ---- filter structures (cached in the context: one host thread per context, freed with it) ----
  bool lds_index = dim_end <= 4096 && 4 * half_subk <= 28;  // ranks fit 12 bits, entries 32
  auto& kc = ctx->kssd;
  const uint64_t cs = table_checksum(h_shuffled_dim, (size_t)dim_size);
  if (kc.half_subk != half_subk || kc.drlevel != drlevel || kc.checksum != cs) {
    RTC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (kc.d_index) { (void)hipFree(kc.d_index); kc.d_index = nullptr; }
    if (kc.d_table) { (void)hipFree(kc.d_table); kc.d_table = nullptr; }
    if (kc.d_bucket) { (void)hipFree(kc.d_bucket); kc.d_bucket = nullptr; }
    if (kc.d_bloom) { (void)hipFree(kc.d_bloom); kc.d_bloom = nullptr; }
    kc.bvar = -1;
    if (lds_index && 4 * half_subk == 24 && !ctx->opt.kssd_cuckoo) {
      // exact index of the kept ids: 8192 buckets of four 16-bit patterns (bucket = 13 bits of dim_id, the pattern
      // holds the other 11 and five the bucket implies -- an unused slot holds a pattern no key of its bucket can
      // produce) when no bucket receives more than four ids; two choices of bucket bits (exact_rank_global)
      std::vector<uint32_t> keys;
      for (int t = 0; t < dim_size; t++)
        if (h_shuffled_dim[t] >= 0 && h_shuffled_dim[t] < dim_end) keys.push_back((uint32_t)t);
      if (keys.size() > (size_t)MAX_LDS_KEEP) return rtc_fail(ctx, RTC_ERR_ARG, "shuffle table is not a permutation");
      const int order[2] = {1, 0};
      for (int oi = 0; oi < 2 && kc.bvar < 0; oi++) {
        const int var = order[oi];
        const size_t nsl = (size_t)BUCKET_BYTES / 2;
        std::vector<uint16_t> pat(nsl), rnk(nsl, 0);
        std::vector<uint8_t> fill(8192, 0);
        for (uint32_t bk = 0; bk < 8192; bk++)  // unused slots: implied bits inverted
          for (int sl = 0; sl < 4; sl++) pat[bk * 4 + sl] = (uint16_t)((~bk & 0x1fu) << (var ? 11 : 3));
        bool fits = true;
        for (uint32_t key : keys) {
          const uint32_t bk = var ? (key >> 11) & 0x1fffu : (key >> 3) & 0x1fffu;
          const uint16_t q = var ? (uint16_t)(key & 0xffffu) : (uint16_t)((key & 0xffu) | (((key >> 16) & 0xffu) << 8));
          if (fill[bk] == 4) { fits = false; break; }
          pat[bk * 4 + fill[bk]] = q;
          rnk[bk * 4 + fill[bk]] = (uint16_t)h_shuffled_dim[key];
          fill[bk]++;
        }
        if (!fits) continue;
        // [patterns | ranks]
        RTC_HIP(ctx, hipMalloc(&kc.d_bucket, (size_t)BUCKET_BYTES * 2));
        RTC_HIP(ctx, hipMemcpy(kc.d_bucket, pat.data(), BUCKET_BYTES, hipMemcpyHostToDevice));
        RTC_HIP(ctx, hipMemcpy((char*)kc.d_bucket + BUCKET_BYTES, rnk.data(), BUCKET_BYTES, hipMemcpyHostToDevice));
        kc.bvar = var;
      }
      if (kc.bvar >= 0) {
        // the forward-strand prefilter over S2 = every kept 12-mer and its reverse complement (sketch_kssd_bloom_kernel):
        // [0, 32 KiB) stage 1: bit c of the map is set when c = 18 bits of a member at one of the four alignments a
        // dword's k-mers have (member >> 0, 2, 4, 6); [32 KiB, 60 KiB) stage 2: blocked Bloom filter, block and the four
        // bits (two per 32-bit half) from a multiplicative hash of the member
        std::vector<uint32_t> bloom(BLOOM_BYTES / 4, 0u);
        uint32_t* bl2 = bloom.data() + CORE_BYTES / 4;
        for (uint32_t key : keys) {
          uint32_t rc = 0;
          for (int i = 0; i < 12; i++) rc |= (((key >> (2 * i)) & 3u) ^ 3u) << (2 * (11 - i));
          for (uint32_t v0 : {key, rc}) {
            const uint32_t v = v0 ^ ((v0 >> 1) & 0x555555u);  // gray24: every base b as b ^ (b >> 1), the prefilter's alphabet
            for (int al = 0; al < 4; al++) {
              const uint32_t core = (v >> (2 * al)) & 0x3ffffu;
              bloom[core >> 5] |= 1u << (core & 31u);
            }
            const uint32_t h = v * BLOOM2_MUL;
            const uint32_t blk = (uint32_t)(((uint64_t)h * BLOOM2_BLOCKS) >> 32);
            bl2[2 * blk] |= (1u << ((h >> 8) & 31u)) | (1u << ((h >> 3) & 31u));
            bl2[2 * blk + 1] |= (1u << ((h >> 13) & 31u)) | (1u << (h & 31u));
          }
        }
        RTC_HIP(ctx, hipMalloc(&kc.d_bloom, BLOOM_BYTES));
        RTC_HIP(ctx, hipMemcpy(kc.d_bloom, bloom.data(), BLOOM_BYTES, hipMemcpyHostToDevice));
      }
    }
    // the cuckoo / HBM structures below serve the k-mer lengths the prefilter kernel does not cover
    if (lds_index) {
      // two-table cuckoo placement of the kept (dim_id -> rank) pairs; smallest tables that work
      const int dimbits = 4 * half_subk;
      std::vector<uint32_t> t1, t2;
      bool placed_all = false;
      size_t nkept = 0;
      const int tries[2][2] = {{13, 12}, {13, 13}};
      for (int tr = 0; tr < 2 && !placed_all; tr++) {
        const int b1 = tries[tr][0], b2 = tries[tr][1], hishift = dimbits - b2;
        struct Slot { int64_t key; uint32_t rank; };
        std::vector<Slot> a1((size_t)1 << b1, Slot{-1, 0}), a2((size_t)1 << b2, Slot{-1, 0});
        placed_all = true; nkept = 0;
        for (int t = 0; t < dim_size && placed_all; t++) {
          if (h_shuffled_dim[t] < 0 || h_shuffled_dim[t] >= dim_end) continue;
          nkept++;
          Slot cur{t, (uint32_t)h_shuffled_dim[t]};
          bool done = false;
          for (int kick = 0; kick < 512 && !done; kick++) {
            const uint32_t s1 = (uint32_t)cur.key & ((1u << b1) - 1u);
            if (a1[s1].key < 0) { a1[s1] = cur; done = true; break; }
            const uint32_t s2 = (uint32_t)cur.key >> hishift;
            if (a2[s2].key < 0) { a2[s2] = cur; done = true; break; }
            // evict alternately from table 1 / table 2
            if (kick & 1) std::swap(cur, a2[s2]); else std::swap(cur, a1[s1]);
          }
          if (!done) placed_all = false;
        }
        if (!placed_all) continue;
        // Encoding.  Table 1 (slot = low b1 bits of the key): key bits b1-1 .. dimbits-1 at their own
        // positions, rank (< 4096) in bits 0..11.  Table 2 (slot = high b2 bits): key bits 0 .. hishift at
        // their own positions, rank in bits 20..31.  Each keeps ONE bit its slot already implies: an
        // unused slot holds that bit inverted, which no key of the slot can match.
        const uint32_t m1key = (dimbits == 32 ? ~0u : ((1u << dimbits) - 1u)) & ~((1u << (b1 - 1)) - 1u);
        const uint32_t m2key = (1u << (hishift + 1)) - 1u;
        t1.resize(a1.size()); t2.resize(a2.size());
        for (size_t q = 0; q < a1.size(); q++)
          t1[q] = a1[q].key < 0 ? ((((uint32_t)~q >> (b1 - 1)) & 1u) << (b1 - 1)) : (((uint32_t)a1[q].key & m1key) | a1[q].rank);
        for (size_t q = 0; q < a2.size(); q++)
          t2[q] = a2[q].key < 0 ? (((uint32_t)~q & 1u) << hishift) : (((uint32_t)a2[q].key & m2key) | (a2[q].rank << 20));
        kc.ck1 = b1; kc.ck2 = b2;
      }
      if (nkept > (size_t)MAX_LDS_KEEP) return rtc_fail(ctx, RTC_ERR_ARG, "shuffle table is not a permutation");
      if (placed_all) {
        RTC_HIP(ctx, hipMalloc(&kc.d_index, (t1.size() + t2.size()) * 4));
        RTC_HIP(ctx, hipMemcpy(kc.d_index, t1.data(), t1.size() * 4, hipMemcpyHostToDevice));
        RTC_HIP(ctx, hipMemcpy((char*)kc.d_index + t1.size() * 4, t2.data(), t2.size() * 4, hipMemcpyHostToDevice));
      } else {  // could not place every key: use the HBM table for this configuration
        RTC_HIP(ctx, hipMalloc((void**)&kc.d_table, (size_t)dim_size * 4));
        RTC_HIP(ctx, hipMemcpy(kc.d_table, h_shuffled_dim, (size_t)dim_size * 4, hipMemcpyHostToDevice));
      }
    } else {
      RTC_HIP(ctx, hipMalloc((void**)&kc.d_table, (size_t)dim_size * 4));
      RTC_HIP(ctx, hipMemcpy(kc.d_table, h_shuffled_dim, (size_t)dim_size * 4, hipMemcpyHostToDevice));
    }
    kc.half_subk = half_subk; kc.drlevel = drlevel; kc.checksum = cs;
  }

  // ---- segments (all segments of a genome append to the same row) ----
  uint64_t total = 0;
  for (uint32_t g = 0; g < n; g++) {
    if (h_off[g + 1] < h_off[g]) return rtc_fail(ctx, RTC_ERR_ARG, "offsets not monotone at genome %u", g);
    total += h_off[g + 1] - h_off[g];
  }
  // prefilter kernel: dim_id of 24 bits in the middle of at most 28 bases
  const bool use_bucket = kc.bvar >= 0 && K >= 18 && K <= 28;
  if (pk) {
    if (!(use_bucket && kc.d_bloom))
      return rtc_fail(ctx, RTC_ERR_UNSUPPORTED, "packed input is sketched by the prefilter kernel only (k 17..28, drlevel 3 or 4): unpack the batch (rtc_unpack_bases_dev) for k=%d, drlevel=%d", kmer_size, drlevel);
    if (h_off[n] > pk->n_bases) return rtc_fail(ctx, RTC_ERR_ARG, "offsets reach %llu, the batch holds %llu bases", (unsigned long long)h_off[n], (unsigned long long)pk->n_bases);
  }
  uint64_t seg_len = total / ((uint64_t)ctx->num_cu * 12);
  const uint64_t min_seg = pk ? (uint64_t)(WGB / 64) * PCHUNK * 8 : use_bucket ? (uint64_t)(WGB / 64) * CHUNK * 16 : 4ull * TILE_BASES_MAX;  // >= 16 (8) chunks per wave
  if (seg_len < min_seg) seg_len = min_seg;
  std::vector<KSegment> segs;
  segs.reserve(n + 1024);
  for (uint32_t g = 0; g < n; g++) {
    const uint64_t b = h_off[g], e = h_off[g + 1], len = e - b;
    uint64_t ns = (len + seg_len / 2) / seg_len;
    if (ns < 1) ns = 1;
    if (ns > 4096) ns = 4096;
    for (uint64_t i = 0; i < ns; i++) segs.push_back(KSegment{b, e, b + len * i / ns, b + len * (i + 1) / ns, g, 0});
  }
  void* ws0 = nullptr;
  const size_t bseg = segs.size() * sizeof(KSegment);
  RTC_TRY(rtc_ws(ctx, 0, bseg + 64, &ws0));
  void* hp = nullptr;
  RTC_TRY(rtc_pinned(ctx, bseg + 64, &hp));
  RTC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  memcpy(hp, segs.data(), bseg);
  RTC_HIP(ctx, hipMemcpyAsync(ws0, hp, bseg, hipMemcpyHostToDevice, ctx->stream));
  RTC_HIP(ctx, hipMemsetAsync(d_cnt, 0, (size_t)n * 4, ctx->stream));

  lds_index = kc.d_index != nullptr;  // false when the cuckoo build fell back to the HBM table
  const uint32_t* d_t1 = (const uint32_t*)kc.d_index;
  P.ck1 = kc.ck1; P.ck2 = kc.ck2;
  P.lshift = 64 - 2 * K;
  P.dimmask = P.dimbits >= 32 ? ~0u : ((1u << P.dimbits) - 1u);
  P.m1key = P.dimmask & ~((1u << (P.ck1 - 1)) - 1u);
  P.m2key = (1u << (P.dimbits - P.ck2 + 1)) - 1u;
  const uint32_t* d_t2 = kc.d_index ? d_t1 + ((size_t)1 << kc.ck1) : nullptr;
  const size_t lds = lds_index ? (((size_t)1 << kc.ck1) + ((size_t)1 << kc.ck2)) * 4 : 16;
  if (pk) {
    if (ctx->opt.verbose) fprintf(stderr, "[kssd] forward-strand prefilter over packed bases, K=%d, %zu segments, %llu runs\n", K, segs.size(), (unsigned long long)pk->n_runs);
    void* ws4 = nullptr;
    RTC_TRY(rtc_ws(ctx, 4, segs.size() * sizeof(uint2) + 64, &ws4));
    hipLaunchKernelGGL(packed_seg_runs_kernel, dim3((uint32_t)((segs.size() + 255) / 256)), dim3(256), 0, ctx->stream, (const KSegment*)ws0,
                       (uint32_t)segs.size(), pk->d_runs, (uint32_t)pk->n_runs, K, (uint2*)ws4);
    RTC_CHECK_LAUNCH(ctx);
    const PackedBatch B{d_seq, pk->n_bases, pk->d_runs, (const uint2*)ws4};
    const uint32_t* d_bk = (const uint32_t*)kc.d_bucket;
    const uint16_t* d_rk = (const uint16_t*)((const char*)kc.d_bucket + BUCKET_BYTES);
    const int lds_bl = BLOOM_BYTES + Q1_BYTES + BQ_BYTES;
#define LAUNCH_PACKED(KK)                                                                                             \
  case KK: {                                                                                                         \
    auto kern = sketch_kssd_packed_kernel<KK>;                                                                       \
    RTC_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bl));        \
    hipLaunchKernelGGL(kern, dim3((uint32_t)segs.size()), dim3(WGB), lds_bl, ctx->stream, B, (const KSegment*)ws0, P,  \
                       (const uint32_t*)kc.d_bloom, d_bk, d_rk, kc.bvar, d_out, stride, d_cnt);                       \
  } break
    switch (K) {
      LAUNCH_PACKED(18); LAUNCH_PACKED(20); LAUNCH_PACKED(22); LAUNCH_PACKED(24); LAUNCH_PACKED(26); LAUNCH_PACKED(28);
      default: return rtc_fail(ctx, RTC_ERR_ARG, "K=%d", K);
    }
#undef LAUNCH_PACKED
  } else if (use_bucket && kc.d_bloom) {
    P.nofast = ctx->opt.kssd_nofast;
    if (ctx->opt.verbose) fprintf(stderr, "[kssd] forward-strand prefilter, K=%d, %zu segments\n", K, segs.size());
    const uint32_t* d_bk = (const uint32_t*)kc.d_bucket;  // the exact index (variant kc.bvar) serves the drain from global memory
    const uint16_t* d_rk = (const uint16_t*)((const char*)kc.d_bucket + BUCKET_BYTES);
    const int lds_bl = BLOOM_BYTES + Q1_BYTES + BQ_BYTES;
#define LAUNCH_BLOOM(KK)                                                                                              \
  case KK: {                                                                                                         \
    auto kern = sketch_kssd_bloom_kernel<KK>;                                                                 \
    RTC_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bl));        \
    hipLaunchKernelGGL(kern, dim3((uint32_t)segs.size()), dim3(WGB), lds_bl, ctx->stream, d_seq,                      \
                       (const KSegment*)ws0, P, (const uint32_t*)kc.d_bloom, d_bk, d_rk, kc.bvar, d_out, stride, d_cnt); \
  } break
    switch (K) {
      LAUNCH_BLOOM(18); LAUNCH_BLOOM(20); LAUNCH_BLOOM(22); LAUNCH_BLOOM(24); LAUNCH_BLOOM(26); LAUNCH_BLOOM(28);
      default: return rtc_fail(ctx, RTC_ERR_ARG, "K=%d", K);
    }
#undef LAUNCH_BLOOM
  } else {
#define LAUNCH_KSSD2(IX, RUN, WARM)                                                                                   \
  do {                                                                                                               \
    auto kern = sketch_kssd_kernel<IX, RUN, WARM>;                                                                   \
    RTC_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));      \
    hipLaunchKernelGGL(kern, dim3((uint32_t)segs.size()), dim3(WG), lds, ctx->stream, d_seq, (const KSegment*)ws0, P, \
                       d_t1, d_t2, (const int32_t*)kc.d_table, d_out, stride, d_cnt);                               \
  } while (0)
#define LAUNCH_KSSD(IX) do { if (K <= 25) LAUNCH_KSSD2(IX, 18, 6); else LAUNCH_KSSD2(IX, 19, 9); } while (0)
    if (ctx->opt.verbose) fprintf(stderr, "[kssd] %s index, K=%d, %zu segments\n", lds_index ? "cuckoo" : "HBM table", K, segs.size());
    if (lds_index) LAUNCH_KSSD(IDX_CUCKOO); else LAUNCH_KSSD(IDX_HBM);
#undef LAUNCH_KSSD2
#undef LAUNCH_KSSD
  }
  RTC_CHECK_LAUNCH(ctx);

  // ---- capacity check, then per-genome sort + dedup ----
  void* ws3 = nullptr;
  RTC_TRY(rtc_ws(ctx, 3, 64, &ws3));
  uint32_t* d_max = (uint32_t*)ws3;
  RTC_HIP(ctx, hipMemsetAsync(d_max, 0, 4, ctx->stream));
  hipLaunchKernelGGL(max_u32_kernel, dim3(std::min<uint32_t>((n + 255) / 256, 1024)), dim3(256), 0, ctx->stream, d_cnt, n, d_max);
  RTC_CHECK_LAUNCH(ctx);
  uint32_t h_max = 0;
  RTC_HIP(ctx, hipMemcpyAsync(&h_max, d_max, 4, hipMemcpyDeviceToHost, ctx->stream));
  RTC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (h_need) *h_need = h_max;
  if (h_max > stride) return rtc_fail(ctx, RTC_ERR_OVERFLOW, "a genome produced %u KSSD tuples, stride is %u", h_max, stride);
  const int cap_max = use64 ? 16384 : 32768;  // one LDS buffer (128 KiB)
  int cap = 1024;
  while (cap < (int)h_max && cap < cap_max) cap <<= 1;
  const size_t lds_s = (size_t)cap * (use64 ? 8 : 4) + (WG / 64 + 1) * 4;
  // rows of up to WAVE_SORT_MAX tuples: one wave each, in registers; the LDS kernel only when some row is longer
  if (use64) hipLaunchKernelGGL(kssd_sort_unique_wave_kernel<uint64_t>, dim3((n + 3) / 4), dim3(256), 0, ctx->stream, (uint64_t*)d_out, stride, d_cnt, n);
  else hipLaunchKernelGGL(kssd_sort_unique_wave_kernel<uint32_t>, dim3((n + 3) / 4), dim3(256), 0, ctx->stream, (uint32_t*)d_out, stride, d_cnt, n);
  RTC_CHECK_LAUNCH(ctx);
  if (h_max <= (uint32_t)WAVE_SORT_MAX) {
    // every row is done
  } else if (use64) {
    auto kern = kssd_sort_unique_kernel<uint64_t>;
    RTC_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s));
    hipLaunchKernelGGL(kern, dim3(n), dim3(WG), lds_s, ctx->stream, (uint64_t*)d_out, stride, d_cnt, cap, WAVE_SORT_MAX);
  } else {
    auto kern = kssd_sort_unique_kernel<uint32_t>;
    RTC_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s));
    hipLaunchKernelGGL(kern, dim3(n), dim3(WG), lds_s, ctx->stream, (uint32_t*)d_out, stride, d_cnt, cap, WAVE_SORT_MAX);
  }
  RTC_CHECK_LAUNCH(ctx);
  if (h_max > (uint32_t)cap) RTC_TRY(kssd_sort_big_rows(ctx, d_out, stride, d_cnt, n, use64, cap));
  return RTC_OK;
}

extern "C" int rtc_sketch_kssd_dev(rtc_ctx* ctx, const uint8_t* d_seq, const uint64_t* h_off, uint32_t n,
                                   int kmer_size, int drlevel, const int32_t* h_shuffled_dim, void* d_out,
                                   uint32_t stride, uint32_t* d_cnt, int* width_out, uint32_t* h_need) {
  return sketch_kssd_impl(ctx, d_seq, nullptr, h_off, n, kmer_size, drlevel, h_shuffled_dim, d_out, stride, d_cnt, width_out, h_need);
}

extern "C" int rtc_sketch_kssd_packed_dev(rtc_ctx* ctx, const uint8_t* d_packed, uint64_t n_bases, const uint64_t* d_runs,
                                          uint64_t n_runs, const uint64_t* h_off, uint32_t n, int kmer_size, int drlevel,
                                          const int32_t* h_shuffled_dim, void* d_out, uint32_t stride, uint32_t* d_cnt,
                                          int* width_out, uint32_t* h_need) {
  const PackedArgs pk{n_bases, d_runs, n_runs};
  const int st = sketch_kssd_impl(ctx, d_packed, &pk, h_off, n, kmer_size, drlevel, h_shuffled_dim, d_out, stride, d_cnt, width_out, h_need);
  // (the capacity read-back has synchronised the stream: what the run check found is known by now)
  return st == RTC_OK && ctx ? rtc_sticky_error(ctx) : st;
}

namespace { __global__ void touch_unit_kernel() {} }
int rtc_touch_sketch_kssd(rtc_ctx* ctx) {
  hipLaunchKernelGGL(touch_unit_kernel, dim3(1), dim3(64), 0, ctx->stream);
  RTC_CHECK_LAUNCH(ctx);
  return RTC_OK;
}

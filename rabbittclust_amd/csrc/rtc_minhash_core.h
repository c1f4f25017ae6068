// rtc_minhash_core.h -- what the two MinHash sketch translation units share (rtc_sketch_minhash.hip: ASCII input,
// rtc_sketch_minhash_packed.hip: the 2-bit staging format): the MurmurHash3 evaluation from LDS product tables, the
// in-LDS bottom-s merge, the partial-sketch merge kernel and the host-side segment plan.  Everything lives in an
// anonymous namespace: each unit gets its own copy.
#pragma once
#include <algorithm>

#include "rtc_internal.h"

namespace {

constexpr int WG = 512;                               // lanes per workgroup (8 waves)
constexpr int NWAVE = WG / 64;
// 19 dwords = 76 owned positions per lane and tile: lane runs 76 B apart keep the set of live 128-B
// lines (one or two lanes per line, ~1500 lanes per CU) inside the 4 MiB L2 of an XCD.  31 dwords is
// 1 % faster but re-reads 56 % of the input from the fabric (measured: TCC_EA0_RDREQ_128B, 19 dwords: 11 %).
constexpr int RUN_DW = 19;                            // dwords of owned bases per lane per tile
constexpr int OWN = RUN_DW * 4;                       // k-mer end positions a lane owns per tile
// warm-up dwords in front of a lane's owned run (they only roll the windows): k-1 bases rounded up so
// that warm-up + run are whole 16-byte loads -- 5 (20 bases, six loads per tile) for the compile-time
// k <= 21, 9 (36 bases >= 31, seven loads) otherwise
__host__ __device__ constexpr int warm_dw(int kt) { return (kt > 0 && kt <= 21) ? 5 : 9; }
static_assert((RUN_DW + warm_dw(21)) % 4 == 0 && (RUN_DW + warm_dw(0)) % 4 == 0, "a lane's window must be whole 16-byte loads");
constexpr int TILE_BASES = WG * RUN_DW * 4;           // bases per tile
// Safe mode appends ONE k-mer per lane between two looks at the candidate count (four until round 5): the room the buffer
// has to guarantee is a quarter, and a third workgroup per CU fits up to s = 3 318 at k = 21 (3 574 with the packed tables)
// instead of 1 782 (2 038) -- clust-greedy's containment sketches of 2 - 3.5 Mbp genomes, and BASELINE config 4's sketches
// no longer need the packed layout (its stride-16 reads cost bank conflicts): 115.6 -> 112.5 ms there, 85.8 -> 81.0 ms at s = 3000.
constexpr int STEP_APPENDS = WG;                      // worst-case appends between two looks at the count in safe mode
constexpr int MIN_ROOM = STEP_APPENDS;                // candidate room the buffer always offers
constexpr uint64_t SENT = ~0ULL;
// per-wave queue of possible candidates (unfinished hash halves) in LDS: see the steady state of the kernel
constexpr int QCAP = 32;                              // entries per wave
constexpr int QDRAIN = 16;                            // drained at a tile end once this many wait
constexpr size_t QUEUE_BYTES = (size_t)NWAVE * QCAP * 16;
// hash tables in LDS (see kmer_hash_parts): per 8 bases of k one 256-entry table of 16-byte entries
// (first four bases of the word) and, where the word has more than four bases, one of 4-byte entries
constexpr size_t LUT_LO_BYTES = 256 * 16, LUT_HI_BYTES = 256 * 4;
// A last word of at most five bases has a table of its own instead: 4^bases entries of 8 bytes holding the
// finished contribution (at most 8 KiB), one ds_read_b64 and two VALU instructions for the word.
__host__ __device__ constexpr int lut_words(int k) { return (k + 7) / 8; }
__host__ __device__ constexpr int lut_last_nb(int k) { return k - 8 * (lut_words(k) - 1); }   // bases of the last word, 1..8
__host__ __device__ constexpr bool lut_direct(int k) { return lut_last_nb(k) <= 5; }
__host__ __device__ constexpr int lut_los(int k) { return lut_words(k) - (lut_direct(k) ? 1 : 0); }  // words with the table pair
__host__ __device__ constexpr int lut_his(int k) { return lut_direct(k) ? lut_los(k) : (k + 3) / 8; }  // word w has a second half iff k > 8w + 4
// Packed layout (pk): lo(b * c) of the 4-byte tables sits in the spare fourth dword of the 16-byte entries instead --
// lut_his(k) KiB less LDS, stride-16 reads (more bank conflicts: the headline shape loses 1.9 %, k = 23 6.5 %).  The
// launch picks it only where it buys a workgroup per CU (k = 21: 3318 < s <= 3574).
__host__ __device__ constexpr size_t lut_direct_bytes(int k) { return lut_direct(k) ? ((size_t)8 << (2 * lut_last_nb(k))) : 0; }
__host__ __device__ constexpr size_t lut_bytes(int k, bool pk) {
  return (size_t)lut_los(k) * LUT_LO_BYTES + (pk ? 0 : (size_t)lut_his(k) * LUT_HI_BYTES) + lut_direct_bytes(k);
}

struct Segment {
  uint64_t g_begin, g_end;  // genome byte range in d_seq
  uint64_t s_begin, s_end;  // k-mer END positions owned by this segment (absolute)
  uint64_t out_off;         // element offset into out buffer
  uint64_t lo_off;          // pass > 0: offset (final buffer) of the largest hash kept by earlier passes
  uint32_t cnt_slot;        // index into cnt buffer
  uint32_t sketch_size;     // hashes to select in this pass
  uint32_t final_slot;      // genome index in the final count buffer
  uint32_t expect;          // pass > 0: run only if the genome already holds exactly this many hashes
  uint32_t partial;         // 1: one of several segments of a genome, writes a partial sketch for the merge kernel
  uint32_t pad;
  uint64_t t0;              // starting threshold (SENT: none), see the kernel
};

struct Ctrl {
  uint64_t T;
  uint64_t T0;       // the threshold while fewer than s hashes are held (SENT, or the segment's starting threshold)
  uint32_t sorted;   // buf[0..sorted) is ascending and distinct (what the last merge left); appends follow it
  uint32_t pad0;
  uint32_t count;
  uint32_t overflow;
  uint32_t saw_max;
  uint32_t scan_base;
  uint32_t wave_tot[NWAVE];
};

// Every LDS object is reached through address_space(3) pointers: the accesses are ds_* instructions
// by construction (not flat ones whose selection depends on what the optimiser can prove).
#define RTC_LDS __attribute__((address_space(3)))
typedef RTC_LDS uint64_t* lds_u64_ptr;
typedef RTC_LDS Ctrl* lds_ctrl_ptr;
typedef RTC_LDS unsigned char* lds_byte_ptr;
struct MergeResult { uint32_t count; uint64_t T; };
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));  // plain vector (HIP's uint4 is a class, unusable through LDS pointers)

// ---- MurmurHash3_x64_128 (first output word) of the canonical k-mer's ASCII bytes ----------------
// MurmurHash3 turns every 64-bit input word w into  rotl(w * c, r) * c'  (c, r, c' = c1, 31, c2 for the
// k1 words, c2, 33, c1 for the k2 words) before mixing it into the state.  With w = a | b << 32
// (a = bytes 0-3, b = bytes 4-7 of the word) multiplication mod 2^64 gives
//     S = w * c :   S.lo = lo(a * c),   S.hi = hi(a * c) + lo(b * c)   (one 32-bit add, no other carry)
// and the rotation by 31 / 33 moves the two halves of S to DISJOINT bit ranges, so
//     rotl(S, r) * c' = X(S.lo) * c' + Y(S.hi) * c'
// where the first term depends on the word's first four bases only: it comes out of an LDS table
// together with hi(a * c) (16-byte entries, one ds_read_b128), lo(b * c) out of a second table
// (4-byte entries), and the second term is a 32 x 64-bit product:
//     r = 31:  Y = S.hi >> 1 | (S.hi & 1) << 63   ->  (S.hi >> 1) * c' + (S.hi << 31) in the top word
//     r = 33:  Y = S.hi << 1 (33 bits)            ->  (S.hi * c') << 1
// i.e. 6 / 5 VALU instructions per word instead of 7 (add, two-instruction rotate, 64 x 64 multiply),
// and no 2-bit -> ASCII expansion at all.  The tables are built once per workgroup for the runtime k.
__device__ __forceinline__ uint64_t fmix64(uint64_t x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return x;
}

// fmix64 without its last multiply and xorshift.  With f = x * FMIX_C2 for the two halves the hash is
// (f1 ^ f1 >> 33) + (f2 ^ f2 >> 33); the last xorshift changes the low words only, so the high words of f1 and
// f2 decide almost every threshold test -- and their SUM is all the test needs, which is the high word of ONE
// product, (a + b) * C, up to a carry: hash_test_word.  The full halves are formed on demand, for the few k-mers
// that pass (mm_finish).
constexpr uint64_t FMIX_C2 = 0xc4ceb9fe1a85ec53ULL;
__device__ __forceinline__ uint64_t fmix64_open(uint64_t x) {
  x ^= x >> 33;
  // cross terms as v_mul_lo + a 32-bit mad (v_mad_u64_u32, low word) and one two-input add: the plain
  // 64-bit product compiles to two v_mul_lo and a v_add3, 1.2 issue cycles more (same-run 82.2 -> 81.5 ms)
  constexpr uint32_t cl = 0xed558ccdu, ch = 0xff51afd7u;
  const uint32_t xl = (uint32_t)x, xh = (uint32_t)(x >> 32);
  const uint64_t D = (uint64_t)xl * cl;
  uint32_t e = xl * ch + xh * cl;
  asm("" : "+v"(e));
  x = __builtin_bit_cast(uint64_t, make_uint2((uint32_t)D, (uint32_t)(D >> 32) + e));
  x ^= x >> 33;
  return x;
}
struct HashParts { uint64_t f1, f2; };  // the two halves before their last multiply
__device__ __forceinline__ uint64_t mm_finish(const HashParts& p) {
  const uint64_t f1 = p.f1 * FMIX_C2, f2 = p.f2 * FMIX_C2;
  return (f1 ^ (f1 >> 33)) + (f2 ^ (f2 >> 33));
}
// The test word.  (f1 + f2) mod 2^64 = (a + b) * C for the halves a, b before their last multiply, and its high
// word is hi(f1) + hi(f2) + c1 (c1: the carry of the two low words), while hi(hash) = hi(f1) + hi(f2) + c2 (c2: the
// carry of the low words AFTER their xorshift).  So w = hi((a + b) * C) + 1 lies in {hi(hash), hi(hash) + 1,
// hi(hash) + 2} (mod 2^32): a hash below T has w <= hi(T) + 2, also when hi(f1) + hi(f2) wraps (w = 0 or 1 then).
// One 64-bit add and the high word of ONE 64 x 64 product (mulhi + two cross products + add3) instead of two.
constexpr uint32_t TEST_SLACK = 2;  // callers compare with hi(T) + TEST_SLACK and need hi(T) + TEST_SLACK < 2^32
__device__ __forceinline__ uint32_t hash_test_word(const HashParts& p) {
  const uint64_t S = p.f1 + p.f2;
  const uint32_t slo = (uint32_t)S, shi = (uint32_t)(S >> 32);
  const uint32_t cl = (uint32_t)FMIX_C2, ch = (uint32_t)(FMIX_C2 >> 32);
  // as a chain of two 32-bit multiply-adds (v_mad_u64_u32, low word) on top of the v_mul_hi: one three-input add fewer
  uint32_t m = __umulhi(slo, cl);
  asm("" : "+v"(m));
  uint32_t r = slo * ch + m;
  asm("" : "+v"(r));
  uint32_t r2 = shi * cl + r;
  asm("" : "+v"(r2));
  return r2 + 1u;
}

constexpr uint64_t MM_C1 = 0x87c37b91114253d5ULL, MM_C2 = 0x4cf5ad432745937fULL;
constexpr uint32_t MASH_SEED = 42;  // the seed of every reference call site (Sketch::MinHash)

__device__ __forceinline__ uint32_t codes_to_ascii(uint32_t e) {
  // e holds 4 base codes, first base in bits 7..6; returns the 4 ASCII bytes, first base lowest
  const uint32_t t = ((e >> 6) & 3u) | (((e >> 4) & 3u) << 8) | (((e >> 2) & 3u) << 16) | ((e & 3u) << 24);
  return __builtin_amdgcn_perm(0u, 0x54474341u, t);
}

struct KParams {
  int k;
  uint32_t seed;
  uint32_t use64;
  int lshift;          // 64 - 2k
  int rc_shift;        // 2k - 2
  uint64_t kmask;      // low 2k bits
  bool packed;         // table layout, see lut_bytes
};

__device__ __forceinline__ KParams make_kparams(int k, uint32_t seed, bool packed) {
  KParams P;
  P.k = k; P.seed = seed; P.use64 = k > 16 ? 1u : 0u; P.packed = packed;
  P.lshift = 64 - 2 * k; P.rc_shift = 2 * k - 2;
  P.kmask = k == 32 ? ~0ULL : ((1ULL << (2 * k)) - 1);
  return P;
}

// Called by all WG threads; the first 256 fill one column each.  Layout: lut_los(k) tables of
// {u64 P, u32 AH, pad} at w * LUT_LO_BYTES, then lut_his(k) tables of u32 BL, then the last word's own table
// (lut_direct(k)): entry i = rotl(w * c, r) * c' of the word whose bases are the 2-bit codes of i.
__device__ __forceinline__ void build_kmer_lut(lds_byte_ptr lut, int k, bool pk) {
  const uint32_t e = threadIdx.x;
  if (e >= 256) return;
  const uint32_t a4 = codes_to_ascii(e);
  const uint32_t hi_base = (uint32_t)(lut_los(k) * LUT_LO_BYTES);
  if (lut_direct(k)) {
    const int wl = lut_words(k) - 1, nb = lut_last_nb(k);
    typedef RTC_LDS uint64_t* lds_u64w_ptr;
    const lds_u64w_ptr dt = (lds_u64w_ptr)(lut + hi_base + (pk ? 0 : (size_t)lut_his(k) * LUT_HI_BYTES));
    for (uint32_t i = e; i < (1u << (2 * nb)); i += 256) {
      const uint32_t e4 = nb >= 4 ? ((i >> (2 * nb - 8)) & 0xffu) : ((i << (8 - 2 * nb)) & 0xffu);   // first four bases, first on top
      const uint32_t am = nb >= 4 ? 0xffffffffu : ((1u << (8 * nb)) - 1u);
      uint64_t wd = (uint64_t)(codes_to_ascii(e4) & am);
      if (nb == 5) wd |= (uint64_t)(codes_to_ascii((i & 3u) << 6) & 0xffu) << 32;
      uint64_t S, K;
      if (wl & 1) { S = wd * MM_C2; K = ((S << 33) | (S >> 31)) * MM_C1; }
      else { S = wd * MM_C1; K = ((S << 31) | (S >> 33)) * MM_C2; }
      dt[i] = K;
    }
  }
#pragma unroll
  for (int w = 0; w < 4; w++) {
    const int nb = k - 8 * w;  // bytes of this word
    if (nb <= 0 || w >= lut_los(k)) break;
    const uint64_t c = (w & 1) ? MM_C2 : MM_C1;
    const uint32_t am = nb >= 4 ? 0xffffffffu : ((1u << (8 * nb)) - 1u);
    const uint64_t A = (uint64_t)(a4 & am) * c;
    const uint32_t AL = (uint32_t)A, AH = (uint32_t)(A >> 32);
    uint64_t P;
    if (w & 1) P = (((uint64_t)(uint32_t)(AL << 1) << 32) | (uint64_t)(AL >> 31)) * MM_C1;   // rotl33 part of S.lo, times c1
    else P = (((uint64_t)(AL >> 1) << 32) | ((uint64_t)(AL & 1u) << 31)) * MM_C2;           // rotl31 part of S.lo, times c2
    typedef RTC_LDS u32x4* lds_u4_ptr;
    const uint32_t bm = nb >= 8 ? 0xffffffffu : (nb > 4 ? ((1u << (8 * (nb - 4))) - 1u) : 0u);
    const uint32_t BL = (a4 & bm) * (uint32_t)c;  // lo(b * c) of the word whose SECOND four bases are e's codes
    const u32x4 ent = {(uint32_t)P, (uint32_t)(P >> 32), AH, pk ? BL : 0u};
    *(lds_u4_ptr)(lut + (size_t)w * LUT_LO_BYTES + (size_t)e * 16) = ent;
    if (nb > 4 && !pk) {
      typedef RTC_LDS uint32_t* lds_u32_ptr;
      *(lds_u32_ptr)(lut + hi_base + (size_t)w * LUT_HI_BYTES + (size_t)e * 4) = BL;
    }
  }
}

// 64-bit rotate as two v_alignbit_b32 (the generic shift/or form costs three to four instructions)
template <int R>
__device__ __forceinline__ uint64_t rotl64c(uint64_t x) {
  static_assert(R > 0 && R < 64 && R != 32, "rotation amount");
  const uint32_t hi = (uint32_t)(x >> 32), lo = (uint32_t)x;
  uint32_t nh, nl;
  if (R < 32) {
    nh = __builtin_amdgcn_alignbit(hi, lo, 32 - R);
    nl = __builtin_amdgcn_alignbit(lo, hi, 32 - R);
  } else {
    nh = __builtin_amdgcn_alignbit(lo, hi, 64 - R);
    nl = __builtin_amdgcn_alignbit(hi, lo, 64 - R);
  }
  // assembled as a register pair and made opaque: otherwise the two halves are re-associated into
  // the 64-bit additions that follow ((lo, 0) + x + (0, hi): one more add and a move)
  uint64_t r = __builtin_bit_cast(uint64_t, make_uint2(nl, nh));
  asm("" : "+v"(r));
  return r;
}

// x*5 as one v_lshl_add_u64 ((x << 2) + x); the compiler's choice is two v_mad_u64_u32 plus moves
__device__ __forceinline__ uint64_t times5(uint64_t x) {
  uint64_t r;
  asm("v_lshl_add_u64 %0, %1, 2, %1" : "=v"(r) : "v"(x));
  return r;
}

// byte B of w, shifted left by `three` (the byte offset of a table entry), as one SDWA shift
template <int B>
__device__ __forceinline__ uint32_t byte_x8(uint32_t w, uint32_t three) {
  uint32_t r;
  if (B == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(three), "v"(w));
  if (B == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(three), "v"(w));
  if (B == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(three), "v"(w));
  if (B == 3) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(three), "v"(w));
  return r;
}

// rotl(S, 31) * c2 for a k1 word: e = {P.lo, P.hi, AH, -}, BL = lo(b * c1)
__device__ __forceinline__ uint64_t word_k1(const u32x4 e, uint32_t BL) {
  const uint32_t SH = e.z + BL, v = SH >> 1;
  const uint64_t R = (uint64_t)v * (uint32_t)MM_C2 + __builtin_bit_cast(uint64_t, make_uint2(e.x, e.y));  // v_mad_u64_u32
  // the high word takes v * hi(c2) + (SH & 1) << 31: t = hi(R) + (SH << 31) as one v_lshl_add_u32 (a fresh register, so it can be the
  // low half of an addend pair) and v * hi(c2) + t as a 32-bit mad (v_mad_u64_u32, low word)
  uint32_t t = (SH << 31) + (uint32_t)(R >> 32);
  asm("" : "+v"(t));
  const uint32_t hi = v * (uint32_t)(MM_C2 >> 32) + t;
  return __builtin_bit_cast(uint64_t, make_uint2((uint32_t)R, hi));
}
// rotl(S, 33) * c1 for a k2 word: (S.hi * c1) << 1 + P = S.hi * (2 c1 mod 2^64) + P, the table value as the
// addend of the v_mad_u64_u32 and the constant's high word as one cross product
__device__ __forceinline__ uint64_t word_k2(const u32x4 e, uint32_t BL) {
  constexpr uint64_t C1X2 = MM_C1 << 1;
  const uint32_t SH = e.z + BL;
  const uint64_t R = (uint64_t)SH * (uint32_t)C1X2 + __builtin_bit_cast(uint64_t, make_uint2(e.x, e.y));  // v_mad_u64_u32
  uint32_t cross = SH * (uint32_t)(C1X2 >> 32);
  asm("" : "+v"(cross));  // keep it a v_mul_lo_u32 + v_add_u32: fused into a second v_mad_u64_u32 it costs two extra moves
  return __builtin_bit_cast(uint64_t, make_uint2((uint32_t)R, (uint32_t)(R >> 32) + cross));
}

// The table words of one k-mer: what its LDS reads return (fields a k does not use are never read or touched).
// Two phases so that a caller can have the reads of one k-mer in flight under the arithmetic of another:
// kmer_loads issues them (offsets + ds_read), kmer_hash_finish consumes them.
struct KmerLoads {
  u32x4 e[4];       // 16-byte entries {P.lo, P.hi, AH, (BL)} of the words with a table pair
  uint32_t bl[4];   // lo(b * c) of their second halves
  uint64_t dt;      // the last word's own table entry (lut_direct)
};
// x: canonical k-mer, 2 bits per base, first base in the top bits (canon << (64 - 2k))
__device__ __forceinline__ KmerLoads kmer_loads(uint64_t x, const KParams& P) {
  const uint32_t hi = (uint32_t)(x >> 32), lo = (uint32_t)x;
  const int k = P.k;
  // the tables start at LDS address 0 (checked at kernel entry), so an LDS address is a table offset:
  // entry offset = code byte << 4 (16-byte entries) or << 2 (4-byte entries), one SDWA shift each;
  // the table bases fold into the ds_read immediates
  typedef const RTC_LDS u32x4* lds_u4_cptr;
  typedef const RTC_LDS uint32_t* lds_u32_cptr;
  const uint32_t four = 4, two = 2;
  const uint32_t hi_base = (uint32_t)(lut_los(k) * LUT_LO_BYTES);
  const bool pk = P.packed;
  const uint32_t dt_base = hi_base + (pk ? 0u : (uint32_t)(lut_his(k) * LUT_HI_BYTES));
  const uint32_t hsh = pk ? four : two;  // shift of the second-half offsets
  // the last word's own table: its 2nb bits sit at the top of the word's 16-bit slot (whatever lies below them is
  // not part of the k-mer and is cut off), entry offset = field << 3
  typedef const RTC_LDS uint64_t* lds_u64_cptr;
  const int dnb = lut_last_nb(k);
#define RTC_DT(H, odd) (*(lds_u64_cptr)(uintptr_t)(dt_base + ((odd) ? (__builtin_amdgcn_ubfe((H), 16 - 2 * dnb, 2 * dnb) << 3) \
                                                                     : (((H) >> (32 - 2 * dnb)) << 3))))
#define RTC_LO(w, off) (*(lds_u4_cptr)(uintptr_t)((off) + (uint32_t)((w) * LUT_LO_BYTES)))
#define RTC_HI(w, off) (*(lds_u32_cptr)(uintptr_t)((off) + (pk ? (uint32_t)((w) * LUT_LO_BYTES) + 12u : hi_base + (uint32_t)((w) * LUT_HI_BYTES))))
  KmerLoads L = {};
  const int dw = lut_direct(k) ? lut_words(k) - 1 : -1;  // the word that has a table of its own
  if (dw == 0) L.dt = RTC_DT(hi, false);
  else { L.e[0] = RTC_LO(0, byte_x8<3>(hi, four)); if (k > 4) L.bl[0] = RTC_HI(0, byte_x8<2>(hi, hsh)); }
  if (dw == 1) L.dt = RTC_DT(hi, true);
  else if (k > 8) { L.e[1] = RTC_LO(1, byte_x8<1>(hi, four)); if (k > 12) L.bl[1] = RTC_HI(1, byte_x8<0>(hi, hsh)); }
  if (dw == 2) L.dt = RTC_DT(lo, false);
  else if (k > 16) { L.e[2] = RTC_LO(2, byte_x8<3>(lo, four)); if (k > 20) L.bl[2] = RTC_HI(2, byte_x8<2>(lo, hsh)); }
  if (dw == 3) L.dt = RTC_DT(lo, true);
  else if (k > 24) { L.e[3] = RTC_LO(3, byte_x8<1>(lo, four)); if (k > 28) L.bl[3] = RTC_HI(3, byte_x8<0>(lo, hsh)); }
#undef RTC_LO
#undef RTC_HI
#undef RTC_DT
  return L;
}
__device__ __forceinline__ HashParts kmer_hash_finish(const KmerLoads& L, const KParams& P) {
  const int k = P.k;
  uint64_t K0 = 0, K1 = 0, K2 = 0, K3 = 0;  // the words' contributions, already rotl(w * c, r) * c'
  const int dw = lut_direct(k) ? lut_words(k) - 1 : -1;
  // The unused fourth dword of a 16-byte entry counts as live until here (an empty asm, no instruction): otherwise the
  // register allocator hands that register to the next read's address while the ds_read_b128 is still in flight and
  // has to wait for it (s_waitcnt lgkmcnt(0) between two reads of one k-mer -- the round trip this split exists to hide).
  if (!P.packed) {
    if (dw != 0) asm volatile("" :: "v"(L.e[0].w));
    if (dw != 1 && k > 8) asm volatile("" :: "v"(L.e[1].w));
    if (dw != 2 && k > 16) asm volatile("" :: "v"(L.e[2].w));
    if (dw != 3 && k > 24) asm volatile("" :: "v"(L.e[3].w));
  }
  if (dw == 0) K0 = L.dt;
  else K0 = word_k1(L.e[0], k > 4 ? L.bl[0] : 0u);
  if (dw == 1) K1 = L.dt;
  else if (k > 8) K1 = word_k2(L.e[1], k > 12 ? L.bl[1] : 0u);
  if (dw == 2) K2 = L.dt;
  else if (k > 16) K2 = word_k1(L.e[2], k > 20 ? L.bl[2] : 0u);
  if (dw == 3) K3 = L.dt;
  else if (k > 24) K3 = word_k2(L.e[3], k > 28 ? L.bl[3] : 0u);
  uint64_t h1 = P.seed, h2 = P.seed;
  uint64_t t0 = K0, t1 = K1;  // tail contributions
  if (k >= 16) {
    // first block with h1 == h2 == seed folded in: 5*(rotl27(seed ^ k1) + seed) + c = 5*rotl27(..) + (5*seed + c)
    h1 = rotl64c<27>(h1 ^ K0);
    h1 = times5(h1) + (5ULL * P.seed + 0x52dce729ULL);
    h2 = rotl64c<31>(h2 ^ K1) + h1;
    h2 = times5(h2) + 0x38495ab5ULL;
    t0 = K2; t1 = K3;
    if (k == 32) {
      h1 ^= K2; h1 = rotl64c<27>(h1); h1 += h2; h1 = times5(h1) + 0x52dce729;
      h2 ^= K3; h2 = rotl64c<31>(h2); h2 += h1; h2 = times5(h2) + 0x38495ab5;
      t0 = 0; t1 = 0;
    }
  }
  const int tail = k & 15;
  if (tail > 8) { h2 ^= t1; }
  if (tail > 0) {
    // h1 ^= k1; h1 ^= len: the low word as one three-input xor (v_bitop3_b32)
    const uint32_t lo = __builtin_amdgcn_bitop3_b32((uint32_t)h1, (uint32_t)t0, (uint32_t)k, 0x96);
    const uint32_t hi = (uint32_t)(h1 >> 32) ^ (uint32_t)(t0 >> 32);
    h1 = __builtin_bit_cast(uint64_t, make_uint2(lo, hi));
  } else {
    h1 ^= (uint64_t)k;
  }
  h2 ^= (uint64_t)k;
  h1 += h2; h2 += h1;
  return HashParts{fmix64_open(h1), fmix64_open(h2)};
}
__device__ __forceinline__ HashParts kmer_hash_parts(uint64_t x, const KParams& P) {
  return kmer_hash_finish(kmer_loads(x, P), P);
}
__device__ __forceinline__ uint64_t kmer_hash(uint64_t x, const KParams& P) {
  const uint64_t h = mm_finish(kmer_hash_parts(x, P));
  return P.use64 ? h : (h & 0xffffffffULL);
}

// ---- block-wide merge: sort buf[0..n), drop duplicates, keep the `s` smallest -----------------
// Normalised bitonic network (every compare-exchange ascending, first step of each merge mirrored)
// over the next power of two >= n; positions >= n stand for +infinity, which an ascending exchange
// never moves, so those exchanges are simply skipped and n may be any number.
__device__ void bitonic_sort_lds(lds_u64_ptr buf, int n) {
  const int t = threadIdx.x;
  int n2 = 2;
  while (n2 < n) n2 <<= 1;
  for (int k = 2; k <= n2; k <<= 1) {
    const int hk = k >> 1;
    for (int i = t; i < (n2 >> 1); i += WG) {
      const int off = i & (hk - 1);
      const int a = ((i - off) << 1) | off;       // block base (i / hk) * k, plus off
      const int b = (a - off) + (k - 1 - off);    // mirrored partner within the block
      if (b < n) {
        const uint64_t va = buf[a], vb = buf[b];
        if (va > vb) { buf[a] = vb; buf[b] = va; }
      }
    }
    __syncthreads();
    for (int j = k >> 2; j > 0; j >>= 1) {
      for (int i = t; i < (n2 >> 1); i += WG) {
        const int a = ((i & ~(j - 1)) << 1) | (i & (j - 1));
        const int b = a | j;
        if (b < n) {
          const uint64_t va = buf[a], vb = buf[b];
          if (va > vb) { buf[a] = vb; buf[b] = va; }
        }
      }
      __syncthreads();
    }
  }
}

// On entry: buf[0..ctrl->count) holds candidates (unsorted, duplicates allowed), all threads
// arrive.  On exit: buf[0..count) ascending distinct, count <= s, ctrl->T updated.
// first index in the ascending run a[0..n) whose value is >= v (STRICT = false) or > v (STRICT = true)
template <bool STRICT>
__device__ __forceinline__ uint32_t lds_bound(lds_u64_ptr a, uint32_t n, uint64_t v) {
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    const uint64_t x = a[mid];
    if (STRICT ? (x <= v) : (x < v)) lo = mid + 1; else hi = mid;
  }
  return lo;
}

__device__ __noinline__ MergeResult merge_block(lds_u64_ptr buf, lds_ctrl_ptr ctrl, int cap, uint32_t s) {
  const int t = threadIdx.x;
  __syncthreads();
  const uint32_t n = ctrl->count < (uint32_t)cap ? ctrl->count : (uint32_t)cap;
  uint32_t c0 = ctrl->sorted < n ? ctrl->sorted : n;  // sorted distinct prefix left by the previous merge
  if (c0 == n && n <= s) {  // nothing was appended since: the prefix is the result (workgroup-uniform)
    const uint64_t keepT = ctrl->T;
    __syncthreads();
    if (t == 0) ctrl->overflow = 0;
    __syncthreads();
    return MergeResult{n, keepT};
  }
  bool saw = false;  // a genuine hash equal to the padding value: remembered, re-appended at the end
  for (uint32_t i = c0 + t; i < n; i += WG) saw |= buf[i] == SENT;
  if (saw) ctrl->saw_max = 1;
  if (t == 0) ctrl->scan_base = 0;
  __syncthreads();
  // Only the appended candidates buf[c0..n) are sorted; they are then merged with the sorted prefix by
  // rank (every element finds its output position with one binary search in the other run: prefix
  // elements go before equal new ones) into the free space behind them.  Sorting n log^2 n elements
  // again at every merge cost 6.6 % of the kernel at s = 1000 and 25 % at s = 2000.  Falls back to
  // sorting everything when the output does not fit (2n > cap) or there is no prefix yet.
  lds_u64_ptr src = buf;
  if (c0 > 0 && 2 * n <= (uint32_t)cap) {
    const uint32_t m = n - c0;
    bitonic_sort_lds(buf + c0, (int)m);  // ends with a barrier
    for (uint32_t idx = t; idx < n; idx += WG) {
      const uint64_t v = buf[idx];
      const uint32_t pos = idx < c0 ? idx + lds_bound<false>(buf + c0, m, v) : (idx - c0) + lds_bound<true>(buf, c0, v);
      buf[n + pos] = v;
    }
    __syncthreads();
    src = buf + n;
  } else {
    bitonic_sort_lds(buf, (int)n);
  }
  // streaming compaction in rounds of WG elements (dest <= src index when in place; disjoint otherwise)
  const uint32_t lane = t & 63, wave = t >> 6;
  for (int r = 0; r < (int)n; r += WG) {
    const int idx = r + t;
    const bool in = idx < (int)n;
    const uint64_t v = in ? src[idx] : SENT;
    const bool keep = in && v != SENT && (idx == 0 || v != src[idx - 1]);
    const uint64_t bal = __ballot(keep);
    const uint32_t before = __popcll(bal & ((1ULL << lane) - 1ULL));
    if (lane == 0) ctrl->wave_tot[wave] = (uint32_t)__popcll(bal);
    __syncthreads();  // all reads of this round done; wave totals visible
    const uint32_t sb = ctrl->scan_base;
    uint32_t base = sb, total = 0;
#pragma unroll
    for (int w = 0; w < NWAVE; w++) {
      const uint32_t wt = ctrl->wave_tot[w];
      if ((uint32_t)w < wave) base += wt;
      total += wt;
    }
    const uint32_t dest = base + before;
    if (keep && dest < s) buf[dest] = v;
    __syncthreads();  // writes done; wave_tot / scan_base may be rewritten
    if (t == 0) ctrl->scan_base = sb + total;  // read again only after the next barrier
  }
  __syncthreads();
  // every thread derives the result itself (no read after the closing barrier: a fast thread may
  // already be appending again by then)
  const uint32_t sbv = ctrl->scan_base;
  const uint32_t c = sbv < s ? sbv : s;
  const uint64_t newT = (c == s && s > 0) ? buf[s - 1] : ctrl->T0;
  __syncthreads();
  if (t == 0) { ctrl->count = c; ctrl->sorted = c; ctrl->T = newT; ctrl->overflow = 0; }
  __syncthreads();
  return MergeResult{c, newT};
}

// One lane's view of a tile: 112 consecutive bases = 36 warm-up + 76 owned k-mer end positions,
// fetched straight from global memory as seven 16-byte loads (no LDS staging: the 38 KiB tile would
// cost most of the occupancy, and each line is still read from HBM once -- neighbouring lanes
// share lines through L2).
// `tile` points at the tile's first base (wave-uniform, lives in SGPRs); rq is the lane's offset
// relative to it and [gb, ge) the genome's extent in the same coordinates (clamped to +-2^30), so the
// lane keeps 32-bit offsets only and the load is an SGPR-base + VGPR-offset global_load_dwordx4.
constexpr int LOAD_BIAS = 64;  // offsets handed to the load are rq + LOAD_BIAS >= 0 (rq >= -4*WARM_DW)
__device__ __forceinline__ uint4 load_bases16(const uint8_t* __restrict__ tile, int rq, int gb, int ge) {
  if (rq >= gb && rq + 16 <= ge)
    return *reinterpret_cast<const uint4*>((tile - LOAD_BIAS) + (uint32_t)(rq + LOAD_BIAS));
  uint32_t ww[4];
#pragma unroll
  for (int d = 0; d < 4; d++) {
    uint32_t x = 0;
#pragma unroll
    for (int b = 0; b < 4; b++) {
      const int p = rq + 4 * d + b;
      const uint32_t ch = (p >= gb && p < ge) ? (tile - LOAD_BIAS)[(uint32_t)(p + LOAD_BIAS)] : (uint32_t)'N';
      x |= ch << (8 * b);
    }
    ww[d] = x;
  }
  return make_uint4(ww[0], ww[1], ww[2], ww[3]);
}

// values every lane holds identically (read from LDS after a barrier): move them to SGPRs so the
// compiler emits scalar branches and compares against scalar operands
__device__ __forceinline__ uint32_t uniform32(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint64_t uniform64(uint64_t v) {
  return ((uint64_t)uniform32((uint32_t)(v >> 32)) << 32) | uniform32((uint32_t)v);
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
  uint32_t expect;     // hashes the genome holds from earlier passes (0 in the first pass)
  uint32_t pass;
  uint32_t t0_used;    // the segments started from a threshold: fewer than s merged hashes flag the genome for a second walk
};

__global__ __launch_bounds__(WG) void merge_partials_kernel(const MergeJob* __restrict__ jobs,
                                                            const uint64_t* __restrict__ parts,
                                                            const uint32_t* __restrict__ pcnt, int cap,
                                                            uint64_t* out,
                                                            uint32_t* cnt, uint32_t* redo, int redo_run) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const lds_u64_ptr buf = (lds_u64_ptr)(lds_byte_ptr)smem;
  const lds_ctrl_ptr ctrl = (lds_ctrl_ptr)((lds_byte_ptr)smem + (size_t)cap * 8);
  const MergeJob jb = jobs[blockIdx.x];
  const int t = threadIdx.x;
  const uint32_t s = jb.sketch_size;
  if (jb.pass > 0 && cnt[jb.cnt_slot] != jb.expect) return;  // genome exhausted by earlier passes
  if (redo_run && redo[jb.cnt_slot] == 0) return;            // second launch: flagged genomes only
  if (t == 0) { ctrl->T = SENT; ctrl->T0 = SENT; ctrl->sorted = 0; ctrl->count = 0; ctrl->overflow = 0; ctrl->saw_max = 0; ctrl->scan_base = 0; }
  __syncthreads();
  uint32_t nmerged = 0;
  for (uint32_t p = 0; p < jb.nparts; p++) {
    const uint32_t pc = pcnt[jb.part_cnt0 + p];
    const uint64_t* src = parts + jb.part_off + (uint64_t)p * jb.stride;
    const uint32_t base = nmerged;
    __syncthreads();
    for (uint32_t i = t; i < pc; i += WG) {
      uint64_t v = src[i];
      if (v == SENT) ctrl->saw_max = 1;
      buf[base + i] = v;  // SENT entries are dropped by the merge
    }
    __syncthreads();
    if (t == 0) ctrl->count = base + pc;
    nmerged = merge_block(buf, ctrl, cap, s).count;
  }
  uint32_t n = nmerged;
  uint64_t* o = out + jb.out_off;
  for (uint32_t i = t; i < n; i += WG) o[i] = buf[i];
  if (t == 0) {
    if (ctrl->saw_max && n < s) { o[n] = SENT; n++; }
    cnt[jb.cnt_slot] = jb.expect + n;
    if (!redo_run && jb.t0_used && n < s) redo[jb.cnt_slot] = 1;  // the genome may hold hashes above its starting threshold
  }
}

inline int pow2ceil(int x) { int p = 1; while (p < x) p <<= 1; return p; }


// ---- host side: the segment plan and the launch sequence of a batch (shared by the two input formats) -------------
struct MinhashPlanInfo {   // handed to `prepare` once the segment table is on the device
  const Segment* d_segs;   // all passes' segments, pass by pass
  size_t nsegs;
  size_t lds;              // dynamic LDS of the sketch kernel
  bool packed_tables;      // table layout (lut_bytes)
};
struct MinhashLaunch {     // one launch of the sketch kernel
  const Segment* d_segs;   // first segment of the launch
  size_t seg_index;        // its index in the table handed to `prepare`
  uint32_t nseg;
  int pass, cap;
  size_t lds;
  bool packed_tables, runtime_k;   // runtime_k: the gated second walk over flagged genomes
  uint64_t* d_parts;
  uint32_t* d_pcnt;
  const uint32_t* d_redo;
};

// tile_bases: bases a workgroup takes per tile (the unit the segment lengths are planned in); min_room: candidate slots the
// kernel's buffer must offer beyond the sketch size.  prepare(MinhashPlanInfo)
// and launch(MinhashLaunch) return a status; everything is enqueued on the context stream.
template <class Prepare, class Launch>
int minhash_run(rtc_ctx* ctx, const uint64_t* h_off, uint32_t n, int k, const uint32_t* h_sizes, uint32_t size,
                uint64_t* d_out, uint32_t stride, uint32_t* d_cnt, uint64_t tile_bases, size_t min_room, Prepare&& prepare, Launch&& launch) {
  RTC_HIP(ctx, hipSetDevice(ctx->device));
  ctx->sketch_gen++;  // sketches on this context change: memos keyed on a sketch buffer are stale

  uint32_t smax = 0;
  for (uint32_t g = 0; g < n; g++) {
    uint32_t s = h_sizes ? h_sizes[g] : size;
    if (s > stride) return rtc_fail(ctx, RTC_ERR_ARG, "sketch size %u of genome %u exceeds stride %u", s, g, stride);
    smax = std::max(smax, s);
  }
  // One pass selects up to CHUNK hashes per genome in LDS; larger sketches take ceil(s/CHUNK) passes
  // over ascending hash ranges (pass p admits only hashes above everything kept so far).
  const uint32_t CHUNK = 6144;
  const uint32_t chunk_max = std::min(smax, CHUNK);
  const uint32_t npass = smax == 0 ? 1 : (smax + CHUNK - 1) / CHUNK;
  // Candidate buffer of the sketch kernel: s entries + room, as large as the LDS share of a workgroup
  // allows at the best occupancy that still leaves MIN_ROOM (3, 2 or 1 workgroups per CU; the sort
  // works on the live count, so the capacity need not be a power of two).  Partial-merge kernel:
  // two s-lists.
  int cap = 0, wgs_per_cu = 1;
  bool packed = false;  // table layout (lut_bytes): packed only where it buys a workgroup per CU
  int wgs_lo = 1, wgs_hi = 3;
  for (int wgs = wgs_hi; wgs >= wgs_lo && cap == 0; wgs--) {
    // 52 / 78 / 156 KiB: measured on MI355X, a 53.3 KiB allocation no longer runs three workgroups per CU
    const size_t share = ((size_t)156 * 1024 / wgs) & ~(size_t)2047;
    for (int pk = 0; pk < 2 && cap == 0; pk++) {
    // (tests: both layouts are exercised for every k by forcing one of them; either gives the same sketches)
    if (pk && (lut_his(k) == 0 || ctx->opt.sketch_no_packed)) break;
    if (!pk && ctx->opt.sketch_packed && lut_his(k) > 0) continue;
    const size_t fixed = lut_bytes(k, pk != 0) + ((sizeof(Ctrl) + 15) & ~(size_t)15) + QUEUE_BYTES;
    // (round 1 measured ~3000 entries of room as the break-even against lost occupancy; with the merge sorting
    // only the new candidates and the express walk a third workgroup per CU wins down to the minimum room:
    // s = 2000 at 10 000 x 5 Mbp 114.5 -> 102.1 ms, the containment sketches of config 4 190 -> 164 ms)
    const size_t want_room = min_room;  // what the kernel's safe mode may append between two looks at the count
    if (share > fixed && (share - fixed) / 8 >= (size_t)chunk_max + want_room) { cap = (int)((share - fixed) / 8); wgs_per_cu = wgs; packed = pk != 0; }
    }
  }
  if (cap == 0) return rtc_fail(ctx, RTC_ERR_UNSUPPORTED, "sketch chunk %u does not fit the LDS", chunk_max);
  // partial-sketch merge: two lists fit 2*chunk_max; twice that lets the rank merge work out of place
  const int cap_merge = (int)std::max<uint32_t>(std::max<uint32_t>(2 * chunk_max, std::min<uint32_t>(4 * chunk_max, 16384)), 1024);
  const size_t lds = (size_t)cap * 8 + lut_bytes(k, packed) + ((sizeof(Ctrl) + 15) & ~(size_t)15) + QUEUE_BYTES;
  const size_t lds_m = (size_t)cap_merge * 8 + sizeof(Ctrl);
  if (lds > (size_t)160 * 1024 || lds_m > (size_t)160 * 1024)
    return rtc_fail(ctx, RTC_ERR_UNSUPPORTED, "sketch chunk %u needs %zu B of LDS (> 160 KiB)", chunk_max, std::max(lds, lds_m));

  // ---- plan segments ----
  uint64_t total = 0;
  for (uint32_t g = 0; g < n; g++) {
    if (h_off[g + 1] < h_off[g]) return rtc_fail(ctx, RTC_ERR_ARG, "offsets not monotone at genome %u", g);
    total += h_off[g + 1] - h_off[g];
  }
  const uint64_t target_segs = (uint64_t)ctx->num_cu * 12;
  uint64_t seg_len = total / target_segs;
  const uint64_t min_seg = 4ull * tile_bases;
  if (seg_len < min_seg) seg_len = min_seg;
  const uint64_t seg_pref = 24ull * tile_bases;  // preferred segment of the few-genomes plan below

  // segments per genome: ~equal-length pieces of seg_len; 1 = the whole genome in one workgroup
  std::vector<uint32_t> nsv(n);
  uint32_t n_single = 0;
  for (uint32_t g = 0; g < n; g++) {
    const uint64_t len = h_off[g + 1] - h_off[g];
    uint64_t ns = (len + seg_len / 2) / seg_len;
    if (ns < 1) ns = 1;
    if (ns > 4096) ns = 4096;
    nsv[g] = (uint32_t)ns;
    if (ns == 1) n_single++;
  }
  // Whole-genome workgroups run in rounds of `slots` at a time; n mod slots leftover genomes would
  // occupy a nearly empty final round for a full round's duration (10 000 genomes on 768 slots:
  // 16 workgroups alone for 7 % of the kernel).  Cut the last leftover genomes into enough segments
  // to fill that round instead; they go through the partial path, launched after the full rounds.
  const uint32_t slots = (uint32_t)ctx->num_cu * (uint32_t)wgs_per_cu;
  if (n_single > slots) {
    uint32_t q = n_single % slots;
    if (q > 0 && q <= slots * 3 / 5) {
      const uint32_t want = slots / q;
      const uint64_t tail_max = 8;  // more pieces fill the round better but serialise in the per-genome merge (measured: 4-8 best)
      for (uint32_t g = n; g-- > 0 && q > 0;) {
        if (nsv[g] != 1) continue;
        const uint64_t len = h_off[g + 1] - h_off[g];
        const uint64_t ns2 = std::min<uint64_t>(std::min<uint64_t>(want, len / min_seg), tail_max);
        if (ns2 >= 2) nsv[g] = (uint32_t)ns2;
        q--;
      }
    }
  }

  // Few genomes (fewer whole-genome workgroups than the chip holds at a time): the segments are dealt out so that
  // their number is a whole multiple of the slots -- every genome gets its share of R x slots segments by largest
  // remainder, so the workgroups are of nearly equal length and the last round is as full as the first (rounding
  // each genome on its own left e.g. 2 000 x 5 Mbp with 4 000 workgroups for 768 slots: 5.2 rounds).
  if (n_single <= slots && total >= (uint64_t)slots * min_seg) {
    uint64_t rounds = total / ((uint64_t)slots * seg_pref);
    if (ctx->opt.sketch_rounds > 0) rounds = (uint64_t)ctx->opt.sketch_rounds;  // tuning
    rounds = std::max<uint64_t>(1, std::min<uint64_t>(rounds, total / ((uint64_t)slots * min_seg)));
    const uint64_t want = rounds * slots;
    std::vector<std::pair<double, uint32_t>> frac;
    frac.reserve(n);
    uint64_t given = 0;
    for (uint32_t g = 0; g < n; g++) {
      const uint64_t len = h_off[g + 1] - h_off[g];
      const double quota = (double)len * (double)want / (double)total;
      uint64_t ns = (uint64_t)quota;
      const uint64_t most = std::min<uint64_t>(std::max<uint64_t>(len / min_seg, 1), 4096);
      if (ns < 1) ns = 1;
      if (ns > most) ns = most;
      nsv[g] = (uint32_t)ns;
      given += ns;
      if (ns < most) frac.emplace_back(quota - (double)ns, g);
    }
    if (given < want) {
      std::sort(frac.begin(), frac.end(), [](const std::pair<double, uint32_t>& a, const std::pair<double, uint32_t>& b) {
        return a.first != b.first ? a.first > b.first : a.second < b.second;
      });
      for (size_t i = 0; i < frac.size() && given < want; i++) { nsv[frac[i].second]++; given++; }
    }
  }

  // starting threshold = factor x the expected s-th smallest hash (0: start from "everything passes").  3 keeps
  // the restart away down to genomes whose distinct k-mers are a third of their length, and every early tile lets
  // 3 s / N of its k-mers through instead of 8 s / N: 50 000 x 1 Mbp 115 -> 107 ms, config 4's sketches 163 -> 150 ms
  // Dense sketches (a genome of fewer than 2 500 k-mers per sketch hash: the containment sketches of clust-greedy, s = length / 1000)
  // start from 2x: there the candidates and their merges are 9 % of the kernel and a third fewer of them is worth more
  // than the margin (50 000 x 1 Mbp at s = 2000: 98.4 -> 92.3 ms, at s = 1000 83.0 -> 81.1 ms; 5 000 k-mers per hash: no difference).
  const int t0_fixed = ctx->opt.sketch_t0_factor;  // (-1: by rule) tests of the restart path / tuning
  auto start_threshold = [&](uint64_t len, uint32_t s) -> uint64_t {
    if (s == 0) return SENT;
    const uint64_t f = t0_fixed >= 0 ? (uint64_t)t0_fixed : (len / s <= 2500 ? 2u : 3u);
    if (f == 0 || len <= f * s) return SENT;
    return (uint64_t)((((unsigned __int128)1 << 64) * (f * s)) / len);
  };
  struct PassPlan { size_t direct0, ndirect, partial0, npartial, job0, njobs; };
  std::vector<PassPlan> plans(npass);
  std::vector<Segment> direct, partial;
  std::vector<MergeJob> jobs;
  uint64_t part_elems_max = 0;
  uint32_t part_slots_max = 0;
  bool any_partial_t0 = false;
  for (uint32_t ps = 0; ps < npass; ps++) {
    PassPlan& pl = plans[ps];
    pl.direct0 = direct.size(); pl.partial0 = partial.size(); pl.job0 = jobs.size();
    uint64_t part_elems = 0;
    uint32_t part_slots = 0;
    for (uint32_t g = 0; g < n; g++) {
      const uint64_t b = h_off[g], e = h_off[g + 1], len = e - b;
      const uint32_t sg_full = h_sizes ? h_sizes[g] : size;
      if (ps > 0 && sg_full <= ps * CHUNK) continue;  // this genome's sketch is complete
      const uint32_t s = std::min(CHUNK, sg_full - ps * CHUNK);
      const uint32_t expect = ps * CHUNK;
      const uint64_t out_off = (uint64_t)g * stride + expect;
      const uint64_t lo_off = ps ? out_off - 1 : 0;
      const uint64_t ns = nsv[g];
      if (ns == 1) {
        const uint64_t t0 = ps == 0 ? start_threshold(len, s) : SENT;
        direct.push_back(Segment{b, e, b, e, out_off, lo_off, g, s, g, expect, 0, 0, t0});
      } else {
        const uint64_t t0 = ps == 0 ? start_threshold(len, s) : SENT;  // the genome's starting threshold, shared by its segments
        if (t0 != SENT) any_partial_t0 = true;
        jobs.push_back(MergeJob{part_elems, part_slots, (uint32_t)ns, out_off, g, s, chunk_max, expect, ps, t0 != SENT ? 1u : 0u});
        for (uint64_t i = 0; i < ns; i++) {
          const uint64_t sb = b + len * i / ns, se = b + len * (i + 1) / ns;
          partial.push_back(Segment{b, e, sb, se, part_elems, lo_off, part_slots, s, g, expect, 1, 0, t0});
          part_elems += chunk_max;
          part_slots++;
        }
      }
    }
    pl.ndirect = direct.size() - pl.direct0; pl.npartial = partial.size() - pl.partial0; pl.njobs = jobs.size() - pl.job0;
    // workgroups are dispatched in table order: longest genomes first, so that the last round ends with the short ones
    // (results go to the genome's own row, the order is free)
    std::stable_sort(direct.begin() + pl.direct0, direct.end(),
                     [](const Segment& a, const Segment& b) { return a.g_end - a.g_begin > b.g_end - b.g_begin; });
    part_elems_max = std::max(part_elems_max, part_elems);
    part_slots_max = std::max(part_slots_max, part_slots);
  }
  // one segment table, pass by pass: [whole-genome segments | partial segments] -- a single launch per
  // pass walks both (workgroups are dispatched in index order, so the partial segments fill the
  // tail of the last whole-genome round)
  std::vector<Segment> segs;
  segs.reserve(direct.size() + partial.size());
  std::vector<size_t> seg0(npass);
  for (uint32_t ps = 0; ps < npass; ps++) {
    const PassPlan& pl = plans[ps];
    seg0[ps] = segs.size();
    segs.insert(segs.end(), direct.begin() + pl.direct0, direct.begin() + pl.direct0 + pl.ndirect);
    segs.insert(segs.end(), partial.begin() + pl.partial0, partial.begin() + pl.partial0 + pl.npartial);
  }
  const size_t bseg = segs.size() * sizeof(Segment);
  const size_t bjobs = jobs.size() * sizeof(MergeJob);
  void* ws0 = nullptr;
  const size_t bredo = any_partial_t0 ? (size_t)n * 4 : 0;
  RTC_TRY(rtc_ws(ctx, 0, bseg + bjobs + bredo + 64, &ws0));
  Segment* d_segs = (Segment*)ws0;
  MergeJob* d_jobs = (MergeJob*)((char*)ws0 + bseg);
  uint32_t* d_redo = (uint32_t*)((char*)ws0 + bseg + bjobs);  // per genome: walk its segments again without the threshold
  void* hp = nullptr;
  RTC_TRY(rtc_pinned(ctx, bseg + bjobs + 64, &hp));
  RTC_HIP(ctx, hipStreamSynchronize(ctx->stream));  // pinned staging may still be in flight
  memcpy(hp, segs.data(), bseg);
  memcpy((char*)hp + bseg, jobs.data(), bjobs);
  RTC_HIP(ctx, hipMemcpyAsync(ws0, hp, bseg + bjobs, hipMemcpyHostToDevice, ctx->stream));
  if (bredo) RTC_HIP(ctx, hipMemsetAsync(d_redo, 0, bredo, ctx->stream));

  uint64_t* d_parts = nullptr;
  uint32_t* d_pcnt = nullptr;
  if (!partial.empty()) {
    void* ws1 = nullptr;
    RTC_TRY(rtc_ws(ctx, 1, part_elems_max * 8 + (size_t)part_slots_max * 4 + 64, &ws1));
    d_parts = (uint64_t*)ws1;
    d_pcnt = (uint32_t*)((char*)ws1 + part_elems_max * 8);
  }


  MinhashPlanInfo pi{d_segs, segs.size(), lds, packed};
  RTC_TRY(prepare(pi));
  RTC_HIP(ctx, hipFuncSetAttribute((const void*)merge_partials_kernel,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_m));
  for (uint32_t ps = 0; ps < npass; ps++) {
    const PassPlan& pl = plans[ps];
    if (pl.ndirect + pl.npartial) {
      RTC_TRY(launch(MinhashLaunch{d_segs + seg0[ps], seg0[ps], (uint32_t)(pl.ndirect + pl.npartial), (int)ps, cap, lds, packed, false,
                                   d_parts, d_pcnt, (const uint32_t*)nullptr}));
    }
    if (pl.npartial) {
      hipLaunchKernelGGL(merge_partials_kernel, dim3((uint32_t)pl.njobs), dim3(WG), lds_m, ctx->stream,
                         d_jobs + pl.job0, d_parts, d_pcnt, cap_merge, d_out, d_cnt, d_redo, 0);
      RTC_CHECK_LAUNCH(ctx);
      if (ps == 0 && any_partial_t0) {
        // genomes the merge flagged (fewer than s hashes below the starting threshold): their segments once more
        // from "everything passes", merged again; every other workgroup of the two launches leaves at once
        // (the runtime-k instantiation: the flagged genomes are few, and the gated launch stays out of the compile-time-k
        // kernel's per-launch statistics)
        RTC_TRY(launch(MinhashLaunch{d_segs + seg0[ps] + pl.ndirect, seg0[ps] + pl.ndirect, (uint32_t)pl.npartial, (int)ps, cap, lds, packed,
                                     true, d_parts, d_pcnt, (const uint32_t*)d_redo}));
        hipLaunchKernelGGL(merge_partials_kernel, dim3((uint32_t)pl.njobs), dim3(WG), lds_m, ctx->stream,
                           d_jobs + pl.job0, d_parts, d_pcnt, cap_merge, d_out, d_cnt, d_redo, 1);
        RTC_CHECK_LAUNCH(ctx);
      }
    }
  }
  return RTC_OK;
}

}  // namespace

// rtc_oracle.cpp -- CPU restatement of the RabbitTClust hot path.  TEST INFRASTRUCTURE ONLY.
// See rtc_oracle.h for the parity status ("parity unpinned" for the MinHash k-mer hash).
// Every function cites the reference file:line (relative to /root/reference) it follows.
#include "rtc_oracle.h"

#include <omp.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <queue>
#include <unordered_map>
#include <unordered_set>
#include <vector>

// ------------------------------------------------------------------------------------------
// MurmurHash3_x64_128.  Public-domain algorithm (Austin Appleby, SMHasher); the reference tree
// only carries its licence notice (LICENSE.txt:16-18), the code itself sits in the absent
// RabbitSketch submodule.  Restated from the published algorithm description.
// ------------------------------------------------------------------------------------------
static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static inline uint64_t fmix64(uint64_t k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdULL;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ULL;
  k ^= k >> 33;
  return k;
}

extern "C" void orc_murmur3_x64_128(const void* key, int len, uint32_t seed, uint64_t out[2]) {
  const uint8_t* data = (const uint8_t*)key;
  const int nblocks = len / 16;
  uint64_t h1 = seed, h2 = seed;
  const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
  for (int i = 0; i < nblocks; i++) {
    uint64_t k1, k2;
    memcpy(&k1, data + 16 * i, 8);
    memcpy(&k2, data + 16 * i + 8, 8);
    k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
    h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
    k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
    h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
  }
  const uint8_t* tail = data + nblocks * 16;
  uint64_t k1 = 0, k2 = 0;
  switch (len & 15) {
    case 15: k2 ^= (uint64_t)tail[14] << 48; /* fallthrough */
    case 14: k2 ^= (uint64_t)tail[13] << 40; /* fallthrough */
    case 13: k2 ^= (uint64_t)tail[12] << 32; /* fallthrough */
    case 12: k2 ^= (uint64_t)tail[11] << 24; /* fallthrough */
    case 11: k2 ^= (uint64_t)tail[10] << 16; /* fallthrough */
    case 10: k2 ^= (uint64_t)tail[9] << 8;   /* fallthrough */
    case 9:  k2 ^= (uint64_t)tail[8];
      k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2; /* fallthrough */
    case 8:  k1 ^= (uint64_t)tail[7] << 56; /* fallthrough */
    case 7:  k1 ^= (uint64_t)tail[6] << 48; /* fallthrough */
    case 6:  k1 ^= (uint64_t)tail[5] << 40; /* fallthrough */
    case 5:  k1 ^= (uint64_t)tail[4] << 32; /* fallthrough */
    case 4:  k1 ^= (uint64_t)tail[3] << 24; /* fallthrough */
    case 3:  k1 ^= (uint64_t)tail[2] << 16; /* fallthrough */
    case 2:  k1 ^= (uint64_t)tail[1] << 8;  /* fallthrough */
    case 1:  k1 ^= (uint64_t)tail[0];
      k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
  }
  h1 ^= (uint64_t)len; h2 ^= (uint64_t)len;
  h1 += h2; h2 += h1;
  h1 = fmix64(h1); h2 = fmix64(h2);
  h1 += h2; h2 += h1;
  out[0] = h1; out[1] = h2;
}

// SMHasher's VerificationTest: hash keys {0},{0,1},...,{0..254} with seed 256-i, hash the
// concatenated results with seed 0, take the first 4 bytes little-endian.  Published value for
// MurmurHash3_x64_128 is 0x6384BA69.
extern "C" uint32_t orc_murmur3_smhasher_verification(void) {
  const int hashbytes = 16;
  uint8_t key[256], hashes[256 * 16], fin[16];
  memset(key, 0, sizeof key);
  for (int i = 0; i < 256; i++) {
    key[i] = (uint8_t)i;
    uint64_t o[2];
    orc_murmur3_x64_128(key, i, (uint32_t)(256 - i), o);
    memcpy(hashes + i * hashbytes, o, 16);
  }
  uint64_t o[2];
  orc_murmur3_x64_128(hashes, hashbytes * 256, 0, o);
  memcpy(fin, o, 16);
  return (uint32_t)fin[0] | ((uint32_t)fin[1] << 8) | ((uint32_t)fin[2] << 16) | ((uint32_t)fin[3] << 24);
}

// ------------------------------------------------------------------------------------------
// MinHash.  Restates the behaviour the reference expects of Sketch::MinHash (RabbitSketch,
// absent): call sites src/SketchInfo.cpp:918-924 (ctor), :942 (update per FASTA record),
// :969 / src/MST.cpp:1317 / src/greedy.cpp:1138 (storeMinHashes).  Algorithm = Mash's
// (LICENSE.txt:16-18,24-26 lineage): uppercase; a k-mer is valid iff all k chars are in ACGT;
// canonical = lexicographically smaller of k-mer and reverse complement; hash = first 64 bits
// of MurmurHash3_x64_128(ascii, k, seed) (first 32 bits when 4^k <= 2^32); keep the s smallest
// distinct hashes.  PARITY UNPINNED against upstream RabbitSketch.
// ------------------------------------------------------------------------------------------
static const int8_t* base_lut() {
  static int8_t lut[256];
  static bool init = false;
  if (!init) {
    for (int i = 0; i < 256; i++) lut[i] = 4;
    lut['A'] = lut['a'] = 0; lut['C'] = lut['c'] = 1;
    lut['G'] = lut['g'] = 2; lut['T'] = lut['t'] = 3;
    init = true;
  }
  return lut;
}

static const uint32_t* ascii4_lut();

struct orc_minhash {
  int k;
  uint32_t s;
  uint32_t seed;
  bool use64;
  std::priority_queue<uint64_t> heap;
  std::unordered_set<uint64_t> set;
};

extern "C" orc_minhash* orc_mh_new(int k, uint32_t sketch_size, uint32_t seed) {
  if (k < 1 || k > 32) return nullptr;
  orc_minhash* m = new orc_minhash();
  m->k = k; m->s = sketch_size; m->seed = seed;
  m->use64 = k > 16;  // Mash: 4^k > 2^32
  (void)base_lut();
  (void)ascii4_lut();
  return m;
}
extern "C" void orc_mh_free(orc_minhash* m) { delete m; }

// 4 bases (8 bits, first base in the top two bits) -> 4 ASCII bytes, first base lowest
static const uint32_t* ascii4_lut() {
  static uint32_t lut[256];
  static bool init = false;
  if (!init) {
    for (int e = 0; e < 256; e++) {
      uint32_t v = 0;
      for (int j = 0; j < 4; j++) v |= (uint32_t)(uint8_t)"ACGT"[(e >> (6 - 2 * j)) & 3] << (8 * j);
      lut[e] = v;
    }
    init = true;
  }
  return lut;
}

static inline uint64_t kmer_hash_packed(uint64_t canon, int k, uint32_t seed, bool use64) {
  // expand the 2-bit word to the ASCII bytes MurmurHash3 consumes, 4 bases per table lookup
  const uint32_t* a4 = ascii4_lut();
  const uint64_t x = k == 32 ? canon : canon << (64 - 2 * k);
  uint32_t w[8];
  for (int d = 0; d < 8; d++) w[d] = a4[(x >> (56 - 8 * d)) & 0xff];
  uint64_t o[2];
  orc_murmur3_x64_128(w, k, seed, o);  // only the first k bytes are read
  return use64 ? o[0] : (o[0] & 0xffffffffULL);
}

extern "C" uint64_t orc_mh_kmer_hash(const char* kmer, int k, uint32_t seed) {
  const int8_t* lut = base_lut();
  uint64_t f = 0, r = 0;
  for (int i = 0; i < k; i++) {
    uint64_t c = (uint64_t)lut[(uint8_t)kmer[i]];
    f = (f << 2) | c;
    r = (r >> 2) | ((3 - c) << (2 * (k - 1)));
  }
  return kmer_hash_packed(f < r ? f : r, k, seed, k > 16);
}

// ---- four k-mers at a time with AVX2 (ours: RabbitSketch's own AVX2/AVX512 MurmurHash3 kernels are
// not in the reference tree).  Same arithmetic as orc_murmur3_x64_128 on 64-bit lanes; the 64 x 64
// multiplies are assembled from 32 x 32 -> 64 products (AVX2 has no vpmullq).  16 <= k <= 32.
#if defined(__AVX2__)
#include <immintrin.h>
#define ORC_HAVE_AVX2 1
static inline __m256i mul64c(__m256i a, uint64_t c) {
  const __m256i cl = _mm256_set1_epi64x((long long)(c & 0xffffffffULL)), ch = _mm256_set1_epi64x((long long)(c >> 32));
  const __m256i ah = _mm256_srli_epi64(a, 32);
  const __m256i lo = _mm256_mul_epu32(a, cl);
  const __m256i cross = _mm256_add_epi64(_mm256_mul_epu32(a, ch), _mm256_mul_epu32(ah, cl));
  return _mm256_add_epi64(lo, _mm256_slli_epi64(cross, 32));
}
static inline __m256i rotl64v(__m256i x, int r) { return _mm256_or_si256(_mm256_slli_epi64(x, r), _mm256_srli_epi64(x, 64 - r)); }
static inline __m256i fmix64v(__m256i k) {
  k = _mm256_xor_si256(k, _mm256_srli_epi64(k, 33));
  k = mul64c(k, 0xff51afd7ed558ccdULL);
  k = _mm256_xor_si256(k, _mm256_srli_epi64(k, 33));
  k = mul64c(k, 0xc4ceb9fe1a85ec53ULL);
  return _mm256_xor_si256(k, _mm256_srli_epi64(k, 33));
}
static inline void kmer_hash_packed_x4(const uint64_t canon[4], int k, uint32_t seed, bool use64, uint64_t out[4]) {
  const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
  const uint32_t* a4 = ascii4_lut();
  alignas(32) uint64_t w[4][4];  // w[word][lane]: the k-mer's ASCII bytes as four little-endian u64 words
  for (int l = 0; l < 4; l++) {
    const uint64_t x = k == 32 ? canon[l] : canon[l] << (64 - 2 * k);
    for (int d = 0; d < 4; d++)
      w[d][l] = (uint64_t)a4[(x >> (56 - 16 * d)) & 0xff] | ((uint64_t)a4[(x >> (48 - 16 * d)) & 0xff] << 32);
  }
  auto bytemask = [](int nb) { return nb >= 8 ? ~0ULL : (nb <= 0 ? 0ULL : ((1ULL << (8 * nb)) - 1)); };
  __m256i h1 = _mm256_set1_epi64x((long long)(uint64_t)seed), h2 = h1;
  const __m256i W0 = _mm256_load_si256((const __m256i*)w[0]), W1 = _mm256_load_si256((const __m256i*)w[1]);
  __m256i W2 = _mm256_load_si256((const __m256i*)w[2]), W3 = _mm256_load_si256((const __m256i*)w[3]);
  auto body = [&](__m256i k1, __m256i k2) {
    k1 = mul64c(rotl64v(mul64c(k1, c1), 31), c2); h1 = _mm256_xor_si256(h1, k1);
    h1 = _mm256_add_epi64(rotl64v(h1, 27), h2);
    h1 = _mm256_add_epi64(_mm256_add_epi64(_mm256_slli_epi64(h1, 2), h1), _mm256_set1_epi64x(0x52dce729));
    k2 = mul64c(rotl64v(mul64c(k2, c2), 33), c1); h2 = _mm256_xor_si256(h2, k2);
    h2 = _mm256_add_epi64(rotl64v(h2, 31), h1);
    h2 = _mm256_add_epi64(_mm256_add_epi64(_mm256_slli_epi64(h2, 2), h2), _mm256_set1_epi64x(0x38495ab5));
  };
  body(W0, W1);
  if (k == 32) body(W2, W3);
  else {
    const int tail = k - 16;  // 0..15 bytes: k2 from bytes 8.., k1 from bytes 0..7 (src: the switch in orc_murmur3_x64_128)
    if (tail > 8) {
      W3 = _mm256_and_si256(W3, _mm256_set1_epi64x((long long)bytemask(tail - 8)));
      h2 = _mm256_xor_si256(h2, mul64c(rotl64v(mul64c(W3, c2), 33), c1));
    }
    if (tail > 0) {
      W2 = _mm256_and_si256(W2, _mm256_set1_epi64x((long long)bytemask(tail)));
      h1 = _mm256_xor_si256(h1, mul64c(rotl64v(mul64c(W2, c1), 31), c2));
    }
  }
  const __m256i len = _mm256_set1_epi64x(k);
  h1 = _mm256_xor_si256(h1, len); h2 = _mm256_xor_si256(h2, len);
  h1 = _mm256_add_epi64(h1, h2); h2 = _mm256_add_epi64(h2, h1);
  h1 = fmix64v(h1); h2 = fmix64v(h2);
  h1 = _mm256_add_epi64(h1, h2);
  _mm256_storeu_si256((__m256i*)out, h1);
  if (!use64) for (int l = 0; l < 4; l++) out[l] &= 0xffffffffULL;
}
// eight k-mers at a time with AVX-512 (vpmullq is a native 64 x 64 multiply); compiled for that target
// only and chosen at run time, the library itself stays x86-64-v3
#define ORC_T512 __attribute__((target("avx512f,avx512dq")))
ORC_T512 static inline __m512i rotl64w(__m512i x, int r) { return _mm512_rol_epi64(x, r); }
ORC_T512 static inline __m512i fmix64w(__m512i k) {
  k = _mm512_xor_si512(k, _mm512_srli_epi64(k, 33));
  k = _mm512_mullo_epi64(k, _mm512_set1_epi64((long long)0xff51afd7ed558ccdULL));
  k = _mm512_xor_si512(k, _mm512_srli_epi64(k, 33));
  k = _mm512_mullo_epi64(k, _mm512_set1_epi64((long long)0xc4ceb9fe1a85ec53ULL));
  return _mm512_xor_si512(k, _mm512_srli_epi64(k, 33));
}
ORC_T512 static void kmer_hash_packed_x8(const uint64_t canon[8], int k, uint32_t seed, bool use64, uint64_t out[8]) {
  const __m512i c1 = _mm512_set1_epi64((long long)0x87c37b91114253d5ULL), c2 = _mm512_set1_epi64((long long)0x4cf5ad432745937fULL);
  const uint32_t* a4 = ascii4_lut();
  alignas(64) uint64_t w[4][8];
  for (int l = 0; l < 8; l++) {
    const uint64_t x = k == 32 ? canon[l] : canon[l] << (64 - 2 * k);
    for (int d = 0; d < 4; d++)
      w[d][l] = (uint64_t)a4[(x >> (56 - 16 * d)) & 0xff] | ((uint64_t)a4[(x >> (48 - 16 * d)) & 0xff] << 32);
  }
  auto bytemask = [](int nb) { return nb >= 8 ? ~0ULL : (nb <= 0 ? 0ULL : ((1ULL << (8 * nb)) - 1)); };
  __m512i h1 = _mm512_set1_epi64((long long)(uint64_t)seed), h2 = h1;
  const __m512i W0 = _mm512_load_si512(w[0]), W1 = _mm512_load_si512(w[1]);
  __m512i W2 = _mm512_load_si512(w[2]), W3 = _mm512_load_si512(w[3]);
#define ORC_BODY8(K1, K2)                                                                                         \
  do {                                                                                                            \
    __m512i k1 = _mm512_mullo_epi64(rotl64w(_mm512_mullo_epi64(K1, c1), 31), c2); h1 = _mm512_xor_si512(h1, k1);   \
    h1 = _mm512_add_epi64(rotl64w(h1, 27), h2);                                                                   \
    h1 = _mm512_add_epi64(_mm512_add_epi64(_mm512_slli_epi64(h1, 2), h1), _mm512_set1_epi64(0x52dce729));          \
    __m512i k2 = _mm512_mullo_epi64(rotl64w(_mm512_mullo_epi64(K2, c2), 33), c1); h2 = _mm512_xor_si512(h2, k2);   \
    h2 = _mm512_add_epi64(rotl64w(h2, 31), h1);                                                                   \
    h2 = _mm512_add_epi64(_mm512_add_epi64(_mm512_slli_epi64(h2, 2), h2), _mm512_set1_epi64(0x38495ab5));          \
  } while (0)
  ORC_BODY8(W0, W1);
  if (k == 32) ORC_BODY8(W2, W3);
  else {
    const int tail = k - 16;
    if (tail > 8) {
      W3 = _mm512_and_si512(W3, _mm512_set1_epi64((long long)bytemask(tail - 8)));
      h2 = _mm512_xor_si512(h2, _mm512_mullo_epi64(rotl64w(_mm512_mullo_epi64(W3, c2), 33), c1));
    }
    if (tail > 0) {
      W2 = _mm512_and_si512(W2, _mm512_set1_epi64((long long)bytemask(tail)));
      h1 = _mm512_xor_si512(h1, _mm512_mullo_epi64(rotl64w(_mm512_mullo_epi64(W2, c1), 31), c2));
    }
  }
#undef ORC_BODY8
  const __m512i len = _mm512_set1_epi64(k);
  h1 = _mm512_xor_si512(h1, len); h2 = _mm512_xor_si512(h2, len);
  h1 = _mm512_add_epi64(h1, h2); h2 = _mm512_add_epi64(h2, h1);
  h1 = fmix64w(h1); h2 = fmix64w(h2);
  h1 = _mm512_add_epi64(h1, h2);
  _mm512_storeu_si512(out, h1);
  if (!use64) for (int l = 0; l < 8; l++) out[l] &= 0xffffffffULL;
}
static int orc_hash_lanes() {  // 8: AVX-512, 4: AVX2, 1: scalar (ORC_HASH_LANES overrides, for timing comparisons)
  static const int lanes = []() {
    int want = 8;
    if (const char* e = getenv("ORC_HASH_LANES")) want = atoi(e);
    __builtin_cpu_init();
    if (want >= 8 && __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq")) return 8;
    return want >= 4 ? 4 : 1;
  }();
  return lanes;
}
#else
#define ORC_HAVE_AVX2 0
static int orc_hash_lanes() { return 1; }
#endif
extern "C" int orc_minhash_impl_avx2(void) { return orc_hash_lanes(); }

static inline void mh_try_insert(orc_minhash* m, uint64_t h) {
  if (m->heap.size() < m->s || h < m->heap.top()) {
    if (m->set.insert(h).second) {
      m->heap.push(h);
      if (m->heap.size() > m->s) {
        m->set.erase(m->heap.top());
        m->heap.pop();
      }
    }
  }
}

extern "C" void orc_mh_update(orc_minhash* m, const char* seq, uint64_t len) {
  const int8_t* lut = base_lut();
  const int k = m->k;
  if (m->s == 0) return;
  const uint64_t mask = k == 32 ? ~0ULL : ((1ULL << (2 * k)) - 1);
  uint64_t f = 0, r = 0;
  int run = 0;
  const int lanes = k >= 16 ? orc_hash_lanes() : 1;  // k-mers per vector hash call (the order of insertion does not change the set)
  uint64_t pend[8], hv[8];
  int np = 0;
  for (uint64_t i = 0; i < len; i++) {
    int c = lut[(uint8_t)seq[i]];
    if (c > 3) { run = 0; continue; }
    f = ((f << 2) | (uint64_t)c) & mask;
    r = (r >> 2) | ((uint64_t)(3 - c) << (2 * (k - 1)));
    if (++run >= k) {
      uint64_t canon = f < r ? f : r;
      if (lanes > 1) {
        pend[np++] = canon;
        if (np == lanes) {
#if ORC_HAVE_AVX2
          if (lanes == 8) kmer_hash_packed_x8(pend, k, m->seed, m->use64, hv);
          else kmer_hash_packed_x4(pend, k, m->seed, m->use64, hv);
#endif
          for (int q = 0; q < lanes; q++) mh_try_insert(m, hv[q]);
          np = 0;
        }
      } else {
        uint64_t h = kmer_hash_packed(canon, k, m->seed, m->use64);
        mh_try_insert(m, h);
      }
    }
  }
  for (int q = 0; q < np; q++) mh_try_insert(m, kmer_hash_packed(pend[q], k, m->seed, m->use64));
}

extern "C" uint32_t orc_mh_store(const orc_minhash* m, uint64_t* out, uint32_t cap) {
  std::vector<uint64_t> v(m->set.begin(), m->set.end());
  std::sort(v.begin(), v.end());
  uint32_t n = (uint32_t)v.size();
  for (uint32_t i = 0; i < n && i < cap; i++) out[i] = v[i];
  return n;
}

extern "C" void orc_sketch_minhash_batch(const uint8_t* seq, const uint64_t* off, uint32_t n, int k,
                                         uint32_t seed, const uint32_t* sizes, uint64_t* out,
                                         uint32_t stride, uint32_t* cnt, int threads) {
  if (threads <= 0) threads = omp_get_max_threads();
  (void)base_lut();
#pragma omp parallel for num_threads(threads) schedule(dynamic)
  for (int64_t g = 0; g < (int64_t)n; g++) {
    orc_minhash* m = orc_mh_new(k, sizes[g], seed);
    // records are separated by non-ACGT bytes, which reset the k-mer window exactly like a
    // record boundary does (src/SketchInfo.cpp:928-948 calls update() once per record)
    orc_mh_update(m, (const char*)seq + off[g], off[g + 1] - off[g]);
    cnt[g] = orc_mh_store(m, out + (size_t)g * stride, stride);
    if (cnt[g] > stride) cnt[g] = stride;
    orc_mh_free(m);
  }
}

// ------------------------------------------------------------------------------------------
// KSSD.  src/SketchInfo.cpp:60-102 (shuffle table), :1019-1048 (parameters/masks),
// :1126-1165 (rolling loop), :1180-1193 (sort).
// ------------------------------------------------------------------------------------------
extern "C" void orc_kssd_params_init(int kmer_size, int drlevel, orc_kssd_params* p) {
  int half_k = (kmer_size + 1) / 2;                       // :1019
  p->half_k = half_k;
  p->kmer_size = half_k * 2;                              // :1020
  p->use64 = half_k - drlevel > 8 ? 1 : 0;                // :1021
  p->half_subk = 6 - drlevel >= 2 ? 6 : drlevel + 2;      // :1022
  p->drlevel = drlevel;
  p->dim_size = 1 << 4 * p->half_subk;                    // :1023
  p->dim_end = 1 << 4 * (p->half_subk - drlevel);         // :1025
  p->id = (half_k << 8) + (p->half_subk << 4) + drlevel;  // :1030
}

// :60-78 shuffle(), :80-89 shuffleN(), :91-102 generate_shuffle_dim().  Uses glibc
// srand()/rand() exactly as the reference does (process-global state; not thread safe).
static int* kssd_shuffle(int* arr, int length, unsigned seed) {
  srand(seed);
  for (int i = length - 1; i > 0; i--) {
    int j = rand() % (i + 1);
    int tmp = arr[i]; arr[i] = arr[j]; arr[j] = tmp;
  }
  return arr;
}
extern "C" int* orc_kssd_shuffle_dim(int half_subk) {
  int dim_size = 1 << 4 * half_subk;
  int* arr = (int*)malloc((size_t)dim_size * sizeof(int));
  for (int i = 0; i < dim_size; i++) arr[i] = i;
  kssd_shuffle(arr, dim_size, 23);
  kssd_shuffle(arr, dim_size, 348842630u);
  return arr;
}

extern "C" uint64_t orc_kssd_sketch(const orc_kssd_params* p, const int* shuffled_dim,
                                    const uint8_t* seq, uint64_t len, uint32_t* out32,
                                    uint64_t* out64, uint64_t cap) {
  const int half_k = p->half_k, half_subk = p->half_subk, drlevel = p->drlevel;
  const int kmerSize = p->kmer_size;
  const int dim_start = 0, dim_end = p->dim_end;
  const int comp_bittl = 64 - 4 * half_k;                 // :1039
  const int half_outctx_len = half_k - half_subk;         // :1040
  const int rev_add_move = 4 * half_k - 2;                // :1041
  const uint64_t tupmask = 0xffffffffffffffffULL >> comp_bittl;                            // :1044
  const uint64_t domask = (tupmask >> (4 * half_outctx_len)) << (2 * half_outctx_len);     // :1045
  const uint64_t undomask = (tupmask ^ domask) & tupmask;                                   // :1046
  const uint64_t undomask1 = undomask & (tupmask >> ((half_k + half_subk) * 2));            // :1047
  const uint64_t undomask0 = undomask ^ undomask1;                                          // :1048
  const int8_t* lut = base_lut();  // == BaseMap :1007-1017 for bytes < 128; >=128 invalid
  std::unordered_set<uint64_t> set;
  uint64_t tuple = 0, rvs = 0;
  int base = 1;
  for (uint64_t j = 0; j < len; j++) {
    int basenum = lut[seq[j]];
    if (basenum < 4) {
      tuple = ((tuple << 2) | (uint64_t)basenum) & tupmask;                                 // :1134
      rvs = (rvs >> 2) + (((uint64_t)basenum ^ 3ULL) << rev_add_move);                      // :1135
      base++;
      if (base > kmerSize) {                                                                // :1139
        uint64_t uni = tuple < rvs ? tuple : rvs;                                           // :1141
        uint32_t dim_id = (uint32_t)((uni & domask) >> (half_outctx_len * 2));              // :1142
        int sd = shuffled_dim[dim_id];
        if (sd >= dim_end || sd < dim_start) continue;                                      // :1054,1145-1148
        int pfilter = sd - dim_start;                                                       // :1149
        uint64_t dr = (((uni & undomask0) |
                        ((uni & undomask1) << (kmerSize * 2 - half_outctx_len * 4))) >>
                       (drlevel * 4)) | (uint64_t)pfilter;                                  // :1150-1152
        if (p->use64) set.insert(dr); else set.insert((uint64_t)(uint32_t)dr);              // :1154-1157
      }
    } else {
      base = 1; tuple = 0; rvs = 0;                                                         // :1160-1164
    }
  }
  std::vector<uint64_t> v(set.begin(), set.end());
  std::sort(v.begin(), v.end());                                                            // :1185,1192
  for (uint64_t i = 0; i < v.size() && i < cap; i++) {
    if (p->use64) out64[i] = v[i]; else out32[i] = (uint32_t)v[i];
  }
  return v.size();
}

// ------------------------------------------------------------------------------------------
// intersections and distances
// ------------------------------------------------------------------------------------------
template <typename T>
static uint32_t common_sorted(const T* a, uint32_t na, const T* b, uint32_t nb) {
  uint32_t i = 0, j = 0, c = 0;
  while (i < na && j < nb) {
    if (a[i] < b[j]) i++;
    else if (b[j] < a[i]) j++;
    else { c++; i++; j++; }
  }
  return c;
}
// Mash's union-truncated estimator (what MinHash::jaccard() is recalled to do, SURVEY.md Appendix B [U];
// call site src/MST.cpp:862-866): denominator = union elements seen, at most sketch_size
extern "C" void orc_mash_counts_u64(const uint64_t* a, uint32_t na, const uint64_t* b, uint32_t nb, uint32_t sketch_size,
                                    uint32_t* common, uint32_t* denom) {
  uint32_t i = 0, j = 0, c = 0, d = 0;
  while (d < sketch_size && i < na && j < nb) {
    if (a[i] < b[j]) i++;
    else if (b[j] < a[i]) j++;
    else { c++; i++; j++; }
    d++;
  }
  if (d < sketch_size) {
    uint32_t rest = (na - i) + (nb - j);
    d += std::min(rest, sketch_size - d);
  }
  *common = c; *denom = d;
}

extern "C" uint32_t orc_common_u64(const uint64_t* a, uint32_t na, const uint64_t* b, uint32_t nb) {
  return common_sorted(a, na, b, nb);
}
extern "C" uint32_t orc_common_u32(const uint32_t* a, uint32_t na, const uint32_t* b, uint32_t nb) {
  return common_sorted(a, na, b, nb);
}

// src/MST.cpp:26-37 calr(), :1292 "int radio = calr(threshold, kmer_size-1)"
extern "C" int orc_mst_radio(double threshold, int kmer_size) {
  return (int)(2.0 * std::exp(threshold * (kmer_size - 1)) - 1.0);
}

// src/MST.cpp:1295 inv_kmer_size, :1489-1515
extern "C" double orc_mst_distance(int common, int size0, int size1, int kmer_size,
                                   int is_containment) {
  const double inv_kmer_size = 1.0 / kmer_size;
  if (!is_containment) {
    int denom = size0 + size1 - common;
    double jaccard = denom == 0 ? 0.0 : (double)common / denom;
    if (jaccard == 1.0) return 0.0;
    if (jaccard == 0.0) return 1.0;
    double ratio = (2.0 * jaccard) / (1.0 + jaccard);
    return -inv_kmer_size * log(ratio);
  } else {
    int denom = size0 < size1 ? size0 : size1;
    double containment = denom == 0 ? 0.0 : (double)common / denom;
    if (containment == 1.0) return 0.0;
    if (containment == 0.0) return 1.0;
    return -inv_kmer_size * log(containment);
  }
}

// src/greedy.cpp:1245-1275
extern "C" double orc_greedy_distance(int common, int sizeRef, int sizeQry, int kmer_size,
                                      int repIsContainment) {
  double dist;
  if (repIsContainment) {
    int minSize = std::min(sizeRef, sizeQry);
    if (minSize == 0) return 1.0;
    double jaccard = (double)common / minSize;
    if (jaccard >= 1.0) dist = 0.0;
    else if (jaccard <= 0.0) dist = 1.0;
    else { dist = -log(2.0 * jaccard / (1.0 + jaccard)) / kmer_size; if (dist > 1.0) dist = 1.0; }
  } else {
    int denom = sizeRef + sizeQry - common;
    if (denom == 0) return 0.0;
    double jaccard = (double)common / denom;
    if (jaccard >= 1.0) dist = 0.0;
    else if (jaccard <= 0.0) dist = 1.0;
    else { dist = -log(2.0 * jaccard / (1.0 + jaccard)) / kmer_size; if (dist > 1.0) dist = 1.0; }
  }
  return dist;
}

// src/greedy.cpp:526-543
extern "C" double orc_kssd_greedy_distance(int common, int size0, int size1, int kmer_size) {
  int denom = size0 + size1 - common;
  if (size0 == 0 || size1 == 0 || denom == 0) return 1.0;
  double jaccard = (double)common / denom;
  if (jaccard == 1.0) return 0.0;
  if (jaccard == 0.0) return 1.0;
  double mashD = (double)-1.0 / kmer_size * log((2 * jaccard) / (1.0 + jaccard));
  return mashD > 1.0 ? 1.0 : mashD;
}

// ------------------------------------------------------------------------------------------
// inverted index (src/SketchInfo.h:95-161 MinHashInvertedIndex / :59-92 KssdInvertedIndex):
// hash -> genome ids in arrival (= id) order, built as at src/SketchInfo.cpp:968-973.
// ------------------------------------------------------------------------------------------
struct Sketches {
  const void* hashes; int width; const uint64_t* start; const uint32_t* len; uint32_t n;
  inline uint64_t at(uint32_t g, uint32_t i) const {
    return width == 8 ? ((const uint64_t*)hashes)[start[g] + i]
                      : (uint64_t)((const uint32_t*)hashes)[start[g] + i];
  }
};
typedef std::unordered_map<uint64_t, std::vector<uint32_t>> InvIndex;

static void build_index(const Sketches& S, InvIndex& idx) {
  uint64_t tot = 0;
  for (uint32_t g = 0; g < S.n; g++) tot += S.len[g];
  idx.reserve(tot);
  for (uint32_t g = 0; g < S.n; g++)
    for (uint32_t i = 0; i < S.len[g]; i++) idx[S.at(g, i)].push_back(g);
}

// epoch-stamped counting, src/MST.cpp:1392-1435: returns candidates in first-touch order
struct Counter {
  std::vector<int> inter, stamp, cand;
  int ep = 1;
  explicit Counter(uint32_t n) : inter(n, 0), stamp(n, 0) {}
  void run(const Sketches& S, const InvIndex& idx, uint32_t i) {
    cand.clear();
    ++ep;
    if (ep == INT_MAX) { std::fill(stamp.begin(), stamp.end(), 0); ep = 1; }
    for (uint32_t t = 0; t < S.len[i]; t++) {
      auto it = idx.find(S.at(i, t));
      if (it == idx.end()) continue;
      for (uint32_t cur : it->second) {
        if (stamp[cur] != ep) { stamp[cur] = ep; inter[cur] = 1; cand.push_back((int)cur); }
        else inter[cur] += 1;
      }
    }
  }
};

extern "C" uint64_t orc_candidate_pairs(const void* hashes, int width, const uint64_t* start,
                                        const uint32_t* len, uint32_t n, orc_cedge* out,
                                        uint64_t cap) {
  Sketches S{hashes, width, start, len, n};
  InvIndex idx;
  build_index(S, idx);
  Counter C(n);
  uint64_t m = 0;
  for (uint32_t i = 0; i < n; i++) {
    if (len[i] == 0) continue;
    C.run(S, idx, i);
    std::vector<int> c = C.cand;
    std::sort(c.begin(), c.end());
    for (int j : c) {
      if ((uint32_t)j >= i) continue;
      if (m < cap) out[m] = orc_cedge{(int)i, j, (uint32_t)C.inter[j]};
      m++;
    }
  }
  return m;
}

// ------------------------------------------------------------------------------------------
// UnionFind (src/UnionFind.h:5-90) and kruskalAlgorithm (src/MST.cpp:59-75)
// ------------------------------------------------------------------------------------------
struct UnionFind {
  std::vector<int> parent, ranks;
  explicit UnionFind(int n) : parent(n), ranks(n, 0) { for (int i = 0; i < n; i++) parent[i] = i; }
  int find(int e) {
    int r = e;
    while (parent[r] != r) r = parent[r];
    while (parent[e] != r) { int nx = parent[e]; parent[e] = r; e = nx; }  // path compression
    return r;
  }
  bool connected(int x, int y) { return find(x) == find(y); }
  void merge(int x, int y) {
    x = find(x); y = find(y);
    if (x != y) {
      if (ranks[x] > ranks[y]) parent[y] = x;
      else if (ranks[x] < ranks[y]) parent[x] = y;
      else { parent[x] = y; ranks[y]++; }
    }
  }
};

static std::vector<orc_edge> kruskal(const std::vector<orc_edge>& graph, int vertices) {
  UnionFind uf(vertices);
  std::vector<orc_edge> tree;
  if (graph.empty()) return tree;
  tree.push_back(graph[0]);
  uf.merge(graph[0].pre, graph[0].suf);
  for (size_t i = 1; i < graph.size(); i++) {
    if (!uf.connected(graph[i].pre, graph[i].suf)) {
      uf.merge(graph[i].pre, graph[i].suf);
      tree.push_back(graph[i]);
    }
  }
  return tree;
}
extern "C" uint64_t orc_kruskal(const orc_edge* sorted, uint64_t m, int vertices, orc_edge* out) {
  std::vector<orc_edge> g(sorted, sorted + m);
  std::vector<orc_edge> t = kruskal(g, vertices);
  for (size_t i = 0; i < t.size(); i++) out[i] = t[i];
  return t.size();
}
static bool cmpEdge(orc_edge a, orc_edge b) { return a.dist < b.dist; }  // src/MST.cpp:17-19

// ------------------------------------------------------------------------------------------
// compute_minhash_mst (src/MST.cpp:1290-1737) == compute_kssd_mst (:216-807) given hashes.
// ------------------------------------------------------------------------------------------
static void mst_row(const Sketches& S, const InvIndex& idx, Counter& C, uint32_t i, int radio,
                    int kmer_size, int is_containment, std::vector<orc_edge>& outv) {
  int size0 = (int)S.len[i];
  if (size0 == 0) return;                                             // :1404
  C.run(S, idx, i);                                                   // :1408-1435
  for (int j : C.cand) {                                              // :1467-1533
    if (j >= (int)i) continue;                                        // :1470
    int common = C.inter[j];
    int size1 = (int)S.len[j];
    if (size1 == 0) continue;                                         // :1477
    int mn = size0 < size1 ? size0 : size1, mx = size0 > size1 ? size0 : size1;
    if (mx > radio * mn) continue;                                    // :1481-1484
    outv.push_back(orc_edge{(int)i, j, orc_mst_distance(common, size0, size1, kmer_size, is_containment)});
  }
}

extern "C" uint64_t orc_mst(const void* hashes, int width, const uint64_t* start,
                            const uint32_t* len, uint32_t n, int kmer_size, int is_containment,
                            double threshold, int threads, orc_edge* out) {
  Sketches S{hashes, width, start, len, n};
  InvIndex idx;
  build_index(S, idx);
  if (threads < 1) threads = 1;
  const int radio = orc_mst_radio(threshold, kmer_size);              // :1292
  const int N = (int)n, subSize = 8;                                  // :1373
  const int tailNum = N % subSize;                                    // :1375 (start_index = 0)
  std::vector<std::vector<orc_edge>> mstArr(threads);
  std::vector<Counter*> ctr(threads);
  for (int t = 0; t < threads; t++) ctr[t] = new Counter(n);
#pragma omp parallel for num_threads(threads) schedule(dynamic)       // :1382
  for (int id = 0; id < N - tailNum; id += subSize) {
    int tid = omp_get_thread_num();
    for (int i = id; i < id + subSize; i++)
      mst_row(S, idx, *ctr[tid], (uint32_t)i, radio, kmer_size, is_containment, mstArr[tid]);
    std::sort(mstArr[tid].begin(), mstArr[tid].end(), cmpEdge);       // :1543
    std::vector<orc_edge> t = kruskal(mstArr[tid], N);                // :1544
    mstArr[tid].swap(t);
  }
  if (tailNum != 0) {                                                 // :1581-1701
    for (int i = N - tailNum; i < N; i++)
      mst_row(S, idx, *ctr[0], (uint32_t)i, radio, kmer_size, is_containment, mstArr[0]);
    if (!mstArr[0].empty()) {
      std::sort(mstArr[0].begin(), mstArr[0].end(), cmpEdge);
      std::vector<orc_edge> t = kruskal(mstArr[0], N);
      mstArr[0].swap(t);
    }
  }
  std::vector<orc_edge> fin;                                          // :1715-1723
  for (int t = 0; t < threads; t++) fin.insert(fin.end(), mstArr[t].begin(), mstArr[t].end());
  std::sort(fin.begin(), fin.end(), cmpEdge);
  std::vector<orc_edge> mst = kruskal(fin, N);
  for (int t = 0; t < threads; t++) delete ctr[t];
  for (size_t i = 0; i < mst.size(); i++) out[i] = mst[i];
  return mst.size();
}

// generateForest (src/MST.cpp:77-85) + generateClusterWithBfs (:109-142)
extern "C" uint32_t orc_forest_clusters(const orc_edge* mst, uint64_t m, double threshold,
                                        int vertices, int* order, uint32_t* cl_off) {
  std::vector<std::vector<int>> G(vertices);
  for (uint64_t e = 0; e < m; e++) {
    if (mst[e].dist <= threshold) {
      G[mst[e].pre].push_back(mst[e].suf);
      G[mst[e].suf].push_back(mst[e].pre);
    }
  }
  std::vector<char> visited(vertices, 0);
  uint32_t ncl = 0, pos = 0;
  cl_off[0] = 0;
  for (int i = 0; i < vertices; i++) {
    if (visited[i]) continue;
    visited[i] = 1;
    std::queue<int> Q;
    Q.push(i);
    order[pos++] = i;
    while (!Q.empty()) {
      int k = Q.front(); Q.pop();
      for (int v : G[k]) {
        if (visited[v]) continue;
        visited[v] = 1; Q.push(v); order[pos++] = v;
      }
    }
    cl_off[++ncl] = pos;
  }
  return ncl;
}

// ------------------------------------------------------------------------------------------
// MinHashGreedyClusterWithInvertedIndex, src/greedy.cpp:986-1399, at -t 1 (one thread: the
// best match is the FIRST best in touched order because the comparisons are strict, :1236,:1277).
// ------------------------------------------------------------------------------------------
extern "C" uint32_t orc_greedy_minhash(const uint64_t* hashes, const uint64_t* start,
                                       const uint32_t* len, const uint32_t* sketch_size_cfg,
                                       uint32_t n, int kmer_size, int is_containment,
                                       double threshold, int* rep_of) {
  if (n == 0) return 0;
  Sketches S{hashes, 8, start, len, n};
  InvIndex index_map;                                                 // rep-only index :1020
  auto add_rep = [&](uint32_t id) {                                   // :1037-1048
    for (uint32_t t = 0; t < S.len[id]; t++) index_map[S.at(id, t)].push_back(id);
  };
  uint32_t nrep = 1;
  rep_of[0] = 0;
  add_rep(0);
  int fixed_sketch_size = (int)sketch_size_cfg[0];                    // :1092
  bool all_fixed_size = true, all_standard_mode = !is_containment;    // :1093-1094
  for (uint32_t i = 1; i < std::min<uint32_t>(100, n); i++) {         // :1097-1103
    if (is_containment || (int)sketch_size_cfg[i] != fixed_sketch_size) {
      all_fixed_size = false; all_standard_mode = false; break;
    }
  }
  int fixed_common_min = 0;
  if (all_fixed_size && all_standard_mode) {                          // :1107-1115
    double x = std::exp(-threshold * kmer_size);
    double jaccard_min = x / (2.0 - x);
    fixed_common_min = (int)std::ceil(jaccard_min * (2 * fixed_sketch_size) / (1.0 + jaccard_min));
  }
  std::vector<uint32_t> cnt(n, 0), mark(n, 0), touched;
  uint32_t cur_mark = 0;
  for (uint32_t j = 1; j < n; j++) {                                  // :1136
    int sizeRef = (int)S.len[j];                                      // :1139
    cur_mark++; touched.clear();
    for (uint32_t t = 0; t < S.len[j]; t++) {                         // :1148-1164
      auto it = index_map.find(S.at(j, t));
      if (it == index_map.end()) continue;
      for (uint32_t rep : it->second) {
        if (mark[rep] != cur_mark) { mark[rep] = cur_mark; cnt[rep] = 1; touched.push_back(rep); }
        else cnt[rep]++;
      }
    }
    int best_common = -1, best_rep = -1;
    double best_dist = std::numeric_limits<double>::max();
    for (uint32_t rep : touched) {                                    // :1198-1283
      int common = (int)cnt[rep];
      int sizeQry = (int)sketch_size_cfg[rep];                        // :1201 getSketchSize()
      bool fast = all_fixed_size && all_standard_mode && !is_containment;
      int common_min;
      if (fast) common_min = fixed_common_min;
      else {
        double x = std::exp(-threshold * kmer_size);
        double jaccard_min = x / (2.0 - x);
        if (is_containment) common_min = (int)std::ceil(jaccard_min * std::min(sizeRef, sizeQry));
        else common_min = (int)std::ceil(jaccard_min * (sizeRef + sizeQry) / (1.0 + jaccard_min));
      }
      if (common < common_min) continue;                              // :1222
      if (fast) {
        if (common > best_common) { best_common = common; best_rep = (int)rep; }     // :1236
      } else {
        double dist = orc_greedy_distance(common, sizeRef, sizeQry, kmer_size, is_containment);
        if (dist <= threshold && dist < best_dist) {                  // :1277
          best_dist = dist; best_common = common; best_rep = (int)rep;
        }
      }
    }
    if (best_rep != -1) rep_of[j] = best_rep;                         // :1321-1325
    else { rep_of[j] = (int)j; nrep++; add_rep(j); }                  // :1327-1336
  }
  return nrep;
}

// KssdGreedyClusterWithInvertedIndex, src/greedy.cpp:566-899 at -t 1 on size-sorted input.
// The periodic prune (:690-702) only removes reps that can no longer reach jaccard_min, so it
// does not change the result and is not restated.
extern "C" uint32_t orc_greedy_kssd(const void* hashes, int width, const uint64_t* start,
                                    const uint32_t* len, uint32_t n, int kmer_size,
                                    double threshold, int* rep_of) {
  if (n == 0) return 0;
  Sketches S{hashes, width, start, len, n};
  InvIndex index_map;
  auto add_rep = [&](uint32_t id) {
    for (uint32_t t = 0; t < S.len[id]; t++) index_map[S.at(id, t)].push_back(id);
  };
  uint32_t nrep = 1;
  rep_of[0] = 0;
  add_rep(0);
  double x = std::exp(-threshold * kmer_size);                        // :652-653
  double jaccard_min = x / (2.0 - x);
  std::vector<uint32_t> cnt(n, 0), mark(n, 0), touched;
  uint32_t cur_mark = 0;
  for (uint32_t j = 1; j < n; j++) {
    int sizeRef = (int)S.len[j];
    cur_mark++; touched.clear();
    for (uint32_t t = 0; t < S.len[j]; t++) {
      auto it = index_map.find(S.at(j, t));
      if (it == index_map.end()) continue;
      for (uint32_t rep : it->second) {
        if (mark[rep] != cur_mark) { mark[rep] = cur_mark; cnt[rep] = 1; touched.push_back(rep); }
        else cnt[rep]++;
      }
    }
    double best_j = -1.0; int best_rep = -1;
    for (uint32_t rep : touched) {                                    // :765-794
      int common = (int)cnt[rep];
      int sizeQry = (int)S.len[rep];
      int common_min = (int)std::ceil(jaccard_min * (sizeRef + sizeQry) / (1.0 + jaccard_min));
      if (common < common_min) continue;
      int denom = sizeRef + sizeQry - common;
      double jac = denom == 0 ? 1.0 : (double)common / denom;
      if (jac > best_j) { best_j = jac; best_rep = (int)rep; }
    }
    if (best_rep != -1) rep_of[j] = best_rep;
    else { rep_of[j] = (int)j; nrep++; add_rep(j); }
  }
  return nrep;
}

// ------------------------------------------------------------------------------------------
// tune_parameters, src/sub_command.cpp:2383-2467
// ------------------------------------------------------------------------------------------
extern "C" orc_tune_result orc_tune_parameters(int greedy, int isSetKmer, int isContainment,
                                               int isJaccard, int kmerSize, double threshold,
                                               int containCompress, int sketchSize,
                                               uint64_t maxSize, uint64_t minSize,
                                               uint64_t averageSize) {
  orc_tune_result r{kmerSize, containCompress, isContainment, 1, 0.0};
  if (isContainment && isJaccard) { r.ok = 0; return r; }              // :2388-2391
  if (greedy) {                                                        // :2392-2408
    if (!isContainment && !isJaccard) { containCompress = (int)(averageSize / 1000); isContainment = 1; }
    else if (!isContainment && isJaccard) { }
    else if (averageSize / containCompress < 10) containCompress = (int)(averageSize / 1000);
  }
  double warning_rate = 0.01, recommend_rate = 0.0001;
  int recommendedKmerSize = (int)ceil(log(maxSize * (1 - recommend_rate) / recommend_rate) / log(4));
  int warningKmerSize = (int)ceil(log(maxSize * (1 - warning_rate) / warning_rate) / log(4));
  if (!isSetKmer) kmerSize = recommendedKmerSize;
  else if (kmerSize < warningKmerSize) kmerSize = recommendedKmerSize;
  else if (kmerSize > recommendedKmerSize + 3) kmerSize = recommendedKmerSize;
  double minJaccard;
  if (!isContainment) minJaccard = 1.0 / sketchSize;                   // :2435
  else minJaccard = 1.0 / (minSize / containCompress);                 // :2439 (integer division)
  double maxDist;
  if (minJaccard >= 1.0) maxDist = 1.0;
  else maxDist = -1.0 / kmerSize * log(2 * minJaccard / (1.0 + minJaccard));
  r.kmer_size = kmerSize; r.contain_compress = containCompress; r.is_containment = isContainment;
  r.max_dist = maxDist;
  if (threshold > maxDist) r.ok = 0;
  return r;
}

// ------------------------------------------------------------------------------------------
// deterministic synthetic genomes (definition shared with csrc/synth kernel)
// ------------------------------------------------------------------------------------------
// word `idx` of the splitmix64 stream started at `seed` (state = seed + (idx+1)*gamma)
static inline uint64_t splitmix64_at(uint64_t seed, uint64_t idx) {
  uint64_t z = seed + (idx + 1) * 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
extern "C" void orc_synth_genome(const orc_synth_desc* d, uint64_t pos0, uint64_t len, uint8_t* out) {
  for (uint64_t t = 0; t < len; t++) {
    uint64_t pos = pos0 + t;
    uint32_t b = (uint32_t)(splitmix64_at(d->fam_seed, pos >> 5) >> (2 * (pos & 31))) & 3;
    if (d->mut_thr) {
      uint32_t v = (uint32_t)(splitmix64_at(d->mut_seed, pos >> 2) >> (16 * (pos & 3))) & 0xFFFF;
      if ((v >> 2) < d->mut_thr) b = (b + 1 + ((v & 3) % 3)) & 3;
    }
    uint8_t c = (uint8_t)"ACGT"[b];
    if (d->n_every && pos >= d->n_every && (pos % d->n_every) < 8) c = 'N';
    out[t] = c;
  }
}

// rtc_bucket_sort.h -- the inverted join's sort, written for its shape (gfx950).
//
// What the join needs of its (hash, genome) records is the reference's index (src/MST.cpp:1408-1435): equal hashes side by
// side, genomes ascending inside a posting list.  The records arrive genome by genome with ascending hashes, K = 10^7 .. 10^8 of
// them, hashes uniform below the largest one.  A general LSD radix sort spends a pass per 8 key bits on that (rocPRIM: a
// histogram pass + 4 passes for u32 KSSD tuples, + 5 for the 40 bits a u64 MinHash set needs: 2.8 / 4.2 ms of the 6.1 / 9.2 ms pair
// phases of BASELINE configs[4] / [2] on one GPU).  Here:
//   * one or two stable partition passes over the TOP bits (up to 8 each) cut the records into 2^B groups of ~1 400: a tile of
//     4 096 records is ranked by wave votes (a lane's peers = the lanes of its 64-record step with the same digit, from one
//     ballot per digit bit), laid out by digit in LDS and written as contiguous runs;
//   * each group is then sorted on ALL its remaining bits inside LDS by one workgroup (the same ranking, 8 bits a pass, no
//     global traffic between the passes) and written once.
// Two or three trips through HBM instead of five or six; every step is stable, so genomes still ascend inside a posting list.
// A group that does not fit its workgroup's LDS (keys that are not spread evenly below the largest one) raises a flag and is
// left alone: the caller then sorts with the library instead -- correctness never depends on the distribution.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rtc_bsort {

constexpr int TILE = 4096, THREADS = 256, WAVES = THREADS / 64, STEPS = TILE / THREADS;  // a wave ranks STEPS steps of 64 consecutive records
constexpr int GROUP_MAX = 2048, GROUP_MEAN = 1400;  // records of a group the local sort takes / aims at
constexpr int GSTEPS = GROUP_MAX / THREADS;

__device__ __forceinline__ uint64_t lanes_below(uint32_t lane) { return (1ULL << lane) - 1ULL; }

// the lanes of this wave (among `valid`) whose digit equals mine
__device__ __forceinline__ uint64_t digit_peers(uint32_t d, int bits, uint64_t valid) {
  uint64_t peers = valid;
  for (int b = 0; b < bits; b++) {  // (uniform trip count)
    const uint64_t m = __ballot((d >> b) & 1u);
    peers &= ((d >> b) & 1u) ? m : ~m;
  }
  return peers;
}

// ---- partition pass, part 1: digits of a tile counted, hist[digit * ntiles + tile] --------------------------------------
template <typename T>
__global__ __launch_bounds__(THREADS) void hist_kernel(const T* __restrict__ keys, uint32_t K, int shift, int bits,
                                                       uint32_t ntiles, uint32_t* __restrict__ hist) {
  __shared__ uint32_t s_h[256];
  const uint32_t nb = 1u << bits, mask = nb - 1u;
  if (threadIdx.x < nb) s_h[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t base = blockIdx.x * TILE;
#pragma unroll
  for (int j = 0; j < STEPS; j++) {
    const uint32_t i = base + j * THREADS + threadIdx.x;
    if (i < K) atomicAdd(&s_h[(uint32_t)(keys[i] >> shift) & mask], 1u);
  }
  __syncthreads();
  if (threadIdx.x < nb) hist[(size_t)threadIdx.x * ntiles + blockIdx.x] = s_h[threadIdx.x];
}

// part 2: hist[digit][*] -> exclusive prefix over the tiles, total[digit]; one workgroup per digit
__global__ __launch_bounds__(THREADS) void scan_kernel(uint32_t* __restrict__ hist, uint32_t ntiles, uint32_t* __restrict__ total) {
  __shared__ uint32_t s_w[WAVES];
  __shared__ uint32_t s_carry;
  uint32_t* h = hist + (size_t)blockIdx.x * ntiles;
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  for (uint32_t t0 = 0; t0 < ntiles; t0 += THREADS) {  // (uniform trip count)
    const uint32_t t = t0 + threadIdx.x;
    const uint32_t v = t < ntiles ? h[t] : 0u;
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t u = (uint32_t)__shfl_up((int)incl, o); if ((int)lane >= o) incl += u; }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    uint32_t before = s_carry;
    for (uint32_t w = 0; w < wave; w++) before += s_w[w];
    if (t < ntiles) h[t] = before + incl - v;
    __syncthreads();
    if (threadIdx.x == THREADS - 1) s_carry = before + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) total[blockIdx.x] = s_carry;
}

// the ranking both sort kernels share: record `step` of this lane (digit d, taking part when `on`) among the records of its
// wave's chunk with the same digit, in record order.  wcnt: the wave's 256 counters in LDS, zero before the first step.
__device__ __forceinline__ uint32_t wave_rank(uint32_t d, int bits, bool on, uint32_t* wcnt, uint32_t lane) {
  const uint64_t valid = __ballot(on);
  uint32_t r = 0;
  if (on) {  // (the votes inside see the lanes that take part only)
    const uint64_t peers = digit_peers(d, bits, valid);
    const uint32_t pre = wcnt[d];  // every peer reads before the first of them writes: one wave, LDS in program order
    r = pre + (uint32_t)__popcll(peers & lanes_below(lane));
    if ((peers & lanes_below(lane)) == 0) wcnt[d] = pre + (uint32_t)__popcll(peers);
  }
  __builtin_amdgcn_wave_barrier();
  return r;
}

// per-digit offsets of a ranked tile / group: s_cnt[w][d] counts -> s_cnt[w][d] = records of digit d in the waves before w,
// s_excl[d] = records of the digits below d (+ all waves); nb <= 256 digits, THREADS = 256 threads
__device__ __forceinline__ void digit_offsets(uint32_t (*s_cnt)[256], uint32_t* s_excl, uint32_t* s_w, uint32_t nb) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t c = 0;
  if (threadIdx.x < nb) {
    uint32_t run = 0;
#pragma unroll
    for (int w = 0; w < WAVES; w++) { const uint32_t v = s_cnt[w][threadIdx.x]; s_cnt[w][threadIdx.x] = run; run += v; }
    c = run;
  }
  uint32_t incl = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const uint32_t u = (uint32_t)__shfl_up((int)incl, o); if ((int)lane >= o) incl += u; }
  if (lane == 63) s_w[wave] = incl;
  __syncthreads();
  uint32_t before = 0;
  for (uint32_t w = 0; w < wave; w++) before += s_w[w];
  if (threadIdx.x < nb) s_excl[threadIdx.x] = before + incl - c;
  __syncthreads();
}

// part 3: a tile ranked, laid out by digit in LDS, written as contiguous runs at (digit base + the tile's offset)
template <typename T>
__global__ __launch_bounds__(THREADS) void scatter_kernel(const T* __restrict__ keys, const uint32_t* __restrict__ vals, uint32_t K,
                                                          int shift, int bits, uint32_t ntiles, const uint32_t* __restrict__ hist,
                                                          const uint32_t* __restrict__ total, T* __restrict__ keys_out,
                                                          uint32_t* __restrict__ vals_out) {
  __shared__ T s_keys[TILE];
  __shared__ uint32_t s_vals[TILE];
  __shared__ uint32_t s_cnt[WAVES][256], s_excl[256], s_gbase[256], s_w[WAVES];
  const uint32_t nb = 1u << bits, mask = nb - 1u;
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int w = 0; w < WAVES; w++) s_cnt[w][threadIdx.x] = 0;
  {  // where each digit's records of THIS tile go: the digits below (all tiles) + this digit in the tiles before
    const uint32_t c = threadIdx.x < nb ? total[threadIdx.x] : 0u;
    uint32_t incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t u = (uint32_t)__shfl_up((int)incl, o); if ((int)lane >= o) incl += u; }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t w = 0; w < wave; w++) before += s_w[w];
    if (threadIdx.x < nb) s_gbase[threadIdx.x] = before + incl - c + hist[(size_t)threadIdx.x * ntiles + blockIdx.x];
    __syncthreads();
  }
  const uint32_t base = blockIdx.x * TILE + wave * (TILE / WAVES);
  T k[STEPS];
  uint32_t v[STEPS], r[STEPS];
#pragma unroll
  for (int j = 0; j < STEPS; j++) {
    const uint32_t i = base + j * 64 + lane;
    k[j] = i < K ? keys[i] : (T)0;
    v[j] = i < K ? vals[i] : 0u;
  }
#pragma unroll
  for (int j = 0; j < STEPS; j++) {
    const uint32_t i = base + j * 64 + lane;
    r[j] = wave_rank((uint32_t)(k[j] >> shift) & mask, bits, i < K, s_cnt[wave], lane);
  }
  __syncthreads();
  digit_offsets(s_cnt, s_excl, s_w, nb);
#pragma unroll
  for (int j = 0; j < STEPS; j++) {
    const uint32_t i = base + j * 64 + lane;
    if (i < K) {
      const uint32_t d = (uint32_t)(k[j] >> shift) & mask;
      const uint32_t p = s_excl[d] + s_cnt[wave][d] + r[j];
      s_keys[p] = k[j];
      s_vals[p] = v[j];
    }
  }
  __syncthreads();
  const uint32_t n = min((uint32_t)TILE, K - blockIdx.x * TILE);
  for (uint32_t p = threadIdx.x; p < n; p += THREADS) {
    const T key = s_keys[p];
    const uint32_t d = (uint32_t)(key >> shift) & mask;
    const uint32_t dst = s_gbase[d] + (p - s_excl[d]);
    keys_out[dst] = key;
    vals_out[dst] = s_vals[p];
  }
}

// gstart[g] = first record whose top bits (key >> shift) are >= g, for g in [0, ngroups]; the records ascend in those bits
template <typename T>
__global__ __launch_bounds__(THREADS) void bounds_kernel(const T* __restrict__ keys, uint32_t K, int shift, uint32_t ngroups,
                                                         uint32_t* __restrict__ gstart) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= K) return;
  const uint64_t p = (uint64_t)(keys[i] >> shift);
  const uint64_t q = i ? (uint64_t)(keys[i - 1] >> shift) + 1 : 0;  // the first group that starts here
  for (uint64_t g = q; g <= p && g < ngroups; g++) gstart[g] = i;
  if (i == K - 1)
    for (uint64_t g = p + 1; g <= ngroups; g++) gstart[g] = K;
}

// a group sorted on its low `lo_bits` bits inside LDS (stable LSD passes of up to 8 bits), written to the other buffer
template <typename T>
__global__ __launch_bounds__(THREADS) void local_kernel(const T* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                        const uint32_t* __restrict__ gstart, int lo_bits, T* __restrict__ keys_out,
                                                        uint32_t* __restrict__ vals_out, uint32_t* __restrict__ too_big) {
  __shared__ T s_k[2][GROUP_MAX];
  __shared__ uint32_t s_v[2][GROUP_MAX];
  __shared__ uint32_t s_cnt[WAVES][256], s_excl[256], s_w[WAVES];
  const uint32_t a = gstart[blockIdx.x], n = gstart[blockIdx.x + 1] - a;
  if (n == 0) return;
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (n > (uint32_t)GROUP_MAX) {  // (uniform) left as it is; the caller sorts with the library instead
    if (threadIdx.x == 0) atomicOr(too_big, 1u);
    return;
  }
  if (n == 1) {
    if (threadIdx.x == 0) { keys_out[a] = keys[a]; vals_out[a] = vals[a]; }
    return;
  }
  for (uint32_t p = threadIdx.x; p < n; p += THREADS) { s_k[0][p] = keys[a + p]; s_v[0][p] = vals[a + p]; }
  // a wave ranks a contiguous quarter of the group (whole steps of 64), in record order
  const uint32_t steps = (n + THREADS - 1) / THREADS;          // per wave
  const uint32_t chunk = steps * 64, first = wave * chunk;
  int cur = 0;
  for (int bit = 0; bit < lo_bits; bit += 8) {  // (uniform)
    const int bits = min(8, lo_bits - bit);
    const uint32_t nb = 1u << bits, mask = nb - 1u;
    for (int w = 0; w < WAVES; w++) s_cnt[w][threadIdx.x] = 0;
    __syncthreads();
    uint32_t r[GSTEPS];
#pragma unroll
    for (int j = 0; j < GSTEPS; j++) {
      if ((uint32_t)j < steps) {  // (uniform)
        const uint32_t p = first + j * 64 + lane;
        const bool on = p < n;
        const uint32_t d = on ? (uint32_t)(s_k[cur][p] >> bit) & mask : 0u;
        r[j] = wave_rank(d, bits, on, s_cnt[wave], lane);
      }
    }
    __syncthreads();
    digit_offsets(s_cnt, s_excl, s_w, nb);
#pragma unroll
    for (int j = 0; j < GSTEPS; j++) {
      if ((uint32_t)j < steps) {
        const uint32_t p = first + j * 64 + lane;
        if (p < n) {
          const T key = s_k[cur][p];
          const uint32_t d = (uint32_t)(key >> bit) & mask;
          const uint32_t q = s_excl[d] + s_cnt[wave][d] + r[j];
          s_k[cur ^ 1][q] = key;
          s_v[cur ^ 1][q] = s_v[cur][p];
        }
      }
    }
    __syncthreads();
    cur ^= 1;
  }
  for (uint32_t p = threadIdx.x; p < n; p += THREADS) { keys_out[a + p] = s_k[cur][p]; vals_out[a + p] = s_v[cur][p]; }
}

// ---- host side -----------------------------------------------------------------------------------------------------------
struct Plan {
  int passes = 0;        // partition passes over the top bits (0 .. 3)
  int bits[3] = {0, 0, 0};
  int top_bits = 0;      // their sum: the records end up in 2^top_bits groups
  int lo_bits = 0;       // bits the local sort handles (0: none, the groups are posting lists already)
  uint32_t ntiles = 0;
  size_t scratch = 0;    // bytes: histogram + totals + group starts + flag
};
inline Plan make_plan(uint64_t K, unsigned end_bit) {
  Plan p;
  int B = 0;
  while (B < 24 && (K >> B) > (uint64_t)GROUP_MEAN) B++;
  if (B > (int)end_bit) B = (int)end_bit;
  p.top_bits = B;
  p.passes = (B + 7) / 8;
  for (int i = 0; i < p.passes; i++) p.bits[i] = B / p.passes + (i < B % p.passes ? 1 : 0);
  p.lo_bits = (int)end_bit - B;
  p.ntiles = (uint32_t)((K + TILE - 1) / TILE);
  const size_t b_hist = (((size_t)256 * p.ntiles * 4) + 255) & ~(size_t)255;
  const size_t b_gs = ((((size_t)1 << B) + 2) * 4 + 255) & ~(size_t)255;
  p.scratch = b_hist + 1024 + b_gs + 256;
  return p;
}

// Sorts the K records in (k_a, v_a) by key, stable; (k_b, v_b) is the other buffer of the same size.  The result is in the
// buffer the call names through *in_a (true: (k_a, v_a)).  d_flag (inside `scratch`) is raised when a group was left unsorted.
// Nothing is synchronised.  Returns hipSuccess or the first launch error.
template <typename T>
hipError_t sort_pairs(const Plan& P, T* k_a, uint32_t* v_a, T* k_b, uint32_t* v_b, uint32_t K, unsigned end_bit, void* scratch,
                      hipStream_t s, bool* in_a, uint32_t** d_flag_out) {
  const size_t b_hist = (((size_t)256 * P.ntiles * 4) + 255) & ~(size_t)255;
  uint32_t* d_hist = (uint32_t*)scratch;
  uint32_t* d_total = (uint32_t*)((char*)scratch + b_hist);
  uint32_t* d_gstart = (uint32_t*)((char*)scratch + b_hist + 1024);
  const size_t b_gs = ((((size_t)1 << P.top_bits) + 2) * 4 + 255) & ~(size_t)255;
  uint32_t* d_flag = (uint32_t*)((char*)scratch + b_hist + 1024 + b_gs);
  *d_flag_out = d_flag;
  hipError_t e = hipMemsetAsync(d_flag, 0, 4, s);
  if (e != hipSuccess) return e;
  T* ki = k_a; uint32_t* vi = v_a; T* ko = k_b; uint32_t* vo = v_b;
  bool a = true;
  int shift = (int)end_bit - P.top_bits;
  for (int pass = 0; pass < P.passes; pass++) {  // LSD over the top bits: lowest of them first
    const int bits = P.bits[pass];
    hipLaunchKernelGGL(hist_kernel<T>, dim3(P.ntiles), dim3(THREADS), 0, s, (const T*)ki, K, shift, bits, P.ntiles, d_hist);
    hipLaunchKernelGGL(scan_kernel, dim3(1u << bits), dim3(THREADS), 0, s, d_hist, P.ntiles, d_total);
    hipLaunchKernelGGL(scatter_kernel<T>, dim3(P.ntiles), dim3(THREADS), 0, s, (const T*)ki, (const uint32_t*)vi, K, shift, bits, P.ntiles,
                       (const uint32_t*)d_hist, (const uint32_t*)d_total, ko, vo);
    std::swap(ki, ko); std::swap(vi, vo); a = !a;
    shift += bits;
  }
  if (P.lo_bits > 0) {
    const uint32_t ngroups = 1u << P.top_bits;
    hipLaunchKernelGGL(bounds_kernel<T>, dim3((K + THREADS - 1) / THREADS), dim3(THREADS), 0, s, (const T*)ki, K, (int)end_bit - P.top_bits, ngroups, d_gstart);
    hipLaunchKernelGGL(local_kernel<T>, dim3(ngroups), dim3(THREADS), 0, s, (const T*)ki, (const uint32_t*)vi, (const uint32_t*)d_gstart,
                       P.lo_bits, ko, vo, d_flag);
    a = !a;
  }
  *in_a = a;
  return hipGetLastError();
}

}  // namespace rtc_bsort

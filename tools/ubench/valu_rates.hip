// Microbenchmark: issue cost (cycles per wave-instruction per SIMD) of the integer ops the sketch
// kernel leans on.  8 chains per lane cross-feeding each other (so the compiler cannot fold the
// loop), 8 waves/SIMD resident -> throughput bound.  Cycles quoted at the measured clock.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITER 32768
#define HIPCHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e)); return; } } while (0)
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed, long long* clk) {
  uint32_t a[8]; uint64_t q[8];
  for (int i = 0; i < 8; i++) { a[i] = seed + threadIdx.x * 17 + i * 0x01010101u; q[i] = ((uint64_t)a[i] << 32) | (a[i] * 7u + 1); }
  const uint64_t c64 = 0x87c37b91114253d5ULL + seed;
  long long t0 = clock64();
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const uint32_t b = a[(i + 1) & 7]; const uint64_t p = q[(i + 1) & 7];
      if (OP == 0) a[i] = a[i] + b;                                    // v_add_u32
      if (OP == 1) a[i] = a[i] * b;                                    // v_mul_lo_u32
      if (OP == 2) a[i] = __umulhi(a[i], b);                           // v_mul_hi_u32
      if (OP == 3) q[i] = (uint64_t)(uint32_t)q[i] * (uint32_t)p + q[i]; // v_mad_u64_u32
      if (OP == 4) q[i] = (q[i] ^ p) * c64;                            // xor64 + full 64x64 multiply by constant
      if (OP == 5) q[i] = (q[i] << 3) + p;                             // v_lshl_add_u64
      if (OP == 6) q[i] = ((q[i] << 31) | (q[i] >> 33)) ^ p;           // rotl64 + xor64
      if (OP == 7) a[i] = __builtin_amdgcn_perm(a[i], b, a[i]);        // v_perm_b32
      if (OP == 8) a[i] = (a[i] & 0xffffffu) * (b & 0xffffffu);          // v_mul_u32_u24
      if (OP == 9) a[i] = a[i] ^ b;                                    // v_xor_b32
      if (OP == 10) a[i] = (a[i] >> 3) ^ b;                            // shift + xor (maybe fused)
      if (OP == 11) q[i] = q[i] >> (b & 31);                           // v_lshrrev_b64
      if (OP == 12) q[i] = q[i] + p;                                   // add64
      if (OP == 13) a[i] = a[i] > b ? a[i] - b : b;                    // cmp + cndmask/sub
      if (OP == 14) a[i] = __builtin_amdgcn_ubfe(a[i], 3, 9) + b;      // v_bfe_u32 + add
    }
  }
  long long t1 = clock64();
  uint32_t r = 0;
  for (int i = 0; i < 8; i++) r ^= a[i] ^ (uint32_t)q[i] ^ (uint32_t)(q[i] >> 32);
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (blockIdx.x == 0 && threadIdx.x == 0) *clk = t1 - t0;
}
template <int OP> void run(const char* name) {
  uint32_t* d; long long* dc; HIPCHK(hipMalloc(&d, 256 * 8 * 256 * 4)); HIPCHK(hipMalloc(&dc, 8));
  hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  dim3 grid(256 * 8), block(256);   // 8 blocks of 4 waves per CU = 8 waves/SIMD
  k<OP><<<grid, block>>>(d, 1, dc); HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipEventRecord(e0)); k<OP><<<grid, block>>>(d, 2, dc); HIPCHK(hipEventRecord(e1)); HIPCHK(hipEventSynchronize(e1));
  float ms; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  long long clk; HIPCHK(hipMemcpy(&clk, dc, 8, hipMemcpyDeviceToHost));
  double ops = 8.0 * ITER * 8;           // wave-ops per SIMD
  printf("%-16s %8.3f ms  -> %.2f cycles per (wave-)op per SIMD at 2.4 GHz\n", name, ms, ms * 1e-3 * 2.4e9 / ops);
  (void)ops; HIPCHK(hipFree(d)); HIPCHK(hipFree(dc));
}
int main() {
  run<0>("v_add_u32"); run<9>("v_xor_b32"); run<10>("shift+xor"); run<13>("cmp+sel"); run<14>("bfe+add");
  run<1>("v_mul_lo_u32"); run<2>("v_mul_hi_u32"); run<8>("v_mul_u32_u24"); run<3>("v_mad_u64_u32");
  run<4>("xor64+mul64"); run<5>("v_lshl_add_u64"); run<12>("add64"); run<6>("rotl64+xor64"); run<11>("v_lshrrev_b64"); run<7>("v_perm_b32");
  return 0;
}

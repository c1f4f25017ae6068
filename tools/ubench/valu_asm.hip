// Per-instruction issue cost on gfx950 (cycles per wave64 instruction per SIMD), measured with
// inline asm so the compiler cannot substitute instructions.  8 chains x 8 waves/SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITER 16384
#define HIPCHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e)); return; } } while (0)

#define KERNEL32(NAME, ASM)                                                                   \
  __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t seed) {                 \
    uint32_t a[8];                                                                            \
    for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 17 + i * 0x01010101u;             \
    for (int it = 0; it < ITER; it++) {                                                       \
      _Pragma("unroll") for (int i = 0; i < 8; i++) {                                         \
        asm volatile(ASM : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));            \
      }                                                                                       \
    }                                                                                         \
    uint32_t r = 0; for (int i = 0; i < 8; i++) r ^= a[i];                                    \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                           \
  }
#define KERNEL64(NAME, ASM)                                                                   \
  __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t seed) {                 \
    uint64_t a[8];                                                                            \
    for (int i = 0; i < 8; i++) a[i] = ((uint64_t)(seed + threadIdx.x * 17) << 32) | (i * 0x01010101u + seed); \
    for (int it = 0; it < ITER; it++) {                                                       \
      _Pragma("unroll") for (int i = 0; i < 8; i++) {                                         \
        asm volatile(ASM : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));            \
      }                                                                                       \
    }                                                                                         \
    uint32_t r = 0; for (int i = 0; i < 8; i++) r ^= (uint32_t)a[i] ^ (uint32_t)(a[i] >> 32); \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                           \
  }

KERNEL32(k_xor, "v_xor_b32 %0, %0, %1")
KERNEL32(k_and, "v_and_b32 %0, %0, %1")
KERNEL32(k_add, "v_add_u32 %0, %0, %1")
KERNEL32(k_sub, "v_sub_u32 %0, %0, %1")
KERNEL32(k_mov, "v_mov_b32 %0, %1")
KERNEL32(k_lshl, "v_lshlrev_b32 %0, 3, %1")
KERNEL32(k_lshr, "v_lshrrev_b32 %0, 3, %1")
KERNEL32(k_lshl_or, "v_lshl_or_b32 %0, %1, 5, %2")
KERNEL32(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
KERNEL32(k_or3, "v_or3_b32 %0, %0, %1, %2")
KERNEL32(k_add3, "v_add3_u32 %0, %0, %1, %2")
KERNEL32(k_lshl_add, "v_lshl_add_u32 %0, %1, 2, %2")
KERNEL32(k_xad, "v_xad_u32 %0, %0, %1, %2")
KERNEL32(k_bfe, "v_bfe_u32 %0, %1, 5, 8")
KERNEL32(k_bfi, "v_bfi_b32 %0, %0, %1, %2")
KERNEL32(k_alignbit, "v_alignbit_b32 %0, %0, %1, 7")
KERNEL32(k_alignbyte, "v_alignbyte_b32 %0, %0, %1, 1")
KERNEL32(k_perm, "v_perm_b32 %0, %0, %1, %2")
KERNEL32(k_bitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96")
KERNEL32(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL32(k_cmp, "v_cmp_lt_u32 vcc, %0, %1")
KERNEL32(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
KERNEL32(k_mul_hi, "v_mul_hi_u32 %0, %0, %1")
KERNEL32(k_mul_u24, "v_mul_u32_u24 %0, %0, %1")
KERNEL32(k_mad_u24, "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL32(k_bcnt, "v_bcnt_u32_b32 %0, %0, %1")
__global__ __launch_bounds__(256) void k_mad64(uint32_t* out, uint32_t seed) {
  uint64_t a[8]; uint32_t b[8];
  for (int i = 0; i < 8; i++) { a[i] = ((uint64_t)(seed + threadIdx.x * 17) << 32) | (i * 0x01010101u + seed); b[i] = seed * 31 + i + threadIdx.x; }
  for (int it = 0; it < ITER; it++) {
    _Pragma("unroll") for (int i = 0; i < 8; i++) {
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(b[(i + 1) & 7]), "v"(b[(i + 2) & 7]) : "vcc");
    }
  }
  uint32_t r = 0; for (int i = 0; i < 8; i++) r ^= (uint32_t)a[i] ^ (uint32_t)(a[i] >> 32);
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
KERNEL64(k_lshl64, "v_lshlrev_b64 %0, 3, %1")
KERNEL64(k_lshr64, "v_lshrrev_b64 %0, 3, %1")
KERNEL64(k_lshl_add64, "v_lshl_add_u64 %0, %1, 2, %2")
KERNEL64(k_cmp64, "v_cmp_lt_u64 vcc, %0, %1")
KERNEL64(k_mov64, "v_mov_b64 %0, %1")

typedef void (*kern_t)(uint32_t*, uint32_t);
static void run(kern_t fn, const char* name, double baseline_ms) {
  uint32_t* d; HIPCHK(hipMalloc(&d, 256 * 8 * 256 * 4));
  hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  dim3 grid(256 * 8), block(256);
  hipLaunchKernelGGL(fn, grid, block, 0, 0, d, 1u); HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipEventRecord(e0)); hipLaunchKernelGGL(fn, grid, block, 0, 0, d, 2u); HIPCHK(hipEventRecord(e1)); HIPCHK(hipEventSynchronize(e1));
  float ms; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  double ops = 8.0 * ITER * 8;
  printf("%-18s %8.3f ms  %5.2f cycles/instr/SIMD @2.4GHz nominal\n", name, ms, ms * 1e-3 * 2.4e9 / ops);
  HIPCHK(hipFree(d));
}
int main() {
#define R(k) run(k, #k, 0)
  R(k_xor); R(k_and); R(k_add); R(k_sub); R(k_mov); R(k_lshl); R(k_lshr); R(k_lshl_or); R(k_and_or); R(k_or3); R(k_add3);
  R(k_lshl_add); R(k_xad); R(k_bfe); R(k_bfi); R(k_alignbit); R(k_alignbyte); R(k_perm); R(k_bitop3); R(k_cndmask); R(k_cmp);
  R(k_mul_lo); R(k_mul_hi); R(k_mul_u24); R(k_mad_u24); R(k_bcnt); R(k_mad64); R(k_lshl64); R(k_lshr64); R(k_lshl_add64); R(k_cmp64); R(k_mov64);
  return 0;
}

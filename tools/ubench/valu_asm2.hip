// Second table of per-instruction issue costs on gfx950 (cycles per wave64 instruction per SIMD): the forms
// valu_asm.hip does not cover -- carry ops, v_cndmask with a live VCC, SDWA / DPP forms, dot4, packed 16-bit ops,
// min/max, shifts by register.  Same harness: 8 independent chains x 8 waves per SIMD, inline asm.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITER 16384
#define HIPCHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e)); return; } } while (0)

// PRE runs once per outer iteration (sets VCC where the instruction reads it)
#define KERNEL32P(NAME, PRE, ASM, CLOB)                                                       \
  __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t seed) {                 \
    uint32_t a[8];                                                                            \
    for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 17 + i * 0x01010101u;             \
    for (int it = 0; it < ITER; it++) {                                                       \
      asm volatile(PRE ::: "vcc");                                                            \
      _Pragma("unroll") for (int i = 0; i < 8; i++) {                                         \
        asm volatile(ASM : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]) : CLOB);     \
      }                                                                                       \
    }                                                                                         \
    uint32_t r = 0; for (int i = 0; i < 8; i++) r ^= a[i];                                    \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                           \
  }
#define KERNEL32(NAME, ASM) KERNEL32P(NAME, "", ASM, "memory")

KERNEL32(k_xor, "v_xor_b32 %0, %0, %1")
KERNEL32(k_or, "v_or_b32 %0, %0, %1")
KERNEL32(k_not, "v_not_b32 %0, %1")
KERNEL32(k_xnor, "v_xnor_b32 %0, %0, %1")
KERNEL32(k_lshr, "v_lshrrev_b32 %0, 3, %1")
KERNEL32(k_lshr_v, "v_lshrrev_b32 %0, %1, %0")
KERNEL32(k_ashr, "v_ashrrev_i32 %0, 3, %1")
KERNEL32(k_lshl1, "v_lshlrev_b32 %0, 1, %1")
KERNEL32(k_lshl_v, "v_lshlrev_b32 %0, %1, %0")
KERNEL32(k_min, "v_min_u32 %0, %0, %1")
KERNEL32(k_max, "v_max_u32 %0, %0, %1")
KERNEL32(k_subrev, "v_subrev_u32 %0, %0, %1")
KERNEL32(k_add_lit, "v_add_u32 %0, 0x12345678, %1")
KERNEL32(k_and_lit, "v_and_b32 %0, 0x03030303, %1")
KERNEL32(k_xor_s, "v_xor_b32 %0, s0, %1")
KERNEL32P(k_cndmask, "s_mov_b32 vcc_lo, 0x55555555\n\ts_mov_b32 vcc_hi, 0x55555555", "v_cndmask_b32 %0, %0, %1, vcc", "memory")
KERNEL32P(k_cndmask_e64, "s_mov_b32 vcc_lo, 0x55555555\n\ts_mov_b32 vcc_hi, 0x55555555", "v_cndmask_b32_e64 %0, %0, %1, vcc", "memory")
KERNEL32P(k_addc, "s_mov_b32 vcc_lo, 0x55555555\n\ts_mov_b32 vcc_hi, 0x55555555", "v_addc_co_u32 %0, vcc, %0, %1, vcc", "vcc")
KERNEL32P(k_add_co, "", "v_add_co_u32 %0, vcc, %0, %1", "vcc")
KERNEL32P(k_add_co_addc, "", "v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %0, vcc, %0, %2, vcc", "vcc")
KERNEL32P(k_sub_co, "", "v_sub_co_u32 %0, vcc, %0, %1", "vcc")
KERNEL32P(k_cmp_e64, "", "v_cmp_lt_u32_e64 s[4:5], %0, %1", "s4")
KERNEL32P(k_cmp_s, "", "v_cmp_lt_u32 vcc, s2, %1", "vcc")
KERNEL32(k_mov_sdwa, "v_mov_b32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2")
KERNEL32(k_mov_sdwa_d1, "v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PAD src0_sel:BYTE_2")
KERNEL32(k_and_sdwa, "v_and_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1")
KERNEL32(k_add_sdwa, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1")
KERNEL32(k_lshl_sdwa, "v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1")
KERNEL32(k_lshr_sdwa, "v_lshrrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:WORD_1")
KERNEL32(k_xor_sdwa, "v_xor_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1")
KERNEL32(k_mov_dpp, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL32(k_xor_dpp, "v_xor_b32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL32(k_add_dpp, "v_add_u32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KERNEL32(k_dot4, "v_dot4_u32_u8 %0, %1, %2, %0")
KERNEL32(k_dot2_u16, "v_dot2_u32_u16 %0, %1, %2, %0")
KERNEL32(k_sad, "v_sad_u32 %0, %0, %1, %2")
KERNEL32(k_sad_u8, "v_sad_u8 %0, %0, %1, %2")
KERNEL32(k_msad_u8, "v_msad_u8 %0, %0, %1, %2")
KERNEL32(k_lerp, "v_lerp_u8 %0, %0, %1, %2")
KERNEL32(k_mad_i24, "v_mad_i32_i24 %0, %0, %1, %2")
KERNEL32(k_mad_u16, "v_mad_u32_u16 %0, %0, %1, %2")
KERNEL32(k_mul_u24_lit, "v_mul_u32_u24 %0, 0x123456, %1")
KERNEL32(k_mul_lo_s, "v_mul_lo_u32 %0, %1, s3")
KERNEL32(k_pk_add_u16, "v_pk_add_u16 %0, %0, %1")
KERNEL32(k_pk_mul_lo_u16, "v_pk_mul_lo_u16 %0, %0, %1")
KERNEL32(k_pk_mad_u16, "v_pk_mad_u16 %0, %0, %1, %2")
KERNEL32(k_pk_lshl_b16, "v_pk_lshlrev_b16 %0, 3, %1 op_sel_hi:[0,1]")
KERNEL32(k_pk_lshr_b16, "v_pk_lshrrev_b16 %0, 3, %1 op_sel_hi:[0,1]")
KERNEL32(k_add_lshl, "v_add_lshl_u32 %0, %0, %1, 4")
KERNEL32(k_bfe_v, "v_bfe_u32 %0, %1, %2, 8")
KERNEL32(k_bfm, "v_bfm_b32 %0, %1, %2")
KERNEL32(k_ffbh, "v_ffbh_u32 %0, %1")
KERNEL32(k_bfrev, "v_bfrev_b32 %0, %1")
KERNEL32(k_mbcnt, "v_mbcnt_lo_u32_b32 %0, %1, %0")
KERNEL32(k_cvt_pk_u8, "v_cvt_pk_u8_f32 %0, %0, %1, %2")
KERNEL32(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
KERNEL32(k_mul_f32, "v_mul_f32 %0, %0, %1")
KERNEL32(k_add_f32, "v_add_f32 %0, %0, %1")
KERNEL32(k_readlane_free, "v_mov_b32 %0, %1\n\tv_xor_b32 %0, %0, %2")

// 64-bit forms
#define KERNEL64(NAME, ASM)                                                                   \
  __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t seed) {                 \
    uint64_t a[8];                                                                            \
    for (int i = 0; i < 8; i++) a[i] = ((uint64_t)(seed + threadIdx.x * 17) << 32) | (i * 0x01010101u + seed); \
    for (int it = 0; it < ITER; it++) {                                                       \
      _Pragma("unroll") for (int i = 0; i < 8; i++) {                                         \
        asm volatile(ASM : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]) : "vcc");    \
      }                                                                                       \
    }                                                                                         \
    uint32_t r = 0; for (int i = 0; i < 8; i++) r ^= (uint32_t)a[i] ^ (uint32_t)(a[i] >> 32); \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                           \
  }
KERNEL64(k_lshl_add64_0, "v_lshl_add_u64 %0, %1, 0, %2")
KERNEL64(k_lshl_add64_s, "v_lshl_add_u64 %0, %1, 0, s[2:3]")
KERNEL64(k_pk_mov, "v_pk_mov_b32 %0, %1, %2 op_sel:[0,1]")
KERNEL64(k_pk_add_f32, "v_pk_add_f32 %0, %1, %2")
KERNEL64(k_add_f64, "v_add_f64 %0, %1, %2")
KERNEL64(k_fma_f64, "v_fma_f64 %0, %1, %2, %0")
KERNEL64(k_mul_f64, "v_mul_f64 %0, %1, %2")
KERNEL64(k_ashr64, "v_ashrrev_i64 %0, 3, %1")

// mad_u64_u32 with a zero addend, and the "32-bit mad" use (only the low word consumed)
__global__ __launch_bounds__(256) void k_mad64_zero(uint32_t* out, uint32_t seed) {
  uint64_t a[8]; uint32_t b[8];
  for (int i = 0; i < 8; i++) { a[i] = seed + i; b[i] = seed * 31 + i + threadIdx.x; }
  for (int it = 0; it < ITER; it++) {
    _Pragma("unroll") for (int i = 0; i < 8; i++) {
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(a[i]) : "v"(b[(i + 1) & 7]), "v"(b[(i + 2) & 7]) : "vcc");
      b[i] ^= (uint32_t)a[i];
    }
  }
  uint32_t r = 0; for (int i = 0; i < 8; i++) r ^= (uint32_t)a[i] ^ (uint32_t)(a[i] >> 32) ^ b[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ __launch_bounds__(256) void k_mad64_sconst(uint32_t* out, uint32_t seed) {
  uint64_t a[8]; uint32_t b[8];
  for (int i = 0; i < 8; i++) { a[i] = ((uint64_t)(seed + threadIdx.x * 17) << 32) | (i * 0x01010101u + seed); b[i] = seed * 31 + i + threadIdx.x; }
  for (int it = 0; it < ITER; it++) {
    _Pragma("unroll") for (int i = 0; i < 8; i++) {
      asm volatile("v_mad_u64_u32 %0, vcc, %1, s3, %0" : "+v"(a[i]) : "v"(b[(i + 1) & 7]) : "vcc");
    }
  }
  uint32_t r = 0; for (int i = 0; i < 8; i++) r ^= (uint32_t)a[i] ^ (uint32_t)(a[i] >> 32);
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
// mixes: does a fast op pair up with a slow one?  (alternating xor / mul_lo: sum of the two or less?)
KERNEL32(k_mix_xor_mul, "v_xor_b32 %0, %0, %1\n\tv_mul_lo_u32 %0, %0, %2")
KERNEL32(k_mix_xor_xor, "v_xor_b32 %0, %0, %1\n\tv_xor_b32 %0, %0, %2")
KERNEL32(k_mix_mul_mul, "v_mul_lo_u32 %0, %0, %1\n\tv_mul_lo_u32 %0, %0, %2")
KERNEL32(k_mix_lshr_xor, "v_lshrrev_b32 %0, 1, %0\n\tv_xor_b32 %0, %0, %1")

typedef void (*kern_t)(uint32_t*, uint32_t);
static void run(kern_t fn, const char* name, double per_iter) {
  uint32_t* d; HIPCHK(hipMalloc(&d, 256 * 8 * 256 * 4));
  hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  dim3 grid(256 * 8), block(256);
  hipLaunchKernelGGL(fn, grid, block, 0, 0, d, 1u); HIPCHK(hipDeviceSynchronize());
  float best = 1e9f;
  for (int rep = 0; rep < 3; rep++) {
    HIPCHK(hipEventRecord(e0)); hipLaunchKernelGGL(fn, grid, block, 0, 0, d, 2u); HIPCHK(hipEventRecord(e1)); HIPCHK(hipEventSynchronize(e1));
    float ms; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  double ops = 8.0 * ITER * 8;
  printf("%-18s %8.3f ms  %5.2f cycles per asm block per SIMD @2.4GHz nominal (%g instr per block)\n", name, best, best * 1e-3 * 2.4e9 / ops, per_iter);
  HIPCHK(hipFree(d));
}
int main() {
  setvbuf(stdout, NULL, _IONBF, 0);
#define R(k) run(k, #k, 1)
#define R2(k) run(k, #k, 2)
  R(k_xor); R(k_or); R(k_not); R(k_xnor); R(k_lshr); R(k_lshr_v); R(k_ashr); R(k_lshl1); R(k_lshl_v); R(k_min); R(k_max); R(k_subrev);
  R(k_add_lit); R(k_and_lit); R(k_xor_s); R(k_cndmask); R(k_cndmask_e64); R(k_addc); R(k_add_co); R2(k_add_co_addc); R(k_sub_co); R(k_cmp_e64); R(k_cmp_s);
  R(k_mov_sdwa); R(k_mov_sdwa_d1); R(k_and_sdwa); R(k_add_sdwa); R(k_lshl_sdwa); R(k_lshr_sdwa); R(k_xor_sdwa); R(k_mov_dpp); R(k_xor_dpp); R(k_add_dpp);
  R(k_dot4); R(k_dot2_u16); R(k_sad); R(k_sad_u8); R(k_msad_u8); R(k_lerp); R(k_mad_i24); R(k_mad_u16); R(k_mul_u24_lit); R(k_mul_lo_s);
  R(k_pk_add_u16); R(k_pk_mul_lo_u16); R(k_pk_mad_u16); R(k_pk_lshl_b16); R(k_pk_lshr_b16); R(k_add_lshl); R(k_bfe_v); R(k_bfm); R(k_ffbh); R(k_bfrev); R(k_mbcnt);
  R(k_cvt_pk_u8); R(k_fma_f32); R(k_mul_f32); R(k_add_f32); R2(k_readlane_free);
  R(k_lshl_add64_0); R(k_lshl_add64_s); R(k_pk_mov); R(k_pk_add_f32); R(k_add_f64); R(k_fma_f64); R(k_mul_f64); R(k_ashr64);
  R(k_mad64_zero); R(k_mad64_sconst);
  R2(k_mix_xor_mul); R2(k_mix_xor_xor); R2(k_mix_mul_mul); R2(k_mix_lshr_xor);
  return 0;
}

// Does the order of "fast" (xor / add / lshr ...) and "slow" (mul / alignbit ...) VALU instructions matter on gfx950?
// Each pattern is ONE asm block of independent instructions over eight registers; 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITER 16384
#define HIPCHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e)); return; } } while (0)
#define F(i) "v_xor_b32 %" #i ", %" #i ", %8\n\t"
#define A(i) "v_add_u32 %" #i ", %" #i ", %8\n\t"
#define H(i) "v_lshrrev_b32 %" #i ", 1, %" #i "\n\t"
#define S(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n\t"
#define B(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 5\n\t"
#define L "ds_read_b32 %9, %10\n\t"
#define W "s_waitcnt lgkmcnt(0)\n\t"
#define N "s_nop 0\n\t"
#define X "s_add_u32 s4, s4, 1\n\t"
#define PAT(NAME, ASM)                                                                        \
  __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t seed) {                 \
    __shared__ uint32_t sm[256];                                                              \
    sm[threadIdx.x] = seed;                                                                   \
    __syncthreads();                                                                          \
    uint32_t a[8], c = seed * 7 + threadIdx.x, ld = 0, ad = (threadIdx.x & 63) * 4;           \
    for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 17 + i * 0x01010101u;             \
    for (int it = 0; it < ITER; it++) {                                                       \
      asm volatile(ASM : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) \
                   : "v"(c), "v"(ld), "v"(ad) : "s4", "scc", "memory");                              \
    }                                                                                         \
    uint32_t r = ld; for (int i = 0; i < 8; i++) r ^= a[i];                                   \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                           \
  }
PAT(p_FFFFFFFF, F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7))
PAT(p_SSSSSSSS, S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7))
PAT(p_FSFSFSFS, F(0) S(1) F(2) S(3) F(4) S(5) F(6) S(7))
PAT(p_FFSSFFSS, F(0) F(1) S(2) S(3) F(4) F(5) S(6) S(7))
PAT(p_FFFFSSSS, F(0) F(1) F(2) F(3) S(4) S(5) S(6) S(7))
PAT(p_FFFSFSSS, F(0) F(1) F(2) S(3) F(4) S(5) S(6) S(7))
PAT(p_FSSSFSSS, F(0) S(1) S(2) S(3) F(4) S(5) S(6) S(7))
PAT(p_FFSSSSSS, F(0) F(1) S(2) S(3) S(4) S(5) S(6) S(7))
PAT(p_AHBSAHBS, A(0) H(1) B(2) S(3) A(4) H(5) B(6) S(7))
PAT(p_ABHSABHS, A(0) B(1) H(2) S(3) A(4) B(5) H(6) S(7))
PAT(p_FdepF, "v_xor_b32 %0, %0, %8\n\tv_xor_b32 %1, %0, %8\n\t" S(2) S(3) "v_xor_b32 %4, %4, %8\n\tv_xor_b32 %5, %4, %8\n\t" S(6) S(7))
PAT(p_FNFSSFNFSS, F(0) N F(1) S(2) S(3) F(4) N F(5) S(6) S(7))
PAT(p_FXFSSFXFSS, F(0) X F(1) S(2) S(3) F(4) X F(5) S(6) S(7))
PAT(p_FLFSSFFSS, F(0) L F(1) S(2) S(3) F(4) F(5) S(6) S(7) W)
PAT(p_FFLSSFFSS, F(0) F(1) L S(2) S(3) F(4) F(5) S(6) S(7) W)
PAT(p_FSLFSFSFS, F(0) S(1) L F(2) S(3) F(4) S(5) F(6) S(7) W)
PAT(p_F6S2, F(0) F(1) F(2) F(3) F(4) F(5) S(6) S(7))
PAT(p_F5S3, F(0) F(1) F(2) F(3) F(4) S(5) S(6) S(7))
PAT(p_FFSFFSFS, F(0) F(1) S(2) F(3) F(4) S(5) F(6) S(7))

typedef void (*kern_t)(uint32_t*, uint32_t);
static void run(kern_t fn, const char* name) {
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
  double blocks = 8.0 * ITER;  // asm blocks per SIMD
  printf("%-14s %8.3f ms  %6.2f cycles per block of 8 VALU per SIMD @2.4GHz nominal\n", name, best, best * 1e-3 * 2.4e9 / blocks);
  HIPCHK(hipFree(d));
}
int main() {
  setvbuf(stdout, NULL, _IONBF, 0);
#define R(k) run(k, #k)
  R(p_FFFFFFFF); R(p_SSSSSSSS); R(p_FSFSFSFS); R(p_FFSSFFSS); R(p_FFFFSSSS); R(p_FFFSFSSS); R(p_FSSSFSSS); R(p_FFSSSSSS);
  R(p_AHBSAHBS); R(p_ABHSABHS); R(p_FdepF); R(p_FNFSSFNFSS); R(p_FXFSSFXFSS); R(p_FLFSSFFSS); R(p_FFLSSFFSS); R(p_FSLFSFSFS);
  R(p_F6S2); R(p_F5S3); R(p_FFSFFSFS);
  return 0;
}

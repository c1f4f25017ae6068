// sort_rates.hip -- measured rates of the rocPRIM primitives the inverted-join pair path is built from
// (radix_sort_pairs u64/u32 keys + u32 values, radix_sort_keys on a narrow bit range, run_length_encode).
// Build: hipcc -O3 --offload-arch=gfx950 -o sort_rates sort_rates.hip ; run: ./sort_rates
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void fill64(uint64_t* k, uint32_t* v, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t x = i * 0x9E3779B97F4A7C15ull + 12345;
  x ^= x >> 31; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 29;
  k[i] = x; v[i] = (uint32_t)(i / 1000);
}
__global__ void fill32(uint32_t* k, uint32_t* v, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t x = i * 0x9E3779B97F4A7C15ull + 12345;
  x ^= x >> 31; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 29;
  k[i] = (uint32_t)(x >> 20); v[i] = (uint32_t)(i / 490);
}
__global__ void fillpairs(uint64_t* k, size_t n, uint32_t ng) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t x = i * 0x9E3779B97F4A7C15ull + 777;
  x ^= x >> 31; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 29;
  uint64_t r = x % ng, c = (x >> 32) % ng;
  k[i] = ((r * ng + c) / 5) * 5;  // some duplicates
}

template <typename K>
int bench_pairs(size_t n, const char* name) {
  K *k0, *k1; uint32_t *v0, *v1;
  CK(hipMalloc(&k0, n * sizeof(K))); CK(hipMalloc(&k1, n * sizeof(K)));
  CK(hipMalloc(&v0, n * 4)); CK(hipMalloc(&v1, n * 4));
  if constexpr (sizeof(K) == 8) fill64<<<(n + 255) / 256, 256>>>((uint64_t*)k0, v0, n);
  else fill32<<<(n + 255) / 256, 256>>>((uint32_t*)k0, v0, n);
  size_t tb = 0;
  CK(rocprim::radix_sort_pairs(nullptr, tb, k0, k1, v0, v1, n, 0, sizeof(K) * 8, 0));
  void* tmp; CK(hipMalloc(&tmp, tb));
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int it = 0; it < 4; it++) {
    hipEventRecord(a, 0);
    CK(rocprim::radix_sort_pairs(tmp, tb, k0, k1, v0, v1, n, 0, sizeof(K) * 8, 0));
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%s radix_sort_pairs n=%zu: %.3f ms (%.2f Gkeys/s), temp %zu MB\n", name, n, ms, n / ms / 1e6, tb >> 20);
  }
  hipFree(k0); hipFree(k1); hipFree(v0); hipFree(v1); hipFree(tmp);
  return 0;
}

int bench_keys(size_t n, uint32_t ng, int bits) {
  uint64_t *k0, *k1;
  CK(hipMalloc(&k0, n * 8)); CK(hipMalloc(&k1, n * 8));
  fillpairs<<<(n + 255) / 256, 256>>>(k0, n, ng);
  size_t tb = 0;
  CK(rocprim::radix_sort_keys(nullptr, tb, k0, k1, n, 0, bits, 0));
  void* tmp; CK(hipMalloc(&tmp, tb));
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int it = 0; it < 3; it++) {
    hipEventRecord(a, 0);
    CK(rocprim::radix_sort_keys(tmp, tb, k0, k1, n, 0, bits, 0));
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("radix_sort_keys u64 bits=%d n=%zu: %.3f ms (%.2f Gkeys/s)\n", bits, n, ms, n / ms / 1e6);
  }
  // run-length encode of the sorted keys
  uint64_t* uq; uint32_t* cnt; size_t* nr;
  CK(hipMalloc(&uq, n * 8)); CK(hipMalloc(&cnt, n * 4)); CK(hipMalloc(&nr, 8));
  size_t tb2 = 0;
  CK(rocprim::run_length_encode(nullptr, tb2, k1, n, uq, cnt, nr, 0));
  void* tmp2; CK(hipMalloc(&tmp2, tb2));
  for (int it = 0; it < 3; it++) {
    hipEventRecord(a, 0);
    CK(rocprim::run_length_encode(tmp2, tb2, k1, n, uq, cnt, nr, 0));
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    size_t h = 0; hipMemcpy(&h, nr, 8, hipMemcpyDeviceToHost);
    printf("run_length_encode n=%zu -> %zu runs: %.3f ms\n", n, h, ms);
  }
  hipFree(k0); hipFree(k1); hipFree(tmp); hipFree(uq); hipFree(cnt); hipFree(nr); hipFree(tmp2);
  return 0;
}

int main() {
  if (bench_pairs<uint64_t>(10'000'000, "u64+u32")) return 1;
  if (bench_pairs<uint64_t>(100'000'000, "u64+u32")) return 1;
  if (bench_pairs<uint32_t>(12'250'000, "u32+u32")) return 1;
  if (bench_pairs<uint32_t>(98'000'000, "u32+u32")) return 1;
  if (bench_keys(13'500'000, 10000, 27)) return 1;
  if (bench_keys(30'000'000, 25000, 30)) return 1;
  if (bench_keys(100'000'000, 100000, 34)) return 1;
  return 0;
}

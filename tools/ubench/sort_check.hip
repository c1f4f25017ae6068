// sort_check.hip -- does rocprim::radix_sort_pairs on a partial bit range sort by those bits, stably?
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>
int check(size_t n, unsigned b0, unsigned b1, uint64_t pool) {
  std::mt19937_64 rng(n + b0);
  std::vector<uint64_t> k(n); std::vector<uint32_t> v(n);
  for (size_t i = 0; i < n; i++) { k[i] = (rng() % pool) * 0x9E3779B97F4A7C15ull; v[i] = (uint32_t)i; }
  uint64_t *k0, *k1; uint32_t *v0, *v1;
  hipMalloc(&k0, n * 8); hipMalloc(&k1, n * 8); hipMalloc(&v0, n * 4); hipMalloc(&v1, n * 4);
  hipMemcpy(k0, k.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(v0, v.data(), n * 4, hipMemcpyHostToDevice);
  size_t tb = 0;
  rocprim::radix_sort_pairs(nullptr, tb, (const uint64_t*)k0, k1, (const uint32_t*)v0, v1, n, 0u, 64u, 0);
  void* tmp; hipMalloc(&tmp, tb);
  hipError_t e = rocprim::radix_sort_pairs(tmp, tb, (const uint64_t*)k0, k1, (const uint32_t*)v0, v1, n, b0, b1, 0);
  hipDeviceSynchronize();
  std::vector<uint64_t> ks(n); std::vector<uint32_t> vs(n);
  hipMemcpy(ks.data(), k1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(vs.data(), v1, n * 4, hipMemcpyDeviceToHost);
  const uint64_t mask = (b1 - b0 == 64) ? ~0ull : (((1ull << (b1 - b0)) - 1) << b0);
  size_t bad_order = 0, bad_stable = 0, bad_pair = 0;
  for (size_t i = 0; i + 1 < n; i++) {
    if ((ks[i] & mask) > (ks[i + 1] & mask)) bad_order++;
    if ((ks[i] & mask) == (ks[i + 1] & mask) && vs[i] > vs[i + 1]) bad_stable++;
  }
  for (size_t i = 0; i < n; i++) if (k[vs[i]] != ks[i]) bad_pair++;
  printf("n=%zu bits [%u,%u) pool %llu: err=%d unordered %zu unstable %zu mispaired %zu\n", n, b0, b1, (unsigned long long)pool, (int)e, bad_order, bad_stable, bad_pair);
  hipFree(k0); hipFree(k1); hipFree(v0); hipFree(v1); hipFree(tmp);
  return 0;
}
int main() {
  for (size_t n : {5000ul, 19225ul, 200000ul, 1000000ul, 2000000ul, 10000000ul})
    for (uint64_t pool : {1ull << 40}) {
      check(n, 40, 64, pool); check(n, 33, 64, pool); check(n, 31, 63, pool); check(n, 12, 44, pool); check(n, 0, 52, pool); check(n, 0, 32, pool); check(n, 20, 63, pool);
      check(n, 19, 51, pool);
    }
  return 0;
}

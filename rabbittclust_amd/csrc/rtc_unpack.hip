// rtc_unpack.hip -- the command lines' staging format back to the byte stream the sketch kernels read.
//
// The reference hands every record to the sketcher as ASCII (kseq buffers, src/SketchInfo.cpp:928-948).  Over PCIe
// that is 1 byte per base and the command line is bound there (19 ms per GB against 5 ms of sketching); the host
// parser therefore packs the bases to 2 bits (rtc_host.cpp: PackedSink) and lists what is not ACGT as runs.  Here
// the packed batch is expanded again in HBM -- 0.25 B read + 1 B written per base at memory speed -- and the runs
// are overwritten with 'N' (any character outside ACGT ends a k-mer the same way, and the gaps between genomes are
// runs too), so the sketch kernels see the stream they would have seen from the ASCII staging buffer.
#include <algorithm>

#include "rtc_internal.h"

namespace {

// one lane: 16 packed bytes = 64 bases = four 16-byte stores; base i at bits 2 (i & 3) of byte i >> 2
__global__ __launch_bounds__(256) void unpack_bases_kernel(const uint4* __restrict__ packed, uint64_t n16, uint4* __restrict__ out) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint4 p = packed[i];
    const uint32_t w[4] = {p.x, p.y, p.z, p.w};
#pragma unroll
    for (int d = 0; d < 4; d++) {
      uint32_t o[4];
#pragma unroll
      for (int b = 0; b < 4; b++) {
        const uint32_t by = (w[d] >> (8 * b)) & 0xffu;                       // four bases
        const uint32_t codes = (by & 3u) | ((by << 6) & 0x300u) | ((by << 12) & 0x30000u) | ((by << 18) & 0x3000000u);
        o[b] = __builtin_amdgcn_perm(0u, 0x54474341u, codes);                // "ACGT"[code] per byte
      }
      out[4 * i + d] = make_uint4(o[0], o[1], o[2], o[3]);
    }
  }
}

// one workgroup per run: 'N' over [start, start + len)
__global__ __launch_bounds__(256) void patch_runs_kernel(const uint64_t* __restrict__ runs, uint8_t* __restrict__ out, uint64_t limit) {
  const uint64_t st = runs[2 * (uint64_t)blockIdx.x], ln = runs[2 * (uint64_t)blockIdx.x + 1];
  const uint64_t en = st + ln < limit ? st + ln : limit;
  for (uint64_t p = st + threadIdx.x; p < en; p += blockDim.x) out[p] = (uint8_t)'N';
}

}  // namespace

extern "C" int rtc_unpack_bases_dev(rtc_ctx* ctx, const uint8_t* d_packed, uint64_t n_bases, const uint64_t* d_runs,
                                    uint64_t n_runs, uint8_t* d_seq) {
  if (!ctx || (n_bases && (!d_packed || !d_seq)) || (n_runs && !d_runs)) return RTC_ERR_ARG;
  if (((uintptr_t)d_packed & 15) || ((uintptr_t)d_seq & 15)) return rtc_fail(ctx, RTC_ERR_ARG, "buffers must be 16-byte aligned");
  if (n_bases & 63) return rtc_fail(ctx, RTC_ERR_ARG, "n_bases must be a multiple of 64 (pad the batch)");
  if (n_runs > 0x7fffffffull) return rtc_fail(ctx, RTC_ERR_ARG, "too many runs");
  if (!n_bases) return RTC_OK;
  RTC_HIP(ctx, hipSetDevice(ctx->device));
  const uint64_t n16 = n_bases / 64;
  const uint32_t grid = (uint32_t)std::min<uint64_t>((n16 + 255) / 256, (uint64_t)ctx->num_cu * 16);
  hipLaunchKernelGGL(unpack_bases_kernel, dim3(grid), dim3(256), 0, ctx->stream, (const uint4*)d_packed, n16, (uint4*)d_seq);
  RTC_CHECK_LAUNCH(ctx);
  if (n_runs) {
    hipLaunchKernelGGL(patch_runs_kernel, dim3((uint32_t)n_runs), dim3(256), 0, ctx->stream, d_runs, d_seq, n_bases);
    RTC_CHECK_LAUNCH(ctx);
  }
  return RTC_OK;
}

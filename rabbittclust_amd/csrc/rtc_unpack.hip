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

// 'N' over every run [start, start + len).  Runs come in two kinds: millions of short ones (every record separator of
// a contig-rich assembly or a FASTQ file is a run of one byte) and a few very long ones (a dropped genome's whole slot,
// the gap behind a genome whose slot was sized from a wrong gzip ISIZE).  One LANE per run writes the short ones and
// lists the long ones; the workgroups of a second launch then share the listed runs in 64 KiB pieces, 16-byte stores
// for the aligned interior.  (One workgroup per run, whatever its length, used to serialise both kinds.)
constexpr uint64_t SHORT_RUN = 64;
constexpr uint32_t LONG_CAP = 65536;  // listed long runs; beyond that a lane writes its run itself

__global__ __launch_bounds__(256) void patch_short_runs_kernel(const uint64_t* __restrict__ runs, uint64_t n_runs, uint8_t* __restrict__ out,
                                                               uint64_t limit, uint32_t* __restrict__ n_long, uint64_t* __restrict__ long_runs) {
  for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_runs; r += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t st = runs[2 * r], ln = runs[2 * r + 1];
    const uint64_t en = st + ln < limit ? st + ln : limit;
    if (ln > SHORT_RUN) {
      const uint32_t at = atomicAdd(n_long, 1u);
      if (at < LONG_CAP) { long_runs[2 * (uint64_t)at] = st; long_runs[2 * (uint64_t)at + 1] = en; continue; }
    }
    for (uint64_t p = st; p < en; p++) out[p] = (uint8_t)'N';
  }
}

__global__ __launch_bounds__(256) void patch_long_runs_kernel(const uint32_t* __restrict__ n_long, const uint64_t* __restrict__ long_runs,
                                                              uint8_t* __restrict__ out) {
  const uint4 n16 = make_uint4(0x4e4e4e4eu, 0x4e4e4e4eu, 0x4e4e4e4eu, 0x4e4e4e4eu);
  const uint32_t nl = min(*n_long, LONG_CAP);
  uint64_t piece = 0;  // every workgroup walks the (short) list and takes its 64 KiB pieces in turn
  for (uint32_t r = 0; r < nl; r++) {
    const uint64_t st = long_runs[2 * (uint64_t)r], en = long_runs[2 * (uint64_t)r + 1];
    for (uint64_t a = st; a < en; a += 65536, piece++) {
      if (piece % gridDim.x != blockIdx.x) continue;
      const uint64_t b = a + 65536 < en ? a + 65536 : en;
      const uint64_t a16 = (a + 15) & ~(uint64_t)15, b16 = b & ~(uint64_t)15;
      if (a16 >= b16) { for (uint64_t p = a + threadIdx.x; p < b; p += blockDim.x) out[p] = (uint8_t)'N'; continue; }
      if (a + threadIdx.x < a16) out[a + threadIdx.x] = (uint8_t)'N';
      for (uint64_t p = a16 + 16 * (uint64_t)threadIdx.x; p < b16; p += 16 * (uint64_t)blockDim.x) *reinterpret_cast<uint4*>(out + p) = n16;
      if (b16 + threadIdx.x < b) out[b16 + threadIdx.x] = (uint8_t)'N';
    }
  }
}

// the run list's contract (ascending by start, disjoint, inside the batch): one pass, a flag in host memory
__global__ __launch_bounds__(256) void check_runs_kernel(const uint64_t* __restrict__ runs, uint64_t n_runs, uint64_t n_bases, uint32_t* __restrict__ bad) {
  for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_runs; r += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t st = runs[2 * r], ln = runs[2 * r + 1], en = st + ln;
    bool ok = en >= st && en <= n_bases;
    if (r + 1 < n_runs) ok = ok && en <= runs[2 * (r + 1)];
    if (!ok) *bad = 1u;
  }
}

}  // namespace

int rtc_check_runs_async(rtc_ctx* ctx, const uint64_t* d_runs, uint64_t n_runs, uint64_t n_bases) {
  if (!n_runs) return RTC_OK;
  if (!ctx->sticky) {
    RTC_HIP(ctx, hipHostMalloc((void**)&ctx->sticky, 64, hipHostMallocMapped));
    *ctx->sticky = 0;
  }
  const uint32_t grid = (uint32_t)std::min<uint64_t>((n_runs + 255) / 256, (uint64_t)ctx->num_cu * 4);
  hipLaunchKernelGGL(check_runs_kernel, dim3(grid), dim3(256), 0, ctx->stream, d_runs, n_runs, n_bases, ctx->sticky);
  RTC_CHECK_LAUNCH(ctx);
  return RTC_OK;
}

int rtc_sticky_error(rtc_ctx* ctx) {
  if (!ctx->sticky || !*(volatile uint32_t*)ctx->sticky) return RTC_OK;
  *ctx->sticky = 0;
  return rtc_fail(ctx, RTC_ERR_ARG, "the run list of an earlier packed batch was not ascending, disjoint and inside the batch: the sketches of that "
                                    "batch are not valid (rtc_sketch_minhash_packed_dev / rtc_sketch_kssd_packed_dev / rtc_unpack_bases_dev)");
}

extern "C" int rtc_unpack_bases_dev(rtc_ctx* ctx, const uint8_t* d_packed, uint64_t n_bases, const uint64_t* d_runs,
                                    uint64_t n_runs, uint8_t* d_seq) {
  if (!ctx || (n_bases && (!d_packed || !d_seq)) || (n_runs && !d_runs)) return RTC_ERR_ARG;
  if (((uintptr_t)d_packed & 15) || ((uintptr_t)d_seq & 15)) return rtc_fail(ctx, RTC_ERR_ARG, "buffers must be 16-byte aligned");
  if (n_bases & 63) return rtc_fail(ctx, RTC_ERR_ARG, "n_bases must be a multiple of 64 (pad the batch)");
  if (!n_bases) return RTC_OK;
  RTC_HIP(ctx, hipSetDevice(ctx->device));
  const uint64_t n16 = n_bases / 64;
  const uint32_t grid = (uint32_t)std::min<uint64_t>((n16 + 255) / 256, (uint64_t)ctx->num_cu * 16);
  hipLaunchKernelGGL(unpack_bases_kernel, dim3(grid), dim3(256), 0, ctx->stream, (const uint4*)d_packed, n16, (uint4*)d_seq);
  RTC_CHECK_LAUNCH(ctx);
  if (n_runs) {
    RTC_TRY(rtc_check_runs_async(ctx, d_runs, n_runs, n_bases));
    void* ws = nullptr;
    RTC_TRY(rtc_ws(ctx, 3, 64 + (size_t)LONG_CAP * 16, &ws));
    uint32_t* d_nlong = (uint32_t*)ws;
    uint64_t* d_long = (uint64_t*)((char*)ws + 64);
    RTC_HIP(ctx, hipMemsetAsync(d_nlong, 0, 4, ctx->stream));
    const uint32_t gs = (uint32_t)std::min<uint64_t>((n_runs + 255) / 256, (uint64_t)ctx->num_cu * 8);
    hipLaunchKernelGGL(patch_short_runs_kernel, dim3(gs), dim3(256), 0, ctx->stream, d_runs, n_runs, d_seq, n_bases, d_nlong, d_long);
    RTC_CHECK_LAUNCH(ctx);
    hipLaunchKernelGGL(patch_long_runs_kernel, dim3((uint32_t)ctx->num_cu), dim3(256), 0, ctx->stream, (const uint32_t*)d_nlong,
                       (const uint64_t*)d_long, d_seq);
    RTC_CHECK_LAUNCH(ctx);
  }
  return RTC_OK;
}

namespace { __global__ void touch_unit_kernel() {} }
int rtc_touch_unpack(rtc_ctx* ctx) {
  hipLaunchKernelGGL(touch_unit_kernel, dim3(1), dim3(64), 0, ctx->stream);
  RTC_CHECK_LAUNCH(ctx);
  return RTC_OK;
}

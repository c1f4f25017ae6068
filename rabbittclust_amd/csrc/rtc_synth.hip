// rtc_synth.hip -- deterministic synthetic genomes written straight into HBM.
// Counter-based: base(pos) depends only on (fam_seed, mut_seed, mut_thr, n_every, pos), so the
// CPU side can regenerate any slice bit-identically.  One lane produces 16 consecutive bases
// and stores them with a single 16-byte write (coalesced 1 KiB per wave-instruction).
#include "rtc_internal.h"

__device__ __forceinline__ uint64_t splitmix64_at(uint64_t seed, uint64_t idx) {
  uint64_t z = seed + (idx + 1) * 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

__device__ __forceinline__ uint8_t synth_base(const rtc_synth_desc& d, uint64_t pos, uint64_t famw,
                                              uint64_t mutw) {
  uint32_t b = (uint32_t)(famw >> (2 * (pos & 31))) & 3u;
  if (d.mut_thr) {
    uint32_t v = (uint32_t)(mutw >> (16 * (pos & 3))) & 0xFFFFu;
    if ((v >> 2) < d.mut_thr) b = (b + 1 + ((v & 3u) % 3u)) & 3u;
  }
  uint8_t c = (uint8_t)((0x54474341u >> (8 * b)) & 0xFFu);  // "ACGT"[b]
  if (d.n_every && pos >= d.n_every && (pos % d.n_every) < 8) c = (uint8_t)'N';
  return c;
}

// grid.y = genome; grid.x strides over 16-base groups of that genome.
__global__ __launch_bounds__(256) void synth_kernel(const rtc_synth_desc* __restrict__ desc,
                                                    const uint64_t* __restrict__ off,
                                                    uint8_t* __restrict__ seq) {
  const uint32_t g = blockIdx.y;
  const rtc_synth_desc d = desc[g];
  const uint64_t g0 = off[g], len = off[g + 1] - g0;
  // 16-byte aligned groups in *absolute* address space so the vector store is aligned
  const uint64_t a0 = g0 & ~15ULL;
  const uint64_t ngroups = (g0 + len - a0 + 15) >> 4;
  for (uint64_t grp = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; grp < ngroups;
       grp += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t abs0 = a0 + (grp << 4);
    uint32_t w[4];
    bool full = abs0 >= g0 && abs0 + 16 <= g0 + len;
    uint8_t bytes[16];
    uint64_t famw = 0, mutw = 0, famw_idx = ~0ULL, mutw_idx = ~0ULL;
#pragma unroll
    for (int t = 0; t < 16; t++) {
      uint64_t ap = abs0 + t;
      uint8_t c = 0;
      if (ap >= g0 && ap < g0 + len) {
        uint64_t pos = ap - g0;
        if ((pos >> 5) != famw_idx) { famw_idx = pos >> 5; famw = splitmix64_at(d.fam_seed, famw_idx); }
        if (d.mut_thr && (pos >> 2) != mutw_idx) { mutw_idx = pos >> 2; mutw = splitmix64_at(d.mut_seed, mutw_idx); }
        c = synth_base(d, pos, famw, mutw);
      }
      bytes[t] = c;
    }
    if (full) {
#pragma unroll
      for (int q = 0; q < 4; q++)
        w[q] = (uint32_t)bytes[4 * q] | ((uint32_t)bytes[4 * q + 1] << 8) |
               ((uint32_t)bytes[4 * q + 2] << 16) | ((uint32_t)bytes[4 * q + 3] << 24);
      *reinterpret_cast<uint4*>(seq + abs0) = make_uint4(w[0], w[1], w[2], w[3]);
    } else {
      for (int t = 0; t < 16; t++) {
        uint64_t ap = abs0 + t;
        if (ap >= g0 && ap < g0 + len) seq[ap] = bytes[t];
      }
    }
  }
}

extern "C" int rtc_synth_genomes_dev(rtc_ctx* ctx, const rtc_synth_desc* h_desc,
                                     const uint64_t* h_off, uint32_t n, uint8_t* d_seq) {
  if (!ctx || !h_desc || !h_off || (n && !d_seq)) return RTC_ERR_ARG;
  if (n == 0) return RTC_OK;
  if (((uintptr_t)d_seq & 15) != 0) return rtc_fail(ctx, RTC_ERR_ARG, "d_seq must be 16-byte aligned");
  RTC_HIP(ctx, hipSetDevice(ctx->device));
  void* ws = nullptr;
  size_t bdesc = (size_t)n * sizeof(rtc_synth_desc), boff = (size_t)(n + 1) * sizeof(uint64_t);
  RTC_TRY(rtc_ws(ctx, 0, bdesc + boff, &ws));
  rtc_synth_desc* d_desc = (rtc_synth_desc*)ws;
  uint64_t* d_off = (uint64_t*)((char*)ws + bdesc);
  RTC_HIP(ctx, hipMemcpyAsync(d_desc, h_desc, bdesc, hipMemcpyHostToDevice, ctx->stream));
  RTC_HIP(ctx, hipMemcpyAsync(d_off, h_off, boff, hipMemcpyHostToDevice, ctx->stream));
  uint64_t maxlen = 0;
  for (uint32_t g = 0; g < n; g++) maxlen = h_off[g + 1] - h_off[g] > maxlen ? h_off[g + 1] - h_off[g] : maxlen;
  uint32_t gx = (uint32_t)((maxlen / 16 + 255) / 256);
  if (gx < 1) gx = 1;
  if (gx > 4096) gx = 4096;
  // grid.y is limited to 65535: loop in slabs
  for (uint32_t g0 = 0; g0 < n; g0 += 65535) {
    uint32_t gy = n - g0 < 65535 ? n - g0 : 65535;
    hipLaunchKernelGGL(synth_kernel, dim3(gx, gy), dim3(256), 0, ctx->stream, d_desc + g0, d_off + g0, d_seq);
    RTC_CHECK_LAUNCH(ctx);
  }
  // h_desc/h_off are pageable: make the async copies safe to outlive the call
  RTC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return RTC_OK;
}

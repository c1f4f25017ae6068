// rtc_sketch_minhash_packed.hip -- bottom-s MinHash sketching straight from the 2-bit staging format (gfx950).
//
// The reference hands Sketch::MinHash::update() ASCII records (src/SketchInfo.cpp:928-948); the command lines stage
// 2-bit codes plus a run list of everything outside ACGT (rtc_host.cpp: PackedSink; include/rtclust.h,
// rtc_unpack_bases_dev for the layout).  rtc_sketch_minhash.hip reads characters -- the batch had to be expanded
// again in HBM in front of a kernel whose first act is to squeeze the characters back into two bits.  This unit is
// the same sketcher fed the packed stream as it crossed PCIe:
//   * a lane owns the 64 k-mer end positions of ONE 16-byte load; a wave's load is 1 KiB of contiguous stream
//     (4 096 bases), every line requested once.  The 32 bases in front of a lane's first come with one 8-byte load
//     of their own: no warm-up bases are walked at all (the ASCII kernel rolls 20 per 76 owned positions);
//   * a byte of the stream IS four bases: the forward window takes the byte with its pairs reversed (one v_bfrev and
//     three bit operations per 16 bases), the reverse-complement window the complemented byte as it lies (the
//     stream's order is the reverse strand's); both windows roll by one v_perm + one v_alignbit -- no SWAR decode,
//     no re-encoding votes, no v_dot4;
//   * what the characters were is the run list's business: a k-mer counts exactly when none of its k bases lies in
//     a run or outside its genome.  A wave asks once per load whether any run can touch its 4 096 bases (a scalar
//     cursor into the segment's run range, minhash_seg_runs_kernel); only a wave that meets one builds per-lane
//     validity masks and takes the general walk.
// The hash (MurmurHash3 from LDS product tables), the threshold test, the candidate queue, the in-LDS merges, the
// segment plan and the partial-sketch merge are rtc_minhash_core.h's, shared with the ASCII unit; results are
// identical to it and to the oracle bit for bit (tests/test_gpu_sketch_minhash_packed.py).
#include "rtc_minhash_core.h"

namespace {

constexpr int P_NL = 1;                        // 16-byte loads (64 bases) per lane and tile
constexpr int P_CHUNK = 64 * 64;               // bases of one wave load
constexpr int P_TILE_BASES = WG * 64 * P_NL;   // bases per tile

struct PackedIn {
  const uint8_t* bytes;     // packed bases: base i at bits 2 (i & 3) of bytes[i >> 2]
  uint64_t n_bases;         // a multiple of 64
  const uint64_t* runs;     // (start, length) pairs, ascending and disjoint
};

// per segment, the first run that ends behind s_begin - (k - 1) and the first that starts at or behind s_end
__global__ __launch_bounds__(256) void minhash_seg_runs_kernel(const Segment* __restrict__ segs, uint32_t nseg, const uint64_t* __restrict__ runs,
                                                               uint32_t n_runs, int k, uint2* __restrict__ seg_runs) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nseg) return;
  const int64_t first = (int64_t)segs[s].s_begin - (k - 1), end = (int64_t)segs[s].s_end;
  uint32_t lo = 0, hi = n_runs;  // runs that end at or before `first` (ends ascend with the starts: the runs are disjoint)
  while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((int64_t)(runs[2 * (uint64_t)mid] + runs[2 * (uint64_t)mid + 1]) <= first) lo = mid + 1; else hi = mid; }
  const uint32_t x = lo;
  hi = n_runs;                   // runs that start before `end`
  while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((int64_t)runs[2 * (uint64_t)mid] < end) lo = mid + 1; else hi = mid; }
  seg_runs[s] = make_uint2(x, lo);
}

// 16 packed bases (first base in the low bits) -> first base on top, every base's two bits in order
__device__ __forceinline__ uint32_t pair_rev(uint32_t x) {
  const uint32_t y = __brev(x);  // pairs in order, the two bits of a pair swapped
  return ((y << 1) & 0xAAAAAAAAu) | ((y >> 1) & 0x55555555u);
}

// v_perm selector of the forward window's roll: new low word = (low word << 8) | byte 3 - q of the pair-reversed dword at
// the byte boundary fsb (operands: S0 = low word -> bytes 4..7, S1 = the dword -> bytes 0..3; 0x0c = a zero byte)
__host__ __device__ constexpr uint32_t fwd_roll_sel(int fsb, int q) {
  uint32_t sel = 0;
  for (int j = 0; j < 4; j++) sel |= (j > fsb / 8 ? (uint32_t)(3 + j) : j == fsb / 8 ? (uint32_t)(3 - q) : 0x0cu) << (8 * j);
  return sel;
}

template <int KT, bool PK>  // as sketch_minhash_kernel (rtc_sketch_minhash.hip)
__global__ __launch_bounds__(WG, 6) void sketch_minhash_packed_kernel(PackedIn B, const Segment* __restrict__ segs,
                                                                   const uint2* __restrict__ seg_runs,
                                                                   int k_arg, uint32_t seed, int cap,
                                                                   uint64_t* out,
                                                                   uint32_t* cnt, int pass_no,
                                                                   uint64_t* parts, uint32_t* pcnt,
                                                                   const uint32_t* __restrict__ redo) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int k = KT > 0 ? KT : k_arg;
  const lds_byte_ptr lds0 = (lds_byte_ptr)smem;
  const lds_byte_ptr lut = lds0;  // at LDS offset 0: table offsets become ds_read immediates
  if ((uint32_t)(uintptr_t)lds0 != 0u) __builtin_trap();
  const lds_u64_ptr buf = (lds_u64_ptr)(lds0 + lut_bytes(k, PK));
  const lds_ctrl_ptr ctrl = (lds_ctrl_ptr)(lds0 + lut_bytes(k, PK) + (size_t)cap * 8);
  const lds_u64_ptr wq = (lds_u64_ptr)(lds0 + lut_bytes(k, PK) + (size_t)cap * 8 + ((sizeof(Ctrl) + 15) & ~(size_t)15)) +
                         (size_t)(threadIdx.x >> 6) * QCAP * 2;
  uint32_t qn = 0;  // entries waiting in this wave's candidate queue (wave-uniform)

  const Segment sg = segs[blockIdx.x];
  if (redo && redo[sg.final_slot] == 0) return;  // workgroup-uniform (second launch: flagged genomes only)
  const KParams P = make_kparams(k, KT > 0 ? MASH_SEED : seed, PK);
  const int t = threadIdx.x;
  const uint32_t lane = t & 63;
  const int wv = (int)uniform32((uint32_t)(t >> 6));
  const uint32_t s = sg.sketch_size;
  const uint2 sr = seg_runs[blockIdx.x];
  const bool has_runs = sr.x != sr.y;  // workgroup-uniform: most segments of a finished genome meet no run at all
  const int64_t nbytes = (int64_t)(B.n_bases >> 2);

  uint64_t lo1 = 0;  // later passes of a large sketch: only hashes above everything kept so far
  if (pass_no > 0) {  // workgroup-uniform
    const bool live = cnt[sg.final_slot] == sg.expect;
    const uint64_t lo = live ? out[sg.lo_off] : SENT;
    if (!live || lo == SENT) {
      if (t == 0 && sg.partial) pcnt[sg.cnt_slot] = 0;
      return;
    }
    lo1 = lo + 1;
  }

  uint64_t Tstart = (pass_no == 0 && !redo) ? sg.t0 : SENT;  // starting threshold, see sketch_minhash_kernel
restart:
  if (t == 0) { ctrl->T = Tstart; ctrl->T0 = Tstart; ctrl->sorted = 0; ctrl->count = 0; ctrl->overflow = 0; ctrl->saw_max = 0; ctrl->scan_base = 0; }
  build_kmer_lut(lut, k, PK);
  __syncthreads();

  uint64_t T = uniform64(Tstart);
  qn = 0;
  bool safe_mode = true;
  const uint32_t room = (uint32_t)cap - s;  // >= MIN_ROOM by construction
  uint32_t rcur = sr.x;  // wave-uniform cursor into the run list: every run in front of it ends before anything this wave still looks at

  auto drain_queue = [&]() {  // as in sketch_minhash_kernel
    if (qn == 0) return;
    HashParts qp{0, 0};
    uint64_t h = 0;
    bool okq = false;
    if (lane < qn) {
      qp = HashParts{wq[2 * lane], wq[2 * lane + 1]};
      h = mm_finish(qp);
      okq = h < T || T == SENT;
    }
    const uint64_t bal = __ballot(okq);
    uint32_t left = 0;
    if (bal) {
      uint32_t base = 0;
      if (lane == 0) base = __hip_atomic_fetch_add(&ctrl->count, (uint32_t)__popcll(bal), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      base = __shfl(base, 0);
      const uint32_t idx = base + (uint32_t)__popcll(bal & ((1ULL << lane) - 1ULL));
      const bool fits = idx < (uint32_t)cap;
      if (okq && fits) buf[idx] = h;
      const uint64_t fbal = __ballot(okq && !fits);
      if (fbal) {
        if (okq && !fits) {
          const uint32_t slot = (uint32_t)__popcll(fbal & ((1ULL << lane) - 1ULL));
          wq[2 * slot] = qp.f1;
          wq[2 * slot + 1] = qp.f2;
          ctrl->overflow = 1;
        }
        left = (uint32_t)__popcll(fbal);
      }
    }
    qn = left;
  };

  uint32_t count_at_tile_start = 0;
  for (uint64_t TB = sg.s_begin & ~63ULL; TB < sg.s_end && s > 0; TB += P_TILE_BASES) {
    const int64_t lo64 = (int64_t)sg.s_begin - (int64_t)TB;
    const int64_t hi64 = (int64_t)sg.s_end - (int64_t)TB;
    const int rel_lo = lo64 < 0 ? 0 : (int)lo64;
    const int rel_hi = hi64 > P_TILE_BASES ? P_TILE_BASES : (int)hi64;
    const bool interior = rel_lo == 0 && rel_hi == P_TILE_BASES;  // every position of the tile is owned
    const int64_t gb64 = (int64_t)sg.g_begin - (int64_t)TB, ge64 = (int64_t)sg.g_end - (int64_t)TB;
    const int gb = gb64 < -(1 << 30) ? -(1 << 30) : (int)gb64;   // genome extent in tile coordinates
    const int ge = ge64 > (1 << 30) ? (1 << 30) : (int)ge64;
    const int64_t tile_byte = (int64_t)(TB >> 2);                 // wave-uniform
    const uint32_t rcur_tile = rcur;                              // a tile walked again starts from here again

    bool redo_tile;
    do {
      redo_tile = false;
      rcur = rcur_tile;
#pragma unroll 1
      for (int j = 0; j < P_NL; j++) {
        const int crel = (wv * P_NL + j) * P_CHUNK;   // this wave's load: first base relative to the tile (wave-uniform)
        const int lrel = crel + 64 * (int)lane;       // the lane's first owned position
        // ---- the lane's 64 bases and the 32 in front of them (zeros outside the buffer: never part of a counted k-mer) ----
        uint32_t cw[4] = {0u, 0u, 0u, 0u}, p2 = 0u, p3 = 0u;
        {
          const int64_t byte = tile_byte + (lrel >> 2);
          if (byte + 16 <= nbytes) {
            const uint4 v = *reinterpret_cast<const uint4*>(B.bytes + byte);
            cw[0] = v.x; cw[1] = v.y; cw[2] = v.z; cw[3] = v.w;
          }
          if (byte >= 8 && byte <= nbytes) {
            const uint2 v = *reinterpret_cast<const uint2*>(B.bytes + byte - 8);
            p2 = v.x; p3 = v.y;
          }
        }
        // ---- can a run touch this wave's bases?  (scalar: the cursor only moves forward) ----
        bool wave_dirty = false;
        if (has_runs) {
          const int64_t first = (int64_t)TB + crel - 32;  // runs that end at or before it are behind this wave for good
          uint32_t rc = rcur;
          while (rc < sr.y) {
            const uint64_t en = B.runs[2 * (uint64_t)rc] + B.runs[2 * (uint64_t)rc + 1];
            if ((int64_t)en > first) break;
            rc++;
            if (rc - rcur == 8) {  // many runs behind: the rest by bisection
              uint32_t lo = rc, hi = sr.y;
              while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((int64_t)(B.runs[2 * (uint64_t)mid] + B.runs[2 * (uint64_t)mid + 1]) <= first) lo = mid + 1; else hi = mid; }
              rc = lo;
              break;
            }
          }
          rcur = uniform32(rc);
          wave_dirty = rcur < sr.y && (int64_t)B.runs[2 * (uint64_t)rcur] < (int64_t)TB + crel + P_CHUNK;
        }

        bool done = false;  // wave-uniform: the express walk took this load
        if constexpr (KT > 16 && KT <= 28) {
          // The steady state: a wave whose 4 096 bases (and the k - 1 in front) lie inside the genome and clear of runs,
          // in a tile interior to the segment, outside safe mode, with a threshold whose high word decides, walks its
          // 64 k-mers per lane as one software pipeline (the table reads of k-mer n + 1 in flight under the arithmetic
          // of k-mer n) -- windows, hash halves, high-word test, possible candidates to the queue.  A full queue hands
          // the whole load to the general walk.
          constexpr int FS = 58 - 2 * KT;           // where a new byte enters the forward window kept top-aligned for a dword's first k-mer
          constexpr int FSB = FS & ~7, FX = FS & 7;  // ... kept at the byte boundary below it; the rest is part of every cut
          const uint32_t Thi_e = (uint32_t)(T >> 32);
          if (!safe_mode && !lo1 && interior && Thi_e < 0xffffffffu - TEST_SLACK && !wave_dirty && crel - (KT - 1) >= gb && crel + P_CHUNK <= ge) {
            const uint32_t Thi1 = Thi_e + TEST_SLACK;
            const uint32_t qn0 = qn;
            // both windows from the 32 bases in front: forward F << FSB (first base on top), reverse complement with the
            // newest base's complement on top -- the complemented stream as it lies
            const uint32_t q2 = pair_rev(p2), q3 = pair_rev(p3);
            uint32_t FThi = FSB ? __builtin_amdgcn_alignbit(q2, q3, 32 - FSB) : q2;
            uint32_t FTlo = FSB ? (q3 << FSB) : q3;
            uint32_t Rhi = ~p3, Rlo = ~p2;
            uint32_t w0 = cw[0], w1 = cw[1], w2 = cw[2], w3 = cw[3];
            bool lost = false;  // wave-uniform: the queue could not take a candidate
            KmerLoads pend = {};
            auto finish_pending = [&]() __attribute__((always_inline)) {
              const HashParts hp = kmer_hash_finish(pend, P);
              const uint64_t mq = __ballot(hash_test_word(hp) <= Thi1);
              if (__builtin_expect(mq != 0, 0)) {  // wave-uniform, rare
                const uint32_t add = (uint32_t)__popcll(mq);
                if (qn + add <= (uint32_t)QCAP) {
                  if (__builtin_amdgcn_inverse_ballot_w64(mq)) {
                    const uint32_t slot = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(mq >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mq, 0u));
                    wq[2 * slot] = hp.f1;
                    wq[2 * slot + 1] = hp.f2;
                  }
                  qn += add;
                } else {
                  lost = true;
                }
              }
            };
#pragma unroll 1
            for (int d = 0; d < 4; d++) {
              const uint32_t PRd = pair_rev(w0), NCd = ~w0;
#pragma unroll
              for (int q = 0; q < 4; q++) {
                // forward: (F << 8 | byte) at the byte boundary FSB; reverse: the complemented byte enters on top
                FThi = __builtin_amdgcn_alignbit(FThi, FTlo, 24);
                FTlo = __builtin_amdgcn_perm(FTlo, PRd, fwd_roll_sel(FSB, q));
                const uint32_t nhi = __builtin_amdgcn_perm(Rhi, NCd, ((uint32_t)q << 24) | 0x00070605u);
                Rlo = __builtin_amdgcn_alignbit(Rhi, Rlo, 8);
                Rhi = nhi;
                const uint64_t FT = ((uint64_t)FThi << 32) | FTlo;
                const uint64_t R = ((uint64_t)Rhi << 32) | Rlo;
#pragma unroll
                for (int b = 0; b < 4; b++) {
                  const uint64_t f = FT << (2 * b + FX);
                  const uint64_t r = R << (6 - 2 * b);
                  const KmerLoads nl = kmer_loads(f < r ? f : r, P);
                  __builtin_amdgcn_sched_barrier(0);
                  if (q > 0 || b > 0 || d > 0) finish_pending();
                  __builtin_amdgcn_sched_barrier(0);
                  pend = nl;
                }
              }
              w0 = w1; w1 = w2; w2 = w3;
            }
            finish_pending();
            if (lost) qn = qn0;  // the candidates this load did queue are found again by the general walk
            else done = true;
          }
        }

        // ---- the general walk: every k, runs, genome and segment edges, safe mode, later passes ----
        if (!done) {  // (safe mode never takes the express walk: every wave meets the barriers of all sixteen steps)
          // validity of the lane's 96 bases: bit i of (M2 : M1 : M0) set = base lrel - 32 + i lies in a run or outside the genome
          uint32_t M[3] = {0u, 0u, 0u};
          const int wstart = lrel - 32;  // tile coordinates
          auto mark = [&](int a, int b) {  // bases [a, b) of the window
            a = a < 0 ? 0 : a;
            b = b > 96 ? 96 : b;
            if (a >= b) return;
#pragma unroll
            for (int w = 0; w < 3; w++) {
              const int la = a - 32 * w < 0 ? 0 : a - 32 * w, lb = b - 32 * w > 32 ? 32 : b - 32 * w;
              if (la < lb) M[w] |= (lb - la == 32) ? ~0u : (((1u << (lb - la)) - 1u) << la);
            }
          };
          if (gb > wstart) mark(0, gb - wstart);
          if (ge < wstart + 96) mark(ge - wstart, 96);
          if (wave_dirty) {
            const int64_t wabs = (int64_t)TB + wstart;
            uint32_t lo = rcur, hi = sr.y;  // the first run that ends behind the window's first base
            while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((int64_t)(B.runs[2 * (uint64_t)mid] + B.runs[2 * (uint64_t)mid + 1]) <= wabs) lo = mid + 1; else hi = mid; }
            for (uint32_t r = lo; r < sr.y; r++) {
              const int64_t st = (int64_t)B.runs[2 * (uint64_t)r] - wabs;
              if (st >= 96) break;
              const int64_t en = st + (int64_t)B.runs[2 * (uint64_t)r + 1];
              mark(st < 0 ? 0 : (int)st, en > 96 ? 96 : (int)en);
            }
          }
          const bool clean = !__any((M[0] | M[1] | M[2]) != 0u);  // wave-uniform
          // windows in the general form (sketch_minhash_kernel's fwd / rc) from the 32 bases in front
          uint64_t fwd = ((uint64_t)pair_rev(p2) << 32) | pair_rev(p3);
          uint64_t rc;
          {
            const uint64_t nc = ~(((uint64_t)p3 << 32) | p2);
            rc = k == 32 ? nc : (nc >> (64 - 2 * k));
          }
          uint32_t g0 = cw[0], g1 = cw[1], g2 = cw[2], g3 = cw[3];
#pragma unroll 1
          for (int d = 0; d < 4; d++) {
            const uint32_t wd = g0;
            g0 = g1; g1 = g2; g2 = g3;
            const uint64_t Wm = d < 2 ? (((uint64_t)M[1] << 32) | M[0]) : (((uint64_t)M[2] << 32) | M[1]);
#pragma unroll
            for (int q = 0; q < 4; q++) {
              const int i0 = 16 * d + 4 * q;           // the step's first position among the lane's 64
              const int rel0 = lrel + i0;              // ... in tile coordinates
              const uint32_t y = (wd >> (8 * q)) & 0xffu;  // four bases, the first lowest
              const uint32_t pack = ((y & 3u) << 6) | ((y & 0xcu) << 2) | ((y >> 2) & 0xcu) | (y >> 6);
              const uint32_t rp = y ^ 0xffu;
              uint64_t canon[4];
              if (k > 28) {
                typedef unsigned __int128 u128;
                const u128 F = ((u128)fwd << 8) | pack;
                const u128 R = (u128)rc | ((u128)rp << (2 * P.k));
#pragma unroll
                for (int b = 0; b < 4; b++) {
                  const uint64_t f = (uint64_t)(F >> (6 - 2 * b)) & P.kmask;
                  const uint64_t r = (uint64_t)(R >> (2 * b + 2)) & P.kmask;
                  canon[b] = (f < r ? f : r) << P.lshift;
                }
                fwd = (uint64_t)F;
                rc = (uint64_t)(R >> 8);
              } else {
                const uint64_t F = (fwd << 8) | pack;
                const uint64_t R = rc | ((uint64_t)rp << (2 * P.k));
#pragma unroll
                for (int b = 0; b < 4; b++) {
                  const uint64_t f = F << (P.lshift - 6 + 2 * b);   // lshift >= 8 here; bits below the window stay (see sketch_minhash_kernel)
                  const uint64_t r = R << (P.lshift - 2 - 2 * b);
                  canon[b] = f < r ? f : r;
                }
                fwd = F;
                rc = R >> 8;
              }
              const bool allok = interior && clean;  // wave-uniform: every k-mer of every lane is valid and owned
              // a k-mer that ends at position i of the lane's 64 is valid when the k bits up to bit 32 + i of the mask are clear
              bool ok[4];
#pragma unroll
              for (int b = 0; b < 4; b++) {
                const int i = (i0 & 31) + b;  // position inside Wm's upper word
                const uint64_t win = (Wm >> (33 + i - k)) & (k == 32 ? 0xffffffffULL : ((1ULL << k) - 1ULL));
                const int rel = rel0 + b;
                ok[b] = win == 0 && rel >= rel_lo && rel < rel_hi;
              }
              // what this wave appends directly (not through its queue): filled by the branches below, appended behind them --
              // in safe mode one k-mer per lane at a time, every wave meeting the same barriers whether it appends or not
              uint64_t am[4] = {0, 0, 0, 0}, ah[4] = {0, 0, 0, 0};
              auto append1 = [&](uint64_t bal, uint64_t hv) __attribute__((always_inline)) {
                if (bal) {  // wave-uniform
                  uint32_t base = 0;
                  if (lane == 0) base = __hip_atomic_fetch_add(&ctrl->count, (uint32_t)__popcll(bal), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                  base = __shfl(base, 0);
                  const uint32_t idx = base + (uint32_t)__popcll(bal & ((1ULL << lane) - 1ULL));
                  if ((bal >> lane) & 1ULL) {
                    if (idx < (uint32_t)cap) buf[idx] = hv;
                    else ctrl->overflow = 1;
                  }
                }
              };
              const uint32_t Thi = (uint32_t)(T >> 32);
              if (allok && P.use64 && !lo1 && Thi < 0xffffffffu - TEST_SLACK) {
                // the high-word test of the steady state (sketch_minhash_kernel), without the pipeline
                HashParts hp[4];
#pragma unroll
                for (int b = 0; b < 4; b++) hp[b] = kmer_hash_parts(canon[b], P);
                const uint32_t Thi1 = Thi + TEST_SLACK;
                uint64_t cm = 0, mq[4];
#pragma unroll
                for (int b = 0; b < 4; b++) {
                  mq[b] = __ballot(hash_test_word(hp[b]) <= Thi1);
                  cm |= mq[b];
                }
                if (cm) {  // wave-uniform, rare
                  const uint32_t add = (uint32_t)(__popcll(mq[0]) + __popcll(mq[1]) + __popcll(mq[2]) + __popcll(mq[3]));
                  if (qn + add <= (uint32_t)QCAP) {
                    uint32_t qb = qn;
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                      if (mq[b]) {  // wave-uniform
                        if ((mq[b] >> lane) & 1ULL) {
                          const uint32_t slot = qb + (uint32_t)__popcll(mq[b] & ((1ULL << lane) - 1ULL));
                          wq[2 * slot] = hp[b].f1;
                          wq[2 * slot + 1] = hp[b].f2;
                        }
                        qb += (uint32_t)__popcll(mq[b]);
                      }
                    }
                    qn = qb;
                  } else {  // queue full: finish and append on the spot
                    uint64_t h[4], m[4];
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                      HashParts qh = hp[b];
                      asm volatile("" : "+v"(qh.f1), "+v"(qh.f2));  // keeps the finishing arithmetic inside this branch
                      h[b] = mm_finish(qh);
                      m[b] = __ballot(h[b] < T);
                      am[b] = m[b]; ah[b] = h[b];
                    }
                  }
                }
              } else {
                uint64_t h[4], m[4];
#pragma unroll
                for (int b = 0; b < 4; b++) h[b] = kmer_hash(canon[b], P);
                if (allok && T != SENT) {
#pragma unroll
                  for (int b = 0; b < 4; b++) m[b] = __ballot(h[b] < T);
                } else {
#pragma unroll
                  for (int b = 0; b < 4; b++) m[b] = __ballot(ok[b] && (h[b] < T || T == SENT));  // T == SENT: sketch not full yet, everything passes
                }
                if (lo1) {
#pragma unroll
                  for (int b = 0; b < 4; b++) m[b] &= __ballot(h[b] >= lo1);
                }
#pragma unroll
                for (int b = 0; b < 4; b++) { am[b] = m[b]; ah[b] = h[b]; }
              }
              if (safe_mode) {
#pragma unroll
                for (int b = 0; b < 4; b++) {
                  // bound the next appends (at most one per lane) so the buffer cannot overflow
                  __syncthreads();
                  const uint32_t cn = uniform32(ctrl->count);
                  if ((uint32_t)cap - cn < (uint32_t)STEP_APPENDS) T = uniform64(merge_block(buf, ctrl, cap, s).T);
                  __syncthreads();
                  append1(am[b], ah[b]);
                }
              } else if (am[0] | am[1] | am[2] | am[3]) {
#pragma unroll
                for (int b = 0; b < 4; b++) append1(am[b], ah[b]);
              }
            }
          }
        }
      }
      __syncthreads();
      if (uniform32(ctrl->overflow)) {
        // the optimistic pass lost candidates: fold what we have, then walk this tile again in safe mode
        const MergeResult mr = merge_block(buf, ctrl, cap, s);
        count_at_tile_start = uniform32(mr.count);
        T = uniform64(mr.T);
        safe_mode = true;
        redo_tile = true;
      }
    } while (redo_tile);

    // ---- end of tile: decide about merging and the next tile's mode (sketch_minhash_kernel's protocol) ----
    const uint32_t cn = uniform32(ctrl->count);
    const uint32_t appended = cn - (count_at_tile_start < cn ? count_at_tile_start : cn);
    const uint32_t half = (uint32_t)cap / 2;
    const bool need_merge = cn > ((half > s + 512 && half < s + room / 2) ? half : s + room / 2);
    safe_mode = appended > room / 4;
    __syncthreads();
    if (need_merge) {
      const MergeResult mr = merge_block(buf, ctrl, cap, s);
      count_at_tile_start = uniform32(mr.count);
      T = uniform64(mr.T);
    }
    else count_at_tile_start = cn;
    if (qn >= (uint32_t)QDRAIN) drain_queue();
  }

  // ---- final fold and write-out ----
  drain_queue();
  {
    const MergeResult mr = merge_block(buf, ctrl, cap, s);
    T = uniform64(mr.T);
  }
  drain_queue();
  uint32_t nfin = merge_block(buf, ctrl, cap, s).count;
  if (nfin < s && Tstart != SENT && !sg.partial) {  // the starting threshold was too optimistic for this genome
    Tstart = SENT;
    __syncthreads();
    goto restart;
  }
  uint64_t* o = (sg.partial ? parts : out) + sg.out_off;
  for (uint32_t i = t; i < nfin; i += WG) o[i] = buf[i];
  if (t == 0) {
    if (ctrl->saw_max && nfin < s) { o[nfin] = SENT; nfin++; }
    if (sg.partial) pcnt[sg.cnt_slot] = nfin;
    else cnt[sg.cnt_slot] = pass_no > 0 ? sg.expect + nfin : nfin;
  }
}

}  // namespace

extern "C" int rtc_sketch_minhash_packed_dev(rtc_ctx* ctx, const uint8_t* d_packed, uint64_t n_bases, const uint64_t* d_runs,
                                             uint64_t n_runs, const uint64_t* h_off, uint32_t n, int k, uint32_t seed,
                                             const uint32_t* h_sizes, uint32_t size, uint64_t* d_out, uint32_t stride,
                                             uint32_t* d_cnt) {
  if (!ctx || !h_off || (n && (!d_packed || !d_out || !d_cnt)) || (n_runs && !d_runs)) return RTC_ERR_ARG;
  if (k < 1 || k > 32) return rtc_fail(ctx, RTC_ERR_ARG, "k=%d outside 1..32", k);
  if (n == 0) return RTC_OK;
  if (((uintptr_t)d_packed & 15) != 0) return rtc_fail(ctx, RTC_ERR_ARG, "d_packed must be 16-byte aligned");
  if ((n_bases & 63) || n_runs >= (1ull << 32))
    return rtc_fail(ctx, RTC_ERR_ARG, "packed batch: n_bases must be a multiple of 64, fewer than 2^32 runs");
  if (h_off[n] > n_bases) return rtc_fail(ctx, RTC_ERR_ARG, "packed batch: the genomes end at base %llu, the buffer holds %llu", (unsigned long long)h_off[n], (unsigned long long)n_bases);

  RTC_TRY(rtc_sticky_error(ctx));
  RTC_TRY(rtc_check_runs_async(ctx, d_runs, n_runs, n_bases));  // asynchronous: a violation surfaces at the next packed call or rtc_ctx_sync
  typedef void (*kern_t)(PackedIn, const Segment*, const uint2*, int, uint32_t, int, uint64_t*, uint32_t*, int, uint64_t*, uint32_t*, const uint32_t*);
  auto pick = [&](bool runtime_k, bool packed) -> kern_t {
    kern_t kern = packed ? sketch_minhash_packed_kernel<0, true> : sketch_minhash_packed_kernel<0, false>;
    switch (!runtime_k && seed == MASH_SEED ? k : 0) {
#define RTC_K(K) case K: kern = packed ? sketch_minhash_packed_kernel<K, true> : sketch_minhash_packed_kernel<K, false>; break;
      RTC_K(16) RTC_K(17) RTC_K(18) RTC_K(19) RTC_K(20) RTC_K(21) RTC_K(22) RTC_K(23) RTC_K(24)
      RTC_K(25) RTC_K(26) RTC_K(27) RTC_K(28) RTC_K(29) RTC_K(30) RTC_K(31) RTC_K(32)
#undef RTC_K
      default: break;
    }
    return kern;
  };
  PackedIn B{d_packed, n_bases, d_runs};
  uint2* d_seg_runs = nullptr;
  auto prepare = [&](const MinhashPlanInfo& pi) -> int {
    RTC_HIP(ctx, hipFuncSetAttribute((const void*)pick(false, pi.packed_tables), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pi.lds));
    RTC_HIP(ctx, hipFuncSetAttribute((const void*)pick(true, pi.packed_tables), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pi.lds));
    void* ws = nullptr;
    RTC_TRY(rtc_ws(ctx, 3, pi.nsegs * sizeof(uint2) + 64, &ws));
    d_seg_runs = (uint2*)ws;
    if (ctx->opt.verbose && !ctx->quiet) fprintf(stderr, "[minhash] sketching over packed bases, k=%d, %zu segments, %llu runs\n", k, pi.nsegs, (unsigned long long)n_runs);
    hipLaunchKernelGGL(minhash_seg_runs_kernel, dim3((uint32_t)((pi.nsegs + 255) / 256)), dim3(256), 0, ctx->stream, pi.d_segs, (uint32_t)pi.nsegs,
                       d_runs, (uint32_t)n_runs, k, d_seg_runs);
    RTC_CHECK_LAUNCH(ctx);
    return RTC_OK;
  };
  auto launch = [&](const MinhashLaunch& L) -> int {
    hipLaunchKernelGGL(pick(L.runtime_k, L.packed_tables), dim3(L.nseg), dim3(WG), L.lds, ctx->stream, B, L.d_segs,
                       (const uint2*)(d_seg_runs + L.seg_index), k, seed, L.cap, d_out, d_cnt, L.pass, L.d_parts, L.d_pcnt, L.d_redo);
    RTC_CHECK_LAUNCH(ctx);
    return RTC_OK;
  };
  return minhash_run(ctx, h_off, n, k, h_sizes, size, d_out, stride, d_cnt, (uint64_t)P_TILE_BASES, (size_t)MIN_ROOM, prepare, launch);
}

namespace { __global__ void touch_unit_kernel() {} }
int rtc_touch_sketch_minhash_packed(rtc_ctx* ctx) {
  hipLaunchKernelGGL(touch_unit_kernel, dim3(1), dim3(64), 0, ctx->stream);
  RTC_CHECK_LAUNCH(ctx);
  return RTC_OK;
}

// rtc_sketch_minhash.hip -- bottom-s MinHash sketching on gfx950.
//
// Replaces Sketch::MinHash::{update,storeMinHashes} driven from src/SketchInfo.cpp:918-942,969
// (reference tree paths).  One workgroup (512 lanes = 8 wave64) walks one *segment* of a genome:
//   * every lane owns 76 consecutive k-mer end positions of a 38 KiB tile and reads its 96 bases
//     (20 warm-up + 76 owned; 112 = 36 + 76 for k > 21) straight from global memory as 16-byte loads, decoding four
//     bases at once and rolling the 2-bit forward / reverse-complement words;
//   * MurmurHash3_x64_128 of the canonical k-mer's ASCII bytes is evaluated from the 2-bit word:
//     the first multiplication of every input word comes out of LDS product tables (linearity of
//     multiplication mod 2^64), leaving seven 64x64 multiplies per k-mer (72 VALU instructions
//     for k = 21, with hand-picked forms for rotates, x5 and the table offsets);
//   * hashes below the running threshold T (kept in SGPRs) are appended to an LDS candidate buffer
//     with one LDS atomic per wave; when the buffer fills, an in-LDS bitonic sort over the live
//     count + dedup keeps the s smallest distinct values and lowers T.
// Large genomes / small batches are split into several segments whose partial sketches are
// merged by merge_partials_kernel (bottom-s is a mergeable summary).  The segments of a genome all start
// from the genome's threshold (3x the expected s-th smallest hash), as a whole-genome workgroup does; a
// genome whose merged partial sketches hold fewer than s hashes is flagged and walked once more without it
// (a second, gated launch over the partial segments -- every other workgroup of it leaves at once).
#include "rtc_minhash_core.h"

namespace {

template <int KT, bool PK>  // KT > 0: k known at compile time (uniform branches fold away); 0: runtime k.  PK: packed tables
// second launch bound: 6 waves/SIMD = 3 workgroups per CU (caps the allocation at 80 VGPRs)
__global__ __launch_bounds__(WG, 6) void sketch_minhash_kernel(const uint8_t* __restrict__ seq,
                                                            const Segment* __restrict__ segs,
                                                            int k_arg, uint32_t seed, int cap,
                                                            uint64_t* out,
                                                            uint32_t* cnt, int pass_no,
                                                            uint64_t* parts, uint32_t* pcnt,
                                                            const uint32_t* __restrict__ redo) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int k = KT > 0 ? KT : k_arg;
  constexpr int WARM_DW = warm_dw(KT);
  const lds_byte_ptr lds0 = (lds_byte_ptr)smem;
  const lds_byte_ptr lut = lds0;  // at LDS offset 0: table offsets become ds_read immediates
  // kmer_hash addresses the tables by absolute LDS address; this kernel has no static LDS, so the
  // dynamic allocation starts at 0 -- trap rather than hash with wrong tables if that ever changes
  if ((uint32_t)(uintptr_t)lds0 != 0u) __builtin_trap();
  const lds_u64_ptr buf = (lds_u64_ptr)(lds0 + lut_bytes(k, PK));
  const lds_ctrl_ptr ctrl = (lds_ctrl_ptr)(lds0 + lut_bytes(k, PK) + (size_t)cap * 8);
  // this wave's candidate queue: QCAP x {f1, f2}
  const lds_u64_ptr wq = (lds_u64_ptr)(lds0 + lut_bytes(k, PK) + (size_t)cap * 8 + ((sizeof(Ctrl) + 15) & ~(size_t)15)) +
                         (size_t)(threadIdx.x >> 6) * QCAP * 2;
  uint32_t qn = 0;  // entries waiting in it (wave-uniform)

  const Segment sg = segs[blockIdx.x];
  // second launch over the partial segments: only the genomes whose merged partial sketches came out short of s
  // under the starting threshold (flagged by merge_partials_kernel) are walked again, from "everything passes"
  if (redo && redo[sg.final_slot] == 0) return;  // workgroup-uniform
  // compile-time-k instantiations serve the reference's seed only (MASH_SEED, the launch sends any other seed to the
  // runtime-k kernel): as an inline constant the two seed xors per k-mer stay fast-class VALU (an SGPR source makes
  // v_xor_b32 a 4.4-cycle instruction, profiles/r03_valu_issue_cost2.txt)
  const KParams P = make_kparams(k, KT > 0 ? MASH_SEED : seed, PK);
  const int t = threadIdx.x;
  const uint32_t lane = t & 63;
  const uint32_t s = sg.sketch_size;
  const bool fastroll = true;     // four bases per step: 64-bit extended windows for k <= 28, 128-bit ones above

  // Sketch sizes beyond one LDS buffer are selected in passes of ascending hash ranges: pass p only
  // admits hashes above the largest one kept so far (lo1 = that hash + 1; 0 in the first pass).
  uint64_t lo1 = 0;
  if (pass_no > 0) {  // workgroup-uniform
    const bool live = cnt[sg.final_slot] == sg.expect;  // genome not exhausted by earlier passes
    const uint64_t lo = live ? out[sg.lo_off] : SENT;
    if (!live || lo == SENT) {
      if (t == 0 && sg.partial) pcnt[sg.cnt_slot] = 0;  // nothing from this segment
      return;
    }
    lo1 = lo + 1;
  }

  // Starting threshold.  A whole-genome workgroup knows how many k-mers are coming: the s-th smallest of N
  // uniform hashes will be near 2^64 * s / N, so it starts at T0 = 3x that (2x for dense sketches; start_threshold, host side) instead of "everything passes".
  // This skips the first tiles' flood of candidates (a dozen merges under per-dword barriers: the cost that
  // grew with s -- 6 % of the kernel at s = 1000, 20 % with 1 Mbp genomes) and changes nothing in the
  // result as long as s distinct hashes below T0 exist (3 s expected); if fewer than s were found -- a
  // genome with few distinct k-mers -- the workgroup simply runs again from T0 = "none".
  // (A partial segment starts from the genome's T0 as well: the union of the segments' hashes below T0 holds the s
  // smallest of the genome whenever s of them exist; if not, the merge flags the genome for the second launch.)
  uint64_t Tstart = (pass_no == 0 && !redo) ? sg.t0 : SENT;
restart:
  if (t == 0) { ctrl->T = Tstart; ctrl->T0 = Tstart; ctrl->sorted = 0; ctrl->count = 0; ctrl->overflow = 0; ctrl->saw_max = 0; ctrl->scan_base = 0; }
  build_kmer_lut(lut, k, PK);
  __syncthreads();

  uint64_t T = uniform64(Tstart);  // scalar registers: the threshold compares write wave masks directly
  qn = 0;
  bool safe_mode = true;
  const uint32_t room = (uint32_t)cap - s;  // >= MIN_ROOM by construction

  // finishes the queued halves (one lane each), keeps those still below T and appends them with ONE LDS
  // atomic for the whole batch; called where cap - count >= NWAVE * QCAP is guaranteed
  auto drain_queue = [&]() {
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
      // No room (other waves filled the buffer meanwhile; cannot happen while the tile-end guarantee holds):
      // the entry stays queued and the overflow flag forces a merge -- nothing is ever dropped here.
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

  uint32_t count_at_tile_start = 0;  // carried in registers: identical in every thread
  for (uint64_t T0 = sg.s_begin & ~15ULL; T0 < sg.s_end && s > 0; T0 += TILE_BASES) {
    // owned positions of this lane relative to T0: [OWN*t, OWN*t + OWN); hash window limits
    const int64_t lo64 = (int64_t)sg.s_begin - (int64_t)T0;
    const int64_t hi64 = (int64_t)sg.s_end - (int64_t)T0;
    const int rel_lo = lo64 < 0 ? 0 : (int)lo64;
    const int rel_hi = hi64 > TILE_BASES ? TILE_BASES : (int)hi64;
    const bool interior = rel_lo == 0 && rel_hi == TILE_BASES;  // every position of the tile is owned
    const uint8_t* tile = seq + T0;                 // wave-uniform
    const int rq0 = OWN * t - 4 * WARM_DW;          // first base of this lane's window, relative to the tile
    const int64_t gb64 = (int64_t)sg.g_begin - (int64_t)T0, ge64 = (int64_t)sg.g_end - (int64_t)T0;
    const int gb = gb64 < -(1 << 30) ? -(1 << 30) : (int)gb64;   // genome extent in tile coordinates
    const int ge = ge64 > (1 << 30) ? (1 << 30) : (int)ge64;

    bool redo;
    do {
      redo = false;
      uint64_t fwd = 0, rc = 0;
      int run = 0;
      bool clean = true;  // wave-uniform: only valid bases in every lane of this wave so far in this pass
      int g0 = 0;
      if constexpr (KT > 16 && KT <= 28) {
        // The steady state without the general walk's per-dword decisions: a wave whose windows lie inside the
        // genome, in a tile interior to the segment, outside safe mode, with a threshold whose high word
        // decides (see below), walks whole 16-byte groups -- validity of the group in one vote, the windows,
        // the hash halves, the high-word test, possible candidates to the queue -- until a group holds a
        // character outside ACGTacgt or the queue is full; the general walk takes over from that group.
        constexpr int NG = (WARM_DW + RUN_DW) / 4;
        const int w0 = (int)uniform32((uint32_t)(t & ~63));
        const int wlo = OWN * w0 - 4 * WARM_DW, whi = OWN * (w0 + 63) - 4 * WARM_DW + 16 * NG;
        const uint32_t Thi_e = (uint32_t)(T >> 32);
        if (!safe_mode && !lo1 && interior && Thi_e < 0xffffffffu - TEST_SLACK && wlo >= gb && whi <= ge) {
          const uint8_t* base = (tile - LOAD_BIAS) + (uint32_t)(rq0 + LOAD_BIAS);
          const uint32_t Thi1 = Thi_e + TEST_SLACK;
          uint4 cur = *reinterpret_cast<const uint4*>(base), nxt1 = *reinterpret_cast<const uint4*>(base + 16);
          // Both extended windows are kept top-aligned for ONE k-mer of a dword, which is then cut without a shift:
          // the forward one as FT = F << FS (first k-mer of the dword on top; the new byte enters at bit FS), the
          // reverse-complement one as R << RE with the newest byte in the top byte (last k-mer of the dword on top):
          // one v_perm (high word) + one v_alignbit (low word) roll it.
          constexpr int FS = 58 - 2 * KT, RE = 56 - 2 * KT;
          static_assert(FS >= 0 && FS + 8 <= 32 && RE >= 0, "express walk: 17 <= k <= 28");
          constexpr uint32_t RSEL = 0x00070605u;  // {rp byte 0, Rhi bytes 3, 2, 1}
          uint64_t FT = 0;
          uint32_t Rhi = 0, Rlo = 0;
          g0 = NG;
#pragma unroll
          for (int g = 0; g < NG; g++) {
            uint4 nxt2 = cur;
            if (g + 2 < NG) nxt2 = *reinterpret_cast<const uint4*>(base + 16 * (g + 2));
            const uint32_t w[4] = {cur.x, cur.y, cur.z, cur.w};
            uint32_t codes[4], bad = 0;
#pragma unroll
            for (int qd = 0; qd < 4; qd++) {
              codes[qd] = ((w[qd] >> 1) ^ (w[qd] >> 2)) & 0x03030303u;
              bad = __builtin_amdgcn_bitop3_b32(bad, __builtin_amdgcn_perm(0u, 0x54474341u, codes[qd]), w[qd], 0xF6);  // bad | (perm ^ w)
            }
            if (__ballot((bad & 0xDFDFDFDFu) != 0u)) { g0 = g; break; }
            const uint64_t FT0 = FT;
            const uint32_t Rhi0 = Rhi, Rlo0 = Rlo, qn0 = qn;
            bool lost = false;  // wave-uniform: the queue could not take this group's candidates
            // The k-mers of a group run as a two-stage pipeline: the table reads of k-mer n + 1 are issued before the
            // arithmetic of k-mer n, so a wave does not park on every LDS round trip (the compiler's own order issues a
            // word's reads right in front of their use: three waits per k-mer); the pipeline drains at the group's end.
            KmerLoads pend = {};
            bool have = false;  // (folds away: everything here is unrolled)
            // finishes the pending k-mer: hash halves, high-word test, a possible candidate to the queue (per k-mer, so
            // that no halves stay live across the dword: the registers go to the reads in flight)
            auto finish_pending = [&]() __attribute__((always_inline)) {
              const HashParts hp = kmer_hash_finish(pend, P);
              const uint64_t mq = __ballot(hash_test_word(hp) <= Thi1);
              if (__builtin_expect(mq != 0, 0)) {  // wave-uniform, rare: kept out of line, the common path falls through
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
#pragma unroll
            for (int qd = 0; qd < 4; qd++) {
              const uint32_t pack = __builtin_amdgcn_udot4(codes[qd], 0x01041040u, 0u, false);
              // reverse-complement byte 255 - (c0 + 4 c1 + 16 c2 + 64 c3) as the LOW BYTE of a dot product with the
              // weights 256 - {1, 4, 16, 64} on top of 255 (only that byte is used: v_perm picks it)
              const uint32_t rp = __builtin_amdgcn_udot4(codes[qd], 0xC0F0FCFFu, 255u, false);
              FT = (FT << 8) | (uint64_t)(pack << FS);  // v_lshlrev_b64 + v_lshl_or_b32
              const uint32_t nhi = __builtin_amdgcn_perm(Rhi, rp, RSEL);
              Rlo = __builtin_amdgcn_alignbit(Rhi, Rlo, 8);
              Rhi = nhi;
              const uint64_t R = ((uint64_t)Rhi << 32) | Rlo;  // = (general walk's R) << RE
              if (g * 4 + qd >= WARM_DW) {
#pragma unroll
                for (int b = 0; b < 4; b++) {
                  const uint64_t f = FT << (2 * b);
                  const uint64_t r = R << (6 - 2 * b);
                  const KmerLoads nl = kmer_loads(f < r ? f : r, P);
                  __builtin_amdgcn_sched_barrier(0);
                  if (have) finish_pending();
                  __builtin_amdgcn_sched_barrier(0);
                  pend = nl;
                  have = true;
                }
              }
            }
            if (have) finish_pending();
            if (lost) { FT = FT0; Rhi = Rhi0; Rlo = Rlo0; qn = qn0; g0 = g; break; }
            cur = nxt1;
            nxt1 = nxt2;
          }
          fwd = FT >> FS;
          rc = (((uint64_t)Rhi << 32) | Rlo) >> (RE + 8);  // the general walk's form
          run = 16 * g0;
        }
      }
      uint4 nxt = g0 < (WARM_DW + RUN_DW) / 4 ? load_bases16(tile, rq0 + 16 * g0, gb, ge) : make_uint4(0u, 0u, 0u, 0u);
      for (int grp = g0; grp < (WARM_DW + RUN_DW) / 4; grp++) {
        const uint4 cur = nxt;
        if (grp + 1 < (WARM_DW + RUN_DW) / 4) nxt = load_bases16(tile, rq0 + 16 * (grp + 1), gb, ge);
        const uint32_t wv4[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
        for (int qd = 0; qd < 4; qd++) {
          const int d = grp * 4 + qd;
          const uint32_t wv = wv4[qd];
          const bool hashing = d >= WARM_DW;  // wave-uniform
          const int rel0 = OWN * t + 4 * (d - WARM_DW);
          // (zero-initialised on purpose: left uninitialised, the generated code issues the tile loads in an
          // order that re-reads 40 % more of the input from the fabric -- measured, tools/pmc_runlen.sh)
          uint64_t canon[4] = {0, 0, 0, 0};  // top-aligned (first base in bit 63); hashing dwords only
          bool ok[4] = {false, false, false, false};  // slow path only; the fast path derives it on demand
          bool allok = false;  // wave-uniform: all four k-mers of every lane are valid and owned
          // ---- decode four bases at once ----
          const uint32_t up = wv & 0xDFDFDFDFu;
          const uint32_t codes4 = ((wv >> 1) ^ (wv >> 2)) & 0x03030303u;  // A,C,G,T (either case) -> 0..3 per byte
          const bool allvalid = __builtin_amdgcn_perm(0u, 0x54474341u, codes4) == up;
          const bool fast = fastroll && __all(allvalid);  // wave-uniform
          clean = clean && fast;
          const int run_in = run;
          if (fast) {
            // pack = c0<<6|c1<<4|c2<<2|c3 ; rp = complement codes in reverse significance; both are
            // byte dot products of the four codes (v_dot4_u32_u8)
            const uint32_t pack = __builtin_amdgcn_udot4(codes4, 0x01041040u, 0u, false);
            const uint32_t rp = __builtin_amdgcn_udot4(codes4, 0x40100401u, 0u, false) ^ 0xffu;
            // bits of fwd above the window shift out when the windows are cut, so it carries unmasked
            if (k > 28) {
              // 2k + 8 bits do not fit 64: the same cuts on 128-bit extended windows (k = 29..32)
              typedef unsigned __int128 u128;
              const u128 F = ((u128)fwd << 8) | pack;
              const u128 R = (u128)rc | ((u128)rp << (2 * P.k));
              if (hashing) {
                allok = interior && clean;
#pragma unroll
                for (int b = 0; b < 4; b++) {
                  const uint64_t f = (uint64_t)(F >> (6 - 2 * b)) & P.kmask;
                  const uint64_t r = (uint64_t)(R >> (2 * b + 2)) & P.kmask;
                  canon[b] = (f < r ? f : r) << P.lshift;
                }
              }
              fwd = (uint64_t)F;
              rc = (uint64_t)(R >> 8);
              run += 4;
            } else {
              const uint64_t F = (fwd << 8) | pack;
              const uint64_t R = rc | ((uint64_t)rp << (2 * P.k));
              if (hashing) {
                // Scalar ownership test for the steady state: in a tile interior to the segment, a wave
                // that has seen only valid bases since the tile began has run = 4d >= 4*WARM_DW >= k-1 in every
                // lane, and every position of the tile is owned.  Anything else takes the per-lane test.
                allok = interior && clean;
                // the four windows are cut out of F / R already top-aligned (one shift + one mask each):
                // the order of two k-mers does not depend on the alignment, and the hash wants them there
                // Bits below the window are NOT cleared: they cannot change which of two different k-mers
                // is smaller (of two equal ones either will do), and the hash never sees them -- table
                // offsets are taken from whole bytes and the tables of a partially filled byte are built
                // from the k-mer's bases only (build_kmer_lut masks by byte count).
#pragma unroll
                for (int b = 0; b < 4; b++) {
                  const uint64_t f = F << (P.lshift - 6 + 2 * b);   // lshift >= 8 in this path
                  const uint64_t r = R << (P.lshift - 2 - 2 * b);
                  canon[b] = f < r ? f : r;
                }
              }
              fwd = F;        // masked by whoever needs exactly 2k bits (the per-base path below)
              rc = R >> 8;    // R < 2^(2k+8) by construction, so this is already < 2^(2k)
              run += 4;
            }
          } else {
#pragma unroll
            for (int b = 0; b < 4; b++) {
              const uint32_t c = (wv >> (8 * b)) & 0xffu;
              const uint32_t code = ((c >> 1) ^ (c >> 2)) & 3u;
              const bool valid = ((c & 0xC0u) == 0x40u) && ((0x0010008Au >> (c & 31u)) & 1u);
              fwd = ((fwd << 2) | code) & P.kmask;
              rc = (rc >> 2) | ((uint64_t)(code ^ 3u) << P.rc_shift);
              run = valid ? run + 1 : 0;
              const int rel = rel0 + b;
              ok[b] = run >= P.k && rel >= rel_lo && rel < rel_hi;
              canon[b] = (fwd < rc ? fwd : rc) << P.lshift;
            }
          }
          if (hashing) {  // wave-uniform
            // appends the k-mers selected by the wave masks m[] (T is scalar: compares write the masks directly)
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
              // The steady state: hash = fin(f1) + fin(f2) where fin() touches the low word only, so
              // hi(hash) = hi(f1) + hi(f2) + carry.  The test word w (hash_test_word) is hi(hash) + {0, 1, 2}: with
              // w > hi(T) + TEST_SLACK the hash cannot be below T -- one 32-bit compare per k-mer and the halves'
              // last multiply is never formed (T != SENT here since hi(T) < 2^32 - 1 - TEST_SLACK).  The few
              // waves holding a possible candidate (~64 s / N of them) finish exactly.
              // four independent hash chains: their LDS table reads and multiplies overlap
              HashParts hp[4];
#pragma unroll
              for (int b = 0; b < 4; b++) hp[b] = kmer_hash_parts(canon[b], P);
              const uint32_t Thi1 = Thi + TEST_SLACK;
              uint64_t cm = 0, mq[4];
#pragma unroll
              for (int b = 0; b < 4; b++) {
                const uint32_t u = hash_test_word(hp[b]);
                mq[b] = __ballot(u <= Thi1);
                cm |= mq[b];
              }
              if (cm) {  // wave-uniform, rare
                // Possible candidates are not finished here (a wave would spend ~35 instructions on what is
                // usually ONE lane's k-mer, ~5 % of the kernel at s = 1000): their two hash halves go to this
                // wave's LDS queue -- slots from the wave masks, no atomics -- and are finished, tested exactly
                // and appended a queue-full at a time (drain_queue, at tile ends where room is guaranteed).
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
                } else {  // queue full (early in a genome, T still high): finish and append on the spot
                  uint64_t h[4], m[4];
#pragma unroll
                  for (int b = 0; b < 4; b++) {
                    // (volatile: keeps the finishing arithmetic inside this branch -- left to itself the
                    // compiler computes it speculatively for every k-mer, which is the cost being avoided)
                    HashParts q = hp[b];
                    asm volatile("" : "+v"(q.f1), "+v"(q.f2));
                    h[b] = mm_finish(q);
                    m[b] = __ballot(h[b] < T);
                    am[b] = m[b]; ah[b] = h[b];
                  }
                }
              }
            } else {
              uint64_t h[4], m[4];
#pragma unroll
              for (int b = 0; b < 4; b++) h[b] = kmer_hash(canon[b], P);
              if (allok && T != SENT) {  // one 64-bit compare per k-mer
#pragma unroll
                for (int b = 0; b < 4; b++) m[b] = __ballot(h[b] < T);
              } else {
#pragma unroll
                for (int b = 0; b < 4; b++) {
                  const int rel = rel0 + b;
                  const bool okb = fast ? (run_in + b + 1 >= P.k && rel >= rel_lo && rel < rel_hi) : ok[b];
                  // T == SENT means "sketch not full yet": everything passes (also a hash == SENT)
                  m[b] = __ballot(okb && (h[b] < T || T == SENT));
                }
              }
              if (lo1) {  // workgroup-uniform: later passes of a large sketch
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
      __syncthreads();
      if (uniform32(ctrl->overflow)) {
        // optimistic pass lost candidates: fold what we have, then redo this tile safely
        // (count may exceed cap: clamp happens inside merge_block)
        const MergeResult mr = merge_block(buf, ctrl, cap, s);
        count_at_tile_start = uniform32(mr.count);
        T = uniform64(mr.T);
        safe_mode = true;
        redo = true;
      }
    } while (redo);

    // ---- end of tile: decide about merging and the next tile's mode ----
    const uint32_t cn = uniform32(ctrl->count);
    const uint32_t appended = cn - (count_at_tile_start < cn ? count_at_tile_start : cn);
    // merge early enough that the rank merge's output still fits behind the candidates (2n <= cap)
    const uint32_t half = (uint32_t)cap / 2;
    const bool need_merge = cn > ((half > s + 512 && half < s + room / 2) ? half : s + room / 2);
    safe_mode = appended > room / 4;
    __syncthreads();  // all reads of ctrl->count done before merge or the next tile's appends
    if (need_merge) {
      const MergeResult mr = merge_block(buf, ctrl, cap, s);
      count_at_tile_start = uniform32(mr.count);
      T = uniform64(mr.T);
    }
    else count_at_tile_start = cn;
    // room is guaranteed here (count <= s + room/2, so cap - count >= MIN_ROOM/2 >= NWAVE * QCAP)
    if (qn >= (uint32_t)QDRAIN) drain_queue();
  }

  // ---- final fold and write-out ----
  drain_queue();
  {
    const MergeResult mr = merge_block(buf, ctrl, cap, s);   // frees room should a queue still hold entries
    T = uniform64(mr.T);
  }
  drain_queue();
  uint32_t n = merge_block(buf, ctrl, cap, s).count;
  if (n < s && Tstart != SENT && !sg.partial) {  // workgroup-uniform: the starting threshold was too optimistic for this genome
    Tstart = SENT;
    __syncthreads();
    goto restart;
  }
  uint64_t* o = (sg.partial ? parts : out) + sg.out_off;
  for (uint32_t i = t; i < n; i += WG) o[i] = buf[i];
  if (t == 0) {
    if (ctrl->saw_max && n < s) { o[n] = SENT; n++; }
    // direct segments accumulate over passes; partial slots hold this pass's count only
    if (sg.partial) pcnt[sg.cnt_slot] = n;
    else cnt[sg.cnt_slot] = pass_no > 0 ? sg.expect + n : n;
  }
}

}  // namespace

extern "C" int rtc_sketch_minhash_dev(rtc_ctx* ctx, const uint8_t* d_seq, const uint64_t* h_off,
                                      uint32_t n, int k, uint32_t seed, const uint32_t* h_sizes,
                                      uint32_t size, uint64_t* d_out, uint32_t stride,
                                      uint32_t* d_cnt) {
  if (!ctx || !h_off || (n && (!d_seq || !d_out || !d_cnt))) return RTC_ERR_ARG;
  if (k < 1 || k > 32) return rtc_fail(ctx, RTC_ERR_ARG, "k=%d outside 1..32", k);
  if (n == 0) return RTC_OK;
  if (((uintptr_t)d_seq & 15) != 0) return rtc_fail(ctx, RTC_ERR_ARG, "d_seq must be 16-byte aligned");
  // compile-time k for 16..32: the values the reference's tune_parameters lands on for Mbp..Gbp
  // genomes (recommended k = ceil(log4(maxSize * 9999)) = 17..23, accepted up to +3), its default 21
  // and the customary 31/32; anything else takes the runtime-k kernel
  typedef void (*kern_t)(const uint8_t*, const Segment*, int, uint32_t, int, uint64_t*, uint32_t*, int, uint64_t*, uint32_t*, const uint32_t*);
  auto pick = [&](bool runtime_k, bool packed) -> kern_t {
    kern_t kern = packed ? sketch_minhash_kernel<0, true> : sketch_minhash_kernel<0, false>;
    switch (!runtime_k && seed == MASH_SEED ? k : 0) {
#define RTC_K(K) case K: kern = packed ? sketch_minhash_kernel<K, true> : sketch_minhash_kernel<K, false>; break;
      RTC_K(16) RTC_K(17) RTC_K(18) RTC_K(19) RTC_K(20) RTC_K(21) RTC_K(22) RTC_K(23) RTC_K(24)
      RTC_K(25) RTC_K(26) RTC_K(27) RTC_K(28) RTC_K(29) RTC_K(30) RTC_K(31) RTC_K(32)
#undef RTC_K
      default: break;
    }
    return kern;
  };
  auto prepare = [&](const MinhashPlanInfo& pi) -> int {
    RTC_HIP(ctx, hipFuncSetAttribute((const void*)pick(false, pi.packed_tables), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pi.lds));
    RTC_HIP(ctx, hipFuncSetAttribute((const void*)pick(true, pi.packed_tables), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pi.lds));
    return RTC_OK;
  };
  auto launch = [&](const MinhashLaunch& L) -> int {
    hipLaunchKernelGGL(pick(L.runtime_k, L.packed_tables), dim3(L.nseg), dim3(WG), L.lds, ctx->stream, d_seq, L.d_segs, k, seed, L.cap, d_out,
                       d_cnt, L.pass, L.d_parts, L.d_pcnt, L.d_redo);
    RTC_CHECK_LAUNCH(ctx);
    return RTC_OK;
  };
  return minhash_run(ctx, h_off, n, k, h_sizes, size, d_out, stride, d_cnt, (uint64_t)TILE_BASES, (size_t)MIN_ROOM, prepare, launch);
}

namespace { __global__ void touch_unit_kernel() {} }
int rtc_touch_sketch_minhash(rtc_ctx* ctx) {
  hipLaunchKernelGGL(touch_unit_kernel, dim3(1), dim3(64), 0, ctx->stream);
  RTC_CHECK_LAUNCH(ctx);
  return RTC_OK;
}

// rtc_host.cpp -- host side of the drop-in (see rtc_host.h).  Plain C++17 + zlib; no GPU code.
#include "rtc_host.h"

#include <ctype.h>
#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <unistd.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <atomic>
#include <zlib.h>
#if defined(__x86_64__)
#include <immintrin.h>  // the AVX2 / AVX-512 tiers of the FASTA intake; other hosts build the portable tier only
#define RTC_HOST_X86 1
#else
#define RTC_HOST_X86 0
#endif

#include <algorithm>
#include <type_traits>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <queue>

namespace rtc {

// =================================================================================================
// FASTA / FASTQ reader.  Behaviour follows klib kseq as the reference drives it
// (src/kseq.h:176-226 kseq_read, :92-140 ks_getuntil2):
//   * records start at a line whose first character is '>' or '@';
//   * name = header text up to the first isspace() character; comment = rest of the header line
//     (absent when the name is terminated by the newline);
//   * sequence = following lines concatenated, up to a line starting with '>', '@' or '+';
//     a trailing '\r' is dropped after each appended line when the accumulated length is > 1;
//   * '+' starts a FASTQ quality block: rest of that line skipped, then quality lines until the
//     quality is as long as the sequence.
// =================================================================================================
namespace {

// Byte source: zlib for gzip files; plain files (no 1f 8b magic) are read with read(2) directly,
// which yields the same bytes zlib's transparent mode would.
// gzip files of ordinary size are inflated in ONE call per member by libdeflate when the host has it (resolved with dlopen,
// no build dependency; about twice zlib's rate on sequence text), from a copy of the whole file in memory; the bytes are
// what gzread yields.  Anything it does not take -- no library, a file of more than 64 MiB, damaged or trailing data --
// goes through zlib's streaming gzread as before (which also defines what a damaged file yields).  RTC_NO_LIBDEFLATE=1: zlib only.
struct Deflate {
  void* (*alloc)() = nullptr;
  int (*gunzip_ex)(void*, const void*, size_t, void*, size_t, size_t*, size_t*) = nullptr;
  void (*release)(void*) = nullptr;
  static const Deflate& get() {
    static const Deflate d = []() {
      Deflate r;
      if (getenv("RTC_NO_LIBDEFLATE")) return r;
      void* h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
      if (!h) h = dlopen("libdeflate.so", RTLD_NOW | RTLD_LOCAL);
      if (!h) return r;
      r.alloc = (void* (*)())dlsym(h, "libdeflate_alloc_decompressor");
      r.gunzip_ex = (int (*)(void*, const void*, size_t, void*, size_t, size_t*, size_t*))dlsym(h, "libdeflate_gzip_decompress_ex");
      r.release = (void (*)(void*))dlsym(h, "libdeflate_free_decompressor");
      if (!r.alloc || !r.gunzip_ex || !r.release) r = Deflate();
      return r;
    }();
    return d;
  }
};

class GzStream {
 public:
  explicit GzStream(const std::string& path) {
    fd_ = open(path.c_str(), O_RDONLY);
    if (fd_ < 0) return;
    unsigned char magic[2] = {0, 0};
    const ssize_t got = pread(fd_, magic, 2, 0);
    if (got == 2 && magic[0] == 0x1f && magic[1] == 0x8b) {
      if (!inflate_whole()) {
        f_ = gzdopen(fd_, "r");
        if (!f_) { close(fd_); fd_ = -1; return; }
        gzbuffer(f_, 1 << 20);
      }
    }
    buf_ = (char*)malloc(BUF + SLACK);
  }
  ~GzStream() {
    if (f_) gzclose(f_);  // closes fd_ too
    else if (fd_ >= 0) close(fd_);
    free(buf_);
    free(mem_);
  }
  GzStream(const GzStream&) = delete;
  GzStream& operator=(const GzStream&) = delete;
  bool ok() const { return fd_ >= 0 && buf_; }
  int getc() {
    if (begin_ >= end_) { if (!fill()) return -1; }
    return (unsigned char)buf_[begin_++];
  }
  // appends to `s` up to (not including) the delimiter class; returns the delimiter char or -1 at EOF.
  // mode 0: isspace()  mode 2: '\n'.  *gotany reports whether the stream had any data left.
  template <typename Sink>
  int get_until(int mode, Sink& s, bool append, bool* gotany) {
    if (!append) s.clear();
    *gotany = false;
    int dret = -1;
    for (;;) {
      if (begin_ >= end_) { if (!fill()) break; }
      int i = begin_;
      if (mode == 2) { const void* p = memchr(buf_ + begin_, '\n', end_ - begin_); i = p ? (int)((const char*)p - buf_) : end_; }
      else { for (; i < end_; ++i) if (isspace((unsigned char)buf_[i])) break; }
      *gotany = true;
      s.append(buf_ + begin_, i - begin_);
      begin_ = i + 1;
      if (i < end_) { dret = (unsigned char)buf_[i]; break; }
    }
    if (mode == 2 && s.size() > 1 && s.back() == '\r') s.pop_back();
    return dret;
  }
  bool eof() const { return eof_ && begin_ >= end_; }
  // Sequence-body fast path (the packed staging of the command lines, PackedSink::take_lines): hands the sink the buffer
  // from a line start on; the sink consumes the blocks that are plain sequence text and says whether it stopped inside a
  // line.  The general code continues from there -- finishing that line first -- and sees exactly the stream kseq's loop
  // would see at this point (src/SketchInfo.cpp:880-948 drives that loop).
  template <typename Sink>
  bool body_lines(Sink& s) {
    bool mid = false;
    if (begin_ < end_) begin_ += (int)s.take_lines(buf_ + begin_, (size_t)(end_ - begin_), &mid);
    return mid;
  }
  static constexpr int SLACK = 128;  // bytes a sink may read beyond the data it is handed (inside the allocation)
  // what decompression cost, summed over the parser threads (rtc_host_inflate_stats: the command lines' metrics)
  static std::atomic<uint64_t>& inflate_ns() { static std::atomic<uint64_t> v{0}; return v; }
  static std::atomic<uint64_t>& inflate_bytes() { static std::atomic<uint64_t> v{0}; return v; }
  static uint64_t now_ns() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec; }

 private:
  static constexpr int BUF = 1 << 18;
  // the whole gzip file inflated into mem_ (every member in turn); false: leave it to zlib
  bool inflate_whole() {
    const Deflate& lib = Deflate::get();
    if (!lib.alloc) return false;
    struct stat st;
    if (fstat(fd_, &st) != 0 || st.st_size < 18 || st.st_size > ((off_t)64 << 20)) return false;  // (each parser thread holds one file inflated)
    const size_t csize = (size_t)st.st_size;
    unsigned char* comp = (unsigned char*)malloc(csize);
    if (!comp) return false;
    size_t have = 0;
    while (have < csize) {
      const ssize_t r = pread(fd_, comp + have, csize - have, (off_t)have);
      if (r < 0 && errno == EINTR) continue;
      if (r <= 0) break;
      have += (size_t)r;
    }
    bool ok = have == csize;
    void* d = ok ? lib.alloc() : nullptr;
    ok = ok && d;
    // One inflated file per parser thread: bounded at 1 GiB and at 64 x the compressed size (sequence text deflates
    // 3-4 x; a file that wants more -- crafted, or mostly one letter -- goes through zlib's streaming gzread, which
    // needs 1 MiB).  The trailing ISIZE (the last member's length mod 2^32) is only a first guess inside that bound.
    const size_t cap_max = std::min<size_t>((size_t)1 << 30, csize * 64 + 65536);
    uint32_t isize;
    memcpy(&isize, comp + csize - 4, 4);
    size_t cap = std::min(cap_max, std::max<size_t>((size_t)isize + 64, csize * 3)), out = 0, in = 0;
    char* mem = ok ? (char*)malloc(cap) : nullptr;
    ok = ok && mem;
    while (ok && in < csize) {
      if (csize - in < 18 || comp[in] != 0x1f || comp[in + 1] != 0x8b) { ok = false; break; }  // trailing bytes that are no member
      size_t used = 0, made = 0;
      const uint64_t t_in = now_ns();
      const int rc = lib.gunzip_ex(d, comp + in, csize - in, mem + out, cap - out, &used, &made);
      inflate_ns() += now_ns() - t_in;
      if (rc == 0) inflate_bytes() += made;
      if (rc == 3) {  // LIBDEFLATE_INSUFFICIENT_SPACE
        if (cap >= cap_max) { ok = false; break; }
        const size_t grown = std::min(cap_max, cap * 2);
        char* bigger = (char*)realloc(mem, grown);
        if (!bigger) { ok = false; break; }
        mem = bigger; cap = grown;
        continue;
      }
      if (rc != 0 || used == 0) { ok = false; break; }
      in += used; out += made;
    }
    if (d) lib.release(d);
    free(comp);
    if (!ok) { free(mem); return false; }
    mem_ = mem; mem_size_ = out; mem_pos_ = 0;
    return true;
  }
  bool fill() {
    if (eof_) return false;
    begin_ = 0;
    if (mem_) {
      const size_t n = std::min<size_t>((size_t)BUF, mem_size_ - mem_pos_);
      memcpy(buf_, mem_ + mem_pos_, n);
      mem_pos_ += n;
      end_ = (int)n;
    } else if (f_) {
      const uint64_t t_in = now_ns();  // (f_ is only ever a gzip stream)
      end_ = gzread(f_, buf_, BUF);
      if (end_ > 0) { inflate_ns() += now_ns() - t_in; inflate_bytes() += (uint64_t)end_; }
    }
    else {
      ssize_t r;
      do { r = read(fd_, buf_, BUF); } while (r < 0 && errno == EINTR);
      end_ = (int)r;
    }
    if (end_ <= 0) { end_ = 0; eof_ = true; return false; }
    return true;
  }
  int fd_ = -1;
  gzFile f_ = nullptr;
  char* mem_ = nullptr;            // a gzip file inflated whole (libdeflate), served through buf_ like the other sources
  size_t mem_size_ = 0, mem_pos_ = 0;
  char* buf_ = nullptr;
  int begin_ = 0, end_ = 0;
  bool eof_ = false;
};

// Sequence sink over a caller-supplied flat buffer.  size()/back()/pop_back() are relative to the
// current record, as kseq's seq.l is.  Writes beyond `cap` are dropped but still counted, so the
// caller learns the capacity a retry needs.
struct FlatSink {
  char* base; size_t cap;
  size_t pos = 0, rec = 0;
  void clear() { rec = pos; }
  void append(const char* p, size_t n) {
    if (pos + n <= cap) memcpy(base + pos, p, n);
    else if (pos < cap) memcpy(base + pos, p, cap - pos);
    pos += n;
  }
  void push_back(char c) { if (pos < cap) base[pos] = c; pos++; }
  size_t size() const { return pos - rec; }
  char back() const { return pos - 1 < cap ? base[pos - 1] : 0; }
  void pop_back() { pos--; }
};

// ---- 2-bit packed sink (the command lines' staging format: a quarter of the bytes over PCIe) ----
// Base i of the slot sits at bits 2 (i & 3) of byte i >> 2 with A, C, G, T (either case) = 0, 1, 2, 3.  Every other
// character -- N, IUPAC codes, the '\n' the reader puts between records -- is written as 0 and listed in `runs` as
// (start, length) in slot coordinates (ascending, adjacent ones merged): rtc_unpack_bases_dev turns the packed
// stream back into the ASCII the sketch kernels read and writes 'N' over the runs, which is all the kernels need
// (any character outside ACGT ends a k-mer the same way).  size()/back()/pop_back() behave as FlatSink's.
static inline uint32_t base_code(unsigned char c) { return ((c >> 1) ^ (c >> 2)) & 3u; }
static inline bool base_valid(unsigned char c) { return ((c & 0xC0u) == 0x40u) && ((0x0010008Au >> (c & 31u)) & 1u); }

struct PackedSink {
  uint8_t* base; size_t cap;           // cap in BASES
  std::vector<uint64_t>* runs;
  size_t pos = 0, rec = 0;             // bases appended so far / where the current record began
  // Lines are copied into a small character buffer first and packed a few KiB at a time: the packing loop then
  // always works on whole 32-base groups, whatever the line length of the file (60, 70, 80 ...).
  static constexpr size_t CH = 8192;
  size_t done = 0, fill = 0;           // stream positions [0, done) are packed, [done, done + fill) wait in buf
  unsigned char buf[CH + 32];  // (+ 32: headroom for the move of the unpacked remainder)
  PackedSink(uint8_t* b, size_t c, std::vector<uint64_t>* r) : base(b), cap(c), runs(r) {}
  void clear() { rec = pos; }
  size_t size() const { return pos - rec; }
  char back() const { return fill ? (char)buf[fill - 1] : 0; }
  void put1(unsigned char c) {         // packs one character at stream position `done`
    if (done < cap) {
      const uint32_t sh = 2 * (done & 3);
      const uint8_t code = base_valid(c) ? (uint8_t)(base_code(c) << sh) : 0;
      if (sh == 0) base[done >> 2] = code; else base[done >> 2] |= code;
      if (!base_valid(c)) {
        if (!runs->empty() && (*runs)[runs->size() - 2] + (*runs)[runs->size() - 1] == done) (*runs)[runs->size() - 1]++;
        else { runs->push_back(done); runs->push_back(1); }
      }
    }
    done++;
  }
  void emit(const unsigned char* p, size_t n);   // packs n characters at stream position `done`
  void flush(bool all) {
    const size_t m = all ? fill : fill & ~(size_t)31;
    emit(buf, m);
    const size_t rest = fill - m;  // < 32 unless everything went out
    for (size_t q = 0; q < rest; q++) buf[q] = buf[m + q];
    fill = rest;
  }
  void append(const char* p, size_t n) {
    while (n) {
      if (fill == CH) flush(false);
      const size_t k = n < CH - fill ? n : CH - fill;
      memcpy(buf + fill, p, k);
      fill += k; pos += k; p += k; n -= k;
    }
  }
  void push_back(char c) { append(&c, 1); }
#if RTC_HOST_X86
  // GzStream::body_lines: p is at a line start of a sequence body with `avail` valid bytes (and GzStream::SLACK readable
  // ones behind them).  Takes 32-byte blocks for as long as they hold nothing but sequence characters and '\n' -- no
  // '>', '@', '+' (a record start when first on a line) and no '\r' -- dropping the '\n's on the way: a block without one is
  // stored as it is, otherwise the pieces between the '\n's are stored one behind the other (each as a 32-byte store that the
  // next one overwrites from its own start; the buffer has the headroom).  No load depends on where the previous line
  // ended.  Returns the bytes consumed; *midline says whether they end inside a line, which the caller then finishes
  // with the general code (as it does everything from the first block that is not plain).
  __attribute__((target("avx2"))) size_t take_lines_avx2(const char* p, size_t avail, bool* midline) {
    const char* const p0 = p;
    const char* const e = p + avail;
    const __m256i nl = _mm256_set1_epi8('\n'), c1 = _mm256_set1_epi8('>'), c2 = _mm256_set1_epi8('@'), c3 = _mm256_set1_epi8('+'),
                  c4 = _mm256_set1_epi8('\r');
    bool mid = false;
    while (e - p >= 32) {
      const __m256i v = _mm256_loadu_si256((const __m256i*)p);
      const __m256i sp = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(v, c1), _mm256_cmpeq_epi8(v, c2)),
                                         _mm256_or_si256(_mm256_cmpeq_epi8(v, c3), _mm256_cmpeq_epi8(v, c4)));
      if (_mm256_movemask_epi8(sp)) break;
      uint32_t m = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, nl));
      if (fill + 64 > CH) flush(false);  // leaves fill < 32
      unsigned char* d = buf + fill;
      if (!m) {
        _mm256_storeu_si256((__m256i*)d, v);
        fill += 32; pos += 32; mid = true;
      } else {
        uint32_t off = 0;
        do {
          const uint32_t k = (uint32_t)__builtin_ctz(m);
          _mm256_storeu_si256((__m256i*)d, _mm256_loadu_si256((const __m256i*)(p + off)));
          d += k - off; off = k + 1; m &= m - 1;
        } while (m);
        _mm256_storeu_si256((__m256i*)d, _mm256_loadu_si256((const __m256i*)(p + off)));
        d += 32 - off;
        const size_t got = (size_t)(d - (buf + fill));
        fill += got; pos += got; mid = off < 32;
      }
      p += 32;
    }
    *midline = mid;
    return (size_t)(p - p0);
  }
  // the same with 64-byte blocks: the four comparisons write mask registers, and ONE vpcompressb drops the '\n's
  __attribute__((target("avx512f,avx512bw,avx512vbmi2"))) size_t take_lines_avx512(const char* p, size_t avail, bool* midline) {
    const char* const p0 = p;
    const char* const e = p + avail;
    const __m512i nl = _mm512_set1_epi8('\n'), c1 = _mm512_set1_epi8('>'), c2 = _mm512_set1_epi8('@'), c3 = _mm512_set1_epi8('+'),
                  c4 = _mm512_set1_epi8('\r');
    bool mid = false;
    while (e - p >= 64) {
      const __m512i v = _mm512_loadu_si512((const void*)p);
      if (_mm512_cmpeq_epi8_mask(v, c1) | _mm512_cmpeq_epi8_mask(v, c2) | _mm512_cmpeq_epi8_mask(v, c3) | _mm512_cmpeq_epi8_mask(v, c4)) break;
      const __mmask64 m = _mm512_cmpeq_epi8_mask(v, nl);
      if (fill + 64 > CH) flush(false);  // leaves fill < 32; the buffer has room for one whole block behind CH - 32
      _mm512_storeu_si512((void*)(buf + fill), _mm512_maskz_compress_epi8(~m, v));
      const size_t got = 64 - (size_t)__builtin_popcountll(m);
      fill += got; pos += got;
      mid = !(m >> 63);
      p += 64;
    }
    *midline = mid;
    return (size_t)(p - p0);
  }
#endif
  size_t take_lines(const char* p, size_t avail, bool* midline);
  void truncate(size_t p) {  // drop everything from base p on (pop_back of a '\r', a record cut short)
    pos = p;
    if (p >= done) { fill = p - done; return; }
    fill = 0; done = p;
    if (p < cap && (p & 3)) base[p >> 2] &= (uint8_t)((1u << (2 * (p & 3))) - 1u);
    while (!runs->empty()) {
      uint64_t& st = (*runs)[runs->size() - 2]; uint64_t& ln = (*runs)[runs->size() - 1];
      if (st >= p) { runs->pop_back(); runs->pop_back(); }
      else { if (st + ln > p) ln = p - st; break; }
    }
  }
  void pop_back() { truncate(pos - 1); }
  void finish() { flush(true); }
};

#if RTC_HOST_X86
// 32 bases at a time: codes by two shifts and a mask, validity by re-encoding the codes (pshufb) and comparing with
// the upper-cased input, packing by two multiply-adds (c0 + 4 c1, then + 16 * (c2 + 4 c3)) and a byte gather
__attribute__((target("avx2"))) static size_t pack_run_avx2(const unsigned char* p, size_t n, uint8_t* out) {
  const __m256i m3 = _mm256_set1_epi8(3), mdf = _mm256_set1_epi8((char)0xDF);
  const __m256i lut = _mm256_setr_epi8('A', 'C', 'G', 'T', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 'A', 'C', 'G', 'T', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0);
  const __m256i w1 = _mm256_set1_epi16(0x0401), w2 = _mm256_set1_epi32(0x00100001);
  const __m256i gather = _mm256_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
  size_t i = 0;
  for (; i + 32 <= n; i += 32) {
    const __m256i x = _mm256_loadu_si256((const __m256i*)(p + i));
    const __m256i c = _mm256_and_si256(_mm256_xor_si256(_mm256_srli_epi16(x, 1), _mm256_srli_epi16(x, 2)), m3);
    const __m256i ok = _mm256_cmpeq_epi8(_mm256_shuffle_epi8(lut, c), _mm256_and_si256(x, mdf));
    if ((uint32_t)_mm256_movemask_epi8(ok) != 0xffffffffu) break;  // a character outside ACGTacgt: the caller goes base by base
    const __m256i q = _mm256_shuffle_epi8(_mm256_madd_epi16(_mm256_maddubs_epi16(c, w1), w2), gather);
    const uint32_t lo = (uint32_t)_mm256_cvtsi256_si32(q), hi = (uint32_t)_mm256_extract_epi32(q, 4);
    memcpy(out + (i >> 2), &lo, 4);
    memcpy(out + (i >> 2) + 4, &hi, 4);
  }
  return i;
}

// the same 64 bases at a time: validity as one 64-bit mask, the 16 packed bytes out of the 16 dwords by one down-convert
__attribute__((target("avx512f,avx512bw"))) static size_t pack_run_avx512(const unsigned char* p, size_t n, uint8_t* out) {
  const __m512i m3 = _mm512_set1_epi8(3), mdf = _mm512_set1_epi8((char)0xDF);
  const __m512i lut = _mm512_broadcast_i32x4(_mm_setr_epi8('A', 'C', 'G', 'T', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0));
  const __m512i w1 = _mm512_set1_epi16(0x0401), w2 = _mm512_set1_epi32(0x00100001);
  size_t i = 0;
  for (; i + 64 <= n; i += 64) {
    const __m512i x = _mm512_loadu_si512((const void*)(p + i));
    const __m512i c = _mm512_and_si512(_mm512_xor_si512(_mm512_srli_epi16(x, 1), _mm512_srli_epi16(x, 2)), m3);
    if (_mm512_cmpeq_epi8_mask(_mm512_shuffle_epi8(lut, c), _mm512_and_si512(x, mdf)) != ~(__mmask64)0) break;
    const __m512i q = _mm512_madd_epi16(_mm512_maddubs_epi16(c, w1), w2);
    _mm_storeu_si128((__m128i*)(out + (i >> 2)), _mm512_cvtepi32_epi8(q));
  }
  return i;
}

#endif

// portable form of the same, 8 bases at a time in a 64-bit word
static size_t pack_run_swar(const unsigned char* p, size_t n, uint8_t* out) {
  size_t i = 0;
  for (; i + 8 <= n; i += 8) {
    uint64_t x;
    memcpy(&x, p + i, 8);
    const uint64_t c = ((x >> 1) ^ (x >> 2)) & 0x0303030303030303ULL;
    const uint64_t b0 = c & 0x0101010101010101ULL, b1 = (c >> 1) & 0x0101010101010101ULL;
    const uint64_t e = 0x4141414141414141ULL + 2 * b0 + 6 * b1 + 0x0B * (b0 & b1);   // A, C, G, T from the codes
    if ((x & 0xDFDFDFDFDFDFDFDFULL) != e) break;
    uint64_t t = (c | (c >> 6)) & 0x000F000F000F000FULL;
    t = (t | (t >> 12)) & 0x000000FF000000FFULL;
    t = t | (t >> 24);
    out[i >> 2] = (uint8_t)t; out[(i >> 2) + 1] = (uint8_t)(t >> 8);
  }
  return i;
}

// SIMD tier of the packer and of the line intake: 0 = the best the CPU has (AVX-512 BW + VBMI2, else AVX2, else portable),
// 1 = portable only, 2 = at most AVX2 (tests run every tier)
static int g_pack_portable = 0;
static int simd_tier() {  // 3: AVX-512, 2: AVX2, 1: portable
#if !RTC_HOST_X86
  return 1;
#else
  static const bool a512 = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vbmi2");
  static const bool a2 = __builtin_cpu_supports("avx2");
  if (g_pack_portable == 1) return 1;
  if (a512 && g_pack_portable != 2) return 3;
  return a2 ? 2 : 1;
#endif
}
size_t PackedSink::take_lines(const char* p, size_t avail, bool* midline) {
  *midline = false;
  const int tier = simd_tier();
  size_t k = 0;
#if RTC_HOST_X86
  if (tier == 3) k = take_lines_avx512(p, avail, midline);
  if (tier >= 2 && avail - k >= 32) {
    // (what the 64-byte blocks left: a block with a special character in it, or fewer than 64 bytes) -- the 32-byte blocks get
    // as far as they can; they start where the wider ones stopped, inside a line or not
    bool mid2 = false;
    const size_t k2 = take_lines_avx2(p + k, avail - k, &mid2);
    if (k2) *midline = mid2;
    k += k2;
  }
#else
  (void)tier; (void)p; (void)avail;
#endif
  return k;
}

void PackedSink::emit(const unsigned char* p, size_t n) {
  const int tier = simd_tier();
  size_t i = 0;
  while (i < n) {
    if ((done & 3) == 0 && done + (n - i) <= cap && n - i >= 8) {  // byte-aligned, room for all: whole groups at once
      size_t k = 0;
#if RTC_HOST_X86
      if (tier == 3) k = pack_run_avx512(p + i, n - i, base + (done >> 2));   // 64 at a time, then the rest of the run below
      if (tier >= 2) k += pack_run_avx2(p + i + k, n - i - k, base + ((done + k) >> 2));
      else
#endif
        k = pack_run_swar(p + i, n - i, base + (done >> 2));
      done += k; i += k;
      if (i >= n) break;
    }
    // up to the next group boundary (or past the character the fast loop stopped at) one by one
    size_t k = 0;
    do { put1(p[i++]); k++; } while (i < n && ((done & 3) != 0 || k < 8));
  }
}

// returns sequence length, -1 at EOF, -2 on truncated quality
template <typename Sink>
int next_record_t(GzStream& ks, int& last_char, std::string& name, std::string& comment, bool& has_comment, Sink& seq) {
  int c;
  if (last_char == 0) {
    while ((c = ks.getc()) != -1 && c != '>' && c != '@') {}
    if (c == -1) return -1;
    last_char = c;
  }
  comment.clear(); seq.clear(); has_comment = false;
  bool got;
  int d = ks.get_until(0, name, false, &got);
  if (!got && ks.eof()) return -1;
  if (d != '\n' && d != -1) { ks.get_until(2, comment, false, &got); has_comment = true; }
  for (;;) {
    if constexpr (std::is_same<Sink, PackedSink>::value) {  // plain sequence text inside the buffer, 32 bytes at a time
      if (ks.body_lines(seq)) { ks.get_until(2, seq, true, &got); continue; }  // it stopped inside a line: the rest of that line
    }
    if ((c = ks.getc()) == -1 || c == '>' || c == '+' || c == '@') break;
    if (c == '\n') continue;
    seq.push_back((char)c);
    ks.get_until(2, seq, true, &got);
  }
  if (c == '>' || c == '@') last_char = c;
  if (c != '+') return (int)seq.size();
  while ((c = ks.getc()) != -1 && c != '\n') {}
  if (c == -1) return -2;
  std::string qual;
  while (true) {
    ks.get_until(2, qual, true, &got);
    if ((!got && ks.eof()) || qual.size() >= seq.size()) break;
  }
  last_char = 0;
  if (qual.size() != seq.size()) return -2;
  return (int)seq.size();
}

int next_record(GzStream& ks, int& last_char, FastaRecord& r) {
  return next_record_t(ks, last_char, r.name, r.comment, r.has_comment, r.seq);
}

}  // namespace
void rtc_host_inflate_stats(double* seconds, uint64_t* bytes_out) {
  if (seconds) *seconds = (double)GzStream::inflate_ns().load() * 1e-9;
  if (bytes_out) *bytes_out = GzStream::inflate_bytes().load();
}

bool read_fasta(const std::string& path, std::vector<FastaRecord>& out) {
  GzStream ks(path);
  if (!ks.ok()) return false;
  int last_char = 0;
  FastaRecord r;
  while (next_record(ks, last_char, r) >= 0) out.push_back(r);
  return true;
}

bool read_genome_file(const std::string& path, std::string& bases, SequenceInfo& first, uint64_t& total_len,
                      uint64_t& n_records) {
  GzStream ks(path);
  if (!ks.ok()) return false;
  int last_char = 0;
  FastaRecord r;
  total_len = 0; n_records = 0;
  int len;
  while ((len = next_record(ks, last_char, r)) >= 0) {
    total_len += (uint64_t)len;                                       // src/SketchInfo.cpp:933
    if (n_records == 0) {                                             // :934-947, only the first record is kept
      first.name = r.name;
      first.comment = r.has_comment ? r.comment : std::string("noName");
      first.strand = 0;
      first.length = len;
    }
    bases.append(r.seq);
    bases.push_back('\n');  // record separator: any non-ACGT byte resets the k-mer window
    n_records++;
  }
  return true;
}

uint64_t genome_slot_bytes(const std::string& path) {
  struct stat st;
  if (stat(path.c_str(), &st) != 0) return 0;
  uint64_t sz = (uint64_t)st.st_size;
  FILE* fp = fopen(path.c_str(), "rb");
  if (!fp) return 0;
  unsigned char magic[2] = {0, 0};
  const bool gz = fread(magic, 1, 2, fp) == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
  if (gz && sz >= 18) {
    uint32_t isize = 0;
    fseek(fp, -4, SEEK_END);
    if (fread(&isize, 4, 1, fp) != 1) isize = 0;
    sz = std::max<uint64_t>(isize, sz);  // single-member guess; the reader reports the exact need on overflow
  }
  fclose(fp);
  return sz + 1;
}

int read_genome_file_flat(const std::string& path, char* dst, uint64_t cap, uint64_t& used, SequenceInfo& first,
                          uint64_t& total_len, uint64_t& n_records) {
  GzStream ks(path);
  if (!ks.ok()) return 1;
  int last_char = 0;
  std::string name, comment;
  bool has_comment = false;
  FlatSink sink{dst, (size_t)cap};
  total_len = 0; n_records = 0;
  int len;
  while ((len = next_record_t(ks, last_char, name, comment, has_comment, sink)) >= 0) {
    total_len += (uint64_t)len;
    if (n_records == 0) {
      first.name = name;
      first.comment = has_comment ? comment : std::string("noName");
      first.strand = 0;
      first.length = len;
    }
    sink.push_back('\n');
    n_records++;
  }
  // a record that ends on a truncated quality block (-2) stops the reader like kseq_read() < 0 does;
  // bases appended for it are not part of any returned record
  if (len == -2) sink.pos = sink.rec;
  used = sink.pos;
  return sink.pos > cap ? 2 : 0;
}

int read_genome_file_packed(const std::string& path, uint8_t* dst, uint64_t cap_bases, uint64_t& used, std::vector<uint64_t>& runs,
                            SequenceInfo& first, uint64_t& total_len, uint64_t& n_records) {
  GzStream ks(path);
  if (!ks.ok()) return 1;
  int last_char = 0;
  std::string name, comment;
  bool has_comment = false;
  runs.clear();
  PackedSink sink(dst, (size_t)cap_bases, &runs);
  total_len = 0; n_records = 0;
  int len;
  while ((len = next_record_t(ks, last_char, name, comment, has_comment, sink)) >= 0) {
    total_len += (uint64_t)len;
    if (n_records == 0) {
      first.name = name;
      first.comment = has_comment ? comment : std::string("noName");
      first.strand = 0;
      first.length = len;
    }
    sink.push_back('\n');
    n_records++;
  }
  if (len == -2) { const size_t want = sink.pos; sink.truncate(sink.rec); if (want > cap_bases) { used = want; return 2; } }
  sink.finish();
  used = sink.pos;
  return sink.pos > cap_bases ? 2 : 0;
}

void pack_force_portable(int on) { g_pack_portable = on; }

size_t pack_bases(const char* seq, size_t n, uint8_t* dst, std::vector<uint64_t>& runs) {
  runs.clear();
  PackedSink sink(dst, n, &runs);
  sink.append(seq, n);
  sink.finish();
  return sink.pos;
}

// =================================================================================================
// calSize (list mode), src/SketchInfo.cpp:438-482,536-552
// =================================================================================================
bool cal_size(const std::string& list_file, uint64_t minLen, uint64_t& maxSize, uint64_t& minSize, uint64_t& averageSize) {
  maxSize = 0; minSize = 1u << 31; averageSize = 0;
  uint64_t totalSize = 0; int number = 0, badNumber = 0;
  std::ifstream ifs(list_file);
  if (!ifs) { std::cerr << "ERROR: calSize(), cannot open the inputFile: " << list_file << std::endl; return false; }
  std::string line;
  while (getline(ifs, line)) {
    if (line.size() < 2) continue;
    uint64_t curSize;
    if (line.substr(line.length() - 2) == "gz") {
      FILE* fp = fopen(line.c_str(), "r");
      if (!fp) { std::cerr << "cannot open the genome file: " << line << std::endl; return false; }
      fseek(fp, -4, SEEK_END);
      int nUnCompress = 0;
      if (fread(&nUnCompress, sizeof(int), 1, fp) != 1) nUnCompress = 0;
      curSize = (uint64_t)(int64_t)nUnCompress;
      fclose(fp);
    } else {
      struct stat statbuf;
      if (stat(line.c_str(), &statbuf) != 0) { std::cerr << "cannot open the genome file: " << line << std::endl; return false; }
      curSize = (uint64_t)statbuf.st_size;
    }
    if (curSize < minLen) { badNumber++; continue; }
    maxSize = std::max(maxSize, curSize); minSize = std::min(minSize, curSize);
    totalSize += curSize; number++;
  }
  if (number == 0) { std::cerr << "ERROR: calSize(), no genome passes the minimum length filter" << std::endl; return false; }
  averageSize = totalSize / number;
  int totalNumber = number + badNumber;
  std::cerr << "\t===the genome number for clustering is: " << number << std::endl;
  std::cerr << "\t===the genome number below the minimum genome length threshold is: " << badNumber << std::endl;
  std::cerr << "\t===the total genome number is: " << totalNumber << std::endl;
  if ((double)badNumber / totalNumber >= 0.2)
    fprintf(stderr, "Warning: there are %d poor quality (length < %ld) genome assemblies in the total %d genome assemblied.\n",
            badNumber, (long)minLen, totalNumber);
  std::cerr << "\t===the totalSize is: " << totalSize << std::endl;
  std::cerr << "\t===the maxSize is: " << maxSize << std::endl;
  std::cerr << "\t===the minSize is: " << minSize << std::endl;
  std::cerr << "\t===the averageSize is: " << averageSize << std::endl;
  return true;
}

int file_length_for_containment(const std::string& path) {
  FILE* fp = fopen(path.c_str(), "r");
  if (!fp) return 0;
  int fileLength = 0;
  if (path.size() >= 2 && path.substr(path.length() - 2) == "gz") {
    fseek(fp, -4, SEEK_END);
    int nUnCompress = 0;
    if (fread(&nUnCompress, sizeof(int), 1, fp) != 1) nUnCompress = 0;
    fileLength = nUnCompress;
  } else {
    fseek(fp, 0, SEEK_END);
    fileLength = (int)ftell(fp);
  }
  fclose(fp);
  return fileLength;
}

// src/sub_command.cpp:2383-2467
bool tune_parameters(bool greedy, bool isSetKmer, uint64_t maxSize, uint64_t minSize, uint64_t averageSize,
                     bool& isContainment, bool isJaccard, int& kmerSize, double threshold, int& containCompress,
                     int sketchSize) {
  if (isContainment && isJaccard) {
    std::cerr << "ERROR: tune_parameters(), conflict distance measurement of Mash distance (fixed-sketch-size) and AAF distance (variable-sketch-size) " << std::endl;
    return false;
  }
  if (greedy) {
    if (!isContainment && !isJaccard) { containCompress = (int)(averageSize / 1000); isContainment = true; }
    else if (!isContainment && isJaccard) { }
    else if (averageSize / containCompress < 10) {
      std::cerr << "the containCompress " << containCompress << " is too large and the sketch size is too small" << std::endl;
      containCompress = (int)(averageSize / 1000);
      std::cerr << "set the containCompress to: " << containCompress << std::endl;
    }
  }
  double warning_rate = 0.01, recommend_rate = 0.0001;
  int recommendedKmerSize = (int)ceil(log(maxSize * (1 - recommend_rate) / recommend_rate) / log(4));
  int warningKmerSize = (int)ceil(log(maxSize * (1 - warning_rate) / warning_rate) / log(4));
  if (!isSetKmer) kmerSize = recommendedKmerSize;
  else if (kmerSize < warningKmerSize) {
    std::cerr << "the kmerSize " << kmerSize << " is too small for the maximum genome size of " << maxSize << std::endl;
    std::cerr << "replace the kmerSize to the: " << recommendedKmerSize << " for reducing the random collision of kmers" << std::endl;
    kmerSize = recommendedKmerSize;
  } else if (kmerSize > recommendedKmerSize + 3) {
    std::cerr << "the kmerSize " << kmerSize << " maybe too large for the maximum genome size of " << maxSize << std::endl;
    std::cerr << "replace the kmerSize to the " << recommendedKmerSize << " for increasing the sensitivity of genome comparison" << std::endl;
    kmerSize = recommendedKmerSize;
  }
  double minJaccard;
  if (!isContainment) minJaccard = 1.0 / sketchSize;
  else minJaccard = 1.0 / (minSize / containCompress);
  double maxDist = minJaccard >= 1.0 ? 1.0 : -1.0 / kmerSize * log(2 * minJaccard / (1.0 + minJaccard));
  std::cerr << "-----the max recommand distance threshold is: " << maxDist << std::endl;
  if (threshold > maxDist) {
    std::cerr << "ERROR: tune_parameters(), the threshold: " << threshold << " is out of the valid distance range estimated by Mash distance or AAF distance" << std::endl;
    std::cerr << "Please set a distance threshold with -d option" << std::endl;
    return false;
  }
  std::cerr << "-----the kmerSize is: " << kmerSize << std::endl;
  std::cerr << "-----the threshold is: " << threshold << std::endl;
  if (isContainment) std::cerr << "-----use the AAF distance (variable-sketch-size), the sketchSize is in proportion with 1/" << containCompress << std::endl;
  else std::cerr << "-----use the Mash distance (fixed-sketch-size), the sketchSize is: " << sketchSize << std::endl;
  return true;
}

// src/sub_command.cpp:2317-2381
bool tune_kssd_parameters(bool isSetKmer, uint64_t maxSize, uint64_t minSize, uint64_t averageSize, bool isContainment,
                          int& kmerSize, double threshold, int drlevel) {
  int compression = 1 << (4 * drlevel);
  int sketchSize = (int)(averageSize / compression);
  double warning_rate = 0.01, recommend_rate = 0.0001;
  int recommendedKmerSize = (int)ceil(log(maxSize * (1 - recommend_rate) / recommend_rate) / log(4));
  int warningKmerSize = (int)ceil(log(maxSize * (1 - warning_rate) / warning_rate) / log(4));
  if (!isSetKmer) kmerSize = recommendedKmerSize;
  else if (kmerSize < warningKmerSize) {
    std::cerr << "the kmerSize " << kmerSize << " is too small for the maximum genome size of " << maxSize << std::endl;
    std::cerr << "replace the kmerSize to the: " << recommendedKmerSize << " for reducing the random collision of kmers" << std::endl;
    kmerSize = recommendedKmerSize;
  } else if (kmerSize > recommendedKmerSize + 3) {
    std::cerr << "the kmerSize " << kmerSize << " maybe too large for the maximum genome size of " << maxSize << std::endl;
    std::cerr << "replace the kmerSize to the " << recommendedKmerSize << " for increasing the sensitivity of genome comparison" << std::endl;
    kmerSize = recommendedKmerSize;
  }
  double minJaccard;
  if (!isContainment) minJaccard = 1.0 / sketchSize;
  else minJaccard = 1.0 / (minSize / compression);
  double maxDist = minJaccard >= 1.0 ? 1.0 : -1.0 / kmerSize * log(2 * minJaccard / (1.0 + minJaccard));
  std::cerr << "-----the max recommand distance threshold is: " << maxDist << std::endl;
  if (threshold > maxDist) {
    std::cerr << "ERROR: tune_parameters(), the threshold: " << threshold << " is out of the valid distance range estimated by Mash distance or AAF distance" << std::endl;
    std::cerr << "Please set a distance threshold with -d option" << std::endl;
    return false;
  }
  std::cerr << "-----the kmerSize is: " << kmerSize << std::endl;
  std::cerr << "-----the threshold is: " << threshold << std::endl;
  return true;
}

// src/SketchInfo.cpp:60-102: two Fisher-Yates passes driven by glibc srand()/rand().
// The generator is restated here (glibc random_r TYPE_3: 31-word additive feedback r[i] = r[i-3] +
// r[i-31], seeded by the Lehmer recurrence 16807*x mod 2^31-1, first 310 outputs discarded, result
// >> 1) so the table does not depend on the C library in use and the draws can be produced a block
// ahead of the swaps, which lets the swap targets be prefetched.  tests/test_cpu_host.py pins the
// table against the golden one made with the C library's own rand().
namespace {
struct GlibcRand {
  int32_t r[34];
  int f, b;  // front / rear indices into the 31-word state
  explicit GlibcRand(unsigned seed) {
    int32_t st[31];
    st[0] = seed ? (int32_t)seed : 1;
    for (int i = 1; i < 31; i++) {
      const long hi = st[i - 1] / 127773, lo = st[i - 1] % 127773;
      long word = 16807 * lo - 2836 * hi;
      if (word < 0) word += 2147483647;
      st[i] = (int32_t)word;
    }
    for (int i = 0; i < 31; i++) r[i] = st[i];
    f = 3; b = 0;
    for (int i = 0; i < 310; i++) next();
  }
  inline int32_t next() {
    const uint32_t v = (uint32_t)r[f] + (uint32_t)r[b];
    r[f] = (int32_t)v;
    if (++f == 31) f = 0;
    if (++b == 31) b = 0;
    return (int32_t)(v >> 1);
  }
};
}  // namespace

std::vector<int32_t> generate_shuffle_dim(int half_subk) {
  const int dim_size = 1 << (4 * half_subk);
  std::vector<int32_t> arr(dim_size);
  for (int i = 0; i < dim_size; i++) arr[i] = i;
  const unsigned seeds[2] = {23u, 348842630u};
  constexpr int BLK = 64;
  int js[2][BLK];
  for (unsigned seed : seeds) {
    GlibcRand g(seed);
    // draws for block `nb` are made (and their targets prefetched) while block `nb-1` is swapped
    auto draw = [&](int* out, int i_hi, int cnt) {
      for (int q = 0; q < cnt; q++) { out[q] = g.next() % (i_hi - q + 1); __builtin_prefetch(&arr[out[q]], 1); }
    };
    int i = dim_size - 1, cur = 0;
    int cnt = std::min(BLK, i);
    draw(js[cur], i, cnt);
    while (cnt > 0) {
      const int i_next = i - cnt;
      const int cnt_next = std::min(BLK, i_next);
      if (cnt_next > 0) draw(js[cur ^ 1], i_next, cnt_next);
      for (int q = 0; q < cnt; q++) std::swap(arr[i - q], arr[js[cur][q]]);
      i = i_next; cnt = cnt_next; cur ^= 1;
    }
  }
  return arr;
}

// =================================================================================================
// on-disk formats
// =================================================================================================
namespace {
template <typename T> void wr(FILE* fp, const T& v) { fwrite(&v, sizeof(T), 1, fp); }
template <typename T> bool rd(FILE* fp, T& v) { return fread(&v, sizeof(T), 1, fp) == 1; }
FILE* open_or_die(const std::string& path, const char* mode, const char* who) {
  FILE* fp = fopen(path.c_str(), mode);
  if (!fp) { std::cerr << "ERROR: " << who << ", cannot open the file: " << path << std::endl; exit(1); }
  return fp;
}
}  // namespace

// src/Sketch_IO.cpp:36-134: info.sketch / info.mst / kssd.info.sketch / kssd.info.mst
void save_genome_info(const std::vector<GenomeInfo>& g, const std::string& folder, const std::string& type,
                      bool sketchByFile, bool kssd) {
  const std::string path = folder + '/' + (kssd ? "kssd.info." : "info.") + type;
  FILE* fp = open_or_die(path, "w+", "save_genome_info()");
  wr(fp, sketchByFile);
  size_t n = g.size();
  wr(fp, n);
  for (const GenomeInfo& s : g) {
    if (sketchByFile) {
      int a = (int)s.fileName.length(), b = (int)s.seq0.name.length(), c = (int)s.seq0.comment.length();
      wr(fp, a); wr(fp, b); wr(fp, c); wr(fp, s.seq0.strand); wr(fp, s.totalSeqLength);
      fwrite(s.fileName.c_str(), 1, a, fp); fwrite(s.seq0.name.c_str(), 1, b, fp); fwrite(s.seq0.comment.c_str(), 1, c, fp);
    } else {
      int b = (int)s.seq0.name.length(), c = (int)s.seq0.comment.length();
      wr(fp, b); wr(fp, c); wr(fp, s.seq0.strand); wr(fp, s.seq0.length);
      fwrite(s.seq0.name.c_str(), 1, b, fp); fwrite(s.seq0.comment.c_str(), 1, c, fp);
    }
    if (kssd) wr(fp, s.use64);
  }
  fclose(fp);
}

bool load_genome_info(const std::string& folder, const std::string& type, std::vector<GenomeInfo>& g, bool kssd,
                      bool& sketchByFile) {
  const std::string path = folder + '/' + (kssd ? "kssd.info." : "info.") + type;
  FILE* fp = fopen(path.c_str(), "r");
  if (!fp) { std::cerr << "ERROR: load_genome_info(), cannot open file: " << path << std::endl; return false; }
  size_t n = 0;
  if (!rd(fp, sketchByFile) || !rd(fp, n)) { fclose(fp); return false; }
  g.clear(); g.reserve(n);
  auto rdstr = [&](int len, std::string& s) { s.resize(len > 0 ? len : 0); return len <= 0 || fread(&s[0], 1, len, fp) == (size_t)len; };
  for (size_t i = 0; i < n; i++) {
    GenomeInfo s; s.id = (int)i;
    if (sketchByFile) {
      int a, b, c;
      if (!rd(fp, a) || !rd(fp, b) || !rd(fp, c) || !rd(fp, s.seq0.strand) || !rd(fp, s.totalSeqLength)) { fclose(fp); return false; }
      if (!rdstr(a, s.fileName) || !rdstr(b, s.seq0.name) || !rdstr(c, s.seq0.comment)) { fclose(fp); return false; }
    } else {
      int b, c;
      if (!rd(fp, b) || !rd(fp, c) || !rd(fp, s.seq0.strand) || !rd(fp, s.seq0.length)) { fclose(fp); return false; }
      if (!rdstr(b, s.seq0.name) || !rdstr(c, s.seq0.comment)) { fclose(fp); return false; }
    }
    if (kssd && !rd(fp, s.use64)) { fclose(fp); return false; }
    g.push_back(std::move(s));
  }
  fclose(fp);
  return true;
}

// src/Sketch_IO.cpp:169-226 (MinHash branch)
void save_minhash_sketches(const std::vector<GenomeInfo>& g, const MinHashSketchFile& f, const std::string& folder,
                           bool sketchByFile) {
  save_genome_info(g, folder, "sketch", sketchByFile, false);
  FILE* fp = open_or_die(folder + "/hash.sketch", "w+", "saveSketch()");
  int sketch_func_id = 0;
  wr(fp, sketch_func_id); wr(fp, f.kmerSize); wr(fp, f.isContainment);
  if (f.isContainment) wr(fp, f.containCompress); else wr(fp, f.sketchSize);
  for (const auto& h : f.hashes) { size_t m = h.size(); wr(fp, m); fwrite(h.data(), sizeof(uint64_t), m, fp); }
  fclose(fp);
  std::cerr << "-----save the sketches into: " << folder << std::endl;
}

// src/Sketch_IO.cpp:284-353
bool load_minhash_sketches(const std::string& folder, std::vector<GenomeInfo>& g, MinHashSketchFile& f, bool& sketchByFile) {
  FILE* fp = fopen((folder + "/hash.sketch").c_str(), "r");
  if (!fp) {
    std::cerr << "ERROR: loadSketches(), cannot open the file: " << folder << "/hash.sketch" << std::endl;
    FILE* t = fopen((folder + "/kssd.hash.sketch").c_str(), "r");
    if (t) { std::cerr << "Do you want to load the kssd sketches directory? Try again with '--fast' option" << std::endl; fclose(t); }
    return false;
  }
  int sketch_func_id = 0;
  if (!rd(fp, sketch_func_id) || sketch_func_id != 0) { fclose(fp); std::cerr << "ERROR: loadSketches(), only MinHash sketch folders are supported" << std::endl; return false; }
  if (!rd(fp, f.kmerSize) || !rd(fp, f.isContainment)) { fclose(fp); return false; }
  if (f.isContainment) { if (!rd(fp, f.containCompress)) { fclose(fp); return false; } }
  else if (!rd(fp, f.sketchSize)) { fclose(fp); return false; }
  if (!load_genome_info(folder, "sketch", g, false, sketchByFile)) { fclose(fp); return false; }
  f.hashes.assign(g.size(), {});
  for (size_t i = 0; i < g.size(); i++) {
    size_t m = 0;
    if (!rd(fp, m)) { fclose(fp); return false; }
    f.hashes[i].resize(m);
    if (m && fread(f.hashes[i].data(), sizeof(uint64_t), m, fp) != m) { fclose(fp); return false; }
  }
  fclose(fp);
  return true;
}

// src/SketchInfo.h:115-160 "MHIDX001".  Entries are written in ascending hash order (the reference
// writes hash-map iteration order; its loader accepts any order).
void save_minhash_index(const MinHashSketchFile& f, const std::string& folder) {
  std::vector<std::pair<uint64_t, uint32_t>> all;
  size_t tot = 0;
  for (const auto& h : f.hashes) tot += h.size();
  all.reserve(tot);
  for (size_t g = 0; g < f.hashes.size(); g++) for (uint64_t h : f.hashes[g]) all.emplace_back(h, (uint32_t)g);
  std::sort(all.begin(), all.end());
  size_t nuniq = 0;
  for (size_t i = 0; i < all.size(); i++) if (i == 0 || all[i].first != all[i - 1].first) nuniq++;
  FILE* fp = fopen((folder + "/minhash.sketch.index").c_str(), "wb");
  if (!fp) { std::cerr << "WARNING: cannot save MinHash index to: " << folder << "/minhash.sketch.index" << std::endl; return; }
  fwrite("MHIDX001", 1, 8, fp);
  wr(fp, nuniq);
  for (size_t i = 0; i < all.size();) {
    size_t j = i;
    while (j < all.size() && all[j].first == all[i].first) j++;
    uint64_t h = all[i].first; uint32_t m = (uint32_t)(j - i);
    wr(fp, h); wr(fp, m);
    for (size_t t = i; t < j; t++) wr(fp, all[t].second);
    i = j;
  }
  fclose(fp);
  std::cerr << "-----MinHash inverted index saved: " << folder << "/minhash.sketch.index (" << nuniq << " unique hashes)" << std::endl;
}

// src/Sketch_IO.cpp:136-167
void save_kssd_sketches(const std::vector<GenomeInfo>& g, const KssdSketchFile& f, const std::string& folder, bool sketchByFile) {
  save_genome_info(g, folder, "sketch", sketchByFile, true);
  FILE* fp = open_or_die(folder + "/kssd.hash.sketch", "w+", "saveSketch()");
  fwrite(&f.info, sizeof(KssdParameters), 1, fp);
  for (size_t i = 0; i < g.size(); i++) {
    if (f.use64) { size_t m = f.h64[i].size(); wr(fp, m); fwrite(f.h64[i].data(), sizeof(uint64_t), m, fp); }
    else { size_t m = f.h32[i].size(); wr(fp, m); fwrite(f.h32[i].data(), sizeof(uint32_t), m, fp); }
  }
  fclose(fp);
  std::cerr << "-----save the kssd sketches into: " << folder << std::endl;
}

// src/Sketch_IO.cpp:228-282
bool load_kssd_sketches(const std::string& folder, std::vector<GenomeInfo>& g, KssdSketchFile& f, bool& sketchByFile) {
  FILE* fp = fopen((folder + "/kssd.hash.sketch").c_str(), "r");
  if (!fp) {
    std::cerr << "ERROR: loadKssdSketches(), cannot open the file: " << folder << "/kssd.hash.sketch" << std::endl;
    FILE* t = fopen((folder + "/hash.sketch").c_str(), "r");
    if (t) { std::cerr << "Do you want to load the minHash sketches directory? Try again without '--fast' option" << std::endl; fclose(t); }
    return false;
  }
  if (fread(&f.info, sizeof(KssdParameters), 1, fp) != 1) { fclose(fp); return false; }
  if (!load_genome_info(folder, "sketch", g, true, sketchByFile) || g.empty()) { fclose(fp); return false; }
  f.use64 = g[0].use64;
  f.h32.assign(f.use64 ? 0 : g.size(), {}); f.h64.assign(f.use64 ? g.size() : 0, {});
  for (size_t i = 0; i < g.size(); i++) {
    size_t m = 0;
    if (!rd(fp, m)) { fclose(fp); return false; }
    if (f.use64) { f.h64[i].resize(m); if (m && fread(f.h64[i].data(), 8, m, fp) != m) { fclose(fp); return false; } }
    else { f.h32[i].resize(m); if (m && fread(f.h32[i].data(), 4, m, fp) != m) { fclose(fp); return false; } }
  }
  fclose(fp);
  return true;
}

// src/SketchInfo.cpp:1379-1467: kssd.sketch.index = {size_t H; hash[H]; u32 count[H]}, .dict = ids
void save_kssd_index(const KssdSketchFile& f, const std::string& folder) {
  std::vector<std::pair<uint64_t, uint32_t>> all;
  const size_t n = f.use64 ? f.h64.size() : f.h32.size();
  for (size_t g = 0; g < n; g++) {
    if (f.use64) for (uint64_t h : f.h64[g]) all.emplace_back(h, (uint32_t)g);
    else for (uint32_t h : f.h32[g]) all.emplace_back((uint64_t)h, (uint32_t)g);
  }
  std::sort(all.begin(), all.end());
  std::vector<uint64_t> keys; std::vector<uint32_t> counts;
  FILE* fd = open_or_die(folder + "/kssd.sketch.dict", "w+", "transSketchesFromIndex");
  for (size_t i = 0; i < all.size();) {
    size_t j = i;
    while (j < all.size() && all[j].first == all[i].first) j++;
    keys.push_back(all[i].first); counts.push_back((uint32_t)(j - i));
    for (size_t t = i; t < j; t++) wr(fd, all[t].second);
    i = j;
  }
  fclose(fd);
  FILE* fi = open_or_die(folder + "/kssd.sketch.index", "w+", "transSketchesFromIndex");
  size_t H = keys.size();
  wr(fi, H);
  if (f.use64) fwrite(keys.data(), 8, H, fi);
  else for (uint64_t k : keys) { uint32_t k32 = (uint32_t)k; wr(fi, k32); }
  fwrite(counts.data(), 4, H, fi);
  fclose(fi);
}

// src/MST_IO.cpp:200-217, :47-70
void save_mst(const std::vector<rtc_edge>& mst, const std::string& folder) {
  FILE* fp = open_or_die(folder + "/edge.mst", "w+", "saveMST()");
  size_t m = mst.size();
  wr(fp, m);
  for (const rtc_edge& e : mst) { wr(fp, e.preNode); wr(fp, e.sufNode); wr(fp, e.dist); }
  fclose(fp);
  std::cerr << "-----save the mst into: " << folder << std::endl;
}
bool load_mst(const std::string& folder, std::vector<rtc_edge>& mst) {
  FILE* fp = fopen((folder + "/edge.mst").c_str(), "r");
  if (!fp) { std::cerr << "ERROR: loadMST(), cannot open the file: " << folder << "/edge.mst" << std::endl; return false; }
  size_t m = 0;
  if (!rd(fp, m)) { fclose(fp); return false; }
  mst.clear(); mst.reserve(m);
  for (size_t i = 0; i < m; i++) {
    rtc_edge e;
    if (!rd(fp, e.preNode) || !rd(fp, e.sufNode) || !rd(fp, e.dist)) { fclose(fp); return false; }
    mst.push_back(e);
  }
  fclose(fp);
  std::cerr << "-----read the mst file from " << folder << "/edge.mst" << std::endl;
  return true;
}

// =================================================================================================
// forest cut + BFS clusters + text output
// =================================================================================================
void save_dense(const std::string& folder, const std::vector<int32_t>& dense, int span, int genome_number) {  // src/MST_IO.cpp:219-233
  const std::string file = folder + "/mst.dense";
  FILE* fp = fopen(file.c_str(), "w+");
  if (!fp) { std::cerr << "ERROR: saveDense(), cannot open the file: " << file; exit(1); }
  fwrite(&genome_number, sizeof(int), 1, fp);
  fwrite(&span, sizeof(int), 1, fp);
  fwrite(dense.data(), sizeof(int32_t), (size_t)span * genome_number, fp);
  fclose(fp);
  std::cerr << "-----save the dense file into: " << folder << std::endl;
}

bool load_dense(const std::string& folder, std::vector<int32_t>& dense, int& span, int& genome_number) {  // src/MST_IO.cpp:12-28
  const std::string file = folder + "/mst.dense";
  FILE* fp = fopen(file.c_str(), "r");
  if (!fp) { std::cerr << "ERROR: saveDense(), cannot open the file: " << file; return false; }
  bool ok = fread(&genome_number, sizeof(int), 1, fp) == 1 && fread(&span, sizeof(int), 1, fp) == 1 && span >= 0 && genome_number >= 0;
  if (ok) {
    dense.resize((size_t)span * genome_number);
    ok = fread(dense.data(), sizeof(int32_t), dense.size(), fp) == dense.size();
  }
  fclose(fp);
  if (ok) std::cerr << "-----read the dense file from: " << file << std::endl;
  return ok;
}

void save_ani(const std::string& folder, const uint64_t ani[101]) {  // src/MST_IO.cpp:235-250
  const std::string file = folder + "/mst.ani";
  FILE* fp = fopen(file.c_str(), "w+");
  if (!fp) { std::cerr << "ERROR: saveANI(), cannot open file: " << file << std::endl; exit(1); }
  fwrite(ani, sizeof(uint64_t), 101, fp);
  fclose(fp);
  std::cerr << "-----save the ani file into: " << file << std::endl;
}

bool load_ani(const std::string& folder, uint64_t ani[101]) {
  FILE* fp = fopen((folder + "/mst.ani").c_str(), "r");
  if (!fp) return false;
  const bool ok = fread(ani, sizeof(uint64_t), 101, fp) == 101;
  fclose(fp);
  return ok;
}

std::vector<int> noise_nodes(const std::vector<std::vector<int>>& cluster, const std::vector<int32_t>& dense, int span,
                             int genome_number, double threshold) {  // src/sub_command.cpp:3077-3092, src/MST.cpp:189-211
  const int alpha = 2;
  const int denseIndex = (int)(threshold / 0.01);
  std::vector<int> total;
  if (denseIndex < 0 || denseIndex >= span) return total;
  for (const std::vector<int>& cl : cluster) {
    if (cl.size() == 1) continue;
    std::vector<std::pair<int, int>> arr;
    for (int element : cl) arr.emplace_back(element, dense[(size_t)denseIndex * genome_number + element]);
    std::sort(arr.begin(), arr.end(), [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.second < b.second; });  // cmpPair
    const int denseQ1 = arr[arr.size() / 4].second;
    int thr = std::max(std::min(denseQ1 - 1, alpha), 0);
    for (const auto& p : arr) { if (p.second <= thr) total.push_back(p.first); else break; }
  }
  return total;
}

std::vector<rtc_edge> modify_forest(const std::vector<rtc_edge>& forest, const std::vector<int>& noise) {  // src/MST.cpp:86-107
  std::vector<char> is_noise;
  for (int v : noise) { if ((size_t)v >= is_noise.size()) is_noise.resize(v + 1, 0); is_noise[v] = 1; }
  std::vector<rtc_edge> out;
  for (const rtc_edge& e : forest) {
    const bool rm = ((size_t)e.preNode < is_noise.size() && is_noise[e.preNode]) || ((size_t)e.sufNode < is_noise.size() && is_noise[e.sufNode]);
    if (!rm) out.push_back(e);
  }
  return out;
}

namespace {
struct DSU {  // src/MST.cpp:40-57
  std::vector<int> p, r;
  explicit DSU(int n) : p(n), r(n, 0) { for (int i = 0; i < n; i++) p[i] = i; }
  int find(int x) { int root = x; while (p[root] != root) root = p[root]; while (p[x] != root) { int nx = p[x]; p[x] = root; x = nx; } return root; }
  int unite(int a, int b) {
    a = find(a); b = find(b);
    if (a == b) return a;
    if (r[a] < r[b]) std::swap(a, b);
    p[b] = a;
    if (r[a] == r[b]) r[a]++;
    return a;
  }
};
std::vector<rtc_edge> by_distance(const std::vector<rtc_edge>& mst) {
  std::vector<rtc_edge> edges = mst;
  std::sort(edges.begin(), edges.end(), [](const rtc_edge& a, const rtc_edge& b) { return a.dist < b.dist; });
  return edges;
}
}  // namespace

// Newick text of the single-linkage dendrogram the MST defines (what src/MST.cpp:1044-1150 prints): edges in ascending
// weight (std::sort with the reference's comparator, so ties fall as they do there) merge the sets of their two ends; a
// merge is an inner node at height = the edge's weight whose first branch leads to the top node of preNode's set and
// whose second to sufNode's, branch length = height difference (never negative); the text starts at the top node of
// genome 0's set.  Merges live in one flat array (node id = leaves + merge number), the sets in a plain parent array
// with the set's current top node kept at its representative, and the text is written by an explicit stack (a
// caterpillar of 100 000 leaves would overflow a recursive writer).
std::string get_newick_tree(const std::vector<GenomeInfo>& g, const std::vector<rtc_edge>& mst, bool sketchByFile) {
  const int leaves = (int)g.size();
  auto label = [&](int v) -> const std::string& { return sketchByFile ? g[v].fileName : g[v].seq0.name; };
  if (leaves == 0) return ";";
  if (leaves == 1) return label(0) + ";";
  struct Merge { int kid[2]; double len[2]; double at; };
  std::vector<Merge> merges;
  merges.reserve(leaves - 1);
  std::vector<int> set_of(leaves), top(leaves);
  for (int v = 0; v < leaves; v++) set_of[v] = top[v] = v;
  auto rep = [&](int v) { while (set_of[v] != v) { set_of[v] = set_of[set_of[v]]; v = set_of[v]; } return v; };
  auto level = [&](int node) { return node < leaves ? 0.0 : merges[node - leaves].at; };
  for (const rtc_edge& e : by_distance(mst)) {
    const int a = rep(e.preNode), b = rep(e.sufNode);
    if (a == b) continue;
    Merge m;
    m.kid[0] = top[a]; m.kid[1] = top[b];
    m.at = e.dist;
    for (int side = 0; side < 2; side++) m.len[side] = std::max(0.0, m.at - level(m.kid[side]));
    merges.push_back(m);
    set_of[b] = a;
    top[a] = leaves + (int)merges.size() - 1;
  }
  std::string text;
  struct Visit { int node; int done; };  // done: how many of an inner node's two branches are written
  std::vector<Visit> path{{top[rep(0)], 0}};
  while (!path.empty()) {
    Visit& v = path.back();
    if (v.node < leaves) {
      text += label(v.node);
    } else if (v.done < 2) {
      text += v.done == 0 ? "(" : ",";
      const int kid = merges[v.node - leaves].kid[v.done++];
      path.push_back({kid, 0});
      continue;
    } else {
      text += ")";
    }
    path.pop_back();
    if (!path.empty()) {  // the subtree just closed hangs on branch done-1 of the node now on top
      const Visit& up = path.back();
      text += ":" + std::to_string(merges[up.node - leaves].len[up.done - 1]);
    }
  }
  return text + ";";
}

static FILE* open_out(const std::string& output, const char* who) {
  FILE* fp = fopen(output.c_str(), "w");
  if (!fp) { std::cerr << "ERROR: " << who << "(), cannot write file: " << output << std::endl; exit(1); }
  return fp;
}

void print_newick_tree(const std::vector<GenomeInfo>& g, const std::vector<rtc_edge>& mst, bool sketchByFile, const std::string& output) {
  FILE* fp = open_out(output, "print_newick_tree");
  fprintf(fp, "%s\n", get_newick_tree(g, mst, sketchByFile).c_str());
  fclose(fp);
}

void print_phylip_tree(const std::vector<GenomeInfo>& g, const std::vector<rtc_edge>& mst, bool sketchByFile, const std::string& output) {
  FILE* fp = open_out(output, "print_phylip_tree");
  fprintf(fp, "1\n%s\n", get_newick_tree(g, mst, sketchByFile).c_str());
  fclose(fp);
}

void print_nexus_tree(const std::vector<GenomeInfo>& g, const std::vector<rtc_edge>& mst, bool sketchByFile, const std::string& output) {  // src/MST_IO.cpp:296-335
  const std::string tree = get_newick_tree(g, mst, sketchByFile);
  FILE* fp = open_out(output, "print_nexus_tree");
  fprintf(fp, "#NEXUS\nBEGIN TAXA;\n  DIMENSIONS NTAX=%zu;\n  TAXLABELS", g.size());
  for (const GenomeInfo& gi : g) {
    std::string lab = sketchByFile ? gi.fileName : gi.seq0.name;
    size_t pos = 0;
    while ((pos = lab.find('\'', pos)) != std::string::npos) { lab.insert(pos, "'"); pos += 2; }  // quotes are doubled
    fprintf(fp, " '%s'", lab.c_str());
  }
  fprintf(fp, ";\nEND;\nBEGIN TREES;\n  TREE tree_1 = [&R] %s\nEND;\n", tree.c_str());
  fclose(fp);
}

void print_linkage_matrix(int N, const std::vector<rtc_edge>& mst, const std::string& output) {  // src/MST.cpp:1246-1287, src/MST_IO.cpp:362-376
  FILE* fp = open_out(output, "print_linkage_matrix");
  if (N > 1) {
    const std::vector<rtc_edge> edges = by_distance(mst);
    DSU dsu(N);
    int next_id = N;
    std::vector<int> cluster_id(N), cluster_size(2 * N - 1);
    for (int i = 0; i < N; i++) { cluster_id[i] = i; cluster_size[i] = 1; }
    for (const rtc_edge& e : edges) {
      const int ru = dsu.find(e.preNode), rv = dsu.find(e.sufNode);
      if (ru == rv) continue;
      const int id_u = cluster_id[ru], id_v = cluster_id[rv];
      const int new_id = next_id++, new_size = cluster_size[id_u] + cluster_size[id_v];
      fprintf(fp, "%d\t%d\t%.6f\t%d\n", id_u, id_v, e.dist, new_size);
      cluster_id[dsu.unite(ru, rv)] = new_id;
      cluster_size[new_id] = new_size;
    }
  }
  fclose(fp);
}

std::vector<rtc_edge> kruskal_algorithm(const std::vector<rtc_edge>& graph, int vertices) {  // src/MST.cpp:59-75
  std::vector<int> parent(vertices), ranks(vertices, 0);             // UnionFind.h: union by rank, path compression
  for (int v = 0; v < vertices; v++) parent[v] = v;
  auto find = [&](int x) { int r = x; while (parent[r] != r) r = parent[r]; while (parent[x] != r) { int nx = parent[x]; parent[x] = r; x = nx; } return r; };
  std::vector<rtc_edge> tree;
  for (const rtc_edge& e : graph) {
    int a = find(e.preNode), b = find(e.sufNode);
    if (a == b) continue;
    if (ranks[a] > ranks[b]) parent[b] = a;
    else if (ranks[a] < ranks[b]) parent[a] = b;
    else { parent[a] = b; ranks[b]++; }
    tree.push_back(e);
  }
  return tree;
}

std::vector<rtc_edge> generate_forest(const std::vector<rtc_edge>& mst, double threshold) {  // src/MST.cpp:77-85
  std::vector<rtc_edge> forest;
  for (const rtc_edge& e : mst) if (e.dist <= threshold) forest.push_back(e);
  return forest;
}

std::vector<std::vector<int>> generate_cluster_with_bfs(const std::vector<rtc_edge>& forest, int vertices) {  // :109-142
  std::vector<std::vector<int>> res, G(vertices);
  for (const rtc_edge& e : forest) { G[e.preNode].push_back(e.sufNode); G[e.sufNode].push_back(e.preNode); }
  std::vector<char> visited(vertices, 0);
  for (int i = 0; i < vertices; i++) {
    if (visited[i]) continue;
    visited[i] = 1;
    std::queue<int> Q; Q.push(i);
    std::vector<int> cl{i};
    while (!Q.empty()) {
      int k = Q.front(); Q.pop();
      for (int v : G[k]) { if (visited[v]) continue; visited[v] = 1; Q.push(v); cl.push_back(v); }
    }
    res.push_back(std::move(cl));
  }
  return res;
}

// skips n bytes of fp; false when the file ends before them
static bool skip_bytes(FILE* fp, size_t n) {
  const long at = ftell(fp);
  if (at < 0 || fseek(fp, 0, SEEK_END) != 0) return false;
  const long end = ftell(fp);
  if (end < 0 || (size_t)(end - at) < n) return false;
  return fseek(fp, at + (long)n, SEEK_SET) == 0;
}

// the representatives' inverted index as the reference's writers lay it out (hash, list length, positions in
// rep_ids); keys ascending here (the reference walks a hash map)
static size_t write_rep_index(FILE* fp, const KssdClusterState& st) {
  std::vector<std::pair<uint64_t, int>> post;
  const size_t rep_count = st.reps.use64 ? st.reps.h64.size() : st.reps.h32.size();
  for (size_t r = 0; r < rep_count; r++) {
    if (st.reps.use64) for (uint64_t h : st.reps.h64[r]) post.emplace_back(h, (int)r);
    else for (uint32_t h : st.reps.h32[r]) post.emplace_back((uint64_t)h, (int)r);
  }
  std::sort(post.begin(), post.end());
  size_t index_size = 0;
  for (size_t i = 0; i < post.size(); i++) if (i == 0 || post[i].first != post[i - 1].first) index_size++;
  wr(fp, index_size);
  for (size_t i = 0; i < post.size();) {
    size_t j = i;
    while (j < post.size() && post[j].first == post[i].first) j++;
    wr(fp, post[i].first);
    const size_t list_size = j - i;
    wr(fp, list_size);
    for (size_t q = i; q < j; q++) wr(fp, post[q].second);
    i = j;
  }
  return index_size;
}

static void write_state_sketch(FILE* fp, const GenomeInfo& g, const KssdSketchFile& sk, size_t i) {
  const bool use64 = sk.use64;
  const size_t n32 = use64 ? 0 : sk.h32[i].size(), n64 = use64 ? sk.h64[i].size() : 0;
  const uint32_t sketchsize = (uint32_t)(n32 + n64);
  wr(fp, g.id); wr(fp, g.totalSeqLength); wr(fp, use64); wr(fp, sketchsize);
  wr(fp, n32); wr(fp, n64);
  if (n32) fwrite(sk.h32[i].data(), 4, n32, fp);
  if (n64) fwrite(sk.h64[i].data(), 8, n64, fp);
  const size_t name_len = g.fileName.size();
  wr(fp, name_len);
  fwrite(g.fileName.data(), 1, name_len, fp);
}

static bool read_state_sketch(FILE* fp, GenomeInfo& g, KssdSketchFile& sk, bool first) {
  bool use64 = false; uint32_t sketchsize = 0; size_t n32 = 0, n64 = 0, name_len = 0;
  bool ok = rd(fp, g.id) && rd(fp, g.totalSeqLength) && rd(fp, use64) && rd(fp, sketchsize) && rd(fp, n32) && rd(fp, n64) &&
            n32 < ((size_t)1 << 32) && n64 < ((size_t)1 << 32);
  if (!ok) return false;
  if (first) sk.use64 = use64;
  std::vector<uint32_t> h32(n32); std::vector<uint64_t> h64(n64);
  if (n32) ok = ok && fread(h32.data(), 4, n32, fp) == n32;
  if (n64) ok = ok && fread(h64.data(), 8, n64, fp) == n64;
  ok = ok && rd(fp, name_len) && name_len < ((size_t)1 << 20);
  if (!ok) return false;
  g.fileName.resize(name_len);
  if (name_len) ok = fread(&g.fileName[0], 1, name_len, fp) == name_len;
  g.use64 = use64;
  g.seq0.name = "N/A"; g.seq0.comment = "N/A";   // printKssdResult's text for sketches without record infos (src/MST_IO.cpp:99-104)
  if (sk.use64) sk.h64.push_back(std::move(h64)); else sk.h32.push_back(std::move(h32));
  return ok;
}

// src/MST_IO.cpp:72-179 (printResult / printKssdResult share the layout)
// src/greedy.cpp:1545-1625
bool save_kssd_cluster_state(const std::string& path, const KssdClusterState& st) {
  FILE* fp = fopen(path.c_str(), "wb");
  if (!fp) { std::cerr << "ERROR: Cannot open file for writing: " << path << std::endl; return false; }
  wr(fp, st.threshold); wr(fp, st.kmer_size);
  wr(fp, st.info.half_k); wr(fp, st.info.half_subk); wr(fp, st.info.drlevel); wr(fp, st.info.genomeNumber);
  const size_t rep_count = st.rep_ids.size();
  wr(fp, rep_count);
  fwrite(st.rep_ids.data(), sizeof(int), rep_count, fp);
  const size_t sketch_count = st.genomes.size();
  wr(fp, sketch_count);
  for (size_t i = 0; i < sketch_count; i++) write_state_sketch(fp, st.genomes[i], st.sk, i);
  const size_t cluster_count = st.clusters.size();
  wr(fp, cluster_count);
  for (const auto& c : st.clusters) {
    const size_t m = c.size();
    wr(fp, m);
    fwrite(c.data(), sizeof(int), m, fp);
  }
  // representatives' inverted index: marker + 64-bit keys
  const char magic[8] = {'K', 'S', 'S', 'I', '0', '2', '\0', '\0'};
  fwrite(magic, 1, 8, fp);
  const size_t index_size = write_rep_index(fp, st);
  fclose(fp);
  std::cerr << "Saved clustering state to: " << path << std::endl
            << "  - " << sketch_count << " genomes" << std::endl
            << "  - " << rep_count << " clusters (representatives)" << std::endl
            << "  - " << index_size << " unique hashes in inverted index" << std::endl;
  return true;
}

// MinHashClusterState::save (src/greedy.cpp:2134-2208): "MINHASH\0", threshold, k, sketch size, containment flag,
// representative ids, every sketch in clustering order {id, length, hashes, file name}, the clusters (members are
// positions in that order, the representative first), the representatives' inverted index hash -> positions in
// representative_ids.  The reference writes the index in hash-map iteration order; here ascending by hash (readers
// rebuild a map from it either way).
bool save_minhash_cluster_state(const std::string& path, const KssdClusterState& st) {
  FILE* fp = fopen(path.c_str(), "wb");
  if (!fp) { std::cerr << "ERROR: Cannot open file for writing: " << path << std::endl; return false; }
  const char magic[8] = {'M', 'I', 'N', 'H', 'A', 'S', 'H', '\0'};
  fwrite(magic, 1, 8, fp);
  wr(fp, st.threshold); wr(fp, st.kmer_size); wr(fp, st.sketch_size); wr(fp, st.is_containment);
  const size_t rep_count = st.rep_ids.size();
  wr(fp, rep_count);
  fwrite(st.rep_ids.data(), sizeof(int), rep_count, fp);
  const size_t sketch_count = st.genomes.size();
  wr(fp, sketch_count);
  for (size_t i = 0; i < sketch_count; i++) {
    const GenomeInfo& g = st.genomes[i];
    wr(fp, g.id); wr(fp, g.totalSeqLength);
    const size_t hash_count = st.sk.h64[i].size();
    wr(fp, hash_count);
    if (hash_count) fwrite(st.sk.h64[i].data(), 8, hash_count, fp);
    const size_t name_len = g.fileName.size();
    wr(fp, name_len);
    fwrite(g.fileName.data(), 1, name_len, fp);
  }
  const size_t cluster_count = st.clusters.size();
  wr(fp, cluster_count);
  for (const auto& c : st.clusters) {
    const size_t m = c.size();
    wr(fp, m);
    fwrite(c.data(), sizeof(int), m, fp);
  }
  std::cerr << "Saving inverted index: ";
  const size_t index_size = write_rep_index(fp, st);
  std::cerr << index_size << " unique hashes..." << std::endl;
  fclose(fp);
  std::cerr << "Saved clustering state to: " << path << std::endl
            << "  - " << sketch_count << " genomes" << std::endl
            << "  - " << rep_count << " clusters (representatives)" << std::endl
            << "  - " << index_size << " unique hashes in inverted index" << std::endl;
  return true;
}

// MinHashClusterState::load (src/greedy.cpp:2210-2302): parameters, representative ids and clusters are kept; the
// sketches are skipped (the caller reloads them from the folder, src/sub_command.cpp:100-139) and so is the index
// (it is rebuilt from the representatives).
bool load_minhash_cluster_state(const std::string& path, KssdClusterState& st) {
  FILE* fp = fopen(path.c_str(), "rb");
  if (!fp) { std::cerr << "ERROR: Cannot open file for reading: " << path << std::endl; return false; }
  char magic[8] = {0};
  if (fread(magic, 1, 8, fp) != 8 || strncmp(magic, "MINHASH", 8) != 0) {
    std::cerr << "ERROR: Invalid file format (not a MinHash cluster state)" << std::endl;
    fclose(fp);
    return false;
  }
  st = KssdClusterState();
  st.minhash = true; st.sk.use64 = true; st.reps.use64 = true;
  bool ok = rd(fp, st.threshold) && rd(fp, st.kmer_size) && rd(fp, st.sketch_size) && rd(fp, st.is_containment);
  size_t rep_count = 0, sketch_count = 0, cluster_count = 0;
  ok = ok && rd(fp, rep_count) && rep_count < ((size_t)1 << 31);
  if (ok) { st.rep_ids.resize(rep_count); ok = fread(st.rep_ids.data(), sizeof(int), rep_count, fp) == rep_count; }
  ok = ok && rd(fp, sketch_count) && sketch_count < ((size_t)1 << 31);
  for (size_t i = 0; ok && i < sketch_count; i++) {
    int id; uint64_t len; size_t hash_count = 0, name_len = 0;
    ok = rd(fp, id) && rd(fp, len) && rd(fp, hash_count) && hash_count < ((size_t)1 << 32) &&
         fseek(fp, (long)(hash_count * 8), SEEK_CUR) == 0 && rd(fp, name_len) && name_len < ((size_t)1 << 20) &&
         fseek(fp, (long)name_len, SEEK_CUR) == 0;
  }
  ok = ok && rd(fp, cluster_count) && cluster_count < ((size_t)1 << 31);
  for (size_t c = 0; ok && c < cluster_count; c++) {
    size_t m = 0;
    ok = rd(fp, m) && m < ((size_t)1 << 31);
    if (!ok) break;
    std::vector<int> cl(m);
    if (m) ok = fread(cl.data(), sizeof(int), m, fp) == m;
    st.clusters.push_back(std::move(cl));
  }
  size_t index_size = 0;
  ok = ok && rd(fp, index_size);
  fclose(fp);
  if (!ok) { std::cerr << "ERROR: truncated or malformed cluster state: " << path << std::endl; return false; }
  std::cerr << "Loading inverted index: " << index_size << " unique hashes..." << std::endl;
  std::cerr << "Loaded clustering state from: " << path << std::endl
            << "  - " << sketch_count << " genomes" << std::endl
            << "  - " << rep_count << " clusters (representatives)" << std::endl
            << "  - " << index_size << " unique hashes in inverted index" << std::endl;
  return true;
}

// src/greedy.cpp:1627-1734 (the inverted index that follows the clusters is not read: it is a function of the
// representatives' sketches)
bool load_kssd_cluster_state(const std::string& path, KssdClusterState& st) {
  FILE* fp = fopen(path.c_str(), "rb");
  if (!fp) { std::cerr << "ERROR: Cannot open file for reading: " << path << std::endl; return false; }
  bool ok = rd(fp, st.threshold) && rd(fp, st.kmer_size) && rd(fp, st.info.half_k) && rd(fp, st.info.half_subk) &&
            rd(fp, st.info.drlevel) && rd(fp, st.info.genomeNumber);
  size_t rep_count = 0, sketch_count = 0, cluster_count = 0;
  ok = ok && rd(fp, rep_count) && rep_count < ((size_t)1 << 31);
  if (ok) { st.rep_ids.resize(rep_count); ok = fread(st.rep_ids.data(), sizeof(int), rep_count, fp) == rep_count; }
  ok = ok && rd(fp, sketch_count) && sketch_count < ((size_t)1 << 31);
  st.genomes.clear(); st.sk.h32.clear(); st.sk.h64.clear(); st.sk.info = st.info;
  for (size_t i = 0; ok && i < sketch_count; i++) {
    GenomeInfo g;
    ok = read_state_sketch(fp, g, st.sk, i == 0);
    if (ok) st.genomes.push_back(g);
  }
  ok = ok && rd(fp, cluster_count) && cluster_count < ((size_t)1 << 31);
  st.clusters.clear();
  for (size_t c = 0; ok && c < cluster_count; c++) {
    size_t m = 0;
    ok = rd(fp, m) && m < ((size_t)1 << 31);
    if (!ok) break;
    std::vector<int> cl(m);
    if (m) ok = fread(cl.data(), sizeof(int), m, fp) == m;
    st.clusters.push_back(std::move(cl));
  }
  fclose(fp);
  if (!ok) { std::cerr << "ERROR: truncated or malformed cluster state: " << path << std::endl; return false; }
  for (int r : st.rep_ids) if (r < 0 || (size_t)r >= st.genomes.size()) { std::cerr << "ERROR: Representative ID " << r << " out of range" << std::endl; return false; }
  if (st.clusters.size() != rep_count) { std::cerr << "ERROR: " << st.clusters.size() << " clusters for " << rep_count << " representatives" << std::endl; return false; }
  st.reps = KssdSketchFile(); st.reps.info = st.info; st.reps.use64 = st.sk.use64; st.rep_genomes.clear();
  for (int r : st.rep_ids) {   // representatives = all_sketches[representative_ids] (src/greedy.cpp:1676-1685)
    st.rep_genomes.push_back(st.genomes[r]);
    if (st.sk.use64) st.reps.h64.push_back(st.sk.h64[r]); else st.reps.h32.push_back(st.sk.h32[r]);
  }
  std::cerr << "Loaded clustering state from: " << path << std::endl
            << "  - " << sketch_count << " genomes" << std::endl
            << "  - " << rep_count << " clusters (representatives)" << std::endl;
  return true;
}

// src/greedy.cpp:2351-2428
bool save_kssd_repdb(const std::string& path, const KssdClusterState& st) {
  FILE* fp = fopen(path.c_str(), "wb");
  if (!fp) { std::cerr << "ERROR: Cannot open RepDB file for writing: " << path << std::endl; return false; }
  fwrite("REPDB002", 1, 8, fp);
  wr(fp, st.threshold); wr(fp, st.kmer_size);
  wr(fp, st.info.half_k); wr(fp, st.info.half_subk); wr(fp, st.info.drlevel); wr(fp, st.info.genomeNumber);
  const size_t rep_count = st.rep_ids.size();
  wr(fp, rep_count);
  for (size_t r = 0; r < rep_count; r++) {
    wr(fp, st.rep_ids[r]);
    write_state_sketch(fp, st.rep_genomes[r], st.reps, r);
  }
  const size_t cluster_count = st.clusters.size();
  wr(fp, cluster_count);
  for (const auto& c : st.clusters) {
    const size_t m = c.size();
    wr(fp, m);
    fwrite(c.data(), sizeof(int), m, fp);
  }
  const size_t all_count = st.genomes.size();
  wr(fp, all_count);
  for (const GenomeInfo& g : st.genomes) {
    const size_t name_len = g.fileName.size();
    wr(fp, name_len);
    fwrite(g.fileName.data(), 1, name_len, fp);
    wr(fp, g.totalSeqLength);
  }
  const size_t index_size = write_rep_index(fp, st);
  fclose(fp);
  std::cerr << "RepDB saved to: " << path << std::endl
            << "  Representatives: " << rep_count << std::endl
            << "  Total genomes:   " << all_count << std::endl
            << "  Inverted index:  " << index_size << " unique hashes" << std::endl;
  return true;
}

// src/greedy.cpp:2430-2537
bool load_kssd_repdb(const std::string& path, KssdClusterState& st) {
  FILE* fp = fopen(path.c_str(), "rb");
  if (!fp) { std::cerr << "ERROR: Cannot open RepDB file for reading: " << path << std::endl; return false; }
  char magic[8] = {0};
  if (fread(magic, 1, 8, fp) != 8 || (memcmp(magic, "REPDB002", 8) != 0 && memcmp(magic, "REPDB001", 8) != 0)) {
    std::cerr << "ERROR: Invalid RepDB file (bad magic): " << path << std::endl;
    fclose(fp);
    return false;
  }
  const bool v2 = memcmp(magic, "REPDB002", 8) == 0;
  bool ok = rd(fp, st.threshold) && rd(fp, st.kmer_size) && rd(fp, st.info.half_k) && rd(fp, st.info.half_subk) &&
            rd(fp, st.info.drlevel) && rd(fp, st.info.genomeNumber);
  size_t rep_count = 0, cluster_count = 0, all_count = 0, index_size = 0;
  ok = ok && rd(fp, rep_count) && rep_count < ((size_t)1 << 31);
  st.rep_ids.clear(); st.rep_genomes.clear(); st.genomes.clear(); st.clusters.clear();
  st.sk = KssdSketchFile(); st.reps = KssdSketchFile(); st.reps.info = st.info; st.sk.info = st.info;
  for (size_t r = 0; ok && r < rep_count; r++) {
    int rid = 0; GenomeInfo g;
    ok = rd(fp, rid) && read_state_sketch(fp, g, st.reps, r == 0);
    if (ok) { st.rep_ids.push_back(rid); st.rep_genomes.push_back(g); }
  }
  st.sk.use64 = st.reps.use64;
  ok = ok && rd(fp, cluster_count) && cluster_count < ((size_t)1 << 31);
  for (size_t c = 0; ok && c < cluster_count; c++) {
    size_t m = 0;
    ok = rd(fp, m) && m < ((size_t)1 << 31);
    if (!ok) break;
    std::vector<int> cl(m);
    if (m) ok = fread(cl.data(), sizeof(int), m, fp) == m;
    st.clusters.push_back(std::move(cl));
  }
  ok = ok && rd(fp, all_count) && all_count < ((size_t)1 << 31);
  for (size_t i = 0; ok && i < all_count; i++) {
    GenomeInfo g; size_t name_len = 0;
    ok = rd(fp, name_len) && name_len < ((size_t)1 << 20);
    if (!ok) break;
    g.fileName.resize(name_len);
    if (name_len) ok = fread(&g.fileName[0], 1, name_len, fp) == name_len;
    ok = ok && rd(fp, g.totalSeqLength);
    g.id = 0; g.use64 = st.reps.use64;   // load_repdb fills only the name and the length of all_sketches
    g.seq0.name = "N/A"; g.seq0.comment = "N/A";
    st.genomes.push_back(g);
  }
  ok = ok && rd(fp, index_size);
  for (size_t i = 0; ok && i < index_size; i++) {   // walked for the truncation check only
    uint64_t h64 = 0; uint32_t h32 = 0; size_t ls = 0;
    ok = (v2 ? rd(fp, h64) : rd(fp, h32)) && rd(fp, ls) && ls < ((size_t)1 << 31) && skip_bytes(fp, ls * sizeof(int));
  }
  fclose(fp);
  if (!ok) { std::cerr << "ERROR: truncated or malformed RepDB: " << path << std::endl; return false; }
  if (st.clusters.size() != rep_count) { std::cerr << "ERROR: " << st.clusters.size() << " clusters for " << rep_count << " representatives" << std::endl; return false; }
  std::cerr << "RepDB loaded from: " << path << std::endl
            << "  Representatives: " << rep_count << std::endl
            << "  Total genomes:   " << all_count << std::endl
            << "  Inverted index:  " << index_size << " unique hashes" << std::endl
            << "  Threshold:       " << st.threshold << std::endl
            << "  Kmer size:       " << st.kmer_size << std::endl;
  return true;
}

// src/greedy.cpp:2789-2862
bool save_minhash_repdb(const std::string& path, const KssdClusterState& st) {
  FILE* fp = fopen(path.c_str(), "wb");
  if (!fp) { std::cerr << "ERROR: Cannot open RepDB file for writing: " << path << std::endl; return false; }
  fwrite("MHREPDB1", 1, 8, fp);
  wr(fp, st.threshold); wr(fp, st.kmer_size); wr(fp, st.sketch_size); wr(fp, st.is_containment);
  const size_t rep_count = st.rep_ids.size();
  wr(fp, rep_count);
  for (size_t r = 0; r < rep_count; r++) {
    const GenomeInfo& g = st.rep_genomes[r];
    wr(fp, st.rep_ids[r]); wr(fp, g.id); wr(fp, g.totalSeqLength); wr(fp, st.is_containment);
    const size_t hash_count = st.reps.h64[r].size();
    wr(fp, hash_count);
    if (hash_count) fwrite(st.reps.h64[r].data(), 8, hash_count, fp);
    const size_t name_len = g.fileName.size();
    wr(fp, name_len);
    fwrite(g.fileName.data(), 1, name_len, fp);
  }
  const size_t cluster_count = st.clusters.size();
  wr(fp, cluster_count);
  for (const auto& c : st.clusters) {
    const size_t m = c.size();
    wr(fp, m);
    fwrite(c.data(), sizeof(int), m, fp);
  }
  const size_t all_count = st.genomes.size();
  wr(fp, all_count);
  for (const GenomeInfo& g : st.genomes) {
    const size_t name_len = g.fileName.size();
    wr(fp, name_len);
    fwrite(g.fileName.data(), 1, name_len, fp);
    wr(fp, g.totalSeqLength);
  }
  const size_t index_size = write_rep_index(fp, st);
  fclose(fp);
  std::cerr << "MinHash RepDB saved to: " << path << std::endl
            << "  Representatives: " << rep_count << std::endl
            << "  Total genomes:   " << all_count << std::endl
            << "  Inverted index:  " << index_size << " unique hashes" << std::endl;
  return true;
}

// src/greedy.cpp:2864-2956
bool load_minhash_repdb(const std::string& path, KssdClusterState& st) {
  FILE* fp = fopen(path.c_str(), "rb");
  if (!fp) { std::cerr << "ERROR: Cannot open RepDB file for reading: " << path << std::endl; return false; }
  char magic[8] = {0};
  if (fread(magic, 1, 8, fp) != 8 || memcmp(magic, "MHREPDB1", 8) != 0) {
    std::cerr << "ERROR: Invalid MinHash RepDB file (bad magic): " << path << std::endl;
    fclose(fp);
    return false;
  }
  st = KssdClusterState();
  st.minhash = true; st.reps.use64 = true; st.sk.use64 = true;
  bool ok = rd(fp, st.threshold) && rd(fp, st.kmer_size) && rd(fp, st.sketch_size) && rd(fp, st.is_containment);
  size_t rep_count = 0, cluster_count = 0, all_count = 0, index_size = 0;
  ok = ok && rd(fp, rep_count) && rep_count < ((size_t)1 << 31);
  for (size_t r = 0; ok && r < rep_count; r++) {
    int rid = 0; GenomeInfo g; bool cont = false; size_t hash_count = 0, name_len = 0;
    ok = rd(fp, rid) && rd(fp, g.id) && rd(fp, g.totalSeqLength) && rd(fp, cont) && rd(fp, hash_count) && hash_count < ((size_t)1 << 32);
    if (!ok) break;
    std::vector<uint64_t> h(hash_count);
    if (hash_count) ok = fread(h.data(), 8, hash_count, fp) == hash_count;
    ok = ok && rd(fp, name_len) && name_len < ((size_t)1 << 20);
    if (!ok) break;
    g.fileName.resize(name_len);
    if (name_len) ok = fread(&g.fileName[0], 1, name_len, fp) == name_len;
    g.seq0.name = "N/A"; g.seq0.comment = "";
    st.rep_ids.push_back(rid); st.rep_genomes.push_back(g); st.reps.h64.push_back(std::move(h));
  }
  ok = ok && rd(fp, cluster_count) && cluster_count < ((size_t)1 << 31);
  for (size_t c = 0; ok && c < cluster_count; c++) {
    size_t m = 0;
    ok = rd(fp, m) && m < ((size_t)1 << 31);
    if (!ok) break;
    std::vector<int> cl(m);
    if (m) ok = fread(cl.data(), sizeof(int), m, fp) == m;
    st.clusters.push_back(std::move(cl));
  }
  ok = ok && rd(fp, all_count) && all_count < ((size_t)1 << 31);
  for (size_t i = 0; ok && i < all_count; i++) {
    GenomeInfo g; size_t name_len = 0;
    ok = rd(fp, name_len) && name_len < ((size_t)1 << 20);
    if (!ok) break;
    g.fileName.resize(name_len);
    if (name_len) ok = fread(&g.fileName[0], 1, name_len, fp) == name_len;
    ok = ok && rd(fp, g.totalSeqLength);
    g.seq0.name = "N/A"; g.seq0.comment = "";   // printRepDBClusterResult's text for genomes without record infos (src/sub_command.cpp:688-693)
    st.genomes.push_back(g);
  }
  ok = ok && rd(fp, index_size);
  for (size_t i = 0; ok && i < index_size; i++) {
    uint64_t h = 0; size_t ls = 0;
    ok = rd(fp, h) && rd(fp, ls) && ls < ((size_t)1 << 31) && skip_bytes(fp, ls * sizeof(int));
  }
  fclose(fp);
  if (!ok) { std::cerr << "ERROR: truncated or malformed RepDB: " << path << std::endl; return false; }
  if (st.clusters.size() != rep_count) { std::cerr << "ERROR: " << st.clusters.size() << " clusters for " << rep_count << " representatives" << std::endl; return false; }
  std::cerr << "MinHash RepDB loaded from: " << path << std::endl
            << "  Representatives: " << rep_count << std::endl
            << "  Total genomes:   " << all_count << std::endl
            << "  Inverted index:  " << index_size << " unique hashes" << std::endl
            << "  Threshold:       " << st.threshold << std::endl
            << "  Kmer size:       " << st.kmer_size << std::endl
            << "  Sketch size:     " << st.sketch_size << std::endl
            << "  Containment:     " << (st.is_containment ? "yes" : "no") << std::endl;
  return true;
}

// src/greedy.cpp:2656-2765
void print_kssd_repdb_stats(const KssdClusterState& st, std::ostream& out) {
  size_t total_genomes = 0;
  for (const auto& cl : st.clusters) total_genomes += cl.size();
  const size_t nrep = st.rep_ids.size();
  auto rep_size = [&](size_t r) { return st.reps.use64 ? st.reps.h64[r].size() : st.reps.h32[r].size(); };
  out << "========================================" << std::endl;
  out << (st.minhash ? "    MinHash RepDB Statistics Report" : "        RepDB Statistics Report") << std::endl;
  out << "========================================" << std::endl << std::endl;
  out << "[Basic Info]" << std::endl;
  out << "  Threshold:              " << st.threshold << std::endl;
  out << "  Kmer size:              " << st.kmer_size << std::endl;
  if (st.minhash) {
    out << "  Sketch size:            " << st.sketch_size << std::endl;
    out << "  Containment mode:       " << (st.is_containment ? "yes" : "no") << std::endl << std::endl;
  } else {
    out << "  KSSD half_k:            " << st.info.half_k << std::endl;
    out << "  KSSD half_subk:         " << st.info.half_subk << std::endl;
    out << "  KSSD drlevel:           " << st.info.drlevel << std::endl << std::endl;
  }
  out << "[Scale]" << std::endl;
  out << "  Total genomes:          " << total_genomes << std::endl;
  out << "  Representatives:        " << nrep << std::endl;
  out << "  Clusters:               " << st.clusters.size() << std::endl;
  const double compression = total_genomes > 0 ? (1.0 - (double)nrep / total_genomes) * 100.0 : 0.0;
  out << "  Compression ratio:      " << std::fixed << std::setprecision(2) << compression << "%" << std::endl << std::endl;
  // the inverted index, from the representatives' sketches
  std::vector<uint64_t> keys;
  for (size_t r = 0; r < nrep; r++) {
    if (st.reps.use64) keys.insert(keys.end(), st.reps.h64[r].begin(), st.reps.h64[r].end());
    else keys.insert(keys.end(), st.reps.h32[r].begin(), st.reps.h32[r].end());
  }
  std::sort(keys.begin(), keys.end());
  size_t unique = 0, max_posting = 0;
  for (size_t i = 0; i < keys.size();) {
    size_t j = i;
    while (j < keys.size() && keys[j] == keys[i]) j++;
    unique++; max_posting = std::max(max_posting, j - i);
    i = j;
  }
  out << "[Inverted Index]" << std::endl;
  out << "  Unique hashes:          " << unique << std::endl;
  out << "  Total postings:         " << keys.size() << std::endl;
  out << "  Avg posting length:     " << std::fixed << std::setprecision(2) << (unique ? (double)keys.size() / unique : 0.0) << std::endl;
  out << "  Max posting length:     " << max_posting << std::endl << std::endl;
  out << "[Cluster Size Distribution]" << std::endl;
  if (!st.clusters.empty()) {
    std::vector<int> sizes;
    size_t singleton = 0;
    for (const auto& cl : st.clusters) { sizes.push_back((int)cl.size()); if (cl.size() <= 1) singleton++; }
    std::sort(sizes.begin(), sizes.end());
    out << "  Min cluster size:       " << sizes.front() << std::endl;
    out << "  Max cluster size:       " << sizes.back() << std::endl;
    out << "  Mean cluster size:      " << std::fixed << std::setprecision(2) << (double)total_genomes / st.clusters.size() << std::endl;
    out << "  Median cluster size:    " << sizes[sizes.size() / 2] << std::endl;
    out << "  Singletons:             " << singleton << " (" << std::fixed << std::setprecision(1)
        << (100.0 * singleton / st.clusters.size()) << "%)" << std::endl;
    out << "  P90 cluster size:       " << sizes[(size_t)(sizes.size() * 0.9)] << std::endl;
    out << "  P95 cluster size:       " << sizes[(size_t)(sizes.size() * 0.95)] << std::endl;
    out << "  P99 cluster size:       " << sizes[(size_t)(sizes.size() * 0.99)] << std::endl;
  }
  if (!st.minhash) out << std::endl << "[Representative Sketch Sizes]" << std::endl;   // (the MinHash report has no such section)
  if (nrep && !st.minhash) {
    size_t min_sk = SIZE_MAX, max_sk = 0, sum_sk = 0;
    for (size_t r = 0; r < nrep; r++) { const size_t z = rep_size(r); min_sk = std::min(min_sk, z); max_sk = std::max(max_sk, z); sum_sk += z; }
    out << "  Min sketch size:        " << min_sk << std::endl;
    out << "  Max sketch size:        " << max_sk << std::endl;
    out << "  Mean sketch size:       " << std::fixed << std::setprecision(1) << (double)sum_sk / nrep << std::endl;
  }
  uint64_t total_seq_len = 0, rep_seq_len = 0;
  for (const GenomeInfo& g : st.genomes) total_seq_len += g.totalSeqLength;
  if (total_seq_len > 0) {
    for (const GenomeInfo& g : st.rep_genomes) rep_seq_len += g.totalSeqLength;
    out << std::endl << "[Genome Coverage]" << std::endl;
    out << "  Total sequence length:  " << total_seq_len << " bp" << std::endl;
    out << "  Representative seq len: " << rep_seq_len << " bp" << std::endl;
    out << "  Coverage ratio:         " << std::fixed << std::setprecision(2) << (100.0 * rep_seq_len / total_seq_len) << "%" << std::endl;
  }
  out << "========================================" << std::endl;
}

void print_result(const std::vector<std::vector<int>>& cluster, const std::vector<GenomeInfo>& g, bool sketchByFile,
                  const std::string& outputFile, double threshold) {
  FILE* fp = fopen(outputFile.c_str(), "w");
  if (!fp) { std::cerr << "Error in printResult(), cannot open file: " << outputFile << std::endl; exit(1); }
  if (threshold >= 0.0) {
    fprintf(fp, "# Clustering threshold: %.6f\n", threshold);
    fprintf(fp, "# Total clusters: %zu\n", cluster.size());
    fprintf(fp, "#\n");
  }
  for (size_t i = 0; i < cluster.size(); i++) {
    fprintf(fp, "the cluster %d is: \n", (int)i);
    for (size_t j = 0; j < cluster[i].size(); j++) {
      const int curId = cluster[i][j];
      if (curId < 0 || curId >= (int)g.size()) continue;
      const GenomeInfo& s = g[curId];
      if (sketchByFile)
        fprintf(fp, "\t%5d\t%6d\t%12dnt\t%20s\t%20s\t%s\n", (int)j, curId, (int)s.totalSeqLength, s.fileName.c_str(),
                s.seq0.name.c_str(), s.seq0.comment.c_str());
      else
        fprintf(fp, "\t%6d\t%6d\t%12dnt\t%20s\t%s\n", (int)j, curId, s.seq0.length, s.seq0.name.c_str(), s.seq0.comment.c_str());
    }
    fprintf(fp, "\n");
  }
  fclose(fp);
}

std::string current_date_time() {
  time_t now = time(0);
  struct tm tstruct = *localtime(&now);
  char buf[80];
  strftime(buf, sizeof(buf), "%Y_%m_%d_%H-%M-%S", &tstruct);
  return buf;
}

}  // namespace rtc

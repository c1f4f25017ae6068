// main.cpp -- clust-mst / clust-greedy command lines on top of the MI355X C ABI.
//
// Mirrors the flag surface and workflow dispatch of the reference's src/main.cpp:113-254,291-671
// for the sketch + all-pairs + cluster path: list-mode input (-l) or one FASTA file whose
// records are the genomes (no -l), MinHash
// and KSSD (--fast) sketching, --presketched / --premsted resume, -e/--no-save, and the same
// intermediate folder (info.sketch, hash.sketch, minhash.sketch.index, kssd.*, info.mst,
// edge.mst).  Built twice: -DGREEDY_CLUST gives clust-greedy, otherwise clust-mst
// (CMakeLists.txt:40-58 of the reference does the same).
// GPUs: every visible MI355X is used (--gpus LIST / RTC_GPUS to choose): one context + one host thread
// per GPU, file batches go round-robin to the GPUs, the sketches stay in HBM (copied to the host only
// to write hash.sketch), are shared among the GPUs with RCCL broadcasts, and the MST runs
// row-sharded with one all-reduce per Boruvka round (rtc_mst_sharded).  Also here: --append (clust-mst, and
// clust-greedy --fast with or without a stored state), --dense, the tree / linkage writers, clust-greedy's
// --save-rep cluster state (KSSD and MinHash) and representative database (--db ..., KSSD and MinHash), clust-greedy
// --append on MinHash sketches, and the dense estimator loops (--inverted-index=false: modifyMST / greedyCluster).
// clust-mst's --db / --save-rep, --auto-threshold and its companions and single-FASTA input to --append / --db are
// outside this path and exit with a message.
#include <math.h>
#include <iomanip>
#include <limits>
#include <sys/stat.h>
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <fstream>
#include <iostream>
#include <numeric>
#include <functional>
#include <mutex>
#include <thread>

#include "rtc_host.h"

using namespace std;
using namespace rtc;

static double get_sec() { struct timeval tv; gettimeofday(&tv, NULL); return (double)tv.tv_sec + (double)tv.tv_usec / 1000000; }

// RTC_METRICS_JSON=<file>: the phase times the reference prints under -DTimer (same labels) and the sizes of the run,
// as one JSON object written when everything else has been written ("metrics / logging").
struct Metrics {
  vector<pair<string, string>> kv;  // in order of arrival; a key set twice keeps the last value
  void put(const string& k, const string& raw) {
    for (auto& e : kv) if (e.first == k) { e.second = raw; return; }
    kv.emplace_back(k, raw);
  }
  void num(const string& k, double v) { char b[64]; snprintf(b, sizeof b, "%.9g", v); put(k, b); }
  void str(const string& k, const string& v) {
    string q = "\"";
    for (char c : v) { if (c == '"' || c == '\\') q += '\\'; if ((unsigned char)c >= 0x20) q += c; }
    put(k, q + "\"");
  }
  void write() const {
    const char* path = getenv("RTC_METRICS_JSON");
    if (!path || !*path) return;
    FILE* f = fopen(path, "w");
    if (!f) { fprintf(stderr, "Warning: cannot write %s\n", path); return; }
    fputs("{", f);
    for (size_t i = 0; i < kv.size(); i++) fprintf(f, "%s\n  \"%s\": %s", i ? "," : "", kv[i].first.c_str(), kv[i].second.c_str());
    fputs("\n}\n", f);
    fclose(f);
  }
};
static Metrics g_metrics;

// The warm-up helper thread (rtc_warmup beside the sketch phase) is joined before the process leaves through exit():
// static destructors and the HIP runtime's teardown must not run while it is still inside a HIP call.
static std::thread* g_gpu_thread = nullptr;  // main's GPU bring-up thread while it runs
#ifdef RTC_MEASURE
// Measurement build only (`make measure` -> bin/clust-mst-measure, tools/cli_timeline.py; the shipped binaries hold none of
// this).  RTC_EXIT_PROBE=1: what the process leaves to the kernel at _exit -- the pageable staging ring and the lanes' device
// staging -- is released by hand, timed, before leaving: says which of them the time between _exit and the parent's wait()
// belongs to.  RTC_EXIT_DELAY_MS / RTC_START_DELAY_MS: a sleep before leaving / before anything else.  RTC_NO_THP=1: staging
// memory without transparent huge pages.
extern "C" int rtc_debug_device_reset(int device);
static std::vector<char*> g_exit_probe_host;
static std::vector<std::pair<rtc_ctx*, void*>> g_exit_probe_dev;
#endif
static std::thread g_warmup_thread;
static std::mutex g_warmup_mutex;
static void join_warmup() {
  std::lock_guard<std::mutex> lk(g_warmup_mutex);
  if (g_warmup_thread.joinable() && g_warmup_thread.get_id() != std::this_thread::get_id()) g_warmup_thread.join();
}
#define CHECK(ctx, call)                                                                         \
  do {                                                                                           \
    int st__ = (call);                                                                           \
    if (st__ != RTC_OK) {                                                                        \
      fprintf(stderr, "ERROR: %s failed (%d): %s\n", #call, st__, rtc_last_error(ctx));          \
      join_warmup();                                                                             \
      exit(1);                                                                                   \
    }                                                                                            \
  } while (0)

struct DeviceSketches {  // sketches resident in HBM in the CSR the pair kernels read
  void* d_hashes = nullptr; uint64_t* d_start = nullptr; uint32_t* d_len = nullptr;
  uint32_t n = 0; int width = 8;
};

struct Gpu {  // one per GPU in use: context, communicator, staging and the resident sketch rows
  rtc_ctx* ctx = nullptr; rtc_comm* comm = nullptr;
  int device = 0;
  void* d_sk = nullptr;         // resident sketches: row g (genome id g) at d_sk + g*stride*width
  uint32_t* d_cnt = nullptr;    // hashes per genome
};

// Sketches produced by sketch_files and left in HBM (every GPU holds all rows after the share step)
struct Resident {
  bool ok = false;              // false: fall back to the host vectors (gzip retry round, KSSD row overflow)
  uint32_t stride = 0; int width = 8;
  std::vector<uint32_t> counts; // per genome id
};

// run fn(g) on one host thread per GPU and wait (a context is used by one thread at a time)
template <typename F>
static void on_all_gpus(std::vector<Gpu>& gpus, F fn) {
  std::vector<std::thread> th;
  for (size_t g = 1; g < gpus.size(); g++) th.emplace_back([&fn, g]() { fn(g); });
  fn(0);
  for (auto& t : th) t.join();
}

static void upload_sketches(rtc_ctx* ctx, const vector<vector<uint64_t>>* h64, const vector<vector<uint32_t>>* h32,
                            DeviceSketches& ds) {
  const uint32_t n = (uint32_t)(h64 ? h64->size() : h32->size());
  ds.n = n; ds.width = h64 ? 8 : 4;
  vector<uint64_t> start(n); vector<uint32_t> len(n);
  uint64_t tot = 0;
  for (uint32_t g = 0; g < n; g++) { start[g] = tot; len[g] = (uint32_t)(h64 ? (*h64)[g].size() : (*h32)[g].size()); tot += len[g]; }
  vector<unsigned char> flat((size_t)tot * ds.width);
  for (uint32_t g = 0; g < n; g++) {
    if (!len[g]) continue;
    if (h64) memcpy(flat.data() + start[g] * 8, (*h64)[g].data(), (size_t)len[g] * 8);
    else memcpy(flat.data() + start[g] * 4, (*h32)[g].data(), (size_t)len[g] * 4);
  }
  CHECK(ctx, rtc_dev_alloc(ctx, flat.size() + 64, &ds.d_hashes));
  CHECK(ctx, rtc_dev_alloc(ctx, (size_t)n * 8 + 64, (void**)&ds.d_start));
  CHECK(ctx, rtc_dev_alloc(ctx, (size_t)n * 4 + 64, (void**)&ds.d_len));
  CHECK(ctx, rtc_copy_h2d(ctx, ds.d_hashes, flat.data(), flat.size()));
  CHECK(ctx, rtc_copy_h2d(ctx, ds.d_start, start.data(), (size_t)n * 8));
  CHECK(ctx, rtc_copy_h2d(ctx, ds.d_len, len.data(), (size_t)n * 4));
}

// the CSR over the resident rows: start[g] = order[g]*stride (order: processing order -> genome id), len from the counts
static void resident_sketches(rtc_ctx* ctx, const Gpu& gp, const Resident& rs, const vector<uint32_t>* order, DeviceSketches& ds) {
  const uint32_t n = (uint32_t)rs.counts.size();
  ds.n = n; ds.width = rs.width; ds.d_hashes = gp.d_sk;
  vector<uint64_t> start(n); vector<uint32_t> len(n);
  for (uint32_t g = 0; g < n; g++) { const uint32_t src = order ? (*order)[g] : g; start[g] = (uint64_t)src * rs.stride; len[g] = rs.counts[src]; }
  CHECK(ctx, rtc_dev_alloc(ctx, (size_t)n * 8 + 64, (void**)&ds.d_start));
  CHECK(ctx, rtc_dev_alloc(ctx, (size_t)n * 4 + 64, (void**)&ds.d_len));
  CHECK(ctx, rtc_copy_h2d(ctx, ds.d_start, start.data(), (size_t)n * 8));
  CHECK(ctx, rtc_copy_h2d(ctx, ds.d_len, len.data(), (size_t)n * 4));
}

static vector<string> read_list(const string& inputFile) {
  fprintf(stderr, "-----input fileList, sketch by file\n");
  ifstream fs(inputFile);
  if (!fs) { fprintf(stderr, "error open the inputFile: %s\n", inputFile.c_str()); exit(1); }
  vector<string> fileList; string fileName;
  while (getline(fs, fileName)) if (!fileName.empty()) fileList.push_back(fileName);
  return fileList;
}

// sketchFiles / sketchFileWithKssd (src/SketchInfo.cpp:865-992, :994-1252): parse on the host
// cores (OpenMP over files), sketch on the GPU in batches of a few GiB of bases.
struct SketchJob {
  bool kssd = false; int kmerSize = 21; int sketchSize = 1000; bool isContainment = false; int containCompress = 1000;
  int drlevel = 3; uint64_t minLen = 10000; int threads = 1;
};

// One batch of consecutive list entries: file f gets the slot [slot_off, slot_off + slot_len) of the
// page-locked staging buffer; parser threads write the bases straight into it.
static char* alloc_pageable(size_t bytes) {  // 2 MiB-aligned, transparent huge pages if the host allows
  void* p = nullptr;
  if (posix_memalign(&p, (size_t)2 << 20, bytes) != 0) return nullptr;
#ifdef RTC_MEASURE
  if (getenv("RTC_NO_THP")) return (char*)p;  // (without huge pages the parse is 15 % slower and the exit 0.04 s longer)
#endif
  madvise(p, bytes, MADV_HUGEPAGE);
  return (char*)p;
}

struct Batch { vector<size_t> files; vector<uint64_t> slot_off, slot_len; uint64_t bytes = 0; };

struct FileResult {           // indexed by list position, so late (retried) files keep list order
  SequenceInfo first; uint64_t total = 0; int flen = 0; bool kept = false;
  vector<uint64_t> h64; vector<uint32_t> h32;
  bool on_host = false;  // h64 / h32 hold the sketch (otherwise it lives in HBM only)
};

// gpus_ready (optional): the HIP runtime and the contexts are still coming up on another thread (main); wait() returns
// when `gpus` is filled, *done says whether it already is.  Everything the host can do alone happens before wait() is
// called -- the list, the slot sizes, the plan, and the PARSING: batch after batch into staging buffers of their own for
// as long as the GPUs are not there (0.05-0.24 s of runtime start-up = 6-28 GB of FASTA at the parser's rate), which
// the lanes then take back to back.
struct GpusReady { std::function<void()> wait; const std::atomic<bool>* done; };
static void sketch_files(vector<Gpu>& gpus, const string& inputFile, const SketchJob& job, vector<GenomeInfo>& genomes,
                         MinHashSketchFile* mh, KssdSketchFile* ks, Resident& rs, bool need_host_hashes,
                         const GpusReady* gpus_ready = nullptr) {
  const double tp00 = get_sec();
  const vector<string> fileList = read_list(inputFile);
  const size_t nfiles = fileList.size();
  const bool verbose = getenv("RTC_VERBOSE") != nullptr;
  // Staging format: the parser threads pack the bases to 2 bits and list everything that is not ACGT as runs
  // (rtc_host: read_genome_file_packed); the GPU expands the batch again in HBM (rtc_unpack_bases_dev) in front of the
  // sketch kernel.  A quarter of the bytes cross PCIe, which is what bounds this command line.  RTC_STAGE_ASCII=1
  // stages the characters themselves (the former path; tests compare the two).
  const bool packed = getenv("RTC_STAGE_ASCII") == nullptr;
  uint64_t BATCH_BYTES = (uint64_t)1 << 30;  // measured best on a 16-core quota: 0.5-1 GiB
  if (const char* e = getenv("RTC_BATCH_BYTES")) BATCH_BYTES = std::max<uint64_t>(strtoull(e, nullptr, 10), 1 << 20);
  vector<int32_t> shuffled;
  std::thread shuffle_thread;  // the 2^24-entry glibc-rand shuffle takes ~0.7 s: built while batch 0 is parsed
  int half_subk = 6;
  if (job.kssd) {
    const int half_k = (job.kmerSize + 1) / 2;
    half_subk = 6 - job.drlevel >= 2 ? 6 : job.drlevel + 2;
    shuffle_thread = std::thread([&shuffled, half_subk]() { shuffled = generate_shuffle_dim(half_subk); });
    ks->info.half_k = half_k; ks->info.half_subk = half_subk; ks->info.drlevel = job.drlevel;
    ks->info.id = (half_k << 8) + (half_subk << 4) + job.drlevel;        // :1030
    ks->info.genomeNumber = (int)fileList.size();                         // :1031
    ks->use64 = half_k - job.drlevel > 8;
  }
  const bool use64 = job.kssd ? ks->use64 : true;
  if (verbose) fprintf(stderr, "[init]  list + shuffle table in %.3fs\n", get_sec() - tp00);

  // ---- slot sizes (upper bound of the bases a file yields), batches of ~BATCH_BYTES ----
  const double tp0 = get_sec();
  vector<uint64_t> slot(nfiles);
  vector<FileResult> res(nfiles);
#pragma omp parallel for num_threads(job.threads) schedule(static)
  for (long i = 0; i < (long)nfiles; i++) {
    slot[i] = genome_slot_bytes(fileList[i]);
    res[i].flen = job.isContainment ? file_length_for_containment(fileList[i]) : 0;
  }
  for (size_t i = 0; i < nfiles; i++)
    if (slot[i] == 0) { fprintf(stderr, "cannot open the genome file: %s\n", fileList[i].c_str()); exit(1); }
  auto plan = [&](const vector<size_t>& files, const vector<uint64_t>& need) {
    vector<Batch> out;
    for (size_t q = 0; q < files.size(); q++) {
      if (out.empty() || (out.back().bytes + need[q] + 64 > BATCH_BYTES && !out.back().files.empty())) out.emplace_back();
      Batch& b = out.back();
      const uint64_t nd = packed ? (need[q] + 63) & ~(uint64_t)63 : need[q];  // packed slots start on 16-byte boundaries
      b.files.push_back(files[q]); b.slot_off.push_back(b.bytes); b.slot_len.push_back(nd);
      b.bytes += nd;
    }
    return out;
  };
  vector<size_t> all(nfiles);
  iota(all.begin(), all.end(), 0);
  vector<Batch> batches = plan(all, slot);
  uint64_t maxb = 0;
  for (const Batch& b : batches) maxb = std::max(maxb, b.bytes);

  // ---- the parser side of one batch: every file of it into its slot of `buf`, the runs of the packed format collected ----
  vector<size_t> retry_files; vector<uint64_t> retry_need;
  vector<size_t> row_file;  // row (= genome id) -> list position
  double parse_s = 0;       // wall time the parser threads took, summed over the batches (the GPU lanes work beside it)
  uint64_t parse_bytes = 0;
  std::atomic<uint64_t> gpu_copy_us{0}, gpu_sketch_us{0}, gpu_batches{0}, runs_total{0};  // the lanes' side of the batches (metrics)
  auto parse_batch = [&](const Batch& b, size_t bi, char* buf, vector<uint64_t>& bruns, int round) -> uint32_t {
    const double t0 = get_sec();
    vector<uint64_t> need(b.files.size(), 0);
    vector<vector<uint64_t>> fruns(packed ? b.files.size() : 0);
#pragma omp parallel for num_threads(job.threads) schedule(dynamic)
    for (long q = 0; q < (long)b.files.size(); q++) {
      FileResult& r = res[b.files[q]];
      uint64_t used = 0, nrec = 0;
      int st;
      if (packed) st = read_genome_file_packed(fileList[b.files[q]], (uint8_t*)buf + b.slot_off[q] / 4, b.slot_len[q], used, fruns[q], r.first, r.total, nrec);
      else st = read_genome_file_flat(fileList[b.files[q]], buf + b.slot_off[q], b.slot_len[q], used, r.first, r.total, nrec);
      if (st == 1) { fprintf(stderr, "cannot open the genome file: %s\n", fileList[b.files[q]].c_str()); exit(1); }
      if (st == 2) { need[q] = used + 1; used = 0; r.kept = false; }
      else r.kept = r.total >= job.minLen;                                 // :963
      if (!r.kept) used = 0;
      // no k-mers in the gap behind the genome, nor in dropped genomes
      if (!packed) memset(buf + b.slot_off[q] + used, 'N', b.slot_len[q] - used);
      else {
        vector<uint64_t>& fr = fruns[q];
        while (!fr.empty() && fr[fr.size() - 2] >= used) { fr.pop_back(); fr.pop_back(); }
        if (!fr.empty() && fr[fr.size() - 2] + fr[fr.size() - 1] > used) fr[fr.size() - 1] = used - fr[fr.size() - 2];
        if (b.slot_len[q] > used) { fr.push_back(used); fr.push_back(b.slot_len[q] - used); }
      }
    }
    if (!packed) memset(buf + b.bytes, 'N', 64);
    else {
      bruns.clear();
      for (size_t q = 0; q < b.files.size(); q++)
        for (size_t e = 0; e + 1 < fruns[q].size(); e += 2) { bruns.push_back(b.slot_off[q] + fruns[q][e]); bruns.push_back(fruns[q][e + 1]); }
      bruns.push_back(b.bytes); bruns.push_back(64);
    }
    uint32_t nkept = 0;
    for (size_t q = 0; q < b.files.size(); q++) {
      if (need[q]) { retry_files.push_back(b.files[q]); retry_need.push_back(need[q]); }
      if (res[b.files[q]].kept) { nkept++; if (round == 0) row_file.push_back(b.files[q]); }
    }
    parse_s += get_sec() - t0;
    parse_bytes += b.bytes;
    if (verbose) fprintf(stderr, "[parse] batch %zu: %zu files, %.2f GB in %.3fs\n", bi, b.files.size(), b.bytes / 1e9, get_sec() - t0);
    return nkept;
  };
  // Pageable staging by default: page-locking costs ~0.15 s/GB up front while the pageable PCIe copy
  // already runs at > 30 GB/s on the MI355X hosts measured; RTC_STAGE_PINNED=1 page-locks instead.
  bool pinned = getenv("RTC_STAGE_PINNED") != nullptr;
  // batches parsed while the GPUs come up (pageable staging only: page-locking needs a context).  The first PRE_RING of
  // them go straight into what becomes the staging ring (every configuration has at least three ring buffers: two lanes
  // + the one being parsed), so they cost no memory of their own -- every GiB of staging a run touches is 0.06 s at
  // process exit (the kernel frees it page by page), which is what a fourth pre-parsed 1-GiB batch in a buffer of its
  // own used to cost a 0.5 s run while saving it 0.05 s.  RTC_PREPARSE_BYTES (default: the three ring buffers) allows
  // more batches, each in a buffer of its own; 0 = parse only once the GPUs are there.
  struct PreBatch { char* buf; vector<uint64_t> runs; uint32_t kept; };
  vector<PreBatch> pre;
  const size_t PRE_RING = 3;
  const uint64_t ring_host_bytes = packed ? maxb / 4 + 64 : maxb + 64;
  if (gpus_ready && !pinned) {
    uint64_t budget = PRE_RING * ring_host_bytes, used = 0;
    if (const char* e = getenv("RTC_PREPARSE_BYTES")) budget = strtoull(e, nullptr, 10);
    while (pre.size() < batches.size() && !gpus_ready->done->load(std::memory_order_acquire)) {
      const Batch& b = batches[pre.size()];
      const uint64_t host_bytes = pre.size() < PRE_RING ? ring_host_bytes : packed ? b.bytes / 4 + 64 : b.bytes + 64;
      if (used + host_bytes > budget) break;
      char* buf = alloc_pageable(host_bytes);
      if (!buf) break;
      used += host_bytes;
      pre.push_back(PreBatch{buf, {}, 0});
      pre.back().kept = parse_batch(b, pre.size() - 1, buf, pre.back().runs, 0);
    }
    if (verbose) fprintf(stderr, "[init]  %zu of %zu batches parsed before the GPUs were up\n", pre.size(), batches.size());
  }
  if (gpus_ready) gpus_ready->wait();
  rtc_ctx* ctx = gpus[0].ctx;
  const size_t G = gpus.size();
  // Two lanes per GPU: lane 0 is the GPU's own context, lane 1 a second context on the same device with a
  // stream of its own, each driven by its own host thread -- the PCIe copy of one batch runs beside the
  // sketch kernel of the previous one.  Both lanes write rows of the same resident sketch buffer.
  struct Lane { rtc_ctx* ctx; void* d_seq; std::thread worker; size_t gpu; bool owned; void* d_packed = nullptr; void* d_runs = nullptr; size_t runs_cap = 0; };
  const size_t LPG = getenv("RTC_SINGLE_LANE") ? 1 : 2;
  vector<Lane> lanes(G * LPG);
  for (size_t l = 0; l < lanes.size(); l++) {
    lanes[l].gpu = l % G; lanes[l].d_seq = nullptr; lanes[l].owned = l >= G; lanes[l].ctx = gpus[l % G].ctx;
    if (lanes[l].owned) {
      CHECK(ctx, rtc_ctx_create(gpus[l % G].device, &lanes[l].ctx));
      CHECK(lanes[l].ctx, rtc_ctx_own_stream(lanes[l].ctx));
    }
  }
  const size_t NL = lanes.size();

  // ---- staging: G+1 host buffers (one being parsed, one per GPU in flight) and one device buffer per GPU ----
  const size_t NSTAGE = NL + 1;
  uint64_t buf_bytes = 0;
  vector<char*> stage(NSTAGE, nullptr);
  vector<vector<uint64_t>> stage_runs(NSTAGE);  // packed staging: the batch's runs of characters outside ACGT, batch coordinates
  auto free_stage = [&]() {
    for (size_t i = 0; i < NSTAGE; i++) {
      if (stage[i] && pinned) CHECK(ctx, rtc_host_free(ctx, stage[i]));
      else free(stage[i]);
      stage[i] = nullptr;
    }
  };
  auto ensure_buffers = [&](uint64_t need) {
    if (need <= buf_bytes) return;
    free_stage();
    for (Lane& l : lanes) {
      if (l.d_seq) { CHECK(l.ctx, rtc_dev_free(l.ctx, l.d_seq)); l.d_seq = nullptr; }
      if (l.d_packed) { CHECK(l.ctx, rtc_dev_free(l.ctx, l.d_packed)); l.d_packed = nullptr; }
    }
    const bool first = buf_bytes == 0;
    buf_bytes = need;
    const uint64_t host_bytes = packed ? buf_bytes / 4 + 64 : buf_bytes + 64;
    for (int i = 0; i < (int)NSTAGE; i++) {
      // the buffers the first batches were parsed into while the GPUs came up ARE ring slots 0 .. PRE_RING - 1 (batch i
      // of the pipeline below lives in slot i % NSTAGE, NSTAGE >= PRE_RING)
      if (first && !pinned && (size_t)i < PRE_RING && (size_t)i < pre.size() && host_bytes == ring_host_bytes) { stage[i] = pre[i].buf; continue; }
      if (pinned && rtc_host_alloc(ctx, host_bytes, (void**)&stage[i]) != RTC_OK) {
        // the host refuses to page-lock this much (ulimit -l): stage through pageable memory instead
        fprintf(stderr, "-----cannot page-lock %.2f GB (%s), staging through pageable memory\n", buf_bytes / 1e9, rtc_last_error(ctx));
        free_stage();
        pinned = false; i = -1;
        continue;
      }
      if (!pinned && !(stage[i] = alloc_pageable(host_bytes))) { fprintf(stderr, "ERROR: cannot allocate %.2f GB of staging memory\n", buf_bytes / 1e9); exit(1); }
    }
    // (the device staging buffers are allocated by the lanes themselves when their first batch arrives: lane_buffers)
  };
  // A lane's device staging, allocated on the lane's own host thread in front of its first copy: the two lanes of a GPU do
  // it side by side and the main thread is already parsing (two 1-GiB hipMallocs and two of 256 MiB used to sit between
  // "the GPUs are there" and the first batch).  The character buffer only where something reads characters.
  auto lane_buffers = [&](Lane& l, bool need_chars) {
    const uint64_t host_bytes = packed ? buf_bytes / 4 + 64 : buf_bytes + 64;
    if (packed && !l.d_packed) CHECK(l.ctx, rtc_dev_alloc(l.ctx, host_bytes, &l.d_packed));
    if (need_chars && !l.d_seq) CHECK(l.ctx, rtc_dev_alloc(l.ctx, buf_bytes + 128, &l.d_seq));
  };
  // Sketching over packed staging: the kernels read the 2-bit stream themselves -- rtc_sketch_minhash_packed_dev for every k,
  // rtc_sketch_kssd_packed_dev for the k-mer lengths the prefilter kernel covers.  RTC_SKETCH_UNPACK=1 (for --fast also the
  // older RTC_KSSD_UNPACK=1) expands every batch to characters in HBM first (the former path; tests compare the stagings)
  std::atomic<bool> direct{packed && getenv("RTC_SKETCH_UNPACK") == nullptr &&
                           (!job.kssd || (getenv("RTC_KSSD_UNPACK") == nullptr && half_subk == 6 && job.drlevel >= 3 &&
                                          (job.kmerSize + 1) / 2 * 2 >= 18 && (job.kmerSize + 1) / 2 * 2 <= 28))};
  ensure_buffers(maxb);
  // Each lane's first PCIe copy costs 17-29 ms instead of 6 (the runtime sets up the stream's copy path) and its buffers 5 ms:
  // the lanes do both now, on their worker threads, while this thread sizes and allocates the resident rows; the pipeline
  // joins a lane's worker before it hands it a batch.
  if (!getenv("RTC_NO_WARMUP"))
    for (size_t l = 0; l < NL; l++) {
      Lane* lp = &lanes[l];
      const char* src = stage[l % NSTAGE];
      const uint64_t warm_bytes = std::min<uint64_t>((uint64_t)8 << 20, packed ? buf_bytes / 4 : buf_bytes);
      lp->worker = std::thread([&lane_buffers, &direct, lp, src, warm_bytes, packed]() {
        lane_buffers(*lp, !direct.load());
        CHECK(lp->ctx, rtc_copy_h2d(lp->ctx, packed ? lp->d_packed : lp->d_seq, src, warm_bytes));
      });
    }
  if (verbose) fprintf(stderr, "[plan] %zu files, %zu batches, %zu GPU(s), staging %zu x %.2f GB, %.3fs\n", nfiles, batches.size(), G, NSTAGE, buf_bytes / 1e9, get_sec() - tp0);

  // ---- resident sketch rows: genome id g (list order among the kept files) owns row g on every GPU ----
  auto size_of = [&](size_t file) -> uint32_t {
    return job.isContainment ? (uint32_t)std::max(res[file].flen / job.containCompress, 100) : (uint32_t)job.sketchSize;  // :919-924
  };
  rs.width = job.kssd ? (use64 ? 8 : 4) : 8;
  rs.stride = 1;
  if (!job.kssd) for (size_t i = 0; i < nfiles; i++) rs.stride = std::max(rs.stride, size_of(i));
  else { uint64_t ms = 0; for (size_t i = 0; i < nfiles; i++) ms = std::max(ms, slot[i]); rs.stride = (uint32_t)(ms / (1ull << (4 * job.drlevel)) * 3 / 2 + 256); }
  rs.ok = getenv("RTC_HOST_SKETCHES") == nullptr;  // the switch forces the upload path (tests compare the two)
  rs.counts.assign(nfiles, 0);
  // The rows share ONE stride (the largest sketch any file can yield), so a single outlier -- one 3 Gbp assembly among
  // 100 000 bacterial genomes -- inflates every row.  The buffer has to fit beside the staging buffers with room left
  // for the clustering phase; when it does not (or the allocation fails) the run keeps per-batch temporary buffers
  // and clusters from the host vectors instead of stopping.
  if (rs.ok) {
    const size_t want = (size_t)nfiles * rs.stride * rs.width + (size_t)nfiles * 4 + 128;
    size_t budget = 0;
    if (const char* e = getenv("RTC_RESIDENT_BUDGET")) budget = strtoull(e, nullptr, 10);  // bytes (tests of the fallback)
    for (Gpu& g : gpus) {
      size_t fr = 0, tot = 0;
      CHECK(g.ctx, rtc_dev_mem_info(g.ctx, &fr, &tot));
      const size_t staging = NL / G * (size_t)(buf_bytes + buf_bytes / 4 + 256);  // the lanes allocate theirs with their first batch
      const size_t lim = budget ? budget : (fr > staging ? fr - staging : 0) / 2;
      if (want > lim) {
        fprintf(stderr, "-----resident sketch rows would take %.2f GB (%zu genomes x %u hashes, the largest genome sets the row) of %.2f GB free: "
                        "sketches go through host memory instead\n", want / 1e9, nfiles, rs.stride, fr / 1e9);
        rs.ok = false;
        break;
      }
    }
  }
  if (rs.ok)
    for (Gpu& g : gpus) {
      if (rtc_dev_alloc(g.ctx, (size_t)nfiles * rs.stride * rs.width + 64, &g.d_sk) != RTC_OK ||
          rtc_dev_alloc(g.ctx, (size_t)nfiles * 4 + 64, (void**)&g.d_cnt) != RTC_OK) {
        fprintf(stderr, "-----cannot keep the sketches resident (%s): sketches go through host memory instead\n", rtc_last_error(g.ctx));
        rs.ok = false;
        break;
      }
      // every count is written by the sketch kernels of its batch; rows of dropped files are never read
    }
  if (!rs.ok)
    for (Gpu& g : gpus) {
      if (g.d_sk) CHECK(g.ctx, rtc_dev_free(g.ctx, g.d_sk));
      if (g.d_cnt) CHECK(g.ctx, rtc_dev_free(g.ctx, g.d_cnt));
      g.d_sk = nullptr; g.d_cnt = nullptr;
    }
  std::atomic<bool> resident_ok{rs.ok};
  struct Placed { uint32_t row0, rows; int gpu; };  // where a batch's sketches live (share step)
  vector<Placed> placed;

  // ---- GPU side of one batch (runs on that GPU's host thread while the next batch is parsed) ----
  // row0 < 0: not resident (retry round), results only go to the host vectors.
  auto gpu_batch = [&](Lane& ln, const Batch& b, const char* h_seq, const vector<uint64_t>* h_runs, long row0) {
    Gpu& gp = gpus[ln.gpu];
    rtc_ctx* c = ln.ctx;
    const double t0 = get_sec();
    vector<uint64_t> off; vector<uint32_t> sizes; vector<size_t> kept;
    for (size_t q = 0; q < b.files.size(); q++) {
      FileResult& r = res[b.files[q]];
      if (!r.kept) continue;
      kept.push_back(b.files[q]);
      off.push_back(b.slot_off[q]);  // a genome extends to the next kept one: the gap holds only 'N'
      sizes.push_back(size_of(b.files[q]));
    }
    const uint32_t nb = (uint32_t)kept.size();
    if (!nb) return;
    off.push_back(b.bytes);
    bool dir = direct.load();  // read once per batch: another lane may switch the run to the unpack path meanwhile
    lane_buffers(ln, !dir);
    const double t0b = get_sec();
    if (packed) {
      const size_t nr = h_runs->size() / 2;
      if (nr > ln.runs_cap) {
        if (ln.d_runs) CHECK(c, rtc_dev_free(c, ln.d_runs));
        ln.runs_cap = nr + nr / 2 + 1024;
        CHECK(c, rtc_dev_alloc(c, ln.runs_cap * 16, &ln.d_runs));
      }
      CHECK(c, rtc_copy_h2d(c, ln.d_packed, h_seq, b.bytes / 4 + 16));
      CHECK(c, rtc_copy_h2d(c, ln.d_runs, h_runs->data(), nr * 16));
      if (!dir)
        CHECK(c, rtc_unpack_bases_dev(c, (const uint8_t*)ln.d_packed, b.bytes + 64, (const uint64_t*)ln.d_runs, nr, (uint8_t*)ln.d_seq));
    } else {
      CHECK(c, rtc_copy_h2d(c, ln.d_seq, h_seq, b.bytes + 64));
    }
    const double t1 = get_sec();
    const bool resident = row0 >= 0 && resident_ok.load();
    const bool to_host = need_host_hashes || !resident;
    const int w = rs.width;
    uint32_t* d_cnt = nullptr;
    if (resident) d_cnt = gp.d_cnt + row0;
    else CHECK(c, rtc_dev_alloc(c, (size_t)nb * 4, (void**)&d_cnt));
    vector<uint32_t> cnt(nb);
    uint32_t stride = rs.stride;
    void* d_out = nullptr;
    if (resident) d_out = (char*)gp.d_sk + (size_t)row0 * rs.stride * w;
    bool row_overflow = false;
    if (!job.kssd) {
      if (!resident) { stride = *std::max_element(sizes.begin(), sizes.end()); CHECK(c, rtc_dev_alloc(c, (size_t)nb * stride * 8, &d_out)); }
      // the batch is sketched as it crossed PCIe (the records update() is handed, src/SketchInfo.cpp:928-948, at 2 bits a base)
      if (dir) CHECK(c, rtc_sketch_minhash_packed_dev(c, (const uint8_t*)ln.d_packed, b.bytes + 64, (const uint64_t*)ln.d_runs, h_runs->size() / 2,
                                                      off.data(), nb, job.kmerSize, 42, sizes.data(), stride, (uint64_t*)d_out, stride, d_cnt));
      else CHECK(c, rtc_sketch_minhash_dev(c, (const uint8_t*)ln.d_seq, off.data(), nb, job.kmerSize, 42, sizes.data(), stride,
                                           (uint64_t*)d_out, stride, d_cnt));
    } else {
      if (!resident) {
        uint64_t maxlen = 0;
        for (uint32_t g = 0; g < nb; g++) maxlen = std::max(maxlen, off[g + 1] - off[g]);
        stride = (uint32_t)(maxlen / (1ull << (4 * job.drlevel)) * 3 / 2 + 256);
        CHECK(c, rtc_dev_alloc(c, (size_t)nb * stride * w, &d_out));
      }
      while (true) {
        int width = 0; uint32_t need = 0;
        int st;
        if (dir) {
          // the batch is sketched as it crossed PCIe; configurations the packed kernel does not serve (a shuffle table
          // its exact index cannot hold) are expanded in HBM after all, from here on for every batch
          st = rtc_sketch_kssd_packed_dev(c, (const uint8_t*)ln.d_packed, b.bytes + 64, (const uint64_t*)ln.d_runs, h_runs->size() / 2,
                                          off.data(), nb, job.kmerSize, job.drlevel, shuffled.data(), d_out, stride, d_cnt, &width, &need);
          if (st == RTC_ERR_UNSUPPORTED) {
            direct.store(false);
            dir = false;
            lane_buffers(ln, true);
            CHECK(c, rtc_unpack_bases_dev(c, (const uint8_t*)ln.d_packed, b.bytes + 64, (const uint64_t*)ln.d_runs, h_runs->size() / 2, (uint8_t*)ln.d_seq));
            continue;
          }
        } else {
          st = rtc_sketch_kssd_dev(c, (const uint8_t*)ln.d_seq, off.data(), nb, job.kmerSize, job.drlevel, shuffled.data(),
                                   d_out, stride, d_cnt, &width, &need);
        }
        if (st == RTC_ERR_OVERFLOW) {
          // a genome with more tuples than a resident row: this batch goes to a wider temporary buffer and
          // the run falls back to the host vectors (the other batches are pulled from HBM at the end)
          if (resident && !row_overflow) { row_overflow = true; resident_ok.store(false); CHECK(c, rtc_dev_alloc(c, (size_t)nb * 4, (void**)&d_cnt)); }
          else CHECK(c, rtc_dev_free(c, d_out));
          stride = need + 64;
          CHECK(c, rtc_dev_alloc(c, (size_t)nb * stride * w, &d_out));
          continue;
        }
        CHECK(c, st);
        break;
      }
    }
    const bool temp = !resident || row_overflow;
    CHECK(c, rtc_copy_d2h(c, cnt.data(), d_cnt, (size_t)nb * 4));
    if (dir) CHECK(c, rtc_ctx_sync(c));  // the stream is idle after the copy: this only reports a run list that broke its contract, against THIS batch
    if (row0 >= 0) for (uint32_t g = 0; g < nb; g++) rs.counts[row0 + g] = cnt[g];
    if (to_host || temp) {
      vector<unsigned char> out((size_t)nb * stride * w);
      CHECK(c, rtc_copy_d2h(c, out.data(), d_out, out.size()));
      for (uint32_t g = 0; g < nb; g++) {
        FileResult& r = res[kept[g]];
        if (w == 8) { const uint64_t* p = (const uint64_t*)out.data() + (size_t)g * stride; r.h64.assign(p, p + cnt[g]); }
        else { const uint32_t* p = (const uint32_t*)out.data() + (size_t)g * stride; r.h32.assign(p, p + cnt[g]); }
        r.on_host = true;
      }
    }
    if (temp) { CHECK(c, rtc_dev_free(c, d_out)); CHECK(c, rtc_dev_free(c, d_cnt)); }
    gpu_copy_us += (uint64_t)((t1 - t0b) * 1e6);
    gpu_sketch_us += (uint64_t)((get_sec() - t1) * 1e6);
    gpu_batches++;
    if (h_runs) runs_total += h_runs->size() / 2;
    if (verbose) fprintf(stderr, "[gpu %d.%d] %u genomes, %.2f GB: alloc %.3fs h2d %.3fs sketch%s %.3fs\n", (int)ln.gpu, (int)ln.owned, nb, b.bytes / 1e9, t0b - t0, t1 - t0b,
                         to_host || temp ? "+d2h" : "", get_sec() - t1);
  };

  // ---- pipeline: parse batch i into stage[i % (G+1)] while the GPU threads work on batches i-1 .. i-G ----
  size_t done_files = 0, bi = 0;
  uint32_t next_row = 0;
  for (int round = 0; round < 2; round++) {  // round 1: files whose slot guess was too small (gzip ISIZE)
    for (const Batch& b : batches) {
      const bool preparsed = round == 0 && bi < pre.size();  // parsed while the GPUs came up, in a buffer of its own
      char* buf = preparsed ? pre[bi].buf : stage[bi % NSTAGE];
      vector<uint64_t>& bruns = preparsed ? pre[bi].runs : stage_runs[bi % NSTAGE];
      const uint32_t nkept = preparsed ? pre[bi].kept : parse_batch(b, bi, buf, bruns, round);
      if (!retry_files.empty()) resident_ok.store(false);  // retried files arrive out of list order: ids are no longer row numbers
      Lane& ln = lanes[bi % NL];
      if (ln.worker.joinable()) ln.worker.join();
      if (shuffle_thread.joinable()) shuffle_thread.join();
      const Batch* bp = &b;
      const long row0 = round == 0 ? (long)next_row : -1;
      if (round == 0) { placed.push_back(Placed{next_row, nkept, (int)ln.gpu}); next_row += nkept; }
      Lane* lnp = &ln;
      const vector<uint64_t>* brp = &bruns;
      ln.worker = std::thread([&gpu_batch, lnp, bp, buf, brp, row0]() { gpu_batch(*lnp, *bp, buf, brp, row0); });
      for (size_t q = 0; q < b.files.size(); q++, done_files++) if (done_files % 10000 == 0) cerr << "---finished sketching: " << done_files << " genomes" << endl;
      bi++;
    }
    for (Lane& l : lanes) if (l.worker.joinable()) l.worker.join();
    if (round == 1 || retry_files.empty()) break;
    batches = plan(retry_files, retry_need);
    maxb = 0;
    for (const Batch& b : batches) maxb = std::max(maxb, b.bytes);
    ensure_buffers(maxb);
    done_files -= retry_files.size();
    retry_files.clear(); retry_need.clear();
  }
  if (shuffle_thread.joinable()) shuffle_thread.join();
  if (parse_s > 0) {  // the host feed: what ONE host can parse bounds what it can hand to 8 GPUs
    g_metrics.num("parse_s", parse_s);
    g_metrics.num("parse_gbp_per_s", (double)parse_bytes / parse_s / 1e9);
    g_metrics.num("parse_gbp_per_s_per_thread", (double)parse_bytes / parse_s / 1e9 / std::max(1, job.threads));
  }
  if (gpu_batches.load()) {  // per batch, on the lane's host thread: copy (+ unpack where a path still needs characters), sketch + read-back of the counts
    g_metrics.num("batches", (double)gpu_batches.load());
    g_metrics.num("gpu_copy_ms_per_batch", gpu_copy_us.load() / 1e3 / gpu_batches.load());
    g_metrics.num("gpu_sketch_ms_per_batch", gpu_sketch_us.load() / 1e3 / gpu_batches.load());
    g_metrics.num("runs_per_genome", (double)runs_total.load() / std::max<size_t>(1, nfiles));
  }
  {
    double inf_s = 0; uint64_t inf_b = 0;
    rtc_host_inflate_stats(&inf_s, &inf_b);
    if (inf_b) { g_metrics.num("inflate_s_all_threads", inf_s); g_metrics.num("inflate_gb_per_s_per_thread", (double)inf_b / inf_s / 1e9); }
  }
  const double tf0 = get_sec();
  // (the second lanes' contexts and every device staging buffer live until the process ends, see below)
  if (pinned) free_stage();
  // Pageable staging is NOT returned here: munmap of GBs of touched pages takes ~0.1 s and holds the
  // address-space lock, which stalls every hipMalloc of the clustering phase that follows (measured:
  // candidate edges 117 ms instead of 3 ms).  The pages go back when the process ends.
#ifdef RTC_MEASURE
  for (auto& p : stage) if (p) g_exit_probe_host.push_back(p);
  for (Lane& l : lanes) { if (l.d_seq) g_exit_probe_dev.push_back({l.ctx, l.d_seq}); if (l.d_packed) g_exit_probe_dev.push_back({l.ctx, l.d_packed}); }
#endif
  for (auto& p : stage) p = nullptr;
  // The device staging buffers stay allocated until the process ends: hipFree of a multi-GB buffer
  // costs ~0.4 s here and the clustering phase needs far less than the 288 GB that are there.
  if (verbose) fprintf(stderr, "[free]  host staging %.3fs\n", get_sec() - tf0);

  rs.ok = resident_ok.load();
  if (!rs.ok) {
    // fall back to the host vectors: pull the batches whose rows were left in HBM only
    for (const Placed& pl : placed) {
      if (!pl.rows || !gpus[pl.gpu].d_sk || res[row_file[pl.row0]].on_host) continue;
      Gpu& gp = gpus[pl.gpu];
      vector<unsigned char> out((size_t)pl.rows * rs.stride * rs.width);
      CHECK(gp.ctx, rtc_copy_d2h(gp.ctx, out.data(), (char*)gp.d_sk + (size_t)pl.row0 * rs.stride * rs.width, out.size()));
      for (uint32_t g = 0; g < pl.rows; g++) {
        FileResult& r = res[row_file[pl.row0 + g]];
        const uint32_t c = rs.counts[pl.row0 + g];
        if (rs.width == 8) { const uint64_t* q = (const uint64_t*)out.data() + (size_t)g * rs.stride; r.h64.assign(q, q + c); }
        else { const uint32_t* q = (const uint32_t*)out.data() + (size_t)g * rs.stride; r.h32.assign(q, q + c); }
        r.on_host = true;
      }
    }
    for (Gpu& g : gpus) {
      if (g.d_sk) CHECK(g.ctx, rtc_dev_free(g.ctx, g.d_sk));
      if (g.d_cnt) CHECK(g.ctx, rtc_dev_free(g.ctx, g.d_cnt));
      g.d_sk = nullptr; g.d_cnt = nullptr;
    }
  } else if (G > 1 || gpus[0].comm) {
    // ---- share: every batch's rows travel from the GPU that sketched them to all others (RCCL broadcast) ----
    const double ts0 = get_sec();
    on_all_gpus(gpus, [&](size_t g) {
      Gpu& gp = gpus[g];
      for (const Placed& pl : placed) {
        if (!pl.rows) continue;
        CHECK(gp.ctx, rtc_comm_broadcast(gp.comm, (char*)gp.d_sk + (size_t)pl.row0 * rs.stride * rs.width,
                                         (size_t)pl.rows * rs.stride * rs.width, pl.gpu));
        CHECK(gp.ctx, rtc_comm_broadcast(gp.comm, gp.d_cnt + pl.row0, (size_t)pl.rows * 4, pl.gpu));
      }
      CHECK(gp.ctx, rtc_ctx_sync(gp.ctx));
    });
    if (verbose) fprintf(stderr, "[share] %zu batches over %zu GPUs (%s) in %.3fs\n", placed.size(), G, rtc_comm_backend(gpus[0].comm), get_sec() - ts0);
  }
  rs.counts.resize(next_row);

  // ---- assemble in list order ----
  for (size_t i = 0; i < nfiles; i++) {
    FileResult& r = res[i];
    if (!r.kept) continue;
    GenomeInfo gi;
    gi.id = (int)genomes.size();                                        // :964-965 (list order here)
    gi.fileName = fileList[i];
    gi.totalSeqLength = r.total;
    gi.seq0 = r.first;
    gi.use64 = job.kssd ? ks->use64 : false;
    genomes.push_back(std::move(gi));
    if (!job.kssd) mh->hashes.push_back(std::move(r.h64));
    else if (use64) ks->h64.push_back(std::move(r.h64));
    else ks->h32.push_back(std::move(r.h32));
  }
}

// "all CPUs of the platform" (src/main.cpp:75-76,113), bounded by what this process may actually
// use: the affinity mask (omp_get_num_procs) and a cgroup-v2 CPU quota.  Oversubscribing a quota
// makes the parser threads time-slice against each other.
// sketchSequences / sketchSequencesWithKssd (src/SketchInfo.cpp:554-862, consumers :205-272, :274-436): without -l the
// input is ONE FASTA file and every record of at least minLen bases is a genome of its own (SequenceInfo{name, comment,
// strand 0, length}).  Records keep their file order (the reference's RabbitFX consumers merge per-thread vectors, so
// its order depends on the thread count; one consumer gives file order).  The records travel to the GPU back to back
// -- the offsets are the genome boundaries -- in batches; the sketches come back to the host vectors.
struct SeqModeSizes { uint64_t maxSize = 1, minSize = 1u << 31, totalSize = 0; int number = 0, badNumber = 0; };

static bool read_sequences(const string& inputFile, uint64_t minLen, vector<FastaRecord>& kept, SeqModeSizes& sz) {
  const size_t dot = inputFile.find_last_of('.');
  const string suf = dot == string::npos ? string() : inputFile.substr(dot + 1);
  if (suf != "fasta" && suf != "fna" && suf != "fa") {
    cerr << "error input format file: " << inputFile << endl << "Only support FASTA files" << endl;
    exit(1);
  }
  vector<FastaRecord> recs;
  if (!read_fasta(inputFile, recs)) { fprintf(stderr, "ERROR: sketchSequences(), cannot open the genome file, %s\n", inputFile.c_str()); return false; }
  for (FastaRecord& r : recs) {   // calSize, sequence branch (src/SketchInfo.cpp:484-535)
    const uint64_t len = r.seq.size();
    if (len < minLen) { sz.badNumber++; continue; }
    sz.maxSize = std::max(sz.maxSize, len); sz.minSize = std::min(sz.minSize, len); sz.totalSize += len; sz.number++;
    kept.push_back(std::move(r));
  }
  if (sz.number == 0) { cerr << "ERROR: calSize(), no sequence passes the minimum length filter" << endl; return false; }
  const int totalNumber = sz.number + sz.badNumber;
  cerr << "\t===the genome number for clustering is: " << sz.number << endl
       << "\t===the genome number below the minimum genome length threshold is: " << sz.badNumber << endl
       << "\t===the total genome number is: " << totalNumber << endl;
  if ((double)sz.badNumber / totalNumber >= 0.2)
    fprintf(stderr, "Warning: there are %d poor quality (length < %ld) genome assemblies in the total %d genome assemblied.\n", sz.badNumber, (long)minLen, totalNumber);
  cerr << "\t===the totalSize is: " << sz.totalSize << endl << "\t===the maxSize is: " << sz.maxSize << endl
       << "\t===the minSize is: " << sz.minSize << endl << "\t===the averageSize is: " << sz.totalSize / sz.number << endl;
  return true;
}

static void sketch_sequences(vector<Gpu>& gpus, vector<FastaRecord>& recs, const SketchJob& job, vector<GenomeInfo>& genomes,
                             MinHashSketchFile* mh, KssdSketchFile* ks) {
  rtc_ctx* c = gpus[0].ctx;
  const size_t n = recs.size();
  vector<int32_t> shuffled;
  int half_subk = 0;
  if (job.kssd) {
    half_subk = 6 - job.drlevel >= 2 ? 6 : job.drlevel + 2;
    shuffled = generate_shuffle_dim(half_subk);
    const int half_k = (job.kmerSize + 1) / 2;
    ks->info.half_k = half_k; ks->info.half_subk = half_subk; ks->info.drlevel = job.drlevel;
    ks->info.id = (half_k << 8) + (half_subk << 4) + job.drlevel; ks->info.genomeNumber = (int)n;
    ks->use64 = half_k - job.drlevel > 8;
  }
  genomes.resize(n);
  for (size_t i = 0; i < n; i++) {
    GenomeInfo& g = genomes[i];
    g.id = (int)i; g.totalSeqLength = recs[i].seq.size(); g.use64 = job.kssd && ks->use64;
    g.seq0.name = recs[i].name; g.seq0.comment = recs[i].has_comment ? recs[i].comment : string();
    g.seq0.strand = 0; g.seq0.length = (int)recs[i].seq.size();
  }
  if (job.kssd) { if (ks->use64) ks->h64.resize(n); else ks->h32.resize(n); } else mh->hashes.resize(n);
  const uint64_t BATCH = (uint64_t)1 << 30;
  void* d_seq = nullptr; uint64_t d_seq_bytes = 0;
  for (size_t i0 = 0; i0 < n;) {
    size_t i1 = i0; uint64_t bytes = 0;
    while (i1 < n && (i1 == i0 || bytes + recs[i1].seq.size() <= BATCH)) bytes += recs[i1++].seq.size();
    const uint32_t nb = (uint32_t)(i1 - i0);
    if (bytes + 64 > d_seq_bytes) { if (d_seq) CHECK(c, rtc_dev_free(c, d_seq)); d_seq_bytes = bytes + 64; CHECK(c, rtc_dev_alloc(c, d_seq_bytes, &d_seq)); }
    vector<char> h_seq(bytes + 64, 'N');
    vector<uint64_t> off(nb + 1, 0);
    vector<uint32_t> sizes(nb);
    for (uint32_t q = 0; q < nb; q++) {
      const string& sq = recs[i0 + q].seq;
      memcpy(h_seq.data() + off[q], sq.data(), sq.size());
      off[q + 1] = off[q] + sq.size();
      sizes[q] = job.isContainment ? (uint32_t)std::max((int)sq.size() / job.containCompress, 100) : (uint32_t)job.sketchSize;   // :224-231
    }
    CHECK(c, rtc_copy_h2d(c, d_seq, h_seq.data(), bytes + 64));
    uint32_t* d_cnt = nullptr; void* d_out = nullptr;
    CHECK(c, rtc_dev_alloc(c, (size_t)nb * 4 + 64, (void**)&d_cnt));
    uint32_t stride = 0; int w = 8;
    if (!job.kssd) {
      stride = *std::max_element(sizes.begin(), sizes.end());
      CHECK(c, rtc_dev_alloc(c, (size_t)nb * stride * 8 + 64, &d_out));
      CHECK(c, rtc_sketch_minhash_dev(c, (const uint8_t*)d_seq, off.data(), nb, job.kmerSize, 42, sizes.data(), stride, (uint64_t*)d_out, stride, d_cnt));
    } else {
      w = ks->use64 ? 8 : 4;
      uint64_t maxlen = 0;
      for (uint32_t q = 0; q < nb; q++) maxlen = std::max(maxlen, off[q + 1] - off[q]);
      stride = (uint32_t)(maxlen / (1ull << (4 * job.drlevel)) * 3 / 2 + 256);
      CHECK(c, rtc_dev_alloc(c, (size_t)nb * stride * w + 64, &d_out));
      while (true) {
        int width = 0; uint32_t need = 0;
        const int st = rtc_sketch_kssd_dev(c, (const uint8_t*)d_seq, off.data(), nb, job.kmerSize, job.drlevel, shuffled.data(), d_out, stride, d_cnt,
                                           &width, &need);
        if (st == RTC_ERR_OVERFLOW) {
          CHECK(c, rtc_dev_free(c, d_out));
          stride = need + 64;
          CHECK(c, rtc_dev_alloc(c, (size_t)nb * stride * w + 64, &d_out));
          continue;
        }
        CHECK(c, st);
        break;
      }
    }
    vector<uint32_t> cnt(nb);
    CHECK(c, rtc_copy_d2h(c, cnt.data(), d_cnt, (size_t)nb * 4));
    vector<unsigned char> out((size_t)nb * stride * w);
    CHECK(c, rtc_copy_d2h(c, out.data(), d_out, out.size()));
    for (uint32_t q = 0; q < nb; q++) {
      if (w == 8) {
        const uint64_t* p = (const uint64_t*)out.data() + (size_t)q * stride;
        if (job.kssd) ks->h64[i0 + q].assign(p, p + cnt[q]); else mh->hashes[i0 + q].assign(p, p + cnt[q]);
      } else {
        const uint32_t* p = (const uint32_t*)out.data() + (size_t)q * stride;
        ks->h32[i0 + q].assign(p, p + cnt[q]);
      }
    }
    CHECK(c, rtc_dev_free(c, d_out)); CHECK(c, rtc_dev_free(c, d_cnt));
    for (size_t i = i0; i < i1; i++) { string().swap(recs[i].seq); }
    i0 = i1;
  }
  if (d_seq) CHECK(c, rtc_dev_free(c, d_seq));
}

static int default_threads() {
  int n = omp_get_num_procs();
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char q[32] = {0}; long period = 0;
    if (fscanf(f, "%31s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
      const long quota = atol(q);
      if (quota > 0) n = (int)std::min<long>(n, std::max<long>(1, (quota + period - 1) / period));
    }
    fclose(f);
  }
  return std::max(n, 1);
}

struct Options {
  string inputFile, outputFile, folder_path, premsted;
  bool newick = false, phylip = false, nexus = false, linkage = false;  // --newick-tree / --phylip-tree / --nexus-tree / --linkage-matrix
  bool dense = false;       // --dense: density maps, ANI histogram and the MST noise-removal pass
  bool has_append = false;  // --append LIST: inputFile holds the genomes to add to --presketched/--premsted DIR
  bool saveRep = false;     // clust-greedy --fast --save-rep: cluster_state.bin beside the sketches
  string repdb_path;        // clust-greedy --fast --db FILE with one of --build / --query / --assign / --append / --stats
  bool db_build = false, db_query = false, db_assign = false, db_stats = false;
  int topk = 5;             // --top-k of --query
  int threads = default_threads();
  bool sketchByFile = false, noSave = false, is_fast = false, isContainment = false, isJaccard = false, isSetKmer = false;
  bool has_threshold = false, has_input = false, has_presketched = false, has_premsted = false, has_output = false;
  double threshold = 0.05;
  int kmerSize = 19, sketchSize = 1000, containCompress = 1000, drlevel = 3;
  uint64_t minLen = 10000;
  bool useIndex = true;  // --inverted-index=false: the dense estimator loops (modifyMST / greedyCluster) instead of the index path
  string gpus;  // --gpus / RTC_GPUS: "all" (default), a count, or a comma list of device ordinals
};

static void unsupported(const char* what) {
  fprintf(stderr, "ERROR: %s is outside the sketch + all-pairs path this build implements\n", what);
  exit(1);
}

static Options parse(int argc, char** argv) {
  Options o;
  auto need = [&](int& i) -> const char* { if (i + 1 >= argc) { fprintf(stderr, "ERROR: option %s requires a value\n", argv[i]); exit(1); } return argv[++i]; };
  for (int i = 1; i < argc; i++) {
    const string a = argv[i];
    if (a == "-t" || a == "--threads") { o.threads = atoi(need(i)); }
    else if (a == "-m" || a == "--min-length") { o.minLen = strtoull(need(i), nullptr, 10); fprintf(stderr, "-----set the filter minimum length: %ld\n", (long)o.minLen); }
    else if (a == "-c" || a == "--containment") { o.containCompress = atoi(need(i)); o.isContainment = true; fprintf(stderr, "-----use AAF distance with containment coefficient, the sketch size is in porportion with 1/%d\n", o.containCompress); }
    else if (a == "-k" || a == "--kmer-size") { o.kmerSize = atoi(need(i)); o.isSetKmer = true; fprintf(stderr, "-----set kmerSize: %d\n", o.kmerSize); }
    else if (a == "-s" || a == "--sketch-size") { o.sketchSize = atoi(need(i)); o.isJaccard = true; fprintf(stderr, "-----set sketchSize:  %d\n", o.sketchSize); }
    else if (a == "-l" || a == "--list") o.sketchByFile = true;
    else if (a == "-e" || a == "--no-save") o.noSave = true;
    else if (a == "-d" || a == "--threshold") { o.threshold = atof(need(i)); o.has_threshold = true; fprintf(stderr, "-----set threshold:  %g\n", o.threshold); }
    else if (a == "-o" || a == "--output") { o.outputFile = need(i); o.has_output = true; }
    else if (a == "-i" || a == "--input") { o.inputFile = need(i); o.has_input = true; }
    else if (a == "--presketched") { o.folder_path = need(i); o.has_presketched = true; }
    else if (a == "--append") { o.inputFile = need(i); o.has_append = true; }
    else if (a == "--fast") o.is_fast = true;
#ifdef GREEDY_CLUST
    else if (a == "--save-rep") o.saveRep = true;
    else if (a == "--db") o.repdb_path = need(i);
    else if (a == "--build") o.db_build = true;
    else if (a == "--query") o.db_query = true;
    else if (a == "--assign") o.db_assign = true;
    else if (a == "--stats") o.db_stats = true;
    else if (a == "--top-k") o.topk = atoi(need(i));
#endif
    else if (a == "--drlevel") o.drlevel = atoi(need(i));
    else if (a == "--gpus") o.gpus = need(i);
#ifndef GREEDY_CLUST
    else if (a == "--dense") o.dense = true;
    else if (a == "--newick-tree") o.newick = true;
    else if (a == "--phylip-tree") o.phylip = true;
    else if (a == "--nexus-tree") o.nexus = true;
    else if (a == "--linkage-matrix") o.linkage = true;
#endif
    else if (a == "--inverted-index") { /* on by default, as in the reference (src/main.cpp:104,129) */ }
    else if (a.rfind("--inverted-index=", 0) == 0) {  // the flag's CLI11 form with a value: =false / =0 switches the index path off
      const string v = a.substr(17);
      o.useIndex = !(v == "false" || v == "0" || v == "off" || v == "no");
    }
#ifndef GREEDY_CLUST
    else if (a == "--premsted") { o.folder_path = need(i); o.has_premsted = true; }
#endif
    else if (a == "-h" || a == "--help") {
#ifdef GREEDY_CLUST
      puts("clust-greedy (MI355X build): greedy incremental clustering module");
#else
      puts("clust-mst (MI355X build): minimum-spanning-tree-based module");
#endif
      puts("  -t,--threads N  -m,--min-length N  -c,--containment N  -k,--kmer-size N  -s,--sketch-size N\n"
           "  -l,--list  -e,--no-save  -d,--threshold X  -o,--output FILE  -i,--input FILE\n"
           "  --presketched DIR  --fast  --drlevel N  --gpus all|N|i,j,.. (default all visible MI355X)\n"
           "  --inverted-index=false (MinHash: the dense loops modifyMST / greedyCluster with MinHash::distance())"
#ifndef GREEDY_CLUST
           "  --premsted DIR  --append LIST (with --presketched/--premsted DIR)"
#else
           "  --append LIST (with --presketched DIR)  --save-rep (cluster_state.bin beside the sketches)\n"
           "  [--fast] --db FILE --build|--query|--assign|--append LIST|--stats [--top-k N] (representative database)"
#endif
      );
      exit(0);
    }
    else if (
#ifndef GREEDY_CLUST
             a == "--db" || a == "--build" || a == "--query" || a == "--assign" || a == "--stats" || a == "--top-k" || a == "--save-rep" ||
#else
             a == "--dense" ||
#endif
              a == "--newick-tree" || a == "--phylip-tree" ||
             a == "--nexus-tree" || a == "--linkage-matrix" || a == "--auto-threshold" || a == "--stability" ||
             a == "--dedup-dist" || a == "--reps-per-cluster" || a == "--buildDB")
      unsupported(a.c_str());
    else { fprintf(stderr, "ERROR: unknown option %s\n", a.c_str()); exit(1); }
  }
  return o;
}

[[maybe_unused]] static void cluster_from_mst(const vector<rtc_edge>& mst, const vector<GenomeInfo>& genomes, bool sketchByFile,
                             const string& outputFile, double threshold) {
  vector<rtc_edge> forest = generate_forest(mst, threshold);
  vector<vector<int>> cl = generate_cluster_with_bfs(forest, (int)genomes.size());
  print_result(cl, genomes, sketchByFile, outputFile, threshold);
  cerr << "-----write the cluster result into: " << outputFile << endl;
  cerr << "-----the cluster number of: " << outputFile << " is: " << cl.size() << endl;
  g_metrics.num("clusters", (double)cl.size());
}

struct Options;
static void write_trees(const Options& o, const vector<GenomeInfo>& genomes, const vector<rtc_edge>& mst, bool byFile);

// --dense noise removal (src/sub_command.cpp:3071-3103): drop the forest edges of low-density nodes, cluster again
[[maybe_unused]] static void remove_noise_and_print(const vector<rtc_edge>& mst, const vector<GenomeInfo>& genomes, bool sketchByFile,
                                                    const string& outputFile, double threshold, const vector<int32_t>& dense, int span) {
  vector<rtc_edge> forest = generate_forest(mst, threshold);
  vector<vector<int>> cl = generate_cluster_with_bfs(forest, (int)genomes.size());
  vector<int> noise = noise_nodes(cl, dense, span, (int)genomes.size(), threshold);
  cerr << "-----the total noiseArr size is: " << noise.size() << endl;
  forest = modify_forest(forest, noise);
  vector<vector<int>> cluster = generate_cluster_with_bfs(forest, (int)genomes.size());
  const string outputFileNew = outputFile + ".removeNoise";
  print_result(cluster, genomes, sketchByFile, outputFileNew);
  cerr << "-----write the cluster without noise into: " << outputFileNew << endl;
  cerr << "-----the cluster number of: " << outputFileNew << " is: " << cluster.size() << endl;
}

[[maybe_unused]] static vector<vector<int>> clusters_from_rep_of(const vector<int32_t>& rep_of) {
  // cluster list in representative-creation order: [rep, members...] (src/greedy.cpp:1355-1367)
  vector<vector<int>> cl; vector<int> cid(rep_of.size(), -1);
  for (size_t i = 0; i < rep_of.size(); i++) if (rep_of[i] == (int32_t)i) { cid[i] = (int)cl.size(); cl.push_back({(int)i}); }
  for (size_t i = 0; i < rep_of.size(); i++) if (rep_of[i] != (int32_t)i) cl[cid[rep_of[i]]].push_back((int)i);
  return cl;
}

// src/sub_command.cpp:3010-3029 (and the same block in the --premsted / --append flows)
[[maybe_unused]] static void write_trees(const Options& o, const vector<GenomeInfo>& genomes, const vector<rtc_edge>& mst, bool byFile) {
  if (o.newick) { const string f = o.outputFile + ".newick.tree"; print_newick_tree(genomes, mst, byFile, f); cerr << "-----write the newick tree into: " << f << endl; }
  if (o.phylip) { const string f = o.outputFile + ".phylip.tree"; print_phylip_tree(genomes, mst, byFile, f); cerr << "-----write the PHYLIP tree into: " << f << endl; }
  if (o.nexus) { const string f = o.outputFile + ".nexus.tree"; print_nexus_tree(genomes, mst, byFile, f); cerr << "-----write the NEXUS tree into: " << f << endl; }
  if (o.linkage) { const string f = o.outputFile + ".linkage.txt"; print_linkage_matrix((int)genomes.size(), mst, f); cerr << "-----write the linkage matrix into: " << f << endl; }
}

#ifndef GREEDY_CLUST
// append_clust_mst / append_clust_mst_fast (src/sub_command.cpp:1532-1759): sketch the new genomes with the
// stored folder's parameters, evaluate only the pairs that involve a new genome (rows >= start_index,
// src/MST.cpp:1375-1383 -- rtc_mst_append), merge that forest with the stored MST (sort + Kruskal,
// :1693-1700), cut, print, and write the combined folder.
static int append_clust_mst(const Options& o, vector<Gpu>& gpus) {
  rtc_ctx* ctx = gpus[0].ctx;
  vector<GenomeInfo> genomes; MinHashSketchFile mh; KssdSketchFile ks; bool byFile = true;
  double t0 = get_sec();
  if (o.is_fast) { if (!load_kssd_sketches(o.folder_path, genomes, ks, byFile)) return 1; }
  else if (!load_minhash_sketches(o.folder_path, genomes, mh, byFile)) return 1;
  vector<rtc_edge> pre_mst;
  if (!load_mst(o.folder_path, pre_mst)) return 1;
  const size_t n_pre = genomes.size();
  if (byFile != o.sketchByFile) {
    cerr << "Warning: append_clust_mst(), the input format of append genomes and pre-sketched genome is not same (single input genome vs. genome list)" << endl;
    cerr << "the output cluster file may not have the genome file name" << endl;
  }
  if (!o.sketchByFile) unsupported("single-FASTA input (run with -l and a genome list)");
  cerr << "-----use the same sketch parameters with pre-generated sketches" << endl;
  SketchJob job;
  job.kssd = o.is_fast; job.minLen = o.minLen; job.threads = o.threads;
  if (o.is_fast) {
    job.kmerSize = ks.info.half_k * 2; job.drlevel = ks.info.drlevel;
    cerr << "---use the KSSD sketches" << endl << "---the half_k is: " << ks.info.half_k << endl
         << "---the half_subk is: " << ks.info.half_subk << endl << "---the drlevel is: " << ks.info.drlevel << endl;
  } else {
    job.kmerSize = mh.kmerSize; job.sketchSize = mh.sketchSize; job.isContainment = mh.isContainment; job.containCompress = mh.containCompress;
    cerr << "---the kmer size is: " << mh.kmerSize << endl;
    if (mh.isContainment) cerr << "---use the AAF distance (variable-sketch-size), the sketch size is in proportion with 1/" << mh.containCompress << endl;
    else cerr << "---use the Mash distance (fixed-sketch-size), the sketch size is: " << mh.sketchSize << endl;
  }
  cerr << "---the thread number is: " << o.threads << endl << "---the threshold is: " << o.threshold << endl;
  vector<GenomeInfo> add; MinHashSketchFile mh2; KssdSketchFile ks2; Resident rs2;
  sketch_files(gpus, o.inputFile, job, add, &mh2, &ks2, rs2, true);
  for (size_t i = 0; i < add.size(); i++) {
    add[i].id = (int)(n_pre + i);
    genomes.push_back(add[i]);
    if (!o.is_fast) mh.hashes.push_back(std::move(mh2.hashes[i]));
    else if (ks.use64) ks.h64.push_back(std::move(ks2.h64[i]));
    else ks.h32.push_back(std::move(ks2.h32[i]));
  }
  ks.info.genomeNumber = (int)genomes.size();
  cerr << "-----the size of sketches (number of genomes or sequences) is: " << genomes.size() << endl;
  cerr << "========time of computing sketch is: " << get_sec() - t0 << "========" << endl;
  const string new_folder = current_date_time();
  if (!o.noSave) {
    string command = "mkdir -p " + new_folder;
    if (system(command.c_str()) != 0) { cerr << "ERROR: cannot create " << new_folder << endl; return 1; }
    if (o.is_fast) { save_kssd_sketches(genomes, ks, new_folder, true); save_kssd_index(ks, new_folder); }
    else { save_minhash_sketches(genomes, mh, new_folder, true); save_minhash_index(mh, new_folder); }
  }
  double t2 = get_sec();
  DeviceSketches ds;
  if (o.is_fast) upload_sketches(ctx, ks.use64 ? &ks.h64 : nullptr, ks.use64 ? nullptr : &ks.h32, ds);
  else upload_sketches(ctx, &mh.hashes, nullptr, ds);
  const int kmer_size = o.is_fast ? ks.info.half_k * 2 : mh.kmerSize;
  const int is_containment = o.is_fast ? (int)o.isContainment : (int)mh.isContainment;
  vector<rtc_edge> append_mst(genomes.size());
  uint64_t ne = 0;
  cerr << "---the start_index is: " << n_pre << endl;
  if (!o.useIndex && !o.is_fast)  // the fallback of append_clust_mst (src/sub_command.cpp:1680): modifyMST from start_index
    CHECK(ctx, rtc_mst_mash(ctx, ds.d_hashes, ds.width, ds.d_start, ds.d_len, ds.n, (uint32_t)n_pre, kmer_size, is_containment, (uint32_t)mh.sketchSize,
                            append_mst.data(), &ne, 0, nullptr, nullptr));
  else
    CHECK(ctx, rtc_mst_append(ctx, ds.d_hashes, ds.width, ds.d_start, ds.d_len, ds.n, (uint32_t)n_pre, kmer_size, is_containment, o.threshold,
                              append_mst.data(), &ne));
  append_mst.resize(ne);
  cerr << "========time of generateMST is: " << get_sec() - t2 << "========" << endl;
  vector<rtc_edge> final_graph(pre_mst);
  final_graph.insert(final_graph.end(), append_mst.begin(), append_mst.end());
  std::sort(final_graph.begin(), final_graph.end(), [](const rtc_edge& a, const rtc_edge& b) { return a.dist < b.dist; });  // cmpEdge
  vector<rtc_edge> final_mst = kruskal_algorithm(final_graph, (int)genomes.size());
  write_trees(o, genomes, final_mst, byFile);
  cluster_from_mst(final_mst, genomes, byFile, o.outputFile, o.threshold);
  if (!o.noSave) { save_genome_info(genomes, new_folder, "mst", true, o.is_fast); save_mst(final_mst, new_folder); }
  return 0;
}
#endif

#ifdef GREEDY_CLUST
// calculate_mash_distance over a known intersection (src/greedy.cpp:103-160)
static double kssd_mash_distance(int cm, int sizeRef, int sizeQry, int kmer_size) {
  const uint64_t uni = (uint64_t)sizeRef + (uint64_t)sizeQry - (uint64_t)cm;
  const double jac = uni == 0 ? 0.0 : (double)cm / (double)uni;
  double dist = 0.0;
  if (jac != 1.0) { dist = -log(2 * jac / (1.0 + jac)) / (double)kmer_size; if (dist > 1.0) dist = 1.0; }
  return dist;
}

// minhash_mash_distance (src/greedy.cpp:2771-2787)
static double minhash_mash_distance(int common, int sizeQry, int sizeRef, int kmer_size, bool is_containment) {
  if (common <= 0) return 1.0;
  double jaccard;
  if (is_containment) {
    const int minSize = std::min(sizeQry, sizeRef);
    if (minSize == 0) return 1.0;
    jaccard = (double)common / minSize;
  } else {
    const int denom = sizeQry + sizeRef - common;
    if (denom == 0) return 0.0;
    jaccard = (double)common / denom;
  }
  if (jaccard >= 1.0) return 0.0;
  if (jaccard <= 0.0) return 1.0;
  const double dist = -log(2.0 * jaccard / (1.0 + jaccard)) / kmer_size;
  return dist > 1.0 ? 1.0 : dist;
}

// one (new genome or query, representative) pair with cm > 0 shared hashes: is it a candidate, and at which distance?
// KSSD: size-ratio and minimum-common filters, then calculate_mash_distance (src/greedy.cpp:1815-1843, :2583-2606).
// MinHash, incremental step: minimum-common filter only (:2040-2076); MinHash query: no filter (:3003-3012).
static bool rep_candidate(const KssdClusterState& st, bool query, int cm, int sizeQry, int sizeRef, double radio, double jaccard_min,
                          double& dist) {
  if (!st.minhash) {
    const double ratio = (double)sizeQry / sizeRef;
    if (ratio > radio || ratio < 1.0 / radio) return false;
    const int min_common_needed = (int)(jaccard_min * (sizeQry + sizeRef) / (1.0 + jaccard_min));
    if (cm < min_common_needed) return false;
    dist = kssd_mash_distance(cm, sizeRef, sizeQry, st.kmer_size);
    return true;
  }
  if (!query) {
    if (sizeRef == 0) return false;
    const int min_common_needed = st.is_containment ? (int)(jaccard_min * std::min(sizeQry, sizeRef))
                                                    : (int)(jaccard_min * (sizeQry + sizeRef) / (1.0 + jaccard_min));
    if (cm < min_common_needed) return false;
    if (!st.is_containment && sizeQry + sizeRef - cm == 0) return false;
  }
  dist = minhash_mash_distance(cm, sizeQry, sizeRef, st.kmer_size, st.is_containment);
  return true;
}

static size_t sketch_count(const KssdSketchFile& f) { return f.use64 ? f.h64.size() : f.h32.size(); }
static size_t sketch_len(const KssdSketchFile& f, size_t i) { return f.use64 ? f.h64[i].size() : f.h32[i].size(); }
static void push_sketch(KssdSketchFile& dst, const KssdSketchFile& src, size_t i) {
  if (dst.use64) dst.h64.push_back(src.h64[i]); else dst.h32.push_back(src.h32[i]);
}

// the representatives' sketches followed by `tail`'s: the set one rectangular intersection launch works on
static void reps_then(const KssdClusterState& st, const KssdSketchFile& tail, KssdSketchFile& work) {
  work = st.reps;
  if (sketch_count(work) == 0) work.use64 = tail.use64;
  for (size_t i = 0; i < sketch_count(tail); i++) push_sketch(work, tail, i);
}

// KssdIncrementalCluster (src/greedy.cpp:1736-1900): every new genome, in order, joins the representative at the
// smallest Mash distance <= threshold among those that share a hash with it and pass the size-ratio and
// minimum-common filters, or becomes a representative itself.  The GPU supplies |A ∩ B| of each new genome against
// the stored representatives and the new genomes before it; the rule runs on the host.  Ties in distance go to the
// earliest representative (the reference's choice depends on hash-map iteration and thread order).  As in the
// reference (:1861-1864) a genome that opens a cluster is recorded as its representative but is not listed among
// the cluster's members -- the cluster starts empty.
static int kssd_incremental_cluster(rtc_ctx* ctx, KssdClusterState& st, const vector<GenomeInfo>& add, const KssdSketchFile& ks2,
                                    bool keep_sketches) {
  const size_t R0 = st.rep_ids.size(), m = add.size(), n_old = st.genomes.size(), n_work = R0 + m;
  const double threshold = st.threshold;
  cerr << "Existing clusters: " << R0 << endl << "New genomes: " << m << endl;
  if (st.minhash && m == 0) { cerr << "ERROR: No new sketches to process" << endl; return 0; }       // src/greedy.cpp:1973-1981
  if (st.minhash && R0 == 0) { cerr << "ERROR: No existing representatives" << endl; return 0; }
  if (m == 0) return 0;
  if (R0 && ks2.use64 != st.reps.use64) { cerr << "ERROR: appended sketches and stored sketches differ in hash width" << endl; return 1; }
  KssdSketchFile work;
  reps_then(st, ks2, work);
  st.reps.use64 = work.use64; if (keep_sketches && n_old == 0) st.sk.use64 = work.use64;
  DeviceSketches ds;
  upload_sketches(ctx, work.use64 ? &work.h64 : nullptr, work.use64 ? nullptr : &work.h32, ds);
  const double radio = 2.0 * exp(threshold * st.kmer_size) - 1.0;                 // calculateMaxSizeRatio, src/greedy.cpp:162-173
  const double x = exp(-threshold * st.kmer_size), jaccard_min = x / (2.0 - x);   // :1751-1753
  const size_t max_block_bytes = (size_t)256 << 20;
  const size_t B = std::max<size_t>(1, std::min<size_t>(m, max_block_bytes / (n_work * 4)));
  uint32_t* d_common = nullptr;
  CHECK(ctx, rtc_dev_alloc(ctx, B * n_work * 4 + 64, (void**)&d_common));
  vector<uint32_t> common(B * n_work);
  vector<int> rep_of_col(n_work, -1);  // column of `work` -> representative index (new genomes: once they open a cluster)
  for (size_t r = 0; r < R0; r++) rep_of_col[r] = (int)r;
  int new_clusters = 0, assigned = 0;
  for (size_t r0 = R0; r0 < n_work; r0 += B) {
    const size_t r1 = std::min(n_work, r0 + B);
    CHECK(ctx, rtc_pair_common_dev(ctx, ds.d_hashes, ds.width, ds.d_start, ds.d_len, (uint32_t)n_work, (uint32_t)r0, (uint32_t)r1, 0,
                                   (uint32_t)n_work, d_common, (uint64_t)n_work, 1, 0));
    CHECK(ctx, rtc_copy_d2h(ctx, common.data(), d_common, (r1 - r0) * n_work * 4));
    for (size_t q = r0; q < r1; q++) {
      const uint32_t* row = common.data() + (q - r0) * n_work;
      const int genome_idx = (int)(n_old + (q - R0));
      const int sizeQry = (int)sketch_len(work, q);
      double best_dist = std::numeric_limits<double>::max();
      int best = -1;
      for (size_t c = 0; c < q; c++) {
        if (rep_of_col[c] < 0 || row[c] == 0) continue;  // candidates: representatives sharing a hash (:1768-1790)
        double dist;
        if (!rep_candidate(st, false, (int)row[c], sizeQry, (int)sketch_len(work, c), radio, jaccard_min, dist)) continue;
        if (dist <= threshold && dist < best_dist) { best_dist = dist; best = rep_of_col[c]; }
      }
      if (best >= 0) { st.clusters[best].push_back(genome_idx); assigned++; }
      else {
        rep_of_col[q] = (int)st.rep_ids.size();
        st.rep_ids.push_back(genome_idx);
        st.rep_genomes.push_back(add[q - R0]);
        push_sketch(st.reps, work, q);
        st.clusters.push_back(vector<int>());
        new_clusters++;
      }
    }
  }
  for (size_t i = 0; i < m; i++) {
    st.genomes.push_back(add[i]);
    if (keep_sketches) push_sketch(st.sk, ks2, i);
  }
  st.info.genomeNumber = (int)st.genomes.size();
  CHECK(ctx, rtc_dev_free(ctx, d_common));
  CHECK(ctx, rtc_dev_free(ctx, ds.d_hashes)); CHECK(ctx, rtc_dev_free(ctx, ds.d_start)); CHECK(ctx, rtc_dev_free(ctx, ds.d_len));
  cerr << "===== Incremental Clustering Results =====" << endl << "Assigned to existing clusters: " << assigned << endl
       << "New clusters created: " << new_clusters << endl << "Total clusters now: " << st.clusters.size() << endl;
  return 0;
}

// the representative part of a state, from clusters whose first member is the representative (src/greedy.cpp:924-934)
static void fill_cluster_state(KssdClusterState& st, double threshold, int kmer_size, const KssdParameters& info,
                               const vector<GenomeInfo>& genomes, const KssdSketchFile& sk, const vector<vector<int>>& cluster) {
  st.threshold = threshold; st.kmer_size = kmer_size; st.info = info; st.genomes = genomes; st.sk = sk; st.clusters = cluster;
  st.rep_ids.clear(); st.rep_genomes.clear();
  st.reps = KssdSketchFile(); st.reps.info = info; st.reps.use64 = sk.use64;
  for (const auto& c : cluster) if (!c.empty()) {
    st.rep_ids.push_back(c[0]);
    st.rep_genomes.push_back(genomes[c[0]]);
    push_sketch(st.reps, sk, c[0]);
  }
}

// KssdInitialClusterWithState (src/greedy.cpp:900-958): the stored sketches sorted by hash count, descending
// (comparator without tie-break, :594-597), clustered as clust-greedy --fast does, kept as a state
static int kssd_initial_state(rtc_ctx* ctx, vector<GenomeInfo>& pre, KssdSketchFile& ks, double threshold, KssdClusterState& st) {
  const size_t n_pre = pre.size();
  struct Item { size_t idx; size_t c; };
  vector<Item> items(n_pre);
  for (size_t i = 0; i < n_pre; i++) items[i] = Item{i, sketch_len(ks, i)};
  std::sort(items.begin(), items.end(), [](const Item& a, const Item& b) { return a.c > b.c; });
  vector<GenomeInfo> genomes; KssdSketchFile all; all.info = ks.info; all.use64 = ks.use64;
  for (const Item& it : items) {
    genomes.push_back(pre[it.idx]);
    if (ks.use64) all.h64.push_back(std::move(ks.h64[it.idx])); else all.h32.push_back(std::move(ks.h32[it.idx]));
  }
  const int kmer_size = ks.info.half_k * 2;
  vector<vector<int>> cluster;
  if (n_pre) {
    DeviceSketches ds;
    upload_sketches(ctx, all.use64 ? &all.h64 : nullptr, all.use64 ? nullptr : &all.h32, ds);
    vector<int32_t> rep_of(n_pre, -1);
    uint32_t ncl = 0;
    CHECK(ctx, rtc_greedy(ctx, ds.d_hashes, ds.width, ds.d_start, ds.d_len, (uint32_t)n_pre, nullptr, kmer_size, 0, 1, threshold,
                          rep_of.data(), &ncl));
    CHECK(ctx, rtc_dev_free(ctx, ds.d_hashes)); CHECK(ctx, rtc_dev_free(ctx, ds.d_start)); CHECK(ctx, rtc_dev_free(ctx, ds.d_len));
    cluster = clusters_from_rep_of(rep_of);
  }
  fill_cluster_state(st, threshold, kmer_size, ks.info, genomes, all, cluster);
  return 0;
}

// compute_kssd_sketches with a state's parameters: the genomes of `list` on the GPU sketcher
static void sketch_for_state(vector<Gpu>& gpus, const Options& o, const string& list, int kmer_size, int drlevel,
                             vector<GenomeInfo>& add, KssdSketchFile& ks2) {
  SketchJob job;
  job.kssd = true; job.kmerSize = kmer_size; job.drlevel = drlevel; job.minLen = o.minLen; job.threads = o.threads;
  MinHashSketchFile mh2; Resident rs2;
  sketch_files(gpus, list, job, add, &mh2, &ks2, rs2, true);
}

// compute_sketches with a MinHash RepDB's parameters (contain_compress is the literal 1000 of mh_repdb_query / _assign /
// _append, src/sub_command.cpp:609,657,717); the sketches come back in the u64 arm of a KssdSketchFile
static void sketch_for_mh_state(vector<Gpu>& gpus, const Options& o, const string& list, const KssdClusterState& st,
                                vector<GenomeInfo>& add, KssdSketchFile& out) {
  SketchJob job;
  job.kssd = false; job.kmerSize = st.kmer_size; job.sketchSize = st.sketch_size; job.isContainment = st.is_containment;
  job.containCompress = 1000; job.minLen = o.minLen; job.threads = o.threads;
  MinHashSketchFile mh2; KssdSketchFile unused; Resident rs2;
  sketch_files(gpus, list, job, add, &mh2, &unused, rs2, true);
  out = KssdSketchFile(); out.use64 = true; out.h64 = std::move(mh2.hashes);
}

static int save_sketch_folder(const vector<GenomeInfo>& g, const KssdSketchFile& ks, string& folder) {
  folder = current_date_time();
  string command = "mkdir -p " + folder;
  if (system(command.c_str()) != 0) { cerr << "ERROR: cannot create " << folder << endl; return 1; }
  save_kssd_sketches(g, ks, folder, true);
  return 0;
}

// append_clust_greedy_fast (src/sub_command.cpp:192-270).  With DIR/cluster_state.bin ("Incremental Update Mode"):
// the stored state's sketches, clusters, threshold and k; the new genomes are clustered against its
// representatives.  Without it ("Initial State Building Mode"): the stored KSSD sketches are clustered as
// clust-greedy --fast would (KssdInitialClusterWithState), then the new genomes as above.  --save-rep (and no -e)
// writes the state back.
static int append_clust_greedy_fast(const Options& o, vector<Gpu>& gpus) {
  rtc_ctx* ctx = gpus[0].ctx;
  const string state_file = o.folder_path + "/cluster_state.bin";
  double t0 = get_sec();
  if (!o.sketchByFile) unsupported("single-FASTA input (run with -l and a genome list)");
  KssdClusterState st;
  struct stat sb;
  bool has_state = stat(state_file.c_str(), &sb) == 0;
  bool byFile = true;
  if (has_state) {
    cerr << "===== Incremental Update Mode (KSSD) =====" << endl << "Found existing cluster state, loading..." << endl;
    has_state = load_kssd_cluster_state(state_file, st);
  }
  if (has_state) {
    cerr << "---the threshold is: " << o.threshold << endl << "---the thread number is: " << o.threads << endl;
  } else {
    vector<GenomeInfo> pre; KssdSketchFile ks;
    if (!load_kssd_sketches(o.folder_path, pre, ks, byFile)) return 1;
    if (byFile != o.sketchByFile) cerr << "Warning: the input format of append genomes and pre-sketched genome is not same" << endl;
    cerr << "===== Initial State Building Mode (KSSD) =====" << endl;
    cerr << "No existing state found, building state from pre-sketched genomes..." << endl;
    cerr << "-----use the same sketch parameters with pre-generated sketches" << endl << "---use the KSSD sketches" << endl
         << "---the half_k is: " << ks.info.half_k << endl << "---the half_subk is: " << ks.info.half_subk << endl
         << "---the drlevel is: " << ks.info.drlevel << endl << "---the threshold is: " << o.threshold << endl;
    if (kssd_initial_state(ctx, pre, ks, o.threshold, st) != 0) return 1;
  }
  // ---- the new genomes, with the stored parameters (KssdIncrementalCluster works with the state's threshold and k) ----
  vector<GenomeInfo> add; KssdSketchFile ks2;
  sketch_for_state(gpus, o, o.inputFile, st.kmer_size, st.info.drlevel, add, ks2);
  cerr << "New genomes sketched: " << add.size() << endl;
  cerr << "========time of computing sketch is: " << get_sec() - t0 << "========" << endl;
  if (!o.noSave) {  // compute_kssd_sketches(isSave): the appended sketches get a folder of their own
    string folder;
    if (save_sketch_folder(add, ks2, folder) != 0) return 1;
  }
  double t2 = get_sec();
  if (kssd_incremental_cluster(ctx, st, add, ks2, true) != 0) return 1;
  if (!o.noSave && o.saveRep) {
    if (!save_kssd_cluster_state(state_file, st)) return 1;
    cerr << "-----saved cluster state (with inverted index) for future incremental updates" << endl;
  }
  print_result(st.clusters, st.genomes, byFile, o.outputFile);
  cerr << "-----write the cluster result into: " << o.outputFile << endl;
  cerr << "-----the cluster number of " << o.outputFile << " is: " << st.clusters.size() << endl;
  cerr << "========time of greedyCluster is: " << get_sec() - t2 << "========" << endl;
  return 0;
}

// ---- RepDB of clust-greedy --fast --db (src/sub_command.cpp:276-475, src/main.cpp:300-334) ----
static void repdb_build_summary(size_t total, size_t reps, const string& db) {
  cerr << "\n===== RepDB Build Summary =====" << endl << "  Total genomes:    " << total << endl << "  Representatives:  " << reps << endl
       << "  Compression:      " << std::fixed << std::setprecision(2) << (1.0 - (double)reps / total) * 100.0 << "%" << endl
       << "  RepDB saved to:   " << db << endl << "===============================" << endl;
}

// repdb_build_from_sketch / repdb_build_from_genome (src/sub_command.cpp:278-334)
static int repdb_build(const Options& o, vector<Gpu>& gpus) {
  rtc_ctx* ctx = gpus[0].ctx;
  vector<GenomeInfo> pre; KssdSketchFile ks; bool byFile = true;
  if (o.has_presketched) {
    if (!load_kssd_sketches(o.folder_path, pre, ks, byFile)) return 1;
    cerr << "===== RepDB Build (from pre-sketched) =====" << endl;
  } else {
    if (!o.sketchByFile) unsupported("single-FASTA input (run with -l and a genome list)");
    int kmerSize = o.kmerSize;
    if (!o.isSetKmer) { kmerSize = 19; cerr << "-----use default kmerSize: " << kmerSize << endl; }
    sketch_for_state(gpus, o, o.inputFile, kmerSize, o.drlevel, pre, ks);
    string folder;
    if (save_sketch_folder(pre, ks, folder) != 0) return 1;   // compute_kssd_sketches(isSave = true)
    cerr << "===== RepDB Build (from genomes) =====" << endl;
  }
  cerr << "  Genomes:    " << pre.size() << endl << "  Threshold:  " << o.threshold << endl << "  Kmer size:  " << ks.info.half_k * 2 << endl;
  if (pre.empty()) { cerr << "ERROR: no genome to cluster" << endl; return 1; }
  KssdClusterState st;
  if (kssd_initial_state(ctx, pre, ks, o.threshold, st) != 0) return 1;
  if (!save_kssd_repdb(o.repdb_path, st)) return 1;
  if (!o.outputFile.empty()) {
    print_result(st.clusters, st.genomes, byFile, o.outputFile, o.threshold);
    cerr << "-----write the cluster result into: " << o.outputFile << endl;
  }
  repdb_build_summary(st.genomes.size(), st.rep_ids.size(), o.repdb_path);
  return 0;
}

struct RepHit { int rep_idx; double distance; };

// KssdClusterState::query_topk for every query (src/greedy.cpp:2539-2637): the representatives that share a hash
// with the query and pass the size-ratio and minimum-common filters, ordered by Mash distance (ties: earlier
// representative first -- the reference's std::sort leaves them in hash-map order), the first `topk` kept.  One
// rectangular intersection launch (rows = queries, columns = representatives) stands in for the index walk.
static int repdb_query_topk(rtc_ctx* ctx, const KssdClusterState& st, const KssdSketchFile& qs, int topk, vector<vector<RepHit>>& out) {
  const size_t R = st.rep_ids.size(), Q = sketch_count(qs);
  out.assign(Q, vector<RepHit>());
  if (R == 0 || Q == 0) return 0;
  if (qs.use64 != st.reps.use64) { cerr << "ERROR: query sketches and the RepDB differ in hash width" << endl; return 1; }
  KssdSketchFile work;
  reps_then(st, qs, work);
  DeviceSketches ds;
  upload_sketches(ctx, work.use64 ? &work.h64 : nullptr, work.use64 ? nullptr : &work.h32, ds);
  const double radio = 2.0 * exp(st.threshold * st.kmer_size) - 1.0;
  const double x = exp(-st.threshold * st.kmer_size), jaccard_min = x / (2.0 - x);
  const size_t B = std::max<size_t>(1, std::min<size_t>(Q, ((size_t)256 << 20) / (R * 4)));
  uint32_t* d_common = nullptr;
  CHECK(ctx, rtc_dev_alloc(ctx, B * R * 4 + 64, (void**)&d_common));
  vector<uint32_t> common(B * R);
  for (size_t q0 = 0; q0 < Q; q0 += B) {
    const size_t q1 = std::min(Q, q0 + B);
    CHECK(ctx, rtc_pair_common_dev(ctx, ds.d_hashes, ds.width, ds.d_start, ds.d_len, (uint32_t)(R + Q), (uint32_t)(R + q0), (uint32_t)(R + q1), 0,
                                   (uint32_t)R, d_common, (uint64_t)R, 0, 0));
    CHECK(ctx, rtc_copy_d2h(ctx, common.data(), d_common, (q1 - q0) * R * 4));
    for (size_t q = q0; q < q1; q++) {
      const uint32_t* row = common.data() + (q - q0) * R;
      const int sizeQry = (int)sketch_len(qs, q);
      vector<RepHit> scored;
      for (size_t r = 0; r < R; r++) {
        if (row[r] == 0) continue;
        double dist;
        if (!rep_candidate(st, true, (int)row[r], sizeQry, (int)sketch_len(st.reps, r), radio, jaccard_min, dist)) continue;
        scored.push_back(RepHit{(int)r, dist});
      }
      std::stable_sort(scored.begin(), scored.end(), [](const RepHit& a, const RepHit& b) { return a.distance < b.distance; });
      if ((int)scored.size() > topk) scored.resize(std::max(topk, 0));
      out[q] = std::move(scored);
    }
  }
  CHECK(ctx, rtc_dev_free(ctx, d_common));
  CHECK(ctx, rtc_dev_free(ctx, ds.d_hashes)); CHECK(ctx, rtc_dev_free(ctx, ds.d_start)); CHECK(ctx, rtc_dev_free(ctx, ds.d_len));
  return 0;
}

static int repdb_load_and_sketch(const Options& o, vector<Gpu>& gpus, KssdClusterState& st, vector<GenomeInfo>& q, KssdSketchFile& qs) {
  if (o.is_fast) { if (!load_kssd_repdb(o.repdb_path, st)) { cerr << "ERROR: Failed to load RepDB from: " << o.repdb_path << endl; return 1; } }
  else if (!load_minhash_repdb(o.repdb_path, st)) { cerr << "ERROR: Failed to load MinHash RepDB from: " << o.repdb_path << endl; return 1; }
  if (!o.sketchByFile) unsupported("single-FASTA input (run with -l and a genome list)");
  if (o.is_fast) sketch_for_state(gpus, o, o.inputFile, st.kmer_size, st.info.drlevel, q, qs);
  else sketch_for_mh_state(gpus, o, o.inputFile, st, q, qs);
  return 0;
}

// mh_repdb_build_from_sketch / mh_repdb_build_from_genome (src/sub_command.cpp:502-586): the sketches in stored /
// list order (no size sort here), MinHashInitialClusterWithState = the MinHash greedy pass, saved as MHREPDB1
static int mh_repdb_build(const Options& o, vector<Gpu>& gpus) {
  rtc_ctx* ctx = gpus[0].ctx;
  vector<GenomeInfo> genomes; MinHashSketchFile mh; bool byFile = true;
  vector<uint32_t> size_cfg;
  if (o.has_presketched) {
    if (!load_minhash_sketches(o.folder_path, genomes, mh, byFile)) return 1;
    if (genomes.empty()) { cerr << "ERROR: no genome to cluster" << endl; return 1; }
    // loadSketches builds MinHash(k, containCompress) in containment mode (src/Sketch_IO.cpp:334): getSketchSize() reports that
    size_cfg.assign(genomes.size(), mh.isContainment ? (uint32_t)mh.containCompress : (uint32_t)mh.sketchSize);
    cerr << "===== MinHash RepDB Build (from pre-sketched) =====" << endl;
  } else {
    if (!o.sketchByFile) unsupported("single-FASTA input (run with -l and a genome list)");
    mh.kmerSize = o.kmerSize; mh.sketchSize = o.sketchSize; mh.isContainment = o.isContainment; mh.containCompress = o.containCompress;
    if (!o.isSetKmer) { mh.kmerSize = 21; cerr << "-----use default kmerSize: " << mh.kmerSize << endl; }
    if (!o.isJaccard) mh.sketchSize = 1000;
    SketchJob job;
    job.kssd = false; job.kmerSize = mh.kmerSize; job.sketchSize = mh.sketchSize; job.isContainment = mh.isContainment;
    job.containCompress = mh.containCompress; job.minLen = o.minLen; job.threads = o.threads;
    KssdSketchFile unused; Resident rs;
    sketch_files(gpus, o.inputFile, job, genomes, &mh, &unused, rs, true);
    cerr << "-----the size of sketches (number of genomes or sequences) is: " << genomes.size() << endl;
    if (genomes.empty()) { cerr << "ERROR: no genome to cluster" << endl; return 1; }
    const string folder = current_date_time();   // compute_sketches(isSave = true)
    string command = "mkdir -p " + folder;
    if (system(command.c_str()) != 0) { cerr << "ERROR: cannot create " << folder << endl; return 1; }
    save_minhash_sketches(genomes, mh, folder, true);
    size_cfg.resize(genomes.size());
    for (size_t i = 0; i < genomes.size(); i++)
      size_cfg[i] = mh.isContainment ? (uint32_t)std::max(file_length_for_containment(genomes[i].fileName) / mh.containCompress, 100)
                                     : (uint32_t)mh.sketchSize;
    cerr << "===== MinHash RepDB Build (from genomes) =====" << endl;
  }
  KssdClusterState st;
  st.minhash = true; st.threshold = o.threshold; st.kmer_size = mh.kmerSize; st.sketch_size = (int)size_cfg[0]; st.is_containment = mh.isContainment;
  cerr << "  Genomes:       " << genomes.size() << endl << "  Threshold:     " << o.threshold << endl << "  Kmer size:     " << st.kmer_size << endl
       << "  Sketch size:   " << st.sketch_size << endl;
  if (o.has_presketched) cerr << "  Containment:   " << (st.is_containment ? "yes" : "no") << endl;
  DeviceSketches ds;
  upload_sketches(ctx, &mh.hashes, nullptr, ds);
  vector<int32_t> rep_of(genomes.size(), -1);
  uint32_t ncl = 0;
  CHECK(ctx, rtc_greedy(ctx, ds.d_hashes, ds.width, ds.d_start, ds.d_len, ds.n, size_cfg.data(), st.kmer_size, (int)st.is_containment, 0,
                        o.threshold, rep_of.data(), &ncl));
  CHECK(ctx, rtc_dev_free(ctx, ds.d_hashes)); CHECK(ctx, rtc_dev_free(ctx, ds.d_start)); CHECK(ctx, rtc_dev_free(ctx, ds.d_len));
  st.clusters = clusters_from_rep_of(rep_of);
  st.genomes = genomes; st.reps.use64 = true;
  for (const auto& c : st.clusters) if (!c.empty()) {
    st.rep_ids.push_back(c[0]);
    st.rep_genomes.push_back(genomes[c[0]]);
    st.reps.h64.push_back(mh.hashes[c[0]]);
  }
  if (!save_minhash_repdb(o.repdb_path, st)) return 1;
  if (!o.outputFile.empty()) {
    print_result(st.clusters, st.genomes, byFile, o.outputFile);
    cerr << "-----write the cluster result into: " << o.outputFile << endl;
  }
  cerr << "\n===== MinHash RepDB Build Summary =====" << endl << "  Total genomes:    " << genomes.size() << endl
       << "  Representatives:  " << st.rep_ids.size() << endl
       << "  Compression:      " << std::fixed << std::setprecision(2) << (1.0 - (double)st.rep_ids.size() / genomes.size()) * 100.0 << "%" << endl
       << "  RepDB saved to:   " << o.repdb_path << endl << "===============================" << endl;
  return 0;
}

// repdb_query (src/sub_command.cpp:337-393)
static int repdb_query(const Options& o, vector<Gpu>& gpus) {
  KssdClusterState st; vector<GenomeInfo> q; KssdSketchFile qs;
  if (repdb_load_and_sketch(o, gpus, st, q, qs) != 0) return 1;
  cerr << (o.is_fast ? "===== RepDB Query =====" : "===== MinHash RepDB Query =====") << endl << "  Query genomes:  " << q.size() << endl << "  Top-k:          " << o.topk << endl
       << "  DB reps:        " << st.rep_ids.size() << endl;
  vector<vector<RepHit>> hits;
  if (repdb_query_topk(gpus[0].ctx, st, qs, o.topk, hits) != 0) return 1;
  FILE* fp = fopen(o.outputFile.c_str(), "w");
  if (!fp) { cerr << "ERROR: Cannot open output file: " << o.outputFile << endl; return 1; }
  fprintf(fp, "#query\trank\trep_name\tdistance\tcluster_id\tcluster_size\n");
  for (size_t i = 0; i < q.size(); i++) {
    string qname = q[i].fileName;
    if (qname.empty()) qname = "query_" + std::to_string(i);
    if (hits[i].empty()) fprintf(fp, "%s\t0\tno_match\t-1\t-1\t0\n", qname.c_str());
    else for (size_t r = 0; r < hits[i].size(); r++) {
      const int ri = hits[i][r].rep_idx;
      fprintf(fp, "%s\t%d\t%s\t%.6f\t%d\t%d\n", qname.c_str(), (int)r + 1, st.rep_genomes[ri].fileName.c_str(), hits[i][r].distance, ri,
              (int)st.clusters[ri].size());
    }
  }
  fclose(fp);
  cerr << "===== Query Results =====" << endl << "  Output: " << o.outputFile << endl << "=========================" << endl;
  return 0;
}

// repdb_assign (src/sub_command.cpp:395-452; KssdClusterState::assign, src/greedy.cpp:2639-2654)
static int repdb_assign(const Options& o, vector<Gpu>& gpus) {
  KssdClusterState st; vector<GenomeInfo> q; KssdSketchFile qs;
  if (repdb_load_and_sketch(o, gpus, st, q, qs) != 0) return 1;
  cerr << (o.is_fast ? "===== RepDB Assignment =====" : "===== MinHash RepDB Assignment =====") << endl << "  Query genomes:  " << q.size() << endl << "  DB reps:        " << st.rep_ids.size() << endl
       << "  Threshold:      " << st.threshold << endl;
  vector<vector<RepHit>> hits;
  if (repdb_query_topk(gpus[0].ctx, st, qs, 1, hits) != 0) return 1;
  FILE* fp = fopen(o.outputFile.c_str(), "w");
  if (!fp) { cerr << "ERROR: Cannot open output file: " << o.outputFile << endl; return 1; }
  fprintf(fp, "#query\tassigned_cluster\trep_name\tdistance\tcluster_size\tstatus\n");
  int assigned = 0, unassigned = 0;
  for (size_t i = 0; i < q.size(); i++) {
    string qname = q[i].fileName;
    if (qname.empty()) qname = "query_" + std::to_string(i);
    if (!hits[i].empty() && hits[i][0].distance <= st.threshold) {
      const int ri = hits[i][0].rep_idx;
      fprintf(fp, "%s\t%d\t%s\t%.6f\t%d\tassigned\n", qname.c_str(), ri, st.rep_genomes[ri].fileName.c_str(), hits[i][0].distance,
              (int)st.clusters[ri].size());
      assigned++;
    } else {
      fprintf(fp, "%s\t-1\tunassigned\t-1\t0\tnovel\n", qname.c_str());
      unassigned++;
    }
  }
  fclose(fp);
  cerr << "===== Assignment Results =====" << endl
       << "  Assigned:    " << assigned << " (" << std::fixed << std::setprecision(1) << (100.0 * assigned / q.size()) << "%)" << endl
       << "  Novel:       " << unassigned << " (" << std::fixed << std::setprecision(1) << (100.0 * unassigned / q.size()) << "%)" << endl
       << "  Output:      " << o.outputFile << endl << "==============================" << endl;
  return 0;
}

// repdb_append (src/sub_command.cpp:454-500)
static int repdb_append(const Options& o, vector<Gpu>& gpus) {
  KssdClusterState st; vector<GenomeInfo> add; KssdSketchFile ks2;
  if (repdb_load_and_sketch(o, gpus, st, add, ks2) != 0) return 1;
  const size_t old_reps = st.rep_ids.size(), old_total = st.genomes.size();
  cerr << (o.is_fast ? "===== RepDB Append =====" : "===== MinHash RepDB Append =====") << endl << "  Existing reps:    " << old_reps << endl << "  Existing genomes: " << old_total << endl
       << "  New genomes:      " << add.size() << endl;
  if (kssd_incremental_cluster(gpus[0].ctx, st, add, ks2, false) != 0) return 1;
  if (!(st.minhash ? save_minhash_repdb(o.repdb_path, st) : save_kssd_repdb(o.repdb_path, st))) return 1;
  if (!o.outputFile.empty()) {   // printKssdResult / printRepDBClusterResult (src/sub_command.cpp:672-702): the same layout
    print_result(st.clusters, st.genomes, true, o.outputFile, st.threshold);
    cerr << "-----write the cluster result into: " << o.outputFile << endl;
  }
  cerr << "\n===== Append Summary =====" << endl << "  New reps added:   " << st.rep_ids.size() - old_reps << endl
       << "  Total reps now:   " << st.rep_ids.size() << endl << "  Total genomes:    " << st.genomes.size() << endl
       << "  RepDB updated:    " << o.repdb_path << endl << "==========================" << endl;
  return 0;
}
// append_clust_greedy (src/sub_command.cpp:23-190): `clust-greedy --append LIST --presketched DIR` on MinHash sketches.
// Without a stored state the old and the new sketches are put together, sorted by genome size (cmpGenomeSize) and
// clustered by the legacy greedyCluster -- every representative, MinHash::distance() (rtc_greedy_mash).  With a
// cluster_state.bin the representatives are taken from the folder's sketches by the stored ids (the reference
// re-reads the folder, :100-139) and the new genomes go through MinHashIncrementalCluster.
static int append_clust_greedy(const Options& o, vector<Gpu>& gpus) {
  rtc_ctx* ctx = gpus[0].ctx;
  const string state_file = o.folder_path + "/cluster_state.bin";
  if (!o.sketchByFile) unsupported("single-FASTA input (run with -l and a genome list)");
  KssdClusterState st;
  struct stat sb;
  bool has_state = stat(state_file.c_str(), &sb) == 0;
  if (has_state) {
    cerr << "===== Incremental Update Mode (MinHash) =====" << endl << "Found existing cluster state, loading..." << endl;
    has_state = load_minhash_cluster_state(state_file, st);
  }
  vector<GenomeInfo> pre; MinHashSketchFile mh; bool byFile = true;
  if (!load_minhash_sketches(o.folder_path, pre, mh, byFile)) return 1;
  auto sketch_new = [&](int kmer, int sketch_size, bool containment, vector<GenomeInfo>& add, MinHashSketchFile& mh2) {
    SketchJob job;
    job.kssd = false; job.kmerSize = kmer; job.sketchSize = sketch_size; job.isContainment = containment;
    job.containCompress = mh.containCompress; job.minLen = o.minLen; job.threads = o.threads;
    KssdSketchFile unused; Resident rs2;
    sketch_files(gpus, o.inputFile, job, add, &mh2, &unused, rs2, true);
  };
  if (!has_state) {
    if (byFile != o.sketchByFile) {
      cerr << "ERROR: append_clust_greedy(), the input format of append genomes and pre-sketched genome is not same (single input genome vs. genome list)" << endl;
      return 1;
    }
    cerr << "-----use the same sketch parameters with pre-generated sketches" << endl << "---the kmer size is: " << mh.kmerSize << endl;
    if (mh.isContainment) cerr << "---use the AAF distance (variable-sketch-size), the sketch size is in proportion with 1/" << mh.containCompress << endl;
    else cerr << "---use the Mash distance (fixed-sketch-size), the sketch size is: " << mh.sketchSize << endl;
    cerr << "---the thread number is: " << o.threads << endl << "---the threshold is: " << o.threshold << endl;
    vector<GenomeInfo> add; MinHashSketchFile mh2;
    sketch_new(mh.kmerSize, mh.sketchSize, mh.isContainment, add, mh2);
    cerr << "-----the size of sketches (number of genomes or sequences) is: " << add.size() << endl;
    // final_sketches = pre + append, sorted by genome size (length descending, id ascending; ids restart in the appended part)
    vector<GenomeInfo> all(pre);
    all.insert(all.end(), add.begin(), add.end());
    vector<size_t> perm(all.size());
    iota(perm.begin(), perm.end(), 0);
    std::stable_sort(perm.begin(), perm.end(), [&](size_t a, size_t b) {
      if (all[a].totalSeqLength != all[b].totalSeqLength) return all[a].totalSeqLength > all[b].totalSeqLength;
      return all[a].id < all[b].id;
    });
    vector<GenomeInfo> genomes; MinHashSketchFile fin = mh; fin.hashes.clear();
    for (size_t q : perm) {
      genomes.push_back(all[q]);
      fin.hashes.push_back(q < pre.size() ? mh.hashes[q] : mh2.hashes[q - pre.size()]);
    }
    if (!o.noSave) {
      const string folder = current_date_time();
      string command = "mkdir -p " + folder;
      if (system(command.c_str()) != 0) { cerr << "ERROR: cannot create " << folder << endl; return 1; }
      save_minhash_sketches(genomes, fin, folder, byFile);
    }
    if (genomes.empty()) { cerr << "ERROR: no genome to cluster" << endl; return 1; }
    DeviceSketches ds;
    upload_sketches(ctx, &fin.hashes, nullptr, ds);
    vector<int32_t> rep_of(genomes.size());
    uint32_t ncl = 0;
    // (sketches loaded from a containment folder report containCompress as their sketch size, src/Sketch_IO.cpp:334;
    // containDistance() does not use it)
    CHECK(ctx, rtc_greedy_mash(ctx, ds.d_hashes, ds.width, ds.d_start, ds.d_len, ds.n, mh.kmerSize, (int)mh.isContainment,
                               (uint32_t)mh.sketchSize, o.threshold, rep_of.data(), &ncl));
    vector<vector<int>> cluster = clusters_from_rep_of(rep_of);
    print_result(cluster, genomes, byFile, o.outputFile);
    cerr << "-----write the cluster result into: " << o.outputFile << endl;
    cerr << "-----the cluster number of " << o.outputFile << " is: " << cluster.size() << endl;
    return 0;
  }
  cerr << "---the threshold is: " << o.threshold << endl << "---the thread number is: " << o.threads << endl;
  if (byFile != o.sketchByFile) cerr << "Warning: the input format of append genomes and pre-sketched genome is not same" << endl;
  st.genomes = pre; st.sk = KssdSketchFile(); st.sk.use64 = true; st.sk.h64 = mh.hashes;
  st.reps = KssdSketchFile(); st.reps.use64 = true; st.rep_genomes.clear();
  for (int rep_id : st.rep_ids) {
    if (rep_id < 0 || (size_t)rep_id >= pre.size()) {
      cerr << "ERROR: Representative ID " << rep_id << " out of range [0, " << pre.size() << "), cannot continue" << endl;
      return 1;
    }
    st.rep_genomes.push_back(pre[rep_id]);
    st.reps.h64.push_back(mh.hashes[rep_id]);
  }
  cerr << "Successfully loaded " << st.rep_ids.size() << " representatives" << endl << "Rebuilding inverted index..." << endl;
  if (st.kmer_size <= 0 || st.sketch_size <= 0) {
    cerr << "ERROR: Invalid kmer_size or sketch_size: kmer=" << st.kmer_size << ", sketch=" << st.sketch_size << endl;
    return 1;
  }
  if (st.is_containment && mh.containCompress <= 0) { cerr << "ERROR: is_containment is true but contain_compress is " << mh.containCompress << endl; return 1; }
  cerr << "Computing sketches for new genomes..." << endl << "  Parameters: kmer_size=" << st.kmer_size << ", sketch_size=" << st.sketch_size
       << ", is_containment=" << st.is_containment << ", contain_compress=" << mh.containCompress << endl;
  vector<GenomeInfo> add; MinHashSketchFile mh2;
  sketch_new(st.kmer_size, st.sketch_size, st.is_containment, add, mh2);
  cerr << "New genomes sketched: " << add.size() << endl;
  if (add.empty()) { cerr << "ERROR: No new sketches generated" << endl; return 0; }
  cerr << "Starting incremental clustering..." << endl;
  KssdSketchFile ks2; ks2.use64 = true; ks2.h64 = std::move(mh2.hashes);
  if (kssd_incremental_cluster(ctx, st, add, ks2, true) != 0) return 1;
  cerr << "Incremental clustering completed" << endl;
  if (!o.noSave && o.saveRep && !save_minhash_cluster_state(state_file, st)) return 1;
  print_result(st.clusters, st.genomes, o.sketchByFile, o.outputFile);
  cerr << "-----write the cluster result into: " << o.outputFile << endl;
  cerr << "-----the cluster number of " << o.outputFile << " is: " << st.clusters.size() << endl;
  cerr << "-----updated cluster state saved" << endl;
  return 0;
}
#endif

// Will this run use exactly one GPU?  Answered without the HIP runtime (its environment must be final before it starts): a
// --gpus / RTC_GPUS choice that names one device, or "all" on a host whose driver topology lists one GPU node.
static bool single_gpu_run(const string& spec) {
  if (spec != "all") {  // as the spec is parsed further down: digits alone are a COUNT (--gpus 4 = four GPUs), a comma list names devices
    if (spec.find(',') == string::npos && spec.find_first_not_of("0123456789") == string::npos && atoi(spec.c_str()) > 0) return atoi(spec.c_str()) == 1;
    int named = 0;
    for (size_t p0 = 0; p0 <= spec.size();) { size_t p1 = spec.find(',', p0); if (p1 == string::npos) p1 = spec.size(); if (p1 > p0) named++; p0 = p1 + 1; }
    return named == 1;
  }
  for (const char* v : {"HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"})
    if (const char* e = getenv(v)) return *e != 0 && strchr(e, ',') == nullptr;
  int gpus = 0;
  for (int node = 0; node < 64; node++) {
    FILE* f = fopen(("/sys/class/kfd/kfd/topology/nodes/" + std::to_string(node) + "/properties").c_str(), "r");
    if (!f) break;
    char key[64]; unsigned long long val;
    while (fscanf(f, "%63s %llu", key, &val) == 2)
      if (strcmp(key, "simd_count") == 0 && val > 0) gpus++;
    fclose(f);
  }
  return gpus == 1;
}

int main(int argc, char** argv) {
#ifdef RTC_MEASURE
  if (const char* e = getenv("RTC_START_DELAY_MS")) usleep(1000 * atoi(e));
#endif
  const double t_main = get_sec();
  Options o = parse(argc, argv);
  // The staging batches leave pageable memory 15 % faster through the runtime's copy kernels than through the SDMA engines
  // (2 048 x 5 Mbp: output 0.112-0.127 s after the runtime is up against 0.128-0.153 s, tools/cli_timeline.py) and the
  // sketch kernels leave the CUs idle two thirds of the time anyway.  Measured on ONE GPU only, so only a run on one GPU
  // gets it: with several, RCCL's collectives and the share step's peer copies would land on the CUs beside the sketch
  // kernels, which nobody has timed.  Has to be in the environment before the runtime starts (hence the look at the
  // driver's topology instead of a device count); a value the user has set is left alone.
  if (single_gpu_run(o.gpus.empty() ? (getenv("RTC_GPUS") ? getenv("RTC_GPUS") : "all") : o.gpus)) setenv("HSA_ENABLE_SDMA", "0", 0);
  if (!o.has_output && !o.db_stats) { cerr << "ERROR: option -o/--output is required (unless --buildDB or --stats is used)" << endl; return 1; }
  if (o.threads < 1) { fprintf(stderr, "-----Invalid thread number %d\n", o.threads); return 1; }
  fprintf(stderr, "-----set the thread number %d\n", o.threads);
  if (!o.has_threshold) { o.threshold = 0.05; cerr << "-----use default threshold: " << o.threshold << endl; }

#ifdef GREEDY_CLUST
  // ---- RepDB mode (src/main.cpp:160-170, :300-370) ----
  const bool db_action = o.db_build || o.db_query || o.db_assign || o.db_stats;
  if (db_action && o.repdb_path.empty()) { cerr << "ERROR: --build / --query / --assign / --stats require --db" << endl; return 1; }
  if ((int)o.db_build + (int)o.db_query + (int)o.db_assign + (int)o.db_stats > 1) { cerr << "ERROR: --build, --query, --assign and --stats exclude each other" << endl; return 1; }
  if (!o.repdb_path.empty()) {
    if (o.db_stats) {
      KssdClusterState st;
      if (o.is_fast) { if (!load_kssd_repdb(o.repdb_path, st)) { cerr << "ERROR: Failed to load RepDB from: " << o.repdb_path << endl; return 1; } }
      else if (!load_minhash_repdb(o.repdb_path, st)) { cerr << "ERROR: Failed to load MinHash RepDB from: " << o.repdb_path << endl; return 1; }
      print_kssd_repdb_stats(st, std::cout);
      return 0;
    }
    if (o.db_build && !o.has_presketched && !o.has_input) { cerr << "ERROR: --build requires --presketched <folder> or -i <genome_list> -l" << endl; return 1; }
    if (o.db_query && !o.has_input) { cerr << "ERROR: --query requires -i <input_file>" << endl; return 1; }
    if (o.db_assign && !o.has_input) { cerr << "ERROR: --assign requires -i <input_file>" << endl; return 1; }
    if (!db_action && !o.has_append) { cerr << "ERROR: --db requires one of: --build, --query, --assign, --append, --stats" << endl; return 1; }
  }
  if (o.has_append && o.has_input) { cerr << "ERROR: --append and -i/--input exclude each other" << endl; return 1; }
  if (o.has_append && !o.has_presketched && o.repdb_path.empty()) { cerr << "ERROR option --append, option --presketched needed" << endl; return 1; }  // src/main.cpp:378-381
#endif
#ifndef GREEDY_CLUST
  // ---- --premsted: no sketching, no GPU (clust_from_mst[_fast], src/sub_command.cpp:1760-1934) ----
  if (o.has_append && o.has_input) { cerr << "ERROR: --append and -i/--input exclude each other" << endl; return 1; }
  if (o.has_append && !o.has_presketched && !o.has_premsted) { cerr << "ERROR option --append, option --presketched or --premsted needed" << endl; return 1; }
  if (o.has_premsted && !o.has_append) {
    vector<GenomeInfo> genomes; vector<rtc_edge> mst; bool byFile = true;
    if (!load_genome_info(o.folder_path, "mst", genomes, o.is_fast, byFile)) return 1;
    if (!load_mst(o.folder_path, mst)) return 1;
    write_trees(o, genomes, mst, byFile);
    cluster_from_mst(mst, genomes, byFile, o.outputFile, o.threshold);
    if (o.dense) {  // clust_from_mst with !no_dense: the stored mst.dense drives the noise pass (src/sub_command.cpp:1795-1822)
      vector<int32_t> dense; int span = 0, gn = 0;
      if (!load_dense(o.folder_path, dense, span, gn) || gn != (int)genomes.size()) return 1;
      remove_noise_and_print(mst, genomes, byFile, o.outputFile, o.threshold, dense, span);
    }
    return 0;
  }
#endif

  // ---- GPUs: one context (and one host thread when it works) per device; RCCL communicators among them ----
  // The HIP runtime, the contexts and the communicators come up on a thread of their own (0.05-0.24 s): a run from a genome
  // list reads the list, sizes its slots and PARSES ITS FIRST BATCH meanwhile (sketch_files); every other flow waits here.
  vector<Gpu> gpus;
  auto init_gpus = [&]() -> int {
    string spec = o.gpus.empty() ? (getenv("RTC_GPUS") ? getenv("RTC_GPUS") : "all") : o.gpus;
    const int ndev = rtc_device_count();
    bool gpus_by_default = false, single_gpu_flow = false;
    vector<int> devs;
    // Sketching from genome files spreads over every GPU of the choice in every flow (lanes on all of them, the rows
    // shared afterwards); of the clustering steps only the row-sharded MST does.  Greedy clustering has a serial
    // dependency on the representative set, --dense / --inverted-index=false evaluate their pairs on one GPU: those
    // cluster on the first GPU.  Flows that sketch little or nothing here (--presketched, --append, --db) take the first
    // GPU of the choice alone and never open a communicator.
    single_gpu_flow =
#ifdef GREEDY_CLUST
        o.has_append || !o.repdb_path.empty() || o.has_presketched;
#else
        o.has_append || ((o.dense || !o.useIndex) && o.has_presketched);
#endif
    if (spec == "all") { gpus_by_default = true; for (int d = 0; d < ndev; d++) devs.push_back(d); }
    else if (spec.find(',') == string::npos && atoi(spec.c_str()) > 0 && spec.find_first_not_of("0123456789") == string::npos) {
      const int want = atoi(spec.c_str());
      if (want > ndev) { fprintf(stderr, "ERROR: --gpus %d but only %d GPU(s) are visible\n", want, ndev); return 1; }
      for (int d = 0; d < want; d++) devs.push_back(d);
    } else {
      size_t p0 = 0;
      while (p0 <= spec.size()) { size_t p1 = spec.find(',', p0); if (p1 == string::npos) p1 = spec.size(); if (p1 > p0) devs.push_back(atoi(spec.substr(p0, p1 - p0).c_str())); p0 = p1 + 1; }
    }
    if (single_gpu_flow && devs.size() > 1) {
      if (!gpus_by_default) fprintf(stderr, "-----this flow runs on one GPU: using GPU %d of the %zu named\n", devs[0], devs.size());
      devs.resize(1);
    }
    if (devs.empty()) { fprintf(stderr, "ERROR: no MI355X context: %s (%d devices)\n", rtc_last_error(nullptr), ndev); return 1; }
    gpus.resize(devs.size());
    vector<rtc_ctx*> ctxs;
    for (size_t g = 0; g < devs.size(); g++) {
      gpus[g].device = devs[g];
      int st = rtc_ctx_create(devs[g], &gpus[g].ctx);
      if (st != RTC_OK) { fprintf(stderr, "ERROR: no MI355X context on device %d: %s\n", devs[g], rtc_last_error(nullptr)); return 1; }
      ctxs.push_back(gpus[g].ctx);
    }
    // RTC_COMM_FORCE_RCCL=1 opens an RCCL communicator also for ONE GPU and sends the run through the share step and
    // rtc_mst_sharded: the loader and every collective of the multi-GPU path, driven from this binary on a one-GPU box
    if (gpus.size() > 1 || (getenv("RTC_COMM_FORCE_RCCL") && !single_gpu_flow)) {
      vector<rtc_comm*> comms(gpus.size(), nullptr);
      const int st = rtc_comm_init_all(ctxs.data(), (int)ctxs.size(), comms.data());
      if (st != RTC_OK) {
        // the default ("all") must not turn a missing or broken RCCL into a dead command line: one GPU does the whole
        // job, as it would on a one-GPU host.  A user who named several GPUs asked for them: that is an error.
        if (!gpus_by_default) { fprintf(stderr, "ERROR: rtc_comm_init_all failed (%d): %s\n", st, rtc_last_error(ctxs[0])); return 1; }
        fprintf(stderr, "Warning: no communicator among the %zu visible GPU(s) (%s); running on GPU %d alone\n", gpus.size(), rtc_last_error(ctxs[0]), gpus[0].device);
        for (size_t g = 1; g < gpus.size(); g++) rtc_ctx_destroy(gpus[g].ctx);
        gpus.resize(1);
      } else {
        for (size_t g = 0; g < gpus.size(); g++) gpus[g].comm = comms[g];
        fprintf(stderr, "-----use %zu GPUs (%s exchange)\n", gpus.size(), rtc_comm_backend(comms[0]));
      }
    }
    return 0;
  };
  int gpu_rc = 0;
  double t_gpus = 0;
  std::atomic<bool> gpus_done{false};
  std::thread gpu_thread([&]() { gpu_rc = init_gpus(); t_gpus = get_sec(); gpus_done.store(true, std::memory_order_release); });
  // whichever way this function is left before the GPUs were asked for (return, exit()): not with that thread inside HIP
  g_gpu_thread = &gpu_thread;
  atexit([]() { if (g_gpu_thread && g_gpu_thread->joinable()) g_gpu_thread->join(); });
  struct GpuJoin { std::thread& t; ~GpuJoin() { if (t.joinable()) t.join(); g_gpu_thread = nullptr; } } gpu_join{gpu_thread};
  rtc_ctx* ctx = nullptr;
  struct Joiner { ~Joiner() { join_warmup(); } } warm;
  bool gpus_up = false;
  const std::function<void()> wait_gpus = [&]() {
    if (gpus_up) return;
    const double t_asked = get_sec();
    gpu_thread.join();
    gpus_up = true;
    if (gpu_rc != 0) { fflush(nullptr); _exit(gpu_rc); }
    ctx = gpus[0].ctx;
    g_metrics.num("hip_init_s", t_gpus - t_main);
    g_metrics.num("hip_init_exposed_s", std::max(0.0, t_gpus - t_asked));  // what the main thread still had to wait for
    if (getenv("RTC_VERBOSE")) fprintf(stderr, "[ctx]   HIP runtime + %zu GPU context(s) in %.3fs (asked for at t+%.3fs, waited %.3fs)\n", gpus.size(),
                                       t_gpus - t_main, t_asked - t_main, std::max(0.0, t_gpus - t_asked));
    // The device code of the pair / MST / greedy phases is mapped at its first launch (~27 ms): a helper thread does
    // that with a toy clustering while this one reads and sketches (rtc_warmup; RTC_NO_WARMUP=1 leaves it out).
    if (!getenv("RTC_NO_WARMUP")) {
      std::vector<int> wdev;
      for (const Gpu& g : gpus) wdev.push_back(g.device);
      g_warmup_thread = std::thread([wdev]() { for (int d : wdev) (void)rtc_warmup(d); });
    }
  };
  const bool list_run = !o.has_presketched && !o.has_append && o.has_input && o.sketchByFile
#ifdef GREEDY_CLUST
                        && o.repdb_path.empty()
#endif
      ;
  const GpusReady gpus_ready{wait_gpus, &gpus_done};
  if (!list_run) wait_gpus();
  Resident rs;
#ifndef GREEDY_CLUST
  if (o.has_append) return append_clust_mst(o, gpus);
#else
  if (!o.repdb_path.empty()) {
    const int rc = o.db_build ? (o.is_fast ? repdb_build(o, gpus) : mh_repdb_build(o, gpus)) : o.db_query ? repdb_query(o, gpus) : o.db_assign ? repdb_assign(o, gpus) : repdb_append(o, gpus);
    for (Gpu& g : gpus) { if (g.comm) rtc_comm_destroy(g.comm); }
    for (Gpu& g : gpus) rtc_ctx_destroy(g.ctx);
    return rc;
  }
  if (o.has_append) {  // src/main.cpp:378-387
    const int rc = o.is_fast ? append_clust_greedy_fast(o, gpus) : append_clust_greedy(o, gpus);
    for (Gpu& g : gpus) { if (g.comm) rtc_comm_destroy(g.comm); }
    for (Gpu& g : gpus) rtc_ctx_destroy(g.ctx);
    return rc;
  }
#endif

  vector<GenomeInfo> genomes;
  MinHashSketchFile mh; KssdSketchFile ks;
  bool sketchByFile = true;
  string folder_path = o.folder_path;
  const bool from_sketches = o.has_presketched;
  bool greedy =
#ifdef GREEDY_CLUST
      true;
#else
      false;
#endif

  double t0 = get_sec();
  if (from_sketches) {
    if (o.is_fast) { if (!load_kssd_sketches(folder_path, genomes, ks, sketchByFile)) return 1; }
    else { if (!load_minhash_sketches(folder_path, genomes, mh, sketchByFile)) return 1; }
    cerr << "-----the size of sketches is: " << genomes.size() << endl;
    cerr << "========time of load genome Infos and sketch Infos is: " << get_sec() - t0 << endl;
  } else {
    if (!o.has_input) { cerr << "ERROR: -i/--input is required" << endl; return 1; }
    sketchByFile = o.sketchByFile;
    uint64_t maxSize, minSize, averageSize;
    vector<FastaRecord> seq_recs;
    if (o.sketchByFile) { if (!cal_size(o.inputFile, o.minLen, maxSize, minSize, averageSize)) return 1; }
    else {
      SeqModeSizes sz;
      if (!read_sequences(o.inputFile, o.minLen, seq_recs, sz)) return 1;
      maxSize = sz.maxSize; minSize = sz.minSize; averageSize = sz.totalSize / sz.number;
    }
    // main.cpp:632 (clust-mst --fast uses the kssd tuner) / :659 (everything else)
    if (o.is_fast && !greedy) { if (!tune_kssd_parameters(o.isSetKmer, maxSize, minSize, averageSize, o.isContainment, o.kmerSize, o.threshold, o.drlevel)) return 1; }
    else if (!tune_parameters(greedy, o.isSetKmer, maxSize, minSize, averageSize, o.isContainment, o.isJaccard, o.kmerSize, o.threshold, o.containCompress, o.sketchSize)) return 1;
    SketchJob job;
    job.kssd = o.is_fast; job.kmerSize = o.kmerSize; job.sketchSize = o.sketchSize; job.isContainment = o.isContainment;
    job.containCompress = o.containCompress; job.drlevel = o.drlevel; job.minLen = o.minLen; job.threads = o.threads;
    if (getenv("RTC_VERBOSE")) fprintf(stderr, "[tune]  cal_size + tune_parameters in %.3fs\n", get_sec() - t0);
    if (o.sketchByFile) sketch_files(gpus, o.inputFile, job, genomes, &mh, &ks, rs, !o.noSave, &gpus_ready);
    else { wait_gpus(); sketch_sequences(gpus, seq_recs, job, genomes, &mh, &ks); }
    wait_gpus();
    mh.kmerSize = o.kmerSize; mh.isContainment = o.isContainment; mh.containCompress = o.containCompress; mh.sketchSize = o.sketchSize;
    cerr << "-----the size of sketches (number of genomes or sequences) is: " << genomes.size() << endl;
    double t1 = get_sec();
    cerr << "========time of computing sketch is: " << t1 - t0 << "========" << endl;
    {
      uint64_t bases = 0;
      for (const GenomeInfo& gi : genomes) bases += o.sketchByFile ? gi.totalSeqLength : (uint64_t)gi.seq0.length;
      g_metrics.num("computing_sketch_s", t1 - t0);
      g_metrics.num("bases", (double)bases);
      g_metrics.num("sketch_gbp_per_s", (double)bases / (t1 - t0) / 1e9);
    }
    folder_path = current_date_time();
    if (!o.noSave) {
      string command = "mkdir -p " + folder_path;
      if (system(command.c_str()) != 0) { cerr << "ERROR: cannot create " << folder_path << endl; return 1; }
      if (o.is_fast) { save_kssd_sketches(genomes, ks, folder_path, sketchByFile); if (!greedy) save_kssd_index(ks, folder_path); }
      else { save_minhash_sketches(genomes, mh, folder_path, sketchByFile); save_minhash_index(mh, folder_path); }
      cerr << "========time of saveSketches is: " << get_sec() - t1 << "========" << endl;
      g_metrics.num("saveSketches_s", get_sec() - t1);
    }
  }
  if (genomes.empty()) { cerr << "ERROR: no genome to cluster" << endl; return 1; }
  const int kmer_size = o.is_fast ? ks.info.half_k * 2 : mh.kmerSize;

  double t2 = get_sec();
#ifdef GREEDY_CLUST
  // ---- clust-greedy: compute_clusters GREEDY branch (src/sub_command.cpp:2894-2922, :1963-1986) ----
  vector<uint32_t> size_cfg, kssd_order;
  if (o.is_fast) {
    // src/greedy.cpp:594-597: sort by hash count, descending, comparator without tie-break
    vector<size_t> perm(genomes.size());
    iota(perm.begin(), perm.end(), 0);
    auto cnt = [&](size_t i) { return rs.ok ? (size_t)rs.counts[i] : (ks.use64 ? ks.h64[i].size() : ks.h32[i].size()); };
    struct Item { size_t idx; size_t c; };
    vector<Item> items(genomes.size());
    for (size_t i = 0; i < items.size(); i++) items[i] = Item{i, cnt(i)};
    std::sort(items.begin(), items.end(), [](const Item& a, const Item& b) { return a.c > b.c; });
    vector<GenomeInfo> g2; KssdSketchFile k2; k2.info = ks.info; k2.use64 = ks.use64;
    for (const Item& it : items) {
      g2.push_back(genomes[it.idx]);
      kssd_order.push_back((uint32_t)it.idx);  // resident rows are addressed through this order, not moved
      if (!rs.ok || (o.saveRep && !o.noSave)) { if (ks.use64) k2.h64.push_back(ks.h64[it.idx]); else k2.h32.push_back(ks.h32[it.idx]); }
    }
    genomes.swap(g2); ks = std::move(k2);
  } else {
    if (from_sketches) {
      // clust_from_sketches GREEDY branch: sort by genome size desc, id asc (src/sub_command.cpp:2657-2660)
      vector<size_t> perm(genomes.size());
      iota(perm.begin(), perm.end(), 0);
      std::sort(perm.begin(), perm.end(), [&](size_t a, size_t b) {
        if (!sketchByFile) {  // cmpSeqSize (src/SketchInfo.cpp:54-58)
          if (genomes[a].seq0.length != genomes[b].seq0.length) return genomes[a].seq0.length > genomes[b].seq0.length;
          return genomes[a].id < genomes[b].id;
        }
        if (genomes[a].totalSeqLength != genomes[b].totalSeqLength) return genomes[a].totalSeqLength > genomes[b].totalSeqLength;
        return genomes[a].id < genomes[b].id;
      });
      vector<GenomeInfo> g2; vector<vector<uint64_t>> h2;
      for (size_t p : perm) { g2.push_back(genomes[p]); h2.push_back(mh.hashes[p]); }
      genomes.swap(g2); mh.hashes.swap(h2);
      // loadSketches builds MinHash(k, containCompress) in containment mode, so getSketchSize() reports
      // containCompress there (src/Sketch_IO.cpp:334); fixed mode reports sketchSize
      size_cfg.assign(genomes.size(), mh.isContainment ? (uint32_t)mh.containCompress : (uint32_t)mh.sketchSize);
    } else {
      size_cfg.resize(genomes.size());
      for (size_t i = 0; i < genomes.size(); i++)
        size_cfg[i] = !mh.isContainment ? (uint32_t)mh.sketchSize
                      : sketchByFile ? (uint32_t)std::max(file_length_for_containment(genomes[i].fileName) / mh.containCompress, 100)
                                     : (uint32_t)std::max(genomes[i].seq0.length / mh.containCompress, 100);
    }
  }
  // greedy has a serial dependency on the representative set: one GPU clusters (replicas only)
  DeviceSketches ds;
  if (rs.ok) resident_sketches(ctx, gpus[0], rs, o.is_fast ? &kssd_order : nullptr, ds);
  else if (o.is_fast) upload_sketches(ctx, ks.use64 ? &ks.h64 : nullptr, ks.use64 ? nullptr : &ks.h32, ds);
  else upload_sketches(ctx, &mh.hashes, nullptr, ds);
  vector<int32_t> rep_of(genomes.size());
  uint32_t ncl = 0;
  if (!o.is_fast && !o.useIndex)  // greedyCluster: every representative, MinHash::distance() (src/sub_command.cpp:2683,2914)
    CHECK(ctx, rtc_greedy_mash(ctx, ds.d_hashes, ds.width, ds.d_start, ds.d_len, ds.n, kmer_size, (int)mh.isContainment, size_cfg[0], o.threshold,
                               rep_of.data(), &ncl));
  else
    CHECK(ctx, rtc_greedy(ctx, ds.d_hashes, ds.width, ds.d_start, ds.d_len, ds.n, o.is_fast ? nullptr : size_cfg.data(), kmer_size,
                          o.is_fast ? 0 : (int)mh.isContainment, o.is_fast ? 1 : 0, o.threshold, rep_of.data(), &ncl));
  vector<vector<int>> cluster = clusters_from_rep_of(rep_of);
  if (!o.is_fast && o.useIndex && o.saveRep && (from_sketches || !o.noSave)) {
    // MinHashInitialClusterWithState + save (src/sub_command.cpp:2673-2679, :2904-2909; src/greedy.cpp:1904-1960)
    KssdClusterState st;
    st.minhash = true; st.threshold = o.threshold; st.kmer_size = kmer_size; st.sketch_size = (int)size_cfg[0]; st.is_containment = mh.isContainment;
    st.genomes = genomes; st.sk.use64 = true; st.sk.h64 = mh.hashes; st.clusters = cluster; st.reps.use64 = true;
    cerr << "Building inverted index from " << cluster.size() << " representatives..." << endl;
    for (const auto& c : cluster) if (!c.empty()) { st.rep_ids.push_back(c[0]); st.rep_genomes.push_back(genomes[c[0]]); st.reps.h64.push_back(mh.hashes[c[0]]); }
    const string state_file = folder_path + "/cluster_state.bin";
    if (!save_minhash_cluster_state(state_file, st)) return 1;
    cerr << "-----saved cluster state (with inverted index) to: " << state_file << endl;
  }
  if (o.is_fast && o.saveRep && !o.noSave && !from_sketches) {  // compute_kssd_clusters, src/sub_command.cpp:1962-1967
    KssdClusterState st;
    KssdParameters info = ks.info;
    info.genomeNumber = (int)genomes.size();
    fill_cluster_state(st, o.threshold, kmer_size, info, genomes, ks, cluster);
    const string state_file = folder_path + "/cluster_state.bin";
    if (!save_kssd_cluster_state(state_file, st)) return 1;
    cerr << "-----saved cluster state (with inverted index) to: " << state_file << endl;
  }
  print_result(cluster, genomes, sketchByFile, o.outputFile);
  cerr << "-----write the cluster result into: " << o.outputFile << endl;
  cerr << "-----the cluster number of " << o.outputFile << " is: " << cluster.size() << endl;
  cerr << "========time of greedyCluster is: " << get_sec() - t2 << "========" << endl;
  g_metrics.num("greedyCluster_s", get_sec() - t2);
  g_metrics.num("clusters", (double)cluster.size());
#else
  // ---- clust-mst: compute_clusters MST branch (src/sub_command.cpp:2924-3053, :1988-2152) ----
  const int is_containment = o.is_fast ? (int)o.isContainment : (int)mh.isContainment;
  vector<rtc_edge> mst(genomes.size());
  uint64_t nedges = 0;
  const size_t G = gpus.size();
  vector<DeviceSketches> dss(G);
  on_all_gpus(gpus, [&](size_t g) {  // every GPU holds the complete sketch set (resident rows, or an upload of the loaded folder)
    if (rs.ok) resident_sketches(gpus[g].ctx, gpus[g], rs, nullptr, dss[g]);
    else if (o.is_fast) upload_sketches(gpus[g].ctx, ks.use64 ? &ks.h64 : nullptr, ks.use64 ? nullptr : &ks.h32, dss[g]);
    else upload_sketches(gpus[g].ctx, &mh.hashes, nullptr, dss[g]);
  });
  vector<int32_t> dense; uint64_t ani[101];
  if (!o.useIndex && !o.is_fast) {  // modifyMST, the dense loop (src/sub_command.cpp:2764,2995): every pair, MinHash::distance()
    const DeviceSketches& ds = dss[0];
    if (o.dense) dense.resize((size_t)DENSE_SPAN * ds.n);
    CHECK(ctx, rtc_mst_mash(ctx, ds.d_hashes, ds.width, ds.d_start, ds.d_len, ds.n, 0, kmer_size, is_containment, (uint32_t)mh.sketchSize, mst.data(),
                            &nedges, o.dense ? DENSE_SPAN : 0, o.dense ? dense.data() : nullptr, o.dense ? ani : nullptr));
  } else if ((G == 1 && !gpus[0].comm) || o.dense) {  // --dense: the histograms are accumulated beside the single-GPU candidate list
    const DeviceSketches& ds = dss[0];
    if (o.dense) dense.resize((size_t)DENSE_SPAN * ds.n);
    CHECK(ctx, rtc_mst_dense(ctx, ds.d_hashes, ds.width, ds.d_start, ds.d_len, ds.n, 0, kmer_size, is_containment, o.threshold, mst.data(), &nedges,
                             o.dense ? DENSE_SPAN : 0, o.dense ? dense.data() : nullptr, o.dense ? ani : nullptr));
  } else {
    // N x N row-sharded across the GPUs, one all-reduce per Boruvka round; every rank ends with the same forest
    vector<vector<rtc_edge>> out(G, vector<rtc_edge>(genomes.size()));
    vector<uint64_t> ne(G, 0);
    vector<rtc_shard_stats> stt(G);
    on_all_gpus(gpus, [&](size_t g) {
      const DeviceSketches& ds = dss[g];
      CHECK(gpus[g].ctx, rtc_mst_sharded(gpus[g].ctx, gpus[g].comm, ds.d_hashes, ds.width, ds.d_start, ds.d_len, ds.n, kmer_size,
                                         is_containment, o.threshold, out[g].data(), &ne[g], &stt[g]));
    });
    for (size_t g = 1; g < G; g++)
      if (ne[g] != ne[0] || memcmp(out[g].data(), out[0].data(), ne[0] * sizeof(rtc_edge)) != 0) { fprintf(stderr, "ERROR: GPU %zu ended with a different forest\n", g); return 1; }
    nedges = ne[0];
    mst.swap(out[0]);
    if (getenv("RTC_VERBOSE"))
      for (size_t g = 0; g < G; g++)
        fprintf(stderr, "[mst gpu %zu] rows %u..%u, %llu candidate edges, pair %.2f ms, boruvka %.2f ms (%u rounds)\n", g, stt[g].row0, stt[g].row1,
                (unsigned long long)stt[g].cand_edges, stt[g].pair_ms, stt[g].mst_ms, stt[g].rounds);
  }
  mst.resize(nedges);
  double t3 = get_sec();
  cerr << "========time of generateMST is: " << t3 - t2 << "========" << endl;
  g_metrics.num("generateMST_s", t3 - t2);
  g_metrics.num("mst_edges", (double)nedges);
  if (!o.noSave && !from_sketches) {
    save_genome_info(genomes, folder_path, "mst", sketchByFile, o.is_fast);
    save_mst(mst, folder_path);
    cerr << "========time of saveMST is: " << get_sec() - t3 << "========" << endl;
    g_metrics.num("saveMST_s", get_sec() - t3);
  }
  write_trees(o, genomes, mst, sketchByFile);
  cluster_from_mst(mst, genomes, sketchByFile, o.outputFile, o.threshold);
  if (o.dense) {
    if (!o.noSave && !from_sketches) { save_ani(folder_path, ani); save_dense(folder_path, dense, DENSE_SPAN, (int)genomes.size()); }
    remove_noise_and_print(mst, genomes, sketchByFile, o.outputFile, o.threshold, dense, DENSE_SPAN);
  }
#endif
  const double t_end = get_sec();
#ifdef GREEDY_CLUST
  g_metrics.str("command", "clust-greedy");
#else
  g_metrics.str("command", "clust-mst");
#endif
  g_metrics.str("sketch", o.is_fast ? "kssd" : "minhash");
  g_metrics.num("gpus", (double)gpus.size());
  g_metrics.num("genomes", (double)genomes.size());
  g_metrics.num("kmer_size", (double)kmer_size);
  g_metrics.num("threshold", o.threshold);
  g_metrics.num("threads", (double)o.threads);
  g_metrics.num("from_sketches", from_sketches ? 1 : 0);
  g_metrics.num("total_s", t_end - t_main);
  g_metrics.write();
  join_warmup();
#ifdef RTC_MEASURE
  if (const char* e = getenv("RTC_EXIT_DELAY_MS")) usleep(1000 * atoi(e));
  if (getenv("RTC_EXIT_PROBE")) {
    const double p0 = get_sec();
    for (char* p : g_exit_probe_host) free(p);
    const double p1 = get_sec();
    for (auto& d : g_exit_probe_dev) (void)rtc_dev_free(d.first, d.second);
    const double p2 = get_sec();
    fprintf(stderr, "[probe] %zu host staging buffers freed in %.3fs, %zu device buffers in %.3fs\n", g_exit_probe_host.size(), p1 - p0,
            g_exit_probe_dev.size(), p2 - p1);
    if (strcmp(getenv("RTC_EXIT_PROBE"), "reset") == 0) {
      for (Gpu& g : gpus) (void)rtc_debug_device_reset(g.device);
      fprintf(stderr, "[probe] hipDeviceReset in %.3fs\n", get_sec() - p2);
      fprintf(stderr, "[exit]  output written at t+%.3fs, contexts released in %.3fs (main entered at %.6f, leaving at %.6f)\n", t_end - t_main, 0.0, t_main, get_sec());
      fflush(nullptr);
      _exit(0);
    }
  }
#endif
  for (Gpu& g : gpus) { if (g.comm) rtc_comm_destroy(g.comm); }
  for (Gpu& g : gpus) rtc_ctx_destroy(g.ctx);
  if (getenv("RTC_VERBOSE")) fprintf(stderr, "[exit]  output written at t+%.3fs, contexts released in %.3fs (main entered at %.6f, leaving at %.6f)\n", t_end - t_main,
                                     get_sec() - t_end, t_main, get_sec());
  // Everything is written and closed: leave without unmapping the GBs of staging memory page by page and without the
  // HIP runtime's own teardown (0.15 s of a 0.9 s run on 41 Gbp); the kernel reclaims both at once.
  std::cout.flush();
  fflush(nullptr);
  _exit(0);
}

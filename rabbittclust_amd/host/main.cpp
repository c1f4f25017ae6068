// main.cpp -- clust-mst / clust-greedy command lines on top of the MI355X C ABI.
//
// Mirrors the flag surface and workflow dispatch of the reference's src/main.cpp:113-254,291-671
// for the sketch + all-pairs + cluster path: list-mode input (-l), MinHash
// and KSSD (--fast) sketching, --presketched / --premsted resume, -e/--no-save, and the same
// intermediate folder (info.sketch, hash.sketch, minhash.sketch.index, kssd.*, info.mst,
// edge.mst).  Built twice: -DGREEDY_CLUST gives clust-greedy, otherwise clust-mst
// (CMakeLists.txt:40-58 of the reference does the same).  Incremental (--append, --db, --save-rep),
// tree writers, --dense, --auto-threshold and single-FASTA mode are outside this path and exit
// with a message.
#include <math.h>
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/time.h>

#include <algorithm>
#include <fstream>
#include <iostream>
#include <numeric>
#include <thread>

#include "rtc_host.h"

using namespace std;
using namespace rtc;

static double get_sec() { struct timeval tv; gettimeofday(&tv, NULL); return (double)tv.tv_sec + (double)tv.tv_usec / 1000000; }

#define CHECK(ctx, call)                                                                         \
  do {                                                                                           \
    int st__ = (call);                                                                           \
    if (st__ != RTC_OK) {                                                                        \
      fprintf(stderr, "ERROR: %s failed (%d): %s\n", #call, st__, rtc_last_error(ctx));          \
      exit(1);                                                                                   \
    }                                                                                            \
  } while (0)

struct DeviceSketches {  // sketches resident in HBM in the CSR the pair kernels read
  void* d_hashes = nullptr; uint64_t* d_start = nullptr; uint32_t* d_len = nullptr;
  uint32_t n = 0; int width = 8;
};

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
  madvise(p, bytes, MADV_HUGEPAGE);
  return (char*)p;
}

struct Batch { vector<size_t> files; vector<uint64_t> slot_off, slot_len; uint64_t bytes = 0; };

struct FileResult {           // indexed by list position, so late (retried) files keep list order
  SequenceInfo first; uint64_t total = 0; int flen = 0; bool kept = false;
  vector<uint64_t> h64; vector<uint32_t> h32;
};

static void sketch_files(rtc_ctx* ctx, const string& inputFile, const SketchJob& job, vector<GenomeInfo>& genomes,
                         MinHashSketchFile* mh, KssdSketchFile* ks) {
  const double tp00 = get_sec();
  const vector<string> fileList = read_list(inputFile);
  const size_t nfiles = fileList.size();
  const bool verbose = getenv("RTC_VERBOSE") != nullptr;
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
#pragma omp parallel for num_threads(job.threads) schedule(dynamic, 16)
  for (long i = 0; i < (long)nfiles; i++) {
    slot[i] = genome_slot_bytes(fileList[i]);
    res[i].flen = job.isContainment ? file_length_for_containment(fileList[i]) : 0;
  }
  for (size_t i = 0; i < nfiles; i++)
    if (slot[i] == 0) { fprintf(stderr, "cannot open the genome file: %s\n", fileList[i].c_str()); exit(1); }
  auto plan = [&](const vector<size_t>& files, const vector<uint64_t>& need) {
    vector<Batch> out;
    for (size_t q = 0; q < files.size(); q++) {
      if (out.empty() || (out.back().bytes + need[q] > BATCH_BYTES && !out.back().files.empty())) out.emplace_back();
      Batch& b = out.back();
      b.files.push_back(files[q]); b.slot_off.push_back(b.bytes); b.slot_len.push_back(need[q]);
      b.bytes += need[q];
    }
    return out;
  };
  vector<size_t> all(nfiles);
  iota(all.begin(), all.end(), 0);
  vector<Batch> batches = plan(all, slot);

  // ---- staging: two page-locked host buffers and one device buffer, reused by every batch ----
  uint64_t buf_bytes = 0;
  char* stage[2] = {nullptr, nullptr};
  void* d_seq = nullptr;
  // Pageable staging by default: page-locking costs ~0.15 s/GB up front while the pageable PCIe copy
  // already runs at > 30 GB/s on the MI355X hosts measured; RTC_STAGE_PINNED=1 page-locks instead.
  bool pinned = getenv("RTC_STAGE_PINNED") != nullptr;
  auto free_stage = [&]() {
    for (int i = 0; i < 2; i++) {
      if (stage[i] && pinned) CHECK(ctx, rtc_host_free(ctx, stage[i]));
      else free(stage[i]);
      stage[i] = nullptr;
    }
  };
  auto ensure_buffers = [&](uint64_t need) {
    if (need <= buf_bytes) return;
    free_stage();
    if (d_seq) CHECK(ctx, rtc_dev_free(ctx, d_seq));
    buf_bytes = need;
    for (int i = 0; i < 2; i++) {
      if (pinned && rtc_host_alloc(ctx, buf_bytes + 64, (void**)&stage[i]) != RTC_OK) {
        // the host refuses to page-lock this much (ulimit -l): stage through pageable memory instead
        fprintf(stderr, "-----cannot page-lock %.2f GB (%s), staging through pageable memory\n", buf_bytes / 1e9, rtc_last_error(ctx));
        free_stage();
        pinned = false; i = -1;
        continue;
      }
      if (!pinned && !(stage[i] = alloc_pageable(buf_bytes + 64))) { fprintf(stderr, "ERROR: cannot allocate %.2f GB of staging memory\n", buf_bytes / 1e9); exit(1); }
    }
    CHECK(ctx, rtc_dev_alloc(ctx, buf_bytes + 64, &d_seq));
  };
  uint64_t maxb = 0;
  for (const Batch& b : batches) maxb = std::max(maxb, b.bytes);
  ensure_buffers(maxb);
  if (verbose) fprintf(stderr, "[plan] %zu files, %zu batches, staging 2 x %.2f GB, %.3fs\n", nfiles, batches.size(), buf_bytes / 1e9, get_sec() - tp0);

  // ---- GPU side of one batch (runs on its own host thread while the next batch is parsed) ----
  auto gpu_batch = [&](const Batch& b, const char* h_seq) {
    const double t0 = get_sec();
    vector<uint64_t> off; vector<uint32_t> sizes; vector<size_t> kept;
    for (size_t q = 0; q < b.files.size(); q++) {
      FileResult& r = res[b.files[q]];
      if (!r.kept) continue;
      kept.push_back(b.files[q]);
      off.push_back(b.slot_off[q]);  // a genome extends to the next kept one: the gap holds only 'N'
      sizes.push_back(job.isContainment ? (uint32_t)std::max(r.flen / job.containCompress, 100) : (uint32_t)job.sketchSize);  // :919-924
    }
    const uint32_t nb = (uint32_t)kept.size();
    if (!nb) return;
    off.push_back(b.bytes);
    CHECK(ctx, rtc_copy_h2d(ctx, d_seq, h_seq, b.bytes + 64));
    const double t1 = get_sec();
    uint32_t* d_cnt = nullptr;
    CHECK(ctx, rtc_dev_alloc(ctx, (size_t)nb * 4, (void**)&d_cnt));
    vector<uint32_t> cnt(nb);
    if (!job.kssd) {
      const uint32_t stride = *std::max_element(sizes.begin(), sizes.end());
      uint64_t* d_out = nullptr;
      CHECK(ctx, rtc_dev_alloc(ctx, (size_t)nb * stride * 8, (void**)&d_out));
      CHECK(ctx, rtc_sketch_minhash_dev(ctx, (const uint8_t*)d_seq, off.data(), nb, job.kmerSize, 42, sizes.data(), stride,
                                        d_out, stride, d_cnt));
      vector<uint64_t> out((size_t)nb * stride);
      CHECK(ctx, rtc_copy_d2h(ctx, out.data(), d_out, out.size() * 8));
      CHECK(ctx, rtc_copy_d2h(ctx, cnt.data(), d_cnt, (size_t)nb * 4));
      for (uint32_t g = 0; g < nb; g++) res[kept[g]].h64.assign(out.begin() + (size_t)g * stride, out.begin() + (size_t)g * stride + cnt[g]);
      CHECK(ctx, rtc_dev_free(ctx, d_out));
    } else {
      uint64_t maxlen = 0;
      for (uint32_t g = 0; g < nb; g++) maxlen = std::max(maxlen, off[g + 1] - off[g]);
      uint32_t stride = (uint32_t)(maxlen / (1ull << (4 * job.drlevel)) * 3 / 2 + 256);
      const int w = use64 ? 8 : 4;
      while (true) {
        void* d_out = nullptr;
        CHECK(ctx, rtc_dev_alloc(ctx, (size_t)nb * stride * w, &d_out));
        int width = 0; uint32_t need = 0;
        int st = rtc_sketch_kssd_dev(ctx, (const uint8_t*)d_seq, off.data(), nb, job.kmerSize, job.drlevel, shuffled.data(),
                                     d_out, stride, d_cnt, &width, &need);
        if (st == RTC_ERR_OVERFLOW) { CHECK(ctx, rtc_dev_free(ctx, d_out)); stride = need + 64; continue; }
        CHECK(ctx, st);
        CHECK(ctx, rtc_copy_d2h(ctx, cnt.data(), d_cnt, (size_t)nb * 4));
        vector<unsigned char> out((size_t)nb * stride * w);
        CHECK(ctx, rtc_copy_d2h(ctx, out.data(), d_out, out.size()));
        for (uint32_t g = 0; g < nb; g++) {
          if (use64) { const uint64_t* p = (const uint64_t*)out.data() + (size_t)g * stride; res[kept[g]].h64.assign(p, p + cnt[g]); }
          else { const uint32_t* p = (const uint32_t*)out.data() + (size_t)g * stride; res[kept[g]].h32.assign(p, p + cnt[g]); }
        }
        CHECK(ctx, rtc_dev_free(ctx, d_out));
        break;
      }
    }
    CHECK(ctx, rtc_dev_free(ctx, d_cnt));
    if (verbose) fprintf(stderr, "[gpu]   %u genomes, %.2f GB: h2d %.3fs sketch+d2h %.3fs\n", nb, b.bytes / 1e9, t1 - t0, get_sec() - t1);
  };

  // ---- pipeline: parse batch i into stage[i&1] while the GPU thread works on batch i-1 ----
  vector<size_t> retry_files; vector<uint64_t> retry_need;
  std::thread worker;
  size_t done_files = 0, bi = 0;
  for (int round = 0; round < 2; round++) {  // round 1: files whose slot guess was too small (gzip ISIZE)
    for (const Batch& b : batches) {
      const double t0 = get_sec();
      char* buf = stage[bi & 1];
      vector<uint64_t> need(b.files.size(), 0);
#pragma omp parallel for num_threads(job.threads) schedule(dynamic)
      for (long q = 0; q < (long)b.files.size(); q++) {
        FileResult& r = res[b.files[q]];
        uint64_t used = 0, nrec = 0;
        char* dst = buf + b.slot_off[q];
        const int st = read_genome_file_flat(fileList[b.files[q]], dst, b.slot_len[q], used, r.first, r.total, nrec);
        if (st == 1) { fprintf(stderr, "cannot open the genome file: %s\n", fileList[b.files[q]].c_str()); exit(1); }
        if (st == 2) { need[q] = used + 1; used = 0; r.kept = false; }
        else r.kept = r.total >= job.minLen;                                 // :963
        if (!r.kept) used = 0;
        memset(dst + used, 'N', b.slot_len[q] - used);  // no k-mers in the gap, nor in dropped genomes
      }
      memset(buf + b.bytes, 'N', 64);
      for (size_t q = 0; q < b.files.size(); q++) if (need[q]) { retry_files.push_back(b.files[q]); retry_need.push_back(need[q]); }
      if (verbose) fprintf(stderr, "[parse] batch %zu: %zu files, %.2f GB in %.3fs\n", bi, b.files.size(), b.bytes / 1e9, get_sec() - t0);
      if (worker.joinable()) worker.join();
      if (shuffle_thread.joinable()) shuffle_thread.join();
      const Batch* bp = &b;
      worker = std::thread([&gpu_batch, bp, buf]() { gpu_batch(*bp, buf); });
      for (size_t q = 0; q < b.files.size(); q++, done_files++) if (done_files % 10000 == 0) cerr << "---finished sketching: " << done_files << " genomes" << endl;
      bi++;
    }
    if (worker.joinable()) worker.join();
    if (round == 1 || retry_files.empty()) break;
    batches = plan(retry_files, retry_need);
    maxb = 0;
    for (const Batch& b : batches) maxb = std::max(maxb, b.bytes);
    ensure_buffers(maxb);
    done_files -= retry_files.size();
    retry_files.clear(); retry_need.clear();
  }
  if (shuffle_thread.joinable()) shuffle_thread.join();
  const double tf0 = get_sec();
  if (pinned) free_stage();
  else {  // returning GBs of touched pages to the kernel takes a while: do it beside the clustering
    char* s0 = stage[0]; char* s1 = stage[1];
    stage[0] = stage[1] = nullptr;
    std::thread([s0, s1]() { free(s0); free(s1); }).detach();
  }
  // The device staging buffer stays allocated until the process ends: hipFree of a multi-GB buffer
  // costs ~0.4 s here and the clustering phase needs far less than the 288 GB that are there.
  (void)d_seq;
  if (verbose) fprintf(stderr, "[free]  host staging %.3fs\n", get_sec() - tf0);

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
  int threads = default_threads();
  bool sketchByFile = false, noSave = false, is_fast = false, isContainment = false, isJaccard = false, isSetKmer = false;
  bool has_threshold = false, has_input = false, has_presketched = false, has_premsted = false, has_output = false;
  double threshold = 0.05;
  int kmerSize = 19, sketchSize = 1000, containCompress = 1000, drlevel = 3;
  uint64_t minLen = 10000;
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
    else if (a == "--fast") o.is_fast = true;
    else if (a == "--drlevel") o.drlevel = atoi(need(i));
    else if (a == "--inverted-index") { /* always on, as in the reference (src/main.cpp:104,129) */ }
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
           "  --presketched DIR  --fast  --drlevel N"
#ifndef GREEDY_CLUST
           "  --premsted DIR"
#endif
      );
      exit(0);
    }
    else if (a == "--append" || a == "--db" || a == "--build" || a == "--query" || a == "--assign" || a == "--stats" ||
             a == "--save-rep" || a == "--top-k" || a == "--dense" || a == "--newick-tree" || a == "--phylip-tree" ||
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
}

[[maybe_unused]] static vector<vector<int>> clusters_from_rep_of(const vector<int32_t>& rep_of) {
  // cluster list in representative-creation order: [rep, members...] (src/greedy.cpp:1355-1367)
  vector<vector<int>> cl; vector<int> cid(rep_of.size(), -1);
  for (size_t i = 0; i < rep_of.size(); i++) if (rep_of[i] == (int32_t)i) { cid[i] = (int)cl.size(); cl.push_back({(int)i}); }
  for (size_t i = 0; i < rep_of.size(); i++) if (rep_of[i] != (int32_t)i) cl[cid[rep_of[i]]].push_back((int)i);
  return cl;
}

int main(int argc, char** argv) {
  Options o = parse(argc, argv);
  if (!o.has_output) { cerr << "ERROR: option -o/--output is required (unless --buildDB or --stats is used)" << endl; return 1; }
  if (o.threads < 1) { fprintf(stderr, "-----Invalid thread number %d\n", o.threads); return 1; }
  fprintf(stderr, "-----set the thread number %d\n", o.threads);
  if (!o.has_threshold) { o.threshold = 0.05; cerr << "-----use default threshold: " << o.threshold << endl; }

#ifndef GREEDY_CLUST
  // ---- --premsted: no sketching, no GPU (clust_from_mst[_fast], src/sub_command.cpp:1760-1934) ----
  if (o.has_premsted) {
    vector<GenomeInfo> genomes; vector<rtc_edge> mst; bool byFile = true;
    if (!load_genome_info(o.folder_path, "mst", genomes, o.is_fast, byFile)) return 1;
    if (!load_mst(o.folder_path, mst)) return 1;
    cluster_from_mst(mst, genomes, byFile, o.outputFile, o.threshold);
    return 0;
  }
#endif

  rtc_ctx* ctx = nullptr;
  { int st = rtc_ctx_create(0, &ctx); if (st != RTC_OK) { fprintf(stderr, "ERROR: no MI355X context: %s\n", rtc_last_error(nullptr)); return 1; } }

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
    if (!o.sketchByFile) unsupported("single-FASTA input (run with -l and a genome list)");
    uint64_t maxSize, minSize, averageSize;
    if (!cal_size(o.inputFile, o.minLen, maxSize, minSize, averageSize)) return 1;
    // main.cpp:632 (clust-mst --fast uses the kssd tuner) / :659 (everything else)
    if (o.is_fast && !greedy) { if (!tune_kssd_parameters(o.isSetKmer, maxSize, minSize, averageSize, o.isContainment, o.kmerSize, o.threshold, o.drlevel)) return 1; }
    else if (!tune_parameters(greedy, o.isSetKmer, maxSize, minSize, averageSize, o.isContainment, o.isJaccard, o.kmerSize, o.threshold, o.containCompress, o.sketchSize)) return 1;
    SketchJob job;
    job.kssd = o.is_fast; job.kmerSize = o.kmerSize; job.sketchSize = o.sketchSize; job.isContainment = o.isContainment;
    job.containCompress = o.containCompress; job.drlevel = o.drlevel; job.minLen = o.minLen; job.threads = o.threads;
    if (getenv("RTC_VERBOSE")) fprintf(stderr, "[tune]  cal_size + tune_parameters in %.3fs\n", get_sec() - t0);
    sketch_files(ctx, o.inputFile, job, genomes, &mh, &ks);
    mh.kmerSize = o.kmerSize; mh.isContainment = o.isContainment; mh.containCompress = o.containCompress; mh.sketchSize = o.sketchSize;
    cerr << "-----the size of sketches (number of genomes or sequences) is: " << genomes.size() << endl;
    double t1 = get_sec();
    cerr << "========time of computing sketch is: " << t1 - t0 << "========" << endl;
    folder_path = current_date_time();
    if (!o.noSave) {
      string command = "mkdir -p " + folder_path;
      if (system(command.c_str()) != 0) { cerr << "ERROR: cannot create " << folder_path << endl; return 1; }
      if (o.is_fast) { save_kssd_sketches(genomes, ks, folder_path, true); if (!greedy) save_kssd_index(ks, folder_path); }
      else { save_minhash_sketches(genomes, mh, folder_path, true); save_minhash_index(mh, folder_path); }
      cerr << "========time of saveSketches is: " << get_sec() - t1 << "========" << endl;
    }
  }
  if (genomes.empty()) { cerr << "ERROR: no genome to cluster" << endl; return 1; }
  const int kmer_size = o.is_fast ? ks.info.half_k * 2 : mh.kmerSize;

  double t2 = get_sec();
#ifdef GREEDY_CLUST
  // ---- clust-greedy: compute_clusters GREEDY branch (src/sub_command.cpp:2894-2922, :1963-1986) ----
  vector<uint32_t> size_cfg;
  if (o.is_fast) {
    // src/greedy.cpp:594-597: sort by hash count, descending, comparator without tie-break
    vector<size_t> perm(genomes.size());
    iota(perm.begin(), perm.end(), 0);
    auto cnt = [&](size_t i) { return ks.use64 ? ks.h64[i].size() : ks.h32[i].size(); };
    struct Item { size_t idx; size_t c; };
    vector<Item> items(genomes.size());
    for (size_t i = 0; i < items.size(); i++) items[i] = Item{i, cnt(i)};
    std::sort(items.begin(), items.end(), [](const Item& a, const Item& b) { return a.c > b.c; });
    vector<GenomeInfo> g2; KssdSketchFile k2; k2.info = ks.info; k2.use64 = ks.use64;
    for (const Item& it : items) { g2.push_back(genomes[it.idx]); if (ks.use64) k2.h64.push_back(ks.h64[it.idx]); else k2.h32.push_back(ks.h32[it.idx]); }
    genomes.swap(g2); ks = std::move(k2);
  } else {
    if (from_sketches) {
      // clust_from_sketches GREEDY branch: sort by genome size desc, id asc (src/sub_command.cpp:2657-2660)
      vector<size_t> perm(genomes.size());
      iota(perm.begin(), perm.end(), 0);
      std::sort(perm.begin(), perm.end(), [&](size_t a, size_t b) {
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
        size_cfg[i] = mh.isContainment ? (uint32_t)std::max(file_length_for_containment(genomes[i].fileName) / mh.containCompress, 100)
                                       : (uint32_t)mh.sketchSize;
    }
  }
  DeviceSketches ds;
  if (o.is_fast) upload_sketches(ctx, ks.use64 ? &ks.h64 : nullptr, ks.use64 ? nullptr : &ks.h32, ds);
  else upload_sketches(ctx, &mh.hashes, nullptr, ds);
  vector<int32_t> rep_of(genomes.size());
  uint32_t ncl = 0;
  CHECK(ctx, rtc_greedy(ctx, ds.d_hashes, ds.width, ds.d_start, ds.d_len, ds.n, o.is_fast ? nullptr : size_cfg.data(), kmer_size,
                        o.is_fast ? 0 : (int)mh.isContainment, o.is_fast ? 1 : 0, o.threshold, rep_of.data(), &ncl));
  vector<vector<int>> cluster = clusters_from_rep_of(rep_of);
  print_result(cluster, genomes, sketchByFile, o.outputFile);
  cerr << "-----write the cluster result into: " << o.outputFile << endl;
  cerr << "-----the cluster number of " << o.outputFile << " is: " << cluster.size() << endl;
  cerr << "========time of greedyCluster is: " << get_sec() - t2 << "========" << endl;
#else
  // ---- clust-mst: compute_clusters MST branch (src/sub_command.cpp:2924-3053, :1988-2152) ----
  DeviceSketches ds;
  if (o.is_fast) upload_sketches(ctx, ks.use64 ? &ks.h64 : nullptr, ks.use64 ? nullptr : &ks.h32, ds);
  else upload_sketches(ctx, &mh.hashes, nullptr, ds);
  const int is_containment = o.is_fast ? (int)o.isContainment : (int)mh.isContainment;
  vector<rtc_edge> mst(genomes.size());
  uint64_t nedges = 0;
  CHECK(ctx, rtc_mst(ctx, ds.d_hashes, ds.width, ds.d_start, ds.d_len, ds.n, kmer_size, is_containment, o.threshold, mst.data(), &nedges));
  mst.resize(nedges);
  double t3 = get_sec();
  cerr << "========time of generateMST is: " << t3 - t2 << "========" << endl;
  if (!o.noSave && !from_sketches) {
    save_genome_info(genomes, folder_path, "mst", true, o.is_fast);
    save_mst(mst, folder_path);
    cerr << "========time of saveMST is: " << get_sec() - t3 << "========" << endl;
  }
  cluster_from_mst(mst, genomes, sketchByFile, o.outputFile, o.threshold);
#endif
  rtc_ctx_destroy(ctx);
  return 0;
}

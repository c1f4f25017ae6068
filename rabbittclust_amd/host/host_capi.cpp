// host_capi.cpp -- small C ABI over the host library, for the CPU test-suite (ctypes).
#include <string.h>

#include "rtc_host.h"

using namespace rtc;

extern "C" {

// name\tcomment\tlength\n<sequence>\n per record ("noName" when the header has no comment), the
// same dump the reference harness (oracle/ref_harness.cpp over kseq.h) produces.
long rtch_fasta_dump(const char* path, char* out, long cap) {
  std::vector<FastaRecord> recs;
  if (!read_fasta(path, recs)) return -1;
  long pos = 0;
  auto put = [&](const char* s, long n) { if (out && pos + n <= cap) memcpy(out + pos, s, n); pos += n; };
  // kseq keeps its comment buffer between records: a record without a comment shows the previous
  // record's text (NULL -> "noName" only while no record has had one).  Only record 0 is ever used
  // by the reference (src/SketchInfo.cpp:944-947); the dump mimics the buffer for all records.
  std::string stale; bool have_stale = false;
  for (const FastaRecord& r : recs) {
    if (r.has_comment) { stale = r.comment; have_stale = true; }
    const std::string cm = r.has_comment ? r.comment : (have_stale ? stale : std::string("noName"));
    const std::string len = std::to_string(r.seq.size());
    put(r.name.data(), (long)r.name.size()); put("\t", 1);
    put(cm.data(), (long)cm.size()); put("\t", 1);
    put(len.data(), (long)len.size()); put("\n", 1);
    put(r.seq.data(), (long)r.seq.size()); put("\n", 1);
  }
  return pos;
}

// loads a sketch folder and writes it again (sketch files + index files): byte-level format check
int rtch_resave_folder(const char* in_dir, const char* out_dir, int kssd) {
  std::vector<GenomeInfo> g; bool byFile = true;
  if (kssd) {
    KssdSketchFile f;
    if (!load_kssd_sketches(in_dir, g, f, byFile)) return 1;
    save_kssd_sketches(g, f, out_dir, byFile);
    save_kssd_index(f, out_dir);
  } else {
    MinHashSketchFile f;
    if (!load_minhash_sketches(in_dir, g, f, byFile)) return 1;
    save_minhash_sketches(g, f, out_dir, byFile);
    save_minhash_index(f, out_dir);
  }
  return 0;
}

// --premsted flow without a GPU: info.mst + edge.mst -> cluster text; also re-saves both files
int rtch_premsted(const char* in_dir, const char* out_dir, const char* out_file, double threshold, int kssd) {
  std::vector<GenomeInfo> g; std::vector<rtc_edge> mst; bool byFile = true;
  if (!load_genome_info(in_dir, "mst", g, kssd != 0, byFile)) return 1;
  if (!load_mst(in_dir, mst)) return 1;
  if (out_dir && out_dir[0]) { save_genome_info(g, out_dir, "mst", byFile, kssd != 0); save_mst(mst, out_dir); }
  std::vector<rtc_edge> forest = generate_forest(mst, threshold);
  std::vector<std::vector<int>> cl = generate_cluster_with_bfs(forest, (int)g.size());
  print_result(cl, g, byFile, out_file, threshold);
  return 0;
}

int rtch_shuffle_dim(int half_subk, int32_t* out) {
  std::vector<int32_t> v = generate_shuffle_dim(half_subk);
  memcpy(out, v.data(), v.size() * sizeof(int32_t));
  return (int)v.size();
}

// returns 1 on success; outputs the tuned values
int rtch_tune(int greedy, int isSetKmer, int isContainment, int isJaccard, int kmerSize, double threshold,
              int containCompress, int sketchSize, uint64_t maxSize, uint64_t minSize, uint64_t avgSize, int* k_out,
              int* compress_out, int* containment_out) {
  bool ic = isContainment != 0;
  int k = kmerSize, cc = containCompress;
  bool ok = tune_parameters(greedy != 0, isSetKmer != 0, maxSize, minSize, avgSize, ic, isJaccard != 0, k, threshold, cc, sketchSize);
  *k_out = k; *compress_out = cc; *containment_out = ic ? 1 : 0;
  return ok ? 1 : 0;
}

int rtch_cal_size(const char* list_file, uint64_t minLen, uint64_t* mx, uint64_t* mn, uint64_t* avg) {
  return cal_size(list_file, minLen, *mx, *mn, *avg) ? 1 : 0;
}

int rtch_file_length(const char* path) { return file_length_for_containment(path); }

// The two genome readers of the sketch driver: mode 0 = read_genome_file (std::string), mode 1 =
// read_genome_file_flat into out[0..cap).  Writes the byte stream (records joined by '\n'), returns
// its length (the needed capacity when cap is too small for mode 1), -1 if the file cannot be opened.
long rtch_genome_bases(const char* path, int flat, char* out, long cap, uint64_t* total, uint64_t* nrec, int* first_len,
                       uint64_t* slot) {
  SequenceInfo first;
  *slot = genome_slot_bytes(path);
  if (flat) {
    uint64_t used = 0;
    const int st = read_genome_file_flat(path, out, (uint64_t)cap, used, first, *total, *nrec);
    if (st == 1) return -1;
    *first_len = first.length;
    return (long)used;
  }
  std::string bases;
  if (!read_genome_file(path, bases, first, *total, *nrec)) return -1;
  *first_len = first.length;
  if ((long)bases.size() <= cap) memcpy(out, bases.data(), bases.size());
  return (long)bases.size();
}

void rtch_pack_force_portable(int on) { pack_force_portable(on); }

// pack_bases: n characters -> ceil(n / 4) packed bytes in `out`, runs (start, length) pairs in runs_out (capacity
// runs_cap u64 values); returns the number of u64 values the runs take (the caller retries when it exceeds runs_cap)
long rtch_pack_bases(const char* seq, long n, unsigned char* out, unsigned long long* runs_out, long runs_cap) {
  std::vector<uint64_t> runs;
  pack_bases(seq, (size_t)n, out, runs);
  for (size_t i = 0; i < runs.size() && (long)i < runs_cap; i++) runs_out[i] = runs[i];
  return (long)runs.size();
}

// read_genome_file_packed: returns status (0 ok, 1 cannot open, 2 capacity); *used bases, *nruns u64 values written to runs_out
int rtch_read_genome_packed(const char* path, unsigned char* out, long cap_bases, long* used, unsigned long long* runs_out, long runs_cap,
                            long* nruns, unsigned long long* total, unsigned long long* nrec) {
  std::vector<uint64_t> runs; SequenceInfo first; uint64_t u = 0, tot = 0, nr = 0;
  const int st = read_genome_file_packed(path, out, (uint64_t)cap_bases, u, runs, first, tot, nr);
  *used = (long)u; *total = tot; *nrec = nr; *nruns = (long)runs.size();
  for (size_t i = 0; i < runs.size() && (long)i < runs_cap; i++) runs_out[i] = runs[i];
  return st;
}
}

// rtc_host.h -- host side of the drop-in: FASTA reading, parameter tuning, on-disk formats,
// cluster extraction and text output, mirroring the reference's interface for this path
// (names, argument meaning, error behaviour).  The compute goes through include/rtclust.h.
#pragma once
#include <stdint.h>

#include <ostream>
#include <string>
#include <vector>

#include "../../include/rtclust.h"

namespace rtc {

// == SequenceInfo / SketchInfo metadata, src/SketchInfo.h:14-39 (reference tree) ==
struct SequenceInfo {
  std::string name, comment;
  int strand = 0;
  int length = 0;
};

struct GenomeInfo {
  int id = 0;
  std::string fileName;          // list mode
  uint64_t totalSeqLength = 0;   // list mode: sum of record lengths
  SequenceInfo seq0;             // list mode: first record; sequence mode: the record
  bool use64 = false;            // KSSD only (kssd.info.* trailing byte)
};

// ---- FASTA / FASTQ reading with the semantics of klib kseq as used at src/SketchInfo.cpp:880-948 ----
struct FastaRecord {
  std::string name, comment;
  bool has_comment = false;  // false -> the reference substitutes "noName"
  std::string seq;
};
// Reads every record of `path` (plain or gzip).  Returns false if the file cannot be opened.
bool read_fasta(const std::string& path, std::vector<FastaRecord>& out);
// Streaming variant used by the sketch driver: appends the records' bases to `bases`, separated by
// '\n' (a non-ACGT byte, so k-mers never span records), and reports first-record metadata.
bool read_genome_file(const std::string& path, std::string& bases, SequenceInfo& first, uint64_t& total_len,
                      uint64_t& n_records);
// Zero-copy variant: writes the same byte stream into dst[0..cap) (e.g. a pinned staging slot).
// Returns 0 ok, 1 cannot open, 2 capacity too small (`used` then holds a capacity that suffices).
int read_genome_file_flat(const std::string& path, char* dst, uint64_t cap, uint64_t& used, SequenceInfo& first,
                          uint64_t& total_len, uint64_t& n_records);
// The same into the 2-bit packed staging format (a quarter of the bytes over PCIe): base i of the stream at bits
// 2 (i & 3) of dst[i >> 2], A/C/G/T (either case) = 0..3; every other character (N, IUPAC codes, the record
// separators) is stored as 0 and listed in `runs` as (start, length) pairs, ascending.  cap_bases: capacity of dst in
// bases (dst holds cap_bases / 4 bytes, cap_bases a multiple of 4).  rtc_unpack_bases_dev restores the ASCII stream
// ('N' over the runs) on the GPU.  Same return values as read_genome_file_flat.
int read_genome_file_packed(const std::string& path, uint8_t* dst, uint64_t cap_bases, uint64_t& used, std::vector<uint64_t>& runs,
                            SequenceInfo& first, uint64_t& total_len, uint64_t& n_records);
size_t pack_bases(const char* seq, size_t n, uint8_t* dst, std::vector<uint64_t>& runs);  // one buffer, for tests
void pack_force_portable(int on);  // tests: 1 = the portable loops only, 2 = at most AVX2, 0 = the best tier the CPU has (AVX-512 BW + VBMI2, AVX2, portable)
// Upper bound of the bytes read_genome_file_flat writes for `path` (exact bound for plain files, the
// ISIZE-based guess for gzip); 0 if the file cannot be opened.
uint64_t genome_slot_bytes(const std::string& path);

// ---- calSize / tune_parameters / tune_kssd_parameters, src/SketchInfo.cpp:438-552, src/sub_command.cpp:2317-2467 ----
bool cal_size(const std::string& list_file, uint64_t minLen, uint64_t& maxSize, uint64_t& minSize, uint64_t& averageSize);
bool tune_parameters(bool greedy, bool isSetKmer, uint64_t maxSize, uint64_t minSize, uint64_t averageSize,
                     bool& isContainment, bool isJaccard, int& kmerSize, double threshold, int& containCompress,
                     int sketchSize);
bool tune_kssd_parameters(bool isSetKmer, uint64_t maxSize, uint64_t minSize, uint64_t averageSize, bool isContainment,
                          int& kmerSize, double threshold, int drlevel);
// file size the reference uses for containment sketch sizes (gz: ISIZE trailer), src/SketchInfo.cpp:892-915
int file_length_for_containment(const std::string& path);

// ---- generate_shuffle_dim, src/SketchInfo.cpp:60-102 (glibc srand/rand) ----
std::vector<int32_t> generate_shuffle_dim(int half_subk);

// ---- on-disk formats (src/Sketch_IO.cpp, src/MST_IO.cpp, src/SketchInfo.h:115-160) ----
struct KssdParameters { int id, half_k, half_subk, drlevel, genomeNumber; };  // src/SketchInfo.h:50-56

void save_genome_info(const std::vector<GenomeInfo>& g, const std::string& folder, const std::string& type,
                      bool sketchByFile, bool kssd);
bool load_genome_info(const std::string& folder, const std::string& type, std::vector<GenomeInfo>& g, bool kssd,
                      bool& sketchByFile);

struct MinHashSketchFile {
  int kmerSize = 21;
  bool isContainment = false;
  int containCompress = 1000, sketchSize = 1000;
  std::vector<std::vector<uint64_t>> hashes;
};
void save_minhash_sketches(const std::vector<GenomeInfo>& g, const MinHashSketchFile& f, const std::string& folder,
                           bool sketchByFile);
bool load_minhash_sketches(const std::string& folder, std::vector<GenomeInfo>& g, MinHashSketchFile& f, bool& sketchByFile);
void save_minhash_index(const MinHashSketchFile& f, const std::string& folder);  // minhash.sketch.index (MHIDX001)

struct KssdSketchFile {
  KssdParameters info{};
  bool use64 = false;
  std::vector<std::vector<uint32_t>> h32;
  std::vector<std::vector<uint64_t>> h64;
};
void save_kssd_sketches(const std::vector<GenomeInfo>& g, const KssdSketchFile& f, const std::string& folder, bool sketchByFile);
bool load_kssd_sketches(const std::string& folder, std::vector<GenomeInfo>& g, KssdSketchFile& f, bool& sketchByFile);
void save_kssd_index(const KssdSketchFile& f, const std::string& folder);  // kssd.sketch.index + .dict

// cluster_state.bin of clust-greedy --fast --save-rep (KssdClusterState::save / ::load, src/greedy.cpp:1545-1734):
// threshold, k, KSSD parameters, representative ids, every sketch in clustering order (id, length, hashes,
// file name), the clusters, and the representatives' inverted index (hash -> positions in rep_ids; written from
// the sketches, skipped when read -- it is rebuilt from them wherever it is needed).
struct KssdClusterState {
  double threshold = 0.05;
  int kmer_size = 0;
  KssdParameters info{};
  std::vector<int> rep_ids;
  std::vector<GenomeInfo> genomes;   // state order: genome i has id i (RepDB: file name and length only)
  KssdSketchFile sk;                 // hashes in the same order (cluster_state.bin; empty in a RepDB)
  std::vector<GenomeInfo> rep_genomes;  // representative r: id, length, file name ...
  KssdSketchFile reps;                  // ... and hashes, r = position in rep_ids
  std::vector<std::vector<int>> clusters;  // clusters[r] belongs to representative r
  // MinHash RepDB (MinHashClusterState, src/greedy.h): u64 hashes in reps.h64, these instead of the KSSD parameters
  bool minhash = false;
  int sketch_size = 0;
  bool is_containment = false;
};
bool save_kssd_cluster_state(const std::string& path, const KssdClusterState& st);
bool load_kssd_cluster_state(const std::string& path, KssdClusterState& st);
// cluster_state.bin of clust-greedy --save-rep on MinHash sketches (MinHashClusterState::save / ::load,
// src/greedy.cpp:2134-2302): "MINHASH", parameters, representative ids, every sketch, clusters (representative first),
// index.  load keeps parameters, representative ids and clusters only -- the reference skips the sketches too and
// re-reads them from the folder.
bool save_minhash_cluster_state(const std::string& path, const KssdClusterState& st);
bool load_minhash_cluster_state(const std::string& path, KssdClusterState& st);
// RepDB of clust-greedy --fast --db (KssdClusterState::save_repdb / ::load_repdb / ::print_stats,
// src/greedy.cpp:2351-2537, :2656-2765): "REPDB002", parameters, the representatives with their sketches, the
// clusters, every genome's file name and length, the representatives' inverted index (64-bit keys; "REPDB001"
// files with 32-bit keys are read too).  The index is written from the sketches and not kept when read.
bool save_kssd_repdb(const std::string& path, const KssdClusterState& st);
bool load_kssd_repdb(const std::string& path, KssdClusterState& st);
void print_kssd_repdb_stats(const KssdClusterState& st, std::ostream& out);
// MinHash twin (MinHashClusterState::save_repdb / ::load_repdb / ::print_stats, src/greedy.cpp:2789-3147): "MHREPDB1"
bool save_minhash_repdb(const std::string& path, const KssdClusterState& st);
bool load_minhash_repdb(const std::string& path, KssdClusterState& st);

void save_mst(const std::vector<rtc_edge>& mst, const std::string& folder);   // edge.mst
bool load_mst(const std::string& folder, std::vector<rtc_edge>& mst);

// ---- --dense by-products: mst.dense / mst.ani (src/MST_IO.cpp:12-45, :219-250) and the noise-removal
// pass of compute_clusters (src/sub_command.cpp:3071-3103; getNoiseNode / modifyForest, src/MST.cpp:86-107,189-211) ----
constexpr int DENSE_SPAN = 100;  // src/common.hpp
void save_dense(const std::string& folder, const std::vector<int32_t>& dense, int span, int genome_number);  // dense: span x n row-major
bool load_dense(const std::string& folder, std::vector<int32_t>& dense, int& span, int& genome_number);
void save_ani(const std::string& folder, const uint64_t ani[101]);
bool load_ani(const std::string& folder, uint64_t ani[101]);
// nodes of every multi-member cluster whose density at the threshold's bucket is <= min(Q1 - 1, alpha = 2)
std::vector<int> noise_nodes(const std::vector<std::vector<int>>& cluster, const std::vector<int32_t>& dense, int span,
                             int genome_number, double threshold);
std::vector<rtc_edge> modify_forest(const std::vector<rtc_edge>& forest, const std::vector<int>& noise);

// ---- forest cut, BFS clusters, result text (src/MST.cpp:77-85,109-142; src/MST_IO.cpp:72-179) ----
// kruskalAlgorithm over a list already sorted by distance (src/MST.cpp:59-75, UnionFind.h:5-90): used to
// merge a stored MST with the forest of the appended rows (append_clust_mst, src/sub_command.cpp:1693-1700)
std::vector<rtc_edge> kruskal_algorithm(const std::vector<rtc_edge>& sorted_graph, int vertices);
std::vector<rtc_edge> generate_forest(const std::vector<rtc_edge>& mst, double threshold);
std::vector<std::vector<int>> generate_cluster_with_bfs(const std::vector<rtc_edge>& forest, int vertices);
void print_result(const std::vector<std::vector<int>>& cluster, const std::vector<GenomeInfo>& g, bool sketchByFile,
                  const std::string& outputFile, double threshold = -1.0);

// ---- tree / linkage writers of clust-mst (src/MST.cpp:1044-1287, src/MST_IO.cpp:252-380): the single-linkage
// dendrogram of the MST (edges by ascending distance, heights = merge distances).  As in the reference only
// the component that contains genome 0 is written when the MST is a forest. ----
std::string get_newick_tree(const std::vector<GenomeInfo>& g, const std::vector<rtc_edge>& mst, bool sketchByFile);
void print_newick_tree(const std::vector<GenomeInfo>& g, const std::vector<rtc_edge>& mst, bool sketchByFile, const std::string& output);
void print_phylip_tree(const std::vector<GenomeInfo>& g, const std::vector<rtc_edge>& mst, bool sketchByFile, const std::string& output);
void print_nexus_tree(const std::vector<GenomeInfo>& g, const std::vector<rtc_edge>& mst, bool sketchByFile, const std::string& output);
void print_linkage_matrix(int n, const std::vector<rtc_edge>& mst, const std::string& output);  // c1 \t c2 \t dist \t size

std::string current_date_time();  // src/common.hpp:36-44

// Time the parser threads spent inside gzip decompression (libdeflate or zlib), summed over threads, and the bytes it produced
// since the process started: the command lines report them (RTC_METRICS_JSON: inflate_gb_per_s_per_thread).
void rtc_host_inflate_stats(double* seconds, uint64_t* bytes_out);

}  // namespace rtc

/*
 * rtc_oracle.h -- CPU restatement of RabbitTClust's sketch + all-pairs-distance hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and only as the
 * checker / reported baseline.  The product (rabbittclust_amd/) never links or calls it.
 *
 * PARITY STATUS
 *   - MinHash k-mer hash (S2): PARITY UNPINNED.  The arithmetic lives in the RabbitSketch
 *     submodule (.gitmodules:4-6), which is an empty directory in /root/reference with no
 *     recoverable pinned commit, and the reference ships no tests / golden vectors.  We restate
 *     the published Mash/RabbitSketch algorithm (MurmurHash3_x64_128 seed 42 over the canonical
 *     ASCII k-mer, bottom-s distinct hashes) and pin only our MurmurHash3 against the public
 *     SMHasher verification constant 0x6384BA69.
 *   - Everything downstream of the hash values (KSSD sketch, intersection counts, distances,
 *     MST, greedy, file formats) restates code that IS in the tree, function by function with
 *     file:line citations.  The reference cannot be compiled here without writing a stand-in for
 *     the absent Sketch.h (every src .cpp file includes it through SketchInfo.h), so there is no
 *     oracle/_ref build of those files; the only reference sources that compile on their own
 *     (kseq.h, UnionFind.h) are built into oracle/_ref/ref_harness and checked against this file.
 *
 * Written in C++17 with a C ABI (not plain C) so that std::sort / glibc rand() behave exactly as
 * they do inside the reference binary (tie order of equal-distance edges, KSSD shuffle table).
 */
#ifndef RTC_ORACLE_H
#define RTC_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- MurmurHash3_x64_128 (public domain algorithm, Austin Appleby) ---- */
void orc_murmur3_x64_128(const void* key, int len, uint32_t seed, uint64_t out[2]);
/* SMHasher VerificationTest value for a 128-bit hash (expects 0x6384BA69). */
uint32_t orc_murmur3_smhasher_verification(void);

/* ---- MinHash (restates Sketch::MinHash as used at src/SketchInfo.cpp:918-942) ---- */
typedef struct orc_minhash orc_minhash;
orc_minhash* orc_mh_new(int k, uint32_t sketch_size, uint32_t seed);
void orc_mh_free(orc_minhash*);
/* one FASTA record (k-mers never span records, src/SketchInfo.cpp:928-948) */
void orc_mh_update(orc_minhash*, const char* seq, uint64_t len);
/* ascending distinct hashes; returns count (<= sketch_size) */
uint32_t orc_mh_store(const orc_minhash*, uint64_t* out, uint32_t cap);
/* k-mers hashed per call in orc_mh_update: 8 (AVX-512 vpmullq), 4 (AVX2) or 1 (scalar) -- our own
 * vectorisations of the same arithmetic; RabbitSketch's AVX2/AVX-512 kernels are absent from the
 * reference tree.  Chosen at run time from the CPU's features. */
int orc_minhash_impl_avx2(void);
/* hash of one canonical k-mer given as ASCII (must be k valid ACGT chars) */
uint64_t orc_mh_kmer_hash(const char* kmer, int k, uint32_t seed);

/* Batch driver: genomes concatenated, records of one genome separated by any non-ACGT byte.
 * off[n+1] byte offsets.  sizes[i] = sketch size of genome i.  out is n*stride u64, cnt[n].
 * threads<=0 -> all cores (OpenMP over genomes, like src/SketchInfo.cpp:878). */
void orc_sketch_minhash_batch(const uint8_t* seq, const uint64_t* off, uint32_t n, int k,
                              uint32_t seed, const uint32_t* sizes, uint64_t* out, uint32_t stride,
                              uint32_t* cnt, int threads);

/* ---- KSSD (restates src/SketchInfo.cpp:60-102, 994-1193) ---- */
typedef struct {
  int half_k, half_subk, drlevel, kmer_size; /* kmer_size = 2*half_k */
  int use64;
  int dim_size, dim_end;
  int id; /* (half_k<<8)+(half_subk<<4)+drlevel, src/SketchInfo.cpp:1030 */
} orc_kssd_params;
void orc_kssd_params_init(int kmer_size, int drlevel, orc_kssd_params* p);
/* glibc srand/rand shuffle table (src/SketchInfo.cpp:60-102); caller frees with free(). */
int* orc_kssd_shuffle_dim(int half_subk);
/* sketch one genome given as records separated by non-ACGT bytes.  Output ascending distinct.
 * out64 used when p->use64 else out32.  Returns number of hashes (may exceed cap: then only
 * cap written). */
uint64_t orc_kssd_sketch(const orc_kssd_params* p, const int* shuffled_dim, const uint8_t* seq,
                         uint64_t len, uint32_t* out32, uint64_t* out64, uint64_t cap);

/* ---- pair intersection + distances ---- */
/* Mash's union-truncated estimator (MinHash::jaccard(), dense loop src/MST.cpp:851-866; [U], unpinned) */
void orc_mash_counts_u64(const uint64_t* a, uint32_t na, const uint64_t* b, uint32_t nb, uint32_t sketch_size,
                         uint32_t* common, uint32_t* denom);
uint32_t orc_common_u64(const uint64_t* a, uint32_t na, const uint64_t* b, uint32_t nb);
uint32_t orc_common_u32(const uint32_t* a, uint32_t na, const uint32_t* b, uint32_t nb);
/* src/MST.cpp:1489-1503 (Jaccard -> Mash) and :1504-1515 (containment -> AAF); no >1 clamp */
double orc_mst_distance(int common, int size0, int size1, int kmer_size, int is_containment);
/* radio filter src/MST.cpp:1292,1481-1484: (int)(2*exp(thr*(k-1))-1) */
int orc_mst_radio(double threshold, int kmer_size);
/* greedy slow path src/greedy.cpp:1240-1282 (Mash formula on containment, clamped to 1) */
double orc_greedy_distance(int common, int size_ref, int size_qry, int kmer_size, int rep_is_containment);
/* src/greedy.cpp:526-543 */
double orc_kssd_greedy_distance(int common, int size0, int size1, int kmer_size);

typedef struct { int pre, suf; double dist; } orc_edge; /* == EdgeInfo, src/MST.h:17-21 */
typedef struct { int pre, suf; uint32_t common; } orc_cedge;

/* all candidate pairs with common>0 (j<i), via the inverted index, before any filter.
 * width 4|8.  start[n], len[n] index into hashes.  Returns number written (sorted by (i,j)). */
uint64_t orc_candidate_pairs(const void* hashes, int width, const uint64_t* start,
                             const uint32_t* len, uint32_t n, orc_cedge* out, uint64_t cap);

/* compute_minhash_mst / compute_kssd_mst restated (src/MST.cpp:1290-1737, 216-807):
 * index-based intersections, j<i, radio filter, distance, per-8-row sort+Kruskal, final
 * sort+Kruskal.  threads>=1 (threads==1 reproduces the reference -t 1 edge order exactly).
 * kssd32_skip_singletons mirrors :472.  Returns number of MST edges written to out (cap>=n). */
uint64_t orc_mst(const void* hashes, int width, const uint64_t* start, const uint32_t* len,
                 uint32_t n, int kmer_size, int is_containment, double threshold, int threads,
                 orc_edge* out);
/* kruskalAlgorithm (src/MST.cpp:59-75) on a pre-sorted list; returns forest size */
uint64_t orc_kruskal(const orc_edge* sorted, uint64_t m, int vertices, orc_edge* out);
/* generateForest + generateClusterWithBfs (src/MST.cpp:77-85,109-142): labels[v]=cluster id in
 * BFS discovery order; order[] = members concatenated in BFS order; cl_off[nclust+1]. Returns nclust */
uint32_t orc_forest_clusters(const orc_edge* mst, uint64_t m, double threshold, int vertices,
                             int* order, uint32_t* cl_off);

/* ---- greedy (restates src/greedy.cpp:986-1399 at -t 1, and :566-899) ----
 * sketch_size_cfg[i] = what rep->getSketchSize() returns (configured size, :1201);
 * is_containment applies to all genomes.  Output: rep_of[i] = representative id (self if rep),
 * in the given order.  Returns number of clusters. */
uint32_t orc_greedy_minhash(const uint64_t* hashes, const uint64_t* start, const uint32_t* len,
                            const uint32_t* sketch_size_cfg, uint32_t n, int kmer_size,
                            int is_containment, double threshold, int* rep_of);
/* KSSD greedy on ALREADY size-sorted input (caller applies the std::sort of :594-597) */
uint32_t orc_greedy_kssd(const void* hashes, int width, const uint64_t* start, const uint32_t* len,
                         uint32_t n, int kmer_size, double threshold, int* rep_of);

/* ---- tune_parameters (src/sub_command.cpp:2383-2467) ---- */
typedef struct {
  int kmer_size; int contain_compress; int is_containment; int ok; double max_dist;
} orc_tune_result;
orc_tune_result orc_tune_parameters(int greedy, int is_set_kmer, int is_containment, int is_jaccard,
                                    int kmer_size, double threshold, int contain_compress,
                                    int sketch_size, uint64_t max_size, uint64_t min_size,
                                    uint64_t avg_size);

/* ---- deterministic synthetic genomes (shared definition with the device generator) ---- */
typedef struct {
  uint64_t fam_seed;   /* seed of the family ancestor sequence */
  uint64_t mut_seed;   /* seed of this member's substitution stream */
  uint32_t mut_thr;    /* substitute where 14-bit draw < mut_thr  (rate = mut_thr/16384) */
  uint32_t n_every;    /* 0: none; else a run of 'N' (length 8) every n_every bases */
} orc_synth_desc;
void orc_synth_genome(const orc_synth_desc* d, uint64_t pos0, uint64_t len, uint8_t* out);

#ifdef __cplusplus
}
#endif
#endif

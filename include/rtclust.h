/*
 * rtclust.h -- C ABI of the MI355X-native sketch + all-pairs-distance path of RabbitTClust.
 *
 * The reference has no FFI; the path sits behind the RabbitSketch C++ class API and the
 * intermediate-folder file formats.  Each entry point below names the reference
 * code it replaces (paths relative to the RabbitTClust tree).  Plain pointers and sizes only;
 * every function returns an rtc_status; no exceptions cross the boundary.
 *
 * Pointer naming: d_* = device (HBM) pointer, h_* = host pointer.  All device work is enqueued
 * on the context's HIP stream (rtc_ctx_set_stream); *_dev entry points do not synchronise unless
 * stated.  A context is bound to one GPU and may be used by one host thread at a time.
 */
#ifndef RTCLUST_H
#define RTCLUST_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  RTC_OK = 0,
  RTC_ERR_ARG = 1,         /* bad argument (null pointer, k out of range, misaligned buffer) */
  RTC_ERR_HIP = 2,         /* HIP runtime failure; see rtc_last_error() */
  RTC_ERR_UNSUPPORTED = 3, /* valid request outside what the GPU path implements */
  RTC_ERR_OVERFLOW = 4,    /* caller-provided output capacity too small; required size reported */
  RTC_ERR_NOMEM = 5,
  RTC_ERR_COMM = 6         /* a collective did not complete within RTC_COMM_TIMEOUT_S (default 120 s) or the communicator was
                              aborted after one: every later call on that communicator returns this at once */
} rtc_status;

typedef struct rtc_ctx rtc_ctx;

/* ---- context ------------------------------------------------------------------------- */
int rtc_device_count(void); /* visible GPUs (0 when there is none or the runtime fails) */
int rtc_ctx_create(int device, rtc_ctx** out);
void rtc_ctx_destroy(rtc_ctx* ctx);
/* Loads the device code of the pair / MST / greedy phases ahead of their first use: the HIP runtime maps a
 * translation unit's code object at its first kernel launch (~27 ms for these phases together on MI355X), which a
 * one-shot command line would otherwise pay between sketching and clustering.  Runs a complete toy clustering on a
 * context of its own; meant for a helper thread beside the sketch phase (the command lines do that).  Thread-safe
 * against work on other contexts. */
int rtc_warmup(int device);
/* The library's switches (RTC_PAIR_JOIN, RTC_EDGE_BUDGET, RTC_SKETCH_T0_FACTOR, RTC_COMM_TIMEOUT_S ...: README) are read from
 * the environment once, by rtc_ctx_create; no call looks at the environment again.  This reads them anew for a context
 * that is already there (tests and tuning runs that change a switch between calls). */
int rtc_ctx_reload_options(rtc_ctx* ctx);
int rtc_ctx_set_stream(rtc_ctx* ctx, void* hip_stream); /* NULL = default stream */
/* Gives the context a non-blocking stream of its own: two contexts on one device, each driven by its
 * own host thread, then overlap (the command lines copy batch i+1 while batch i is sketched). */
int rtc_ctx_own_stream(rtc_ctx* ctx);
int rtc_ctx_sync(rtc_ctx* ctx);
const char* rtc_last_error(const rtc_ctx* ctx); /* ctx may be NULL: last context-less failure */
const char* rtc_version(void);
/* device properties: out[0]=CU count, out[1]=LDS bytes per workgroup, out[2]=wavefront size */
int rtc_device_info(rtc_ctx* ctx, int out[3]);

/* device memory for hosts that do not link a HIP runtime themselves (the C++ CLI) */
int rtc_dev_alloc(rtc_ctx* ctx, size_t bytes, void** d_ptr);
int rtc_dev_free(rtc_ctx* ctx, void* d_ptr);
/* free / total HBM of the context's GPU in bytes (the command lines size their resident sketch rows against it) */
int rtc_dev_mem_info(rtc_ctx* ctx, size_t* free_bytes, size_t* total_bytes);
int rtc_copy_h2d(rtc_ctx* ctx, void* d_dst, const void* h_src, size_t bytes); /* synchronous */
int rtc_copy_d2h(rtc_ctx* ctx, void* h_dst, const void* d_src, size_t bytes); /* synchronous */
int rtc_memset_dev(rtc_ctx* ctx, void* d_ptr, int value, size_t bytes);
/* page-locked host staging memory: the CLI parses FASTA files straight into it (the reference's
 * per-thread kseq buffers, src/SketchInfo.cpp:880-948) so the PCIe copy runs at link speed */
int rtc_host_alloc(rtc_ctx* ctx, size_t bytes, void** h_ptr);
int rtc_host_free(rtc_ctx* ctx, void* h_ptr);

/* ---- 2-bit packed staging (command lines) -------------------------------------------------- */
/* The reference feeds the sketcher ASCII records (src/SketchInfo.cpp:928-948).  The command lines send a quarter of
 * that over PCIe: base i of a batch at bits 2 (i & 3) of d_packed[i >> 2] with A, C, G, T = 0..3, and everything that
 * is not ACGT (N, IUPAC codes, record separators, the gaps between genomes) as d_runs[2 r] = start, d_runs[2 r + 1] =
 * length.  This call writes the ASCII stream the sketch kernels read to d_seq[0 .. n_bases): "ACGT"[code], 'N' over the
 * runs.  n_bases a multiple of 64, both buffers 16-byte aligned.  Context stream, asynchronous. */
int rtc_unpack_bases_dev(rtc_ctx* ctx, const uint8_t* d_packed, uint64_t n_bases, const uint64_t* d_runs, uint64_t n_runs,
                         uint8_t* d_seq);

/* ---- timing of the last launches (HIP events on the context stream) -------------------- */
/* Brackets subsequently enqueued work; rtc_timer_stop synchronises and returns milliseconds. */
int rtc_timer_start(rtc_ctx* ctx);
int rtc_timer_stop(rtc_ctx* ctx, float* ms_out);

/* ---- synthetic genomes (benchmark / test input) -------------------------- */
typedef struct {
  uint64_t fam_seed; /* ancestor stream */
  uint64_t mut_seed; /* this member's substitution stream */
  uint32_t mut_thr;  /* substitute where a 14-bit draw < mut_thr (rate = mut_thr/16384) */
  uint32_t n_every;  /* 0: none; else an 8-base run of 'N' every n_every bases */
} rtc_synth_desc;
/* Writes genome g's bases (ASCII ACGT/N) to d_seq[h_off[g] .. h_off[g+1]). */
int rtc_synth_genomes_dev(rtc_ctx* ctx, const rtc_synth_desc* h_desc, const uint64_t* h_off,
                          uint32_t n, uint8_t* d_seq);

/* ---- MinHash sketching ----------------------------------------------------------------- */
/* Replaces, for a batch of genomes, `new Sketch::MinHash(k, size)` + `update(seq)` per FASTA
 * record + `storeMinHashes()`  (src/SketchInfo.cpp:918-924, :942, :969; RabbitSketch library).
 * d_seq: concatenated genomes, 16-byte aligned; records of one genome are separated by any
 *   non-ACGT byte (k-mers never span records); lower case is folded to upper.
 * h_off[n+1]: byte offsets of the genomes in d_seq.  h_sizes[n]: sketch size per genome
 *   (fixed-size mode: all equal; containment mode: max(fileBytes/compress,100),
 *   src/SketchInfo.cpp:919-924), or NULL to use `size` for all.
 * d_out: n * stride u64; genome g's ascending distinct hashes at d_out + g*stride;
 * d_cnt[n]: number of hashes produced (< size only when the genome has fewer distinct k-mers).
 * Hash: first 64 bits of MurmurHash3_x64_128(canonical k-mer ASCII, k, seed) for k > 16,
 * first 32 bits for k <= 16 (Mash / RabbitSketch convention). 1 <= k <= 32. */
int rtc_sketch_minhash_dev(rtc_ctx* ctx, const uint8_t* d_seq, const uint64_t* h_off, uint32_t n,
                           int k, uint32_t seed, const uint32_t* h_sizes, uint32_t size,
                           uint64_t* d_out, uint32_t stride, uint32_t* d_cnt);

/* The same sketches straight from a batch in the 2-bit staging format (layout: rtc_unpack_bases_dev above) -- the records
 * `update()` is handed one by one (src/SketchInfo.cpp:928-948) as they crossed PCIe, 0.25 B per base read once, no ASCII
 * copy in HBM.  d_packed (16-byte aligned) holds n_bases / 4 bytes, n_bases a multiple of 64 and >= h_off[n];
 * d_runs[2 r], d_runs[2 r + 1] = start and length of run r of characters outside ACGT, ascending by start and disjoint
 * (record separators and the gaps between genomes are runs too): a k-mer counts exactly when none of its k characters
 * lies in a run nor outside its genome's [h_off[g], h_off[g + 1]) -- what update() does with a character outside ACGT.
 * Every k in 1..32 and every sketch size; everything else as rtc_sketch_minhash_dev, whose results it reproduces bit
 * for bit.  The run list's contract is checked on the device beside the sketching (no host round trip): a violation is
 * reported as RTC_ERR_ARG by rtc_ctx_sync -- call it before the sketches of a packed batch are consumed -- or, failing
 * that, by the next packed call on the context. */
int rtc_sketch_minhash_packed_dev(rtc_ctx* ctx, const uint8_t* d_packed, uint64_t n_bases, const uint64_t* d_runs,
                                  uint64_t n_runs, const uint64_t* h_off, uint32_t n, int k, uint32_t seed,
                                  const uint32_t* h_sizes, uint32_t size, uint64_t* d_out, uint32_t stride,
                                  uint32_t* d_cnt);

/* ---- KSSD sketching (--fast) ------------------------------------------------------------- */
/* Replaces the per-file body of sketchFileWithKssd (src/SketchInfo.cpp:994-1252): 2-bit rolling
 * k-mer (k rounded up to even, :1019-1020), canonical min, shuffled-dimension filter, dr_tuple,
 * dedup, ascending sort.  h_shuffled_dim: the 2^(4*half_subk) table of generate_shuffle_dim
 * (:91-102; built on the host with the same glibc srand/rand calls).
 * width_out: 4 (u32 hashes) or 8 (u64) as decided by half_k - drlevel > 8 (:1021).
 * d_out: n * stride elements of that width; d_cnt[n] = hashes per genome.
 * Returns RTC_ERR_OVERFLOW if some genome yields more than `stride` hashes; h_need (optional)
 * then holds the required stride. */
int rtc_sketch_kssd_dev(rtc_ctx* ctx, const uint8_t* d_seq, const uint64_t* h_off, uint32_t n,
                        int kmer_size, int drlevel, const int32_t* h_shuffled_dim, void* d_out,
                        uint32_t stride, uint32_t* d_cnt, int* width_out, uint32_t* h_need);
/* The same sketches straight from a batch in the 2-bit staging format (see rtc_unpack_bases_dev for the layout): the
 * records sketchFileWithKssd walks (src/SketchInfo.cpp:1120-1166) as they crossed PCIe, 0.25 B per base read once, no
 * ASCII copy in HBM.  d_packed (16-byte aligned) holds n_bases / 4 bytes, n_bases a multiple of 64; d_runs[2 r],
 * d_runs[2 r + 1] = start and length of run r of characters outside ACGT, ascending by start and disjoint -- a k-mer
 * counts exactly when none of its characters lies in a run (:1136-1139, :1160-1164) nor outside its genome's
 * [h_off[g], h_off[g + 1]).  Everything else as rtc_sketch_kssd_dev, whose results it reproduces bit for bit.
 * Returns RTC_ERR_UNSUPPORTED outside the prefilter kernel's configurations (17 <= kmer_size <= 28 with half_subk = 6,
 * i.e. drlevel 3 (the default) or 4: at most 4 096 kept dimensions): callers then expand the batch with rtc_unpack_bases_dev and call rtc_sketch_kssd_dev. */
int rtc_sketch_kssd_packed_dev(rtc_ctx* ctx, const uint8_t* d_packed, uint64_t n_bases, const uint64_t* d_runs,
                               uint64_t n_runs, const uint64_t* h_off, uint32_t n, int kmer_size, int drlevel,
                               const int32_t* h_shuffled_dim, void* d_out, uint32_t stride, uint32_t* d_cnt,
                               int* width_out, uint32_t* h_need);

/* ---- all-pairs sorted-sketch intersection ----------------------------------------------- */
/* common[i][j] = |A_i ∩ A_j| for i in [row0,row1), j in [col0,col1): the integers that
 * compute_minhash_mst / compute_kssd_mst obtain through their inverted index
 * (src/MST.cpp:1408-1435, :428-487) and that modifyMST's distance()/jaccard() calls derive
 * (src/MST.cpp:851-866).  Sketches: `d_hashes` (u64 when width==8, u32 when width==4), genome g
 * occupies d_hashes[d_start[g] .. d_start[g]+d_len[g]) ascending and distinct.
 * d_common: (row1-row0) x ld u32, row-major.  lower_only != 0: only entries with j < i are
 * defined (others are left untouched).  algo: 0 = auto, 1 = per-pair merge (generic),
 * 2 = LDS mask-table tiles. */
int rtc_pair_common_dev(rtc_ctx* ctx, const void* d_hashes, int width, const uint64_t* d_start,
                        const uint32_t* d_len, uint32_t n, uint32_t row0, uint32_t row1,
                        uint32_t col0, uint32_t col1, uint32_t* d_common, uint64_t ld,
                        int lower_only, int algo);

/* The dense loop's estimator: what Sketch::MinHash::jaccard()/distance() hand to modifyMST
 * (src/MST.cpp:851-866) is Mash's union-truncated Jaccard, NOT the set-Jaccard of the index path: merge
 * the two ascending lists, stop after `sketch_size` elements of the union; d_common = shared elements
 * among them, d_denom = union elements seen (sketch_size unless both lists run out); distance on the host
 * = -ln(2j/(1+j))/k with j = common/denom.  RabbitSketch is absent from the reference tree: this restates
 * the published Mash algorithm and is parity-unpinned like the k-mer hash. */
int rtc_pair_mash_dev(rtc_ctx* ctx, const void* d_hashes, int width, const uint64_t* d_start,
                      const uint32_t* d_len, uint32_t n, uint32_t sketch_size, uint32_t row0, uint32_t row1,
                      uint32_t col0, uint32_t col1, uint32_t* d_common, uint32_t* d_denom, uint64_t ld);

/* ---- candidate edges ----------------------------------------------------------------------- */
typedef struct { uint32_t i, j, common; } rtc_cedge; /* i > j */
/* Scans the common matrix produced above and appends every pair the reference would turn into
 * an EdgeInfo: j < i, common > 0, both sketches non-empty, max(|A|,|B|) <= radio*min(|A|,|B|)
 * (src/MST.cpp:1468-1487; radio = (int)(2*exp(threshold*(k-1))-1), :1292).  d_count is a u64
 * counter the caller zeroes; edges beyond `cap` are counted but not stored. */
int rtc_extract_edges_dev(rtc_ctx* ctx, const uint32_t* d_common, uint64_t ld, uint32_t row0,
                          uint32_t row1, uint32_t col0, uint32_t col1, const uint32_t* d_len,
                          int radio, rtc_cedge* d_edges, uint64_t cap, uint64_t* d_count);

/* Fused form of the two calls above for the tile rows [row0,row1) x cols [col0,col1): the
 * surviving (i, j, common), j < i, of the same filters (src/MST.cpp:1468-1487) are appended
 * directly -- no dense matrix is written.  radio < 0 disables the size-ratio test (greedy
 * clustering filters on the host).  d_count as above.  Two device paths with identical results:
 * the inverted join (the reference's index, src/MST.cpp:1408-1435, as a device sort of
 * (hash, genome) + a count of every column's partner lists in on-chip tables; cost ~ hashes +
 * co-occurrences) where the tile is sparse enough for it to win, otherwise the tiled kernel
 * (cost ~ rows x cols x s / 64, independent of the data).  RTC_PAIR_JOIN=0 in the environment
 * disables the join, =2 takes it wherever its scratch fits.
 * Overflow protocol: a count beyond `cap` on return means the list was too short -- grow it to at least the count and call
 * again from the old count.  When the join's density sample says that a list is too short for the set before the tiled
 * kernel has run, the count comes back as an ESTIMATE above `cap` with nothing appended (one launch saved); the repeated
 * call always runs to the end and returns the exact count. */
int rtc_pair_edges_dev(rtc_ctx* ctx, const void* d_hashes, int width, const uint64_t* d_start,
                       const uint32_t* d_len, uint32_t n, uint32_t row0, uint32_t row1, uint32_t col0,
                       uint32_t col1, int radio, rtc_cedge* d_edges, uint64_t cap, uint64_t* d_count);
/* Which path the last rtc_pair_edges_dev of this context took: 0 none yet, 1 per-pair merge kernel,
 * 2 tiled kernel, 3 inverted join (measurement: bench.py names the kernels of the pair phase by it). */
int rtc_pair_last_path(const rtc_ctx* ctx);
/* Which paths this context has taken since it was created (tests and measurement): out[0] tiles the inverted join took,
 * out[1] tiles the tiled kernel took, out[2] tiles of the merge kernel, out[3] candidate lists contracted to their forest
 * between row chunks, out[4] greedy runs replayed from one global join, out[5] query blocks of greedy's block loop, out[6]
 * estimates rtc_pair_edges_dev handed back instead of a launch; out[7] reserved. */
int rtc_diag_counters(const rtc_ctx* ctx, uint64_t out[8]);
/* Duration of this context's last tiled pair kernel launch (rtc_pair_last_path == 2), from HIP events recorded on the
 * stream it was launched on; waits for the launch to finish (measurement: bench.py's roofline_dist). */
int rtc_pair_last_kernel_ms(rtc_ctx* ctx, float* ms_out);

/* ---- minimum spanning forest over candidate edges (Boruvka, order-exact integer weights) -- */
/* One Boruvka round primitive for row-sharded multi-GPU use: for every current component c
 * (d_comp[v] = component label of vertex v) computes the minimum key over the local edges that
 * leave c.  Pass 1 (d_wkey): weight key = bit pattern of the exact rational similarity order
 * (see DESIGN.md); pass 2 (d_ekey): (i<<32|j) among edges attaining d_wkey.  Between the passes
 * the caller all-reduces (MIN) d_wkey across ranks; after pass 2 it all-reduces d_ekey. */
int rtc_boruvka_minweight_dev(rtc_ctx* ctx, const rtc_cedge* d_edges, uint64_t m,
                              const uint32_t* d_len, int is_containment, const uint32_t* d_comp,
                              uint32_t n, uint64_t* d_wkey);
int rtc_boruvka_minedge_dev(rtc_ctx* ctx, const rtc_cedge* d_edges, uint64_t m,
                            const uint32_t* d_len, int is_containment, const uint32_t* d_comp,
                            uint32_t n, const uint64_t* d_wkey, uint64_t* d_ekey);

/* After d_ekey is final (all-reduced), the rank owning each winning edge publishes its `common`
 * into d_ecommon[component] (others leave 0; all-reduce(MAX) across ranks). */
int rtc_boruvka_fetch_dev(rtc_ctx* ctx, const rtc_cedge* d_edges, uint64_t m, const uint32_t* d_comp,
                          uint32_t n, const uint64_t* d_ekey, uint32_t* d_ecommon);

/* Fixed-size mode (every sketch holds exactly s hashes -- the -s configs): the distance
 * (src/MST.cpp:1489-1503) is monotone in `common` alone, so ONE u64 key per component carries weight,
 * edge and count:  key = (s - common) << 2B | i << B | j,  B = rtc_boruvka_key_bits(n, s) (0 when the
 * key would not fit 63 bits -> use the three-pass form).  Across GPUs: one all-reduce(MIN) per round. */
int rtc_boruvka_key_bits(uint32_t n, uint32_t s_fixed);
int rtc_boruvka_minkey_dev(rtc_ctx* ctx, const rtc_cedge* d_edges, uint64_t m, const uint32_t* d_comp,
                           uint32_t n, uint32_t s_fixed, uint64_t* d_key);

/* Round state on the device: d_comp[v] = v, forest counter d_nsel[0] = 0 (d_nsel: two u64). */
int rtc_boruvka_init_dev(rtc_ctx* ctx, uint32_t n, uint32_t* d_comp, uint64_t* d_nsel);
/* Union step of a round on the device (kruskalAlgorithm's union-find work, src/MST.cpp:59-75):
 * every component hooks onto the one its (all-reduced) minimum edge leads to, chosen edges are
 * appended to d_sel (capacity n) and d_comp is relabelled.  s_fixed != 0: d_key holds fused keys;
 * s_fixed == 0: d_key holds edge ids (i<<32|j, the d_ekey of the three-pass form) and d_ecommon the
 * counts.  d_succ: n u32 of scratch.  *h_added = edges added this round (0: forest complete).
 * Synchronises the stream (reads one counter back). */
int rtc_boruvka_union_dev(rtc_ctx* ctx, uint32_t n, uint32_t s_fixed, const uint64_t* d_key,
                          const uint32_t* d_ecommon, uint32_t* d_comp, uint32_t* d_succ, rtc_cedge* d_sel,
                          uint64_t* d_nsel, uint32_t* h_added);

/* All rounds on ONE GPU behind one call: the minimum spanning forest (kruskalAlgorithm's result, src/MST.cpp:59-75) of
 * a device-resident candidate list.  d_sel: n entries; *h_n_sel edges are written; h_rounds may be NULL.  Synchronous. */
int rtc_msf_dev(rtc_ctx* ctx, const rtc_cedge* d_edges, uint64_t m, const uint32_t* d_len, uint32_t n, int is_containment,
                rtc_cedge* d_sel, uint64_t* h_n_sel, int* h_rounds);

/* Host helper closing one Boruvka round: unions the components joined by the winning edges
 * (h_ekey[c] = i<<32|j or 0x7FFF...F for none), appends them to h_sel (capacity n) and relabels
 * h_comp[v] with the new root vertex ids.  *h_added == 0 means the forest is complete. */
int rtc_boruvka_merge_host(uint32_t n, const uint64_t* h_ekey, const uint32_t* h_ecommon, uint32_t* h_comp,
                           rtc_cedge* h_sel, uint64_t* h_n_sel, uint64_t* h_added);

/* EdgeInfo of the reference (src/MST.h:17-21); the on-disk edge.mst record (src/MST_IO.cpp:200-217) */
typedef struct { int32_t preNode, sufNode; double dist; } rtc_edge;

/* Host helper: selected forest edges (i, j, common) -> EdgeInfo records with the reference's
 * double arithmetic (src/MST.cpp:1295,1489-1515), sorted by (dist, preNode, sufNode). */
int rtc_edges_to_mst_host(const rtc_cedge* h_sel, uint64_t m, const uint32_t* h_len, int kmer_size,
                          int is_containment, rtc_edge* h_out);

/* Whole single-GPU MST step: compute_minhash_mst / compute_kssd_mst (src/MST.cpp:1290-1737,
 * :216-807) from device-resident sketches.  Distances are evaluated on the HOST with the
 * reference's expression order (src/MST.cpp:1295,1489-1515) so doubles are bit-identical.
 * h_edges_out must hold n entries; *h_n_edges receives the forest size.  Synchronous. */
int rtc_mst(rtc_ctx* ctx, const void* d_hashes, int width, const uint64_t* d_start,
            const uint32_t* d_len, uint32_t n, int kmer_size, int is_containment, double threshold,
            rtc_edge* h_edges_out, uint64_t* h_n_edges);

/* The start_index form of the same functions (src/MST.cpp:1375-1383, used by append_clust_mst,
 * src/sub_command.cpp:1532-1759): only rows i >= start_index of the pair space (all columns j < i)
 * are evaluated -- the pairs that involve an appended genome -- and the forest over those edges is
 * returned; the caller merges it with the stored MST (sort + kruskalAlgorithm, :1693-1700). */
int rtc_mst_append(rtc_ctx* ctx, const void* d_hashes, int width, const uint64_t* d_start,
                   const uint32_t* d_len, uint32_t n, uint32_t start_index, int kmer_size, int is_containment,
                   double threshold, rtc_edge* h_edges_out, uint64_t* h_n_edges);

/* The same with the --dense by-products (src/MST.cpp:1333-1352, :1517-1530, :1703-1713): for every
 * candidate pair (the pairs that become EdgeInfo records) with distance d, both genomes are counted
 * in every radius bucket t with t/dense_span >= d, and ANI bin (int)((1-d)*100) is incremented.
 * h_dense: dense_span x n int32 row-major (mst.dense layout, src/MST_IO.cpp:219-233), h_ani: 101 u64
 * (mst.ani).  dense_span = 0: plain rtc_mst_append.  Buckets use the host doubles of the edge weights. */
int rtc_mst_dense(rtc_ctx* ctx, const void* d_hashes, int width, const uint64_t* d_start, const uint32_t* d_len,
                  uint32_t n, uint32_t start_index, int kmer_size, int is_containment, double threshold,
                  rtc_edge* h_edges_out, uint64_t* h_n_edges, int dense_span, int32_t* h_dense, uint64_t* h_ani);

/* The dense loop modifyMST (src/MST.cpp:809-1018; reached when the index path is switched off, src/sub_command.cpp:2764,
 * :2995, :1680): EVERY pair i < j with j >= start_index is an edge -- no filters -- weighted by MinHash::distance()
 * (the union-truncated estimator of rtc_pair_mash_dev with `sketch_size`; is_containment != 0: containDistance(),
 * -ln(|A n B| / min(|A|, |B|)) / k) and the minimum spanning TREE over them is returned (pairs without a common hash
 * weigh 1), records {i, j, dist} with i < j as modifyMST builds them.  dense_span / h_dense / h_ani as rtc_mst_dense
 * (here every pair is counted, :868-879).  The estimator restates the published Mash algorithm: parity-unpinned like
 * the k-mer hash (RabbitSketch is absent from the reference tree). */
int rtc_mst_mash(rtc_ctx* ctx, const void* d_hashes, int width, const uint64_t* d_start, const uint32_t* d_len,
                 uint32_t n, uint32_t start_index, int kmer_size, int is_containment, uint32_t sketch_size,
                 rtc_edge* h_edges_out, uint64_t* h_n_edges, int dense_span, int32_t* h_dense, uint64_t* h_ani);

/* ---- multi-GPU: RCCL collectives over xGMI and the sharded clust-mst step ----------------- */
/* The reference is one shared-memory process (OpenMP over 8-row blocks of the pair space,
 * src/MST.cpp:1382, and over files, src/SketchInfo.cpp:878).  Here one rtc_comm per rtc_ctx (= per
 * GPU); ranks are processes (one per GPU, the id travels through the launcher's own channel) or
 * host threads of one process.  Every rank sketches its block of genomes; the sketches are gathered
 * into the canonical order (genome g of rank r at row r*n_local + g); the strict lower triangle of
 * the pair space is cut into contiguous row ranges of equal cost; each Boruvka round all-reduces
 * (MIN) one u64 key per component (fixed sketch sizes) or three small arrays (variable sizes). */
typedef struct rtc_comm rtc_comm;
#define RTC_COMM_ID_BYTES 128
int rtc_comm_unique_id(void* id_out /* RTC_COMM_ID_BYTES, created on one rank, passed to all */);
int rtc_comm_init_rank(rtc_ctx* ctx, int nranks, int rank, const void* id, rtc_comm** out); /* collective */
/* One process, one context per GPU, one host thread per context afterwards: communicators for
 * ctxs[0..n).  Contexts that share a device (RCCL rejects duplicate GPUs) get an in-process
 * exchange instead -- the way the protocol is exercised on a one-GPU box. */
int rtc_comm_init_all(rtc_ctx** ctxs, int n, rtc_comm** comms_out);
void rtc_comm_destroy(rtc_comm* comm);
int rtc_comm_rank(const rtc_comm* comm);
int rtc_comm_size(const rtc_comm* comm);
const char* rtc_comm_backend(const rtc_comm* comm); /* "rccl" | "in-process" | "single" */
/* in-place all-reduce on the context stream; dtype 0 = int64, 1 = uint32, 2 = uint64 (the Boruvka key arrays);
 * op 0 = MIN, 1 = MAX */
int rtc_comm_all_reduce(rtc_comm* comm, void* d_buf, size_t count, int dtype, int op);
/* the same for up to 64 host values (agreeing on strides, counts); synchronises */
int rtc_comm_all_reduce_host(rtc_comm* comm, int64_t* h_vals, size_t count, int op);
/* Rows [a,b) of every rank's block of a canonical global buffer (rank r owns rows
 * [r*n_local, (r+1)*n_local), row_bytes each) travel to all ranks, in place (grouped broadcasts).
 * async != 0: on the communicator's side stream, ordered after the work enqueued so far on the
 * context stream; rtc_comm_wait makes the context stream wait for it. */
int rtc_comm_gather_rows(rtc_comm* comm, void* d_global, size_t row_bytes, uint32_t n_local, uint32_t a,
                         uint32_t b, int async);
int rtc_comm_wait(rtc_comm* comm);
/* d_buf[0..bytes) of rank `root` replaces every other rank's copy (context stream) */
int rtc_comm_broadcast(rtc_comm* comm, void* d_buf, size_t bytes, int root);
/* h_bounds[world+1]: row ranges of the strict lower triangle of equal cost, row i costing
 * (i + fixed_cols) columns (fixed_cols: the per-row-block table build; rtc_mst_sharded uses the measured
 * 1.84 x mean sketch size). */
int rtc_triangle_rows(uint32_t n, int world, double fixed_cols, uint32_t* h_bounds);
/* sketchFiles' sketch loop (src/SketchInfo.cpp:878-976) for this rank's genomes, written into its
 * block of the global buffers (d_out_global: size*n_local*stride u64, d_cnt_global: size*n_local)
 * and gathered to all ranks; the gather of the first part overlaps the sketching of the rest. */
int rtc_sketch_minhash_sharded(rtc_ctx* ctx, rtc_comm* comm, const uint8_t* d_seq, const uint64_t* h_off,
                               uint32_t n_local, int k, uint32_t seed, const uint32_t* h_sizes, uint32_t size,
                               uint64_t* d_out_global, uint32_t stride, uint32_t* d_cnt_global);
/* The same phase for a rank whose genomes are resident as batches in the 2-bit staging format (layout: rtc_unpack_bases_dev;
 * what both command lines stage, src/SketchInfo.cpp:928-948 being the records they hold): one call per batch, in the order
 * of the rank's rows.  The batch's n_batch genomes become rows [row_first, row_first + n_batch) of this rank's block of
 * n_local rows; they are sketched straight from the packed bases (rtc_sketch_minhash_packed_dev) and their gather starts on
 * the communicator's side stream behind the sketch kernel, i.e. it travels beside the NEXT batch's kernel.  last != 0 marks
 * the rank's final batch: it is cut in two parts (as rtc_sketch_minhash_sharded cuts a rank's genomes) and the call returns
 * with the context stream waiting for every gather.  Every rank passes the same sequence of (row_first, n_batch) and the
 * same n_local / stride (checked on the first batch).  With one rank the calls sketch into the rows and nothing travels. */
int rtc_sketch_minhash_packed_sharded(rtc_ctx* ctx, rtc_comm* comm, const uint8_t* d_packed, uint64_t n_bases,
                                      const uint64_t* d_runs, uint64_t n_runs, const uint64_t* h_off, uint32_t n_batch,
                                      uint32_t row_first, uint32_t n_local, int last, int k, uint32_t seed,
                                      const uint32_t* h_sizes, uint32_t size, uint64_t* d_out_global, uint32_t stride,
                                      uint32_t* d_cnt_global);
/* --fast: sketchFileWithKssd (src/SketchInfo.cpp:994-1252) over the rank's packed batches, same protocol
 * (rtc_sketch_kssd_packed_dev per batch).  d_out_global: size * n_local * stride tuples of *width_out bytes (4 or 8,
 * src/SketchInfo.cpp:1021).  KSSD sketches vary in length and the rows travel at the caller's stride, so a tight one saves
 * link time.  A batch whose longest sketch exceeds the stride is still gathered -- the ranks' collectives stay matched --
 * and the call with last != 0 returns RTC_ERR_OVERFLOW on EVERY rank with *h_need = the longest sketch any rank produced;
 * the caller repeats the phase with wider rows. */
int rtc_sketch_kssd_packed_sharded(rtc_ctx* ctx, rtc_comm* comm, const uint8_t* d_packed, uint64_t n_bases,
                                   const uint64_t* d_runs, uint64_t n_runs, const uint64_t* h_off, uint32_t n_batch,
                                   uint32_t row_first, uint32_t n_local, int last, int kmer_size, int drlevel,
                                   const int32_t* h_shuffled_dim, void* d_out_global, uint32_t stride,
                                   uint32_t* d_cnt_global, int* width_out, uint32_t* h_need);
typedef struct {
  uint32_t row0, row1;  /* this rank's rows of the pair space */
  uint64_t cand_edges;  /* candidate edges it produced */
  uint32_t rounds, s_fixed, contractions, pad;
  float pair_ms, mst_ms;
} rtc_shard_stats;
/* rtc_mst across the ranks of `comm` (sketches: the complete canonical set, on every rank).  Every
 * rank receives the identical forest, identical to rtc_mst's on one GPU.  stats may be NULL. */
int rtc_mst_sharded(rtc_ctx* ctx, rtc_comm* comm, const void* d_hashes, int width, const uint64_t* d_start,
                    const uint32_t* d_len, uint32_t n, int kmer_size, int is_containment, double threshold,
                    rtc_edge* h_edges_out, uint64_t* h_n_edges, rtc_shard_stats* stats);

/* ---- greedy incremental clustering ------------------------------------------------------- */
/* MinHashGreedyClusterWithInvertedIndex at -t 1 (src/greedy.cpp:986-1399) and
 * KssdGreedyClusterWithInvertedIndex (:566-899; caller sorts by size first, :594-597).
 * Genomes are processed in the given order; the GPU computes query-batch x representative
 * intersections, the host applies the reference's filter / best-match / tie rules.
 * h_size_cfg[n]: what getSketchSize() returns for each genome (configured size, :1201); NULL for
 * KSSD.  h_rep_of[n] receives the representative of each genome (itself if it is one). */
int rtc_greedy(rtc_ctx* ctx, const void* d_hashes, int width, const uint64_t* d_start,
               const uint32_t* d_len, uint32_t n, const uint32_t* h_size_cfg, int kmer_size,
               int is_containment, int is_kssd, double threshold, int32_t* h_rep_of,
               uint32_t* h_n_clusters);

/* greedyCluster (src/greedy.cpp:285-351), the legacy loop without index and filters: every genome is measured
 * against every current representative with MinHash::distance() (Mash's union-truncated estimator over
 * `sketch_size`) or, for containment sketches, containDistance(); it joins the nearest one within the threshold
 * (earliest of equals) or opens a cluster.  Reached with the index path switched off and by `clust-greedy --append`
 * on MinHash sketches without a stored state (src/sub_command.cpp:91).  Estimator: parity-unpinned (RabbitSketch). */
int rtc_greedy_mash(rtc_ctx* ctx, const void* d_hashes, int width, const uint64_t* d_start, const uint32_t* d_len,
                    uint32_t n, int kmer_size, int is_containment, uint32_t sketch_size, double threshold,
                    int32_t* h_rep_of, uint32_t* h_n_clusters);

#ifdef __cplusplus
}
#endif
#endif /* RTCLUST_H */

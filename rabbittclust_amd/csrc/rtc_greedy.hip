// rtc_greedy.hip -- greedy incremental clustering with GPU query-batch x representative counts.
//
// Replaces MinHashGreedyClusterWithInvertedIndex (src/greedy.cpp:986-1399 in the reference tree)
// and KssdGreedyClusterWithInvertedIndex (:566-899).  The reference walks genomes serially and,
// per query, probes a dynamic inverted index of the representatives' hashes.  Here a batch of
// queries is intersected against every current representative AND against the earlier genomes of
// the same batch in one tiled all-pairs launch (a virtual sketch set [reps..., batch...] built
// from CSR offsets, no copy of hashes); the pair kernel emits the non-zero counts as a compact list and the
// host replays the reference's serial decisions -- common_min filter (:1205-1225), best match with
// the reference's strict comparisons (:1233-1282), first-touched-wins tie rule of a -t 1 run --
// so the clusters are identical to the reference at -t 1.
#include <math.h>

#include <algorithm>
#include <limits>
#include <unordered_map>
#include <vector>

#include "rtc_internal.h"

namespace {

// src/greedy.cpp:1245-1275
double greedy_distance(int common, int sizeRef, int sizeQry, int kmer_size, bool repIsContainment) {
  double dist;
  if (repIsContainment) {
    int minSize = std::min(sizeRef, sizeQry);
    if (minSize == 0) return 1.0;
    double jaccard = (double)common / minSize;
    if (jaccard >= 1.0) dist = 0.0;
    else if (jaccard <= 0.0) dist = 1.0;
    else { dist = -log(2.0 * jaccard / (1.0 + jaccard)) / kmer_size; if (dist > 1.0) dist = 1.0; }
  } else {
    int denom = sizeRef + sizeQry - common;
    if (denom == 0) return 0.0;
    double jaccard = (double)common / denom;
    if (jaccard >= 1.0) dist = 0.0;
    else if (jaccard <= 0.0) dist = 1.0;
    else { dist = -log(2.0 * jaccard / (1.0 + jaccard)) / kmer_size; if (dist > 1.0) dist = 1.0; }
  }
  return dist;
}

template <typename T>
uint32_t first_shared_pos(const T* q, uint32_t nq, const T* r, uint32_t nr) {
  uint32_t i = 0, j = 0;
  while (i < nq && j < nr) {
    if (q[i] < r[j]) i++;
    else if (r[j] < q[i]) j++;
    else return i;
  }
  return 0xFFFFFFFFu;
}

struct Cand { uint32_t vcol; uint32_t common; };

}  // namespace

extern "C" int rtc_greedy(rtc_ctx* ctx, const void* d_hashes, int width, const uint64_t* d_start,
                          const uint32_t* d_len, uint32_t n, const uint32_t* h_size_cfg, int kmer_size,
                          int is_containment, int is_kssd, double threshold, int32_t* h_rep_of,
                          uint32_t* h_n_clusters) {
  if (!ctx || !h_rep_of || !h_n_clusters || (n && (!d_start || !d_len))) return RTC_ERR_ARG;
  if (width != 4 && width != 8) return rtc_fail(ctx, RTC_ERR_ARG, "width must be 4 or 8");
  if (!is_kssd && !h_size_cfg) return rtc_fail(ctx, RTC_ERR_ARG, "h_size_cfg is required for MinHash greedy");
  *h_n_clusters = 0;
  if (n == 0) return RTC_OK;
  RTC_HIP(ctx, hipSetDevice(ctx->device));

  // ---- host copies: geometry always, hashes for the first-touched tie rule ----
  std::vector<uint64_t> h_start(n);
  std::vector<uint32_t> h_len(n);
  RTC_HIP(ctx, hipMemcpyAsync(h_start.data(), d_start, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
  RTC_HIP(ctx, hipMemcpyAsync(h_len.data(), d_len, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
  RTC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  // Sketches are fetched on demand (two small copies per tie) and kept while they fit 256 MiB:
  // ties are rare, and a host copy of every sketch would cost more than the clustering itself
  // (2 GB for 50 000 containment sketches).
  std::unordered_map<uint32_t, std::vector<unsigned char>> sketch_cache;
  size_t cache_bytes = 0;
  int fetch_status = RTC_OK;
  auto fetch = [&](uint32_t g) -> const unsigned char* {
    auto it = sketch_cache.find(g);
    if (it != sketch_cache.end()) return it->second.data();
    std::vector<unsigned char> v((size_t)h_len[g] * width + 8);
    if (h_len[g]) {
      hipError_t e = hipMemcpyAsync(v.data(), (const unsigned char*)d_hashes + h_start[g] * width, (size_t)h_len[g] * width,
                                    hipMemcpyDeviceToHost, ctx->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
      if (e != hipSuccess) fetch_status = rtc_fail(ctx, RTC_ERR_HIP, "sketch fetch for the tie rule -> %s", hipGetErrorString(e));
    }
    cache_bytes += v.size();
    return sketch_cache.emplace(g, std::move(v)).first->second.data();
  };
  auto first_pos = [&](uint32_t q, uint32_t r) -> uint32_t {
    if (cache_bytes > ((size_t)256 << 20)) { sketch_cache.clear(); cache_bytes = 0; }  // only between lookups
    const unsigned char* pq = fetch(q);
    const unsigned char* pr = fetch(r);  // pq stays valid: unordered_map does not move mapped values on insert
    if (width == 8) return first_shared_pos((const uint64_t*)pq, h_len[q], (const uint64_t*)pr, h_len[r]);
    return first_shared_pos((const uint32_t*)pq, h_len[q], (const uint32_t*)pr, h_len[r]);
  };

  // ---- reference constants ----
  const double x = exp(-threshold * kmer_size);                              // :1109 / :652
  const double jaccard_min = x / (2.0 - x);
  bool fast = false;
  int fixed_common_min = 0;
  if (!is_kssd) {
    int fixed_sketch_size = (int)h_size_cfg[0];                              // :1092
    bool all_fixed = true, all_std = !is_containment;                        // :1093-1094
    for (uint32_t i = 1; i < std::min<uint32_t>(100, n); i++)                // :1097-1103
      if (is_containment || (int)h_size_cfg[i] != fixed_sketch_size) { all_fixed = false; all_std = false; break; }
    fast = all_fixed && all_std && !is_containment;
    if (fast) fixed_common_min = (int)ceil(jaccard_min * (2 * fixed_sketch_size) / (1.0 + jaccard_min));  // :1112
  }

  // ---- state ----
  std::vector<uint32_t> reps;           // representative genome ids in creation order
  std::vector<uint32_t> rep_order(n, 0xFFFFFFFFu);  // genome id -> creation index
  reps.push_back(0); rep_order[0] = 0; h_rep_of[0] = 0;                      // :1075-1079

  const uint32_t B = 1024;
  std::vector<uint64_t> v_start; std::vector<uint32_t> v_len;
  uint64_t* dv_start = nullptr; uint32_t* dv_len = nullptr; size_t dv_cap = 0;
  rtc_cedge* d_edges = nullptr; uint64_t ecap = 1u << 22;
  unsigned long long* d_count = nullptr;
  std::vector<rtc_cedge> h_edges;
  int st = RTC_OK;
  auto cleanup = [&]() {
    if (dv_start) (void)hipFree(dv_start);
    if (dv_len) (void)hipFree(dv_len);
    if (d_edges) (void)hipFree(d_edges);
    if (d_count) (void)hipFree(d_count);
  };
#define G_TRY(call) do { st = (call); if (st != RTC_OK) { cleanup(); return st; } } while (0)
#define G_HIP(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { cleanup(); return rtc_fail(ctx, RTC_ERR_HIP, "%s -> %s", #call, hipGetErrorString(e__)); } } while (0)
  G_HIP(hipMalloc(&d_edges, ecap * sizeof(rtc_cedge)));
  ctx->free_hbm_at = -1.0;  // the pair phase sizes its scratch from rtc_free_hbm: this allocation changed it
  G_HIP(hipMalloc(&d_count, 8));

  // ---- serial replay of the reference's decisions for one query: candidates [c0, c1), `resolve` maps a candidate's
  // column to its genome id and says whether that genome is a representative ----
  auto decide = [&](uint32_t q, const Cand* c0, const Cand* c1, auto&& resolve) {
    const int sizeRef = (int)h_len[q];                                      // :1139 / :684
    int best_common = -1; double best_dist = std::numeric_limits<double>::max(); double best_jac = -1.0;
    uint32_t best_rep = 0xFFFFFFFFu;
    std::vector<uint32_t> ties;
    for (const Cand* pc = c0; pc != c1; pc++) {
      const Cand& cd = *pc;
      uint32_t rep;
      if (!resolve(cd.vcol, rep)) continue;  // not a representative
      const int common = (int)cd.common;
      if (is_kssd) {
        const int sizeQry = (int)h_len[rep];                                // :768
        const int common_min = (int)ceil(jaccard_min * (sizeRef + sizeQry) / (1.0 + jaccard_min));  // :774
        if (common < common_min) continue;
        const int denom = sizeRef + sizeQry - common;
        const double jac = denom == 0 ? 1.0 : (double)common / denom;      // :785-786
        if (jac > best_jac) { best_jac = jac; best_rep = rep; ties.clear(); ties.push_back(rep); }
        else if (jac == best_jac) ties.push_back(rep);
      } else {
        const int sizeQry = (int)h_size_cfg[rep];                           // :1201 getSketchSize()
        int common_min;
        if (fast) common_min = fixed_common_min;                            // :1206-1208
        else if (is_containment) common_min = (int)ceil(jaccard_min * std::min(sizeRef, sizeQry));  // :1216
        else common_min = (int)ceil(jaccard_min * (sizeRef + sizeQry) / (1.0 + jaccard_min));       // :1218
        if (common < common_min) continue;                                  // :1222
        if (fast) {
          if (common > best_common) { best_common = common; best_rep = rep; ties.clear(); ties.push_back(rep); }  // :1236
          else if (common == best_common) ties.push_back(rep);
        } else {
          const double dist = greedy_distance(common, sizeRef, sizeQry, kmer_size, is_containment != 0);
          if (dist <= threshold) {                                          // :1277
            if (dist < best_dist) { best_dist = dist; best_rep = rep; ties.clear(); ties.push_back(rep); }
            else if (dist == best_dist) ties.push_back(rep);
          }
        }
      }
    }
    if (ties.size() > 1) {
      // -t 1 reference order: candidates are visited in first-touch order = (position of the first
      // query hash they share, representative creation order); strict comparisons keep the first.
      uint32_t bp = 0xFFFFFFFFu, bo = 0xFFFFFFFFu;
      for (uint32_t r : ties) {
        const uint32_t pos = first_pos(q, r), ord = rep_order[r];
        if (pos < bp || (pos == bp && ord < bo)) { bp = pos; bo = ord; best_rep = r; }
      }
    }
    if (best_rep != 0xFFFFFFFFu) {
      h_rep_of[q] = (int32_t)best_rep;                                      // :1321-1325
    } else {
      h_rep_of[q] = (int32_t)q;                                             // :1327-1336
      rep_order[q] = (uint32_t)reps.size();
      reps.push_back(q);
    }
  };

  // ---- one inverted join over the whole set when it is sparse enough (rtc_pairs_join.hip): every pair (q, j < q)
  // sharing a hash, once; the replay keeps the candidates that are representatives when their query comes up --
  // what the reference's representative-only index returns (src/greedy.cpp:1150-1196) ----
  bool global_done = false;
  uint64_t global_pair_budget = (uint64_t)1 << 27;
  if (ctx->opt.has_greedy_global_pairs) global_pair_budget = ctx->opt.greedy_global_pairs;  // tests of the fall-through
  if (n > B) {
    int handled = 0;
    uint64_t m = 0;
    while (true) {
      G_HIP(hipMemsetAsync(d_count, 0, 8, ctx->stream));
      G_TRY(rtc_pair_edges_join(ctx, d_hashes, width, d_start, d_len, n, 1, n, 0, n - 1, -1, d_edges, ecap, (uint64_t*)d_count, 0.5,
                                &handled));
      if (!handled) break;
      unsigned long long cnt = 0;
      G_HIP(hipMemcpyAsync(&cnt, d_count, 8, hipMemcpyDeviceToHost, ctx->stream));
      G_HIP(hipStreamSynchronize(ctx->stream));
      if (cnt > global_pair_budget) { handled = 0; break; }
      if (cnt <= ecap) { m = cnt; break; }
      // The whole set's candidate pairs at once: 12 B each here, ~20 B each on the host while they are bucketed.  The
      // join's cost rule weighs time, not this memory: beyond the budget (2^27 pairs = 1.6 GB + 2.7 GB by default), or
      // when the larger list cannot be allocated, the block loop below takes over -- it never holds more than one block
      // of queries' candidates.
      const uint64_t old_cap = ecap;
      (void)hipFree(d_edges); d_edges = nullptr;
      ecap = cnt + cnt / 4;
      if (hipMalloc(&d_edges, ecap * sizeof(rtc_cedge)) != hipSuccess) {
        (void)hipGetLastError();
        d_edges = nullptr;
        ecap = old_cap;
        G_HIP(hipMalloc(&d_edges, ecap * sizeof(rtc_cedge)));
        ctx->free_hbm_at = -1.0;  // the pair phase sizes its scratch from rtc_free_hbm: this allocation changed it
        handled = 0;
        break;
      }
    }
    if (handled) {
      h_edges.resize(m);
      if (m) {
        G_HIP(hipMemcpyAsync(h_edges.data(), d_edges, m * sizeof(rtc_cedge), hipMemcpyDeviceToHost, ctx->stream));
        G_HIP(hipStreamSynchronize(ctx->stream));
      }
      std::vector<uint64_t> qoff((size_t)n + 1, 0);
      for (uint64_t e = 0; e < m; e++) qoff[(size_t)h_edges[e].i + 1]++;
      for (uint32_t i = 0; i < n; i++) qoff[i + 1] += qoff[i];
      std::vector<Cand> cands(m);
      {
        std::vector<uint64_t> cur(qoff.begin(), qoff.begin() + n);
        for (uint64_t e = 0; e < m; e++) cands[cur[h_edges[e].i]++] = Cand{h_edges[e].j, h_edges[e].common};
      }
      std::vector<rtc_cedge>().swap(h_edges);
      for (uint32_t q = 1; q < n; q++) {
        decide(q, cands.data() + qoff[q], cands.data() + qoff[q + 1], [&](uint32_t vcol, uint32_t& rep) {
          rep = vcol;
          return rep_order[rep] != 0xFFFFFFFFu;
        });
        if (fetch_status != RTC_OK) { cleanup(); return fetch_status; }
      }
      global_done = true;
      ctx->diag[4]++;
    }
  }

  for (uint32_t q0 = 1; q0 < n && !global_done; q0 += B) {
    const uint32_t q1 = std::min(n, q0 + B), nb = q1 - q0;
    ctx->diag[5]++;
    const uint32_t nr = (uint32_t)reps.size();
    const uint32_t nv = nr + nb;
    // virtual sketch set: [representatives in creation order..., this batch in processing order...]
    v_start.resize(nv); v_len.resize(nv);
    for (uint32_t i = 0; i < nr; i++) { v_start[i] = h_start[reps[i]]; v_len[i] = h_len[reps[i]]; }
    for (uint32_t i = 0; i < nb; i++) { v_start[nr + i] = h_start[q0 + i]; v_len[nr + i] = h_len[q0 + i]; }
    if (nv > dv_cap) {
      if (dv_start) (void)hipFree(dv_start);
      if (dv_len) (void)hipFree(dv_len);
      dv_start = nullptr; dv_len = nullptr;
      dv_cap = (size_t)nv + nv / 2 + B;
      G_HIP(hipMalloc(&dv_start, dv_cap * 8));
      G_HIP(hipMalloc(&dv_len, dv_cap * 4));
    }
    G_HIP(hipMemcpyAsync(dv_start, v_start.data(), (size_t)nv * 8, hipMemcpyHostToDevice, ctx->stream));
    G_HIP(hipMemcpyAsync(dv_len, v_len.data(), (size_t)nv * 4, hipMemcpyHostToDevice, ctx->stream));
    // candidate (query, earlier genome, common) triples straight from the pair kernel (no dense matrix);
    // the reference's filters need the configured sizes and run on the host below
    uint64_t m = 0;
    while (true) {
      G_HIP(hipMemsetAsync(d_count, 0, 8, ctx->stream));
      G_TRY(rtc_pair_edges_dev(ctx, d_hashes, width, dv_start, dv_len, nv, nr, nv, 0, nv - 1, -1, d_edges, ecap,
                               (uint64_t*)d_count));
      unsigned long long cnt = 0;
      G_HIP(hipMemcpyAsync(&cnt, d_count, 8, hipMemcpyDeviceToHost, ctx->stream));
      G_HIP(hipStreamSynchronize(ctx->stream));
      if (cnt <= ecap) { m = cnt; break; }
      (void)hipFree(d_edges); d_edges = nullptr;
      ecap = cnt + cnt / 4;
      G_HIP(hipMalloc(&d_edges, ecap * sizeof(rtc_cedge)));
      ctx->free_hbm_at = -1.0;  // the pair phase sizes its scratch from rtc_free_hbm: this allocation changed it
    }
    h_edges.resize(m);
    if (m) {
      G_HIP(hipMemcpyAsync(h_edges.data(), d_edges, m * sizeof(rtc_cedge), hipMemcpyDeviceToHost, ctx->stream));
      G_HIP(hipStreamSynchronize(ctx->stream));
    }
    // bucket candidates by query (virtual row index)
    std::vector<uint32_t> qoff(nb + 1, 0);
    for (uint64_t e = 0; e < m; e++) qoff[h_edges[e].i - nr + 1]++;
    for (uint32_t i = 0; i < nb; i++) qoff[i + 1] += qoff[i];
    std::vector<Cand> cands(m);
    {
      std::vector<uint32_t> cur(qoff.begin(), qoff.begin() + nb);
      for (uint64_t e = 0; e < m; e++) cands[cur[h_edges[e].i - nr]++] = Cand{h_edges[e].j, h_edges[e].common};
    }
    for (uint32_t bi = 0; bi < nb; bi++) {
      decide(q0 + bi, cands.data() + qoff[bi], cands.data() + qoff[bi + 1], [&](uint32_t vcol, uint32_t& rep) {
        if (vcol < nr) { rep = reps[vcol]; return true; }
        rep = q0 + (vcol - nr);
        return rep_order[rep] != 0xFFFFFFFFu;
      });
    }
    if (fetch_status != RTC_OK) { cleanup(); return fetch_status; }
  }
  cleanup();
#undef G_TRY
#undef G_HIP
  *h_n_clusters = (uint32_t)reps.size();
  return RTC_OK;
}

// greedyCluster (src/greedy.cpp:285-351): the legacy greedy loop that measures every genome against EVERY current
// representative with MinHash::distance() / containDistance() -- no index, no filters.  It is what clust-greedy
// runs when the index path is switched off and what `clust-greedy --append` runs on MinHash sketches without a
// stored state (src/sub_command.cpp:91).  Row blocks of the dense estimator matrix come from the GPU
// (rtc_pair_mash_dev: Mash's union-truncated counts; containment: full intersections), the serial
// decisions are replayed on the host: a genome joins the representative at the smallest distance <= threshold
// (the reference keeps the candidates in a map<double, int>: the smallest distance wins, of equal ones the first
// inserted, i.e. the earliest representative at -t 1), otherwise it becomes a representative.
extern "C" int rtc_greedy_mash(rtc_ctx* ctx, const void* d_hashes, int width, const uint64_t* d_start,
                               const uint32_t* d_len, uint32_t n, int kmer_size, int is_containment,
                               uint32_t sketch_size, double threshold, int32_t* h_rep_of, uint32_t* h_n_clusters) {
  if (!ctx || !h_rep_of || !h_n_clusters || (n && (!d_start || !d_len))) return RTC_ERR_ARG;
  if (!is_containment && sketch_size == 0) return rtc_fail(ctx, RTC_ERR_ARG, "sketch_size 0");
  *h_n_clusters = 0;
  if (n == 0) return RTC_OK;
  RTC_HIP(ctx, hipSetDevice(ctx->device));
  std::vector<uint32_t> len(n);
  RTC_HIP(ctx, hipMemcpyAsync(len.data(), d_len, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
  RTC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const uint32_t B = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint32_t>(n, 262140), ((uint64_t)1 << 24) / n));
  void* ws = nullptr;
  RTC_TRY(rtc_ws(ctx, 2, (size_t)B * n * 8 + 64, &ws));
  uint32_t* d_common = (uint32_t*)ws;
  uint32_t* d_denom = d_common + (size_t)B * n;
  std::vector<uint32_t> common((size_t)B * n), denom(is_containment ? 0 : (size_t)B * n);
  h_rep_of[0] = 0;  // :293-294
  uint32_t ncl = 1;
  for (uint32_t r0 = 1; r0 < n; r0 += B) {
    const uint32_t r1 = std::min(n, r0 + B);
    if (is_containment) RTC_TRY(rtc_pair_common_dev(ctx, d_hashes, width, d_start, d_len, n, r0, r1, 0, r1 - 1, d_common, n, 1, 0));
    else RTC_TRY(rtc_pair_mash_dev(ctx, d_hashes, width, d_start, d_len, n, sketch_size, r0, r1, 0, r1 - 1, d_common, d_denom, n));
    RTC_HIP(ctx, hipMemcpyAsync(common.data(), d_common, (size_t)(r1 - r0) * n * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (!is_containment) RTC_HIP(ctx, hipMemcpyAsync(denom.data(), d_denom, (size_t)(r1 - r0) * n * 4, hipMemcpyDeviceToHost, ctx->stream));
    RTC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (uint32_t q = r0; q < r1; q++) {
      const uint32_t* crow = common.data() + (size_t)(q - r0) * n;
      const uint32_t* drow = is_containment ? nullptr : denom.data() + (size_t)(q - r0) * n;
      double best = std::numeric_limits<double>::max();
      int32_t best_rep = -1;
      for (uint32_t c = 0; c < q; c++) {
        if (h_rep_of[c] != (int32_t)c) continue;  // representatives only
        double dist;
        if (is_containment) {  // MinHash::containDistance(): -ln(|A n B| / min(|A|, |B|)) / k
          const uint32_t mn = std::min(len[q], len[c]);
          const double cj = mn ? (double)crow[c] / mn : 0.0;
          dist = cj == 0.0 ? 1.0 : (cj == 1.0 ? 0.0 : -(1.0 / kmer_size) * log(cj));  // the in-tree operation order (src/MST.cpp:1295,1515)
        } else {               // MinHash::distance(): Mash
          const double j = drow[c] ? (double)crow[c] / drow[c] : 0.0;
          dist = j == 0.0 ? 1.0 : (j == 1.0 ? 0.0 : -log(2.0 * j / (1.0 + j)) / kmer_size);
          if (dist > 1.0) dist = 1.0;
        }
        if (dist <= threshold && dist < best) { best = dist; best_rep = (int32_t)c; }  // :319-326, :334-336
      }
      if (best_rep >= 0) h_rep_of[q] = best_rep;
      else { h_rep_of[q] = (int32_t)q; ncl++; }
    }
  }
  *h_n_clusters = ncl;
  return RTC_OK;
}

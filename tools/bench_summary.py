"""One screen of a bench.py run: headline, phases, roofline fraction, then each extra workload.  Usage: bench_summary.py <file.jsonl>
(reads the full object -- the {"headline": ..., "extra": ...} line in front of the compact last line; older files: the last line)"""
import json, sys
objs = [json.loads(x) for x in open(sys.argv[1]) if x.startswith("{")]
full = next((o for o in objs if "headline" in o), None)
d = dict(full["headline"], extra=full.get("extra") or {}) if full else objs[-1]
if "value" not in d:  # --only lines
    d = {"value": None, "ms_per_step": None, "phase_ms": {}, "roofline": {"frac": None}, "extra": d.get("extra", {})}
print(d["value"], d["ms_per_step"], {k: round(v, 3) for k, v in d["phase_ms"].items() if k.endswith("_ms")}, d["roofline"]["frac"], d["roofline"].get("physical_frac"))
if d.get("cpu_baseline", {}).get("dist_fit"):
    print("cpu", d["cpu_baseline"]["sketch_gbp_per_sec"], d["cpu_baseline"]["dist_fit"])
for k, v in (d.get("extra") or {}).items():
    if "error" in v: print(k, "ERROR", v); continue
    if k in ("kssd", "kssd_packed"): print(k, v["ms_per_step"], v["phase_ms"]["sketch_ms"], v["roofline"]["frac"], v["roofline"].get("physical_frac"), v["roofline"]["traffic"])
    elif k in ("minhash_ascii", "minhash_packed"): print(k, v["ms_per_step"], v["phase_ms"]["sketch_ms"], v["roofline"]["frac"], v["roofline"]["traffic"])
    elif k == "greedy": print(k, v["sketch_ms"], v["sketch_ms_packed"], v["greedy_s"], v["roofline"]["frac"], v["roofline_packed"]["frac"], v["roofline"]["traffic"])
    elif k == "weak_first_point": print(k, v["ms_per_step"])
    elif k == "dense_pairs": print(k, v["pair_path"], v["pair_ms"], v["pair_kernel_ms"], v["mst_ms"], v["first_call_pair_ms"], v["roofline_dist"]["frac"], "| 25000 u32:", {a: v["u32_25000"].get(a) for a in ("pair_path", "pair_ms", "pair_kernel_ms", "mst_ms", "first_call_pair_ms", "cand_edges")})
    elif k in ("config3_1gpu", "config5_1gpu"):
        print(k, {a: v.get(a) for a in ("total_s", "sketch_s", "pair_ms", "mst_ms", "clusters", "cand_edges", "cpu_extrapolated_s", "gpu_vs_cpu_extrapolated")})
        print("   cpu", {a: v.get("cpu_extrapolated", {}).get(a) for a in ("sketch_s", "dist_s", "dist_fit")})
        print("   shards", v.get("row_shards_on_one_gpu"))
    elif k == "cli": print(k, {m: (round(v[m]["wall_s"], 3), round(v[m]["end_to_end_gbp_per_sec"], 1), v[m].get("parse_gbp_per_sec_per_thread")) for m in v if isinstance(v[m], dict) and "wall_s" in v[m]})

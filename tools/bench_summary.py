import json,sys
"""One screen of a bench.py line: headline, phases, roofline fraction, then each extra workload.  Usage: bench_summary.py <file.jsonl>"""
d=json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1])
if "value" not in d: d = {"value": None, "ms_per_step": None, "phase_ms": {}, "roofline": {"frac": None}, "extra": d.get("extra", {})}  # --only lines
print(d["value"], d["ms_per_step"], {k:round(v,3) for k,v in d["phase_ms"].items() if k.endswith("_ms")}, d["roofline"]["frac"])
for k,v in d["extra"].items():
    if "error" in v: print(k,"ERROR",v); continue
    if k in ("kssd", "kssd_packed"): print(k, v["ms_per_step"], v["phase_ms"]["sketch_ms"], v["roofline"]["frac"], v["roofline"].get("physical_frac"), v["roofline"]["traffic"])
    elif k=="greedy": print(k, v["sketch_ms"], v["greedy_s"], v["roofline"]["frac"], v["roofline"]["traffic"])
    elif k=="weak_first_point": print(k, v["ms_per_step"])
    elif k=="dense_pairs": print(k, v["pair_path"], v["pair_ms"], v["pair_kernel_ms"], v["mst_ms"], v["roofline_dist"]["frac"], v["roofline_dist"]["traffic"])
    elif k=="cli": print(k, {m:(round(v[m]["wall_s"],3), round(v[m]["end_to_end_gbp_per_sec"],1), v[m]["hip_init_exposed_s"]) for m in ("minhash","fast")})

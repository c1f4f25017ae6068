"""bench.py end to end on the GPU box: the default line must carry every extra workload WITHOUT an error entry (an extra that
breaks is reported as {"error": ...} so that it never costs the headline -- which also means nobody notices unless a test
looks), the dense extra must have gone through the tiled kernel by the cost rule's own choice, and the roofline objects
must be complete."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_default_line_has_every_extra_without_error(ctx):
    import gc
    import torch
    gc.collect()
    torch.cuda.empty_cache()  # the child needs ~130 GB for the north-star extras: nothing cached by earlier tests of this process may stay
    env = {k: v for k, v in os.environ.items() if not k.startswith("RTC_PAIR")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--extra-steps", "1",
                        "--cli-genomes", "64", "--cpu-sample-genomes", "32", "--cpu-sample-sketches", "2000"],
                       capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["steps"] == 1 and line["value"] > 0
    for key in ("roofline", "roofline_dist", "cpu_baseline", "extra"):
        assert key in line, key
    assert line["roofline"]["frac"] > 0 and line["cpu_baseline"].get("value"), line["cpu_baseline"]
    assert line["roofline_dist"]["survey_8d_bytes_per_pair"] == 16000.0
    ex = line["extra"]
    assert set(ex) == {"minhash_packed", "kssd", "kssd_packed", "greedy", "weak_first_point", "dense_pairs", "config3_1gpu", "config5_1gpu", "cli"}
    for name, v in ex.items():
        assert "error" not in v, (name, v)
    # the same genomes sketched from characters and from the 2-bit staging format: same sketches, same forest
    assert ex["kssd_packed"]["mean_sketch_size"] == ex["kssd"]["mean_sketch_size"] and ex["kssd_packed"]["mst_edges"] == ex["kssd"]["mst_edges"]
    assert ex["kssd_packed"]["roofline"]["kernel"] == "sketch_kssd_packed_kernel" and ex["kssd_packed"]["roofline"]["physical_frac"] > 0
    # the headline's genomes from the 2-bit staging format: same forest, fewer milliseconds of sketching than bytes would suggest
    mp = ex["minhash_packed"]
    assert mp["mst_edges"] == line["mst_edges"] and mp["roofline"]["physical_frac"] > 0 and mp["phase_ms"]["sketch_ms"] > 0
    # the north-star configurations on one GPU: whole job, clusters, and the CPU side labelled as extrapolated
    for name, n in (("config3_1gpu", 100000), ("config5_1gpu", 200000)):
        c = ex[name]
        assert c["total_s"] > 0 and c["sketch_s"] > 0 and c["pair_ms"] > 0 and c["mst_ms"] > 0 and c["pair_path"] in (2, 3)
        assert n // 10 <= c["clusters"] < n and c["mst_edges"] > n // 2, (name, c["clusters"], c["mst_edges"])
        assert c["cpu_extrapolated_s"] > 0 and "EXTRAPOLATED" in c["cpu_extrapolated"]["label"] and c["cpu_extrapolated"]["sample"]
    assert ex["greedy"]["packed_sketches_identical"] and ex["greedy"]["sketch_ms_packed"] > 0
    d = ex["dense_pairs"]
    assert d["pair_path"] == 2 and d["pair_kernel_ms"] > 0 and d["cand_edges"] >= 10 * 1000 * 999 // 2
    assert d["roofline_dist"]["bytes_per_pair"] == 16000.0 and d["roofline_dist"]["algorithmic_frac"] > 0
    for mode in ("gz", "contigs"):
        c = ex["cli"][mode]
        assert c["wall_s"] > 0 and c["clusters"] > 0 and c["gpu_sketch_ms_per_batch"] > 0, (mode, c)
    assert ex["cli"]["gz"]["inflate_gb_per_sec_per_thread"] > 0 and ex["cli"]["contigs"]["runs_per_genome"] > 100
    assert ex["cli"]["greedy"]["clusters"] > 0 and ex["cli"]["greedy"]["greedyCluster_s"] is not None
    for mode in ("minhash", "fast"):
        c = ex["cli"][mode]
        assert c["genomes"] == 64 and c["wall_s"] > 0 and c["computing_sketch_s"] > 0 and c["parse_gbp_per_sec_per_thread"] > 0

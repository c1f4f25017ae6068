"""bench.py end to end on the GPU box.  The driver reads ONE thing of a round: the last stdout line of `bench.py`.  So the
tests here are about the shape of the evidence as much as about its content: the last line parses, is the compact headline
(< 4 096 bytes) with `roofline` and `cpu_baseline`, no string anywhere in any printed line is long or looks like an array
that slipped into an f-string, every extra workload ran WITHOUT an error entry (an extra that breaks is reported as
{"error": ...} so that it never costs the headline -- which also means nobody notices unless a test looks), and the
headline is printed before the extras start."""
import json
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

sys.path.insert(0, ROOT)


def _strings(o, path=""):
    if isinstance(o, dict):
        for k, v in o.items():
            yield from _strings(v, f"{path}.{k}")
    elif isinstance(o, list):
        for i, v in enumerate(o):
            yield from _strings(v, f"{path}[{i}]")
    elif isinstance(o, str):
        yield path, o


def check_lines(stdout):
    """every JSON line of a bench run obeys the output protocol; returns (compact headline, full object or None)"""
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert lines, stdout[-2000:]
    last = stdout.rstrip("\n").splitlines()[-1]
    assert last == lines[-1], "the compact headline must be the LAST stdout line"
    assert len(last.encode()) < 4096, len(last.encode())
    objs = [json.loads(ln) for ln in lines]  # every line parses
    for o in objs:
        for path, sv in _strings(o):
            assert len(sv) <= 400, (path, len(sv))
            assert not re.search(r"\[\s*\d{10}", sv), (path, sv[:80])  # an array repr inside a string (round 5's bug)
    head = objs[-1]
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in head, key
    assert "workload" in head["config"] and len(head["config"]["workload"]) < 300
    for key in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in head["roofline"], key
    full = next((o for o in objs if "headline" in o), None)
    return head, full, objs


def test_bench_default_line_has_every_extra_without_error(ctx, tmp_path):
    import gc
    import torch
    gc.collect()
    torch.cuda.empty_cache()  # the child needs ~130 GB for the north-star extras: nothing cached by earlier tests of this process may stay
    env = {k: v for k, v in os.environ.items() if not k.startswith("RTC_PAIR")}
    xj = str(tmp_path / "bench_extra.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--extra-steps", "1",
                        "--cli-genomes", "64", "--cli-genomes-large", "128", "--cpu-sample-genomes", "32", "--cpu-sample-sketches", "2000",
                        "--extra-json", xj],
                       capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line, full, objs = check_lines(r.stdout)
    assert len(objs) == 3 and objs[0]["value"] == line["value"] and "extra_scalars" not in objs[0], "headline first, extras, headline again"
    assert line["n_gpus"] == 1 and line["steps"] == 1 and line["value"] > 0
    assert line["config"]["staging"] == "packed" and line["roofline"]["kernel"].startswith("sketch_minhash_packed_kernel")
    assert line["roofline"]["frac"] > 0 and line["roofline"]["physical_frac"] > 0 and line["cpu_baseline"]["value"], line["cpu_baseline"]
    assert line["extra_errors"] == [], line["extra_errors"]
    sc = line["extra_scalars"]
    for k in ("minhash_packed_ms", "minhash_ascii_ms", "kssd_frac", "kssd_packed_frac", "kssd_packed_physical_frac", "greedy_frac",
              "dense_pair_kernel_ms", "dense_first_call_ms", "dense25k_pair_ms", "config3_total_s", "config5_total_s", "cli_gbp_per_sec", "cli_large_gbp_per_sec"):
        assert sc[k] is not None and sc[k] > 0, (k, sc)
    assert json.load(open(xj)) == full
    head = full["headline"]
    assert head["roofline_dist"]["survey_8d_bytes_per_pair"] == 16000.0 and head["cpu_baseline"]["dist_fit"]["n"] == [500, 1000, 2000]
    ex = full["extra"]
    assert set(ex) == {"minhash_ascii", "kssd", "kssd_packed", "greedy", "weak_first_point", "dense_pairs", "config3_1gpu", "config5_1gpu", "cli"}
    for name, v in ex.items():
        assert "error" not in v, (name, v)
    # the same genomes sketched from characters and from the 2-bit staging format: same sketches, same forest
    assert ex["kssd_packed"]["mean_sketch_size"] == ex["kssd"]["mean_sketch_size"] and ex["kssd_packed"]["mst_edges"] == ex["kssd"]["mst_edges"]
    assert ex["kssd_packed"]["roofline"]["kernel"] == "sketch_kssd_packed_kernel" and ex["kssd_packed"]["roofline"]["physical_frac"] > 0
    ma = ex["minhash_ascii"]
    assert ma["mst_edges"] == line["mst_edges"] and ma["roofline"]["kernel"].startswith("sketch_minhash_kernel") and ma["phase_ms"]["sketch_ms"] > 0
    # the north-star configurations on one GPU: whole job, clusters, and the CPU side labelled as extrapolated by the fitted law
    for name, n in (("config3_1gpu", 100000), ("config5_1gpu", 200000)):
        c = ex[name]
        assert c["total_s"] > 0 and c["sketch_s"] > 0 and c["pair_ms"] > 0 and c["mst_ms"] > 0 and c["pair_path"] in (2, 3)
        assert n // 10 <= c["clusters"] < n and c["mst_edges"] > n // 2, (name, c["clusters"], c["mst_edges"])
        cx = c["cpu_extrapolated"]
        assert c["cpu_extrapolated_s"] > 0 and "EXTRAPOLATED" in cx["label"] and cx["sample"] and len(cx["dist_fit"]["n"]) == 3
        assert cx["dist_s"] < n * (n - 1) / 2 / (2000 * 1999 / 2) * cx["dist_fit"]["s"][-1], "the distance law is not the pair count's"
    assert ex["greedy"]["packed_sketches_identical"] and ex["greedy"]["sketch_ms_packed"] > 0
    d = ex["dense_pairs"]
    assert d["pair_path"] == 2 and d["pair_kernel_ms"] > 0 and d["cand_edges"] >= 10 * 1000 * 999 // 2
    assert "10 families of 1000 genomes" in d["workload"]
    assert d["roofline_dist"]["bytes_per_pair"] == 16000.0 and d["roofline_dist"]["algorithmic_frac"] > 0
    for mode in ("gz", "contigs"):
        c = ex["cli"][mode]
        assert c["wall_s"] > 0 and c["clusters"] > 0 and c["gpu_sketch_ms_per_batch"] > 0, (mode, c)
    assert ex["cli"]["gz"]["inflate_gb_per_sec_per_thread"] > 0 and ex["cli"]["contigs"]["runs_per_genome"] > 100
    assert ex["cli"]["greedy"]["clusters"] > 0 and ex["cli"]["greedy"]["greedyCluster_s"] is not None
    for mode in ("minhash", "fast"):
        c = ex["cli"][mode]
        assert c["genomes"] == 64 and c["wall_s"] > 0 and c["computing_sketch_s"] > 0 and c["parse_gbp_per_sec_per_thread"] > 0
    assert ex["cli"]["minhash_large"]["genomes"] == 128


def test_bench_strong_scaling_one_rank_small(tmp_path):
    """`--scaling strong` on one GPU: the job's genomes as packed batches through rtc_sketch_minhash_packed_sharded +
    rtc_mst_sharded with a communicator of one rank -- the N = 1 point of the strong-scaling curve, here at a size of seconds;
    the weak-scaling run over the same genomes as ONE batch gives the same forest size and cluster count."""
    args = ["--steps", "1", "--warmup", "1", "--genomes", "5000", "--length", "100000", "--no-cpu-baseline", "--no-extra",
            "--extra-json", str(tmp_path / "x.json")]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--scaling", "strong"] + args, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line, full, _ = check_lines(r.stdout)
    assert line["scaling"] == "strong" and line["config"]["genomes_total"] == 5000 and line["config"]["collectives"].startswith("rtc_comm (single")
    assert line["mst_edges"] > 2500 and 500 <= line["clusters"] < 5000 and line["phase_ms"]["sketch_ms"] > 0
    assert full["headline"]["config"]["batches_per_gpu"] == 1
    k = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--scaling", "strong", "--mode", "kssd"] + args, capture_output=True, text=True, timeout=600)
    assert k.returncode == 0, k.stderr[-2000:]
    kl, _, _ = check_lines(k.stdout)
    assert kl["dtype"] == "u32" and kl["roofline"]["kernel"] == "sketch_kssd_packed_kernel" and kl["mst_edges"] > 500

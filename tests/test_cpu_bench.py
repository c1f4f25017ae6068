"""bench.py's output protocol, host logic only (no GPU): the compact headline the driver parses stays a valid, bounded JSON
line whatever the extra workloads return -- round 5 lost its evidence to a numpy array formatted into a string."""
import json
import os
import sys

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


def test_compact_line_fits_whatever_the_extras_hold():
    """host logic only: the compact line stays below the limit and valid when the extras are huge, broken or absent"""
    import numpy as np
    import bench
    line = {"metric": "m", "value": 1.0, "unit": "u", "n_gpus": 1, "steps": 1, "warmup": 0, "ms_per_step": 1.0, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "w" * 250, "genomes_total": 1, "staging": "packed"}, "sketch_gbp_per_sec": 1.0, "dist_pairs_per_sec": 1.0,
            "mst_edges": 1, "clusters": 1, "phase_ms": {"sketch_ms": 1.0, "pair_ms": float("nan")},
            "roofline": {"bound": "hbm", "kernel": "k", "achieved": 1.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.1, "traffic": None},
            "cpu_baseline": {"value": 1.0, "unit": "u", "cores": 1, "kind": "port", "sample": "s" * 1000}}
    huge = {"dense_pairs": {"workload": f"{np.arange(2000)}", "pair_kernel_ms": np.float32(1.5)}, "cli": {"error": "x" * 5000},
            "junk": {str(i): "y" * 300 for i in range(200)}}
    c = bench.compact_line(line, huge)
    s = json.dumps(c)
    assert len(s) < 4096 and json.loads(s)["cpu_baseline"]["sample"] == "s" * 200
    assert c["extra_scalars"]["dense_pair_kernel_ms"] == 1.5 and "cli" in c["extra_errors"]
    assert c["phase_ms"]["pair_ms"] is None  # NaN never reaches the line
    full = bench.sanitize({"extra": huge})
    assert all(len(sv) <= 400 for _, sv in _strings(full))
    assert len(json.dumps(bench.compact_line(line))) < 4096


def test_north_star_genomes_do_not_depend_on_the_rank_count():
    """the strong-scaling job is the same 100 000 genomes at every N: a rank's descriptors are a slice of the job's"""
    import numpy as np
    import bench
    from rabbittclust_amd import api
    whole = bench.north_star_descs(api, "minhash", 0, 7500)
    assert len(whole) == 7500
    for world in (2, 3, 4):
        n_local = 7500 // world
        parts = [bench.north_star_descs(api, "minhash", r * n_local, (r + 1) * n_local) for r in range(world)]
        assert np.array_equal(np.concatenate(parts), whole[: n_local * world])
    assert not np.array_equal(bench.north_star_descs(api, "kssd", 0, 10), whole[:10])

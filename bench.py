#!/usr/bin/env python3
"""bench.py -- MinHash sketch + all-pairs Mash distance (clust-mst hot path) on MI355X.

One step = one pass of the hot path over one batch of synthetic genomes already resident in HBM:
  sketch (k=21, s=1000) -> [N>1: all-gather sketches over RCCL] -> row-sharded N x N sorted-sketch
  intersection -> candidate edges -> minimum spanning forest (Boruvka; N>1: all-reduce(min) per round).
Workload at N=1: BASELINE.json configs[1] = 10k x 5 Mbp synthetic genomes (1 000 families of 10,
substitution rate U[0,0.08]).  N>1: every rank brings its own 10k genomes (weak scaling), the pair
space is (N*10k)^2/2 row-sharded across ranks.

Prints ONE JSON line on rank 0 (see README / DESIGN.md for field definitions).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from rabbittclust_amd import api  # noqa: E402
from rabbittclust_amd import pipeline  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def measured_traffic(kernel, args):
    """HBM-side bytes per launch from the committed rocprofv3 PMC passes (request-size counters,
    profiles/r01_pmc_traffic.json), valid only for the workload they were collected on."""
    try:
        prof = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
        w = prof["workload"]
        if (w["genomes"], w["length"], w["k"], w["s"]) != (args.genomes, args.length, args.k, args.s):
            return None
        return prof["kernels"][kernel]["hbm_bytes_per_launch"]
    except Exception:
        return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genomes", type=int, default=10000, help="genomes per GPU")
    ap.add_argument("--length", type=int, default=5_000_000)
    ap.add_argument("--family", type=int, default=10)
    ap.add_argument("-k", type=int, default=21)
    ap.add_argument("-s", type=int, default=1000)
    ap.add_argument("--threshold", type=float, default=0.05)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-genomes", type=int, default=0, help="0 = 16 per usable core (>= 64)")
    ap.add_argument("--cpu-sample-sketches", type=int, default=8000)
    return ap.parse_args()


def usable_cores():
    """Host cores this process may actually use: affinity mask and cgroup-v2 CPU quota, not the
    machine's core count (the GPU boxes expose 256 CPUs under a 16-CPU quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max" and int(period) > 0:
            n = min(n, max(1, -(-int(quota) // int(period))))
    except Exception:
        pass
    return max(n, 1)


def cpu_baseline(args, ctx, seq, off, sketches_host):
    """Oracle ("port") timed on this box's host cores on a bounded sample of the same workload."""
    from oracle import pyoracle as O
    cores = usable_cores()
    ns = min(args.cpu_sample_genomes or max(64, 16 * cores), len(off) - 1)
    L = int(off[1] - off[0])
    sub = seq[: ns * L].cpu().numpy()
    suboff = np.ascontiguousarray(off[: ns + 1])
    t0 = time.time()
    sk = O.sketch_minhash_batch(sub, suboff, args.k, args.s, threads=cores)
    t_sk = time.time() - t0
    for g in range(min(ns, 4)):
        assert np.array_equal(sk[g], sketches_host[g]), "cpu baseline sketch differs from GPU sketch"
    npair = min(args.cpu_sample_sketches, len(sketches_host))
    flat, start, lens = O.to_csr(sketches_host[:npair])
    t0 = time.time()
    O.mst(flat, start, lens, args.k, 0, args.threshold, threads=cores)
    t_mst = time.time() - t0
    pairs = npair * (npair - 1) // 2
    return {
        "value": pairs / t_mst, "unit": "genome-pairs/s", "cores": cores, "kind": "port",
        "sketch_gbp_per_sec": ns * L / t_sk / 1e9,
        "sample": (f"sketch: {ns} x {L} bp genomes in {t_sk:.2f}s on {cores} threads (OpenMP over genomes, "
                   f"scalar MurmurHash3 port; RabbitSketch's AVX2 kernel is absent from the reference tree); "
                   f"distance: index-based compute_minhash_mst restatement on {npair} of the same sketches "
                   f"({pairs} pairs, only pairs sharing a hash are touched) in {t_mst:.2f}s"),
    }


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1 or os.environ.get("RTC_FORCE_DIST") == "1":  # the env switch exercises the RCCL path on one GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    ctx = api.Context(local)

    n_local = args.genomes
    n_fam = max(1, n_local // args.family)
    n_local = n_fam * args.family
    desc = api.synth_family_descs(n_fam, args.family, global_seed=42 + 1000 * rank)
    off = np.arange(n_local + 1, dtype=np.uint64) * np.uint64(args.length)
    seq = ctx.synth_genomes(desc, off)
    ctx.sync()

    pipe = pipeline.MstPipeline(ctx, k=args.k, sketch_size=args.s, threshold=args.threshold,
                                dist=dist, rank=rank, world=world)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        pipe.step(seq, off)
    barrier()
    t0 = time.perf_counter()
    phases = []
    for _ in range(args.steps):
        phases.append(pipe.step(seq, off))
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=ctx.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    n_total = n_local * world
    pairs = n_total * (n_total - 1) // 2
    bases_total = float(n_local) * args.length * world
    ms_step = dt / args.steps * 1e3
    ph = {k: float(np.mean([p[k] for p in phases])) for k in phases[0]}

    if rank == 0:
        sk_ms = ph["sketch_ms"]
        algo_bytes = float(n_local) * args.length + n_local * args.s * 8.0
        achieved = algo_bytes / (sk_ms * 1e-3) / 1e9
        dist_pairs_local = ph["pairs_local"]
        dist_ach = dist_pairs_local * 2 * args.s * 8.0 / (ph["pair_ms"] * 1e-3) / 1e9
        line = {
            "metric": "genome_pairs_per_sec_end_to_end (sketch + all-pairs Mash distance + MST), k=21 s=1000",
            "value": pairs / (dt / args.steps),
            "unit": "genome-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"{n_total} x {args.length} bp synthetic genomes ({n_local}/GPU), MinHash k={args.k} "
                                   f"s={args.s}, sketch + all-pairs + MST at d={args.threshold}",
                       "genomes_per_gpu": n_local, "genome_length": args.length, "k": args.k,
                       "sketch_size": args.s, "sharding": f"rows/{world}"},
            "sketch_gbp_per_sec": bases_total / (sk_ms * 1e-3) / 1e9,
            "dist_pairs_per_sec": pairs / (ph["dist_ms"] * 1e-3),
            "phase_ms": ph,
            "mst_edges": int(phases[-1]["mst_edges"]),
            "roofline": {"bound": "hbm", "kernel": "sketch_minhash_kernel", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": measured_traffic("sketch_minhash_kernel", args),
                         "note": "algorithmic bytes = 1 B/base + 8 B/hash out per launch; traffic = rocprofv3 PMC "
                                 "bytes per launch (profiles/r01_pmc_traffic.json); the kernel is integer-VALU-issue "
                                 "bound (~95 VALU instructions per k-mer at ~4.1 cycles each), see DESIGN.md 3.1"},
            "roofline_dist": {"bound": "hbm", "kernel": "pair kernel", "achieved": dist_ach, "peak": HBM_PEAK_GBS,
                              "unit": "GB/s", "frac": dist_ach / HBM_PEAK_GBS,
                              "traffic": measured_traffic("pair_tiled_kernel", args),
                              "note": "algorithmic bytes = (|A|+|B|)*8 = 16000 B/pair; tiles are reused from "
                                      "LDS/L2 so this may exceed 1"},
        }
        if not args.no_cpu_baseline and world == 1:
            try:
                line["cpu_baseline"] = cpu_baseline(args, ctx, seq, off, pipe.last_sketches.to_host())
            except Exception as e:  # the baseline is a reported extra; never lose the GPU line
                line["cpu_baseline"] = {"value": None, "error": repr(e)}
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

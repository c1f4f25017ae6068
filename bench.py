#!/usr/bin/env python3
"""bench.py -- sketch + all-pairs Mash distance (clust-mst hot path) on MI355X.

One step = one pass of the hot path over one batch of synthetic genomes already resident in HBM:
  sketch (MinHash k=21 s=1000, or KSSD with --mode kssd) -> [N>1: gather of the sketch rows over RCCL] ->
  row-sharded N x N sorted-sketch intersection with fused candidate-edge emission -> minimum spanning
  forest (device Boruvka; N>1: one all-reduce(MIN) per round in fixed-size mode).

Workloads (BASELINE.json configs; 1 000-family synthetic genomes, substitution rate U[0,0.08]), the batch resident in the
command lines' 2-bit staging format (--staging packed, north_star's "packed sequence"; --staging ascii: as characters):
  N=1            configs[1] = 10 000 x 5 Mbp (--mode kssd: 25 000 x 2 Mbp).
  N>1            --scaling strong (the default): configs[2] = exactly 100 000 x 5 Mbp (--mode kssd: configs[4] =
                 200 000 x 2 Mbp) split over the N ranks -- north_star's own job at every N; its N=1 point is
                 `config3_total_s` / `config5_total_s` of the N=1 line (`bench.py --gpus 1 --scaling strong` runs it as
                 the headline).  --scaling weak: 12 500 (25 000) genomes per GPU, i.e. the same job only at N=8.

`python bench.py --gpus N` with N > 1 and no torchrun environment re-launches itself under torch.distributed.run with N
ranks (one per GPU, backend nccl = RCCL); `n_gpus` in the output is the world size the process group reports.

Output (rank 0, stdout): the compact headline object (< 4 KB: metric, value, roofline, cpu_baseline, phase_ms and the
scalars lifted from the extra workloads) is printed as soon as it is known, the extra workloads follow as ONE line
{"extra": ...} (also written to bench_extra.json), and the compact headline -- now with the extras' scalars -- is
printed again as the LAST line.  README.md ("Reading the bench line") explains every field; no prose travels in the line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
def _profile_jsons():
    """committed PMC summaries, newest round first (profiles/rNN_*pmc_traffic.json)"""
    import glob
    return [os.path.basename(f) for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_*pmc_traffic.json")), reverse=True)]


PROFILE_JSON = _profile_jsons()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--mode", choices=("minhash", "kssd"), default="minhash")
    ap.add_argument("--scaling", choices=("weak", "strong"), default=None,
                    help="strong: BASELINE configs[2] (100 000 x 5 Mbp; --mode kssd: configs[4], 200 000 x 2 Mbp) split over the ranks, "
                         "the default for N > 1; weak: a fixed number of genomes per GPU (N = 1: configs[1], the default there)")
    ap.add_argument("--genomes", type=int, default=0, help="weak: genomes per GPU, strong: genomes of the whole job (0 = the BASELINE shape)")
    ap.add_argument("--length", type=int, default=0, help="bases per genome (0 = 5 000 000 minhash / 2 000 000 kssd)")
    ap.add_argument("--family", type=int, default=10)
    ap.add_argument("-k", type=int, default=21)
    ap.add_argument("-s", type=int, default=1000)
    ap.add_argument("--drlevel", type=int, default=3)
    ap.add_argument("--threshold", type=float, default=0.05)
    ap.add_argument("--staging", choices=("ascii", "packed"), default="packed",
                    help="the batch resident in HBM in the command lines' 2-bit staging format (rtc_sketch_minhash_packed_dev / "
                         "rtc_sketch_kssd_packed_dev: north_star's packed sequence, what clust-mst / clust-greedy run) or as characters "
                         "(rtc_sketch_minhash_dev / rtc_sketch_kssd_dev); the roofline stays at SURVEY 8(d)'s 1 B/base either way")
    ap.add_argument("--comm", choices=("native", "torch"), default="native",
                    help="N>1 collectives: the C ABI's own RCCL communicator (rtc_comm_*, what the C++ hosts use) "
                         "or torch.distributed's; both are RCCL over xGMI (strong scaling is native only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true",
                    help="N=1 only: skip the extra workloads timed after the headline region")
    ap.add_argument("--extra-steps", type=int, default=3)
    ap.add_argument("--only", choices=tuple(n for n, _ in EXTRAS), default=None,
                    help="run ONE of the extra workloads alone and print {\"extra\": {...}} (the profile collection's driver)")
    ap.add_argument("--cli-genomes", type=int, default=2048, help="extra.cli: FASTA files written to /dev/shm")
    ap.add_argument("--cli-genomes-large", type=int, default=8192,
                    help="extra.cli: one more MinHash run of the command line on this many genomes when /dev/shm has the room (0: skip)")
    ap.add_argument("--cpu-sample-genomes", type=int, default=0, help="0 = 1024 (SURVEY 8d: >= 1k genomes)")
    ap.add_argument("--cpu-sample-sketches", type=int, default=8000)
    ap.add_argument("--extra-json", default=os.path.join(ROOT, "bench_extra.json"), help="where the full extras object is also written")
    return ap.parse_args()


def respawn_under_torchrun(args):
    """`python bench.py --gpus N` (N > 1) outside torchrun: launch the N ranks ourselves."""
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} requested but {have} GPU(s) visible; refusing to run fewer ranks")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    # the hosts' driver only supports dmabuf IPC: without HSA_ENABLE_IPC_MODE_LEGACY=0 RCCL's buffer exchange between the
    # rank processes fails with "hipIpcGetMemHandle: invalid argument"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def measured_traffic(kernel, workload):
    """HBM-side bytes per launch from the committed rocprofv3 PMC passes (request-size counters),
    valid only for the workload they were collected on."""
    for name in PROFILE_JSON:
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", name)))
            w = prof["workload"]
            key = next((k for k in prof["kernels"] if k in kernel), None)  # profile keys are kernel-name stems
            if key is not None and all(w.get(k) == v for k, v in workload.items()):
                return prof["kernels"][key]["hbm_bytes_per_launch"], name
        except Exception:
            continue
    return None, None


def measured_valu(kernel, workload, n_xcd=8, n_simd=1024):
    """VALU issue figures of the dominant kernel from the same committed PMC passes: wave64 VALU instructions per
    launch, busy cycles per launch (GRBM_GUI_ACTIVE is summed over the 8 XCDs) and the cycles one SIMD had per
    VALU instruction it issued.  Against the issue cost of the kernel's instruction mix (profiles/
    r02_valu_issue_cost.txt: ~2.5 cycles for xor / and / add / right shift / v_bitop3, ~4.2-4.7 for multiplies,
    64-bit and three-operand forms; ~3.9 for the sketch kernels' mix) this says how busy the VALU pipe is."""
    for name in PROFILE_JSON:
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", name)))
            key = next((k for k in prof["kernels"] if k in kernel), None)
            if key is None or not all(prof["workload"].get(k) == v for k, v in workload.items()):
                continue
            kd = prof["kernels"][key]
            insts, cyc = kd["SQ_INSTS_VALU_per_launch"], kd["GRBM_GUI_ACTIVE_per_launch"] / n_xcd
            return {"wave_insts_per_launch": insts, "gpu_cycles_per_launch": cyc,
                    "insts_per_kmer_wave_step": kd.get("derived", {}).get("valu_insts_per_step"),
                    "simd_cycles_per_wave_inst": cyc * n_simd / insts, "source": "profiles/" + name}
        except Exception:
            continue
    return None


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


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(args, mode, seq, off, sketches_host, shuffled):
    """Oracle ("port") timed on this box's host cores on a bounded sample of the same workload.  The distance half is timed at
    three sample sizes (n, n/2, n/4 of the same sketches) and fitted as t(N) = a N + b N^2: the reference's index loop
    (src/MST.cpp:1408-1470) touches only pairs that share a hash -- for families of fixed size that is linear in N -- and
    closes every 8-row block with a Kruskal pass whose UnionFind is built over all N vertices (:1543-1546), the quadratic term."""
    import numpy as np
    from oracle import pyoracle as O
    cores = usable_cores()
    n = len(off) - 1
    ns = min(args.cpu_sample_genomes or 1024, n)
    L = int(off[1] - off[0])
    sub = seq[: ns * L].cpu().numpy()
    suboff = np.ascontiguousarray(off[: ns + 1])
    t0 = time.time()
    if mode == "minhash":
        sk = O.sketch_minhash_batch(sub, suboff, args.k, args.s, threads=cores)
        impl = O.minhash_impl()
    else:
        sk = O.sketch_kssd_batch(sub, suboff, shuffled, args.k, args.drlevel, threads=cores)
        impl = "KSSD restatement of src/SketchInfo.cpp:994-1252"
    t_sk = time.time() - t0
    for g in range(min(ns, 4)):
        assert np.array_equal(sk[g], sketches_host[g]), "cpu baseline sketch differs from GPU sketch"
    npair = min(args.cpu_sample_sketches, len(sketches_host))
    kk = args.k if mode == "minhash" else 2 * ((args.k + 1) // 2)
    fit_n, fit_s = [], []
    for m in sorted({max(64, npair // 4), max(64, npair // 2), npair}):
        flat, start, lens = O.to_csr(sketches_host[:m], dtype=sketches_host[0].dtype)
        t0 = time.time()
        O.mst(flat, start, lens, kk, 0, args.threshold, threads=cores)
        fit_n.append(m)
        fit_s.append(time.time() - t0)
    t_mst = fit_s[-1]
    A = np.array([[float(m), float(m) * m] for m in fit_n])
    coef = np.linalg.lstsq(A, np.array(fit_s), rcond=None)[0] if len(fit_n) >= 2 else np.array([t_mst / npair, 0.0])
    a, b = float(max(coef[0], 0.0)), float(max(coef[1], 0.0))
    if a == 0.0 and b == 0.0:
        a = t_mst / npair
    pairs = npair * (npair - 1) // 2
    return {
        "value": pairs / t_mst, "unit": "genome-pairs/s", "cores": cores, "kind": "port", "cpu": cpu_model(),
        "sketch_gbp_per_sec": ns * L / t_sk / 1e9,
        "dist_fit": {"law": "t(N) = a*N + b*N^2", "n": fit_n, "s": [round(x, 4) for x in fit_s], "a_s_per_genome": a, "b_s_per_genome2": b},
        "sample": (f"sketch: {ns} x {L} bp on {cores} threads in {t_sk:.2f}s ({impl}); distance: index-based "
                   f"compute_{'minhash' if mode == 'minhash' else 'kssd'}_mst restatement on {npair} of the same sketches in {t_mst:.2f}s")[:200],
    }


def cpu_dist_seconds(cpu, n):
    """the fitted law of cpu_baseline's distance half at n genomes"""
    f = cpu["dist_fit"]
    return f["a_s_per_genome"] * n + f["b_s_per_genome2"] * float(n) * n


def _mean_phases(phases):
    import numpy as np
    return {k: float(np.mean([p[k] for p in phases])) for k in phases[0]}


def extra_kssd(args, ctx, api, pipeline, steps, packed=False):
    """configs[4] per-GPU shape: 25 000 x 2 Mbp, --fast k=21 drlevel=3, sketch + all-pairs + MST.  packed: the batch is
    resident in the command lines' 2-bit staging format and sketched from it (rtc_sketch_kssd_packed_dev)."""
    import numpy as np
    import torch
    from rabbittclust_amd import host
    n, L = 25000, 2_000_000
    shuffled = host.generate_shuffle_dim(6 if 6 - args.drlevel >= 2 else args.drlevel + 2)
    desc = api.synth_family_descs(n // 10, 10, global_seed=42)
    off = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    seq = ctx.synth_genomes(desc, off)
    ctx.sync()
    if packed:  # outside the timed region: the parser's work on the host side of the command lines
        seq = api.pack_staging(seq, int(off[-1]))
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    pipe = pipeline.MstPipeline(ctx, k=args.k, threshold=args.threshold, mode="kssd", drlevel=args.drlevel, shuffled_dim=shuffled)
    pipe.step(seq, off)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ph = _mean_phases([pipe.step(seq, off) for _ in range(steps)])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    sk = pipe.last_sketches
    algo = float(n) * L + float(sk.len.sum().item()) * sk.width
    ach = algo / (ph["sketch_ms"] * 1e-3) / 1e9
    kernel = "sketch_kssd_packed_kernel" if packed else "sketch_kssd_bloom_kernel"
    wl = {"genomes": n, "length": L, "k": args.k, "s": args.s, "mode": "kssd", "staging": "packed" if packed else None}
    traffic, src = measured_traffic(kernel, wl)
    pairs = n * (n - 1) // 2
    out = {
        "workload": f"{n} x {L} bp synthetic genomes, KSSD --fast k={args.k} drlevel={args.drlevel}, sketch + all-pairs + MST"
                    + (", batch resident in the 2-bit staging format" if packed else ""),
        "steps": steps, "ms_per_step": dt * 1e3, "genome_pairs_per_sec": pairs / dt, "dtype": "u64" if sk.width == 8 else "u32",
        "sketch_gbp_per_sec": float(n) * L / (ph["sketch_ms"] * 1e-3) / 1e9, "mean_sketch_size": float(sk.len.float().mean().item()),
        "phase_ms": ph, "mst_edges": int(ph["mst_edges"]),
        "roofline": {"bound": "hbm", "kernel": kernel, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": ("profiles/" + src) if src else None},
    }
    if packed:
        phys = float(n) * L / 4 + float(sk.len.sum().item()) * sk.width
        out["roofline"]["physical_achieved"] = phys / (ph["sketch_ms"] * 1e-3) / 1e9
        out["roofline"]["physical_frac"] = out["roofline"]["physical_achieved"] / HBM_PEAK_GBS
    return out


def extra_kssd_packed(args, ctx, api, pipeline, steps):
    return extra_kssd(args, ctx, api, pipeline, steps, packed=True)


def extra_greedy(args, ctx, api, pipeline, steps):
    """configs[3]: 50 000 prefix genomes of 0.4 .. 2 Mbp, -c 1000 containment sketches + rtc_greedy."""
    import numpy as np
    n, L, fam = 50000, 2_000_000, 10
    rng = np.random.default_rng(1)
    desc = api.synth_family_descs(n // fam, fam, global_seed=43, max_rate=0.04)
    for f in range(n // fam):  # true prefixes of one genome
        desc[f * fam:(f + 1) * fam] = desc[f * fam]
    frac = rng.uniform(0.2, 1.0, size=n)
    frac[::fam] = 1.0
    lens = (frac * L).astype(np.uint64) // 16 * 16
    off = np.zeros(n + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    seq = ctx.synth_genomes(desc, off)
    sizes = np.maximum((lens.astype(np.float64) * 1.0125 / 1000).astype(np.uint32), 100)  # max(fileBytes / 1000, 100)
    ctx.sync()
    sk_ms, gr_s, ncl = [], [], 0
    for it in range(steps + 1):
        ctx.timer_start()
        sk = ctx.sketch_minhash(seq, off, k=args.k, sizes=sizes)
        ms = ctx.timer_stop()
        t0 = time.perf_counter()
        ncl, rep = ctx.greedy(sk, args.threshold, size_cfg=sizes, is_containment=True)
        g = time.perf_counter() - t0
        if it:
            sk_ms.append(ms)
            gr_s.append(g)
    bases = float(off[-1])
    algo = bases + float(sizes.sum()) * 8
    ach = algo / (float(np.mean(sk_ms)) * 1e-3) / 1e9
    traffic, src = measured_traffic("sketch_minhash_kernel", {"genomes": n, "mode": "greedy"})
    # the same genomes from the 2-bit staging format (what clust-greedy sketches per batch): same sketches, timed the same way
    import torch
    ref_hashes, ref_len = sk.hashes.clone(), sk.len.clone()
    del sk
    pb = api.pack_staging(seq, int(off[-1]))
    del seq
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    pk_ms = []
    for it in range(steps + 1):
        ctx.timer_start()
        skp = ctx.sketch_minhash_packed(pb, off, k=args.k, sizes=sizes)
        ms = ctx.timer_stop()
        if it:
            pk_ms.append(ms)
    same = bool(torch.equal(skp.hashes, ref_hashes) and torch.equal(skp.len, ref_len))
    ach_p = algo / (float(np.mean(pk_ms)) * 1e-3) / 1e9
    return {
        "workload": f"{n} prefix genomes of {int(lens.min())} .. {int(lens.max())} bp ({bases / 1e9:.1f} Gbp), clust-greedy -c 1000 "
                    f"(containment sketches of {int(sizes.min())} .. {int(sizes.max())} hashes), d={args.threshold}",
        "steps": steps, "sketch_ms": float(np.mean(sk_ms)), "sketch_gbp_per_sec": bases / (float(np.mean(sk_ms)) * 1e-3) / 1e9,
        "greedy_s": float(np.mean(gr_s)), "genomes_per_sec": n / (float(np.mean(sk_ms)) * 1e-3 + float(np.mean(gr_s))),
        "sketch_ms_packed": float(np.mean(pk_ms)), "packed_sketches_identical": same,
        "roofline_packed": {"bound": "hbm", "kernel": "sketch_minhash_packed_kernel<21, false>", "achieved": ach_p, "peak": HBM_PEAK_GBS,
                            "unit": "GB/s", "frac": ach_p / HBM_PEAK_GBS},
        "clusters": int(ncl), "dtype": "u64",
        "roofline": {"bound": "hbm", "kernel": "sketch_minhash_kernel<21, false>", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": ("profiles/" + src) if src else None},
    }


def _extra_minhash_staging(args, ctx, api, pipeline, steps, packed):
    """The headline workload (configs[1], 10 000 x 5 Mbp) in the staging the headline did NOT use: resident in the command lines'
    2-bit staging format (rtc_sketch_minhash_packed_dev, what clust-mst / clust-greedy run per batch) or as characters
    (rtc_sketch_minhash_dev, the library boundary's other form).  Same genomes, same forest."""
    import numpy as np
    import torch
    n, L = 10000, 5_000_000
    desc = api.synth_family_descs(n // args.family, args.family, global_seed=42)
    off = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    seq = ctx.synth_genomes(desc, off)
    ctx.sync()
    if packed:
        pb = api.pack_staging(seq, int(off[-1]))  # the host parser's work, outside the timed region
        del seq
        seq = pb
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    pipe = pipeline.MstPipeline(ctx, k=args.k, sketch_size=args.s, threshold=args.threshold)
    pipe.step(seq, off)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ph = _mean_phases([pipe.step(seq, off) for _ in range(steps)])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    sk = pipe.last_sketches
    hashes = float(sk.len.sum().item())
    algo = float(n) * L + hashes * 8
    sec = ph["sketch_ms"] * 1e-3
    kernel = "sketch_minhash_packed_kernel" if packed else "sketch_minhash_kernel"
    wl = {"genomes": n, "length": L, "k": args.k, "s": args.s, "mode": "minhash", "staging": "packed" if packed else None}
    traffic, src = measured_traffic(kernel, wl)
    out = {
        "workload": f"{n} x {L} bp synthetic genomes, MinHash k={args.k} s={args.s}, batch resident "
                    + ("in the 2-bit staging format" if packed else "as characters") + f", sketch + all-pairs + MST at d={args.threshold}",
        "steps": steps, "ms_per_step": dt * 1e3, "genome_pairs_per_sec": n * (n - 1) // 2 / dt, "dtype": "u64",
        "sketch_gbp_per_sec": float(n) * L / sec / 1e9, "phase_ms": ph, "mst_edges": int(ph["mst_edges"]),
        "roofline": {"bound": "hbm", "kernel": kernel + "<21, false>", "achieved": algo / sec / 1e9, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": algo / sec / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_source": ("profiles/" + src) if src else None, "valu_issue": measured_valu(kernel, wl)},
    }
    if packed:
        phys = float(n) * L / 4 + hashes * 8
        out["roofline"]["physical_achieved"] = phys / sec / 1e9
        out["roofline"]["physical_frac"] = phys / sec / 1e9 / HBM_PEAK_GBS
    return out


def extra_minhash_packed(args, ctx, api, pipeline, steps):
    return _extra_minhash_staging(args, ctx, api, pipeline, steps, True)


def extra_minhash_ascii(args, ctx, api, pipeline, steps):
    return _extra_minhash_staging(args, ctx, api, pipeline, steps, False)


NORTH_STAR = {"minhash": {"n": 100000, "L": 5_000_000, "chunk": 10000, "seed0": 500, "config": 2},
              "kssd": {"n": 200000, "L": 2_000_000, "chunk": 25000, "seed0": 900, "config": 4}}
NS_UNIT = 2500  # genomes per seed unit: genome g of the job is member g % 2500 of synth_family_descs(250, 10, seed0 + g // 2500)


def north_star_descs(api, mode, g0, g1):
    """descriptors of the job's genomes [g0, g1): the same 100 000 (200 000) genomes whatever the rank count"""
    import numpy as np
    parts = []
    for u in range(g0 // NS_UNIT, (g1 + NS_UNIT - 1) // NS_UNIT):
        d = api.synth_family_descs(NS_UNIT // 10, 10, global_seed=NORTH_STAR[mode]["seed0"] + u)
        parts.append(d[max(g0 - u * NS_UNIT, 0):min(g1 - u * NS_UNIT, NS_UNIT)])
    return np.concatenate(parts)


def north_star_batches(ctx, api, mode, g0, g1, L):
    """This rank's genomes [g0, g1) of the job, resident in HBM as batches in the 2-bit staging format -- what the command lines
    stage per lane (synthesis and the host parser's packing: outside every timed region).  Every rank cuts its rows into the
    same batch sizes (the sharded entry points' protocol)."""
    import numpy as np
    import torch
    n_local = g1 - g0
    nb = max(1, -(-n_local // NORTH_STAR[mode]["chunk"]))
    sizes = [n_local // nb + (1 if i < n_local % nb else 0) for i in range(nb)]
    batches, a = [], g0
    for m in sizes:
        off = np.arange(m + 1, dtype=np.uint64) * np.uint64(L)
        seq = ctx.synth_genomes(north_star_descs(api, mode, a, a + m), off)
        ctx.sync()
        batches.append((api.pack_staging(seq, int(off[-1])), off))
        del seq
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        a += m
    return batches


def north_star_job(args, ctx, api, pipeline, comm, mode, n_total, L, steps, warmup, barrier=None):
    """The north-star job (BASELINE configs[2] / configs[4]) on the ranks of `comm` (an api.Comm; one rank: the whole job on
    this GPU): every rank's genomes resident as packed batches, rtc_sketch_*_packed_sharded per batch into the global sketch
    rows, rtc_mst_sharded (row range of the triangle -> candidate edges -> Boruvka with one all-reduce per round -> host
    distances), the forest cut at d on the host.  Returns (phase means, per-step records, first-step seconds, setup seconds,
    pipe)."""
    import numpy as np
    import torch
    from rabbittclust_amd import host
    world, rank = comm.size, comm.rank
    n_local = n_total // world
    free, _ = torch.cuda.mem_get_info()
    need = n_local * L / 4 * 1.1 + min(n_local, NORTH_STAR[mode]["chunk"]) * L * 1.6 + n_total * (args.s * 8 if mode == "minhash" else 3000)
    if free < need:
        raise MemoryError(f"{free / 1e9:.0f} GB of HBM free, {need / 1e9:.0f} GB needed")
    t_setup = time.perf_counter()
    batches = north_star_batches(ctx, api, mode, rank * n_local, (rank + 1) * n_local, L)
    t_setup = time.perf_counter() - t_setup
    shuffled = host.generate_shuffle_dim(6 if 6 - args.drlevel >= 2 else args.drlevel + 2) if mode == "kssd" else None
    pipe = pipeline.MstPipeline(ctx, k=args.k, sketch_size=args.s, threshold=args.threshold, mode=mode, drlevel=args.drlevel,
                                shuffled_dim=shuffled, comm=pipeline.NativeComm(comm))
    sync = barrier or torch.cuda.synchronize
    rec, first = [], None
    for it in range(warmup):
        t0 = time.perf_counter()
        pipe.step(batches)
        torch.cuda.synchronize()
        first = first or time.perf_counter() - t0
    sync()
    t_all = time.perf_counter()
    for it in range(steps):
        t0 = time.perf_counter()
        ph = pipe.step(batches)
        ph["clusters"] = float(n_total - int(np.count_nonzero(pipe.last_mst["dist"] <= args.threshold)))  # the forest cut (src/MST.cpp:1155-1183)
        ph["total_s"] = time.perf_counter() - t0
        rec.append(ph)
    sync()
    t_all = time.perf_counter() - t_all
    return _mean_phases(rec), rec, first, t_setup, pipe, t_all, batches


def _north_star_1gpu(args, ctx, api, pipeline, steps, mode):
    """A north-star configuration on ONE GPU, whole job, through the same C entry points the N-rank runs use (a communicator of
    one rank): this is the N = 1 point of `bench.py --gpus N --scaling strong`."""
    import numpy as np
    ns = NORTH_STAR[mode]
    n, L = ns["n"], ns["L"]
    comm = api.Comm.init_rank(ctx, 1, 0, None)
    try:
        ph, rec, first, t_setup, pipe, t_all, batches = north_star_job(args, ctx, api, pipeline, comm, mode, n, L, steps, 1)
        pairs = n * (n - 1) // 2
        sk = pipe.last_sketches
        out = {
            "workload": (f"{n} x {L} bp synthetic genomes ({n * L / 1e9:.0f} Gbp), "
                         + (f"MinHash k={args.k} s={args.s}" if mode == "minhash" else f"KSSD --fast k={args.k} drlevel={args.drlevel}")
                         + f", ONE GPU, {len(batches)} resident 2-bit batches ({n * L / 4e9:.0f} GB), sketch + all {pairs:.3g} pairs + MST + "
                         f"clusters at d={args.threshold}; BASELINE configs[{ns['config']}]"),
            "steps": steps, "total_s": t_all / steps, "sketch_s": ph["sketch_ms"] * 1e-3, "pair_ms": ph["pair_ms"], "mst_ms": ph["mst_ms"],
            "pair_path": int(round(ph["pair_path"])), "cand_edges": int(ph["cand_edges"]), "boruvka_rounds": ph["boruvka_rounds"],
            "mst_edges": int(ph["mst_edges"]), "clusters": int(ph["clusters"]), "genome_pairs_per_sec": pairs / (t_all / steps),
            "sketch_gbp_per_sec": n * L / (ph["sketch_ms"] * 1e-3) / 1e9, "first_step_total_s": first, "setup_s": t_setup,
            "dtype": "u64" if sk.width == 8 else "u32", "mean_sketch_size": float(sk.len.float().mean().item()),
        }
        # What each rank of an N-GPU run of this job would do in its pair phase, measured here one rank after the other on the
        # resident sketch set (the row ranges rtc_mst_sharded cuts: equal cost, not equal pairs): the balance of the split and the
        # part of the step that shrinks with N.  The sketch phase divides by N (whole genomes per rank); the Boruvka rounds do not.
        try:
            from rabbittclust_amd.pipeline import triangle_row_ranges
            import torch
            ppipe = pipeline.MstPipeline(ctx, k=sk.k, sketch_size=args.s, threshold=args.threshold)
            shards = {}
            for W in (2, 4, 8):
                b = triangle_row_ranges(n, W, fixed_cols=1.84 * out["mean_sketch_size"])
                ms, ed = [], []
                for r in range(W):
                    for rep in range(2):  # (the first call sizes the list)
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        _, m = ppipe.candidate_edges(sk, b[r], b[r + 1])
                        torch.cuda.synchronize()
                        dt = (time.perf_counter() - t0) * 1e3
                    ms.append(dt)
                    ed.append(int(m))
                shards[str(W)] = {"pair_ms_max": max(ms), "pair_ms_min": min(ms), "pair_ms_sum": sum(ms), "cand_edges_max": max(ed),
                                  "sketch_s_per_rank": out["sketch_s"] / W}
            out["row_shards_on_one_gpu"] = shards
            ppipe = None
        except Exception as e:
            out["row_shards_on_one_gpu"] = {"error": repr(e)[:300]}
        if not args.no_cpu_baseline:
            # this job's own bounded CPU sample: its first genomes again as characters, its first sketches
            try:
                nsmp = min(args.cpu_sample_genomes or 1024, n)
                soff = np.arange(nsmp + 1, dtype=np.uint64) * np.uint64(L)
                sseq = ctx.synth_genomes(north_star_descs(api, mode, 0, nsmp), soff)
                ctx.sync()
                npair = min(args.cpu_sample_sketches, n)
                stride = sk.hashes.numel() // sk.n
                sub = api.SketchSet(sk.hashes.view(sk.n, stride)[:npair].reshape(-1), sk.start[:npair], sk.len[:npair], sk.width, sk.k, sk.kind)
                cpu = cpu_baseline(args, mode, sseq, soff, sub.to_host(), pipe.shuffled_dim)
                del sseq
            except Exception as e:
                cpu = {"value": None, "error": repr(e)}
            _cpu_extrapolation(out, n, L, pairs, cpu)
        return out
    finally:
        comm.close()


def _cpu_extrapolation(out, n, L, pairs, cpu):
    """The reference's CPU path on this job, EXTRAPOLATED from the job's own bounded CPU sample: the sketch half is
    linear in bases; the distance half follows the law fitted from three sample sizes (cpu_baseline)."""
    if not cpu or not cpu.get("value"):
        if cpu and cpu.get("error"):
            out["cpu_extrapolated"] = {"error": cpu["error"][:300]}
        return
    sk_s = n * L / 1e9 / cpu["sketch_gbp_per_sec"]
    pr_s = cpu_dist_seconds(cpu, n)
    out["cpu_extrapolated_s"] = sk_s + pr_s
    out["cpu_extrapolated"] = {
        "sketch_s": sk_s, "dist_s": pr_s, "cores": cpu.get("cores"), "kind": cpu.get("kind"), "dist_fit": cpu["dist_fit"],
        "label": "EXTRAPOLATED, not measured: bases / the sample's sketch rate + the distance law t(N) = a N + b N^2 fitted on "
                 "three sizes of the job's own sketches (index loop linear in N for fixed family size, per-block Kruskal quadratic)",
        "sample": cpu.get("sample")}
    out["gpu_vs_cpu_extrapolated"] = out["cpu_extrapolated_s"] / out["total_s"]


def extra_config3_1gpu(args, ctx, api, pipeline, steps):
    return _north_star_1gpu(args, ctx, api, pipeline, min(steps, 2), "minhash")


def extra_config5_1gpu(args, ctx, api, pipeline, steps):
    return _north_star_1gpu(args, ctx, api, pipeline, min(steps, 2), "kssd")


def extra_weak_first_point(args, ctx, api, pipeline, steps):
    """12 500 x 5 Mbp MinHash: the per-GPU load of the N>1 runs on one GPU (= `bench.py --gpus 1 --genomes 12500`), so that the
    1 -> 8 curve has a first point with the same per-GPU work."""
    import numpy as np
    import torch
    n, L = 12500, 5_000_000
    free, _ = torch.cuda.mem_get_info()
    if free < n * L * 1.2:
        raise MemoryError(f"{free / 1e9:.0f} GB free")
    desc = api.synth_family_descs(n // args.family, args.family, global_seed=42)
    off = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    seq = ctx.synth_genomes(desc, off)
    ctx.sync()
    pipe = pipeline.MstPipeline(ctx, k=args.k, sketch_size=args.s, threshold=args.threshold)
    pipe.step(seq, off)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ph = _mean_phases([pipe.step(seq, off) for _ in range(steps)])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {
        "workload": f"{n} x {L} bp, MinHash k={args.k} s={args.s}: the per-GPU load of the N>1 runs on one GPU "
                    f"(also `python bench.py --gpus 1 --genomes {n}`)",
        "steps": steps, "ms_per_step": dt * 1e3, "genome_pairs_per_sec": n * (n - 1) // 2 / dt,
        "sketch_gbp_per_sec": float(n) * L / (ph["sketch_ms"] * 1e-3) / 1e9, "phase_ms": ph,
    }


DENSE_N, DENSE_FAM, DENSE_RATE, DENSE_L = 10000, 10, 0.01, 500_000


def _dense_u32_25000(args, ctx, api, pipeline, steps):
    """25 000 KSSD sketches (u32 tuples, ~488 each) in 25 families of 1 000 near-identical 2 Mbp genomes: the dense regime at the
    size where wider column blocks of the tiled kernel were costed to start paying; default dispatch."""
    import numpy as np
    import torch
    from rabbittclust_amd import host
    n, fam, L = 25000, 25, 2_000_000
    desc = api.synth_family_descs(fam, n // fam, global_seed=44, max_rate=DENSE_RATE)
    off = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    seq = ctx.synth_genomes(desc, off)
    sk = ctx.sketch_kssd(seq, off, host.generate_shuffle_dim(6), kmer_size=args.k, drlevel=3)
    ctx.sync()
    del seq
    torch.cuda.empty_cache()
    pipe = pipeline.MstPipeline(ctx, k=sk.k, threshold=args.threshold)
    rec = []
    for it in range(steps + 1):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record()
        edges, m = pipe.candidate_edges(sk, 0, n)
        ev[1].record()
        path = ctx.pair_last_path()
        kms = ctx.pair_last_kernel_ms() if path == 2 else float("nan")
        sel, rounds = pipe.boruvka(sk, edges, m)
        mst = pipe.finish(sk, sel)
        ev[2].record()
        torch.cuda.synchronize()
        rec.append({"pair_ms": ev[0].elapsed_time(ev[1]), "mst_ms": ev[1].elapsed_time(ev[2]), "kernel_ms": kms, "cand_edges": float(m),
                    "pair_path": float(path), "mst_edges": float(len(mst))})
    first, ph = rec[0], _mean_phases(rec[1:])
    return {"workload": f"{n} u32 KSSD sketches of {float(sk.len.float().mean().item()):.0f} tuples: {fam} families of {n // fam} genomes ({L} bp, "
                        f"substitution rate <= {DENSE_RATE}), all-pairs candidate edges + MST; default dispatch",
            "pair_path": int(round(ph["pair_path"])), "pair_ms": ph["pair_ms"], "pair_kernel_ms": ph["kernel_ms"], "mst_ms": ph["mst_ms"],
            "first_call_pair_ms": first["pair_ms"], "cand_edges": int(ph["cand_edges"]), "mst_edges": int(ph["mst_edges"]),
            "pair_phase_pairs_per_sec": n * (n - 1) // 2 / (ph["pair_ms"] * 1e-3)}


def extra_dense_pairs(args, ctx, api, pipeline, steps):
    """The dense regime of the distance half: 10 000 u64 sketches (s = 1000) in 10 families of 1 000 near-identical genomes
    (substitution rate <= 1 %): posting lists as long as a family, 3.5e9 co-occurrences -- the input on which the reference's
    posting-list walk (src/MST.cpp:1412-1435) goes quadratic and for which the tiled N x N kernel exists.  The cost rule of
    rtc_pair_edges_dev picks the path itself (no environment override): `pair_path` must read 2.  One step = candidate edges
    of the whole lower triangle (plan + pair_tiled_kernel) + Boruvka + host distances."""
    import numpy as np
    import torch
    n, fam = DENSE_N, DENSE_FAM
    desc = api.synth_family_descs(fam, n // fam, global_seed=42, max_rate=DENSE_RATE)
    off = np.arange(n + 1, dtype=np.uint64) * np.uint64(DENSE_L)
    pipe = pipeline.MstPipeline(ctx, k=args.k, sketch_size=args.s, threshold=args.threshold)
    pairs = n * (n - 1) // 2
    # code objects warm, sketch set cold: a toy set of its own (256 u32 sketches of two families, other buffers) takes the same
    # device path once -- what rtc_warmup does for the command lines -- so that the first launch on the real set below pays
    # what a real run pays for a set it meets once (the look at its density), not the runtime's first mapping of the kernels
    # (u32 sketches: the toy goes through the u32 instantiation of the tiled kernel, so that the kernel statistics of this run's
    # pair_tiled_kernel<unsigned long, ...> are the real set's launches only; the code object is one per translation unit)
    rng = np.random.default_rng(7)
    toy_rows = []
    for f in range(2):
        toy_base = np.unique(rng.integers(0, 1 << 32, size=1100, dtype=np.uint64).astype(np.uint32))[:1000]
        for m in range(128):
            v = toy_base.copy()
            v[rng.integers(0, len(v), size=8)] = rng.integers(0, 1 << 32, size=8, dtype=np.uint64).astype(np.uint32)
            toy_rows.append(np.unique(v))
    toy = api.SketchSet.from_host(toy_rows, ctx.device, k=args.k, kind="kssd", width=4)
    toy_pipe = pipeline.MstPipeline(ctx, k=args.k, sketch_size=args.s, threshold=args.threshold)
    te, tm = toy_pipe.candidate_edges(toy, 0, toy.n)
    toy_path = ctx.pair_last_path()
    toy_pipe.finish(toy, toy_pipe.boruvka(toy, te, tm)[0])
    sk = ctx.sketch_minhash(ctx.synth_genomes(desc, off), off, k=args.k, size=args.s)  # a fresh sketch generation: no memo of this set anywhere
    ctx.sync()
    rec = []
    for it in range(steps + 1):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record()
        edges, m = pipe.candidate_edges(sk, 0, n)
        ev[1].record()
        path = ctx.pair_last_path()
        kms = ctx.pair_last_kernel_ms() if path == 2 else float("nan")
        sel, rounds = pipe.boruvka(sk, edges, m)
        mst = pipe.finish(sk, sel)
        ev[2].record()
        torch.cuda.synchronize()
        rec.append({"pair_ms": ev[0].elapsed_time(ev[1]), "mst_ms": ev[1].elapsed_time(ev[2]), "kernel_ms": kms,
                    "cand_edges": float(m), "pair_path": float(path), "boruvka_rounds": float(rounds), "mst_edges": float(len(mst))})
    first, ph = rec[0], _mean_phases(rec[1:])
    width = sk.width
    avg_len = float(sk.len.float().mean().item())
    bytes_pair = 2 * avg_len * width
    wl = {"genomes": n, "mode": "dense_pairs"}
    traffic, src = measured_traffic("pair_tiled_kernel", wl)
    kern_s = ph["kernel_ms"] * 1e-3
    algo = pairs * bytes_pair / kern_s / 1e9
    big = None
    try:  # the same regime at BASELINE configs[4]'s per-GPU shape: 25 000 u32 KSSD sketches (2 Mbp genomes), 25 families of 1 000
        big = _dense_u32_25000(args, ctx, api, pipeline, steps)
    except Exception as e:
        big = {"error": repr(e)[:300]}
    return {
        "u32_25000": big,
        "workload": f"{n} u64 sketches of {int(avg_len)} hashes: {fam} families of {n // fam} genomes ({DENSE_L} bp, substitution rate "
                    f"<= {DENSE_RATE}), all-pairs candidate edges + MST at d={args.threshold}; default dispatch",
        "steps": steps, "pair_path": int(round(ph["pair_path"])), "pair_ms": ph["pair_ms"], "pair_kernel_ms": ph["kernel_ms"],
        "mst_ms": ph["mst_ms"], "dist_pairs_per_sec": pairs / ((first["pair_ms"] + first["mst_ms"]) * 1e-3),
        "dist_pairs_per_sec_steady": pairs / ((ph["pair_ms"] + ph["mst_ms"]) * 1e-3),
        "pair_phase_pairs_per_sec": pairs / (ph["pair_ms"] * 1e-3), "cand_edges": int(ph["cand_edges"]),
        "mst_edges": int(ph["mst_edges"]), "boruvka_rounds": ph["boruvka_rounds"], "dtype": "u64",
        "first_call_pair_ms": first["pair_ms"], "first_call_mst_ms": first["mst_ms"], "first_call_pair_path": int(first["pair_path"]),
        "warmup_toy_pair_path": int(toy_path),
        "roofline_dist": {"bound": "hbm", "kernel": "pair_tiled_kernel", "bytes_per_pair": bytes_pair,
                          "algorithmic_achieved": algo, "algorithmic_frac": algo / HBM_PEAK_GBS,
                          "achieved": (traffic / kern_s / 1e9) if traffic else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": (traffic / kern_s / 1e9 / HBM_PEAK_GBS) if traffic else None, "traffic": traffic,
                          "traffic_source": ("profiles/" + src) if src else None},
    }


def extra_cli(args, ctx, api, pipeline, steps):
    """The drop-in command line from FASTA files, on the driver's clock: `--cli-genomes` x 5 Mbp genomes written as 80-column
    FASTA to /dev/shm (outside the timed region), then bin/clust-mst -l ... -e with RTC_METRICS_JSON, MinHash and --fast.
    The page cache is warm (tmpfs); PCIe, parsing, HIP start-up and process exit are all inside `wall_s`.  The runs are a second
    apart so that none is charged the driver's asynchronous teardown of the one before it."""
    import shutil
    import tempfile
    import numpy as np
    n, L = int(args.cli_genomes), 5_000_000
    binp = os.path.join(ROOT, "rabbittclust_amd", "bin", "clust-mst")
    if not os.path.exists(binp):
        raise FileNotFoundError(binp)
    need = int(n * L * 1.15) + (64 << 20)  # the plain files and, beside them, an eighth of them gzip'd
    where = None  # tmpfs when it has the room (the page cache is warm either way: the files were just written)
    for cand in ("/dev/shm", tempfile.gettempdir()):
        try:
            if os.path.isdir(cand) and shutil.disk_usage(cand).free > need:
                where = cand
                break
        except OSError:
            pass
    if where is None:
        raise OSError(f"no directory with {need / 1e9:.1f} GB free for the FASTA files")
    tmp = tempfile.mkdtemp(prefix="rtc_bench_cli_", dir=where)
    try:
        t0 = time.time()
        desc = api.synth_family_descs(max(1, n // 8), 8, global_seed=77)[:n]
        nl = np.full((L // 80, 1), 10, dtype=np.uint8)
        paths = []
        chunk = 256
        for c0 in range(0, n, chunk):
            c1 = min(n, c0 + chunk)
            off = np.arange(c1 - c0 + 1, dtype=np.uint64) * np.uint64(L)
            seq = ctx.synth_genomes(desc[c0:c1], off).cpu().numpy()
            for g in range(c0, c1):
                p = os.path.join(tmp, f"g{g:05d}.fna")
                body = np.concatenate([seq[(g - c0) * L:(g - c0 + 1) * L].reshape(-1, 80), nl], axis=1).tobytes()
                with open(p, "wb") as f:
                    f.write(f">g{g} synthetic\n".encode() + body)
                paths.append(p)
            del seq
        with open(os.path.join(tmp, "list.txt"), "w") as f:
            f.write("\n".join(paths) + "\n")
        t_write = time.time() - t0
        out = {"workload": f"{n} x {L} bp genomes as 80-column FASTA files in {where} (written in {t_write:.1f}s, outside the timed "
                           f"region), bin/clust-mst -l -i list -k {args.k} -d {args.threshold} -e, page cache warm, best of three runs "
                           f"started a second after the previous process left",
               "host_cores": usable_cores()}
        def run_cli(tag, list_file, extra, n_files, bases, tool="clust-mst", reps=3):
            best = None
            for rep in range(reps):  # from the second run on the code objects and the files' pages are warm
                # A process that has left is not gone: the driver tears its GPU state down asynchronously (~0.25 s of work), and a
                # process launched inside that window pays it in its own HIP start-up (0.07 -> 0.13-0.26 s) or at its own exit
                # (0.001 -> 0.12 s): tools/cli_timeline.py, TL_SLEEP=0 against 1.  One command line is one process.
                time.sleep(1.0)
                mj = os.path.join(tmp, f"metrics_{tag}.json")
                env = dict(os.environ, RTC_METRICS_JSON=mj)
                t0 = time.perf_counter()
                r = subprocess.run([os.path.join(os.path.dirname(binp), tool), "-l", "-i", list_file, "-k", str(args.k), "-d", str(args.threshold), "-e",
                                    "-o", os.path.join(tmp, f"out_{tag}.cluster")] + extra, capture_output=True, text=True, cwd=tmp, env=env)
                wall = time.perf_counter() - t0
                if r.returncode != 0:
                    raise RuntimeError(f"{tool} {' '.join(extra)} rc={r.returncode}: {r.stderr[-400:]}")
                m = json.load(open(mj))
                cur = {"wall_s": wall, "end_to_end_gbp_per_sec": bases / wall / 1e9,
                       "computing_sketch_s": m.get("computing_sketch_s"), "sketch_phase_gbp_per_sec": m.get("sketch_gbp_per_s"),
                       "generateMST_s": m.get("generateMST_s"), "greedyCluster_s": m.get("greedyCluster_s"), "total_s": m.get("total_s"), "threads": m.get("threads"),
                       "genomes": m.get("genomes"), "clusters": m.get("clusters"), "mst_edges": m.get("mst_edges"),
                       "parse_s": m.get("parse_s"), "parse_gbp_per_sec": m.get("parse_gbp_per_s"),
                       "parse_gbp_per_sec_per_thread": m.get("parse_gbp_per_s_per_thread"), "hip_init_s": m.get("hip_init_s"), "hip_init_exposed_s": m.get("hip_init_exposed_s"),
                       "batches": m.get("batches"), "gpu_copy_ms_per_batch": m.get("gpu_copy_ms_per_batch"),
                       "gpu_sketch_ms_per_batch": m.get("gpu_sketch_ms_per_batch"), "runs_per_genome": m.get("runs_per_genome"),
                       "inflate_gb_per_sec_per_thread": m.get("inflate_gb_per_s_per_thread")}
                if best is None or cur["wall_s"] < best["wall_s"]:
                    best = cur
            return best

        plain_list = os.path.join(tmp, "list.txt")
        out["minhash"] = run_cli("minhash", plain_list, ["-s", str(args.s)], n, n * L)
        out["fast"] = run_cli("fast", plain_list, ["--fast"], n, n * L)
        out["greedy"] = run_cli("greedy", plain_list, ["-c", "1000"], n, n * L, tool="clust-greedy")  # config 3's command line: containment sketches of 5 000 hashes
        # ---- the shapes sketchFiles actually opens (src/SketchInfo.cpp:880-948): gzip'd files, many-contig assemblies ----
        ngz = min(n, max(16, n // 8))
        t0 = time.time()
        subprocess.run("head -%d list.txt | xargs -P %d -n 4 gzip -6 -k" % (ngz, usable_cores()), shell=True, cwd=tmp, check=True)
        t_gz = time.time() - t0
        gz_paths = [p + ".gz" for p in paths[:ngz]]
        with open(os.path.join(tmp, "list_gz.txt"), "w") as f:
            f.write("\n".join(gz_paths) + "\n")
        gz_bytes = sum(os.path.getsize(p) for p in gz_paths)
        g = run_cli("gz", os.path.join(tmp, "list_gz.txt"), ["-s", str(args.s)], ngz, ngz * L)
        g["workload"] = (f"the first {ngz} of the same genomes as .fna.gz (gzip -6, {gz_bytes / 1e9:.2f} GB compressed, written in {t_gz:.1f}s outside "
                         "the timed region), MinHash")
        out["gz"] = g
        for p in gz_paths:
            os.unlink(p)
        for p in paths:
            os.unlink(p)
        # 200-contig assemblies: records of 5-45 kbp (a multiple of the 80-column line), three N runs of 10-500 bases per genome
        t0 = time.time()
        rng = np.random.default_rng(5)
        cpaths = []
        for c0 in range(0, n, chunk):
            c1 = min(n, c0 + chunk)
            off = np.arange(c1 - c0 + 1, dtype=np.uint64) * np.uint64(L)
            seq = ctx.synth_genomes(desc[c0:c1], off).cpu().numpy()
            for gi in range(c0, c1):
                a = seq[(gi - c0) * L:(gi - c0 + 1) * L].copy()
                for _ in range(3):
                    st = int(rng.integers(0, L - 600))
                    a[st:st + int(rng.integers(10, 500))] = ord("N")
                body = np.concatenate([a.reshape(-1, 80), nl], axis=1).tobytes()
                nlines = L // 80
                cuts = [0]
                while cuts[-1] < nlines:
                    cuts.append(min(nlines, cuts[-1] + int(rng.integers(63, 563))))
                p = os.path.join(tmp, f"c{gi:05d}.fna")
                with open(p, "wb") as f:
                    f.write(b"".join(f">g{gi}_contig{r} synthetic\n".encode() + body[cuts[r] * 81:cuts[r + 1] * 81] for r in range(len(cuts) - 1)))
                cpaths.append(p)
            del seq
        with open(os.path.join(tmp, "list_contigs.txt"), "w") as f:
            f.write("\n".join(cpaths) + "\n")
        t_c = time.time() - t0
        c = run_cli("contigs", os.path.join(tmp, "list_contigs.txt"), ["-s", str(args.s)], n, n * L)
        c["workload"] = (f"{n} x {L} bp assemblies of ~200 contigs (records of 5-45 kbp) with three N runs each, plain FASTA (written in {t_c:.1f}s "
                         "outside the timed region), MinHash: every record separator and N stretch is a run of the staging format")
        out["contigs"] = c
        for p in cpaths:
            os.unlink(p)
        # ---- the command line at scale: --cli-genomes-large files (8 192 x 5 Mbp = 41 GB of FASTA), MinHash, when the box has the room ----
        nl_ = int(args.cli_genomes_large)
        if nl_ > n:
            need_l = int(nl_ * L * 1.02) + (1 << 30)
            avail = 0
            try:
                for ln in open("/proc/meminfo"):
                    if ln.startswith("MemAvailable:"):
                        avail = int(ln.split()[1]) * 1024
            except OSError:
                pass
            free_l = shutil.disk_usage(where).free
            if free_l < need_l + (4 << 30) or (where == "/dev/shm" and avail < 2 * need_l):
                out["minhash_large"] = {"skipped": f"{nl_} x {L} bp needs {need_l / 1e9:.0f} GB in {where}: {free_l / 1e9:.0f} GB free there, "
                                                   f"{avail / 1e9:.0f} GB of host memory available"}
            else:
                t0 = time.time()
                desc_l = api.synth_family_descs(max(1, nl_ // 8), 8, global_seed=78)[:nl_]
                lpaths = []
                for c0 in range(0, nl_, chunk):
                    c1 = min(nl_, c0 + chunk)
                    off = np.arange(c1 - c0 + 1, dtype=np.uint64) * np.uint64(L)
                    seq = ctx.synth_genomes(desc_l[c0:c1], off).cpu().numpy()
                    for g in range(c0, c1):
                        p = os.path.join(tmp, f"l{g:05d}.fna")
                        with open(p, "wb") as f:
                            f.write(f">l{g} synthetic\n".encode() + np.concatenate([seq[(g - c0) * L:(g - c0 + 1) * L].reshape(-1, 80), nl], axis=1).tobytes())
                        lpaths.append(p)
                    del seq
                with open(os.path.join(tmp, "list_large.txt"), "w") as f:
                    f.write("\n".join(lpaths) + "\n")
                t_l = time.time() - t0
                big = run_cli("minhash_large", os.path.join(tmp, "list_large.txt"), ["-s", str(args.s)], nl_, nl_ * L, reps=2)
                big["workload"] = f"{nl_} x {L} bp genomes as FASTA files in {where} ({nl_ * L * 1.0125 / 1e9:.0f} GB, written in {t_l:.0f}s outside the timed region), MinHash, best of two"
                out["minhash_large"] = big
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


EXTRAS = (("minhash_ascii", extra_minhash_ascii), ("minhash_packed", extra_minhash_packed), ("kssd", extra_kssd), ("kssd_packed", extra_kssd_packed), ("greedy", extra_greedy),
          ("weak_first_point", extra_weak_first_point), ("dense_pairs", extra_dense_pairs), ("config3_1gpu", extra_config3_1gpu),
          ("config5_1gpu", extra_config5_1gpu), ("cli", extra_cli))


def extra_workloads(args, ctx, api, pipeline, only=None):
    """N=1 only, AFTER the timed headline region: the other single-GPU shapes of BASELINE.json and the two regimes the headline
    workload does not reach, on the same clock.  Each gets `--extra-steps` timed steps after one warm-up.  A failing extra
    never costs the headline line: its entry becomes {"error": ...} (tests/test_gpu_bench.py asserts there is none)."""
    import torch
    out = {}
    steps = max(1, args.extra_steps)
    skip = "minhash_" + getattr(args, "_headline_staging", "packed")  # the headline already ran the workload in that staging
    for name, fn in EXTRAS:
        if (only and name != only) or (not only and name == skip):
            continue
        t0 = time.perf_counter()
        try:
            out[name] = fn(args, ctx, api, pipeline, steps)
        except Exception as e:
            out[name] = {"error": repr(e)[:300]}
        out[name]["extra_wall_s"] = round(time.perf_counter() - t0, 2)
        import gc
        gc.collect()
        torch.cuda.empty_cache()
    return out


# ---- the output protocol -----------------------------------------------------------------------------------------------
MAX_STR = 400       # no string value of any printed object is longer (tests/test_gpu_bench.py)
MAX_COMPACT = 4000  # bytes of the compact headline line


def sanitize(obj):
    """numpy scalars / arrays -> plain JSON values, NaN -> None, strings cut at MAX_STR: whatever an extra returns, the
    printed line stays valid JSON of bounded strings (an array that slipped into an f-string cost round 5 its evidence)."""
    import math
    import numpy as np
    if isinstance(obj, dict):
        return {str(k): sanitize(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [sanitize(v) for v in obj]
    if isinstance(obj, np.ndarray):
        return sanitize(obj.tolist()) if obj.size <= 64 else f"<array of {obj.size}>"
    if isinstance(obj, np.generic):
        obj = obj.item()
    if isinstance(obj, float):
        return obj if math.isfinite(obj) else None
    if isinstance(obj, str):
        return obj if len(obj) <= MAX_STR else obj[:MAX_STR - 3] + "..."
    if obj is None or isinstance(obj, (bool, int)):
        return obj
    return sanitize(str(obj))


def _r(x, nd=4):
    """a float rounded to nd significant digits (the compact line), anything else unchanged"""
    if isinstance(x, float) and x == x and x not in (float("inf"), float("-inf")) and x != 0.0:
        from math import floor, log10
        return round(x, max(0, nd - 1 - int(floor(log10(abs(x))))))
    return x


def _get(d, *path):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


def compact_line(line, extras=None):
    """The headline object the driver parses: the contract's keys, roofline, cpu_baseline, phase_ms and the scalars lifted
    from the extras.  Always < MAX_COMPACT bytes: optional keys are dropped from the end of `optional` until it fits."""
    rf, cb = line["roofline"], line.get("cpu_baseline") or {}
    out = {k: line[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                "vs_baseline", "dtype", "data")}
    out["config"] = {k: line["config"][k] for k in ("workload", "genomes_total", "genomes_per_gpu", "genome_length", "k", "sketch_size", "staging",
                                                    "sharding", "collectives", "pair_path") if k in line["config"]}
    out["sketch_gbp_per_sec"] = line["sketch_gbp_per_sec"]
    out["dist_pairs_per_sec"] = line["dist_pairs_per_sec"]
    out["mst_edges"], out["clusters"] = line["mst_edges"], line.get("clusters")
    out["roofline"] = {k: rf.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "physical_frac", "kernel_ms",
                                              "traffic_source")}
    out["cpu_baseline"] = ({k: cb.get(k) for k in ("value", "unit", "cores", "cpu", "kind", "sketch_gbp_per_sec", "sample")}
                           if cb.get("value") else {"value": None, "error": str(cb.get("error", "skipped"))[:200]})
    if cb.get("sample"):
        out["cpu_baseline"]["sample"] = cb["sample"][:200]
    out["phase_ms"] = {k: line["phase_ms"][k] for k in ("sketch_ms", "gather_ms", "pair_ms", "mst_ms", "dist_ms", "cand_edges", "boruvka_rounds")
                       if k in line["phase_ms"]}
    optional = []
    if line.get("per_rank"):
        pr = line["per_rank"]
        optional.append(("per_rank", {"sketch_ms_max": pr["phase_ms_max"]["sketch_ms"], "gather_ms_max": pr["phase_ms_max"]["gather_ms"],
                                      "pair_ms_max": pr["phase_ms_max"]["pair_ms"], "mst_ms_max": pr["phase_ms_max"]["mst_ms"],
                                      "pair_ms_min": pr["phase_ms_min"]["pair_ms"], "all_reduce_bytes_per_step": pr["all_reduce_bytes_per_step"]}))
    if extras is not None:
        x = extras
        sc = {
            "minhash_packed_ms": line["phase_ms"]["sketch_ms"] if line["config"].get("staging") == "packed" else _get(x, "minhash_packed", "phase_ms", "sketch_ms"),
            "minhash_ascii_ms": line["phase_ms"]["sketch_ms"] if line["config"].get("staging") == "ascii" else _get(x, "minhash_ascii", "phase_ms", "sketch_ms"),
            "kssd_frac": _get(x, "kssd", "roofline", "frac"),
            "kssd_packed_frac": _get(x, "kssd_packed", "roofline", "frac"),
            "kssd_packed_physical_frac": _get(x, "kssd_packed", "roofline", "physical_frac"),
            "greedy_frac": _get(x, "greedy", "roofline", "frac"),
            "dense_pair_kernel_ms": _get(x, "dense_pairs", "pair_kernel_ms"),
            "dense_first_call_ms": _get(x, "dense_pairs", "first_call_pair_ms"),
            "dense25k_pair_ms": _get(x, "dense_pairs", "u32_25000", "pair_ms"),
            "config3_total_s": _get(x, "config3_1gpu", "total_s"),
            "config3_pair_ms": _get(x, "config3_1gpu", "pair_ms"),
            "config3_mst_ms": _get(x, "config3_1gpu", "mst_ms"),
            "config3_clusters": _get(x, "config3_1gpu", "clusters"),
            "config3_vs_cpu_extrapolated": _get(x, "config3_1gpu", "gpu_vs_cpu_extrapolated"),
            "config5_total_s": _get(x, "config5_1gpu", "total_s"),
            "config5_pair_ms": _get(x, "config5_1gpu", "pair_ms"),
            "config5_mst_ms": _get(x, "config5_1gpu", "mst_ms"),
            "config5_vs_cpu_extrapolated": _get(x, "config5_1gpu", "gpu_vs_cpu_extrapolated"),
            "cli_gbp_per_sec": _get(x, "cli", "minhash", "end_to_end_gbp_per_sec"),
            "cli_large_gbp_per_sec": _get(x, "cli", "minhash_large", "end_to_end_gbp_per_sec"),
        }
        optional.append(("extra_scalars", sc))
        failed = [k for k, v in x.items() if isinstance(v, dict) and "error" in v]
        optional.append(("extra_errors", failed))
        optional.append(("extra_file", "bench_extra.json (and the {\"extra\": ...} line above)"))
    for k, v in optional:
        out[k] = v
    out = sanitize(out)

    def rounded(o):
        if isinstance(o, dict):
            return {k: rounded(v) for k, v in o.items()}
        if isinstance(o, list):
            return [rounded(v) for v in o]
        return _r(o, 6)
    out = rounded(out)
    drop = [k for k, _ in optional][::-1] + ["phase_ms"]
    while len(json.dumps(out)) > MAX_COMPACT and drop:
        out.pop(drop.pop(0), None)
    return out


def emit(obj):
    print(json.dumps(sanitize(obj)), flush=True)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn_under_torchrun(args)

    import numpy as np
    import torch
    from rabbittclust_amd import api, pipeline

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1 or os.environ.get("RTC_FORCE_DIST") == "1":  # the env switch exercises the RCCL path on one GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        world = dist.get_world_size()  # what RCCL actually sees
        rank = dist.get_rank()
    if args.gpus != world and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but the process group has {world} rank(s); reporting n_gpus={world}",
              file=sys.stderr)
    ctx = api.Context(local)
    if args.only:
        ex = extra_workloads(args, ctx, api, pipeline, only=args.only)
        emit({"extra": ex})
        return

    mode = args.mode
    scaling = args.scaling or ("strong" if world > 1 else "weak")
    packed = args.staging == "packed"
    length = args.length or NORTH_STAR[mode]["L"]
    shuffled = None
    if mode == "kssd":
        from rabbittclust_amd import host
        shuffled = host.generate_shuffle_dim(6 if 6 - args.drlevel >= 2 else args.drlevel + 2)

    # ---- communicator ----
    comm, comm_kind, nat = None, "none", None
    if dist is not None:
        comm, comm_kind = pipeline.TorchComm(dist, rank, world), "torch.distributed nccl (RCCL)"
    if (dist is not None and args.comm == "native") or scaling == "strong":
        # the id is created on rank 0 and travels through the process group the launcher set up
        ok = 1
        try:
            uid = [api.Comm.unique_id(ctx.lib) if (rank == 0 and dist is not None) else None]
            if dist is not None:
                dist.broadcast_object_list(uid, src=0)
            nat = api.Comm.init_rank(ctx, world, rank, uid[0])
            t = torch.tensor([rank + 1], dtype=torch.int64, device=ctx.device)
            nat.all_reduce(t, "max")
            ok = int(int(t.item()) == world)
        except Exception as e:  # stay on torch.distributed's communicator (also RCCL), say so
            print(f"bench.py rank {rank}: native communicator unavailable ({e!r})", file=sys.stderr)
            ok = 0
        if dist is not None:
            flag = torch.tensor([ok], dtype=torch.int64, device=ctx.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)  # all ranks take the same path
            ok = int(flag.item())
        if ok == 1:
            comm, comm_kind = pipeline.NativeComm(nat), f"rtc_comm ({nat.backend}, C ABI)"
        elif scaling == "strong":
            sys.exit("bench.py: --scaling strong goes through the C ABI's communicator (rtc_comm_*), which is unavailable here")

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    cpu_seq = cpu_off = None
    if scaling == "strong":
        # ---- north_star's own job, split over the ranks: rtc_sketch_*_packed_sharded + rtc_mst_sharded ----
        if not packed:
            sys.exit("bench.py: --scaling strong keeps the genomes resident in the 2-bit staging format (--staging packed)")
        n_total = (args.genomes or NORTH_STAR[mode]["n"]) // (world * 10) * (world * 10)
        n_local = n_total // world
        ph, phases, first, t_setup, pipe, dt, batches = north_star_job(args, ctx, api, pipeline, nat, mode, n_total, length, args.steps,
                                                                       args.warmup, barrier=barrier)
        n_batches = len(batches)
    else:
        n_local = args.genomes or ((10000 if world == 1 else 12500) if mode == "minhash" else 25000)
        n_fam = max(1, n_local // args.family)
        n_local = n_fam * args.family
        n_total = n_local * world
        desc = api.synth_family_descs(n_fam, args.family, global_seed=42 + 1000 * rank)
        off = np.arange(n_local + 1, dtype=np.uint64) * np.uint64(length)
        seq = ctx.synth_genomes(desc, off)
        ctx.sync()
        ns_cpu = min(args.cpu_sample_genomes or 1024, n_local)
        cpu_seq, cpu_off = seq, off
        if packed:  # the batch as the command lines hand it over; packing is the host parser's work, outside the timed region
            cpu_seq = seq[: ns_cpu * length].clone()  # the characters of the CPU baseline's sample
            pb = api.pack_staging(seq, int(off[-1]))
            del seq
            seq = [(pb, off)] if isinstance(comm, pipeline.NativeComm) else pb  # behind the C ABI: the sharded packed entry points
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
        pipe = pipeline.MstPipeline(ctx, k=args.k, sketch_size=args.s, threshold=args.threshold,
                                    mode=mode, drlevel=args.drlevel, shuffled_dim=shuffled,
                                    comm=comm or pipeline.TorchComm(None, rank, world))
        n_batches = 1
        for _ in range(args.warmup):
            pipe.step(seq, off)
        barrier()
        t0 = time.perf_counter()
        phases = []
        for _ in range(args.steps):
            p = pipe.step(seq, off)
            p["clusters"] = float(n_total - int(np.count_nonzero(pipe.last_mst["dist"] <= args.threshold)))
            phases.append(p)
        barrier()
        dt = time.perf_counter() - t0
        ph = _mean_phases(phases)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=ctx.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    pairs = n_total * (n_total - 1) // 2
    bases_total = float(n_total) * length
    ms_step = dt / args.steps * 1e3
    ranks_ph = None
    if dist is not None:  # every rank's phase means, so that the N>1 line shows the spread and not only rank 0
        ranks_ph = [None] * world
        dist.all_gather_object(ranks_ph, ph)

    if rank == 0:
        sk_ms = ph["sketch_ms"]
        sk_all = pipe.last_sketches
        width = sk_all.width
        hashes_local = float(sk_all.len.sum().item()) / world
        algo_bytes = float(n_local) * length + hashes_local * width  # 1 B/base in + width B/hash out, this rank's launches of a step
        achieved = algo_bytes / (sk_ms * 1e-3) / 1e9
        avg_len = float(sk_all.len.float().mean().item())
        wl = {"genomes": n_local, "length": length, "k": args.k, "s": args.s, "mode": mode, "staging": "packed" if packed else None}
        sk_kernel = (("sketch_minhash_packed_kernel" if packed else "sketch_minhash_kernel") if mode == "minhash" else
                     "sketch_kssd_packed_kernel" if packed else "sketch_kssd_bloom_kernel")
        sk_traffic, sk_src = measured_traffic(sk_kernel, wl)
        if sk_traffic is None and n_batches > 1:  # the committed PMC pass is per launch of a 10 000-genome batch
            per = n_local // n_batches
            t1, sk_src = measured_traffic(sk_kernel, dict(wl, genomes=per))
            sk_traffic = t1 * n_batches if t1 else None
        pair_path = int(round(ph.get("pair_path", 2.0)))
        pr_kernel = "pair_join_phase" if pair_path == 3 else "pair_tiled_kernel"
        pr_traffic, pr_src = measured_traffic(pr_kernel, dict(wl, staging=None))  # (the pair phase does not see the staging: its PMC pass is the character run's)
        pr_ach = pr_traffic / (ph["pair_ms"] * 1e-3) / 1e9 if pr_traffic else None
        survey_bytes_pair = 2 * avg_len * width  # (|A| + |B|) * w per genome pair
        what = "MinHash k=%d s=%d" % (args.k, args.s) if mode == "minhash" else "KSSD --fast k=%d drlevel=%d" % (args.k, args.drlevel)
        what += ", batch resident in the 2-bit staging format" if packed else ", batch resident as characters"
        line = {
            "metric": ("genome_pairs_per_sec_end_to_end (sketch + all-pairs Mash distance + MST), k=21 s=1000"
                       if mode == "minhash" else
                       "genome_pairs_per_sec_end_to_end (KSSD sketch + all-pairs Mash distance + MST), --fast k=21"),
            "value": pairs / (dt / args.steps),
            "unit": "genome-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "u64" if width == 8 else "u32", "data": "synthetic",
            "config": {"workload": f"{n_total} x {length} bp synthetic genomes ({n_local}/GPU), {what}, "
                                   f"sketch + all-pairs + MST at d={args.threshold}"
                                   + (f" (BASELINE configs[{NORTH_STAR[mode]['config']}], the same genomes at every N)" if scaling == "strong" and not args.genomes else ""),
                       "genomes_total": n_total, "genomes_per_gpu": n_local, "genome_length": length, "k": args.k,
                       "sketch_size": args.s if mode == "minhash" else round(avg_len, 1), "staging": args.staging, "sharding": f"rows/{world}",
                       "collectives": comm_kind, "pair_path": {3: "inverted join (rtc_pairs_join.hip)", 2: "tiled kernel (rtc_pairs_tiled.hip)"}.get(pair_path, "merge kernel"),
                       "batches_per_gpu": n_batches},
            "sketch_gbp_per_sec": bases_total / (sk_ms * 1e-3) / 1e9,
            "dist_pairs_per_sec": pairs / (ph["dist_ms"] * 1e-3),
            "per_gpu": {"sketch_gbp_per_sec": float(n_local) * length / (sk_ms * 1e-3) / 1e9,
                        "genome_pairs_per_sec": pairs / (dt / args.steps) / world,
                        "dist_pairs_per_sec_rank0": ph["pairs_local"] / (ph["dist_ms"] * 1e-3)},
            "phase_ms": ph,
            "mst_edges": int(phases[-1]["mst_edges"]), "clusters": int(phases[-1]["clusters"]),
            "roofline": {"bound": "hbm", "kernel": sk_kernel + ("<21, false>" if mode == "minhash" and args.k == 21 else ""), "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": sk_traffic, "traffic_source": ("profiles/" + sk_src) if sk_src else None,
                         "kernel_ms": sk_ms / n_batches, "launches_per_step": n_batches,
                         "algorithmic_bytes_per_launch": algo_bytes / n_batches,
                         "valu_issue": measured_valu(sk_kernel, wl)},
            "roofline_dist": {"bound": "hbm", "kernel": pr_kernel, "achieved": pr_ach, "peak": HBM_PEAK_GBS,
                              "unit": "GB/s", "frac": (pr_ach / HBM_PEAK_GBS) if pr_ach else None,
                              "traffic": pr_traffic, "traffic_source": ("profiles/" + pr_src) if pr_src else None,
                              "records_achieved": hashes_local * world * (width + 4) / (ph["pair_ms"] * 1e-3) / 1e9,
                              "survey_8d_bytes_per_pair": survey_bytes_pair,
                              "survey_8d_achieved": ph["pairs_local"] * survey_bytes_pair / (ph["pair_ms"] * 1e-3) / 1e9},
        }
        if packed:
            phys = float(n_local) * length / 4 + hashes_local * width
            line["roofline"]["physical_achieved"] = phys / (sk_ms * 1e-3) / 1e9
            line["roofline"]["physical_frac"] = line["roofline"]["physical_achieved"] / HBM_PEAK_GBS
        if scaling == "strong":
            line["first_step_s"], line["setup_s"] = first, t_setup
        if ranks_ph:
            keys = ("sketch_ms", "gather_ms", "pair_ms", "mst_ms", "dist_ms", "cand_edges", "pairs_local")
            s_fixed = pipe.fixed_size(sk_all)
            rounds = ph["boruvka_rounds"]
            line["per_rank"] = {
                "phase_ms_min": {k: min(r[k] for r in ranks_ph) for k in keys},
                "phase_ms_max": {k: max(r[k] for r in ranks_ph) for k in keys},
                "boruvka_rounds": rounds,
                "all_reduce_bytes_per_step": rounds * n_total * (8 if s_fixed else 16)}
        if not args.no_cpu_baseline:
            try:
                if cpu_seq is None:  # strong scaling: rank 0's first genomes again as characters
                    ns_cpu = min(args.cpu_sample_genomes or 1024, n_local)
                    cpu_off = np.arange(ns_cpu + 1, dtype=np.uint64) * np.uint64(length)
                    cpu_seq = ctx.synth_genomes(north_star_descs(api, mode, 0, ns_cpu), cpu_off)
                    ctx.sync()
                npair = min(args.cpu_sample_sketches, n_local)
                stride = sk_all.hashes.numel() // sk_all.n
                sub = api.SketchSet(sk_all.hashes.view(sk_all.n, stride)[:npair].reshape(-1), sk_all.start[:npair], sk_all.len[:npair],
                                    width, sk_all.k, sk_all.kind)
                line["cpu_baseline"] = cpu_baseline(args, mode, cpu_seq, cpu_off, sub.to_host(), shuffled)
                if scaling == "strong":  # the whole job on this box's host cores, by the fitted law
                    _cpu_extrapolation(line.setdefault("job_vs_cpu", {"total_s": dt / args.steps}), n_total, length, pairs, line["cpu_baseline"])
            except Exception as e:  # the baseline is a reported extra; never lose the GPU line
                line["cpu_baseline"] = {"value": None, "error": repr(e)}
        del cpu_seq
        # ---- the evidence: the compact headline at once, the extras after it, the compact headline again as the LAST line ----
        emit(compact_line(line))
        want_extra = world == 1 and dist is None and not args.no_extra and mode == "minhash" and not args.genomes and not args.length and scaling == "weak"
        extras = None
        if want_extra:
            seq = batches = pb = None
            pipe.last_sketches = pipe._rows = None
            pipe = None
            if nat is not None:
                nat.close()
                nat = None
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            args._headline_staging = args.staging
            extras = extra_workloads(args, ctx, api, pipeline)
        full = sanitize({"headline": line, "extra": extras} if extras is not None else {"headline": line})
        emit(full)
        try:
            with open(args.extra_json, "w") as f:
                json.dump(full, f, indent=1)
        except OSError as e:
            print(f"bench.py: could not write {args.extra_json}: {e}", file=sys.stderr)
        emit(compact_line(line, extras))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

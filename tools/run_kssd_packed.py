#!/usr/bin/env python3
"""KSSD sketch phase from the three forms a batch can take in HBM, same synthetic genomes (SURVEY.md 8d families):
  ascii          rtc_sketch_kssd_dev over characters (the library boundary's form; bench.py --mode kssd)
  unpack+ascii   rtc_unpack_bases_dev + rtc_sketch_kssd_dev (what the command lines ran until round 4)
  packed         rtc_sketch_kssd_packed_dev over the 2-bit staging format
Prints one JSON line: milliseconds (best of --reps, HIP events on the context stream), Gbase/s, and whether the three
tuple sets are identical.  The batch is packed on the GPU with torch in 256 MiB pieces (test plumbing)."""
import argparse
import json
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rabbittclust_amd import api  # noqa: E402
from rabbittclust_amd.host import generate_shuffle_dim  # noqa: E402


def pack_on_gpu(seq, n_bases):
    out = torch.empty(n_bases // 4, dtype=torch.uint8, device=seq.device)
    step = 1 << 28
    for a in range(0, n_bases, step):
        b = min(a + step, n_bases)
        x = seq[a:b]
        if x.numel() < b - a:
            x = torch.cat([x, torch.full((b - a - x.numel(),), ord("N"), dtype=torch.uint8, device=seq.device)])
        c = ((x >> 1) ^ (x >> 2)) & 3
        c = c.view(-1, 4)
        out[a // 4:b // 4] = c[:, 0] | (c[:, 1] << 2) | (c[:, 2] << 4) | (c[:, 3] << 6)
    return out


def runs_on_gpu(seq, n):
    """(start, length) of every maximal stretch outside ACGTacgt, found in pieces of 2^30 characters (a stretch across a
    seam comes out as two runs that touch, which the format allows)"""
    parts = []
    step = 1 << 30
    for a in range(0, n, step):
        up = seq[a:min(a + step, n)] & 0xDF
        bad = ~((up == 65) | (up == 67) | (up == 71) | (up == 84))
        idx = torch.nonzero(bad).view(-1)
        if idx.numel() == 0:
            continue
        brk = torch.ones_like(idx, dtype=torch.bool)
        brk[1:] = idx[1:] != idx[:-1] + 1
        last = torch.ones_like(idx, dtype=torch.bool)
        last[:-1] = brk[1:]
        starts, ends = idx[brk], idx[last] + 1
        parts.append(torch.stack([starts + a, ends - starts], dim=1).reshape(-1))
    if not parts:
        return torch.zeros(0, dtype=torch.int64, device=seq.device)
    return torch.cat(parts).contiguous()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genomes", type=int, default=5000)
    ap.add_argument("--length", type=int, default=2_000_000)
    ap.add_argument("-k", type=int, default=21)
    ap.add_argument("--drlevel", type=int, default=3)
    ap.add_argument("--n-every", type=int, default=0, help="an 8-base run of N every so many bases (0: none)")
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    ctx = api.Context(0)
    n, L = args.genomes, args.length
    desc = api.synth_family_descs(max(n // 10, 1), 10, global_seed=42, n_every=args.n_every)[:n]
    off = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    seq = ctx.synth_genomes(desc, off)
    ctx.sync()
    total = int(off[-1])
    n_bases = (total + 63) // 64 * 64 + 64
    packed = pack_on_gpu(seq, n_bases)
    runs = runs_on_gpu(seq, total)
    runs = torch.cat([runs, torch.tensor([total, n_bases - total], dtype=torch.int64, device=runs.device)])
    torch.cuda.synchronize()
    half_subk = 6 if 6 - args.drlevel >= 2 else args.drlevel + 2
    sd = generate_shuffle_dim(half_subk)
    stride = int(L / 16 ** args.drlevel * 1.5) + 256
    unp = torch.empty(n_bases, dtype=torch.uint8, device=seq.device)

    def t_ascii():
        return ctx.sketch_kssd(seq, off, sd, kmer_size=args.k, drlevel=args.drlevel, stride=stride)

    def t_unpack():
        ctx.check(ctx.lib.rtc_unpack_bases_dev(ctx.h, packed.data_ptr(), n_bases, runs.data_ptr(), runs.numel() // 2, unp.data_ptr()))
        return ctx.sketch_kssd(unp, off, sd, kmer_size=args.k, drlevel=args.drlevel, stride=stride)

    def t_packed():
        return ctx.sketch_kssd_packed(packed, n_bases, runs, off, sd, kmer_size=args.k, drlevel=args.drlevel, stride=stride)

    res, sets = {}, {}
    for name, fn in (("ascii", t_ascii), ("unpack+ascii", t_unpack), ("packed", t_packed)):
        best = None
        for r in range(args.reps + 1):
            ctx.timer_start()
            sk = fn()
            ms = ctx.timer_stop()
            if r and (best is None or ms < best):
                best = ms
        res[name] = {"ms": round(best, 3), "gbase_per_s": round(total / best / 1e6, 1)}
        sets[name] = (sk.hashes.clone(), sk.len.clone())
    same = all(torch.equal(sets["ascii"][1], sets[x][1]) for x in sets)
    if same:
        ln = sets["ascii"][1].to(torch.int64)
        m = torch.arange(stride, device=ln.device)[None, :] < ln[:, None]
        for x in ("unpack+ascii", "packed"):
            same = same and bool(torch.equal(sets["ascii"][0].view(n, -1)[m], sets[x][0].view(n, -1)[m]))
    print(json.dumps({"workload": {"genomes": n, "length": L, "k": args.k, "drlevel": args.drlevel, "runs": int(runs.numel() // 2)},
                      "identical": bool(same), **res}))


if __name__ == "__main__":
    main()

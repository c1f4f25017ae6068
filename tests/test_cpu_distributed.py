"""CPU suite: the N>1 path on 2 gloo ranks.  The row sharding, the per-round all-reduce protocol of
the Boruvka loop (rabbittclust_amd.pipeline.boruvka_rounds), the sketch all-gather and the host-side
union are exercised exactly as on GPUs; only the three per-round device primitives are replaced by a
numpy stand-in defined HERE (test code), so no GPU is needed."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEY_NONE = 0x7FFFFFFFFFFFFFFF


class NumpyBoruvkaBackend:
    """Same contract as pipeline.HipBoruvkaBackend over this rank's local (i, j, common) edges."""

    def __init__(self, edges, lens, is_containment):
        self.e, self.lens, self.ic = edges, lens.astype(np.int64), is_containment
        self.device = torch.device("cpu")
        i, j, c = edges[:, 0], edges[:, 1], edges[:, 2].astype(np.float64)
        sa, sb = self.lens[i], self.lens[j]
        denom = np.minimum(sa, sb).astype(np.float64) if is_containment else (sa + sb - edges[:, 2]).astype(np.float64)
        J = c / denom
        self.key = (np.uint64(0x4000000000000000) - J.view(np.uint64)).astype(np.uint64)
        self.id = (i.astype(np.uint64) << np.uint64(32)) | j.astype(np.uint64)

    def _cross(self, comp):
        c = comp.numpy().view(np.uint32)
        ci, cj = c[self.e[:, 0]], c[self.e[:, 1]]
        return ci, cj, ci != cj

    def minweight(self, comp, wkey):
        w = np.full(len(comp), KEY_NONE, dtype=np.uint64)
        ci, cj, x = self._cross(comp)
        np.minimum.at(w, ci[x], self.key[x]); np.minimum.at(w, cj[x], self.key[x])
        wkey.copy_(torch.from_numpy(w.view(np.int64)))

    def minedge(self, comp, wkey, ekey):
        w = wkey.numpy().view(np.uint64)
        e = np.full(len(comp), KEY_NONE, dtype=np.uint64)
        ci, cj, x = self._cross(comp)
        a = x & (self.key == w[ci]); b = x & (self.key == w[cj])
        np.minimum.at(e, ci[a], self.id[a]); np.minimum.at(e, cj[b], self.id[b])
        ekey.copy_(torch.from_numpy(e.view(np.int64)))

    def fetch(self, comp, ekey, ecommon):
        ek = ekey.numpy().view(np.uint64)
        out = np.zeros(len(comp), dtype=np.uint32)
        ci, cj, x = self._cross(comp)
        a = x & (ek[ci] == self.id); b = x & (ek[cj] == self.id)
        out[ci[a]] = self.e[a, 2]; out[cj[b]] = self.e[b, 2]
        ecommon.copy_(torch.from_numpy(out.view(np.int32)))


def _make_sketches(seed, n):
    rng = np.random.default_rng(seed)
    pool = np.unique(rng.integers(1, 1 << 60, size=1500, dtype=np.uint64))
    return [np.sort(rng.choice(pool, size=int(rng.integers(5, 150)), replace=False)) for _ in range(n)]


def _worker(rank, world, port, n, containment, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rabbittclust_amd import _lib, api, pipeline
    lib = _lib.load()
    # each rank "sketches" its own n/world genomes, then all-gathers (strided layout, like the GPU path)
    sk_all = _make_sketches(77, n)
    n_local = n // world
    stride = 160
    mine = sk_all[rank * n_local:(rank + 1) * n_local]
    h = np.zeros((n_local, stride), dtype=np.uint64)
    ln = np.zeros(n_local, dtype=np.int32)
    for g, s in enumerate(mine):
        h[g, :len(s)] = s; ln[g] = len(s)
    local = api.SketchSet(torch.from_numpy(h.view(np.int64).reshape(-1)), torch.arange(n_local) * stride,
                          torch.from_numpy(ln), 8, 21, "minhash")
    pipe = pipeline.MstPipeline.__new__(pipeline.MstPipeline)
    pipe.world, pipe.rank, pipe.dist = world, rank, dist
    sk = pipe.gather_sketches(local)
    got = sk.to_host()
    assert all(np.array_equal(a, b) for a, b in zip(got, sk_all))
    # two-part all-gather used by the overlapped multi-GPU sketch phase: global order is
    # [part A of rank 0..W-1 | part B of rank 0..W-1]
    split = pipeline.MstPipeline.split_point(n_local)
    assert 0 < split < n_local
    called = []
    skp, works = pipe.gather_parts(torch.from_numpy(h.view(np.int64)), torch.from_numpy(ln), [(0, split), (split, n_local)], 21,
                                   before_part=lambda a, b: called.append((a, b)))
    for w in works:
        w.wait()
    assert called == [(0, split), (split, n_local)]
    want_order = [r * n_local + i for r in range(world) for i in range(split)] + \
                 [r * n_local + i for r in range(world) for i in range(split, n_local)]
    gotp = skp.to_host()
    assert len(gotp) == n and all(np.array_equal(gotp[q], sk_all[g]) for q, g in enumerate(want_order))
    # this rank's rows of the strict lower triangle
    b = pipeline.triangle_row_ranges(n, world)
    radio = api.mst_radio(0.05, 21)
    edges = []
    for i in range(max(b[rank], 1), b[rank + 1]):
        for j in range(i):
            c = len(np.intersect1d(sk_all[i], sk_all[j]))
            la, lb = len(sk_all[i]), len(sk_all[j])
            if c and max(la, lb) <= radio * min(la, lb):
                edges.append((i, j, c))
    edges = np.array(edges, dtype=np.uint32).reshape(-1, 3)
    lens = np.array([len(s) for s in sk_all], dtype=np.uint32)
    backend = NumpyBoruvkaBackend(edges, lens, containment)
    sel, rounds = pipeline.boruvka_rounds(backend, n, lib, dist, world)
    out = np.zeros(max(len(sel), 1), dtype=api.EDGE_DT)
    sel = np.ascontiguousarray(sel)
    import ctypes as C
    st = lib.rtc_edges_to_mst_host(sel.ctypes.data_as(C.c_void_p), len(sel), lens.ctypes.data_as(C.c_void_p), 21,
                                   int(containment), out.ctypes.data_as(C.c_void_p))
    assert st == 0
    q.put((rank, out[:len(sel)].copy(), rounds, len(edges)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("containment", [False, True])
def test_two_rank_boruvka_matches_oracle(oracle, containment):
    n, world = 90, 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, containment, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    sk_all = _make_sketches(77, n)
    flat, start, lens = oracle.to_csr(sk_all)
    want = oracle.mst(flat, start, lens, 21, containment, 0.05, threads=1)
    # both ranks hold the identical forest, equal (as a weight multiset) to the oracle's
    assert np.array_equal(res[0][1], res[1][1])
    assert np.array_equal(np.sort(res[0][1]["dist"]).view(np.uint64), np.sort(want["dist"]).view(np.uint64))
    assert res[0][3] > 0 and res[1][3] > 0  # both shards contributed edges


def test_triangle_row_ranges_balance():
    from rabbittclust_amd.pipeline import triangle_row_ranges
    for n, w in ((100000, 8), (10000, 4), (17, 3), (1, 2)):
        b = triangle_row_ranges(n, w)
        assert b[0] == 0 and b[-1] == n and all(b[i] <= b[i + 1] for i in range(w))
        if n >= 1000:
            work = [(b[i + 1] ** 2 - b[i] ** 2) / 2 for i in range(w)]
            assert max(work) / (sum(work) / w) < 1.02


def test_triangle_row_ranges_with_fixed_row_cost():
    """Cost-aware split: row i weighs (i + fixed_cols); ranges stay contiguous, cover [0, n) and
    balance that weight, so the rank that owns the short rows at the top gets fewer pairs."""
    from rabbittclust_amd.pipeline import triangle_row_ranges
    n, w, c = 80000, 8, 8800.0
    b = triangle_row_ranges(n, w, fixed_cols=c)
    assert b[0] == 0 and b[-1] == n and all(b[i] < b[i + 1] for i in range(w))
    cost = [(b[i + 1] ** 2 - b[i] ** 2) / 2 + c * (b[i + 1] - b[i]) for i in range(w)]
    assert max(cost) / (sum(cost) / w) < 1.01
    pairs = [(b[i + 1] ** 2 - b[i] ** 2) / 2 for i in range(w)]
    assert pairs[0] < pairs[-1]
    assert triangle_row_ranges(n, w, fixed_cols=0.0) == triangle_row_ranges(n, w)

"""CPU suite: the N>1 path on 2 gloo ranks.  The row sharding, the per-round all-reduce protocol of
the Boruvka loop (rabbittclust_amd.pipeline.boruvka_rounds), the sketch all-gathers (MinHash two-part
and KSSD one-part) into the canonical genome order are exercised exactly as on GPUs; only the
per-round device primitives are replaced by a numpy stand-in defined HERE (test code), so no GPU is
needed."""
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
    """Same contract as pipeline.HipBoruvkaBackend over this rank's local (i, j, common) edges: the
    per-round minimum passes and the hooking union (rtc_mst.hip: boruvka_hook_kernel /
    boruvka_relabel_kernel) restated in numpy."""

    def __init__(self, edges, lens, is_containment, n):
        self.e, self.lens, self.ic, self.n = edges, lens.astype(np.int64), is_containment, n
        self.device = torch.device("cpu")
        i, j, c = edges[:, 0], edges[:, 1], edges[:, 2].astype(np.float64)
        sa, sb = self.lens[i], self.lens[j]
        denom = np.minimum(sa, sb).astype(np.float64) if is_containment else (sa + sb - edges[:, 2]).astype(np.float64)
        J = c / denom
        self.key = (np.uint64(0x4000000000000000) - J.view(np.uint64)).astype(np.uint64)
        self.id = (i.astype(np.uint64) << np.uint64(32)) | j.astype(np.uint64)

    def init(self):
        self.comp = np.arange(self.n, dtype=np.uint32)
        self.sel = []

    def _cross(self):
        ci, cj = self.comp[self.e[:, 0]], self.comp[self.e[:, 1]]
        return ci, cj, ci != cj

    def minkey(self, s_fixed, key):
        from rabbittclust_amd import _lib
        B = _lib.load().rtc_boruvka_key_bits(self.n, s_fixed)
        assert B > 0
        k = ((np.uint64(s_fixed) - self.e[:, 2].astype(np.uint64)) << np.uint64(2 * B)) | \
            (self.e[:, 0].astype(np.uint64) << np.uint64(B)) | self.e[:, 1].astype(np.uint64)
        w = np.full(self.n, KEY_NONE, dtype=np.uint64)
        ci, cj, x = self._cross()
        np.minimum.at(w, ci[x], k[x]); np.minimum.at(w, cj[x], k[x])
        key.copy_(torch.from_numpy(w.view(np.int64)))

    def minweight(self, wkey):
        w = np.full(self.n, KEY_NONE, dtype=np.uint64)
        ci, cj, x = self._cross()
        np.minimum.at(w, ci[x], self.key[x]); np.minimum.at(w, cj[x], self.key[x])
        wkey.copy_(torch.from_numpy(w.view(np.int64)))

    def minedge(self, wkey, ekey):
        w = wkey.numpy().view(np.uint64)
        e = np.full(self.n, KEY_NONE, dtype=np.uint64)
        ci, cj, x = self._cross()
        a = x & (self.key == w[ci]); b = x & (self.key == w[cj])
        np.minimum.at(e, ci[a], self.id[a]); np.minimum.at(e, cj[b], self.id[b])
        ekey.copy_(torch.from_numpy(e.view(np.int64)))

    def fetch(self, ekey, ecommon):
        ek = ekey.numpy().view(np.uint64)
        out = np.zeros(self.n, dtype=np.uint32)
        ci, cj, x = self._cross()
        a = x & (ek[ci] == self.id); b = x & (ek[cj] == self.id)
        out[ci[a]] = self.e[a, 2]; out[cj[b]] = self.e[b, 2]
        ecommon.copy_(torch.from_numpy(out.view(np.int32)))

    def _round_edge(self, s_fixed, key, ecommon, c):
        k = int(key[c])
        if k == KEY_NONE:
            return None
        if s_fixed:
            from rabbittclust_amd import _lib
            B = _lib.load().rtc_boruvka_key_bits(self.n, s_fixed)
            mask = (1 << B) - 1
            return (k >> B) & mask, k & mask, s_fixed - (k >> (2 * B))
        return k >> 32, k & 0xFFFFFFFF, int(ecommon[c])

    def union(self, s_fixed, key, ecommon):
        key = key.numpy().view(np.uint64)
        ecommon = None if ecommon is None else ecommon.numpy().view(np.uint32)
        comp, n = self.comp, self.n
        succ = np.arange(n, dtype=np.uint32)
        added = 0

        def other(c):
            ed = self._round_edge(s_fixed, key, ecommon, c)
            if ed is None:
                return None, None
            ci, cj = comp[ed[0]], comp[ed[1]]
            return (cj if ci == c else ci), ed

        for v in range(n):
            if comp[v] != v:
                continue
            d, ed = other(v)
            if d is None:
                continue
            d2, _ = other(d)
            mutual = d2 is not None and d2 == v
            if mutual and v < d:
                self.sel.append(ed); added += 1
            else:
                succ[v] = d
                if not mutual:
                    self.sel.append(ed); added += 1
        for v in range(n):
            r = comp[v]
            while succ[r] != r:
                r = succ[r]
            comp[v] = r
        return added

    def selected(self):
        from rabbittclust_amd import api
        out = np.zeros(len(self.sel), dtype=api.CEDGE_DT)
        for q, (i, j, c) in enumerate(self.sel):
            out[q] = (i, j, c)
        return out


class CountingComm:
    """pipeline.TorchComm that counts the collectives (one all-reduce per round in fixed-size mode)."""

    def __init__(self, inner):
        self.inner, self.reduces = inner, 0
        self.dist, self.rank, self.world = inner.dist, inner.rank, inner.world

    active = True

    def all_reduce_min(self, t):
        self.reduces += 1
        self.inner.all_reduce_min(t)

    def all_reduce_max(self, t):
        self.reduces += 1
        self.inner.all_reduce_max(t)

    def all_gather(self, out, inp, async_op=False):
        return self.inner.all_gather(out, inp, async_op=async_op)


def _make_sketches(seed, n, fixed=0, width=8):
    """variable sizes 5..149 (the containment / KSSD shape) or all `fixed` (the -s shape)"""
    rng = np.random.default_rng(seed)
    hi = 1 << 60 if width == 8 else 1 << 31
    dt = np.uint64 if width == 8 else np.uint32
    pool = np.unique(rng.integers(1, hi, size=1500, dtype=np.uint64)).astype(dt)
    return [np.sort(rng.choice(pool, size=fixed or int(rng.integers(5, 150)), replace=False)) for _ in range(n)]


def _worker(rank, world, port, n, containment, fixed, width, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rabbittclust_amd import _lib, api, pipeline
    lib = _lib.load()
    np_dt, t_dt = (np.uint64, np.int64) if width == 8 else (np.uint32, np.int32)
    # each rank "sketches" its own n/world genomes, then all-gathers (strided layout, like the GPU path)
    sk_all = _make_sketches(77, n, fixed, width)
    n_local = n // world
    stride = 160
    mine = sk_all[rank * n_local:(rank + 1) * n_local]
    h = np.zeros((n_local, stride), dtype=np_dt)
    ln = np.zeros(n_local, dtype=np.int32)
    for g, s in enumerate(mine):
        h[g, :len(s)] = s; ln[g] = len(s)
    local = api.SketchSet(torch.from_numpy(h.view(t_dt).reshape(-1)), torch.arange(n_local) * stride,
                          torch.from_numpy(ln), width, 21, "minhash" if width == 8 else "kssd")
    comm = CountingComm(pipeline.TorchComm(dist, rank, world))
    pipe = pipeline.MstPipeline(None, comm=comm)
    pipe._check_equal_counts(n_local)
    sk = pipe.gather_sketches(local)
    got = sk.to_host()
    assert all(np.array_equal(a, b) for a, b in zip(got, sk_all))
    # two-part all-gather used by the overlapped multi-GPU sketch phase: the assembled order is canonical
    # (genome g of rank r at r*n_local + g), so multi-GPU node ids equal single-GPU node ids
    split = pipeline.MstPipeline.split_point(n_local)
    assert 0 < split < n_local
    called = []
    finish, works = pipe.gather_parts(torch.from_numpy(h.view(t_dt)), torch.from_numpy(ln), [(0, split), (split, n_local)],
                                      21, before_part=lambda a, b: called.append((a, b)), width=width)
    for w in works:
        w.wait()
    skp = finish()
    assert called == [(0, split), (split, n_local)]
    gotp = skp.to_host()
    assert len(gotp) == n and all(np.array_equal(gotp[g], sk_all[g]) for g in range(n))
    # one-part gather (the KSSD step) takes the no-copy route
    finish1, works1 = pipe.gather_parts(torch.from_numpy(h.view(t_dt)), torch.from_numpy(ln), [(0, n_local)], 21, width=width)
    for w in works1:
        w.wait()
    assert all(np.array_equal(a, b) for a, b in zip(finish1().to_host(), sk_all))
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
    backend = NumpyBoruvkaBackend(edges, lens, containment, n)
    s_fixed = fixed if lib.rtc_boruvka_key_bits(n, fixed) else 0
    before = comm.reduces
    sel, rounds = pipeline.boruvka_rounds(backend, n, comm, s_fixed)
    reduces = comm.reduces - before
    out = np.zeros(max(len(sel), 1), dtype=api.EDGE_DT)
    sel = np.ascontiguousarray(sel)
    import ctypes as C
    st = lib.rtc_edges_to_mst_host(sel.ctypes.data_as(C.c_void_p), len(sel), lens.ctypes.data_as(C.c_void_p), 21,
                                   int(containment), out.ctypes.data_as(C.c_void_p))
    assert st == 0
    q.put((rank, out[:len(sel)].copy(), rounds, len(edges), reduces))
    dist.barrier()
    dist.destroy_process_group()


def _run_two_ranks(n, containment, fixed, width):
    world = 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, containment, fixed, width, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize("containment", [False, True])
def test_two_rank_boruvka_matches_oracle(oracle, containment):
    """variable sketch sizes: three all-reduces per round (MIN weight, MIN edge id, MAX common)"""
    n = 90
    res = _run_two_ranks(n, containment, 0, 8)
    sk_all = _make_sketches(77, n)
    flat, start, lens = oracle.to_csr(sk_all)
    want = oracle.mst(flat, start, lens, 21, containment, 0.05, threads=1)
    # both ranks hold the identical forest, equal (as a weight multiset) to the oracle's
    assert np.array_equal(res[0][1], res[1][1])
    assert np.array_equal(np.sort(res[0][1]["dist"]).view(np.uint64), np.sort(want["dist"]).view(np.uint64))
    assert res[0][3] > 0 and res[1][3] > 0  # both shards contributed edges
    assert res[0][4] == 3 * res[0][2]


def test_two_rank_boruvka_fixed_size_one_allreduce_per_round(oracle):
    """fixed sketch size (BASELINE configs 2/3): the fused (s - common | i | j) key needs exactly one
    all-reduce(MIN) per Boruvka round and gives the oracle's forest weights"""
    n, fixed = 90, 120
    res = _run_two_ranks(n, False, fixed, 8)
    sk_all = _make_sketches(77, n, fixed)
    flat, start, lens = oracle.to_csr(sk_all)
    want = oracle.mst(flat, start, lens, 21, False, 0.05, threads=1)
    assert np.array_equal(res[0][1], res[1][1])
    assert np.array_equal(np.sort(res[0][1]["dist"]).view(np.uint64), np.sort(want["dist"]).view(np.uint64))
    assert res[0][2] >= 2 and res[0][4] == res[0][2], "one all-reduce per round"


def test_two_rank_kssd_shape_u32_sketches(oracle):
    """the --fast (KSSD) multi-GPU step: u32 tuples of varying count, one-part all-gather, row-sharded
    Boruvka; forest weights equal compute_kssd_mst's"""
    n = 90
    res = _run_two_ranks(n, False, 0, 4)
    sk_all = _make_sketches(77, n, 0, 4)
    flat, start, lens = oracle.to_csr(sk_all, dtype=np.uint32)
    want = oracle.mst(flat, start, lens, 21, False, 0.05, threads=1)  # the worker evaluates distances with k = 21 too
    assert np.array_equal(res[0][1], res[1][1])
    assert np.array_equal(np.sort(res[0][1]["dist"]).view(np.uint64), np.sort(want["dist"]).view(np.uint64))
    assert res[0][4] == 3 * res[0][2]


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

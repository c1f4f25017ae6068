"""clust-mst hot path as one step: sketch -> [all-gather] -> row-sharded all-pairs -> edges -> MSF.

Mirrors clust_from_genomes -> compute_sketches -> compute_clusters of the reference
(src/sub_command.cpp:2302-2315, :2858-2889, :2924-3053) from the sketching call down to the
`vector<EdgeInfo> mst`.  Multi-GPU: one process per GPU; sketches are all-gathered once (RCCL),
the strict lower triangle of the N x N pair space is split into contiguous row ranges of equal
area, and every Boruvka round all-reduces (MIN) two u64 arrays (weight key, then edge id).
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from .api import CEDGE_DT, EDGE_DT, SketchSet, _np_ptr, _t_ptr, mst_radio

KEY_NONE = 0x7FFFFFFFFFFFFFFF


def triangle_row_ranges(n, world, fixed_cols=0.0):
    """Contiguous row ranges [b_r, b_{r+1}) of the strict lower triangle with ~equal cost.
    Row i costs (i + fixed_cols): i pairs plus the per-row-block work that does not depend on the
    row's length (building the LDS tables of its 64-row block costs as much as ~8.8 columns per
    sketch hash on MI355X: tools/sim_rank.py).  With fixed_cols = 0 the split is by pair count,
    boundaries n*sqrt(r/world)."""
    c = float(fixed_cols)
    area = n * n / 2.0 + c * n
    b = [int(round(-c + math.sqrt(c * c + 2.0 * area * r / world))) for r in range(world + 1)]
    b[0], b[-1] = 0, n
    for r in range(1, world + 1):
        b[r] = min(max(b[r], b[r - 1]), n)
    return b


class HipBoruvkaBackend:
    """Per-round primitives on this rank's candidate edges (device tensors, HIP kernels)."""

    def __init__(self, ctx, sk, edges, m, is_containment):
        self.ctx, self.sk, self.edges, self.m, self.ic = ctx, sk, edges, m, int(is_containment)
        self.device = ctx.device

    def minweight(self, comp, wkey):
        c = self.ctx
        c.check(c.lib.rtc_boruvka_minweight_dev(c.h, _t_ptr(self.edges), self.m, _t_ptr(self.sk.len), self.ic,
                                                _t_ptr(comp), self.sk.n, _t_ptr(wkey)))

    def minedge(self, comp, wkey, ekey):
        c = self.ctx
        c.check(c.lib.rtc_boruvka_minedge_dev(c.h, _t_ptr(self.edges), self.m, _t_ptr(self.sk.len), self.ic,
                                              _t_ptr(comp), self.sk.n, _t_ptr(wkey), _t_ptr(ekey)))

    def fetch(self, comp, ekey, ecommon):
        c = self.ctx
        c.check(c.lib.rtc_boruvka_fetch_dev(c.h, _t_ptr(self.edges), self.m, _t_ptr(comp), self.sk.n,
                                            _t_ptr(ekey), _t_ptr(ecommon)))


def boruvka_rounds(backend, n, lib, dist=None, world=1):
    """Boruvka over row-sharded candidate edges.  Per round: local per-component minimum weight key
    -> all-reduce(MIN) -> local minimum edge id among edges attaining it -> all-reduce(MIN) ->
    owner publishes `common` -> all-reduce(MAX); then every rank applies the identical host-side
    union (rtc_boruvka_merge_host).  `backend` supplies the three local primitives."""
    dev = backend.device
    wkey = torch.empty(n, dtype=torch.int64, device=dev)
    ekey = torch.empty(n, dtype=torch.int64, device=dev)
    ecommon = torch.empty(n, dtype=torch.int32, device=dev)
    comp_h = np.arange(n, dtype=np.uint32)
    comp = torch.empty(n, dtype=torch.int32, device=dev)
    sel = np.zeros(max(n, 1), dtype=CEDGE_DT)
    nsel, added = C.c_uint64(0), C.c_uint64(0)
    rounds = 0
    for _ in range(64):
        comp.copy_(torch.from_numpy(comp_h.view(np.int32)))
        backend.minweight(comp, wkey)
        if world > 1:
            dist.all_reduce(wkey, op=dist.ReduceOp.MIN)
        backend.minedge(comp, wkey, ekey)
        if world > 1:
            dist.all_reduce(ekey, op=dist.ReduceOp.MIN)
        backend.fetch(comp, ekey, ecommon)
        if world > 1:
            dist.all_reduce(ecommon, op=dist.ReduceOp.MAX)
        ekey_h = np.ascontiguousarray(ekey.cpu().numpy().view(np.uint64))
        ecommon_h = np.ascontiguousarray(ecommon.cpu().numpy().view(np.uint32))
        st = lib.rtc_boruvka_merge_host(n, _np_ptr(ekey_h), _np_ptr(ecommon_h), _np_ptr(comp_h),
                                        _np_ptr(sel), C.byref(nsel), C.byref(added))
        if st != _lib.RTC_OK:
            raise _lib.RtcError(st, "rtc_boruvka_merge_host")
        rounds += 1
        if added.value == 0:
            break
    return sel[: nsel.value], rounds


class MstPipeline:
    def __init__(self, ctx, k=21, sketch_size=1000, threshold=0.05, is_containment=False,
                 dist=None, rank=0, world=1, row_chunk_bytes=2 << 30):
        self.ctx, self.k, self.s, self.threshold = ctx, k, sketch_size, threshold
        self.is_containment = is_containment
        self.dist, self.rank, self.world = dist, rank, world
        self.row_chunk_bytes = row_chunk_bytes
        self.last_sketches = None
        self.last_mst = None
        self._edge_cap = 1 << 20
        self._edges = None

    # ---- pieces ---------------------------------------------------------------------------------
    def gather_sketches(self, sk):
        """All ranks end up with every genome's sketch (strided layout, stride = sketch_size)."""
        if self.world == 1:
            return sk
        n_local = sk.n
        stride = sk.hashes.numel() // max(n_local, 1)
        hashes = torch.empty(self.world * sk.hashes.numel(), dtype=sk.hashes.dtype, device=sk.hashes.device)
        lens = torch.empty(self.world * n_local, dtype=sk.len.dtype, device=sk.len.device)
        self.dist.all_gather_into_tensor(hashes, sk.hashes.contiguous())
        self.dist.all_gather_into_tensor(lens, sk.len.contiguous())
        n = self.world * n_local
        start = torch.arange(n, dtype=torch.int64, device=sk.hashes.device) * stride
        return SketchSet(hashes, start, lens, sk.width, sk.k, sk.kind)

    @staticmethod
    def split_point(n_local, slots=0):
        """Genomes sketched before the first all-gather is started: ~3/4 of them, so that the gather
        of the first part hides behind the sketching of the rest and only a quarter's gather is
        exposed.  With `slots` (workgroups the GPU runs at once, one genome each) the first part is a
        whole number of full rounds, so the extra launch boundary costs no idle tail."""
        if n_local < 8:
            return n_local
        if slots > 0 and n_local >= 2 * slots:
            return max(slots, int(0.8 * n_local) // slots * slots)
        return (3 * n_local) // 4

    def _gather_part(self, g_hashes, g_len, base, out_part, cnt_part):
        """Start the all-gathers of one locally sketched row range into rows [base, base + W*m) of the
        global buffers (rank r's rows land at base + r*m).  Returns the async work handles."""
        m = out_part.shape[0]
        rows = slice(base, base + self.world * m)
        return [self.dist.all_gather_into_tensor(g_hashes[rows].view(-1), out_part.reshape(-1), async_op=True),
                self.dist.all_gather_into_tensor(g_len[rows], cnt_part.contiguous(), async_op=True)]

    def gather_parts(self, out, cnt, parts, k, kind="minhash", before_part=None):
        """All-gather the local row ranges `parts` = [(a, b), ...] one after the other into the global
        layout [part 0 of rank 0..W-1 | part 1 of rank 0..W-1 | ...]; `before_part(a, b)` (if given)
        is called right before a part's collectives are started -- the multi-GPU step sketches the
        part there, so the previous part's all-gather runs beside it.  Every rank passes the same
        parts.  Returns (SketchSet over the global buffers, work handles to wait on)."""
        n_local, stride = out.shape
        n = self.world * n_local
        g_hashes = torch.empty((n, stride), dtype=out.dtype, device=out.device)
        g_len = torch.empty(n, dtype=cnt.dtype, device=cnt.device)
        works, base = [], 0
        for a, b in parts:
            if b <= a:
                continue
            if before_part is not None:
                before_part(a, b)
            works += self._gather_part(g_hashes, g_len, base, out[a:b], cnt[a:b])
            base += self.world * (b - a)
        start = torch.arange(n, dtype=torch.int64, device=out.device) * stride
        return SketchSet(g_hashes.view(-1), start, g_len, 8, k, kind), works

    def sketch_and_gather(self, seq, off, sizes=None):
        """Multi-GPU sketch phase: sketch part A, start its all-gather, sketch part B, start its
        all-gather.  The collective of part A runs on RCCL's stream beside the sketch kernel of B."""
        ctx = self.ctx
        off = np.ascontiguousarray(off, dtype=np.uint64)
        n_local = len(off) - 1
        stride = int(np.max(sizes)) if sizes is not None else int(self.s)
        out = torch.empty((n_local, max(stride, 1)), dtype=torch.int64, device=ctx.device)
        cnt = torch.zeros(n_local, dtype=torch.int32, device=ctx.device)
        split = self.split_point(n_local, slots=3 * ctx.num_cu())

        def sketch_part(a, b):
            ctx.sketch_minhash_into(seq, off[a:b + 1], out[a:b], cnt[a:b], k=self.k, size=self.s,
                                    sizes=None if sizes is None else sizes[a:b])

        return self.gather_parts(out, cnt, [(0, split), (split, n_local)], self.k, before_part=sketch_part)

    def candidate_edges(self, sk, row0, row1):
        """Dense common counts for rows [row0,row1) x cols [0,row) in chunks, filtered into a
        compact (i, j, common) list on the device."""
        ctx = self.ctx
        n = sk.n
        radio = mst_radio(self.threshold, sk.k)
        count = torch.zeros(1, dtype=torch.int64, device=ctx.device)
        if self._edges is None or self._edges.shape[0] < self._edge_cap:
            self._edges = torch.empty((self._edge_cap, 3), dtype=torch.int32, device=ctx.device)
        rows_per = max(64, min(max(row1 - row0, 1), self.row_chunk_bytes // (max(n, 1) * 4)))
        rows_per = ((rows_per + 63) // 64) * 64  # whole 64-row blocks, one launch when it fits
        m = 0
        r0 = max(row0, 1)
        common = None
        while r0 < row1:
            r1 = min(row1, r0 + rows_per)
            c1 = r1 - 1
            if common is None or common.shape[0] < (r1 - r0) or common.shape[1] < c1:
                common = torch.empty((rows_per, max(n, 1)), dtype=torch.int32, device=ctx.device)
            ctx.check(ctx.lib.rtc_pair_common_dev(ctx.h, _t_ptr(sk.hashes), sk.width, _t_ptr(sk.start),
                                                  _t_ptr(sk.len), n, r0, r1, 0, c1, _t_ptr(common),
                                                  common.stride(0), 1, 0))
            while True:
                ctx.check(ctx.lib.rtc_extract_edges_dev(ctx.h, _t_ptr(common), common.stride(0), r0, r1, 0, c1,
                                                        _t_ptr(sk.len), radio, _t_ptr(self._edges),
                                                        self._edges.shape[0], _t_ptr(count)))
                cnt = int(count.item())
                if cnt <= self._edges.shape[0]:
                    m = cnt
                    break
                self._edge_cap = max(cnt + cnt // 2, self._edge_cap * 2)
                bigger = torch.empty((self._edge_cap, 3), dtype=torch.int32, device=ctx.device)
                bigger[:m] = self._edges[:m]
                self._edges = bigger
                count.fill_(m)
            r0 = r1
        return self._edges, m

    def boruvka(self, sk, edges, m):
        backend = HipBoruvkaBackend(self.ctx, sk, edges, m, self.is_containment)
        return boruvka_rounds(backend, sk.n, self.ctx.lib, self.dist, self.world)

    def finish(self, sk, sel):
        """(i, j, common) -> EdgeInfo records with the reference's double arithmetic, sorted."""
        lens_h = sk.len.cpu().numpy().view(np.uint32)
        out = np.zeros(max(len(sel), 1), dtype=EDGE_DT)
        sel = np.ascontiguousarray(sel)
        st = self.ctx.lib.rtc_edges_to_mst_host(_np_ptr(sel), len(sel), _np_ptr(lens_h), sk.k,
                                                int(self.is_containment), _np_ptr(out))
        if st != _lib.RTC_OK:
            raise _lib.RtcError(st, "rtc_edges_to_mst_host")
        return out[: len(sel)]

    # ---- one step ---------------------------------------------------------------------------------
    def step(self, seq, off, sizes=None):
        ctx = self.ctx
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        ev[0].record()
        if self.dist is None:
            sk = ctx.sketch_minhash(seq, off, k=self.k, size=self.s, sizes=sizes)
            ev[1].record()
        else:
            sk, works = self.sketch_and_gather(seq, off, sizes)
            ev[1].record()  # local sketching done; what follows is the exposed rest of the all-gathers
            for w in works:
                w.wait()
        ev[2].record()
        b = triangle_row_ranges(sk.n, self.world, fixed_cols=8.8 * self.s if self.world > 1 else 0.0)
        row0, row1 = b[self.rank], b[self.rank + 1]
        edges, m = self.candidate_edges(sk, row0, row1)
        ev[3].record()
        sel, rounds = self.boruvka(sk, edges, m)
        mst = self.finish(sk, sel)
        ev[4].record()
        torch.cuda.synchronize()
        self.last_sketches, self.last_mst = sk, mst
        pairs_local = (row1 * (row1 - 1) - row0 * (row0 - 1)) // 2 if row1 > 0 else 0
        return {
            "sketch_ms": ev[0].elapsed_time(ev[1]),
            "gather_ms": ev[1].elapsed_time(ev[2]),
            "pair_ms": ev[2].elapsed_time(ev[3]),
            "mst_ms": ev[3].elapsed_time(ev[4]),
            "dist_ms": ev[2].elapsed_time(ev[4]),
            "pairs_local": float(pairs_local),
            "cand_edges": float(m),
            "boruvka_rounds": float(rounds),
            "mst_edges": float(len(mst)),
        }

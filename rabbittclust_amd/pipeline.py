"""clust-mst hot path as one step: sketch -> [all-gather] -> row-sharded all-pairs -> edges -> MSF.

Mirrors clust_from_genomes -> compute_sketches -> compute_clusters of the reference
(src/sub_command.cpp:2302-2315, :2858-2889, :2924-3053; --fast: :1934-1951, :1953-2152) from the
sketching call down to the `vector<EdgeInfo> mst`.  Multi-GPU: one process per GPU; sketches are
all-gathered once (RCCL) into the canonical order rank*n_local + i, the strict lower triangle of the
N x N pair space is split into contiguous row ranges of equal cost, and every Boruvka round
all-reduces (MIN) ONE u64 key array when all sketches have the same size (three small arrays
otherwise); the union step runs on the device, identically on every rank.
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
    row's length (building the LDS tables of its 64-row block costs as much as ~1.84 columns per
    sketch hash on MI355X: tools/sim_rank.py).  With fixed_cols = 0 the split is by pair count,
    boundaries n*sqrt(r/world)."""
    c = float(fixed_cols)
    area = n * n / 2.0 + c * n
    b = [int(round(-c + math.sqrt(c * c + 2.0 * area * r / world))) for r in range(world + 1)]
    b[0], b[-1] = 0, n
    for r in range(1, world + 1):
        b[r] = min(max(b[r], b[r - 1]), n)
    return b


class TorchComm:
    """The collectives of the multi-GPU step over torch.distributed (backend "nccl" = RCCL over xGMI
    on the GPUs, "gloo" in the CPU tests).  world == 1 without a process group: no-ops."""

    def __init__(self, dist=None, rank=0, world=1):
        self.dist, self.rank, self.world = dist, rank, world

    @property
    def active(self):
        return self.dist is not None

    def all_reduce_min(self, t):
        if self.dist is not None:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)

    def all_reduce_max(self, t):
        if self.dist is not None:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)

    def all_gather(self, out, inp, async_op=False):
        return self.dist.all_gather_into_tensor(out, inp, async_op=async_op)


class NativeComm:
    """The same collectives through the C ABI's own RCCL communicator (rtc_comm_*): what the C++ hosts
    use.  With it MstPipeline.step runs the multi-GPU phases as two C calls
    (rtc_sketch_minhash_sharded, rtc_mst_sharded)."""

    def __init__(self, comm):
        self.c, self.rank, self.world, self.dist = comm, comm.rank, comm.size, None

    active = True

    def all_reduce_min(self, t):
        self.c.all_reduce(t, "min")

    def all_reduce_max(self, t):
        self.c.all_reduce(t, "max")


class HipBoruvkaBackend:
    """Per-round primitives on this rank's candidate edges (device tensors, HIP kernels)."""

    def __init__(self, ctx, sk, edges, m, is_containment):
        self.ctx, self.sk, self.edges, self.m, self.ic = ctx, sk, edges, m, int(is_containment)
        self.device = ctx.device
        n = sk.n
        self.comp = torch.empty(max(n, 1), dtype=torch.int32, device=self.device)
        self.succ = torch.empty(max(n, 1), dtype=torch.int32, device=self.device)
        self.sel = torch.empty((max(n, 1), 3), dtype=torch.int32, device=self.device)
        self.nsel = torch.zeros(2, dtype=torch.int64, device=self.device)

    def init(self):
        c = self.ctx
        c.check(c.lib.rtc_boruvka_init_dev(c.h, self.sk.n, _t_ptr(self.comp), _t_ptr(self.nsel)))

    def minkey(self, s_fixed, key):
        c = self.ctx
        c.check(c.lib.rtc_boruvka_minkey_dev(c.h, _t_ptr(self.edges), self.m, _t_ptr(self.comp), self.sk.n,
                                             int(s_fixed), _t_ptr(key)))

    def minweight(self, wkey):
        c = self.ctx
        c.check(c.lib.rtc_boruvka_minweight_dev(c.h, _t_ptr(self.edges), self.m, _t_ptr(self.sk.len), self.ic,
                                                _t_ptr(self.comp), self.sk.n, _t_ptr(wkey)))

    def minedge(self, wkey, ekey):
        c = self.ctx
        c.check(c.lib.rtc_boruvka_minedge_dev(c.h, _t_ptr(self.edges), self.m, _t_ptr(self.sk.len), self.ic,
                                              _t_ptr(self.comp), self.sk.n, _t_ptr(wkey), _t_ptr(ekey)))

    def fetch(self, ekey, ecommon):
        c = self.ctx
        c.check(c.lib.rtc_boruvka_fetch_dev(c.h, _t_ptr(self.edges), self.m, _t_ptr(self.comp), self.sk.n,
                                            _t_ptr(ekey), _t_ptr(ecommon)))

    def union(self, s_fixed, key, ecommon):
        """Device union; returns the number of forest edges this round added (identical on every rank)."""
        c = self.ctx
        added = C.c_uint32(0)
        c.check(c.lib.rtc_boruvka_union_dev(c.h, self.sk.n, int(s_fixed), _t_ptr(key),
                                            _t_ptr(ecommon) if ecommon is not None else None, _t_ptr(self.comp),
                                            _t_ptr(self.succ), _t_ptr(self.sel), _t_ptr(self.nsel), C.byref(added)))
        return int(added.value)

    def selected(self):
        ns = int(self.nsel[0].item())
        return np.ascontiguousarray(self.sel[:ns].cpu().numpy().view(np.uint32)).view(CEDGE_DT).reshape(-1)


def boruvka_rounds(backend, n, comm=None, s_fixed=0):
    """Boruvka over row-sharded candidate edges.  Fixed-size mode (s_fixed = the common sketch size,
    fused key fits): per round ONE local pass -> ONE all-reduce(MIN) -> device union.  Otherwise:
    minimum weight key -> all-reduce(MIN) -> minimum edge id among edges attaining it ->
    all-reduce(MIN) -> owner publishes `common` -> all-reduce(MAX) -> device union.  Every rank
    applies the identical union to the identical reduced arrays.  `backend` supplies the primitives."""
    comm = comm or TorchComm()
    dev = backend.device
    wkey = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
    ekey = ecommon = None
    if not s_fixed:
        ekey = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
        ecommon = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    backend.init()
    rounds = 0
    for _ in range(64):
        if s_fixed:
            backend.minkey(s_fixed, wkey)
            comm.all_reduce_min(wkey)
            added = backend.union(s_fixed, wkey, None)
        else:
            backend.minweight(wkey)
            comm.all_reduce_min(wkey)
            backend.minedge(wkey, ekey)
            comm.all_reduce_min(ekey)
            backend.fetch(ekey, ecommon)
            comm.all_reduce_max(ecommon)
            added = backend.union(0, ekey, ecommon)
        rounds += 1
        if added == 0:
            break
    return backend.selected(), rounds


def _gather_stride(sk):
    """Row length at which a rank's KSSD sketches travel: its longest sketch, rounded up to 4 tuples (at least 4)."""
    longest = int(sk.len.max().item()) if sk.n else 0
    return max(4, (longest + 3) // 4 * 4)


class MstPipeline:
    """mode "minhash": rtc_sketch_minhash_dev; mode "kssd": rtc_sketch_kssd_dev (--fast, u32/u64 tuples); handed an
    api.PackedBatch instead of characters, the step sketches it as it is (rtc_sketch_minhash_packed_dev /
    rtc_sketch_kssd_packed_dev)."""

    def __init__(self, ctx, k=21, sketch_size=1000, threshold=0.05, is_containment=False,
                 dist=None, rank=0, world=1, mode="minhash", drlevel=3, shuffled_dim=None, comm=None):
        self.ctx, self.k, self.s, self.threshold = ctx, k, sketch_size, threshold
        self.is_containment = is_containment
        self.comm = comm or TorchComm(dist, rank, world)
        self.dist, self.rank, self.world = self.comm.dist, self.comm.rank, self.comm.world
        self.mode, self.drlevel, self.shuffled_dim = mode, drlevel, shuffled_dim
        self.last_sketches = None
        self.last_mst = None
        self._edge_cap = 1 << 20
        self._edges = None
        self._sel = None

    # ---- pieces ---------------------------------------------------------------------------------
    def _check_equal_counts(self, n_local):
        """Every rank must bring the same number of genomes (the gathered order is rank*n_local + i)."""
        t = torch.tensor([n_local, -n_local], dtype=torch.int64, device=self.ctx.device if self.ctx else "cpu")
        self.comm.all_reduce_max(t)
        if int(t[0].item()) != n_local or int(-t[1].item()) != n_local:
            raise ValueError("multi-GPU step: ranks hold different genome counts "
                             f"(this rank {n_local}, max {int(t[0].item())}, min {int(-t[1].item())})")

    def gather_sketches(self, sk):
        """All ranks end up with every genome's sketch in canonical order (genome g of rank r at
        r*n_local + g; strided layout)."""
        if not self.comm.active:
            return sk
        n_local = sk.n
        stride = sk.hashes.numel() // max(n_local, 1)
        hashes = torch.empty(self.world * sk.hashes.numel(), dtype=sk.hashes.dtype, device=sk.hashes.device)
        lens = torch.empty(self.world * n_local, dtype=sk.len.dtype, device=sk.len.device)
        self.comm.all_gather(hashes, sk.hashes.contiguous())
        self.comm.all_gather(lens, sk.len.contiguous())
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

    def gather_parts(self, out, cnt, parts, k, kind="minhash", before_part=None, width=8):
        """All-gather the local row ranges `parts` = [(a, b), ...] one after the other; `before_part(a, b)`
        (if given) is called right before a part's collectives are started -- the multi-GPU step
        sketches the part there, so the previous part's all-gather runs beside it.  Every rank passes
        the same parts.  Returns (finish, works): wait on `works`, then finish() assembles the
        canonical global order (genome g of rank r at r*n_local + g) and returns the SketchSet."""
        n_local, stride = out.shape
        W = self.world
        n = W * n_local
        staged, works = [], []
        for a, b in parts:
            if b <= a:
                continue
            if before_part is not None:
                before_part(a, b)
            m = b - a
            th = torch.empty((W, m, stride), dtype=out.dtype, device=out.device)
            tl = torch.empty((W, m), dtype=cnt.dtype, device=cnt.device)
            works.append(self.comm.all_gather(th.view(-1), out[a:b].reshape(-1), async_op=True))
            works.append(self.comm.all_gather(tl.view(-1), cnt[a:b].contiguous(), async_op=True))
            staged.append((a, b, th, tl))

        def finish():
            if len(staged) == 1 and staged[0][0] == 0 and staged[0][1] == n_local:
                g_hashes, g_len = staged[0][2].view(n, stride), staged[0][3].view(n)
            else:
                g_hashes = torch.empty((W, n_local, stride), dtype=out.dtype, device=out.device)
                g_len = torch.empty((W, n_local), dtype=cnt.dtype, device=cnt.device)
                for a, b, th, tl in staged:
                    g_hashes[:, a:b] = th
                    g_len[:, a:b] = tl
                g_hashes, g_len = g_hashes.view(n, stride), g_len.view(n)
            start = torch.arange(n, dtype=torch.int64, device=out.device) * stride
            return SketchSet(g_hashes.view(-1), start, g_len, width, k, kind)

        return finish, works

    def sketch_and_gather(self, seq, off, sizes=None):
        """Multi-GPU MinHash sketch phase: sketch part A, start its all-gather, sketch part B, start its
        all-gather.  The collective of part A runs on RCCL's stream beside the sketch kernel of B."""
        ctx = self.ctx
        off = np.ascontiguousarray(off, dtype=np.uint64)
        n_local = len(off) - 1
        self._check_equal_counts(n_local)
        stride = int(np.max(sizes)) if sizes is not None else int(self.s)
        if sizes is not None:  # containment mode: per-rank sizes differ -> agree on the row stride
            t = torch.tensor([stride], dtype=torch.int64, device=ctx.device)
            self.comm.all_reduce_max(t)
            stride = int(t.item())
        out = torch.empty((n_local, max(stride, 1)), dtype=torch.int64, device=ctx.device)
        cnt = torch.zeros(n_local, dtype=torch.int32, device=ctx.device)
        split = self.split_point(n_local, slots=3 * ctx.num_cu())

        def sketch_part(a, b):
            from .api import PackedBatch
            if isinstance(seq, PackedBatch):
                ctx.sketch_minhash_packed(seq, off[a:b + 1], k=self.k, size=self.s, sizes=None if sizes is None else sizes[a:b],
                                          out=out[a:b], cnt=cnt[a:b])
            else:
                ctx.sketch_minhash_into(seq, off[a:b + 1], out[a:b], cnt[a:b], k=self.k, size=self.s,
                                        sizes=None if sizes is None else sizes[a:b])

        return self.gather_parts(out, cnt, [(0, split), (split, n_local)], self.k, before_part=sketch_part)

    def _sketch_kssd(self, seq, off):
        """characters (rtc_sketch_kssd_dev) or a batch in the 2-bit staging format (rtc_sketch_kssd_packed_dev)"""
        from .api import PackedBatch
        if isinstance(seq, PackedBatch):
            return self.ctx.sketch_kssd_packed(seq, None, None, off, self.shuffled_dim, kmer_size=self.k, drlevel=self.drlevel)
        return self.ctx.sketch_kssd(seq, off, self.shuffled_dim, kmer_size=self.k, drlevel=self.drlevel)

    def sketch_kssd_and_gather(self, seq, off):
        """Multi-GPU KSSD (--fast) sketch phase (sketchFileWithKssd on every rank's genomes, then one
        all-gather): sketch sizes vary per genome, so the ranks first agree on the row stride."""
        ctx = self.ctx
        sk = self._sketch_kssd(seq, off)
        if not self.comm.active:
            return (lambda: sk), []
        n_local = sk.n
        self._check_equal_counts(n_local)
        stride = sk.hashes.numel() // max(n_local, 1)
        # rows travel at the longest sketch of any rank (a multiple of 4 tuples), not at the allocated stride: the
        # allocation is 1.5x the expected count + 256, i.e. twice what a 2 Mbp genome yields
        t = torch.tensor([_gather_stride(sk)], dtype=torch.int64, device=ctx.device)
        self.comm.all_reduce_max(t)
        gstride = int(t.item())
        rows = sk.hashes.view(n_local, stride)
        if gstride != stride:
            wide = torch.zeros((n_local, gstride), dtype=rows.dtype, device=rows.device)
            m = min(stride, gstride)
            wide[:, :m] = rows[:, :m]
            rows = wide
        return self.gather_parts(rows, sk.len.contiguous(), [(0, n_local)], sk.k, kind="kssd", width=sk.width)

    def candidate_edges(self, sk, row0, row1):
        """Candidate (i, j, common) triples of rows [row0,row1) x cols [0,row): emitted by the pair
        kernel itself (reference filters src/MST.cpp:1468-1487), compacted on the device."""
        ctx = self.ctx
        n = sk.n
        radio = mst_radio(self.threshold, sk.k)
        count = torch.zeros(1, dtype=torch.int64, device=ctx.device)
        r0 = max(row0, 1)
        while True:
            if self._edges is None or self._edges.shape[0] < self._edge_cap:
                self._edges = torch.empty((self._edge_cap, 3), dtype=torch.int32, device=ctx.device)
            if r0 >= row1:
                return self._edges, 0
            count.zero_()
            ctx.check(ctx.lib.rtc_pair_edges_dev(ctx.h, _t_ptr(sk.hashes), sk.width, _t_ptr(sk.start), _t_ptr(sk.len),
                                                 n, r0, row1, 0, row1 - 1, radio, _t_ptr(self._edges),
                                                 self._edges.shape[0], _t_ptr(count)))
            cnt = int(count.item())
            if cnt <= self._edges.shape[0]:
                return self._edges, cnt
            self._edge_cap = cnt + cnt // 8  # the kernel counted everything: exact need, one redo
            self._edges = None

    def fixed_size(self, sk):
        """Common sketch size when every sketch has it and the fused Boruvka key fits, else 0."""
        if sk.n < 2:
            return 0
        mm = torch.stack((sk.len.min(), sk.len.max())).cpu().numpy()
        if int(mm[0]) != int(mm[1]) or int(mm[0]) <= 0:
            return 0
        return int(mm[0]) if self.ctx.lib.rtc_boruvka_key_bits(sk.n, int(mm[0])) else 0

    def boruvka(self, sk, edges, m):
        if not self.comm.active:  # one GPU: all rounds behind one C call (rtc_msf_dev)
            ctx = self.ctx
            if self._sel is None or self._sel.shape[0] < max(sk.n, 1):
                self._sel = torch.empty((max(sk.n, 1), 3), dtype=torch.int32, device=ctx.device)
            nsel, rounds = C.c_uint64(), C.c_int()
            ctx.check(ctx.lib.rtc_msf_dev(ctx.h, _t_ptr(edges), m, _t_ptr(sk.len), sk.n, int(self.is_containment), _t_ptr(self._sel),
                                          C.byref(nsel), C.byref(rounds)))
            sel = np.ascontiguousarray(self._sel[:nsel.value].cpu().numpy().view(np.uint32)).view(CEDGE_DT).reshape(-1)
            return sel, int(rounds.value)
        backend = HipBoruvkaBackend(self.ctx, sk, edges, m, self.is_containment)
        return boruvka_rounds(backend, sk.n, self.comm, self.fixed_size(sk))

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

    def _kssd_sketch_and_gather_native(self, seq, off, ev_sketched):
        """--fast sketch phase over the C ABI's communicator.  KSSD sketches vary in length, so a part's rows travel at the
        longest sketch any rank produced for it (agreed with one host all-reduce; the allocation is twice that).  The genomes
        are sketched in two parts like the MinHash step: the first part's rows travel on the communicator's side stream
        while the second part is sketched, and the gathered set keeps the two parts as two blocks of rows with a stride
        each (`start` carries the layout; no copy into a common stride)."""
        ctx, comm = self.ctx, self.comm.c
        W, n_local = comm.size, len(off) - 1
        cut = self.split_point(n_local) if W > 1 else n_local
        parts = [(0, cut), (cut, n_local)] if 0 < cut < n_local else [(0, n_local)]
        off = np.ascontiguousarray(off, dtype=np.uint64)
        blocks, width, kk, dt = [], 4, 0, torch.int32
        for i, (a, b) in enumerate(parts):
            loc = self._sketch_kssd(seq, off[a:b + 1])
            if i == len(parts) - 1:
                ev_sketched.record()
            m = b - a
            stride = loc.hashes.numel() // max(m, 1)
            v = comm.all_reduce_host([_gather_stride(loc), n_local, -n_local, loc.width, -loc.width], "max")
            if int(v[1]) != n_local or int(-v[2]) != n_local:
                raise ValueError("multi-GPU step: ranks hold different genome counts")
            if int(v[3]) != loc.width or int(-v[4]) != loc.width:
                raise ValueError("multi-GPU step: ranks disagree on the tuple width")
            g = int(v[0])
            g_h = torch.empty((W * m, g), dtype=loc.hashes.dtype, device=ctx.device)
            g_l = torch.empty(W * m, dtype=torch.int32, device=ctx.device)
            w = min(stride, g)
            g_h[comm.rank * m:(comm.rank + 1) * m, :w] = loc.hashes.view(m, stride)[:, :w]
            g_l[comm.rank * m:(comm.rank + 1) * m] = loc.len
            last = i == len(parts) - 1
            comm.gather_rows(g_h, m, 0, m, async_=not last)   # the side stream starts behind what the context stream holds so far
            comm.gather_rows(g_l, m, 0, m, async_=not last)
            blocks.append((m, g, g_h, g_l))
            width, kk, dt = loc.width, loc.k, loc.hashes.dtype
        if len(parts) > 1:
            comm.wait()
        if len(blocks) == 1:
            m, g, g_h, g_l = blocks[0]
            start = torch.arange(W * m, dtype=torch.int64, device=ctx.device) * g
            return SketchSet(g_h.view(-1), start, g_l, width, kk, "kssd")
        # canonical order: genome q of rank r at r * n_local + q; block i holds rank r's part i at rows [r * m_i, (r + 1) * m_i)
        hashes = torch.cat([blk[2].view(-1) for blk in blocks])
        starts, lens, base = [], [], 0
        for m, g, g_h, g_l in blocks:
            starts.append((torch.arange(W * m, dtype=torch.int64, device=ctx.device) * g + base).view(W, m))
            lens.append(g_l.view(W, m))
            base += W * m * g
        return SketchSet(hashes, torch.cat(starts, dim=1).reshape(-1).contiguous(), torch.cat(lens, dim=1).reshape(-1).contiguous(),
                         width, kk, "kssd")

    # ---- one step ---------------------------------------------------------------------------------
    def step_native(self, seq, off, sizes=None):
        """The step with both multi-GPU phases behind the C ABI (NativeComm)."""
        ctx, comm = self.ctx, self.comm.c
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev[0].record()
        if isinstance(seq, list):  # the rank's genomes as batches in the 2-bit staging format: [(PackedBatch, off), ...]
            rows = getattr(self, "_rows", (None, None))
            sk = ctx.sketch_packed_sharded(comm, seq, mode=self.mode, k=self.k, size=self.s, drlevel=self.drlevel,
                                           shuffled_dim=self.shuffled_dim, out=rows[0], cnt=rows[1])
            self._rows = (sk.hashes.view(sk.n, -1), sk.len)  # the global rows are reused by the next step
            ev[1].record()
        elif self.mode == "kssd":
            sk = self._kssd_sketch_and_gather_native(seq, off, ev[1])
        else:
            sk = ctx.sketch_minhash_sharded(comm, seq, off, k=self.k, size=self.s, sizes=sizes)
            ev[1].record()
        ev[2].record()
        mst, st = ctx.mst_sharded(comm, sk, self.threshold, self.is_containment)
        ev[3].record()
        torch.cuda.synchronize()
        self.last_sketches, self.last_mst = sk, mst
        row0, row1 = int(st.row0), int(st.row1)
        pairs_local = (row1 * (row1 - 1) - row0 * (row0 - 1)) // 2 if row1 > 0 else 0
        return {
            "sketch_ms": ev[0].elapsed_time(ev[1]),
            "gather_ms": ev[1].elapsed_time(ev[2]),
            "pair_ms": float(st.pair_ms),
            "mst_ms": ev[2].elapsed_time(ev[3]) - float(st.pair_ms),  # Boruvka + forest read-back + host distances: the rest of the call
            "boruvka_ms": float(st.mst_ms),                            # its device rounds alone (HIP events inside rtc_mst_sharded)
            "dist_ms": ev[2].elapsed_time(ev[3]),
            "pairs_local": float(pairs_local),
            "cand_edges": float(st.cand_edges),
            "pair_path": float(ctx.pair_last_path()),
            "boruvka_rounds": float(st.rounds),
            "mst_edges": float(len(mst)),
        }

    def step(self, seq, off=None, sizes=None):
        from .api import PackedBatch
        if isinstance(seq, list) and not isinstance(self.comm, NativeComm):
            raise TypeError("a list of packed batches goes through the C ABI's sharded entry points: give the pipeline a NativeComm")
        if isinstance(self.comm, NativeComm):
            return self.step_native(seq, off, sizes)
        ctx = self.ctx
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        ev[0].record()
        if self.mode == "kssd":
            finish, works = self.sketch_kssd_and_gather(seq, off)
        elif not self.comm.active:
            if isinstance(seq, PackedBatch):
                sk0 = ctx.sketch_minhash_packed(seq, off, k=self.k, size=self.s, sizes=sizes)
            else:
                sk0 = ctx.sketch_minhash(seq, off, k=self.k, size=self.s, sizes=sizes)
            finish, works = (lambda: sk0), []
        else:
            finish, works = self.sketch_and_gather(seq, off, sizes)
        ev[1].record()  # local sketching done; what follows is the exposed rest of the all-gathers
        for w in works:
            w.wait()
        sk = finish()
        ev[2].record()
        fixed_cols = 1.84 * float(sk.len.float().mean().item()) if self.world > 1 else 0.0
        b = triangle_row_ranges(sk.n, self.world, fixed_cols=fixed_cols)
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
            "pair_path": float(ctx.pair_last_path()),
            "boruvka_rounds": float(rounds),
            "mst_edges": float(len(mst)),
        }

"""Host-side Python mirror of the sketch + all-pairs path, over the C ABI (include/rtclust.h).

PyTorch is plumbing here: device buffers (torch tensors), the current HIP stream and
torch.distributed.  All compute goes through librtclust_hip.so; nothing in this module has a
CPU fallback.  Names follow the reference's vocabulary (genomes, sketches, hashes, MST edges).
"""
import ctypes as C
import os
import math

import numpy as np
import torch

from . import _lib
from ._lib import RtcError, SynthDesc

SYNTH_DT = np.dtype([("fam_seed", "<u8"), ("mut_seed", "<u8"), ("mut_thr", "<u4"), ("n_every", "<u4")])
CEDGE_DT = np.dtype([("i", "<u4"), ("j", "<u4"), ("common", "<u4")])
EDGE_DT = np.dtype([("preNode", "<i4"), ("sufNode", "<i4"), ("dist", "<f8")])  # edge.mst record


def _np_ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _t_ptr(t):
    return C.c_void_p(t.data_ptr())


class SketchSet:
    """Device-resident sketches: `hashes` flat tensor (int64 view of u64, or int32 view of u32),
    `start` (int64, element offsets) and `len` (int32) per genome -- the CSR the pair kernels read."""

    def __init__(self, hashes, start, length, width, k, kind):
        self.hashes, self.start, self.len, self.width, self.k, self.kind = hashes, start, length, width, k, kind

    @property
    def n(self):
        return int(self.len.numel())

    def to_host(self):
        np_dt = np.uint64 if self.width == 8 else np.uint32
        flat = self.hashes.cpu().numpy().view(np_dt).reshape(-1)
        start = self.start.cpu().numpy().astype(np.uint64)
        ln = self.len.cpu().numpy().astype(np.uint32)
        return [flat[int(s):int(s) + int(l)].copy() for s, l in zip(start, ln)]

    @staticmethod
    def from_host(sketches, device, k=21, kind="minhash", width=8):
        np_dt = np.uint64 if width == 8 else np.uint32
        t_dt = torch.int64 if width == 8 else torch.int32
        lens = np.array([len(s) for s in sketches], dtype=np.int32)
        start = np.zeros(len(sketches), dtype=np.int64)
        if len(sketches) > 1:
            start[1:] = np.cumsum(lens[:-1], dtype=np.int64)
        flat = (np.concatenate([np.asarray(s, dtype=np_dt) for s in sketches])
                if len(sketches) and lens.sum() else np.zeros(0, dtype=np_dt))
        flat = np.ascontiguousarray(flat)
        h = torch.from_numpy(flat.view(np.int64 if width == 8 else np.int32).copy()).to(device)
        if h.numel() == 0:
            h = torch.zeros(2, dtype=t_dt, device=device)
        return SketchSet(h, torch.from_numpy(start).to(device), torch.from_numpy(lens).to(device), width, k, kind)


_LIVE = None  # weak set of the live contexts (reload_all_options)


def reload_all_options():
    """every live Context reads the RTC_* switches again (the library reads them once, at rtc_ctx_create): tests that flip one"""
    for c in list(_LIVE or ()):
        if c.h:
            c.reload_options()


class Context:
    """One context per GPU (one process per GPU in multi-GPU runs)."""

    def __init__(self, device=0):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RtcError(_lib.RTC_ERR_HIP, "no HIP device visible: the MI355X path has no CPU fallback")
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        torch.cuda.init()
        torch.empty(1, device=self.device)  # force the HIP runtime torch bundles to initialise first
        h = C.c_void_p()
        st = self.lib.rtc_ctx_create(device, C.byref(h))
        if st != _lib.RTC_OK:
            raise RtcError(st, "rtc_ctx_create: " + self.lib.rtc_last_error(None).decode(errors="replace"))
        self.h = h
        global _LIVE
        if _LIVE is None:
            import weakref
            _LIVE = weakref.WeakSet()
        _LIVE.add(self)
        self.use_torch_stream()

    def close(self):
        if getattr(self, "h", None):
            self.lib.rtc_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, st):
        if st != _lib.RTC_OK:
            raise RtcError(st, self.lib.rtc_last_error(self.h).decode(errors="replace"))

    def reload_options(self):
        """the library's RTC_* switches are read when the context is created: read them again (tests that flip one)"""
        self.check(self.lib.rtc_ctx_reload_options(self.h))

    def env(self, **switches):
        """context manager: RTC_* switches set in the environment (None: unset) and read by this context, restored on exit"""
        import contextlib

        @contextlib.contextmanager
        def scope():
            old = {k: os.environ.get(k) for k in switches}
            try:
                for k, v in switches.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = str(v)
                self.reload_options()
                yield self
            finally:
                for k, v in old.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
                self.reload_options()
        return scope()

    def num_cu(self):
        info = (C.c_int * 3)()
        self.check(self.lib.rtc_device_info(self.h, info))
        return int(info[0])

    def use_torch_stream(self):
        s = torch.cuda.current_stream(self.device).cuda_stream
        self.check(self.lib.rtc_ctx_set_stream(self.h, C.c_void_p(s)))

    def sync(self):
        self.check(self.lib.rtc_ctx_sync(self.h))

    def pair_last_path(self):
        """Path of the last pair_edges call: 0 none, 1 merge kernel, 2 tiled kernel, 3 inverted join."""
        return int(self.lib.rtc_pair_last_path(self.h))

    def diag(self):
        """rtc_diag_counters as a dict: which paths this context has taken since it was created"""
        a = (C.c_uint64 * 8)()
        self.check(self.lib.rtc_diag_counters(self.h, a))
        names = ("join_tiles", "tiled_tiles", "merge_tiles", "contractions", "greedy_global", "greedy_blocks", "estimates")
        return {k: int(a[i]) for i, k in enumerate(names)}

    def pair_last_kernel_ms(self):
        """Duration of the last tiled pair kernel launch (HIP events on its launch stream)."""
        ms = C.c_float()
        self.check(self.lib.rtc_pair_last_kernel_ms(self.h, C.byref(ms)))
        return float(ms.value)

    def timer_start(self):
        self.check(self.lib.rtc_timer_start(self.h))

    def timer_stop(self):
        ms = C.c_float()
        self.check(self.lib.rtc_timer_stop(self.h, C.byref(ms)))
        return float(ms.value)

    # ---- inputs -------------------------------------------------------------------------------
    def synth_genomes(self, desc, off):
        """desc: numpy SYNTH_DT[n]; off: u64[n+1].  Returns uint8 tensor with all genomes."""
        desc = np.ascontiguousarray(desc, dtype=SYNTH_DT)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        n = len(desc)
        seq = torch.empty(int(off[-1]) + 64, dtype=torch.uint8, device=self.device)
        self.check(self.lib.rtc_synth_genomes_dev(self.h, _np_ptr(desc), _np_ptr(off), n, _t_ptr(seq)))
        return seq

    def upload_sequences(self, seq_np):
        t = torch.empty(len(seq_np) + 64, dtype=torch.uint8, device=self.device)
        t[:len(seq_np)] = torch.from_numpy(np.array(seq_np, dtype=np.uint8, copy=True)).to(self.device)
        return t

    # ---- sketching ----------------------------------------------------------------------------
    def sketch_minhash(self, seq, off, k=21, size=1000, sizes=None, seed=42):
        """Sketch::MinHash(k,size) + update() + storeMinHashes() for every genome.
        Returns a SketchSet (strided: start[g] = g*stride)."""
        off = np.ascontiguousarray(off, dtype=np.uint64)
        n = len(off) - 1
        if sizes is not None:
            sizes = np.ascontiguousarray(sizes, dtype=np.uint32)
            stride = int(sizes.max()) if n else 1
        else:
            stride = int(size)
        stride = max(stride, 1)
        out = torch.empty((max(n, 1), stride), dtype=torch.int64, device=self.device)
        cnt = torch.zeros(max(n, 1), dtype=torch.int32, device=self.device)
        self.check(self.lib.rtc_sketch_minhash_dev(
            self.h, _t_ptr(seq), _np_ptr(off), n, k, seed,
            _np_ptr(sizes) if sizes is not None else None, int(size), _t_ptr(out), stride, _t_ptr(cnt)))
        start = torch.arange(n, dtype=torch.int64, device=self.device) * stride
        return SketchSet(out.view(-1), start, cnt[:n], 8, k, "minhash")

    def sketch_minhash_into(self, seq, off, out, cnt, k=21, size=1000, sizes=None, seed=42):
        """Same as sketch_minhash, into caller-provided rows: `out` (len(off)-1, stride) int64 and `cnt`
        int32 views (used by the multi-GPU step, which sketches in two parts so that the all-gather
        of the first overlaps the sketching of the second).  `off` may start at any base offset."""
        off = np.ascontiguousarray(off, dtype=np.uint64)
        n = len(off) - 1
        assert out.is_contiguous() and cnt.is_contiguous() and out.shape[0] == n and cnt.shape[0] == n
        if sizes is not None:
            sizes = np.ascontiguousarray(sizes, dtype=np.uint32)
        self.check(self.lib.rtc_sketch_minhash_dev(
            self.h, _t_ptr(seq), _np_ptr(off), n, k, seed,
            _np_ptr(sizes) if sizes is not None else None, int(size), _t_ptr(out), int(out.shape[1]), _t_ptr(cnt)))

    def sketch_minhash_packed(self, packed, off, k=21, size=1000, sizes=None, seed=42, n_bases=None, runs=None, out=None, cnt=None):
        """sketch_minhash over a batch in the 2-bit staging format (a PackedBatch, or `packed` uint8 device tensor of
        n_bases / 4 bytes with `runs` int64 (start, length) pairs).  `out` / `cnt`: caller-provided rows as in sketch_minhash_into."""
        if isinstance(packed, PackedBatch):
            packed, n_bases, runs = packed.packed, packed.n_bases, packed.runs
        off = np.ascontiguousarray(off, dtype=np.uint64)
        n = len(off) - 1
        if sizes is not None:
            sizes = np.ascontiguousarray(sizes, dtype=np.uint32)
            stride = int(sizes.max()) if n else 1
        else:
            stride = int(size)
        stride = max(stride, 1)
        if out is None:
            out = torch.empty((max(n, 1), stride), dtype=torch.int64, device=self.device)
            cnt = torch.zeros(max(n, 1), dtype=torch.int32, device=self.device)
        else:
            assert out.is_contiguous() and cnt.is_contiguous() and out.shape[0] >= n and cnt.shape[0] >= n
            stride = int(out.shape[1])
        n_runs = int(runs.numel() // 2) if runs is not None else 0
        self.check(self.lib.rtc_sketch_minhash_packed_dev(
            self.h, _t_ptr(packed), int(n_bases), _t_ptr(runs) if n_runs else None, n_runs, _np_ptr(off), n, k, seed,
            _np_ptr(sizes) if sizes is not None else None, int(size), _t_ptr(out), stride, _t_ptr(cnt)))
        start = torch.arange(n, dtype=torch.int64, device=self.device) * stride
        return SketchSet(out.view(-1), start, cnt[:n], 8, k, "minhash")

    def sketch_kssd(self, seq, off, shuffled_dim, kmer_size=21, drlevel=3, stride=None):
        """sketchFileWithKssd's per-file body.  shuffled_dim: int32[2^(4*half_subk)] from the host."""
        off = np.ascontiguousarray(off, dtype=np.uint64)
        n = len(off) - 1
        sd = np.ascontiguousarray(shuffled_dim, dtype=np.int32)
        half_k = (kmer_size + 1) // 2
        use64 = half_k - drlevel > 8
        maxlen = int((off[1:] - off[:-1]).max()) if n else 0
        if stride is None:
            keep = 1.0 / (16 ** drlevel)
            stride = int(maxlen * keep * 1.5) + 256
        while True:
            t_dt = torch.int64 if use64 else torch.int32
            out = torch.empty((max(n, 1), stride), dtype=t_dt, device=self.device)
            cnt = torch.zeros(max(n, 1), dtype=torch.int32, device=self.device)
            width = C.c_int()
            need = C.c_uint32()
            st = self.lib.rtc_sketch_kssd_dev(self.h, _t_ptr(seq), _np_ptr(off), n, kmer_size, drlevel,
                                              _np_ptr(sd), _t_ptr(out), stride, _t_ptr(cnt),
                                              C.byref(width), C.byref(need))
            if st == _lib.RTC_ERR_OVERFLOW:
                stride = int(need.value) + 64
                continue
            self.check(st)
            break
        start = torch.arange(n, dtype=torch.int64, device=self.device) * stride
        return SketchSet(out.view(-1), start, cnt[:n], int(width.value), half_k * 2, "kssd")

    def sketch_kssd_packed(self, packed, n_bases, runs, off, shuffled_dim, kmer_size=21, drlevel=3, stride=None):
        """sketch_kssd over a batch in the 2-bit staging format: `packed` uint8 device tensor of n_bases / 4 bytes,
        `runs` int64 device tensor of (start, length) pairs (ascending, disjoint) for everything outside ACGT.
        (`packed` may be a PackedBatch; n_bases and runs are then taken from it.)"""
        if isinstance(packed, PackedBatch):
            packed, n_bases, runs = packed.packed, packed.n_bases, packed.runs
        off = np.ascontiguousarray(off, dtype=np.uint64)
        n = len(off) - 1
        sd = np.ascontiguousarray(shuffled_dim, dtype=np.int32)
        half_k = (kmer_size + 1) // 2
        use64 = half_k - drlevel > 8
        maxlen = int((off[1:] - off[:-1]).max()) if n else 0
        if stride is None:
            stride = int(maxlen / (16 ** drlevel) * 1.5) + 256
        n_runs = int(runs.numel() // 2) if runs is not None else 0
        while True:
            out = torch.empty((max(n, 1), stride), dtype=torch.int64 if use64 else torch.int32, device=self.device)
            cnt = torch.zeros(max(n, 1), dtype=torch.int32, device=self.device)
            width = C.c_int()
            need = C.c_uint32()
            st = self.lib.rtc_sketch_kssd_packed_dev(self.h, _t_ptr(packed), int(n_bases), _t_ptr(runs) if n_runs else None,
                                                     n_runs, _np_ptr(off), n, kmer_size, drlevel, _np_ptr(sd), _t_ptr(out),
                                                     stride, _t_ptr(cnt), C.byref(width), C.byref(need))
            if st == _lib.RTC_ERR_OVERFLOW:
                stride = int(need.value) + 64
                continue
            self.check(st)
            break
        start = torch.arange(n, dtype=torch.int64, device=self.device) * stride
        return SketchSet(out.view(-1), start, cnt[:n], int(width.value), half_k * 2, "kssd")

    # ---- all pairs ----------------------------------------------------------------------------
    def pair_common(self, sk, row0=0, row1=None, col0=0, col1=None, lower_only=False, algo=0, out=None):
        n = sk.n
        row1 = n if row1 is None else row1
        col1 = n if col1 is None else col1
        ld = col1 - col0
        if out is None:
            out = torch.zeros((max(row1 - row0, 1), max(ld, 1)), dtype=torch.int32, device=self.device)
        self.check(self.lib.rtc_pair_common_dev(self.h, _t_ptr(sk.hashes), sk.width, _t_ptr(sk.start),
                                                _t_ptr(sk.len), n, row0, row1, col0, col1, _t_ptr(out),
                                                out.stride(0), int(lower_only), algo))
        return out

    def pair_mash(self, sk, sketch_size, row0=0, row1=None, col0=0, col1=None):
        """Mash's union-truncated estimator per pair (modifyMST's distance(), D3): (common, denom) tensors."""
        n = sk.n
        row1 = n if row1 is None else row1
        col1 = n if col1 is None else col1
        shape = (max(row1 - row0, 1), max(col1 - col0, 1))
        common = torch.zeros(shape, dtype=torch.int32, device=self.device)
        denom = torch.zeros(shape, dtype=torch.int32, device=self.device)
        self.check(self.lib.rtc_pair_mash_dev(self.h, _t_ptr(sk.hashes), sk.width, _t_ptr(sk.start), _t_ptr(sk.len), n,
                                              int(sketch_size), row0, row1, col0, col1, _t_ptr(common), _t_ptr(denom),
                                              common.stride(0)))
        return common, denom

    def extract_edges(self, common, sk, row0, row1, col0, col1, radio, cap):
        edges = torch.empty((max(cap, 1), 3), dtype=torch.int32, device=self.device)
        count = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.check(self.lib.rtc_extract_edges_dev(self.h, _t_ptr(common), common.stride(0), row0, row1,
                                                  col0, col1, _t_ptr(sk.len), radio, _t_ptr(edges), cap,
                                                  _t_ptr(count)))
        return edges, count

    def mst_dense(self, sk, threshold, is_containment=False, span=100, start_index=0):
        """rtc_mst_dense: (edge.mst records, dense[span, n] int32, ani[101] u64) -- the --dense by-products."""
        n = sk.n
        out = np.zeros(max(n, 1), dtype=EDGE_DT)
        dense = np.zeros((span, max(n, 1)), dtype=np.int32)
        ani = np.zeros(101, dtype=np.uint64)
        m = C.c_uint64()
        self.check(self.lib.rtc_mst_dense(self.h, _t_ptr(sk.hashes), sk.width, _t_ptr(sk.start), _t_ptr(sk.len), n,
                                          int(start_index), sk.k, int(is_containment), float(threshold), _np_ptr(out),
                                          C.byref(m), span, _np_ptr(dense), _np_ptr(ani)))
        return out[:m.value].copy(), dense[:, :n], ani

    def pair_edges(self, sk, row0, row1, col0, col1, radio, cap):
        """Fused form of pair_common + extract_edges (no dense matrix).  Returns (edges, count)."""
        edges = torch.empty((max(cap, 1), 3), dtype=torch.int32, device=self.device)
        count = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.check(self.lib.rtc_pair_edges_dev(self.h, _t_ptr(sk.hashes), sk.width, _t_ptr(sk.start), _t_ptr(sk.len),
                                               sk.n, row0, row1, col0, col1, int(radio), _t_ptr(edges), cap,
                                               _t_ptr(count)))
        return edges, int(count.item())

    def mst(self, sk, threshold, is_containment=False, start_index=0):
        """compute_minhash_mst / compute_kssd_mst: returns numpy EDGE_DT array (edge.mst records).
        start_index > 0: only rows >= start_index (the --append form, src/MST.cpp:1375-1383)."""
        n = sk.n
        out = np.zeros(max(n, 1), dtype=EDGE_DT)
        m = C.c_uint64()
        self.check(self.lib.rtc_mst_append(self.h, _t_ptr(sk.hashes), sk.width, _t_ptr(sk.start), _t_ptr(sk.len),
                                           n, int(start_index), sk.k, int(is_containment), float(threshold),
                                           _np_ptr(out), C.byref(m)))
        return out[:m.value].copy()

    def sketch_minhash_sharded(self, comm, seq, off, k=21, size=1000, sizes=None, stride=None, seed=42):
        """Multi-GPU sketch phase behind the C ABI: this rank's genomes into its block of the canonical
        global buffers, gathered to every rank (two parts, the first gather overlapping the second
        sketch launch).  Returns the global SketchSet (comm.size * n_local genomes)."""
        off = np.ascontiguousarray(off, dtype=np.uint64)
        n_local = len(off) - 1
        if sizes is not None:
            sizes = np.ascontiguousarray(sizes, dtype=np.uint32)
        if stride is None:
            stride = int(sizes.max()) if sizes is not None else int(size)
            if sizes is not None and comm.size > 1:  # per-rank maxima differ: agree
                stride = int(comm.all_reduce_host([stride], "max")[0])
        n = comm.size * n_local
        out = torch.empty((max(n, 1), max(stride, 1)), dtype=torch.int64, device=self.device)
        cnt = torch.zeros(max(n, 1), dtype=torch.int32, device=self.device)
        self.check(self.lib.rtc_sketch_minhash_sharded(
            self.h, comm.h, _t_ptr(seq), _np_ptr(off), n_local, k, seed,
            _np_ptr(sizes) if sizes is not None else None, int(size), _t_ptr(out), max(stride, 1), _t_ptr(cnt)))
        start = torch.arange(n, dtype=torch.int64, device=self.device) * max(stride, 1)
        return SketchSet(out.view(-1), start, cnt[:n], 8, k, "minhash")

    def sketch_packed_sharded(self, comm, batches, mode="minhash", k=21, size=1000, seed=42, drlevel=3, shuffled_dim=None,
                              stride=None, out=None, cnt=None):
        """Multi-GPU sketch phase behind the C ABI for a rank whose genomes are resident as batches in the 2-bit staging
        format: `batches` = [(PackedBatch, off), ...] in the order of the rank's rows (every rank: the same batch sizes).
        rtc_sketch_minhash_packed_sharded / rtc_sketch_kssd_packed_sharded per batch; a batch's gather travels beside the
        next batch's sketch kernel.  Returns the global SketchSet (comm.size * n_local genomes, canonical order).
        `out` / `cnt`: caller-provided global rows (reused between steps)."""
        offs = [np.ascontiguousarray(o, dtype=np.uint64) for _, o in batches]
        n_local = sum(len(o) - 1 for o in offs)
        n = comm.size * n_local
        kssd = mode == "kssd"
        if kssd:
            sd = np.ascontiguousarray(shuffled_dim, dtype=np.int32)
            half_k = (k + 1) // 2
            width = 8 if half_k - drlevel > 8 else 4
            kk = half_k * 2
            if stride is None:  # 1.25 x the expected count of the longest genome: ample for genomes of one length
                longest = max(int((o[1:] - o[:-1]).max()) for o in offs)
                stride = (int(longest / (16 ** drlevel) * 1.25) + 64 + 3) // 4 * 4
                stride = int(comm.all_reduce_host([stride], "max")[0])
        else:
            width, kk = 8, k
            stride = int(size) if stride is None else int(stride)
        t_dt = torch.int64 if width == 8 else torch.int32
        while True:
            if out is None or out.shape != (max(n, 1), stride) or out.dtype != t_dt:
                out = torch.empty((max(n, 1), stride), dtype=t_dt, device=self.device)
                cnt = torch.zeros(max(n, 1), dtype=torch.int32, device=self.device)
            row = 0
            st = _lib.RTC_OK
            need = C.c_uint32()
            for i, ((pb, _), off) in enumerate(zip(batches, offs)):
                nb = len(off) - 1
                last = int(i == len(batches) - 1)
                n_runs = int(pb.runs.numel() // 2) if pb.runs is not None else 0
                runs = _t_ptr(pb.runs) if n_runs else None
                if kssd:
                    w = C.c_int()
                    st = self.lib.rtc_sketch_kssd_packed_sharded(self.h, comm.h, _t_ptr(pb.packed), pb.n_bases, runs, n_runs, _np_ptr(off), nb,
                                                                 row, n_local, last, k, drlevel, _np_ptr(sd), _t_ptr(out), stride,
                                                                 _t_ptr(cnt), C.byref(w), C.byref(need))
                else:
                    st = self.lib.rtc_sketch_minhash_packed_sharded(self.h, comm.h, _t_ptr(pb.packed), pb.n_bases, runs, n_runs, _np_ptr(off),
                                                                    nb, row, n_local, last, k, seed, None, int(size), _t_ptr(out), stride,
                                                                    _t_ptr(cnt))
                if st != _lib.RTC_OK:
                    break
                row += nb
            if kssd and st == _lib.RTC_ERR_OVERFLOW:  # every rank got it, with the same need: wider rows, once more
                stride = (int(need.value) + 64 + 3) // 4 * 4
                out = None
                continue
            self.check(st)
            break
        start = torch.arange(n, dtype=torch.int64, device=self.device) * stride
        return SketchSet(out.view(-1), start, cnt[:n], width, kk, "kssd" if kssd else "minhash")

    def mst_sharded(self, comm, sk, threshold, is_containment=False):
        """rtc_mst across the ranks of `comm`; returns (edge.mst records, ShardStats)."""
        n = sk.n
        out = np.empty(max(n, 1), dtype=EDGE_DT)  # (the call writes the first m records; the caller gets a view of them)
        m = C.c_uint64()
        stats = _lib.ShardStats()
        self.check(self.lib.rtc_mst_sharded(self.h, comm.h, _t_ptr(sk.hashes), sk.width, _t_ptr(sk.start), _t_ptr(sk.len),
                                            n, sk.k, int(is_containment), float(threshold), _np_ptr(out), C.byref(m),
                                            C.byref(stats)))
        return out[:m.value], stats

    def mst_mash(self, sk, sketch_size, is_containment=False, start_index=0, span=0):
        """modifyMST (the dense loop): spanning tree over EVERY pair, Mash-estimator / containDistance weights.
        Returns edge.mst records {i < j}; with span > 0 also (dense[span, n], ani[101])."""
        n = sk.n
        out = np.zeros(max(n, 1), dtype=EDGE_DT)
        dense = np.zeros((max(span, 1), max(n, 1)), dtype=np.int32)
        ani = np.zeros(101, dtype=np.uint64)
        m = C.c_uint64()
        self.check(self.lib.rtc_mst_mash(self.h, _t_ptr(sk.hashes), sk.width, _t_ptr(sk.start), _t_ptr(sk.len), n,
                                         int(start_index), sk.k, int(is_containment), int(sketch_size), _np_ptr(out),
                                         C.byref(m), int(span), _np_ptr(dense) if span else None, _np_ptr(ani) if span else None))
        return (out[:m.value].copy(), dense[:, :n], ani) if span else out[:m.value].copy()

    def greedy_mash(self, sk, threshold, sketch_size, is_containment=False):
        """greedyCluster (legacy loop, every representative, Mash estimator): (n_clusters, rep_of)."""
        n = sk.n
        rep = np.zeros(max(n, 1), dtype=np.int32)
        nc = C.c_uint32()
        self.check(self.lib.rtc_greedy_mash(self.h, _t_ptr(sk.hashes), sk.width, _t_ptr(sk.start), _t_ptr(sk.len), n, sk.k,
                                            int(is_containment), int(sketch_size), float(threshold), _np_ptr(rep), C.byref(nc)))
        return int(nc.value), rep[:n].copy()

    def greedy(self, sk, threshold, size_cfg=None, is_containment=False):
        n = sk.n
        rep = np.zeros(max(n, 1), dtype=np.int32)
        nc = C.c_uint32()
        cfg = None
        if size_cfg is not None:
            cfg = np.ascontiguousarray(np.broadcast_to(np.asarray(size_cfg, dtype=np.uint32), (n,)))
        self.check(self.lib.rtc_greedy(self.h, _t_ptr(sk.hashes), sk.width, _t_ptr(sk.start), _t_ptr(sk.len),
                                       n, _np_ptr(cfg) if cfg is not None else None, sk.k,
                                       int(is_containment), int(sk.kind == "kssd"), float(threshold),
                                       _np_ptr(rep), C.byref(nc)))
        return int(nc.value), rep[:n].copy()


class Comm:
    """One rtc_comm (RCCL communicator of one GPU / context).  Ranks are processes (init_rank, the id
    from rank 0 travels through the launcher's channel, e.g. torch.distributed) or threads of one
    process (init_all)."""

    def __init__(self, ctx, handle):
        self.ctx, self.h = ctx, handle
        lib = ctx.lib
        self.rank, self.size = lib.rtc_comm_rank(handle), lib.rtc_comm_size(handle)
        self.backend = lib.rtc_comm_backend(handle).decode()

    @staticmethod
    def unique_id(lib=None):
        lib = lib or _lib.load()
        buf = (C.c_char * 128)()
        st = lib.rtc_comm_unique_id(buf)
        if st != _lib.RTC_OK:
            raise RtcError(st, "rtc_comm_unique_id: " + lib.rtc_last_error(None).decode(errors="replace"))
        return bytes(buf)

    @staticmethod
    def init_rank(ctx, nranks, rank, uid):
        h = C.c_void_p()
        buf = (C.c_char * 128).from_buffer_copy(uid) if uid is not None else None
        ctx.check(ctx.lib.rtc_comm_init_rank(ctx.h, nranks, rank, buf, C.byref(h)))
        return Comm(ctx, h)

    @staticmethod
    def init_all(ctxs):
        n = len(ctxs)
        hs = (C.c_void_p * n)(*[c.h for c in ctxs])
        out = (C.c_void_p * n)()
        st = ctxs[0].lib.rtc_comm_init_all(hs, n, out)
        ctxs[0].check(st)
        return [Comm(c, C.c_void_p(out[i])) for i, c in enumerate(ctxs)]

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.rtc_comm_destroy(self.h)
            self.h = None

    def all_reduce(self, t, op="min"):
        dtype = {torch.int64: 0, torch.int32: 1}[t.dtype]
        self.ctx.check(self.ctx.lib.rtc_comm_all_reduce(self.h, _t_ptr(t), t.numel(), dtype, 0 if op == "min" else 1))

    def all_reduce_host(self, vals, op="max"):
        a = np.ascontiguousarray(vals, dtype=np.int64)
        self.ctx.check(self.ctx.lib.rtc_comm_all_reduce_host(self.h, _np_ptr(a), len(a), 0 if op == "min" else 1))
        return a

    def gather_rows(self, t_global, n_local, a, b, async_=False):
        row_bytes = t_global.numel() * t_global.element_size() // (self.size * n_local)
        self.ctx.check(self.ctx.lib.rtc_comm_gather_rows(self.h, _t_ptr(t_global), row_bytes, n_local, a, b, int(async_)))

    def wait(self):
        self.ctx.check(self.ctx.lib.rtc_comm_wait(self.h))


# ---- host-side arithmetic shared by CLI-equivalent flows (reference expression order) -------------
def mst_radio(threshold, kmer_size):
    """src/MST.cpp:26-37,1292: (int)(2*exp(thr*(k-1)) - 1)."""
    return int(2.0 * math.exp(threshold * (kmer_size - 1)) - 1.0)


def mst_distance(common, size0, size1, kmer_size, is_containment):
    """src/MST.cpp:1295,1489-1515 evaluated with the same operation order (host libm)."""
    inv_k = 1.0 / kmer_size
    if not is_containment:
        denom = size0 + size1 - common
        jac = 0.0 if denom == 0 else common / denom
        if jac == 1.0:
            return 0.0
        if jac == 0.0:
            return 1.0
        ratio = (2.0 * jac) / (1.0 + jac)
        return -inv_k * math.log(ratio)
    denom = min(size0, size1)
    c = 0.0 if denom == 0 else common / denom
    if c == 1.0:
        return 0.0
    if c == 0.0:
        return 1.0
    return -inv_k * math.log(c)


class PackedBatch:
    """A batch in the command lines' 2-bit staging format, resident in HBM (include/rtclust.h, rtc_unpack_bases_dev):
    `packed` uint8[n_bases / 4], `runs` int64[2 r] = (start, length) of every stretch outside ACGT (ascending)."""

    def __init__(self, packed, n_bases, runs):
        self.packed, self.n_bases, self.runs = packed, int(n_bases), runs


def pack_staging(seq, total):
    """Characters resident in HBM -> PackedBatch, what the command lines' parser produces on the host (rtc_host.cpp:
    PackedSink).  Torch plumbing for benchmarks and tests, in pieces that keep the temporaries small; not a product path."""
    total = int(total)
    n_bases = (total + 63) // 64 * 64 + 64
    dev = seq.device
    packed = torch.empty(n_bases // 4, dtype=torch.uint8, device=dev)
    runs = []
    step = 1 << 28
    for a in range(0, n_bases, step):
        b = min(a + step, n_bases)
        x = seq[a:min(b, total)]
        if x.numel() < b - a:
            x = torch.cat([x, torch.full((b - a - x.numel(),), ord("N"), dtype=torch.uint8, device=dev)])
        c = (((x >> 1) ^ (x >> 2)) & 3).view(-1, 4)
        packed[a // 4:b // 4] = c[:, 0] | (c[:, 1] << 2) | (c[:, 2] << 4) | (c[:, 3] << 6)
        up = x & 0xDF
        idx = torch.nonzero(~((up == 65) | (up == 67) | (up == 71) | (up == 84))).view(-1)
        if idx.numel():
            first = torch.ones_like(idx, dtype=torch.bool)
            first[1:] = idx[1:] != idx[:-1] + 1
            last = torch.ones_like(idx, dtype=torch.bool)
            last[:-1] = first[1:]
            starts, ends = idx[first], idx[last] + 1
            runs.append(torch.stack([starts + a, ends - starts], dim=1).reshape(-1))  # a stretch across a seam: two runs that touch
        del x, c, up, idx
    runs = torch.cat(runs).contiguous() if runs else torch.zeros(0, dtype=torch.int64, device=dev)
    return PackedBatch(packed, n_bases, runs)


def synth_family_descs(n_families, per_family, global_seed=42, max_rate=0.08, n_every=0):
    """SURVEY.md 8d synthetic design: families of `per_family` members, member 0 is the ancestor,
    member m>0 carries substitutions at rate U[0,max_rate] drawn from a counter-based stream."""
    def mix(x):
        x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = x
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)
    n = n_families * per_family
    d = np.zeros(n, dtype=SYNTH_DT)
    for f in range(n_families):
        fs = mix((global_seed << 20) ^ (f * 2 + 1))
        for m in range(per_family):
            g = f * per_family + m
            ms = mix(fs ^ (m * 0x632BE59BD9B4E019 & 0xFFFFFFFFFFFFFFFF))
            rate = 0.0 if m == 0 else max_rate * ((mix(ms ^ 0xABCDEF) >> 11) / float(1 << 53))
            d[g] = (fs, ms, int(rate * 16384), n_every)
    return d

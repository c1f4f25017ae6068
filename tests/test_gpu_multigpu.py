"""GPU: the multi-GPU path behind the C ABI on a one-GPU box.

RCCL refuses two ranks on one device, so `rtc_comm_init_all` over two contexts of the same device
uses its in-process exchange; everything above the exchange -- canonical sketch blocks, two-part
gather, triangle row ranges, per-round all-reduce + device union (rtc_mst_sharded), the CLI's
`--gpus 0,0` mode -- is the code that runs with RCCL on distinct GPUs.  The RCCL calls themselves are
driven with one rank (RTC_COMM_FORCE_RCCL=1): ncclAllReduce and the grouped in-place ncclBroadcast."""
import os
import subprocess
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _reload_options():
    """the library reads its RTC_* switches when a context is created: every live context reads them again"""
    from rabbittclust_amd import api
    api.reload_all_options()

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "rabbittclust_amd", "bin")


def _threads(fns):
    err, out = [], [None] * len(fns)

    def run(i):
        try:
            out[i] = fns[i]()
        except BaseException as e:  # noqa: BLE001 -- reported to the main thread
            err.append(e)
    ts = [threading.Thread(target=run, args=(i,)) for i in range(len(fns))]
    for t in ts:
        t.start()
    for t in ts:
        t.join(300)
    assert not any(t.is_alive() for t in ts), "a rank hung"
    if err:
        raise err[0]
    return out


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("fixed", [True, False])
def test_sharded_step_in_process_ranks_equal_single_gpu(oracle, world, fixed):
    """world contexts on device 0, one host thread each: sharded sketch + gather + sharded MST.  Every
    rank must end with the global sketches in canonical order and the forest of the single-GPU run."""
    import torch
    from rabbittclust_amd import api
    ctxs = [api.Context(0) for _ in range(world)]
    comms = api.Comm.init_all(ctxs)
    assert [c.rank for c in comms] == list(range(world)) and all(c.backend == "in-process" for c in comms)
    n_local, L, s = 40, 50_000, 300
    descs = [api.synth_family_descs(n_local // 5, 5, global_seed=100 + 7 * (r // 2)) for r in range(world)]  # ranks 0/1 share families
    off = np.arange(n_local + 1, dtype=np.uint64) * L
    seqs = [ctxs[r].synth_genomes(descs[r], off) for r in range(world)]
    sizes = None if fixed else [np.array([200 + 25 * ((g + r) % 5) for g in range(n_local)], dtype=np.uint32) for r in range(world)]
    torch.cuda.synchronize()

    def rank_fn(r):
        def f():
            sk = ctxs[r].sketch_minhash_sharded(comms[r], seqs[r], off, k=21, size=s, sizes=None if fixed else sizes[r])
            mst, st = ctxs[r].mst_sharded(comms[r], sk, 0.05)
            ctxs[r].sync()
            return sk.to_host(), mst, (st.row0, st.row1, st.cand_edges, st.rounds, st.s_fixed)
        return f

    res = _threads([rank_fn(r) for r in range(world)])
    # reference: everything on one context
    one = api.Context(0)
    want_sk = []
    for r in range(world):
        want_sk += one.sketch_minhash(seqs[r], off, k=21, size=s, sizes=None if fixed else sizes[r]).to_host()
    for r in range(world):
        assert len(res[r][0]) == world * n_local
        assert all(np.array_equal(a, b) for a, b in zip(res[r][0], want_sk)), f"rank {r}: gathered sketches differ"
    dev = api.SketchSet.from_host(want_sk, one.device, k=21)
    want_mst = one.mst(dev, 0.05)
    flat, start, lens = oracle.to_csr(want_sk)
    omst = oracle.mst(flat, start, lens, 21, 0, 0.05)
    assert np.array_equal(np.sort(want_mst["dist"]).view(np.uint64), np.sort(omst["dist"]).view(np.uint64))
    rows = []
    for r in range(world):
        assert np.array_equal(res[r][1], want_mst), f"rank {r}: forest differs from the single-GPU forest"
        rows.append(res[r][2][:2])
        assert (res[r][2][4] == s) == fixed
    assert rows[0][0] == 0 and rows[-1][1] == world * n_local and all(rows[i][1] == rows[i + 1][0] for i in range(world - 1))
    assert sum(x[2][2] for x in res) > 0
    for c in comms:
        c.close()
    for c in ctxs + [one]:
        c.close()


@pytest.mark.parametrize("mode", ["minhash", "kssd"])
@pytest.mark.parametrize("world", [8, 3])
def test_packed_sharded_step_in_process_ranks_equal_single_launch(oracle, mode, world):
    """The strong-scaling step from PACKED input: `world` contexts on device 0, one host thread each, every rank's genomes
    resident as batches in the 2-bit staging format (unequal batch sizes, several batches per rank, N runs inside), sketched
    by rtc_sketch_minhash_packed_sharded / rtc_sketch_kssd_packed_sharded into the global rows, then rtc_mst_sharded.  Every
    rank must end with the sketches of the single launch over all genomes (bit for bit, canonical order) and its forest."""
    import torch
    from rabbittclust_amd import api, host, pipeline
    ctxs = [api.Context(0) for _ in range(world)]
    comms = api.Comm.init_all(ctxs)
    assert all(c.backend == "in-process" for c in comms)
    n_local, L = 27, (60_000 if mode == "minhash" else 300_000)
    sizes = [11, 9, 7]  # the rank's batches
    sd = host.generate_shuffle_dim(6) if mode == "kssd" else None
    one = api.Context(0)
    desc = api.synth_family_descs(world * n_local // 9 + 1, 9, global_seed=31, n_every=20_011)[: world * n_local]
    off_all = np.arange(world * n_local + 1, dtype=np.uint64) * L
    seq_all = one.synth_genomes(desc, off_all)
    one.sync()
    if mode == "minhash":
        want = one.sketch_minhash(seq_all, off_all, k=21, size=250)
    else:
        want = one.sketch_kssd(seq_all, off_all, sd, kmer_size=21, drlevel=3)
    want_mst = one.mst(want, 0.05)
    want_sk = want.to_host()
    assert len(want_mst) > world * n_local // 2
    batches = []
    for r in range(world):
        bs, a = [], r * n_local
        for m in sizes:
            o = np.arange(m + 1, dtype=np.uint64) * L
            piece = seq_all[a * L:(a + m) * L].clone()
            bs.append((api.pack_staging(piece, m * L), o))
            a += m
        batches.append(bs)
    torch.cuda.synchronize()

    def rank_fn(r):
        def f():
            pipe = pipeline.MstPipeline(ctxs[r], k=21, sketch_size=250, threshold=0.05, mode=mode, shuffled_dim=sd,
                                        comm=pipeline.NativeComm(comms[r]))
            st = pipe.step(batches[r])
            again = pipe.step(batches[r])  # the global rows are reused by the next step
            ctxs[r].sync()
            return pipe.last_sketches.to_host(), pipe.last_mst, st, again
        return f

    res = _threads([rank_fn(r) for r in range(world)])
    for r in range(world):
        got_sk, got_mst, st, again = res[r]
        assert len(got_sk) == world * n_local
        assert all(np.array_equal(a, b) for a, b in zip(got_sk, want_sk)), f"rank {r}: gathered sketches differ from the single launch"
        assert np.array_equal(got_mst, want_mst), f"rank {r}: forest differs from the single-GPU forest"
        assert st["mst_edges"] == again["mst_edges"] == len(want_mst)
    assert sum(x[2]["pairs_local"] for x in res) == world * n_local * (world * n_local - 1) // 2
    flat, start, lens = oracle.to_csr(want_sk, dtype=want_sk[0].dtype)
    omst = oracle.mst(flat, start, lens, want.k, 0, 0.05)
    assert np.array_equal(np.sort(want_mst["dist"]).view(np.uint64), np.sort(omst["dist"]).view(np.uint64))
    for c in comms:
        c.close()
    for c in ctxs + [one]:
        c.close()


def test_packed_sharded_kssd_overflow_is_agreed_by_all_ranks():
    """rows too narrow for one rank's sketches: every rank gets RTC_ERR_OVERFLOW with the same need, nobody hangs; the api
    wrapper then repeats the phase with wider rows."""
    import torch
    from rabbittclust_amd import _lib, api, host
    import ctypes as C
    world = 2
    ctxs = [api.Context(0) for _ in range(world)]
    comms = api.Comm.init_all(ctxs)
    sd = host.generate_shuffle_dim(6)
    n_local, Ls = 12, [120_000, 400_000]  # rank 1's genomes yield ~100 tuples, rank 0's ~30
    batches = []
    for r in range(world):
        off = np.arange(n_local + 1, dtype=np.uint64) * Ls[r]
        seq = ctxs[r].synth_genomes(api.synth_family_descs(3, 4, global_seed=5 + r), off)
        ctxs[r].sync()
        batches.append([(api.pack_staging(seq, int(off[-1])), off)])
    torch.cuda.synchronize()

    def raw(r):
        def f():
            c, pb, off = ctxs[r], batches[r][0][0], batches[r][0][1]
            out = torch.empty((world * n_local, 48), dtype=torch.int32, device=c.device)
            cnt = torch.zeros(world * n_local, dtype=torch.int32, device=c.device)
            w, need = C.c_int(), C.c_uint32()
            st = c.lib.rtc_sketch_kssd_packed_sharded(c.h, comms[r].h, api._t_ptr(pb.packed), pb.n_bases, api._t_ptr(pb.runs) if pb.runs.numel() else None,
                                                      pb.runs.numel() // 2, api._np_ptr(off), n_local, 0, n_local, 1, 21, 3, api._np_ptr(np.ascontiguousarray(sd, dtype=np.int32)),
                                                      api._t_ptr(out), 48, api._t_ptr(cnt), C.byref(w), C.byref(need))
            return st, int(need.value), int(w.value)
        return f
    res = _threads([raw(r) for r in range(world)])
    assert res[0][0] == res[1][0] == _lib.RTC_ERR_OVERFLOW and res[0][1] == res[1][1] > 48 and res[0][2] == res[1][2] == 4
    # the wrapper: too tight a stride on purpose, then the agreed need
    def wrapped(r):
        def f():
            sk = ctxs[r].sketch_packed_sharded(comms[r], batches[r], mode="kssd", k=21, drlevel=3, shuffled_dim=sd, stride=48)
            ctxs[r].sync()
            return sk.to_host()
        return f
    got = _threads([wrapped(r) for r in range(world)])
    one = api.Context(0)
    want = []
    for r in range(world):
        want += one.sketch_kssd_packed(batches[r][0][0], None, None, batches[r][0][1], sd, kmer_size=21, drlevel=3).to_host()
    for r in range(world):
        assert all(np.array_equal(a, b) for a, b in zip(got[r], want))
    for c in comms:
        c.close()
    for c in ctxs + [one]:
        c.close()


def test_rccl_calls_single_rank(oracle):
    """One rank, communicator forced through RCCL: ncclCommInitRank, ncclAllReduce(MIN/MAX) and the
    grouped in-place ncclBroadcast run on the GPU; the sharded step equals rtc_mst."""
    import torch
    from rabbittclust_amd import api, pipeline
    os.environ["RTC_COMM_FORCE_RCCL"] = "1"
    _reload_options()
    try:
        ctx = api.Context(0)
        comm = api.Comm.init_rank(ctx, 1, 0, api.Comm.unique_id(ctx.lib))
    finally:
        del os.environ["RTC_COMM_FORCE_RCCL"]
        _reload_options()
    assert comm.backend == "rccl" and comm.size == 1
    t = torch.tensor([5, -3, 1 << 40], dtype=torch.int64, device=ctx.device)
    comm.all_reduce(t, "min"); comm.all_reduce(t, "max")
    assert t.tolist() == [5, -3, 1 << 40]
    assert comm.all_reduce_host([7, -2], "max").tolist() == [7, -2]
    desc = api.synth_family_descs(8, 5, global_seed=3)
    L = 60_000
    off = np.arange(len(desc) + 1, dtype=np.uint64) * L
    seq = ctx.synth_genomes(desc, off)
    pipe = pipeline.MstPipeline(ctx, k=21, sketch_size=400, threshold=0.05, comm=pipeline.NativeComm(comm))
    st = pipe.step(seq, off)
    ref = pipeline.MstPipeline(ctx, k=21, sketch_size=400, threshold=0.05)
    ref.step(seq, off)
    assert all(np.array_equal(a, b) for a, b in zip(pipe.last_sketches.to_host(), ref.last_sketches.to_host()))
    assert np.array_equal(pipe.last_mst, ref.last_mst) and st["mst_edges"] == len(ref.last_mst) > 0
    comm.close()
    ctx.close()


def test_native_kssd_step_in_process_ranks(oracle):
    """--fast (KSSD) multi-GPU step: local sketch, stride agreement, gather, sharded u32 MST."""
    import torch
    from rabbittclust_amd import api, host, pipeline
    world = 2
    ctxs = [api.Context(0) for _ in range(world)]
    comms = api.Comm.init_all(ctxs)
    sd = host.generate_shuffle_dim(6)
    n_local = 20
    Ls = [150_000, 260_000]  # different lengths -> different local strides
    seqs, offs = [], []
    for r in range(world):
        desc = api.synth_family_descs(4, 5, global_seed=50)  # same families on both ranks, different lengths: prefixes overlap
        off = np.arange(n_local + 1, dtype=np.uint64) * Ls[r]
        seqs.append(ctxs[r].synth_genomes(desc, off)); offs.append(off)
    torch.cuda.synchronize()

    def rank_fn(r):
        def f():
            pipe = pipeline.MstPipeline(ctxs[r], k=21, threshold=0.05, mode="kssd", shuffled_dim=sd, comm=pipeline.NativeComm(comms[r]))
            pipe.step(seqs[r], offs[r])
            return pipe.last_sketches.to_host(), pipe.last_mst
        return f

    res = _threads([rank_fn(r) for r in range(world)])
    one = api.Context(0)
    want = []
    for r in range(world):
        want += one.sketch_kssd(seqs[r], offs[r], sd, kmer_size=21, drlevel=3).to_host()
    assert want[0].dtype == np.uint32
    g0 = seqs[0][: Ls[0]].cpu().numpy()
    assert np.array_equal(want[0], oracle.kssd_sketch(g0, 21, 3))
    for r in range(world):
        assert all(np.array_equal(a, b) for a, b in zip(res[r][0], want))
    dev = api.SketchSet.from_host(want, one.device, k=22, kind="kssd", width=4)
    ref = one.mst(dev, 0.05)
    assert len(ref) > 0 and np.array_equal(res[0][1], ref) and np.array_equal(res[1][1], ref)
    flat, start, lens = oracle.to_csr(want, dtype=np.uint32)
    omst = oracle.mst(flat, start, lens, 22, 0, 0.05)
    assert np.array_equal(np.sort(ref["dist"]).view(np.uint64), np.sort(omst["dist"]).view(np.uint64))
    for c in comms:
        c.close()
    for c in ctxs + [one]:
        c.close()


def test_mismatched_ranks_are_reported(oracle):
    """ranks bringing different genome counts must fail with RTC_ERR_ARG on every rank, not hang"""
    import torch
    from rabbittclust_amd import _lib, api
    ctxs = [api.Context(0) for _ in range(2)]
    comms = api.Comm.init_all(ctxs)
    L = 30_000
    ns = [10, 15]
    seqs, offs = [], []
    for r in range(2):
        desc = api.synth_family_descs(ns[r] // 5, 5, global_seed=9)
        off = np.arange(ns[r] + 1, dtype=np.uint64) * L
        seqs.append(ctxs[r].synth_genomes(desc, off)); offs.append(off)
    torch.cuda.synchronize()

    def rank_fn(r):
        def f():
            try:
                ctxs[r].sketch_minhash_sharded(comms[r], seqs[r], offs[r], k=21, size=100)
            except _lib.RtcError as e:
                return e.status
            return 0
        return f
    assert _threads([rank_fn(0), rank_fn(1)]) == [_lib.RTC_ERR_ARG, _lib.RTC_ERR_ARG]


def _cli(args, cwd, env=None):
    r = subprocess.run(args, cwd=cwd, capture_output=True, text=True, timeout=900, env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stderr[-3000:]
    return r.stderr


@pytest.mark.parametrize("fast", [False, True])
def test_clust_mst_two_gpu_mode_equals_one_gpu(oracle, tmp_path, fast):
    """clust-mst --gpus 0,0 (two contexts, two host threads, batches round-robin, share step, sharded MST)
    must write byte-identical hash.sketch / edge.mst / cluster text to the --gpus 1 run; RTC_HOST_SKETCHES=1
    (sketches uploaded from the host vectors instead of staying in HBM) as well."""
    from test_gpu_cli import _write_family_fastas
    tmp = str(tmp_path)
    L = 1_800_000
    lst, paths, seqs = _write_family_fastas(oracle, tmp, 4, 4, L, seed=12)
    outs = {}
    for tag, gpus, env in (("one", "1", {}), ("two", "0,0", {}), ("host", "0,0", {"RTC_HOST_SKETCHES": "1"})):
        d = os.path.join(tmp, tag)
        os.makedirs(d)
        out = os.path.join(d, "res.out")
        cmd = [os.path.join(BIN, "clust-mst"), "-l", "-i", lst, "-k", "21", "-d", "0.05", "-t", "4", "--gpus", gpus, "-o", out]
        cmd += ["--fast"] if fast else ["-s", "600"]
        err = _cli(cmd, d, dict(env, RTC_BATCH_BYTES=str(5 << 20), RTC_VERBOSE="1"))  # 5 MiB batches: several per GPU
        if tag != "one":
            assert "use 2 GPUs (in-process exchange)" in err
        folder = [os.path.join(d, x) for x in os.listdir(d) if os.path.isdir(os.path.join(d, x))][0]
        files = sorted(f for f in os.listdir(folder) if "info" not in f)
        outs[tag] = (open(out).read(), {f: open(os.path.join(folder, f), "rb").read() for f in files})
    assert outs["one"][0] == outs["two"][0] == outs["host"][0]
    assert outs["one"][1].keys() == outs["two"][1].keys()
    for f in outs["one"][1]:
        assert outs["one"][1][f] == outs["two"][1][f] == outs["host"][1][f], f
    assert ("kssd.hash.sketch" if fast else "hash.sketch") in outs["one"][1] and "edge.mst" in outs["one"][1]


def test_clust_mst_no_save_keeps_sketches_on_device(oracle, tmp_path):
    """-e: nothing is written and no sketch travels to the host; the clusters equal the saving run's"""
    from test_gpu_cli import _write_family_fastas, _parse_clusters, _partition
    tmp = str(tmp_path)
    lst, paths, seqs = _write_family_fastas(oracle, tmp, 3, 3, 1_800_000, seed=14)
    a, b = os.path.join(tmp, "a.out"), os.path.join(tmp, "b.out")
    err = _cli([os.path.join(BIN, "clust-mst"), "-l", "-i", lst, "-k", "21", "-s", "500", "-d", "0.05", "-e", "-o", a], tmp, {"RTC_VERBOSE": "1"})
    assert "sketch+d2h" not in err and "[gpu 0.0]" in err
    assert not [d for d in os.listdir(tmp) if os.path.isdir(os.path.join(tmp, d))]
    _cli([os.path.join(BIN, "clust-mst"), "-l", "-i", lst, "-k", "21", "-s", "500", "-d", "0.05", "-o", b], tmp)
    assert _partition(_parse_clusters(a)) == _partition(_parse_clusters(b))
    assert open(a).read() == open(b).read()


def _rccl_path_for_bare_binary():
    """where a process WITHOUT torch finds RCCL: the ROCm tree, or the wheel's copy named explicitly"""
    for p in ("/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"):
        if os.path.exists(p):
            return None  # the loader's own search reaches it
    import torch
    p = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    return p if os.path.exists(p) else None


@pytest.mark.parametrize("fast", [False, True])
def test_clust_mst_bare_binary_drives_rccl(oracle, tmp_path, fast):
    """RTC_COMM_FORCE_RCCL=1 clust-mst --gpus 1 in a subprocess that never imports torch: the binary's own dlopen
    of librccl, ncclCommInitAll, the share step's in-place ncclBroadcast and rtc_mst_sharded's ncclAllReduce per
    Boruvka round -- the calls of the multi-GPU path, with one rank.  Output byte-identical to the plain run."""
    from test_gpu_cli import _write_family_fastas
    tmp = str(tmp_path)
    lst, paths, seqs = _write_family_fastas(oracle, tmp, 4, 3, 1_500_000, seed=31)
    outs = {}
    explicit = _rccl_path_for_bare_binary()
    for tag, env in (("plain", {}), ("rccl", dict({"RTC_COMM_FORCE_RCCL": "1"}, **({"RTC_RCCL_LIB": explicit} if explicit else {})))):
        d = os.path.join(tmp, tag)
        os.makedirs(d)
        out = os.path.join(d, "res.out")
        cmd = [os.path.join(BIN, "clust-mst"), "-l", "-i", lst, "-k", "21", "-d", "0.05", "-t", "4", "--gpus", "1", "-o", out]
        cmd += ["--fast"] if fast else ["-s", "500"]
        err = _cli(cmd, d, dict(env, RTC_BATCH_BYTES=str(6 << 20), RTC_VERBOSE="1"))
        if tag == "rccl":
            assert "use 1 GPUs (rccl exchange)" in err and "[comm]  RCCL from " in err and "[share]" in err and "[mst gpu 0]" in err, err[-2000:]
        folder = [os.path.join(d, x) for x in os.listdir(d) if os.path.isdir(os.path.join(d, x))][0]
        files = sorted(f for f in os.listdir(folder) if "info" not in f)
        outs[tag] = (open(out).read(), {f: open(os.path.join(folder, f), "rb").read() for f in files})
    assert outs["plain"] == outs["rccl"]


def test_clust_mst_without_rccl_falls_back_to_one_gpu(oracle, tmp_path):
    """The default GPU choice ("all") survives a missing RCCL: a warning, then the whole job on one GPU with the same
    output.  A user who NAMED the GPUs gets the error instead."""
    from test_gpu_cli import _write_family_fastas
    tmp = str(tmp_path)
    lst, paths, seqs = _write_family_fastas(oracle, tmp, 3, 3, 1_200_000, seed=32)
    base = [os.path.join(BIN, "clust-mst"), "-l", "-i", lst, "-k", "21", "-s", "400", "-d", "0.05", "-t", "4", "-e"]
    broken = {"RTC_COMM_FORCE_RCCL": "1", "RTC_RCCL_LIB": os.path.join(tmp, "no-such-librccl.so")}
    a, b = os.path.join(tmp, "a.out"), os.path.join(tmp, "b.out")
    _cli(base + ["-o", a], tmp)
    err = _cli(base + ["-o", b], tmp, broken)
    assert "Warning: no communicator" in err and "running on GPU 0 alone" in err
    assert open(a).read() == open(b).read()
    r = subprocess.run(base + ["--gpus", "1", "-o", os.path.join(tmp, "c.out")], cwd=tmp, capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, **broken))
    assert r.returncode == 1 and "rtc_comm_init_all failed" in r.stderr and "no-such-librccl.so" in r.stderr


def test_clust_greedy_sketches_on_every_gpu_and_clusters_on_one(oracle, tmp_path):
    """clust-greedy from genome files: the sketch phase uses every GPU of the choice (--gpus 0,0: two lanes' contexts and
    the share step), the clustering runs on the first; from a sketch folder there is nothing to spread: one context, no
    communicator.  Same clusters every way."""
    from test_gpu_cli import _write_family_fastas
    tmp = str(tmp_path)
    lst, paths, seqs = _write_family_fastas(oracle, tmp, 3, 3, 1_200_000, seed=33)
    a, b, c = os.path.join(tmp, "a.out"), os.path.join(tmp, "b.out"), os.path.join(tmp, "c.out")
    base = [os.path.join(BIN, "clust-greedy"), "-l", "-i", lst, "-k", "21", "-s", "400", "-d", "0.05", "-t", "4"]
    _cli(base + ["-e", "--gpus", "1", "-o", a], tmp)
    err = _cli(base + ["--gpus", "0,0", "-o", b], tmp, {"RTC_VERBOSE": "1"})
    assert "use 2 GPUs" in err and "2 GPU context(s)" in err and "[share]" in err
    assert open(a).read() == open(b).read()
    folder = [d for d in os.listdir(tmp) if os.path.isdir(os.path.join(tmp, d)) and os.path.exists(os.path.join(tmp, d, "hash.sketch"))]
    assert len(folder) == 1
    err = _cli([os.path.join(BIN, "clust-greedy"), "--presketched", os.path.join(tmp, folder[0]), "-d", "0.05", "-t", "4", "--gpus", "0,0", "-o", c],
               tmp, {"RTC_VERBOSE": "1"})
    assert "this flow runs on one GPU" in err and "exchange)" not in err and "1 GPU context(s)" in err
    part = lambda f: sorted(tuple(sorted(ln.split("\t")[-1] for ln in blk.strip().split("\n")[1:])) for blk in open(f).read().split("the cluster")[1:])
    assert part(a) == part(c)


@pytest.mark.parametrize("mode", ["minhash-c", "kssd"])
def test_resident_rows_over_budget_fall_back_to_host_vectors(oracle, tmp_path, mode):
    """the resident sketch buffer (files x largest row) is sized against the free HBM; over the budget the run
    keeps per-batch temporaries and clusters from the host vectors -- same files, same clusters, no abort."""
    from test_gpu_cli import _write_family_fastas
    tmp = str(tmp_path)
    lst, paths, seqs = _write_family_fastas(oracle, tmp, 3, 3, 1_500_000, seed=34)
    outs = {}
    for tag, env in (("resident", {}), ("budget", {"RTC_RESIDENT_BUDGET": "4096"})):
        d = os.path.join(tmp, tag)
        os.makedirs(d)
        out = os.path.join(d, "res.out")
        cmd = [os.path.join(BIN, "clust-mst"), "-l", "-i", lst, "-k", "21", "-d", "0.05", "-t", "4", "--gpus", "1", "-o", out]
        cmd += ["--fast"] if mode == "kssd" else ["-c", "1000"]
        err = _cli(cmd, d, dict(env, RTC_BATCH_BYTES=str(6 << 20)))
        assert ("sketches go through host memory instead" in err) == (tag == "budget")
        folder = [os.path.join(d, x) for x in os.listdir(d) if os.path.isdir(os.path.join(d, x))][0]
        files = sorted(f for f in os.listdir(folder) if "info" not in f)
        outs[tag] = (open(out).read(), {f: open(os.path.join(folder, f), "rb").read() for f in files})
    assert outs["resident"] == outs["budget"]


def test_collective_watchdog_fails_fast_when_a_rank_skips_a_round(oracle):
    """A rank that never arrives (it took another branch, failed before the call) must not leave the others inside the
    collective: with RTC_COMM_TIMEOUT_S=1 the waiting rank returns RTC_ERR_COMM after about a second, and the
    communicator group stays broken -- the late rank's own next collective fails at once instead of hanging."""
    import time
    import torch
    from rabbittclust_amd import _lib, api
    ctxs = [api.Context(0) for _ in range(2)]
    comms = api.Comm.init_all(ctxs)
    os.environ["RTC_COMM_TIMEOUT_S"] = "1"
    _reload_options()
    try:
        def rank0():
            t = torch.arange(8, dtype=torch.int64, device=ctxs[0].device)
            comms[0].all_reduce(t, "min")          # round 1: both ranks
            t0 = time.time()
            try:
                comms[0].all_reduce(t, "min")      # round 2: rank 1 never comes
            except _lib.RtcError as e:
                return e.status, time.time() - t0
            return 0, time.time() - t0

        def rank1():
            t = torch.arange(8, dtype=torch.int64, device=ctxs[1].device) + 5
            comms[1].all_reduce(t, "min")          # round 1
            time.sleep(2.5)                        # "skips" round 2, comes back later for another collective
            t0 = time.time()
            try:
                comms[1].all_reduce_host([1, 2], "max")
            except _lib.RtcError as e:
                return e.status, time.time() - t0
            return 0, time.time() - t0
        (s0, dt0), (s1, dt1) = _threads([rank0, rank1])
        assert s0 == _lib.RTC_ERR_COMM and 0.8 < dt0 < 2.4, (s0, dt0)
        assert s1 == _lib.RTC_ERR_COMM and dt1 < 0.5, (s1, dt1)
        assert "RTC_COMM_TIMEOUT_S" in ctxs[0].lib.rtc_last_error(ctxs[0].h).decode()
    finally:
        os.environ.pop("RTC_COMM_TIMEOUT_S", None)
        _reload_options()
        for c in comms:
            c.close()

"""GPU parity: all-pairs intersection counts vs the CPU oracle (exact integers)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _make_sketches(rng, n, smin, smax, pool_bits=20, dtype=np.uint64):
    """Random sorted distinct sets drawn from a small pool so intersections are frequent."""
    out = []
    for _ in range(n):
        s = int(rng.integers(smin, smax + 1))
        v = np.unique(rng.integers(0, 1 << pool_bits, size=s * 2, dtype=np.uint64))[:s]
        # spread into the full range but keep collisions between genomes
        v = (v * np.uint64(0x9E3779B97F4A7C15)) if dtype == np.uint64 else v
        out.append(np.sort(v.astype(dtype)))
    return out


def _oracle_matrix(oracle, sk):
    n = len(sk)
    m = np.zeros((n, n), dtype=np.int64)
    for i in range(n):
        for j in range(n):
            m[i, j] = oracle.common(sk[i], sk[j])
    return m


@pytest.mark.parametrize("algo", [1, 0])
@pytest.mark.parametrize("width", [8, 4])
def test_pair_common_matches_oracle(ctx, oracle, algo, width):
    from rabbittclust_amd import api
    rng = np.random.default_rng(width * 10 + algo)
    dt = np.uint64 if width == 8 else np.uint32
    sk = _make_sketches(rng, 70, 0, 300, pool_bits=12, dtype=dt)
    sk[3] = np.zeros(0, dtype=dt)
    sk[5] = sk[4].copy()
    dev = api.SketchSet.from_host(sk, ctx.device, width=width)
    got = ctx.pair_common(dev, algo=algo).cpu().numpy()
    want = _oracle_matrix(oracle, sk)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("algo", [1, 0])
def test_pair_common_subtile_and_lower(ctx, oracle, algo):
    from rabbittclust_amd import api
    rng = np.random.default_rng(3)
    sk = _make_sketches(rng, 150, 50, 120, pool_bits=11)
    dev = api.SketchSet.from_host(sk, ctx.device)
    want = _oracle_matrix(oracle, sk)
    got = ctx.pair_common(dev, row0=37, row1=131, col0=5, col1=150, algo=algo).cpu().numpy()
    assert np.array_equal(got, want[37:131, 5:150])
    low = ctx.pair_common(dev, lower_only=True, algo=algo).cpu().numpy()
    il = np.tril_indices(150, -1)
    assert np.array_equal(low[il], want[il])


def test_pair_common_on_real_sketches(ctx, oracle):
    from rabbittclust_amd import api
    desc = api.synth_family_descs(6, 5, global_seed=11)
    L = 120_000
    off = np.arange(len(desc) + 1, dtype=np.uint64) * L
    seq = ctx.synth_genomes(desc, off)
    dev = ctx.sketch_minhash(seq, off, k=21, size=1000)
    host = dev.to_host()
    want = _oracle_matrix(oracle, host)
    for algo in (1, 0):
        got = ctx.pair_common(dev, algo=algo).cpu().numpy()
        assert np.array_equal(got, want), algo
    assert want[1, 0] > 100 and want[5, 0] == 0  # families share hashes, strangers do not

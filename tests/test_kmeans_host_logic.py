"""Host logic of audio_tokens_amd.ops.Kmeans / IndexFlatL2 (the FAISS training recipe and its
data-parallel variant) driven with the CPU stand-in backend, against the oracle and the golden
vectors.  Includes the world_size-2 gloo run of the sharded path."""
import os
import socket
import warnings
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

G = Path(__file__).resolve().parent / "golden"


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.fixture()
def cpu_be():
    from oracle_backend import OracleBackend
    return OracleBackend()


def test_train_matches_golden_plain_and_split(cpu_be):
    from audio_tokens_amd.ops import Kmeans
    g = np.load(G / "kmeans.npz")
    km = Kmeans(64, 64, niter=20, backend=cpu_be)
    obj = km.train(g["a_x"])
    assert np.array_equal(bits(km.centroids), bits(g["a_centroids"]))
    assert [s["nsplit"] for s in km.iteration_stats] == list(g["a_nsplit"])
    np.testing.assert_allclose(km.obj, g["a_obj"], rtol=2e-6)
    np.testing.assert_allclose([s["imbalance_factor"] for s in km.iteration_stats], g["a_imbalance"], rtol=1e-12)
    assert obj == pytest.approx(float(g["a_obj"][-1]), rel=2e-6)
    assert km.index.ntotal == 64
    kb = Kmeans(64, 48, niter=6, backend=cpu_be)
    kb.train(g["b_x"])
    assert np.array_equal(bits(kb.centroids), bits(g["b_centroids"]))
    assert [s["nsplit"] for s in kb.iteration_stats] == list(g["b_nsplit"])


def test_train_subsample_and_warm_start(cpu_be):
    from audio_tokens_amd.ops import Kmeans
    g = np.load(G / "kmeans.npz")
    km = Kmeans(8, 64, niter=5, backend=cpu_be)
    km.train(g["c_x"], init_centroids=g["c_init"])
    assert np.array_equal(bits(km.centroids), bits(g["c_centroids"]))
    # the reference's loop: train(batch0); train(batch1, init_centroids=kmeans.centroids)
    km2 = Kmeans(8, 64, niter=5, backend=cpu_be)
    km2.train(g["c_x"][:9000])
    first = km2.centroids.copy()
    km2.train(g["c_x"][9000:], init_centroids=km2.centroids)
    import oracle
    r1 = oracle.kmeans_train(g["c_x"][:9000], 64, niter=5)
    r2 = oracle.kmeans_train(g["c_x"][9000:], 64, niter=5, init_centroids=r1.centroids)
    assert np.array_equal(bits(first), bits(r1.centroids)) and np.array_equal(bits(km2.centroids), bits(r2.centroids))


def test_train_error_behaviour(cpu_be):
    from audio_tokens_amd.ops import Kmeans
    x = np.random.default_rng(0).standard_normal((10, 4)).astype(np.float32)
    with pytest.raises(RuntimeError, match="should be at least as large as number of clusters"):
        Kmeans(4, 16, backend=cpu_be).train(x)
    bad = x.copy(); bad[2, 1] = np.inf
    with pytest.raises(RuntimeError, match="NaN's or Inf's"):
        Kmeans(4, 4, backend=cpu_be).train(bad)
    km = Kmeans(4, 10, niter=3, backend=cpu_be)       # n == k: points become centroids
    km.train(x)
    assert np.array_equal(km.centroids, x) and km.iteration_stats[0]["nsplit"] == 0
    with pytest.raises(NotImplementedError):
        Kmeans(4, 2, spherical=True, backend=cpu_be)


def test_index_flat_l2(cpu_be):
    from audio_tokens_amd.ops import IndexFlatL2
    g = np.load(G / "tokenizer.npz")
    index = IndexFlatL2(64, backend=cpu_be)
    assert index.ntotal == 0
    index.add(g["c"][:100]); index.add(g["c"][100:])
    assert index.ntotal == 256
    D, I = index.search(g["x"], 1)
    assert D.shape == (4000, 1) and I.shape == (4000, 1) and I.dtype == np.int64 and D.dtype == np.float32
    assert np.array_equal(I[:, 0], g["ids"]) and np.array_equal(bits(D[:, 0]), bits(g["dis"]))
    with pytest.raises(NotImplementedError):
        index.search(g["x"], 5)
    index.reset()
    assert index.ntotal == 0


# ---- world_size 2 under gloo ------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _cuts(n, world):
    """Uneven contiguous blocks of n rows for `world` ranks (rank order = row order, as the sharded k-means assumes)."""
    w = np.arange(1, world + 1, dtype=np.float64) ** 0.5 + 0.37
    edges = np.concatenate([[0], np.floor(np.cumsum(w / w.sum()) * n).astype(np.int64)])
    edges[-1] = n
    return edges


def _worker(rank, world, port, case, q, exchange="auto"):
    import sys
    sys.path.insert(0, str(Path(__file__).resolve().parent))
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle_backend import OracleBackend
        from audio_tokens_amd.ops import Kmeans
        g = np.load(G / "kmeans.npz")
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            if case == "plain":
                x, cut = g["a_x"], 1000
                local = x[:cut] if rank == 0 else x[cut:]
                km = Kmeans(64, 64, niter=20, distributed=True, backend=OracleBackend())
                km.exchange = exchange
                km.train(local)
            elif case == "many":   # any world size: uneven blocks of the subsampled case, cold start then warm start
                x = g["c_x"]
                e = _cuts(len(x), world)
                local = x[e[rank]:e[rank + 1]]
                km = Kmeans(8, 64, niter=5, distributed=True, backend=OracleBackend())
                km.exchange = exchange
                km.train(local, init_centroids=g["c_init"])
                km.train(local[: len(local) * 3 // 4], init_centroids=km.centroids)
            else:  # subsampled: 20000 rows, k=64 -> 16384 rows kept, spread over both ranks
                x, cut = g["c_x"], 12345
                local = x[:cut] if rank == 0 else x[cut:]
                km = Kmeans(8, 64, niter=5, distributed=True, backend=OracleBackend())
                km.exchange = exchange
                km.train(local, init_centroids=g["c_init"])
        q.put((rank, km.centroids.copy(), [s["nsplit"] for s in km.iteration_stats], km.obj.copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case,exchange", [("plain", "auto"), ("subsampled", "auto"), ("plain", "scatter"), ("subsampled", "scatter")])
def test_sharded_kmeans_gloo_world2(case, exchange, oracle):
    """Both forms of the per-iteration exchange (all-gather of whole partials; all-to-all + ordered local sum +
    all-gather) give the oracle's two-shard bits on both ranks."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, case, q, exchange)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    g = np.load(G / "kmeans.npz")
    assert np.array_equal(bits(res[0][1]), bits(res[1][1])), "ranks disagree"
    if case == "plain":
        assert np.array_equal(bits(res[0][1]), bits(g["d_centroids"]))   # oracle's two-shard variant
        np.testing.assert_allclose(res[0][3], g["d_obj"], rtol=2e-6)
    else:
        shard = (np.arange(20000) >= 12345).astype(np.int32)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            r = oracle.kmeans_train(g["c_x"], 64, niter=5, init_centroids=g["c_init"], shard=shard, n_shards=2)
        assert np.array_equal(bits(res[0][1]), bits(r.centroids))


@pytest.mark.parametrize("world,exchange", [(4, "gather"), (4, "scatter"), (8, "gather"), (8, "scatter")])
def test_sharded_kmeans_gloo_world4_and_8(world, exchange, oracle):
    """The N-GPU form at N = 4 and 8: uneven row blocks, the subsample permutation over the global
    row index, a cold-started and a warm-started training -- every rank ends with the bits of the oracle's n_shards mode
    (partials added in ascending shard order), in both forms of the exchange."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, "many", q, exchange)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    g = np.load(G / "kmeans.npz")
    x = g["c_x"]
    e = _cuts(len(x), world)
    shard = (np.searchsorted(e, np.arange(len(x)), side="right") - 1).astype(np.int32)
    keep = np.concatenate([np.arange(e[r], e[r] + (e[r + 1] - e[r]) * 3 // 4) for r in range(world)])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        r1 = oracle.kmeans_train(x, 64, niter=5, init_centroids=g["c_init"], shard=shard, n_shards=world)
        r2 = oracle.kmeans_train(x[keep], 64, niter=5, init_centroids=r1.centroids, shard=shard[keep], n_shards=world)
    for rank, cent, nsplit, obj in res:
        assert np.array_equal(bits(cent), bits(r2.centroids)), f"rank {rank} differs from the oracle's {world}-shard result"
        assert nsplit == list(r2.nsplit)
        np.testing.assert_allclose(obj, r2.obj, rtol=2e-5)      # (the objective is a statistic summed in another order: DESIGN.md section 2)


def test_prefetch_preserves_order_and_errors():
    from audio_tokens_amd.utils.prefetch import prefetch
    assert list(prefetch(iter(range(7)), depth=2)) == list(range(7))
    assert list(prefetch(iter(()))) == []

    def boom():
        yield 1
        raise ValueError("bad file")

    it = prefetch(boom())
    assert next(it) == 1
    with pytest.raises(ValueError, match="bad file"):
        next(it)

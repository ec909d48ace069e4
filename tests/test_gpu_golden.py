"""GPU parity against the committed golden vectors (no oracle needed at run time) and the
faiss-shaped Python operators on the real HIP backend."""
import warnings
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = Path(__file__).resolve().parent / "golden"


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def test_tokenizer_golden(be):
    from audio_tokens_amd.ops import IndexFlatL2
    g = np.load(G / "tokenizer.npz")
    index = IndexFlatL2(64)
    index.add(g["c"])
    D, I = index.search(g["x"], 1)
    assert I.dtype == np.int64 and D.dtype == np.float32 and I.shape == (4000, 1)
    assert np.array_equal(I[:, 0], g["ids"]) and np.array_equal(bits(D[:, 0]), bits(g["dis"]))
    D, I = index.search(g["x"][:7], 1)                       # n < 20: faiss' direct form
    assert np.array_equal(I[:, 0], g["ids_small"]) and np.array_equal(bits(D[:, 0]), bits(g["dis_small"]))
    # device tensors in -> device tensors out
    Dt, It = index.search(torch.from_numpy(g["x"]).cuda(), 1)
    assert It.is_cuda and np.array_equal(It.cpu().numpy()[:, 0], g["ids"])


def test_kmeans_golden_all_cases(be):
    from audio_tokens_amd.ops import Kmeans
    g = np.load(G / "kmeans.npz")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        km = Kmeans(64, 64, niter=20)
        km.train(g["a_x"])
        assert np.array_equal(bits(km.centroids), bits(g["a_centroids"]))
        assert [s["nsplit"] for s in km.iteration_stats] == list(g["a_nsplit"])
        np.testing.assert_allclose(km.obj, g["a_obj"], rtol=2e-6)
        assert np.array_equal(km._last_assign.cpu().numpy(), g["a_assign"])
        kb = Kmeans(64, 48, niter=6)
        kb.train(g["b_x"])                                   # empty clusters -> split_clusters every iteration
        assert np.array_equal(bits(kb.centroids), bits(g["b_centroids"]))
        assert [s["nsplit"] for s in kb.iteration_stats] == list(g["b_nsplit"])
        kc = Kmeans(8, 64, niter=5)
        kc.train(g["c_x"], init_centroids=g["c_init"])       # mt19937 subsample, generic-d kernels
        assert np.array_equal(bits(kc.centroids), bits(g["c_centroids"]))


def test_logmel_golden(be):
    g = np.load(G / "logmel.npz")
    for nm in (64, 128):
        got = be.logmel(g["wave"], n_mels=nm).cpu().numpy()
        ref = g[f"logmel_{nm}"]
        P, Pr = 10.0 ** (got.astype(np.float64) / 10), 10.0 ** (ref.astype(np.float64) / 10)
        tol = 2e-5 * Pr + 1e-9 * Pr.max(axis=-2, keepdims=True) + 1e-14
        assert (np.abs(P - Pr) <= tol).all()
        # a user-supplied filterbank (e.g. torchaudio's own) gives the same answer as the built-in
        got2 = be.logmel(g["wave"], n_mels=nm, fb=g[f"fb_{nm}"]).cpu().numpy()
        assert np.array_equal(bits(got2), bits(got))


def test_logmel_spectrogram_operator(be):
    from audio_tokens_amd.ops import LogMelSpectrogram
    g = np.load(G / "logmel.npz")
    op = LogMelSpectrogram(sample_rate=22050, n_mels=64, n_fft=512, hop_length=128)
    one = op(torch.from_numpy(g["wave"][:1]))                # [1, L] like the reference's call
    assert tuple(one.shape) == (1, 64, 173)
    allc = op(g["wave"])
    assert np.array_equal(bits(allc[0].cpu().numpy()), bits(one[0].cpu().numpy()))


def test_kmeans_pruned_path_matches_oracle(be, oracle, switches):
    """A train() large enough to take the pruned / coarse-to-fine path (k >= 1024), cold start and
    warm start, against the oracle; and the same with pruning switched off."""
    from audio_tokens_amd.ops import Kmeans
    switches(filter_stats=1)                                 # (the sweeps count what they skip only when asked to)
    be.prune_stats(reset=True)
    rng = np.random.default_rng(9)
    cen = rng.standard_normal((2048, 64))
    x = (cen[rng.integers(0, 2048, 60000)] + 0.4 * rng.standard_normal((60000, 64))).astype(np.float32)
    x = oracle.l2norm_rows(x)
    x[:300] = x[300:600]                                    # duplicates
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        r1 = oracle.kmeans_train(x[:40000], 2048, niter=8)
        r2 = oracle.kmeans_train(x[40000:], 2048, niter=8, init_centroids=r1.centroids)
        # tiles per wave of the Lloyd sweep: the default at this size (one), and two / four forced
        for prune, nb in ((True, 0), (True, 2), (True, 4), (False, 0)):
            switches(filter_nb=nb)
            km = Kmeans(64, 2048, niter=8)
            km.prune = prune
            km.train(x[:40000])
            assert np.array_equal(bits(km.centroids), bits(r1.centroids)), f"prune={prune} nb={nb} cold"
            assert [s["nsplit"] for s in km.iteration_stats] == list(r1.nsplit)
            km.train(x[40000:], init_centroids=km.centroids)
            assert np.array_equal(bits(km.centroids), bits(r2.centroids)), f"prune={prune} nb={nb} warm"
    a, t = be.prune_stats()
    assert 0 < a < t                                         # something was really skipped


def test_kmeans_visiting_order_path_matches_oracle(be, oracle):
    """Enough rows per cluster (>= 96) for the (cluster, distance) visiting order -- its sort runs on a stream of
    its own beside the accumulations -- cold start and warm start against the oracle."""
    from audio_tokens_amd.ops import Kmeans
    rng = np.random.default_rng(19)
    cen = rng.standard_normal((1024, 64))
    x = (cen[rng.integers(0, 1024, 230000)] + 0.5 * rng.standard_normal((230000, 64))).astype(np.float32)
    x = oracle.l2norm_rows(x)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        r1 = oracle.kmeans_train(x[:120000], 1024, niter=6)
        r2 = oracle.kmeans_train(x[120000:], 1024, niter=6, init_centroids=r1.centroids)
        km = Kmeans(64, 1024, niter=6)
        km.train(x[:120000])
        assert np.array_equal(bits(km.centroids), bits(r1.centroids))
        # (the objective is reported from a fixed fp64 summation tree, the oracle adds in faiss' fp32 order)
        assert np.allclose(km.obj, r1.obj, rtol=2e-5, atol=0)
        assert [s["nsplit"] for s in km.iteration_stats] == list(r1.nsplit)
        km.train(x[120000:], init_centroids=km.centroids)
        assert np.array_equal(bits(km.centroids), bits(r2.centroids))
        assert np.allclose(km.obj, r2.obj, rtol=2e-5, atol=0)


def test_index_flat_l2_large_search_uses_exact_pruning(be, oracle):
    from audio_tokens_amd.ops import IndexFlatL2
    rng = np.random.default_rng(10)
    c = oracle.l2norm_rows(rng.standard_normal((4096, 64)).astype(np.float32))
    x = oracle.l2norm_rows((c[rng.integers(0, 4096, 100000)] + 0.05 * rng.standard_normal((100000, 64))).astype(np.float32))
    ids_o, dis_o = oracle.assign(x, c)
    index = IndexFlatL2(64)
    index.add(c)
    D, I = index.search(x, 1)
    assert index._prune is not None
    assert np.array_equal(I[:, 0], ids_o) and np.array_equal(bits(D[:, 0]), bits(dis_o))
    index.prune = False
    D2, I2 = index.search(x, 1)
    assert np.array_equal(I2, I) and np.array_equal(bits(D2), bits(D))


def test_full_size_properties(be):
    """BASELINE-size launch (2 097 152 x 64 vs 8192): size-independent checks."""
    gen = torch.Generator(device="cuda").manual_seed(1)
    n, d, k = 2097152, 64, 8192
    x = torch.nn.functional.normalize(torch.randn(n, d, device="cuda", generator=gen), dim=1)
    c = torch.nn.functional.normalize(torch.randn(k, d, device="cuda", generator=gen), dim=1)
    ids, dis = be.assign(x, c)
    assert int(ids.min()) >= 0 and int(ids.max()) < k
    # the reported distance is the distance to the reported centroid (recomputed in fp64)
    sel = torch.randint(0, n, (20000,), device="cuda", generator=gen)
    d64 = ((x[sel].double() - c[ids[sel]].double()) ** 2).sum(1)
    assert float((dis[sel].double() - d64).abs().max()) < 5e-6
    # no other centroid is closer (dense fp64 check on a sample)
    sub = sel[:2000]
    full = torch.cdist(x[sub].double(), c.double()) ** 2
    assert float((d64[:2000] - full.min(1).values).max()) < 5e-6
    # permuting the rows permutes the answer; a row that is a centroid maps to it with distance ~0
    perm = torch.randperm(n, device="cuda", generator=gen)
    ids_p, _ = be.assign(x[perm].contiguous(), c)
    assert torch.equal(ids_p, ids[perm])
    ids_c, dis_c = be.assign(c, c)
    assert torch.equal(ids_c, torch.arange(k, device="cuda")) and float(dis_c.max()) < 1e-6
    # centroid sums: totals are conserved and counts add up
    part = be.centroid_accum(x, ids, k)
    counts = part[k * d:]
    assert float(counts.sum()) == n
    tot = part[: k * d].view(k, d).double().sum(0)
    assert float((tot - x.double().sum(0)).abs().max()) < 1e-2


def test_logmel_user_filterbanks_are_compared_by_value(be, oracle):
    """two different filterbanks of the same shape back to back -- the second one lands at the
    address the allocator just freed -- and one rewritten in place must each produce their own spectrogram."""
    g = np.load(G / "logmel.npz")
    wave = torch.from_numpy(g["wave"]).to(be.device)
    fb_a = g["fb_64"].copy()
    fb_b = fb_a[:, ::-1].copy()                       # same shape, different values
    def run(fb):
        t = torch.from_numpy(fb).to(be.device)        # a fresh temporary per call, as backend.logmel does for numpy
        out = be.logmel(wave, n_mels=64, fb=t)
        del t
        return out

    a1, b1, a2 = run(fb_a), run(fb_b), run(fb_a)
    assert torch.equal(a1, a2) and not torch.equal(a1, b1)
    assert torch.equal(b1, a1.flip(1))                # mel axis reversed with the filterbank
    t = torch.from_numpy(fb_a).to(be.device)
    x1 = be.logmel(wave, n_mels=64, fb=t)
    t.copy_(torch.from_numpy(fb_b))                   # rewritten in place: same address, new values
    x2 = be.logmel(wave, n_mels=64, fb=t)
    assert torch.equal(x1, a1) and torch.equal(x2, b1)

"""The oracle against the committed golden vectors (tests/golden/*.npz, produced by
tests/golden/make_golden.py) and against the independent implementations that exist in the
container.  CPU only.  Parity status of the oracle itself: UNPINNED (no reference fixtures exist;
see oracle/oracle.c) -- these tests pin it against drift and against torch / numpy / sklearn /
libstdc++ / transformers where those overlap with it."""
import warnings
from pathlib import Path

import numpy as np
import pytest
import torch

G = Path(__file__).resolve().parent / "golden"


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


# ---- golden ---------------------------------------------------------------------------------
def test_golden_rng(oracle):
    g = np.load(G / "rng.npz")
    assert np.array_equal(oracle.rand_perm(1000, 1234), g["perm_1000_1234"])
    assert np.array_equal(oracle.rand_perm(50000, 1235)[:512], g["perm_50000_1235_head"])
    assert np.array_equal(oracle.mt19937_raw(1234, 64), g["raw_1234"])


def test_golden_logmel(oracle):
    g = np.load(G / "logmel.npz")
    for nm in (64, 128):
        got = np.stack([oracle.logmel(w, n_mels=nm) for w in g["wave"]])
        assert np.array_equal(bits(got), bits(g[f"logmel_{nm}"]))
        assert np.array_equal(oracle.mel_filterbank(22050, 512, nm), g[f"fb_{nm}"])


def test_golden_tokenizer(oracle):
    g = np.load(G / "tokenizer.npz")
    for ref in (False, True):
        ids, dis = oracle.assign(g["x"], g["c"], ref=ref)
        assert np.array_equal(ids, g["ids"]) and np.array_equal(bits(dis), bits(g["dis"]))
    ids, dis = oracle.assign(g["x"][:7], g["c"])
    assert np.array_equal(ids, g["ids_small"]) and np.array_equal(bits(dis), bits(g["dis_small"]))
    assert (g["ids"][100:110] == np.arange(10)).all()      # duplicate centroids: lowest index wins
    assert (g["dis"][100:110] == 0).all()


def test_golden_kmeans(oracle):
    g = np.load(G / "kmeans.npz")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        r = oracle.kmeans_train(g["a_x"], 64, niter=20)
        assert np.array_equal(bits(r.centroids), bits(g["a_centroids"])) and np.array_equal(r.assign, g["a_assign"])
        assert np.array_equal(r.obj, g["a_obj"]) and np.array_equal(r.nsplit, g["a_nsplit"])
        rb = oracle.kmeans_train(g["b_x"], 48, niter=6)
        assert np.array_equal(bits(rb.centroids), bits(g["b_centroids"])) and np.array_equal(rb.nsplit, g["b_nsplit"])
        assert g["b_nsplit"].sum() > 0
        rc = oracle.kmeans_train(g["c_x"], 64, niter=5, init_centroids=g["c_init"])
        assert np.array_equal(bits(rc.centroids), bits(g["c_centroids"])) and np.array_equal(rc.sub_perm, g["c_sub_perm"])
        rd = oracle.kmeans_train(g["a_x"], 64, niter=20, shard=g["d_shard"], n_shards=2)
        assert np.array_equal(bits(rd.centroids), bits(g["d_centroids"]))


# ---- independent pins -----------------------------------------------------------------------
def test_mt19937_is_libstdcxx_mt19937(oracle):
    assert np.array_equal(oracle.mt19937_raw(1234, 5000), oracle.std_mt19937_raw(1234, 5000))
    assert np.array_equal(oracle.mt19937_raw(5489, 10000)[-1:], np.array([4123659995], np.uint32))  # the C++ standard's check value
    for n, s in [(1, 1234), (2, 1234), (1000, 1234), (100003, 1235)]:
        assert np.array_equal(oracle.rand_perm(n, s), oracle.std_rand_perm(n, s))
    p = oracle.rand_perm(100003, 1234)
    assert np.array_equal(np.sort(p), np.arange(100003))


@pytest.mark.parametrize("d", [3, 8, 20, 64, 128, 130, 640])
def test_l2norm_rows_is_numpy_bit_for_bit(oracle, d):
    rng = np.random.default_rng(d)
    x = (rng.standard_normal((500, d)) * rng.uniform(0.01, 100, (500, 1))).astype(np.float32)
    x[5] = 0
    ref = x / (np.linalg.norm(x, axis=1, keepdims=True) + 1e-10)   # cluster_creator.py:64-66
    assert ref.dtype == np.float32
    assert np.array_equal(bits(oracle.l2norm_rows(x)), bits(ref))


@pytest.mark.parametrize("n_mels", [64, 128])
def test_filterbank_vs_transformers(oracle, n_mels):
    from transformers.audio_utils import mel_filter_bank
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = mel_filter_bank(num_frequency_bins=257, num_mel_filters=n_mels, min_frequency=0.0,
                              max_frequency=11025.0, sampling_rate=22050, norm=None, mel_scale="htk")
    fb = oracle.mel_filterbank(22050, 512, n_mels)
    assert np.abs(fb - ref).max() < 1e-6
    assert ((fb > 0) == (ref > 0)).all()


def _torch_logmel(w, fb):
    """torchaudio's MelSpectrogram + AmplitudeToDB written with torch ops (fp32 pipeline)."""
    S = torch.stft(torch.from_numpy(w), 512, 128, 512, torch.hann_window(512), center=True,
                   pad_mode="reflect", return_complex=True)
    mel = torch.matmul(S.abs().pow(2.0).transpose(-1, -2), torch.from_numpy(fb)).transpose(-1, -2)
    return (10 * torch.log10(torch.clamp(mel, min=1e-10))).numpy()


def test_logmel_vs_torch_stft_pipeline(oracle):
    g = np.load(G / "logmel.npz")
    for nm in (64, 128):
        fb = g[f"fb_{nm}"]
        for w, ref in zip(g["wave"], g[f"logmel_{nm}"]):
            t = _torch_logmel(w, fb)
            assert t.shape == ref.shape == (nm, 173)
            P, Pr = 10.0 ** (t.astype(np.float64) / 10), 10.0 ** (ref.astype(np.float64) / 10)
            tol = 2e-5 * Pr + 1e-9 * Pr.max(axis=0, keepdims=True) + 1e-14
            assert (np.abs(P - Pr) <= tol).all()          # same tolerance the GPU kernel is held to
    # a silent clip is exactly -100 dB in both
    z = np.zeros(22050, np.float32)
    assert (oracle.logmel(z) == -100.0).all() and (_torch_logmel(z, g["fb_64"]) == -100.0).all()


def test_assign_vs_float64_brute_force(oracle):
    g = np.load(G / "tokenizer.npz")
    x, c = g["x"].astype(np.float64), g["c"].astype(np.float64)
    d2 = ((x[:, None, :] - c[None]) ** 2).sum(-1)
    best = d2.min(1)
    picked = d2[np.arange(len(x)), g["ids"]]
    assert (picked - best <= 1e-6).all()                  # the fp32 winner is within rounding of the true one
    assert np.abs(g["dis"] - picked).max() < 2e-6


def test_lloyd_vs_sklearn(oracle):
    from sklearn.cluster import KMeans
    rng = np.random.default_rng(2)
    k, d, n = 32, 64, 6000
    centers = rng.standard_normal((k, d)) * 3
    x = oracle.l2norm_rows((centers[rng.integers(0, k, n)] + rng.standard_normal((n, d))).astype(np.float32))
    r = oracle.kmeans_train(x, k, niter=20)
    assert r.nsplit.sum() == 0                              # no empty cluster: plain Lloyd
    init = x[oracle.rand_perm(n, 1235)[:k]]                 # faiss: rand_perm(n, seed + 1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        km = KMeans(n_clusters=k, init=init, n_init=1, algorithm="lloyd", tol=0, max_iter=20).fit(x)
    assert np.abs(km.cluster_centers_ - r.centroids).max() < 1e-5


def test_kmeans_error_behaviour(oracle):
    x = np.random.default_rng(0).standard_normal((10, 4)).astype(np.float32)
    with pytest.raises(RuntimeError, match="at least as large as number of clusters"):
        oracle.kmeans_train(x, 16)
    x[3, 2] = np.nan
    with pytest.raises(RuntimeError, match="NaN"):
        oracle.kmeans_train(x, 4)
    # n == k corner case: the points become the centroids
    y = np.random.default_rng(1).standard_normal((8, 4)).astype(np.float32)
    assert np.array_equal(oracle.kmeans_train(y, 8, niter=3).centroids, y)

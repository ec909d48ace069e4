"""ctypes front end of the CPU oracle (oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package;
nothing under audio_tokens_amd/ does.  Parity status: UNPINNED (see the header of oracle.c) --
the reference has no tests and its arithmetic lives in torchaudio 2.4.1 / faiss 1.8.0, neither
of which is available offline.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB = None
_MTC = None

_f32p = ctypes.POINTER(ctypes.c_float)
_f64p = ctypes.POINTER(ctypes.c_double)
_i32p = ctypes.POINTER(ctypes.c_int32)
_i64p = ctypes.POINTER(ctypes.c_int64)
_u32p = ctypes.POINTER(ctypes.c_uint32)


def build(force: bool = False) -> None:
    """Compile liboracle.so / libmtcheck.so with the committed Makefile."""
    if force or not (_HERE / "liboracle.so").exists() or not (_HERE / "libmtcheck.so").exists():
        subprocess.run(["make", "-C", str(_HERE)] + (["-B"] if force else []), check=True,
                       stdout=subprocess.DEVNULL)


def _lib():
    global _LIB
    if _LIB is None:
        build()
        L = ctypes.CDLL(str(_HERE / "liboracle.so"))
        L.orc_mt19937_raw.argtypes = [ctypes.c_uint32, ctypes.c_int64, _u32p]
        L.orc_rand_perm.argtypes = [_i32p, ctypes.c_int64, ctypes.c_int64]
        L.orc_l2norm_rows.argtypes = [_f32p, ctypes.c_int64, ctypes.c_int, _f32p]
        L.orc_mel_filterbank.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, _f32p]
        L.orc_mel_filterbank.restype = ctypes.c_int
        L.orc_num_frames.argtypes = [ctypes.c_int64, ctypes.c_int]
        L.orc_num_frames.restype = ctypes.c_int64
        L.orc_logmel.argtypes = [_f32p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                 ctypes.c_int, _f32p, _f32p]
        L.orc_logmel.restype = ctypes.c_int
        L.orc_resample_length.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int]
        L.orc_resample_length.restype = ctypes.c_int64
        L.orc_resample.argtypes = [_f32p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, _f32p]
        L.orc_resample.restype = ctypes.c_int
        for fn in (L.orc_assign, L.orc_assign_ref):
            fn.argtypes = [_f32p, ctypes.c_int64, ctypes.c_int, _f32p, ctypes.c_int, _i64p, _f32p]
        L.orc_split_clusters.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int64, _f32p, _f32p]
        L.orc_split_clusters.restype = ctypes.c_int
        L.orc_kmeans_train.argtypes = [_f32p, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_int, _f32p, _i32p, ctypes.c_int, _f32p, _f64p,
                                       _i32p, _i64p]
        L.orc_kmeans_train.restype = ctypes.c_int
        L.orc_num_threads.restype = ctypes.c_int
        _LIB = L
    return _LIB


def _mtc():
    global _MTC
    if _MTC is None:
        build()
        M = ctypes.CDLL(str(_HERE / "libmtcheck.so"))
        M.mtc_raw.argtypes = [ctypes.c_uint32, ctypes.c_int64, _u32p]
        M.mtc_rand_perm.argtypes = [_i32p, ctypes.c_size_t, ctypes.c_int64]
        M.mtc_rand_floats.argtypes = [ctypes.c_uint32, ctypes.c_int64, _f32p]
        _MTC = M
    return _MTC


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a: np.ndarray, t):
    return a.ctypes.data_as(t) if a is not None else None


# -- rng -------------------------------------------------------------------------------------
def mt19937_raw(seed: int, n: int) -> np.ndarray:
    out = np.empty(n, np.uint32)
    _lib().orc_mt19937_raw(seed, n, _p(out, _u32p))
    return out


def rand_perm(n: int, seed: int) -> np.ndarray:
    out = np.empty(n, np.int32)
    _lib().orc_rand_perm(_p(out, _i32p), n, seed)
    return out


def std_mt19937_raw(seed: int, n: int) -> np.ndarray:
    out = np.empty(n, np.uint32)
    _mtc().mtc_raw(seed, n, _p(out, _u32p))
    return out


def std_rand_perm(n: int, seed: int) -> np.ndarray:
    out = np.empty(n, np.int32)
    _mtc().mtc_rand_perm(_p(out, _i32p), n, seed)
    return out


def std_rand_floats(seed: int, n: int) -> np.ndarray:
    out = np.empty(n, np.float32)
    _mtc().mtc_rand_floats(seed, n, _p(out, _f32p))
    return out


# -- row normalisation -----------------------------------------------------------------------
def l2norm_rows(x) -> np.ndarray:
    x = _f32(x)
    n, d = x.shape
    y = np.empty_like(x)
    _lib().orc_l2norm_rows(_p(x, _f32p), n, d, _p(y, _f32p))
    return y


# -- log-mel ---------------------------------------------------------------------------------
def mel_filterbank(sample_rate: int, n_fft: int, n_mels: int) -> np.ndarray:
    fb = np.empty((n_fft // 2 + 1, n_mels), np.float32)
    rc = _lib().orc_mel_filterbank(sample_rate, n_fft, n_mels, _p(fb, _f32p))
    assert rc == 0
    return fb


def num_frames(L: int, hop: int) -> int:
    return int(_lib().orc_num_frames(L, hop))


def logmel(wave, sample_rate=22050, n_fft=512, hop=128, n_mels=64, fb=None) -> np.ndarray:
    """wave [L] -> [n_mels, T] float32 (mel-major, the reference's .npy layout)."""
    wave = _f32(wave).reshape(-1)
    T = num_frames(wave.shape[0], hop)
    out = np.empty((n_mels, T), np.float32)
    fbp = _p(_f32(fb), _f32p) if fb is not None else None
    rc = _lib().orc_logmel(_p(wave, _f32p), wave.shape[0], sample_rate, n_fft, hop, n_mels, fbp,
                           _p(out, _f32p))
    if rc != 0:
        raise ValueError("orc_logmel: bad arguments")
    return out


def resample(wave, orig_freq: int, new_freq: int) -> np.ndarray:
    """torchaudio.transforms.Resample(orig_freq, new_freq)(wave) for one clip: [L] -> [ceil(L*new/orig)]."""
    wave = _f32(wave).reshape(-1)
    if orig_freq == new_freq:
        return wave.copy()
    out = np.empty(int(_lib().orc_resample_length(wave.shape[0], orig_freq, new_freq)), np.float32)
    if _lib().orc_resample(_p(wave, _f32p), wave.shape[0], orig_freq, new_freq, _p(out, _f32p)) != 0:
        raise ValueError("orc_resample: bad arguments")
    return out


# -- search / k-means ------------------------------------------------------------------------
def assign(x, c, ref: bool = False):
    """IndexFlatL2(c).search(x, 1) -> (ids int64 [n], dis float32 [n])."""
    x, c = _f32(x), _f32(c)
    n, d = x.shape
    k = c.shape[0]
    assert c.shape[1] == d
    ids = np.empty(n, np.int64)
    dis = np.empty(n, np.float32)
    fn = _lib().orc_assign_ref if ref else _lib().orc_assign
    fn(_p(x, _f32p), n, d, _p(c, _f32p), k, _p(ids, _i64p), _p(dis, _f32p))
    return ids, dis


def split_clusters(hassign, centroids, n: int):
    h = _f32(hassign).copy()
    c = _f32(centroids).copy()
    k, d = c.shape
    ns = _lib().orc_split_clusters(d, k, n, _p(h, _f32p), _p(c, _f32p))
    return ns, h, c


class KmeansResult:
    __slots__ = ("centroids", "obj", "imbalance", "nsplit", "n_used", "sub_perm", "assign")


def kmeans_train(x, k: int, niter: int = 20, init_centroids=None, shard=None,
                 n_shards: int = 1) -> KmeansResult:
    """faiss.Kmeans(d, k, niter=niter).train(x, init_centroids=...) -- one Clustering::train."""
    x = _f32(x)
    n, d = x.shape
    init = _f32(init_centroids) if init_centroids is not None else None
    if init is not None:
        assert init.shape == (k, d)
    sh = np.ascontiguousarray(shard, np.int32) if shard is not None else None
    cent = np.zeros((k, d), np.float32)
    stats = np.zeros((max(niter, 1), 4), np.float64)
    n_used = min(n, 256 * k)
    sub_perm = np.empty(n_used, np.int32)
    asg = np.empty(n_used, np.int64)
    rc = _lib().orc_kmeans_train(_p(x, _f32p), n, d, k, niter, _p(init, _f32p), _p(sh, _i32p),
                                 n_shards, _p(cent, _f32p), _p(stats, _f64p),
                                 _p(sub_perm, _i32p), _p(asg, _i64p))
    if rc == -2:
        raise RuntimeError(f"Number of training points ({n}) should be at least as large as "
                           f"number of clusters ({k})")
    if rc == -3:
        raise RuntimeError("input contains NaN's or Inf's")
    assert rc == 0
    r = KmeansResult()
    r.centroids = cent
    r.obj = stats[:niter, 0].astype(np.float32)
    r.imbalance = stats[:niter, 1]
    r.nsplit = stats[:niter, 2].astype(np.int64)
    r.n_used = n_used
    r.sub_perm = sub_perm
    r.assign = asg
    return r


def num_threads() -> int:
    return int(_lib().orc_num_threads())

#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz with the CPU oracle (oracle/oracle.c).

The reference (danavery/audio-tokens) ships no tests, fixtures or golden vectors, and its arithmetic
lives in torchaudio 2.4.1 / faiss 1.8.0 which cannot be imported in the build container (ordinary
ModuleNotFoundError, nothing was denied).  These vectors are therefore produced by the build's own
restatement: they pin the oracle against drift and give the GPU tests
something to compare with that does not need the oracle at run time.  Inputs are stored next to the
outputs; nothing here is copied from the reference.

    python tests/golden/make_golden.py
"""
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent))
import oracle  # noqa: E402


def unit(rng, n, d):
    return oracle.l2norm_rows(rng.standard_normal((n, d)).astype(np.float32))


def logmel_case():
    rng = np.random.default_rng(20241016)
    L, sr = 22050, 22050
    t = np.arange(L) / sr
    wave = np.stack([
        0.4 * np.sin(2 * np.pi * 440 * t),                                   # tone
        0.3 * np.sin(2 * np.pi * (100 * t + 4000 * t * t)),                  # chirp
        0.05 * rng.standard_normal(L),                                       # noise
        np.where(np.arange(L) == 9000, 0.9, 0.0) + np.where(np.arange(L) > 15000, 1e-3 * rng.standard_normal(L), 0.0),
    ]).astype(np.float32)                                                    # silence + click + faint tail
    out = {"wave": wave}
    for nm in (64, 128):
        out[f"logmel_{nm}"] = np.stack([oracle.logmel(w, n_mels=nm) for w in wave])
        out[f"fb_{nm}"] = oracle.mel_filterbank(sr, 512, nm)
    np.savez_compressed(HERE / "logmel.npz", **out)


def kmeans_cases():
    rng = np.random.default_rng(4242)
    # (a) plain Lloyd, no subsampling, random init by the FAISS rule
    centers = rng.standard_normal((64, 64)) * 2
    x = oracle.l2norm_rows((centers[rng.integers(0, 64, 2048)] + rng.standard_normal((2048, 64))).astype(np.float32))
    r = oracle.kmeans_train(x, 64, niter=20)
    out = {"a_x": x, "a_centroids": r.centroids, "a_obj": r.obj, "a_nsplit": r.nsplit, "a_assign": r.assign,
           "a_imbalance": r.imbalance}
    # (b) engineered to hit split_clusters: heavy duplicates -> empty clusters every iteration
    base = unit(rng, 24, 64)
    xb = np.repeat(base, 40, axis=0)
    xb[::7] = unit(rng, len(xb[::7]), 64)
    rb = oracle.kmeans_train(xb, 48, niter=6)
    assert rb.nsplit.sum() > 0
    out.update({"b_x": xb, "b_centroids": rb.centroids, "b_obj": rb.obj, "b_nsplit": rb.nsplit, "b_assign": rb.assign})
    # (c) n > 256 k: mt19937 subsample permutation, d = 8 (generic-d kernel path), given init
    xc = unit(rng, 20000, 8)
    init = unit(rng, 64, 8)
    rc = oracle.kmeans_train(xc, 64, niter=5, init_centroids=init)
    out.update({"c_x": xc, "c_init": init, "c_centroids": rc.centroids, "c_obj": rc.obj, "c_sub_perm": rc.sub_perm,
                "c_assign": rc.assign})
    # (d) two-shard data-parallel variant of (a): partial sums per shard, added in shard order
    shard = (np.arange(2048) >= 1000).astype(np.int32)
    rd = oracle.kmeans_train(x, 64, niter=20, shard=shard, n_shards=2)
    out.update({"d_shard": shard, "d_centroids": rd.centroids, "d_obj": rd.obj, "d_assign": rd.assign})
    np.savez_compressed(HERE / "kmeans.npz", **out)


def tokenizer_case():
    rng = np.random.default_rng(77)
    x = unit(rng, 4000, 64)
    c = unit(rng, 256, 64)
    c[200:210] = c[0:10]            # exact duplicate centroids: lowest index must win
    x[100:110] = c[200:210]         # rows identical to them: clamped distance 0, tie
    ids, dis = oracle.assign(x, c)
    ids_ref, dis_ref = oracle.assign(x, c, ref=True)
    assert np.array_equal(ids, ids_ref) and np.array_equal(dis.view(np.uint32), dis_ref.view(np.uint32))
    xs = x[:7]
    ids_s, dis_s = oracle.assign(xs, c)   # n < 20: faiss' direct form
    np.savez_compressed(HERE / "tokenizer.npz", x=x, c=c, ids=ids, dis=dis, ids_small=ids_s, dis_small=dis_s)


def rng_case():
    np.savez_compressed(HERE / "rng.npz", perm_1000_1234=oracle.rand_perm(1000, 1234),
                        perm_50000_1235_head=oracle.rand_perm(50000, 1235)[:512],
                        raw_1234=oracle.mt19937_raw(1234, 64))


if __name__ == "__main__":
    oracle.build()
    logmel_case(); kmeans_cases(); tokenizer_case(); rng_case()
    for p in sorted(HERE.glob("*.npz")):
        print(p.name, p.stat().st_size)

"""BASELINE.json's configs run as workloads.

configs[0] (110 clips, n_mels=64, vocab 256) and a two-batch cut of configs[1] (vocab 500, 128 000-row Lloyd)
go through DevicePipeline at full clip length and are compared with the oracle END TO END: the oracle is fed
the frames the device produced (log-mel is a floating-point kernel held to its tolerance elsewhere; everything
after it is bit-exact by contract) and must reproduce every centroid and every token.  configs[2] / configs[4]
(n_mels=128, vocab 8192 / 16 384): Kmeans.train against the oracle at sizes it finishes in seconds -- cold, warm,
filter on and off -- and size-independent properties at the full Lloyd shapes.  configs[3]: a full 20-iteration
training at 2 097 152 x 8192 x 64 with every acceleration on against the same training on plain dense sweeps."""
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _oracle_pipeline(oracle, frames, T, n_train, k, niter, batch_clips):
    """ClusterCreator.run + SpecTokenizer on host frames, the reference's batch loop (cluster_creator.py:49-59)."""
    cent = None
    for c0 in range(0, n_train, batch_clips):
        c1 = min(n_train, c0 + batch_clips)
        cent = oracle.kmeans_train(frames[c0 * T:c1 * T], k, niter=niter, init_centroids=cent).centroids
    return oracle.l2norm_rows(cent)


def test_config0_whole_pipeline_against_the_oracle(be, oracle):
    """configs[0]: 99 + 11 ten-second clips, n_mels=64, vocab_size=256, niter=20 (Lloyd on a 65 536-row subsample
    of 170 577 frames), tokens for all 189 530 frames."""
    from audio_tokens_amd.pipeline import DevicePipeline
    from audio_tokens_amd.synth import synth_clips
    wave = synth_clips(110, L=220500, seed=4242, device=be.device)
    pipe = DevicePipeline(n_mels=64, vocab_size=256, niter=20, clustering_batch_size=10000, backend=be)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = pipe.run(wave[:99], wave[99:])
    T = res.frames_per_clip
    assert T == 1723
    frames = be.logmel(wave, frame_major=True, l2norm=True).cpu().numpy()
    # the device log-mel against the oracle's on a few clips (tolerance of tests/test_gpu_ops.py::test_logmel_vs_oracle)
    spec = be.logmel(wave[:3]).cpu().numpy()
    for i in range(3):
        ref = oracle.logmel(wave[i].cpu().numpy())
        P, Pr = 10.0 ** (spec[i].astype(np.float64) / 10), 10.0 ** (ref.astype(np.float64) / 10)
        assert (np.abs(P - Pr) <= 2e-5 * Pr + 1e-9 * Pr.max(0, keepdims=True) + 1e-14).all()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        cent = _oracle_pipeline(oracle, frames, T, 99, 256, 20, 10000)
    assert np.array_equal(bits(res.centroids.cpu().numpy()), bits(cent))
    ids, _ = oracle.assign(frames, cent)
    assert np.array_equal(res.tokens_train.cpu().numpy(), ids[:99 * T])
    assert np.array_equal(res.tokens_val.cpu().numpy(), ids[99 * T:])
    assert len(res.kmeans_stats) == 1 and len(res.kmeans_stats[0]) == 20


def test_config1_two_batches_against_the_oracle(be, oracle):
    """configs[1] (n_mels=64, vocab_size=500, 128 000-row Lloyd per batch of 10 000 files): 10 000 + 1 200 train
    clips = one full batch and a warm-started second one, 700 validation clips; 20.5 M frames tokenised on the
    device, 1.3 M of them (first, last and validation frames) checked against the oracle."""
    from audio_tokens_amd.pipeline import DevicePipeline
    from audio_tokens_amd.synth import synth_clips
    n_tr, n_va, k = 11200, 700, 500
    wave = synth_clips(n_tr + n_va, L=220500, seed=4242, device=be.device)
    pipe = DevicePipeline(n_mels=64, vocab_size=k, niter=20, clustering_batch_size=10000, backend=be)
    res = pipe.run(wave[:n_tr], wave[n_tr:])
    T = res.frames_per_clip
    frames = torch.empty(((n_tr + n_va) * T, 64), dtype=torch.float32)
    for c0 in range(0, n_tr + n_va, 2000):
        c1 = min(n_tr + n_va, c0 + 2000)
        frames[c0 * T:c1 * T] = be.logmel(wave[c0:c1], frame_major=True, l2norm=True).cpu()
    frames = frames.numpy()
    cent = _oracle_pipeline(oracle, frames, T, n_tr, k, 20, 10000)
    assert np.array_equal(bits(res.centroids.cpu().numpy()), bits(cent))
    tok = torch.cat([res.tokens_train, res.tokens_val]).cpu().numpy()
    lo, hi = 50 * T, (n_tr - 50) * T
    for a, b in ((0, lo), (hi, (n_tr + n_va) * T)):
        ids, _ = oracle.assign(frames[a:b], cent)
        assert np.array_equal(tok[a:b], ids)
    assert (lo + (n_tr + n_va) * T - hi) >= 1_000_000
    assert [len(s) for s in res.kmeans_stats] == [20, 20]


@pytest.mark.parametrize("k,n1,n2", [(2048, 60000, 40000), (8192, 70000, 30000), (16384, 70000, 30000)])
def test_kmeans_d128_pruned_path_matches_oracle(be, oracle, switches, k, n1, n2):
    """configs[2] / configs[4] feature width and table sizes (k = 16 384 is configs[4]'s vocabulary: 512 groups, the
    bucket path's and split_clusters' largest LDS tables): assign_f16filter_kernel<128,...>, its fused pre-pass and the
    d=128 redo paths inside a full train(), cold start then warm start, filter on and off, against the oracle."""
    from audio_tokens_amd.ops import Kmeans
    rng = np.random.default_rng(k)
    cen = rng.standard_normal((k, 128))
    x = (cen[rng.integers(0, k, n1 + n2)] + 0.5 * rng.standard_normal((n1 + n2, 128))).astype(np.float32)
    x = oracle.l2norm_rows(x)
    x[:200] = x[200:400]                                    # duplicates
    niter = 6 if k < 16384 else 4
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        r1 = oracle.kmeans_train(x[:n1], k, niter=niter)
        r2 = oracle.kmeans_train(x[n1:], k, niter=niter, init_centroids=r1.centroids)
        for label, use_filter, prune in (("filtered", True, True), ("fp32 pruned", False, True), ("dense", True, False)):
            switches(filter=use_filter)
            km = Kmeans(128, k, niter=niter, backend=be)
            km.prune = prune
            km.train(x[:n1])
            assert np.array_equal(bits(km.centroids), bits(r1.centroids)), f"{label} cold"
            assert [s["nsplit"] for s in km.iteration_stats] == list(r1.nsplit), label
            assert np.allclose(km.obj, r1.obj, rtol=2e-5, atol=0)
            km.train(x[n1:], init_centroids=km.centroids)
            assert np.array_equal(bits(km.centroids), bits(r2.centroids)), f"{label} warm"
            assert [s["nsplit"] for s in km.iteration_stats] == list(r2.nsplit), label


@pytest.mark.parametrize("n,k", [(2097152, 8192), (3000000, 16384)])
def test_full_size_d128_lloyd_shape_properties(be, n, k):
    """The Lloyd shapes of configs[2] (2 097 152 x 8192 x 128) and configs[4] (16 384 centroids x 128; 3 M of its
    4 194 304 rows) are beyond the oracle in a test: the exact pruned + filtered d=128 sweep is held to bit equality
    with the dense fp32 sweep (pinned to the oracle at small sizes), self-consistency and re-entrance."""
    from audio_tokens_amd.synth import synth_clips
    clips = -(-n // 1723)
    x = torch.empty((clips * 1723, 128), dtype=torch.float32, device=be.device)
    for c0 in range(0, clips, 600):
        c1 = min(clips, c0 + 600)
        wave = synth_clips(c1 - c0, L=220500, seed=4242, first_clip=c0, device=be.device)
        be.logmel(wave, 22050, 512, 128, 128, frame_major=True, l2norm=True, out=x[c0 * 1723:c1 * 1723])
    x = x[:n].contiguous()
    d = 128
    g = torch.Generator(device="cuda").manual_seed(5)
    pick = torch.randperm(n, device="cuda", generator=g)[:k]
    c = x[pick].clone()
    c += 1e-3 * torch.randn(k, d, device="cuda", generator=g)      # distinct centroids (silent frames repeat)
    ids_d, dis_d = be.assign(x, c)                                  # dense fp32 sweep
    cperm = be.from_host(be.group_rows_kd(be.to_host(c)))
    dmin = be.group_min_dist(c, cperm)
    ids_c, dis_c = be.assign_c2f(x, c, cperm, dmin, coherent=True)  # guess generator + fused filtered sweep
    assert torch.equal(ids_c, ids_d) and torch.equal(dis_c.view(torch.int32), dis_d.view(torch.int32))
    ids_p, dis_p = be.assign_pruned(x, c, be.visit_order(ids_d, dis_d, k), cperm, dmin)   # guided by the answer
    assert torch.equal(ids_p, ids_d) and torch.equal(dis_p.view(torch.int32), dis_d.view(torch.int32))
    own, own_d = be.assign_c2f(c.repeat(8, 1), c, cperm, dmin)
    assert torch.equal(own.view(8, k), torch.arange(k, device="cuda").expand(8, k)) and float(own_d.max()) == 0.0
    probe = torch.randint(0, k, (n,), device="cuda", generator=g)   # a random other centroid per row is never closer
    d_probe = ((x - c[probe]) ** 2).sum(1)
    assert bool((dis_d <= d_probe + 1e-5).all())
    assert int(torch.bincount(ids_d, minlength=k).sum()) == n


@pytest.mark.parametrize("d,k,niter,n_clips", [(64, 8192, 20, 1218), (128, 8192, 20, 1218), (128, 16384, 5, 2440)])
def test_full_size_20_iterations_accelerated_equals_dense(be, d, k, niter, n_clips):
    """configs[3] (d=64) and configs[2] (d=128) Lloyd shape, all 20 iterations of one FAISS-recipe training on
    2 097 152 rows x 8192 clusters: pruning + fp16 filter + guess generators against plain dense fp32 sweeps.
    Centroids, repairs and objectives must be bit-equal, and so must a warm-started second training.
    configs[4]'s shape (d=128, 16 384 clusters): the training frames of 2 440 clips (4 204 120 rows) are cut to the
    256 k = 4 194 304-row subsample by the device permutation inside train(), five iterations cold and five warm."""
    from audio_tokens_amd.ops import Kmeans
    from audio_tokens_amd.synth import synth_clips
    rows = 256 * k
    x = torch.empty((n_clips * 1723, d), dtype=torch.float32, device=be.device)
    for c0 in range(0, n_clips, 610):
        c1 = min(n_clips, c0 + 610)
        wave = synth_clips(c1 - c0, L=220500, seed=99, first_clip=c0, device=be.device)
        be.logmel(wave, 22050, 512, 128, d, frame_major=True, l2norm=True, out=x[c0 * 1723:c1 * 1723])
        del wave
    assert x.shape[0] >= rows
    x2 = x[:rows * 3 // 4]
    # (k = 8192: exactly 256 k rows, no subsampling; k = 16 384: a few thousand rows more, so train() subsamples)
    x = x[-rows:].contiguous() if k == 8192 else x
    runs = {}
    for prune in (True, False):
        km = Kmeans(d, k, niter=niter, backend=be)
        km.prune = prune
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            km.train(x)
            first = (km.centroids_device.clone(), list(km.obj), [s["nsplit"] for s in km.iteration_stats])
            km.train(x2, init_centroids=km.centroids_device)
        runs[prune] = first + (km.centroids_device.clone(), list(km.obj), [s["nsplit"] for s in km.iteration_stats])
    a, b = runs[True], runs[False]
    assert torch.equal(a[0].view(torch.int32), b[0].view(torch.int32)), "cold training: centroids differ"
    assert a[1] == b[1] and a[2] == b[2]
    assert torch.equal(a[3].view(torch.int32), b[3].view(torch.int32)), "warm training: centroids differ"
    assert a[4] == b[4] and a[5] == b[5]
    assert all(later <= earlier * (1 + 1e-6) for earlier, later in zip(a[1][1:], a[1][2:]))

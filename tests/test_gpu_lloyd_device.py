"""GPU parity of the pieces that took the host out of the Lloyd loop (csrc/randperm.hip, csrc/lloyd.hip):
the FAISS subsample permutation and split_clusters computed on the device must give the bits of the
sequential host forms (which tests/test_oracle_golden.py pins to libstdc++'s std::mt19937 and the oracle)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("n,m", [(1, 1), (2, 1), (2, 2), (3, 3), (7, 7), (100, 100), (100, 99), (100, 1), (625, 625),
                                 (1000, 37), (5000, 4999), (70000, 70000), (200000, 65536), (1000003, 131072),
                                 (4307500, 2097152), (17230000, 2097152), (17230000, 4194304)])
def test_rand_perm_device_equals_host(be, n, m):
    """at_rand_perm_prefix_device == first m entries of faiss' rand_perm(n, seed): full permutations (every
    position touched many times), self swaps, prefixes, the benchmark's sizes; two seeds."""
    for seed in (1234, 1235):
        want = be.rand_perm_prefix(n, seed, m)
        got = be.rand_perm_prefix_device(n, seed, m).cpu().numpy()
        assert np.array_equal(got, want), f"n={n} m={m} seed={seed}: {(got != want).sum()} entries differ"
    if m == n:   # a permutation
        assert np.array_equal(np.sort(got), np.arange(n, dtype=np.int32))


def test_rand_perm_device_beside_other_work(be):
    """Its workspace is its own: a permutation queued on a side stream while the main stream sorts and
    accumulates gives the same bits."""
    rng = np.random.default_rng(3)
    x = torch.from_numpy(rng.standard_normal((200000, 64)).astype(np.float32)).to(be.device)
    ids = torch.from_numpy(rng.integers(0, 500, 200000)).to(be.device)
    want = be.rand_perm_prefix(3000000, 1234, 500000)
    side = torch.cuda.Stream(be.device)
    side.wait_stream(torch.cuda.current_stream(be.device))
    with torch.cuda.stream(side):
        got = be.rand_perm_prefix_device(3000000, 1234, 500000)
    for _ in range(4):
        be.centroid_accum(x, ids, 500)
    torch.cuda.current_stream(be.device).wait_stream(side)
    assert np.array_equal(got.cpu().numpy(), want)


def _split_case(rng, k, d, n, n_empty, big=None):
    """Counts with n_empty zeros summing to n, and centroids."""
    h = np.zeros(k, np.float32)
    alive = rng.permutation(k)[: k - n_empty]
    w = rng.random(len(alive)) ** 3 + 1e-3
    cnt = np.maximum(1, np.floor(w / w.sum() * (n - len(alive)))).astype(np.int64)
    if big is not None:
        cnt[:] = 1
        cnt[rng.integers(0, len(alive), big)] = 2          # hardly any donor: long acceptance scans
    cnt[0] += n - cnt.sum()
    assert cnt[0] >= 1
    h[alive] = cnt
    c = rng.standard_normal((k, d)).astype(np.float32)
    c[h == 0] = 0
    return h, c


@pytest.mark.parametrize("k,d,n,n_empty,big", [(64, 64, 4096, 3, None), (48, 8, 1000, 17, None), (8192, 64, 2097152, 1, None),
                                               (8192, 64, 2097152, 40, None), (500, 64, 128000, 0, None),
                                               (1000, 128, 1003, 5, 3), (300, 640, 90000, 299, None), (16384, 128, 4194304, 7, None),
                                               (8192, 64, 2097152, 700, None), (20000, 64, 5120000, 30, None)])
def test_split_clusters_device_equals_host(be, k, d, n, n_empty, big):
    """(8192 clusters with 700 empty ones draw ~5.7 M numbers: past the context's resident 4.19 M-draw stream, into
    the kernel's own generator; k = 20 000 keeps the probabilities in global memory)"""
    rng = np.random.default_rng(k + n_empty)
    h, c = _split_case(rng, k, d, n, n_empty, big)
    h_ref, c_ref = h.copy(), c.copy()
    ns_ref = be.split_clusters(h_ref, c_ref, n)
    assert ns_ref == n_empty
    hd, cd = be.from_host(h), be.from_host(c)
    nsd = torch.full((1,), -7, dtype=torch.int32, device=be.device)
    be.split_clusters_device(hd, cd, n, nsd)
    assert int(nsd.item()) == ns_ref
    assert np.array_equal(bits(hd.cpu().numpy()), bits(h_ref))
    assert np.array_equal(bits(cd.cpu().numpy()), bits(c_ref))


def test_lloyd_stats(be):
    rng = np.random.default_rng(11)
    k, d, world = 500, 64, 3
    h = rng.integers(0, 5000, k).astype(np.float32)
    off, total = be.part_layout(k, d)
    parts = torch.zeros((world, total), dtype=torch.float32, device=be.device)
    objs = [123.456789012345, 1e-3, 98765.4321]
    for r, o in enumerate(objs):
        parts[r, off:off + 2].view(torch.float64)[0] = o
    st = torch.zeros((2,), dtype=torch.float64, device=be.device)
    be.lloyd_stats(be.from_host(h), parts, k, d, st)
    st = st.cpu().numpy()
    assert st[0] == (objs[0] + objs[1]) + objs[2]
    hd = h.astype(np.float64)
    assert st[1] == (hd * hd).sum() * k / (hd.sum() ** 2)


@pytest.mark.parametrize("k,d,n,n_empty", [(64, 64, 4096, 3), (8192, 64, 2097152, 0), (8192, 64, 2097152, 40), (500, 64, 128000, 0)])
def test_lloyd_stats_split_in_one_launch(be, k, d, n, n_empty):
    """at_lloyd_stats_split_f32 == at_lloyd_stats_f64 followed by at_split_clusters_f32, bit for bit: the statistics
    describe the counts as the assignment left them, the repair then changes counts and centroids."""
    rng = np.random.default_rng(k + n_empty + 1)
    h, c = _split_case(rng, k, d, n, n_empty, None)
    off, total = be.part_layout(k, d)
    parts = torch.zeros((2, total), dtype=torch.float32, device=be.device)
    parts[0, off:off + 2].view(torch.float64)[0] = 3.25
    parts[1, off:off + 2].view(torch.float64)[0] = 1e-7
    h1, c1, h2, c2 = be.from_host(h), be.from_host(c), be.from_host(h), be.from_host(c)
    st1 = torch.zeros((2,), dtype=torch.float64, device=be.device)
    st2 = torch.zeros((2,), dtype=torch.float64, device=be.device)
    ns1 = torch.full((1,), -7, dtype=torch.int32, device=be.device)
    ns2 = torch.full((1,), -7, dtype=torch.int32, device=be.device)
    be.lloyd_stats(h1, parts, k, d, st1)
    be.split_clusters_device(h1, c1, n, ns1)
    be.lloyd_stats_split(h2, c2, n, ns2, parts, st2)
    assert int(ns1.item()) == int(ns2.item()) == n_empty
    assert torch.equal(st1.view(torch.int64), st2.view(torch.int64))
    assert torch.equal(h1.view(torch.int32), h2.view(torch.int32)) and torch.equal(c1.view(torch.int32), c2.view(torch.int32))


@pytest.mark.parametrize("sync", [True, False])
def test_kmeans_train_without_host_round_trips_matches_oracle(be, oracle, sync):
    """The restructured loop (device permutation, device split_clusters, statistics read at the end) against
    the oracle: a subsampled cold start that has to repair empty clusters, then a warm start."""
    rng = np.random.default_rng(77)
    d, k = 64, 64
    x = oracle.l2norm_rows((rng.standard_normal((40000, d)) + 3 * rng.standard_normal((1, d))).astype(np.float32))
    x[:3000] = x[0]                              # a heavy duplicate: its cluster mates come out empty
    from audio_tokens_amd.ops import Kmeans
    km = Kmeans(d, k, niter=8, backend=be)
    loss = km.train(x, sync=sync)
    r = oracle.kmeans_train(x, k, niter=8)
    assert np.array_equal(bits(km.centroids), bits(r.centroids))
    assert [s["nsplit"] for s in km.iteration_stats] == list(r.nsplit)
    assert sum(r.nsplit) > 0, "the case is meant to exercise split_clusters"
    # (the objective is a statistic: summed in double on the device, sequentially in float by faiss / the oracle)
    np.testing.assert_allclose(km.obj, r.obj, rtol=2e-5)
    np.testing.assert_allclose([s["imbalance_factor"] for s in km.iteration_stats], r.imbalance, rtol=1e-12)
    assert (loss is None) == (not sync)
    x2 = oracle.l2norm_rows(rng.standard_normal((30000, d)).astype(np.float32))
    km.train(x2, init_centroids=km.centroids_device, sync=sync)
    r2 = oracle.kmeans_train(x2, k, niter=8, init_centroids=r.centroids)
    assert np.array_equal(bits(km.centroids), bits(r2.centroids))
    assert [s["nsplit"] for s in km.iteration_stats] == list(r2.nsplit)


def test_split_clusters_device_without_donor_terminates(be):
    """Every cluster has at most one member, so no scan can ever accept: faiss would spin forever; the kernel
    gives up after 64 cycles over the clusters and reports -1."""
    k, d = 64, 8
    h = np.ones(k, np.float32)
    h[5] = 0
    hd, cd = be.from_host(h), be.zeros((k, d))
    nsd = torch.zeros((1,), dtype=torch.int32, device=be.device)
    be.split_clusters_device(hd, cd, 1000, nsd)
    assert int(nsd.item()) == -1

"""GPU parity: every device operator of the C ABI against the CPU oracle on the same inputs.

Bar: bit-exact for ids / distances / centroid sums / row normalisation (integer-like and
order-pinned fp32 work); log-mel within the tolerance stated in test_logmel (fp32 FFT vs the
oracle's exact value)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _unit_rows(rng, n, d, oracle):
    return oracle.l2norm_rows(rng.standard_normal((n, d)).astype(np.float32))


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("n,d,k", [(1000, 64, 256), (4099, 64, 500), (513, 128, 100), (2048, 128, 8192),
                                   (100000, 64, 8192), (3000, 64, 64), (777, 64, 1)])
def test_assign_bit_exact(be, oracle, n, d, k):
    rng = np.random.default_rng(n * 7 + k)
    x = _unit_rows(rng, n, d, oracle)
    c = _unit_rows(rng, k, d, oracle)
    ids_o, dis_o = oracle.assign(x, c)
    ids, dis = be.assign(x, c)
    ids, dis = ids.cpu().numpy(), dis.cpu().numpy()
    assert np.array_equal(ids, ids_o), f"{(ids != ids_o).sum()} of {n} ids differ"
    assert np.array_equal(bits(dis), bits(dis_o))


@pytest.mark.parametrize("n,d,k", [(70000, 64, 500), (66000, 128, 300), (100000, 64, 1000), (65536, 64, 128)])
def test_assign_unguided_small_tables_bit_exact(be, oracle, n, d, k):
    """Tables too small to prune (configs[1]: vocab_size 500) go through the fp16-split filter with every group
    visited (HipBackend.assign_unguided; IndexFlatL2.assign takes it for 128 <= k < 1024 and >= 65536 rows): same ids
    and distances as the oracle's brute force, exact ties and duplicated centroids included."""
    from audio_tokens_amd.ops import IndexFlatL2
    rng = np.random.default_rng(n + k)
    centers = _unit_rows(rng, k, d, oracle)
    x = oracle.l2norm_rows((centers[rng.integers(0, k, n)] + 0.3 * rng.standard_normal((n, d))).astype(np.float32))
    c = centers.copy()
    c[k // 2: k // 2 + 10] = c[0:10]                  # duplicates: the lowest index wins
    x[:10] = c[k // 2: k // 2 + 10]
    x[10:20] *= 7.5                                   # un-normalised rows
    ids_o, dis_o = oracle.assign(x, c)
    ids, dis = be.assign_unguided(x, c)
    assert np.array_equal(ids.cpu().numpy(), ids_o)
    assert np.array_equal(bits(dis.cpu().numpy()), bits(dis_o))
    index = IndexFlatL2(d, backend=be)
    index.add(c)
    be.filter_stats()
    D, I = index.search(x, 1)
    assert be.filter_stats()[0] == n                  # (the search went through the filter sweep)
    assert np.array_equal(I[:, 0], ids_o) and np.array_equal(bits(D[:, 0]), bits(dis_o))


def test_assign_unnormalised_and_ties(be, oracle):
    rng = np.random.default_rng(5)
    # un-normalised rows, duplicated centroids (exact ties -> lowest index), duplicated rows
    x = (rng.standard_normal((5000, 64)) * rng.uniform(0.1, 30, (5000, 1))).astype(np.float32)
    c = (rng.standard_normal((300, 64)) * 5).astype(np.float32)
    c[200:250] = c[0:50]          # exact duplicates of lower-index centroids
    x[100:150] = c[200:250]       # rows identical to a duplicated centroid: dis clamps to 0 on both
    ids_o, dis_o = oracle.assign(x, c)
    ids, dis = be.assign(x, c)
    assert np.array_equal(ids.cpu().numpy(), ids_o)
    assert np.array_equal(bits(dis.cpu().numpy()), bits(dis_o))
    assert (ids_o[100:150] < 200).all()


@pytest.mark.parametrize("n,d,k", [(5000, 64, 300), (70000, 64, 8192), (3000, 128, 1000), (40000, 128, 4096)])
def test_assign_hinted_is_hint_independent(be, oracle, n, d, k):
    """at_assign_hinted_f32 must return the brute-force answer whatever it is told."""
    rng = np.random.default_rng(n + d + k)
    x = _unit_rows(rng, n, d, oracle)
    c = _unit_rows(rng, k, d, oracle)
    c[k // 2: k // 2 + 20] = c[0:20]                 # duplicate centroids: lowest index must win
    x[:20] = c[k // 2: k // 2 + 20]                  # rows sitting on them (clamped distance 0)
    ids_o, dis_o = oracle.assign(x, c)
    xt, ct = be._f32(x), be._f32(c)
    truth = torch.from_numpy(ids_o).to(be.device)
    hints = {
        "truth": truth,
        "random": torch.from_numpy(rng.integers(0, k, n)).to(be.device),
        "none": torch.full((n,), -1, dtype=torch.int64, device=be.device),
        "dup_high": torch.where(truth < 20, truth + k // 2, truth),   # the higher-index twin of the winner
        "mixed": torch.where(torch.arange(n, device=be.device) % 3 == 0, truth, (truth + 7) % k),
    }
    for name, hint in hints.items():
        part, order = be.centroid_accum(xt, torch.clamp(hint, min=0), k, want_order=True)
        for od in (None, order):
            ids, dis = be.assign_hinted(xt, ct, hint.contiguous(), od)
            assert np.array_equal(ids.cpu().numpy(), ids_o), f"hint={name} order={'yes' if od is not None else 'no'}"
            assert np.array_equal(bits(dis.cpu().numpy()), bits(dis_o)), name
    # the order returned by centroid_accum is the stable sort by (id, row)
    part, (order, sorted_ids) = be.centroid_accum(xt, truth, k, want_order=True)
    ref_order = np.argsort(ids_o, kind="stable")
    assert np.array_equal(order.cpu().numpy().view(np.uint32), ref_order.astype(np.uint32))
    assert np.array_equal(sorted_ids.cpu().numpy().view(np.uint32), ids_o[ref_order].astype(np.uint32))


@pytest.mark.parametrize("n,d,k,lds_kernel", [(6000, 64, 300, False), (70000, 64, 8192, False), (3000, 128, 1000, False),
                                              (50000, 128, 4096, False), (20000, 64, 2048, False),
                                              (33333, 128, 1500, False), (33333, 64, 1500, True), (25001, 64, 16384, False),
                                              (30001, 128, 16384, False)])   # (last: the shape of BASELINE configs[4])
def test_assign_pruned_is_exact(be, oracle, switches, n, d, k, lds_kernel):
    """at_assign_pruned_f32 == brute force, bit for bit, for good guesses (where it prunes), bad
    guesses, missing guesses and exact ties."""
    if lds_kernel:
        switches(prune_kernel=0)   # the LDS-DMA form of the d=64 kernel
    rng = np.random.default_rng(n * 3 + d + k)
    # clustered data so that pruning actually happens
    centers = _unit_rows(rng, k, d, oracle)
    x = oracle.l2norm_rows((centers[rng.integers(0, k, n)] + 0.05 * rng.standard_normal((n, d))).astype(np.float32))
    c = oracle.l2norm_rows((centers + 0.01 * rng.standard_normal((k, d))).astype(np.float32))
    c[k // 2: k // 2 + 20] = c[0:20]
    x[:20] = c[k // 2: k // 2 + 20]
    ids_o, dis_o = oracle.assign(x, c)
    xt, ct = be._f32(x), be._f32(c)
    cperm = be.from_host(be.group_rows_kd(c))
    dmin = be.group_min_dist(ct, cperm)
    # dmin really is a lower bound of the true distances
    cp = cperm.cpu().numpy().reshape(-1, 32)
    for g in (0, cp.shape[0] // 2, cp.shape[0] - 1):
        mem = cp[g][cp[g] >= 0]
        true = np.sqrt(((c[:, None, :].astype(np.float64) - c[None, mem].astype(np.float64)) ** 2).sum(-1)).min(1)
        got = dmin[:, g].cpu().numpy()
        assert (got <= true + 1e-12).all()
        assert (got >= true - 0.02).all()                      # and not uselessly loose
    truth = torch.from_numpy(ids_o).to(be.device)
    dtruth = torch.from_numpy(dis_o).to(be.device)
    cases = {
        "truth": (truth, dtruth),
        "random": (torch.from_numpy(rng.integers(0, k, n)).to(be.device), None),
        "none": (torch.full((n,), -1, dtype=torch.int64, device=be.device), None),
        "dup_high": (torch.where(truth < 20, truth + k // 2, truth), dtruth),
        "mixed": (torch.where(torch.arange(n, device=be.device) % 5 == 0, (truth + 11) % k, truth), dtruth),
    }
    for name, (hint, hd) in cases.items():
        order = be.visit_order(hint.contiguous(), hd, k)
        for flt in (False, True):                                   # fp32 sweep alone / behind the fp16-split filter
            ids, dis = be.assign_pruned(xt, ct, order, cperm, dmin, filter=flt)
            assert np.array_equal(ids.cpu().numpy(), ids_o), f"hint={name} filter={flt}: {(ids.cpu().numpy() != ids_o).sum()} ids differ"
            assert np.array_equal(bits(dis.cpu().numpy()), bits(dis_o)), (name, flt)
            ids, none = be.assign_pruned(xt, ct, order, cperm, dmin, want_dist=False, filter=flt)
            assert none is None and np.array_equal(ids.cpu().numpy(), ids_o), (name, flt)
    # unguided coarse-to-fine search: same answer again
    ids, dis = be.assign_c2f(xt, ct, cperm, dmin)
    assert np.array_equal(ids.cpu().numpy(), ids_o) and np.array_equal(bits(dis.cpu().numpy()), bits(dis_o))
    # ... and with the one-launch guess generator for rows in their own order
    ids, dis = be.assign_c2f(xt, ct, cperm, dmin, coherent=True)
    assert np.array_equal(ids.cpu().numpy(), ids_o) and np.array_equal(bits(dis.cpu().numpy()), bits(dis_o))
    # the visiting order is a permutation sorted by guess
    order, hs = be.visit_order(truth, dtruth, k)
    o = order.cpu().numpy().view(np.uint32).astype(np.int64)
    assert np.array_equal(np.sort(o), np.arange(n))
    assert np.array_equal(hs.cpu().numpy().view(np.uint32), ids_o[o].astype(np.uint32))
    assert (np.diff(ids_o[o]) >= 0).all()


def _filter_eps(d, H):
    """The a-priori bound of csrc/filter.hip on |P16 + |x|^2 - true squared distance| (header there)."""
    u, q, m = 2.0 ** -24, np.sqrt(d) * 2.0 ** -25, 3.0 * d / 16.0
    return H * (34 * m * u * 1.01 + 3 * 2.0 ** -22 + d * u * 1.01 + 2 * u + 2 * q) + 4 * q


@pytest.mark.parametrize("scale_x,scale_c,d", [(1.0, 1.0, 64), (30.0, 0.02, 64), (0.003, 7.0, 64), (1000.0, 1000.0, 64),
                                               (1.0, 1.0, 128), (0.01, 50.0, 128)])
def test_filter_error_bound(be, oracle, scale_x, scale_c, d):
    """Stage 1 alone: the approximate distances stay inside the bound the acceptance test assumes
    (measured against float64), on unit rows and on badly scaled ones, with sign-alternating data
    that makes the inner products cancel."""
    rng = np.random.default_rng(17)
    n, k = 40000, 2048
    x = (rng.standard_normal((n, d)) * rng.choice([1e-3, 1.0], (n, d), p=[0.3, 0.7])).astype(np.float32)
    x = oracle.l2norm_rows(x) * np.float32(scale_x)
    c = oracle.l2norm_rows((x[rng.integers(0, n, k)] / np.float32(scale_x) + 0.02 * rng.standard_normal((k, d))).astype(np.float32))
    c = (c * np.float32(scale_c)).astype(np.float32)
    xt, ct = be._f32(x), be._f32(c)
    ids_o, dis_o = oracle.assign(x, c)
    cperm = be.from_host(be.group_rows_kd(c))
    dmin = be.group_min_dist(ct, cperm)
    order = be.visit_order(torch.from_numpy(ids_o).to(be.device), torch.from_numpy(dis_o).to(be.device), k)
    win, approx, listed = be.filter_probe(xt, ct, order, cperm, dmin)
    win, approx = win.cpu().numpy(), approx.cpu().numpy().astype(np.float64)
    ok = win >= 0
    assert ok.mean() > 0.99
    x64, c64 = x.astype(np.float64), c.astype(np.float64)
    P_true = (c64[win[ok]] ** 2).sum(1) - 2.0 * (x64[ok] * c64[win[ok]]).sum(1)
    H = (x64[ok] ** 2).sum(1) + (c64 ** 2).sum(1).max()
    err = np.abs(approx[ok, 0] - P_true)
    eps = _filter_eps(d, H)
    assert (err <= eps).all(), f"max err/eps {np.max(err / eps):.3f}"
    assert np.max(err / eps) < 0.25           # the budget is meant to be generous
    # rows whose stage-1 winner is not the contract's arg-min must all have been listed
    assert (win != ids_o).sum() <= listed
    ids, dis = be.assign_pruned(xt, ct, order, cperm, dmin, filter=True)
    assert np.array_equal(ids.cpu().numpy(), ids_o) and np.array_equal(bits(dis.cpu().numpy()), bits(dis_o))


def test_filter_near_ties_and_bad_values(be, oracle):
    """Rows engineered to sit exactly on, and within 1e-7 ... 1e-3 of, the bisector of two centroids;
    duplicated centroids; a row and a centroid outside the fp16 range; a NaN row: always the fp32
    contract's answer."""
    rng = np.random.default_rng(23)
    n, d, k = 30000, 64, 1024
    c = _unit_rows(rng, k, d, oracle)
    c[700:720] = c[100:120]                                        # exact duplicates
    a, b = rng.integers(0, k, n), rng.integers(0, k, n)
    t = np.float32(0.5) + rng.choice([0.0, 1e-7, -1e-7, 1e-6, -1e-5, 1e-4, -1e-3], n).astype(np.float32)
    x = (c[a] * t[:, None] + c[b] * (np.float32(1) - t)[:, None] + 0.002 * rng.standard_normal((n, d))).astype(np.float32)
    x[:2000] = (c[a[:2000]] * np.float32(0.5) + c[b[:2000]] * np.float32(0.5)).astype(np.float32)   # exact midpoints
    x[2000:2100] = c[rng.integers(0, k, 100)]                      # rows equal to centroids
    for label in ("plain", "big_row", "big_centroid", "nan_row"):
        xx, cc = x.copy(), c.copy()
        if label == "big_row":
            xx[5, 3] = 70000.0
        if label == "big_centroid":
            cc[9, 1] = 40000.0
        if label == "nan_row":
            xx[7, 0] = np.nan
        xt, ct = be._f32(xx), be._f32(cc)
        ids_d, dis_d = be.assign(xt, ct)                            # dense fp32 sweep (pinned to the oracle elsewhere)
        if label == "plain":
            ids_o, dis_o = oracle.assign(xx, cc)
            assert np.array_equal(ids_d.cpu().numpy(), ids_o) and np.array_equal(bits(dis_d.cpu().numpy()), bits(dis_o))
        cperm = be.from_host(be.group_rows_kd(cc))
        dmin = be.group_min_dist(ct, cperm)
        for hint in (ids_d, torch.from_numpy(a).to(be.device), torch.full((n,), -1, dtype=torch.int64, device=be.device)):
            order = be.visit_order(hint.contiguous(), None, k)
            be.filter_stats()
            ids, dis = be.assign_pruned(xt, ct, order, cperm, dmin, filter=True)
            rows, listed = be.filter_stats()
            assert rows == n and 0 < listed <= n
            if label == "big_centroid":
                assert listed == n                                   # the whole call falls back to fp32
            same = ids == ids_d
            if label == "nan_row":                                   # a NaN row has no defined winner
                same[7] = True
            assert bool(same.all()), label
            keep = torch.ones(n, dtype=torch.bool, device=be.device)
            if label == "nan_row":
                keep[7] = False
            assert torch.equal(dis[keep].view(torch.int32), dis_d[keep].view(torch.int32)), label


@pytest.mark.parametrize("n", [1, 5, 19])
def test_assign_small_batch_form(be, oracle, n):
    rng = np.random.default_rng(n)
    x = _unit_rows(rng, n, 64, oracle)
    c = _unit_rows(rng, 300, 64, oracle)
    ids_o, dis_o = oracle.assign(x, c)
    ids, dis = be.assign(x, c)
    assert np.array_equal(ids.cpu().numpy(), ids_o)
    assert np.array_equal(bits(dis.cpu().numpy()), bits(dis_o))


@pytest.mark.parametrize("d,k", [(8, 20), (40, 33), (80, 500), (640, 64), (192, 700), (50, 77), (6, 9), (1280, 40)])
def test_assign_generic_d(be, oracle, d, k):
    rng = np.random.default_rng(d + k)
    x = _unit_rows(rng, 1500, d, oracle)
    c = _unit_rows(rng, k, d, oracle)
    ids_o, dis_o = oracle.assign(x, c)
    ids, dis = be.assign(x, c)
    assert np.array_equal(ids.cpu().numpy(), ids_o)
    assert np.array_equal(bits(dis.cpu().numpy()), bits(dis_o))


@pytest.mark.parametrize("variant", [0, 4, 5, 6])
def test_assign_anyd_shapes_at_d640(be, oracle, switches, variant):
    """use_convolution's d = 640 (ten 64-feature chunks, the x chunk of the next stage prefetched into a second
    register set): every shape of the chunked MFMA kernel against the oracle, including a k that leaves the last
    128-centroid tile mostly empty and exact duplicates."""
    switches(assign_variant=variant)
    rng = np.random.default_rng(640 + variant)
    n, d, k = 3000, 640, 700
    x = _unit_rows(rng, n, d, oracle)
    c = _unit_rows(rng, k, d, oracle)
    c[600:620] = c[0:20]
    x[:20] = c[600:620]
    ids_o, dis_o = oracle.assign(x, c)
    ids, dis = be.assign(x, c)
    assert np.array_equal(ids.cpu().numpy(), ids_o)
    assert np.array_equal(bits(dis.cpu().numpy()), bits(dis_o))


@pytest.mark.parametrize("n,d", [(1, 64), (1000, 64), (777, 128), (300, 3), (100, 130), (64, 640), (5000, 20)])
def test_l2norm_rows_bit_exact_vs_numpy(be, n, d):
    rng = np.random.default_rng(n + d)
    x = (rng.standard_normal((n, d)) * rng.uniform(1e-3, 1e2, (n, 1))).astype(np.float32)
    if n > 3:
        x[3] = 0.0  # silent frame: 0 / 1e-10 = 0
    ref = x / (np.linalg.norm(x, axis=1, keepdims=True) + 1e-10)
    got = be.l2norm_rows(x).cpu().numpy()
    assert ref.dtype == np.float32
    diff = np.argwhere(bits(got) != bits(ref))
    assert diff.size == 0, f"{len(diff)} of {got.size} values differ, first at {diff[0]}: got {got[tuple(diff[0])]!r} want {ref[tuple(diff[0])]!r}"


def test_gather_rows(be):
    rng = np.random.default_rng(0)
    for d in (64, 128, 7):
        x = rng.standard_normal((1000, d)).astype(np.float32)
        idx = rng.integers(0, 1000, 333).astype(np.int32)
        got = be.gather_rows(x, idx).cpu().numpy()
        assert np.array_equal(got, x[idx])


@pytest.mark.parametrize("n,d,k", [(5000, 64, 37), (20000, 64, 500), (3000, 128, 64), (2000, 640, 10), (1000, 8, 5)])
def test_centroid_accum_is_the_sequential_sum(be, n, d, k):
    rng = np.random.default_rng(n + d + k)
    x = rng.standard_normal((n, d)).astype(np.float32)
    ids = rng.integers(0, k, n).astype(np.int64)
    ids[ids == 3] = 4  # cluster 3 empty
    # skewed: one heavy cluster
    ids[: n // 3] = 1
    sums = np.zeros((k, d), np.float32)
    counts = np.zeros(k, np.float32)
    for i in range(n):  # ascending i, fp32 adds: what FAISS' owning thread does
        sums[ids[i]] += x[i]
        counts[ids[i]] += 1
    part = be.centroid_accum(be._f32(x), torch.from_numpy(ids).to(be.device), k).cpu().numpy()
    assert np.array_equal(bits(part[: k * d].reshape(k, d)), bits(sums))
    assert np.array_equal(part[k * d:], counts)
    cent, h = be.centroid_finalize(torch.from_numpy(part).to(be.device), k, d)
    ref = sums.copy()
    nz = counts > 0
    ref[nz] = sums[nz] * (np.float32(1.0) / counts[nz])[:, None]
    assert np.array_equal(bits(cent.cpu().numpy()), bits(ref))
    assert np.array_equal(h.cpu().numpy(), counts)


@pytest.mark.parametrize("buckets", [1, 0])
@pytest.mark.parametrize("n,d,k,n_long", [(300000, 64, 5000, 3), (400000, 64, 6400, 24), (120000, 128, 2000, 1), (90000, 8, 1500, 2),
                                          (50000, 6, 40, 2), (300000, 64, 500, 3)])
def test_centroid_accum_long_lists_both_paths(be, switches, buckets, n, d, k, n_long):
    """Member lists by the bucket path (count / scan / scatter / in-LDS order; long lists by ordered compaction, more
    than sixteen of them through the rank-sort fallback) and by the radix sort: the sequential sums, the counts, the
    (id, row) order and the ids in that order, bit for bit -- including ids outside [0, k)."""
    switches(accum_buckets=buckets)
    rng = np.random.default_rng(n + k + n_long)
    x = rng.standard_normal((n, d)).astype(np.float32)
    ids = rng.integers(0, k, n)
    heavy = rng.choice(k, n_long, replace=False)
    pick = rng.random(n) < 0.7
    ids[pick] = heavy[rng.integers(0, n_long, int(pick.sum()))]      # 70 % of the rows in the heavy clusters
    ids[rng.integers(0, n, 50)] = -1                                  # ids outside [0, k): a trailing bucket nobody sums
    ids[rng.integers(0, n, 50)] = k + 7
    xt = be._f32(x)
    idt = torch.from_numpy(ids).to(be.device)
    for _ in range(2):                                                # (twice: the scan leaves the counters clean)
        part, (order, sorted_ids) = be.centroid_accum(xt, idt, k, want_order=True)
    valid = (ids >= 0) & (ids < k)
    key = np.where(valid, ids, k)
    ref_order = np.argsort(key, kind="stable")
    nv = int(valid.sum())
    got_order = order.cpu().numpy().view(np.uint32)
    assert np.array_equal(got_order[:nv], ref_order[:nv].astype(np.uint32))
    assert np.array_equal(np.sort(got_order[nv:]), np.sort(ref_order[nv:]).astype(np.uint32))
    assert np.array_equal(sorted_ids.cpu().numpy().view(np.uint32), key[ref_order].astype(np.uint32))
    sums = np.zeros((k, d), np.float32)
    np.add.at(sums, ids[valid], x[valid])                             # unbuffered, ascending row: the sequential fp32 sum
    counts = np.bincount(ids[valid], minlength=k).astype(np.float32)
    p = part.cpu().numpy()
    assert np.array_equal(bits(p[: k * d].reshape(k, d)), bits(sums))
    assert np.array_equal(p[k * d: k * d + k], counts)
    assert int((counts > 2048).sum()) >= min(n_long, 1)
    assert (buckets == 0) or (n > 64 * k) or (d % 4) or int((counts > 2048).sum()) == n_long


@pytest.mark.parametrize("d", [64, 128])
def test_centroid_accum_sort_path_with_more_long_clusters_than_early_slots(be, switches, d):
    """WS_LONG_PRED keeps [count | EARLY_MAX predicted ids | k generation marks].  With the marks laid over the last id
    (round 2) a call with >= 16 long clusters wrote a cluster id into done[0] while cluster 0's thread read it: a long
    cluster 0 could be left un-summed.  Sort path, 20 long clusters INCLUDING cluster 0, four calls in a row (the second
    one onward sums the predicted clusters early), every call against the sequential sum, bit for bit."""
    switches(accum_buckets=0)
    n, k, n_long = 360000, 3000, 20
    rng = np.random.default_rng(d)
    x = rng.standard_normal((n, d)).astype(np.float32)
    xt = be._f32(x)
    heavy = np.concatenate([[0], 1 + rng.choice(k - 1, n_long - 1, replace=False)])
    for call in range(4):
        ids = rng.integers(0, k, n)
        pick = rng.random(n) < 0.75
        ids[pick] = heavy[rng.integers(0, n_long, int(pick.sum()))]
        if call == 2:                                   # the long set changes between calls: predictions partly wrong
            ids[ids == heavy[3]] = heavy[4]
        idt = torch.from_numpy(ids).to(be.device)
        part = be.centroid_accum(xt, idt, k)
        sums = np.zeros((k, d), np.float32)
        np.add.at(sums, ids, x)
        counts = np.bincount(ids, minlength=k).astype(np.float32)
        p = part.cpu().numpy()
        assert int((counts > 2048).sum()) >= 17 and counts[0] > 2048
        assert np.array_equal(p[k * d: k * d + k], counts), call
        assert np.array_equal(bits(p[: k * d].reshape(k, d)), bits(sums)), call


@pytest.mark.parametrize("n_mels,nk,ks", [(64, 10, 3), (128, 10, 3), (40, 4, 5), (64, 1, 1)])
def test_conv1d_mel_matches_torch(be, n_mels, nk, ks):
    """use_convolution: Conv1d(1, nk, ks, padding=ks//2) along the mel axis by at_conv1d_mel_f32 against
    torch's conv1d on the CPU (tolerance: the two sum the taps in their own order) and, exactly, against the taps added
    in ascending order in float64 rounded once per fma; layout feature = mel * nk + kernel."""
    torch.manual_seed(n_mels + nk)
    conv = torch.nn.Conv1d(1, nk, ks, padding=ks // 2)
    x = torch.randn(5000, n_mels) * 20 - 30
    with torch.no_grad():
        want = conv(x.unsqueeze(1)).transpose(1, 2).reshape(x.shape[0], -1)
    got = be.conv1d_mel(x, conv.weight.detach(), conv.bias.detach(), padding=ks // 2).cpu()
    assert got.shape == want.shape
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-5)
    # the kernel's own definition: bias, then fmaf per tap in ascending order (an fma = exact product, one rounding)
    w = conv.weight.detach().reshape(nk, ks).double().numpy(); b = conv.bias.detach().double().numpy()
    xp = np.pad(x.numpy().astype(np.float64), ((0, 0), (ks // 2, ks // 2)))
    acc = np.broadcast_to(b.astype(np.float32)[None, None, :], (x.shape[0], n_mels, nk)).copy()
    for t in range(ks):
        acc = (acc.astype(np.float64) + xp[:, t:t + n_mels, None] * w[None, None, :, t]).astype(np.float32)
    assert np.array_equal(bits(got.numpy().reshape(x.shape[0], n_mels, nk)), bits(acc))


def test_new_entry_points_edges(be):
    """Edges of round 3's entry points: no bias / empty input / a convolution that is not 'same' (at_conv1d_mel_f32);
    the background stream is one stream per context, usable as a torch stream; an empty batch of clips for the fused
    min-max form."""
    from audio_tokens_amd import _lib
    x = torch.randn(100, 64, device="cuda")
    w = torch.randn(4, 1, 3)
    y = be.conv1d_mel(x, w, None, padding=1).cpu()
    want = torch.nn.functional.conv1d(x.cpu().unsqueeze(1), w, None, padding=1).transpose(1, 2).reshape(100, -1)
    assert torch.allclose(y, want, rtol=1e-5, atol=1e-5)
    assert be.conv1d_mel(x[:0], w, None, padding=1).shape == (0, 256)
    with pytest.raises(_lib.NativeError, match="only 'same' convolutions"):
        be.conv1d_mel(x, w, None, padding=0)
    s1, s2 = be.background_stream(), be.background_stream()
    assert s1.cuda_stream == s2.cuda_stream and s1.cuda_stream != torch.cuda.current_stream().cuda_stream
    with torch.cuda.stream(s1):
        z = be.l2norm_rows(x)
    s1.synchronize()
    assert torch.equal(z, be.l2norm_rows(x))
    assert tuple(be.logmel_minmax(torch.zeros(0, 22050, device="cuda")).shape) == (0, 64, 173)


def test_sum_and_nonfinite(be):
    rng = np.random.default_rng(1)
    v = rng.random(1_000_003).astype(np.float32)
    s = be.sum_f64(v).item()
    assert abs(s - v.astype(np.float64).sum()) <= 1e-9 * s
    assert not be.any_nonfinite(v)
    v[12345] = np.inf
    assert be.any_nonfinite(v)
    v[12345] = np.nan
    assert be.any_nonfinite(v)
    # one launch, the last workgroup adds the partials: the same bits every time, at every size, and the arrival
    # counter is back at zero after each call
    for n in (0, 1, 255, 256, 257, 70_000, 262_144 + 5, 3_000_001):
        w = rng.standard_normal(n).astype(np.float32)
        wt = be.from_host(w)
        first = be.sum_f64(wt).item()
        for _ in range(3):
            assert be.sum_f64(wt).item() == first
        assert abs(first - w.astype(np.float64).sum()) <= 1e-9 * max(1.0, np.abs(w).astype(np.float64).sum())


def _logmel_tolerance(got, ref):
    """fp32 FFT noise floor: |dP| <= 2e-5*P + 1e-9*max_frame(P) in the power domain (a frame's
    bins far below its peak are rounding noise in torchaudio's fp32 pipeline too)."""
    P, Pr = 10.0 ** (got.astype(np.float64) / 10), 10.0 ** (ref.astype(np.float64) / 10)
    pmax = Pr.max(axis=-2, keepdims=True)
    return np.abs(P - Pr) <= 2e-5 * Pr + 1e-9 * pmax + 1e-10 * 1e-4


@pytest.mark.parametrize("n_mels", [64, 128])
def test_logmel_vs_oracle(be, oracle, n_mels):
    rng = np.random.default_rng(3)
    L = 22050
    t = np.arange(L) / 22050.0
    clips = np.stack([
        0.4 * np.sin(2 * np.pi * 440 * t),
        0.3 * np.sin(2 * np.pi * (100 * t + 4000 * t * t)),
        0.05 * rng.standard_normal(L),
        0.3 * np.sin(2 * np.pi * 1000 * t) + 1e-3 * rng.standard_normal(L),
        np.where(np.arange(L) == 9000, 0.9, 0.0),
        np.zeros(L),
    ]).astype(np.float32)
    ref = np.stack([oracle.logmel(c, n_mels=n_mels) for c in clips])
    got = be.logmel(clips, n_mels=n_mels).cpu().numpy()
    assert got.shape == ref.shape == (6, n_mels, 173)
    ok = _logmel_tolerance(got, ref)
    assert ok.all(), f"{(~ok).sum()} of {ok.size} bins outside tolerance; worst dB diff {np.abs(got - ref)[~ok].max()}"
    assert (got[5] == -100.0).all()                       # digital silence -> exactly -100 dB
    # frame-major layout and the fused row normalisation
    fm = be.logmel(clips, n_mels=n_mels, frame_major=True).cpu().numpy()
    assert np.array_equal(fm.reshape(6, 173, n_mels), got.transpose(0, 2, 1))
    fmn = be.logmel(clips, n_mels=n_mels, frame_major=True, l2norm=True).cpu().numpy()
    ref_n = fm / (np.linalg.norm(fm, axis=1, keepdims=True) + 1e-10)
    assert np.array_equal(bits(fmn), bits(ref_n))


@pytest.mark.parametrize("hop,n_mels", [(64, 64), (100, 40), (256, 64), (400, 20), (512, 6), (128, 136)])
def test_logmel_other_hops_and_widths(be, oracle, hop, n_mels):
    """Hops that take the register-prefetch staging with 32 or 16 frames per block and the plain staging
    (hop > 307), odd clip strides (unaligned rows: the per-sample path), widths whose unit rows are
    normalised inside the kernel (8..128) and behind it."""
    rng = np.random.default_rng(hop + n_mels)
    L = 30001
    clips = (0.1 * rng.standard_normal((5, L))).astype(np.float32)
    clips[3, 5000:] = 0.0
    ref = np.stack([oracle.logmel(c, hop=hop, n_mels=n_mels) for c in clips])
    got = be.logmel(clips, 22050, 512, hop, n_mels).cpu().numpy()
    assert got.shape == ref.shape
    ok = _logmel_tolerance(got, ref)
    assert ok.all(), f"{(~ok).sum()} of {ok.size} bins outside tolerance"
    fm = be.logmel(clips, 22050, 512, hop, n_mels, frame_major=True).cpu().numpy()
    assert np.array_equal(fm.reshape(5, -1, n_mels), got.transpose(0, 2, 1))
    fmn = be.logmel(clips, 22050, 512, hop, n_mels, frame_major=True, l2norm=True).cpu().numpy()
    ref_n = fm / (np.linalg.norm(fm, axis=1, keepdims=True) + 1e-10)
    assert np.array_equal(bits(fmn), bits(ref_n))


@pytest.mark.parametrize("n_fft,hop,n_mels", [(1024, 512, 64), (1024, 256, 128), (256, 64, 40), (2048, 512, 64), (64, 16, 8),
                                              (4096, 1024, 128), (128, 32, 20)])   # (every pass sequence: 8,4 / 8,8 / 8,8,2 / 8,8,8 / 8,8,8,2 / 8,8,8,4)
def test_logmel_other_nfft(be, oracle, n_fft, hop, n_mels):
    """n_fft is a configuration knob of the reference (audio_tokens_config.py:39; its README documents 1024 / 512):
    every power of two other than the tuned 512 goes through the general kernel (Stockham radix-8 passes, one template
    instance per size) -- same tolerance against the oracle, both layouts, unit rows, a user filterbank, silence exactly
    -100 dB."""
    rng = np.random.default_rng(n_fft + hop)
    L = 30001
    clips = (0.1 * rng.standard_normal((4, L))).astype(np.float32)
    clips[1] = (0.3 * np.sin(2 * np.pi * 1234.5 * np.arange(L) / 22050)).astype(np.float32)
    clips[3, 4000:] = 0.0
    ref = np.stack([oracle.logmel(c, n_fft=n_fft, hop=hop, n_mels=n_mels) for c in clips])
    got = be.logmel(clips, 22050, n_fft, hop, n_mels).cpu().numpy()
    assert got.shape == ref.shape == (4, n_mels, 1 + L // hop)
    ok = _logmel_tolerance(got, ref)
    assert ok.all(), f"{(~ok).sum()} of {ok.size} bins outside tolerance"
    assert (got[3, :, -3:] == -100.0).all()
    fm = be.logmel(clips, 22050, n_fft, hop, n_mels, frame_major=True).cpu().numpy()
    assert np.array_equal(fm.reshape(4, -1, n_mels), got.transpose(0, 2, 1))
    fmn = be.logmel(clips, 22050, n_fft, hop, n_mels, frame_major=True, l2norm=True).cpu().numpy()
    assert np.array_equal(bits(fmn), bits(fm / (np.linalg.norm(fm, axis=1, keepdims=True) + 1e-10)))
    fb = oracle.mel_filterbank(22050, n_fft, n_mels)[:, ::-1].copy()          # a user filterbank: mel axis reversed
    rev = be.logmel(clips, 22050, n_fft, hop, n_mels, fb=fb).cpu().numpy()
    assert _logmel_tolerance(rev[:, ::-1], ref).all()
    again = be.logmel(clips, 22050, n_fft, hop, n_mels).cpu().numpy()          # and back to the library's own
    assert np.array_equal(bits(again), bits(got))


def test_logmel_full_length_clip_shapes(be, oracle):
    rng = np.random.default_rng(4)
    w = (0.1 * rng.standard_normal((3, 220500))).astype(np.float32)
    got = be.logmel(w).cpu().numpy()
    assert got.shape == (3, 64, 1723)
    ref = oracle.logmel(w[1])
    assert _logmel_tolerance(got[1], ref).all()


# ---------------------------------------------------------------------------------------------
# resampler
@pytest.mark.parametrize("orig_freq,new_freq", [(44100, 22050), (48000, 22050), (16000, 22050), (22050, 16000)])
def test_resample_matches_oracle(be, oracle, orig_freq, new_freq):
    rng = np.random.default_rng(orig_freq)
    for L, B in ((1, 1), (37, 3), (20001, 2), (orig_freq, 4)):
        w = rng.standard_normal((B, L)).astype(np.float32)
        got = be.to_host(be.resample(w, orig_freq, new_freq))
        assert got.shape == (B, be.resample_length(L, orig_freq, new_freq))
        taps = be.resample_taps(orig_freq, new_freq)[0]
        # fp32 fma chain of K taps against the oracle's double dot: |err| <= K * 2^-24 * sum|t||x|
        tol = taps.shape[1] * 2.0 ** -24 * np.abs(taps).sum(1).max() * np.abs(w).max()
        for b in range(B):
            want = oracle.resample(w[b], orig_freq, new_freq)
            assert np.abs(got[b] - want).max() <= tol


def test_resample_kernels_agree_bitwise(be, switches):
    """The LDS-tiled kernel and the plain one run the same ascending-k fma chain."""
    for orig_freq, new_freq, L in ((44100, 22050, 50001), (48000, 22050, 30011), (11025, 22050, 999),
                                   (96000, 22050, 40000), (16000, 22050, 16000)):
        w = torch.randn(3, L, device=be.device)
        switches(resample_simple=0)
        tiled = be.resample(w, orig_freq, new_freq)
        switches(resample_simple=1)
        plain = be.resample(w, orig_freq, new_freq)
        assert torch.equal(tiled, plain), (orig_freq, new_freq)


def test_resample_strided_rows_and_class(be, oracle):
    from audio_tokens_amd.ops import Resample
    w = torch.randn(3, 5000, device=be.device)
    view = w[:, :4096]                                   # row stride 5000, length 4096
    got = be.to_host(Resample(48000, 22050, backend=be)(view))
    for b in range(3):
        want = oracle.resample(view[b].cpu().numpy(), 48000, 22050)
        np.testing.assert_allclose(got[b], want, rtol=0, atol=1e-5)
    same = Resample(22050, 22050, backend=be)(view)
    assert same is view
    one = Resample(44100, 22050, backend=be)(view[0])
    assert tuple(one.shape) == (2048,)
    with pytest.raises(ValueError):
        Resample(44100.5, 22050, backend=be)


@pytest.mark.parametrize("k", [500, 8192, 20000])
def test_token_histogram(be, k):
    rng = np.random.default_rng(k)
    ids = rng.integers(-1, k, 300001)            # -1 = "no token": skipped
    ids[:1000] = 7                                # one heavy bin
    got = be.token_histogram(ids, k).cpu().numpy()
    want = np.bincount(ids[ids >= 0], minlength=k)
    assert got.dtype == np.int64 and np.array_equal(got, want)
    assert int(be.token_histogram(np.zeros(0, np.int64), k).sum()) == 0


def test_full_size_lloyd_shape_properties(be):
    """BASELINE.json configs[3]'s Lloyd shape (2 097 152 rows x 8192 centroids x 64) is beyond the CPU
    oracle's reach in a test, so the exact pruned + filtered sweep is held to size-independent
    properties there: bit equality with the dense fp32 sweep (itself pinned to the oracle at small
    sizes), every centroid is its own nearest neighbour at distance 0, no sampled centroid is closer
    than the reported one, and a second sweep guided by the result reproduces it."""
    from audio_tokens_amd.synth import synth_clips
    wave = synth_clips(1218, L=220500, seed=4242, device=be.device)
    x = be.logmel(wave, 22050, 512, 128, 64, frame_major=True, l2norm=True)
    n, d, k = x.shape[0], 64, 8192
    assert n >= 2097152
    x = x[:2097152].contiguous()
    n = x.shape[0]
    g = torch.Generator(device="cuda").manual_seed(3)
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
    own, own_d = be.assign_c2f(c.repeat(16, 1), c, cperm, dmin)      # 131 072 queries = the centroids themselves
    assert torch.equal(own.view(16, k), torch.arange(k, device="cuda").expand(16, k)) and float(own_d.max()) == 0.0
    probe = torch.randint(0, k, (n,), device="cuda", generator=g)   # a random other centroid per row is never closer
    d_probe = ((x - c[probe]) ** 2).sum(1)
    assert bool((dis_d <= d_probe + 1e-5).all())
    assert int(torch.bincount(ids_d, minlength=k).sum()) == n


def test_full_size_kmeans_is_independent_of_the_acceleration(be, switches):
    """One FAISS-recipe training at the benchmark's size (2.1 M rows, 8192 clusters): the fp16-split
    filter + fused pre-pass + exact pruning must leave centroids and objectives exactly where the
    plain pruned fp32 sweep puts them, and the objective must not rise between Lloyd iterations."""
    import warnings
    from audio_tokens_amd.ops import Kmeans
    from audio_tokens_amd.synth import synth_clips
    wave = synth_clips(1218, L=220500, seed=99, device=be.device)
    x = be.logmel(wave, 22050, 512, 128, 64, frame_major=True, l2norm=True)[:2097152].contiguous()
    runs = {}
    for label, use_filter in (("filtered", True), ("fp32", False)):
        switches(filter=use_filter)
        km = Kmeans(64, 8192, niter=4, backend=be)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            km.train(x)
            first = (km.centroids_device.clone(), [s["obj"] for s in km.iteration_stats], [s["nsplit"] for s in km.iteration_stats])
            km.train(x[:1000000], init_centroids=km.centroids_device)     # warm start: previous grouping, regrouped beside
        runs[label] = first + (km.centroids_device.clone(), [s["obj"] for s in km.iteration_stats])
    (ca, oa, sa, wa, woa), (cb, ob, sb, wb, wob) = runs["filtered"], runs["fp32"]
    assert torch.equal(ca.view(torch.int32), cb.view(torch.int32)) and torch.equal(wa.view(torch.int32), wb.view(torch.int32))
    assert oa == ob and sa == sb and woa == wob
    assert all(later <= earlier * (1 + 1e-6) for earlier, later in zip(oa[1:], oa[2:]))   # (iteration 1 may repair empties)


@pytest.mark.parametrize("sw", [{"filter_screen": 0}, {"filter_fused": 0}, {"filter_nb": 4},
                                {"filter_nb": 2, "filter_wps2": 1}, {"dmin_kernel": 0},
                                {"c2f_fused": 0}, {"filter": 0, "prune_nb": 1}, {"filter": 0, "prune_nb": 4}, {"filter_sync": 1},
                                {"assign_variant": 1}])
def test_ab_switches_leave_the_bits_alone(be, oracle, switches, sw):
    """Every A/B switch (include/at_debug.h, README.md) selects another route to the same ids and distances."""
    switches(**sw)
    rng = np.random.default_rng(5)
    n, d, k = 70000, 64, 2048
    centers = _unit_rows(rng, k, d, oracle)
    x = oracle.l2norm_rows((centers[rng.integers(0, k, n)] + 0.05 * rng.standard_normal((n, d))).astype(np.float32))
    c = oracle.l2norm_rows((centers + 0.01 * rng.standard_normal((k, d))).astype(np.float32))
    c[1000:1010] = c[0:10]
    ids_o, dis_o = oracle.assign(x, c)
    xt, ct = be._f32(x), be._f32(c)
    cperm = be.from_host(be.group_rows_kd(c))
    dmin = be.group_min_dist(ct, cperm)
    guess = torch.from_numpy(np.where(rng.random(n) < 0.8, ids_o, rng.integers(0, k, n))).to(be.device)
    ids, dis = be.assign_pruned(xt, ct, be.visit_order(guess.contiguous(), None, k), cperm, dmin)
    assert np.array_equal(ids.cpu().numpy(), ids_o) and np.array_equal(bits(dis.cpu().numpy()), bits(dis_o))
    ids, dis = be.assign_c2f(xt, ct, cperm, dmin, coherent=True)
    assert np.array_equal(ids.cpu().numpy(), ids_o) and np.array_equal(bits(dis.cpu().numpy()), bits(dis_o))


@pytest.mark.parametrize("ng,d,nnb", [(256, 64, 8), (256, 64, 4), (37, 128, 4), (3, 64, 8), (512, 64, 8)])
def test_group_neighbours_names_the_nearest_means(be, ng, d, nnb):
    """The neighbour table (a heuristic input, it never decides a result) names g itself and the groups
    whose means are nearest to it."""
    rng = np.random.default_rng(ng + nnb)
    m = rng.standard_normal((ng, d)).astype(np.float32)
    m[1] = m[0]                                                     # a tie: the lower index first
    bits_dev = be.group_neighbours(be.from_host(m), nnb).cpu().numpy().view(np.uint32)
    assert bits_dev.shape == (ng, (ng + 31) // 32)
    d2 = ((m[:, None, :].astype(np.float64) - m[None, :, :]) ** 2).sum(-1)
    for g in range(ng):
        chosen = [j for j in range(ng) if (bits_dev[g, j >> 5] >> (j & 31)) & 1]
        assert g in chosen and len(chosen) == min(nnb, ng)
        others = np.delete(np.arange(ng), g)
        kth = np.sort(d2[g, others])[min(nnb, ng) - 2] if min(nnb, ng) > 1 else 0.0
        assert all(j == g or d2[g, j] <= kth * (1 + 1e-5) for j in chosen)


@pytest.mark.parametrize("k,niter", [(2048, 3), (2048, 0), (64, 2)])
def test_kmeans_rejects_non_finite_input(be, oracle, k, niter):
    """faiss' isfinite check: the error comes before anything of the call is kept (pruned path at k = 2048,
    plain path at k = 64, no iteration at all), and the object stays usable."""
    from audio_tokens_amd.ops import Kmeans
    rng = np.random.default_rng(k + niter)
    x = oracle.l2norm_rows(rng.standard_normal((90000, 64)).astype(np.float32))
    km = Kmeans(64, k, niter=niter, backend=be)
    for bad_value, where in ((np.nan, 12345), (np.inf, 89999)):
        xb = x.copy()
        xb[where, 7] = bad_value
        with pytest.raises(RuntimeError, match="NaN's or Inf's"):
            km.train(xb)
        assert km.iteration_stats == [] and getattr(km, "centroids", None) is None
    km.train(x)                                                    # and the object is still usable
    assert km.centroids.shape == (k, 64) and np.isfinite(km.centroids).all()


def test_logmel_propagates_nonfinite_samples_like_the_oracle(be, oracle):
    """torch.clamp (AmplitudeToDB) leaves a NaN power NaN; so does the oracle; the kernels' `s > 1e-10 ? log : -100`
    used to turn it into -100 dB (found in round 2 through the fused non-finite verdict)."""
    import torch
    from audio_tokens_amd.synth import synth_clips
    wave = synth_clips(2, L=22050, seed=5, device="cuda")
    for bad in (float("nan"), float("inf")):
        w = wave.clone()
        w[1, 9000] = bad
        for n_fft in (512, 1024):
            got = be.logmel(w, n_fft=n_fft, hop=128, n_mels=64).cpu().numpy()
            want = oracle.logmel(w[1].cpu().numpy(), n_fft=n_fft, hop=128, n_mels=64)
            assert np.isfinite(got[0]).all()
            assert np.array_equal(np.isnan(got[1]), np.isnan(want)), (bad, n_fft)
            assert np.isnan(want).any()

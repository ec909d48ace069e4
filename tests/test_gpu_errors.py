"""Error reporting of the native layer (round 2's intermittent 'kernel launch failed: invalid resource handle').

ROCm 7.2 keeps the error of a failed HIP call pending in the calling thread until somebody calls hipGetLastError(),
whatever succeeds in between (tools/probes/event_probe.hip).  Round 2 checked its launches with a bare
hipGetLastError() and therefore blamed the next launch for any earlier failure nobody had consumed -- its own
hipEventElapsedTime on timing events among them.  These tests pin the rules of csrc/at_internal.h:
a launch is judged by hipLaunchKernel's own return code, a pending error is consumed, counted and (strict mode)
reported as what it is, and the statistics ring never reads a timing event that was not recorded."""
import numpy as np
import pytest
import torch

from audio_tokens_amd import _lib

pytestmark = pytest.mark.gpu


def _unit(rng, n, d):
    x = rng.standard_normal((n, d)).astype(np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True)


@pytest.fixture()
def strict(be):
    """(strict_errors is process-wide; the suite runs with it on -- conftest.py)"""
    before = be.debug_get("strict_errors")
    yield lambda v: be.debug_set("strict_errors", v)
    be.debug_set("strict_errors", before)
    be.diag_errors(reset=True)


def test_a_pending_error_is_not_blamed_on_the_next_launch(be, strict):
    rng = np.random.default_rng(0)
    x = be._f32(_unit(rng, 4096, 64) * 3.0)
    out = torch.empty_like(x)
    ref = be.l2norm_rows(x).clone()
    torch.cuda.synchronize()
    be.diag_errors(reset=True)

    # tolerant mode (the product's default): the call goes through, the stray error is consumed and counted
    strict(0)
    _lib.check(be.lib.at_debug_leave_error_pending())
    be.l2norm_rows(x, out=out)
    torch.cuda.synchronize()                       # (torch's own checks see a clean thread too)
    assert torch.equal(out, ref)
    dg = be.diag_errors()
    assert dg["stale_seen"] == 1 and dg["stale_last_code"] == 400, dg      # hipErrorInvalidResourceHandle
    assert "l2norm.hip" in dg["where"], dg

    # strict mode (the test-suite's): the call fails and says what it found, not "kernel launch failed"
    strict(1)
    _lib.check(be.lib.at_debug_leave_error_pending())
    with pytest.raises(_lib.NativeError, match="stale error from an earlier call.*invalid resource handle"):
        be.l2norm_rows(x, out=out)
    # ... and the error was consumed by that report: the next call is clean
    be.l2norm_rows(x, out=out)
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    assert be.diag_errors()["stale_seen"] == 2


def _exact_calls(be, xt, ct, cperm, k, rounds):
    ids, dis = be.assign(xt, ct)
    for _ in range(rounds):
        dmin = be.group_min_dist(ct, cperm)
        ids, dis = be.assign_pruned(xt, ct, be.visit_order(ids, dis, k), cperm, dmin, filter=True, image_current=True)
    return ids


def test_statistics_ring_reads_timing_events_only_when_they_were_recorded(be, switches, strict):
    """The path of the round-2 failure: exact calls push a slot per call; resolving a slot must not touch timing
    events that its call did not record (hipEventElapsedTime on such a pair fails with 'invalid resource handle'
    and used to stay pending until the next launch check)."""
    strict(1)
    rng = np.random.default_rng(1)
    n, d, k = 40000, 64, 2048
    x, c = _unit(rng, n, d), _unit(rng, k, d)
    xt, ct = be._f32(x), be._f32(c)
    cperm = be.from_host(be.group_rows_kd(c))
    ref = be.assign(xt, ct)[0]
    be.filter_stats()
    be.diag_errors(reset=True)

    switches(filter_timing=0)                       # the product's setting: no timing events at all
    assert torch.equal(_exact_calls(be, xt, ct, cperm, k, 5), ref)
    rows, listed, ms, sweeps, _, _ = be.filter_stats(timing=True)
    assert rows == 5 * n and sweeps == 0 and ms == 0.0

    switches(filter_timing=1)                       # bench.py's setting
    assert torch.equal(_exact_calls(be, xt, ct, cperm, k, 5), ref)
    rows, listed, ms, sweeps, _, _ = be.filter_stats(timing=True)
    assert rows == 5 * n and sweeps == 5 and ms > 0.0

    # the switch flipped while slots are pending, more calls than the ring has slots, and the synchronous form in between
    for r in range(3):
        switches(filter_timing=r & 1)
        assert torch.equal(_exact_calls(be, xt, ct, cperm, k, 40), ref)
        switches(filter_sync=1)
        assert torch.equal(_exact_calls(be, xt, ct, cperm, k, 2), ref)
        switches(filter_sync=0)
    rows, listed, ms, sweeps, _, _ = be.filter_stats(timing=True)
    assert rows == 3 * 42 * n and sweeps == 42       # the timed round only (r = 1): 40 asynchronous + 2 synchronous calls
    dg = be.diag_errors()
    assert dg["tolerated"] == 0 and dg["stale_seen"] == 0, dg


def test_whole_training_leaves_no_error_behind(be, strict):
    """The product call chain of processors/cluster_creator.py:49-56 (the one the round-2 traceback names) in strict
    mode: cold and warm training with every helper stream and thread in play, then nothing pending, nothing tolerated."""
    from audio_tokens_amd.ops import Kmeans
    strict(1)
    be.diag_errors(reset=True)
    rng = np.random.default_rng(2)
    x = be._f32(_unit(rng, 300000, 64))
    km = Kmeans(64, 1024, niter=20, backend=be)
    km.train(x)
    km.train(x, init_centroids=km.centroids_device)
    torch.cuda.synchronize()
    dg = be.diag_errors()
    assert dg["stale_seen"] == 0 and dg["tolerated"] == 0, dg

"""The sharded k-means on real HIP kernels with world_size 2.

With two or more GPUs visible the ranks take one device each and exchange their partials over RCCL
(torch.distributed backend "nccl"): the production path of an N-GPU run.  On a one-GPU box the two ranks
share cuda:0 and exchange through gloo (host-staged), which exercises the same host logic and the same
kernels.  Either way the result must equal the oracle's two-shard variant, bit for bit, on both ranks."""
import os
import socket
import warnings
from pathlib import Path

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
G = Path(__file__).resolve().parent / "golden"


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    use_rccl = torch.cuda.device_count() >= world          # (counting devices does not initialise the GPU)
    dev = rank if use_rccl else 0
    torch.cuda.set_device(dev)
    if use_rccl:   # the process group first: RCCL binds the rank to its device
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from audio_tokens_amd.ops import Kmeans
        from audio_tokens_amd.pipeline import DevicePipeline
        from audio_tokens_amd.synth import synth_clips
        g = np.load(G / "kmeans.npz")
        out = {}
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            x, cut = g["a_x"], 1000
            km = Kmeans(64, 64, niter=20, distributed=True)
            km.train(x[:cut] if rank == 0 else x[cut:])
            out["plain"] = km.centroids.copy()
            x, cut = g["c_x"], 12345
            kc = Kmeans(8, 64, niter=5, distributed=True)
            kc.train(x[:cut] if rank == 0 else x[cut:], init_centroids=g["c_init"])
            out["sub"] = kc.centroids.copy()
            # big enough for the exact pruned sweep (k >= 1024, >= 4096 rows per rank)
            rng = np.random.default_rng(5)
            cen = rng.standard_normal((1024, 64))
            xb = (cen[rng.integers(0, 1024, 30000)] + 0.3 * rng.standard_normal((30000, 64))).astype(np.float32)
            xb /= np.linalg.norm(xb, axis=1, keepdims=True)
            kp = Kmeans(64, 1024, niter=6, distributed=True)
            assert kp.prune
            kp.train(xb[:14000] if rank == 0 else xb[14000:])
            out["pruned"] = kp.centroids.copy()
            ks = Kmeans(64, 1024, niter=6, distributed=True)
            ks.exchange = "scatter"          # all-to-all + ordered local sum + all-gather instead of the all-gather
            ks.train(xb[:14000] if rank == 0 else xb[14000:])
            out["pruned_scatter"] = ks.centroids.copy()
            # whole device pipeline, 2 ranks x 3 clips
            wave = synth_clips(6, L=22050 * 2, seed=11, device="cuda")
            mine = wave[rank * 3:(rank + 1) * 3]
            res = DevicePipeline(n_mels=64, vocab_size=32, niter=4, clustering_batch_size=6,
                                 distributed=True).run(mine[:2], mine[2:])
            out["pipe_centroids"] = res.centroids.cpu().numpy()
            out["pipe_tokens"] = res.tokens_train.cpu().numpy()
            out["backend"] = dist.get_backend()
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_two_ranks_one_gpu(oracle, be):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    import torch
    assert res[0]["backend"] == res[1]["backend"] == ("nccl" if torch.cuda.device_count() >= 2 else "gloo")
    g = np.load(G / "kmeans.npz")
    assert np.array_equal(bits(res[0]["plain"]), bits(res[1]["plain"]))
    assert np.array_equal(bits(res[0]["plain"]), bits(g["d_centroids"]))
    shard = (np.arange(20000) >= 12345).astype(np.int32)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        r = oracle.kmeans_train(g["c_x"], 64, niter=5, init_centroids=g["c_init"], shard=shard, n_shards=2)
    assert np.array_equal(bits(res[0]["sub"]), bits(r.centroids)) and np.array_equal(bits(res[1]["sub"]), bits(r.centroids))
    rng = np.random.default_rng(5)
    cen = rng.standard_normal((1024, 64))
    xb = (cen[rng.integers(0, 1024, 30000)] + 0.3 * rng.standard_normal((30000, 64))).astype(np.float32)
    xb /= np.linalg.norm(xb, axis=1, keepdims=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        rp = oracle.kmeans_train(xb, 1024, niter=6, shard=(np.arange(30000) >= 14000).astype(np.int32), n_shards=2)
    assert np.array_equal(bits(res[0]["pruned"]), bits(rp.centroids)) and np.array_equal(bits(res[1]["pruned"]), bits(rp.centroids))
    assert np.array_equal(bits(res[0]["pruned_scatter"]), bits(rp.centroids)) and np.array_equal(bits(res[1]["pruned_scatter"]), bits(rp.centroids))
    # pipeline: both ranks hold the same centroids; the oracle reproduces them from the same frames
    assert np.array_equal(bits(res[0]["pipe_centroids"]), bits(res[1]["pipe_centroids"]))
    from audio_tokens_amd.synth import synth_clips
    wave = synth_clips(6, L=22050 * 2, seed=11, device="cuda")
    frames = be.logmel(wave, frame_major=True, l2norm=True).cpu().numpy()
    T = frames.shape[0] // 6
    # global batch = rank 0's two train clips then rank 1's two train clips (clips 0,1,3,4)
    xb = np.concatenate([frames[0:2 * T], frames[3 * T:5 * T]])
    sh = np.repeat(np.array([0, 1], np.int32), 2 * T)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ro = oracle.kmeans_train(xb, 32, niter=4, shard=sh, n_shards=2)
    cent = oracle.l2norm_rows(ro.centroids)
    assert np.array_equal(bits(res[0]["pipe_centroids"]), bits(cent))
    ids, _ = oracle.assign(frames, cent)
    assert np.array_equal(res[0]["pipe_tokens"], ids[0:2 * T]) and np.array_equal(res[1]["pipe_tokens"], ids[3 * T:5 * T])


def _rccl_single_rank_worker(port, q):
    import sys
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        import audio_tokens_amd.ops as ops
        from audio_tokens_amd.backend import default_backend
        be = default_backend()

        class OneRankRccl(ops._Dist):   # a one-rank group still goes through every collective of the sharded path
            def __init__(self, enabled, group=None):
                super().__init__(enabled, group)
                self.on, self.world, self.rank, self.host_staged = bool(enabled), 1, 0, False

        d = OneRankRccl(True)
        out = {"backend": dist.get_backend()}
        v = torch.randn(100003, device="cuda")
        out["gather"] = bool(torch.equal(d.all_gather_parts(v)[0], v))
        out["scatter"] = bool(torch.equal(d.reduce_in_rank_order(be, v), v))
        out["sizes"] = d.all_gather_sizes(77, be.device)
        out["flag"] = d.any_flag(True, be.device)
        out["f64"] = float(d.all_gather_f64(torch.tensor([1.25], dtype=torch.float64, device="cuda"))[0])
        rows = torch.randn(5, 8, device="cuda")
        out["bits"] = bool(torch.equal(d.sum_bits(rows.clone()), rows))
        # and a whole training through the distributed code path (device collectives, no host staging)
        rng = np.random.default_rng(3)
        x = rng.standard_normal((50000, 64)).astype(np.float32)
        x /= np.linalg.norm(x, axis=1, keepdims=True)
        plain = ops.Kmeans(64, 1024, niter=5)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            plain.train(x)
            ops._Dist = OneRankRccl
            for exchange in ("gather", "scatter"):
                km = ops.Kmeans(64, 1024, niter=5, distributed=True)
                km.exchange = exchange
                km.train(x)
                out[exchange + "_train"] = bool(np.array_equal(bits(km.centroids), bits(plain.centroids)))
        q.put(out)
    finally:
        dist.destroy_process_group()


def test_rccl_collectives_execute_with_one_rank(be):
    """The pool's boxes have one GPU, and RCCL refuses two ranks on one device, so the two-rank test above runs over
    gloo there.  This one makes the `nccl` branch of ops._Dist execute anyway: a one-rank RCCL group, device tensors
    through every collective the sharded k-means uses (all-gather, all-to-all, all-reduce), and a whole training
    through the distributed code path in both exchange forms -- which with one shard must equal the plain training."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_single_rank_worker, args=(_free_port(), q))
    p.start()
    out = q.get(timeout=300)
    p.join(60)
    assert p.exitcode == 0
    assert out["backend"] == "nccl"
    assert out["gather"] and out["scatter"] and out["bits"] and out["flag"] is True
    assert out["sizes"] == [77] and out["f64"] == 1.25
    assert out["gather_train"] and out["scatter_train"]


def _rccl_cabi_worker(q):
    import ctypes
    import sys
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    from audio_tokens_amd import _lib
    from audio_tokens_amd.backend import default_backend
    torch.cuda.set_device(0)
    be = default_backend()
    # the communicator is the CALLER's: a one-rank one made with the same RCCL the wrappers will find (loaded globally
    # first, so that their lookup among the process's symbols resolves to this copy)
    rccl = ctypes.CDLL("/opt/rocm/lib/librccl.so", mode=ctypes.RTLD_GLOBAL)

    class UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]

    uid = UniqueId()
    assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
    comm = ctypes.c_void_p()
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
    try:
        n = 100003
        part = torch.randn(n, device="cuda")
        parts = torch.full((n,), -1.0, device="cuda")
        out = torch.full((n,), -2.0, device="cuda")
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        ptr = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731
        _lib.check(be.lib.at_comm_allgather_f32(be.ctx.handle, comm, ptr(part), ptr(parts), n, stream))
        _lib.check(be.lib.at_comm_allreduce_ordered_f32(be.ctx.handle, comm, ptr(part), ptr(parts), ptr(out), n, stream))
        torch.cuda.synchronize()
        q.put({"gather": bool(torch.equal(parts, part)), "reduce": bool(torch.equal(out, part))})
    finally:
        rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        rccl.ncclCommDestroy(comm)


def test_rccl_wrappers_of_the_c_abi_with_one_rank(be):
    """at_comm_allgather_f32 / at_comm_allreduce_ordered_f32 (the exchange for hosts without torch.distributed) on a
    one-rank communicator the test creates with RCCL itself: with one rank both must return the rank's own part."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_cabi_worker, args=(q,))
    p.start()
    out = q.get(timeout=300)
    p.join(60)
    assert p.exitcode == 0
    assert out == {"gather": True, "reduce": True}

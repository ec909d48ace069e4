"""MI355X-native stand-ins for the three third-party operators the reference's stage classes call.

    torchaudio.transforms.MelSpectrogram + AmplitudeToDB  ->  LogMelSpectrogram
        (processors/spectrogram_generator.py:28-34,123-126 of danavery/audio-tokens)
    torchaudio.transforms.Resample                        ->  Resample
        (processors/spectrogram_generator.py:117-121)
    faiss.Kmeans                                          ->  Kmeans
        (processors/cluster_creator.py:42-56)
    faiss.IndexFlatL2                                     ->  IndexFlatL2
        (processors/spec_tokenizer.py:123-127, 77)

Same constructor arguments, method names, return types and error behaviour as the originals for the
subset the reference uses.  All arithmetic happens in libaudio_tokens_amd.so (HIP, gfx950); this
file is the host-side orchestration: the FAISS training recipe (subsample permutation, random
initialisation, 20 Lloyd iterations, empty-cluster repair) and, when torch.distributed is
initialised and `distributed=True`, the data-parallel variant in which every rank owns a block of
rows and the per-cluster partial sums/counts are exchanged once per iteration.
"""
from __future__ import annotations

import collections
import os
import sys
import time
from collections import OrderedDict

import numpy as np
import torch

from .backend import default_backend

__all__ = ["LogMelSpectrogram", "Resample", "Kmeans", "IndexFlatL2", "normalize_rows"]


def _is_host(x) -> bool:
    return isinstance(x, np.ndarray) or (isinstance(x, torch.Tensor) and x.device.type == "cpu")


def normalize_rows(x, backend=None):
    """x / (||x||_2 + 1e-10) row-wise, bit-identical to the reference's numpy expression
    (cluster_creator.py:64-66).  numpy in -> numpy out; device tensor in -> device tensor out."""
    be = backend or default_backend()
    host = _is_host(x)
    y = be.l2norm_rows(x)
    return be.to_host(y) if host else y


class LogMelSpectrogram:
    """MelSpectrogram(sample_rate, n_mels, n_fft, hop_length) followed by AmplitudeToDB(), fused.

    __call__(waveform [..., L]) -> [..., n_mels, T] float32 in dB (torchaudio's layout)."""

    def __init__(self, sample_rate=22050, n_fft=512, hop_length=128, n_mels=64, fb=None, backend=None):
        self.sample_rate, self.n_fft, self.hop_length, self.n_mels = sample_rate, n_fft, hop_length, n_mels
        self.backend = backend or default_backend()
        self.fb = None if fb is None else self.backend._f32(fb)

    def to(self, device):  # torchaudio-style chaining; the backend already pins the device
        return self

    def __call__(self, waveform):
        be = self.backend
        w = be._f32(waveform)
        lead = w.shape[:-1]
        out = be.logmel(w.reshape(-1, w.shape[-1]), self.sample_rate, self.n_fft, self.hop_length,
                        self.n_mels, fb=self.fb)
        return out.reshape(*lead, self.n_mels, out.shape[-1])

    def frames(self, waveform, l2norm=False):
        """[n_clips, L] -> frame-major [n_clips*T, n_mels] (optionally row-normalised): the matrix
        ClusterCreator / SpecTokenizer build from the .npy files, without the round trip."""
        be = self.backend
        return be.logmel(be._f32(waveform), self.sample_rate, self.n_fft, self.hop_length, self.n_mels,
                         fb=self.fb, frame_major=True, l2norm=l2norm)


class Resample:
    """torchaudio.transforms.Resample(orig_freq, new_freq) with its defaults (sinc_interp_hann,
    lowpass_filter_width 6, rolloff 0.99).  __call__(waveform [..., L]) -> [..., ceil(L*new/orig)]
    on the device; equal rates return the input unchanged, as torchaudio does."""

    def __init__(self, orig_freq=16000, new_freq=16000, backend=None):
        if int(orig_freq) != orig_freq or int(new_freq) != new_freq:
            raise ValueError("Frequencies must be of integer type to ensure quality resampling computation.")
        self.orig_freq, self.new_freq = int(orig_freq), int(new_freq)
        if self.orig_freq <= 0 or self.new_freq <= 0:
            raise ValueError("Original frequency and desired frequecy should be positive")
        self.backend = backend or default_backend()

    def to(self, device):
        return self

    def __call__(self, waveform):
        if self.orig_freq == self.new_freq:
            return waveform
        be = self.backend
        w = be._f32(waveform)
        lead = w.shape[:-1]
        out = be.resample(w.reshape(-1, w.shape[-1]), self.orig_freq, self.new_freq)
        return out.reshape(*lead, out.shape[-1])


class _Dist:
    """Thin view of torch.distributed for the sharded k-means (one process per GPU, RCCL).

    With the gloo backend (CPU tests, or several test ranks sharing one GPU) device tensors are
    staged through the host; with nccl (= RCCL) the collectives run on the device tensors."""

    def __init__(self, enabled, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.on = bool(enabled) and dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self.group = group
        self.world = dist.get_world_size(group) if self.on else 1
        self.rank = dist.get_rank(group) if self.on else 0
        self.host_staged = self.on and dist.get_backend(group) == "gloo"

    def _all_reduce(self, t, op):
        if self.host_staged and t.device.type != "cpu":
            h = t.cpu()
            self.dist.all_reduce(h, op=op, group=self.group)
            t.copy_(h)
        else:
            self.dist.all_reduce(t, op=op, group=self.group)
        return t

    def _all_gather(self, t):
        """t [m] -> [world * m], rank order."""
        if self.host_staged and t.device.type != "cpu":
            h = t.cpu()
            out = torch.empty(self.world * h.numel(), dtype=h.dtype)
            self.dist.all_gather_into_tensor(out, h.reshape(-1), group=self.group)
            return out.to(t.device)
        out = torch.empty(self.world * t.numel(), dtype=t.dtype, device=t.device)
        self.dist.all_gather_into_tensor(out, t.reshape(-1), group=self.group)
        return out

    def all_gather_sizes(self, n_loc, device):
        if not self.on:
            return [n_loc]
        out = self._all_gather(torch.tensor([n_loc], dtype=torch.int64, device=device))
        return [int(v) for v in out.cpu()]

    def any_flag(self, flag: bool, device) -> bool:
        if not self.on:
            return flag
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device)
        return bool(self._all_reduce(t, self.dist.ReduceOp.MAX).item())

    def all_gather_parts(self, part):
        """part [m] float32 -> [world, m], rank order."""
        if not self.on:
            return part.unsqueeze(0)
        return self._all_gather(part).view(self.world, part.numel())

    def reduce_in_rank_order(self, be, v):
        """v [m] float32 -> the sum over the ranks of v, added in ascending rank order, identical on every rank:
        all-to-all of the ranks' slices, ordered local sum of the slice this rank owns (at_sum_parts_f32), all-gather
        of the reduced slices.  Moves 2 (N-1)/N m floats per rank where the all-gather of whole partials moves
        (N-1) m -- the form for large tables (BASELINE.json's configs[4]: 8.45 MB per partial)."""
        if not self.on:
            return v
        w, m = self.world, v.numel()
        chunk = -(-m // w)
        chunk = (chunk + 3) // 4 * 4
        host = self.host_staged and v.device.type != "cpu"
        src = torch.zeros(w * chunk, dtype=v.dtype, device="cpu" if host else v.device)
        src[:m] = v.cpu() if host else v
        recv = torch.empty_like(src)
        self.dist.all_to_all_single(recv, src, group=self.group)          # recv[r] = rank r's copy of MY slice
        mine = be.sum_parts((recv.to(v.device) if host else recv).view(w, chunk))
        mine = mine.cpu() if host else mine
        out = torch.empty(w * chunk, dtype=v.dtype, device=mine.device)
        self.dist.all_gather_into_tensor(out, mine, group=self.group)
        out = out.to(v.device) if host else out
        return out[:m]

    def all_gather_f64(self, v):
        """v float64 [1] -> [world] in rank order."""
        if not self.on:
            return v.reshape(1)
        return self._all_gather(v.reshape(-1))

    def sum_bits(self, rows):
        """Exact merge of float32 rows of which exactly one rank holds a non-zero copy."""
        if not self.on:
            return rows
        bits = rows.view(torch.int32)
        return self._all_reduce(bits, self.dist.ReduceOp.SUM).view(torch.float32)

    def sum_f64(self, v):
        return self._all_reduce(v, self.dist.ReduceOp.SUM) if self.on else v


class Kmeans:
    """faiss.Kmeans(d, k, niter=, verbose=, gpu=) for the reference's use (cluster_creator.py:42-48).

    ClusteringParameters are FAISS's defaults: nredo=1, seed=1234, max_points_per_centroid=256,
    min_points_per_centroid=39, no spherical / int / frozen centroids.  Each train() call is one
    faiss Clustering::train: subsample to k*256 rows with rand_perm(n, seed), initialise from
    `init_centroids` or from rand_perm(n_sub, seed+1), then niter x {nearest centroid, objective,
    ascending-index centroid sums, 1/count, split_clusters}.

    distributed=True (and torch.distributed initialised): `x` is this rank's block of the global
    row-concatenation in rank order; the result is identical on every rank.
    """

    def __init__(self, d, k, niter=20, verbose=False, gpu=True, seed=1234, max_points_per_centroid=256,
                 min_points_per_centroid=39, distributed=False, process_group=None, backend=None, **kwargs):
        unsupported = {kk: v for kk, v in kwargs.items() if kk not in ("nredo",) or v != 1}
        if unsupported:
            raise NotImplementedError(f"Kmeans: unsupported ClusteringParameters {sorted(unsupported)}")
        self.d, self.k, self.niter, self.verbose = int(d), int(k), int(niter), bool(verbose)
        self.gpu = gpu  # accepted for signature compatibility; this implementation is GPU-only
        self.seed = int(seed)
        self.max_points_per_centroid = int(max_points_per_centroid)
        self.min_points_per_centroid = int(min_points_per_centroid)
        self.backend = backend or default_backend()
        self._dist_enabled, self._group = distributed, process_group
        self.centroids_device = None   # [k, d] device tensor after train(); `.centroids` is its numpy copy (lazy)
        self._centroids_host = None
        self._iteration_stats, self._stats_pending = [], None
        self.index = None
        self.phase_seconds = None
        self.prune = True  # exact pruning of Lloyd iterations 2..niter (d = 64 / 128 only)
        self.order_beside = True   # the visiting-order sort on a stream of its own (A/B aid)
        self.exchange = "auto"     # "gather" | "scatter" | "auto": how the per-iteration partials are combined (see train)

    # ------------------------------------------------------------------------------------
    def train(self, x, init_centroids=None, sync=True, check_finite=True):
        """-> final objective (float), like faiss.  sync=False returns None without waiting for the device:
        centroids_device is valid in stream order, `.centroids` / `.obj` / `.iteration_stats` wait when read.
        check_finite=False skips faiss' NaN/Inf scan of x (a host round trip) for callers that have made it."""
        be = self.backend
        k, d = self.k, self.d
        dist = _Dist(self._dist_enabled, self._group)
        if isinstance(x, np.ndarray):
            assert x.flags.c_contiguous, "x must be C-contiguous"  # faiss asserts the same
        x = be._f32(x)
        assert x.dim() == 2 and x.shape[1] == d, f"expected [n, {d}], got {tuple(x.shape)}"
        n_loc = x.shape[0]
        sizes = dist.all_gather_sizes(n_loc, be.device)
        off = sum(sizes[:dist.rank])
        n = sum(sizes)
        if n < k:
            raise RuntimeError(f"Error: 'nx >= k' failed: Number of training points ({n}) should be at "
                               f"least as large as number of clusters ({k})")
        if check_finite and dist.any_flag(be.any_nonfinite(x) if n_loc else False, be.device):
            raise RuntimeError("Error: 'std::isfinite(x_in[i])' failed: input contains NaN's or Inf's")

        # ---- subsample_training_set -------------------------------------------------------
        if n > k * self.max_points_per_centroid:
            ns = k * self.max_points_per_centroid
            if self.verbose:
                print(f"Sampling a subset of {ns} / {n} for training")
            perm = self._subsample_perm(n, ns)      # device int32 [ns]: faiss' rand_perm(n, seed)[:ns]
            if dist.on:
                mine = (perm >= off) & (perm < off + n_loc)
                pos = torch.nonzero(mine).reshape(-1)              # subsample positions this rank owns (ascending)
                xs = be.gather_rows(x, (perm[mine] - off).to(torch.int32).contiguous())
            else:
                pos = None
                xs = be.gather_rows(x, perm)
        else:
            ns = n
            if ns < k * self.min_points_per_centroid:
                print(f"WARNING clustering {ns} points to {k} centroids: please provide at least "
                      f"{k * self.min_points_per_centroid} training points", file=sys.stderr)
            xs = x
            pos = off + torch.arange(n_loc, dtype=torch.int64, device=x.device) if dist.on else None

        def rows_at(positions):
            """Rows of the (global) subsample at `positions` [m] -> device [m, d], on every rank."""
            positions = np.asarray(positions, dtype=np.int64)
            if not dist.on:
                return be.gather_rows(xs, positions.astype(np.int32))
            out = be.zeros((len(positions), d))
            want = be.from_host(positions)
            if pos.numel():
                j = torch.searchsorted(pos, want).clamp_(max=pos.numel() - 1)
                ok = pos[j] == want
                if bool(ok.any()):
                    out[ok] = be.gather_rows(xs, j[ok].to(torch.int32).contiguous())
            return dist.sum_bits(out)

        self._stats_pending = None
        self._iteration_stats = []
        if ns == k:  # faiss corner case: the training set becomes the centroids
            cent = rows_at(np.arange(k))
            self._iteration_stats.append(dict(obj=0.0, time=0.0, time_search=0.0, imbalance_factor=1.0, nsplit=0))
            return self._finish(cent, sync)

        if init_centroids is not None:
            cent = be._f32(init_centroids).clone()
            assert tuple(cent.shape) == (k, d), f"init_centroids must be [{k}, {d}]"
        else:
            cent = rows_at(be.rand_perm_prefix(ns, self.seed + 1, k))

        # ---- Lloyd iterations -------------------------------------------------------------
        # Nothing below waits for the device: empty clusters are repaired by a device kernel (same draws, same
        # bits as faiss' host loop) and {objective, imbalance, nsplit} of every iteration stay in two small device
        # arrays that are read once, after the last iteration (or per iteration when verbose).
        niter = self.niter
        stats_dev = be.zeros((max(niter, 1), 2), torch.float64)
        nsplit_dev = be.zeros((max(niter, 1),), torch.int32)
        obj_off, part_len = be.part_layout(k, d)
        # the exchange: an all-gather of whole partials (one collective, (N-1) x the table per rank) up to 4 MB,
        # all-to-all + ordered local sum + all-gather (2 (N-1)/N x the table) above; both add in rank order
        scatter_exchange = dist.on and (self.exchange == "scatter" or (self.exchange == "auto" and part_len * 4 > (4 << 20)))
        t0 = time.time()
        prof = self.phase_seconds  # None, or a dict that collects per-phase wall time (debug aid)

        def lap(name, t_prev):
            if prof is None:
                return t_prev
            be.synchronize()
            now = time.perf_counter()
            prof[name] = prof.get(name, 0.0) + (now - t_prev)
            return now

        # Iterations after the first reuse the previous assignment as a guess.  With d = 64/128 the
        # sweep is also pruned (exact: see csrc/prune.hip); the spatial grouping of the centroids it
        # relies on is computed once per train() -- it only affects how much gets skipped.
        prune = (self.prune and hasattr(be, "assign_pruned") and d in (64, 128) and k >= 1024
                 and (k + 31) // 32 <= 512 and xs.shape[0] >= 4096)
        ids = dis = order = vorder = None
        # With few rows per cluster on this rank (sharded runs) a cluster fills a tile or two and sorting its
        # rows by distance buys nothing: the member-list order of the accumulation doubles as the visiting
        # order and the second sort of the iteration is dropped.
        member_order = prune and xs.shape[0] < 96 * k   # (measured: -8 % per iteration at 32 rows per cluster)
        regrouping = None   # host grouping of newer centroids, under way on the helper thread
        regrouping_is_late = False

        def regroup_beside(c):
            """Host grouping of the centroids `c` on the helper thread; the copy to the host is queued on the
            stream and waited for there, not here."""
            h, ready = be.to_host_async(c)

            def job():
                ready.synchronize()
                return be.group_rows_kd(h.numpy())
            return _grouper().submit(job)

        if prune:
            # The spatial grouping only decides how much the exact sweep can skip.  A warm start begins with
            # the grouping the previous train() ended with while the host regroups the new initial centroids
            # beside the first iterations; a cold start groups its initial centroids before it can begin.
            upload = getattr(be, "from_host_async", be.from_host)   # (the small table goes up without draining the stream)
            cached = getattr(self, "_cperm_cache", None)
            if init_centroids is not None and cached is not None and cached[0] == (k, d):
                cperm = cached[1]
                # (continuing from this object's own result: that grouping is of centroids a few iterations
                # older than these, nothing to redo)
                if not (init_centroids is getattr(self, "centroids_device", None)
                        or init_centroids is self.__dict__.get("_centroids_host")):
                    regrouping = regroup_beside(cent)
            else:
                cperm = be.from_host(be.group_rows_kd(be.to_host(cent)))
            gnbr = None
        # a last regrouping near the end: the next warm start and the tokeniser's index begin with it
        late_regroup = niter - 6 if niter >= 10 else -1
        # The host queues an iteration several times faster than the device runs it.  Left alone it would be a whole
        # training ahead, and a regrouping -- which needs the device to have reached the centroids it groups, plus
        # ~3 ms of host work -- would arrive after the host had queued every iteration that could have used it.  So
        # the host stays at most three iterations ahead (waiting on an event three iterations old: the device always
        # has queued work), measures the device's iteration time from those events, and submits the late regrouping
        # early enough to be finished when the last iteration is (short iterations: a shard of an N-GPU run).
        paced = collections.deque()
        iter_ms = None
        last_done = None
        timed_events = hasattr(be, "record_event_timed")

        def late_due(it):
            if late_regroup < 0 or it < niter // 2:
                return False
            return it >= late_regroup or (iter_ms is not None and (niter - it) * iter_ms <= 4.5)

        def pruned_assign(it):
            """Queues iteration `it`'s exact search over the current centroids -> (ids, dis)."""
            nonlocal cperm, regrouping, regrouping_is_late, gnbr
            if regrouping is not None and regrouping.done():
                cperm = upload(regrouping.result())
                regrouping = None
            elif (it == 2 and init_centroids is None and regrouping is None) or (late_due(it) and not regrouping_is_late):
                # cold start: regroup once the centroids have settled (taken up when the host is done); near the end:
                # for the next warm start and the tokeniser (replaces one still under way: these centroids are newer)
                regrouping_is_late = late_due(it)
                regrouping = regroup_beside(cent)
            dmin = be.group_min_dist(cent, cperm)
            if ids is None:   # no previous assignment yet: coarse-to-fine exact search
                gnbr = be.group_neighbours(be.group_means(cent, cperm), 8)
                return be.assign_c2f(xs, cent, cperm, dmin, gnbr)
            return be.assign_pruned(xs, cent, vorder if vorder is not None else be.visit_order(ids, dis, k),
                                    cperm, dmin, image_current=True)

        for it in range(niter):
            tp = time.perf_counter()
            if prune:
                ids, dis = pruned_assign(it)
            elif ids is None:
                ids, dis = be.assign(xs, cent)
            else:  # same answer, guided by the previous assignment and its member-list order
                ids, dis = be.assign_hinted(xs, cent, ids, order)
            tp = lap("assign", tp)
            part = be.empty((part_len,))
            obj_view = part[obj_off:obj_off + 2].view(torch.float64)             # this rank's objective rides along
            beside = prune and not member_order and self.order_beside and it + 1 < niter
            sjoin = None
            if not beside:
                if prune and hasattr(be, "sum_beside"):
                    sjoin = be.sum_beside(dis, obj_view)      # beside the accumulation; joined before the partial is used
                else:
                    be.sum_f64(dis, out=obj_view)
            if prune:
                # the next iteration's visiting order depends on this assignment only: its sort runs behind
                # the short-list accumulation while the long lists are still being summed on the side stream
                if member_order:
                    part, vorder = be.centroid_accum(xs, ids, k, out=part, want_order=True, defer_join=True)
                elif beside:
                    # a third stream, beside both accumulations: the visiting order of the next iteration and the objective
                    vjoin = be.visit_order_beside(ids, dis, k, sum_out=obj_view)
                    be.centroid_accum(xs, ids, k, out=part, defer_join=True)
                    vorder = vjoin()
                else:
                    be.centroid_accum(xs, ids, k, out=part, defer_join=True)
                    vorder = be.visit_order(ids, dis, k) if it + 1 < niter else None
                be.centroid_accum_join()
                if sjoin is not None:
                    sjoin()
            else:
                part, order = be.centroid_accum(xs, ids, k, out=part, want_order=True)
            tp = lap("accumulate", tp)
            if scatter_exchange:
                # (N-1)/N of the partial out, the reduced table back: sums and counts added in rank order by
                # at_sum_parts_f32 on the slice each rank owns; the objectives travel as N doubles
                red = be.empty((1, part_len))
                red[0, :obj_off] = dist.reduce_in_rank_order(be, part[:obj_off])
                red[0, obj_off:] = 0.0
                objs = dist.all_gather_f64(part[obj_off:obj_off + 2].view(torch.float64))
                cent, hassign = be.centroid_finalize(red, k, d)
                parts, objs = red, objs.contiguous()
            else:
                parts, objs = dist.all_gather_parts(part), None
                cent, hassign = be.centroid_finalize(parts, k, d)
            if hasattr(be, "lloyd_stats_split"):    # statistics + repair of empty clusters: two single-workgroup passes, one launch
                be.lloyd_stats_split(hassign, cent, ns, nsplit_dev[it:it + 1], parts, stats_dev[it], objs=objs)
            else:
                be.lloyd_stats(hassign, parts, k, d, stats_dev[it], objs=objs)
                be.split_clusters_device(hassign, cent, ns, nsplit_dev[it:it + 1])
            tp = lap("exchange+finalize+split", tp)
            if prune and timed_events:
                paced.append(be.record_event_timed())
                if len(paced) > 3:
                    done = paced.popleft()
                    done.synchronize()
                    if last_done is not None:
                        iter_ms = last_done.elapsed_time(done)
                    last_done = done
        self._last_assign = ids
        self._stats_pending = (stats_dev[:niter], nsplit_dev[:niter], t0)
        if self.verbose:   # faiss' per-iteration lines, printed once the iterations are through (no wait inside the loop)
            for it, st in enumerate(self.iteration_stats):
                print(f"  Iteration {it} ({st['time']:.2f} s, search {st['time_search']:.2f} s): "
                      f"objective={st['obj']:g} imbalance={st['imbalance_factor']:.3f} nsplit={st['nsplit']}", flush=True)
        if prune:
            if regrouping is not None and (regrouping.done() or regrouping_is_late):
                cperm = upload(regrouping.result())     # (submitted early enough to be done, or about to be)
            self._cperm_cache = ((k, d), cperm)
            self._grouping_of_result = cperm
        return self._finish(cent, sync)

    # -- results ----------------------------------------------------------------------------
    def _subsample_perm(self, n, m):
        """faiss' rand_perm(n, seed)[:m] on the device.  A pure function of (n, seed, m); the file batches of one
        run (one Kmeans object, as in ClusterCreator.run) mostly share n, so the last two are kept."""
        key = (int(n), self.seed, int(m))
        cache = self.__dict__.setdefault("_perm_cache", OrderedDict())
        hit = cache.get(key)
        if hit is None:
            hit = cache[key] = self.backend.rand_perm_prefix_device(n, self.seed, m)
            while len(cache) > 2:
                cache.popitem(last=False)
        return hit

    def _read_stats(self, stats_dev, nsplit_dev, t0):
        be = self.backend
        st, ns = be.to_host(stats_dev), be.to_host(nsplit_dev)     # (waits for the iterations queued so far)
        if (ns < 0).any():
            raise RuntimeError("Kmeans: split_clusters found no donor (every cluster has at most one point)")
        el = time.time() - t0
        return [dict(obj=float(np.float32(st[i, 0])), time=el * (i + 1) / len(ns), time_search=el * (i + 1) / len(ns),
                     imbalance_factor=float(st[i, 1]), nsplit=int(ns[i])) for i in range(len(ns))]

    @property
    def iteration_stats(self):
        """faiss' ClusteringIterationStats of the last train() (read back from the device on first use; `time`
        and `time_search` are the elapsed time spread evenly, the iterations are not timed one by one)."""
        if self._stats_pending is not None:
            self._iteration_stats = self._read_stats(*self._stats_pending)
            self._stats_pending = None
        return self._iteration_stats

    @property
    def obj(self):
        return np.array([s["obj"] for s in self.iteration_stats], dtype=np.float32)

    @property
    def centroids(self):
        """numpy [k, d], like faiss (copied from the device on first use)."""
        if self._centroids_host is None and self.centroids_device is not None:
            self._centroids_host = self.backend.to_host(self.centroids_device)
        return self._centroids_host

    def lend_grouping(self, table) -> None:
        """Offers the spatial grouping this object ended with to an IndexFlatL2 later filled with `table`
        (e.g. the normalised centroids a tokeniser loads): nearby rows stay nearby, and the grouping only
        ever decides how much the exact search skips."""
        cached = getattr(self, "_cperm_cache", None)
        be = self.backend
        if cached is not None and hasattr(be, "remember_grouping"):
            table = be.to_host(table) if isinstance(table, torch.Tensor) else np.asarray(table)
            if table.shape == cached[0]:
                be.remember_grouping(table, cached[1])

    def _finish(self, cent, sync=True):
        self.centroids_device = cent
        self._centroids_host = None
        self.index = IndexFlatL2(self.d, backend=self.backend)
        self.index.add(cent)
        be = self.backend
        if getattr(self, "_grouping_of_result", None) is not None and hasattr(be, "remember_grouping"):
            # for the index a tokeniser later builds from these very centroids (needs their host copy: sync)
            if sync:
                be.remember_grouping(self.centroids, self._grouping_of_result)
            self._grouping_of_result = None
        if not sync:
            return None
        st = self.iteration_stats
        return float(st[-1]["obj"]) if st else 0.0


_GROUPER = None


def _grouper():
    """One helper thread for the host-side spatial grouping (a C call that releases the GIL)."""
    global _GROUPER
    if _GROUPER is None:
        from concurrent.futures import ThreadPoolExecutor
        _GROUPER = ThreadPoolExecutor(max_workers=1, thread_name_prefix="at-grouping")
    return _GROUPER


class IndexFlatL2:
    """faiss.IndexFlatL2(d) with add / search(x, 1) / reset / ntotal (spec_tokenizer.py:123-127,77)."""

    def __init__(self, d, backend=None):
        self.d = int(d)
        self.backend = backend or default_backend()
        self._c = None
        self._prune = None   # (cperm, dmin, gnbr) for the coarse-to-fine exact search, built lazily
        self.prune = True
        self.rows_coherent = True   # queries arrive as consecutive frames of clips (spec_tokenizer.py:66-78)

    @property
    def ntotal(self) -> int:
        return 0 if self._c is None else int(self._c.shape[0])

    def reset(self) -> None:
        self._c = None
        self._prune = None

    def add(self, c) -> None:
        c = self.backend._f32(c)
        assert c.dim() == 2 and c.shape[1] == self.d, f"expected [n, {self.d}]"
        self._c = c.clone() if self._c is None else torch.cat([self._c, c], 0)
        self._prune = None

    def assign(self, x, want_dist=True):
        """Device tensors in, (ids [n] int64, dis [n] float32 or None) out: the search itself.  Large
        searches against large tables go through the exact coarse-to-fine pruned sweep."""
        be = self.backend
        c = self._c
        k = c.shape[0]
        if (self.prune and hasattr(be, "assign_c2f") and self.d in (64, 128) and k >= 1024
                and (k + 31) // 32 <= 512 and x.shape[0] >= 65536):
            if self._prune is None:
                c_host = be.to_host(c)
                cperm = be.recall_grouping(c_host) if hasattr(be, "recall_grouping") else None
                if cperm is None or cperm.numel() != ((k + 31) // 32) * 32:
                    cperm = be.from_host(be.group_rows_kd(c_host))
                # 4 neighbour groups for the one-launch guess generator (measured: 1-4 equal, 8 is 6 % slower)
                self._prune = (cperm, be.group_min_dist(c, cperm), be.group_neighbours(be.group_means(c, cperm), 4))
            cperm, dmin, gnbr = self._prune
            return be.assign_c2f(x, c, cperm, dmin, gnbr, want_dist=want_dist, coherent=self.rows_coherent)
        if (self.prune and hasattr(be, "assign_unguided") and self.d in (64, 128) and 128 <= k < 1024 and x.shape[0] >= 65536
                and getattr(be, "switches", {}).get("filter", True)):
            # too few centroids to prune (configs[1]: k = 500), enough rows to pay for the set-up: the same fp16-split
            # filter, every group visited
            if self._prune is None:
                self._prune = (be.from_host(be.group_rows_kd(be.to_host(c))), None, None)
            return be.assign_unguided(x, c, want_dist=want_dist, cperm=self._prune[0])
        return be.assign(x, c, want_dist=want_dist)

    def search(self, x, k=1):
        if k != 1:
            raise NotImplementedError("IndexFlatL2.search: only k=1 is implemented (the reference's use)")
        host = _is_host(x)
        be = self.backend
        x = be._f32(x)
        assert x.dim() == 2 and x.shape[1] == self.d, f"expected [n, {self.d}]"
        if self._c is None:
            D = torch.full((x.shape[0], 1), float("inf"), device=be.device)
            I = torch.full((x.shape[0], 1), -1, dtype=torch.int64, device=be.device)
        else:
            ids, dis = self.assign(x)
            D, I = dis.unsqueeze(1), ids.unsqueeze(1)
        return (be.to_host(D), be.to_host(I)) if host else (D, I)

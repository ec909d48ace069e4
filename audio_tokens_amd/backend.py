"""Tensor-level front end of the C ABI: PyTorch-ROCm tensors in, PyTorch-ROCm tensors out.

`HipBackend` is the only compute backend the package ships.  It hands `data_ptr()`s and the
current HIP stream to libaudio_tokens_amd.so and never computes anything itself; if the library
or a gfx950 device is missing it raises.  (tests/ drive the same host logic with a CPU stand-in
built on the oracle, to cover the multi-rank code path under gloo; that stand-in lives in tests/,
not here.)
"""
from __future__ import annotations

import ctypes
import os

import numpy as np
import torch

from . import _lib

_vp = ctypes.c_void_p


def _ptr(t) -> _vp:
    return _vp(t.data_ptr()) if t is not None else _vp(None)


def _np_ptr(a: np.ndarray) -> _vp:
    return _vp(a.ctypes.data)


class HostHelpers:
    """The sequential, RNG-driven host pieces of the FAISS recipe (native C++, no GPU needed)."""

    def __init__(self):
        self.lib = _lib.load()
        self._groupings = {}   # fingerprint of a centroid table -> device grouping (remember_grouping)

    def rand_perm(self, n: int, seed: int) -> np.ndarray:
        out = np.empty(n, np.int32)
        _lib.check(self.lib.at_rand_perm_mt19937(n, seed, _np_ptr(out)))
        return out

    def rand_perm_prefix(self, n: int, seed: int, m: int) -> np.ndarray:
        out = np.empty(m, np.int32)
        _lib.check(self.lib.at_rand_perm_prefix_mt19937(n, seed, m, _np_ptr(out)))
        return out

    def mel_filterbank(self, sample_rate: int, n_fft: int, n_mels: int) -> np.ndarray:
        fb = np.empty((n_fft // 2 + 1, n_mels), np.float32)
        _lib.check(self.lib.at_mel_filterbank_host(sample_rate, n_fft, n_mels, _np_ptr(fb)))
        return fb

    def num_frames(self, L: int, hop: int) -> int:
        return int(self.lib.at_num_frames(L, hop))

    def group_rows_kd(self, rows: np.ndarray, leaf: int = 32) -> np.ndarray:
        """Spatial grouping of a host table into groups of `leaf` rows (-1 padded permutation)."""
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        k, d = rows.shape
        out = np.empty(((k + leaf - 1) // leaf) * leaf, np.int32)
        _lib.check(self.lib.at_group_rows_kd_host(_np_ptr(rows), k, d, leaf, _np_ptr(out)))
        return out

    # The grouping never decides a result, so a weak fingerprint of the table is enough to hand the grouping
    # a Kmeans ended with to the IndexFlatL2 later built from the same centroids (a mismatch costs speed only).
    @staticmethod
    def _table_fingerprint(rows: np.ndarray):
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        return rows.shape, int(rows.view(np.uint32).sum(dtype=np.uint64))

    def remember_grouping(self, rows: np.ndarray, cperm) -> None:
        cache = self._groupings
        if len(cache) >= 4:
            cache.pop(next(iter(cache)))
        cache[self._table_fingerprint(rows)] = cperm

    def recall_grouping(self, rows: np.ndarray):
        return self._groupings.get(self._table_fingerprint(rows))

    def resample_taps(self, orig_freq: int, new_freq: int):
        """-> (taps float32 [new, 2*width + orig], orig, new, width): torchaudio's sinc_interp_hann kernel."""
        o, nw, w = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        _lib.check(self.lib.at_resample_taps_host(orig_freq, new_freq, ctypes.byref(o), ctypes.byref(nw),
                                                  ctypes.byref(w), None, 0))
        taps = np.empty((nw.value, 2 * w.value + o.value), np.float32)
        _lib.check(self.lib.at_resample_taps_host(orig_freq, new_freq, ctypes.byref(o), ctypes.byref(nw),
                                                  ctypes.byref(w), _np_ptr(taps), taps.size))
        return taps, o.value, nw.value, w.value

    def resample_length(self, L: int, orig_freq: int, new_freq: int) -> int:
        return int(self.lib.at_resample_length(L, orig_freq, new_freq))

    @staticmethod
    def part_layout(k, d):
        """Packed per-rank partial of one Lloyd iteration, in floats: sums [k*d], counts [k], padding to an even
        offset, then the rank's objective as one double -> (offset of the double, total length)."""
        p = k * d + k
        off = p + (p & 1)
        return off, off + 2

    def split_clusters(self, hassign: np.ndarray, centroids: np.ndarray, n: int) -> int:
        """In place on two C-contiguous float32 host arrays; returns nsplit."""
        assert hassign.dtype == np.float32 and centroids.dtype == np.float32
        assert hassign.flags.c_contiguous and centroids.flags.c_contiguous
        k, d = centroids.shape
        ns = ctypes.c_int(0)
        _lib.check(self.lib.at_split_clusters_host(d, k, n, _np_ptr(hassign), _np_ptr(centroids),
                                                   ctypes.byref(ns)))
        return int(ns.value)


class HipBackend(HostHelpers):
    def __init__(self, device=None):
        super().__init__()
        if not torch.cuda.is_available():
            raise RuntimeError("audio_tokens_amd: no ROCm device visible (torch.cuda.is_available() "
                               "is False) and there is no CPU fallback")
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError(f"audio_tokens_amd: device must be a ROCm GPU, got {device}")
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = device
        self.ctx = _lib.context(device.index)
        self.assign_trace = None  # set to a list to collect (kind, n, d, k, start_event, end_event)
        self.assign_trace_only = None   # a set of kinds: only those are traced (every traced launch is two events on the stream)
        # host-side A/B switches, read from the environment once (the native ones: at_debug.h, self.debug_set)
        self.switches = {"filter": os.environ.get("AT_FILTER", "1") != "0",
                         "c2f_fused": os.environ.get("AT_C2F_FUSED", "1") != "0"}

    # -- development switches (include/at_debug.h) ------------------------------------------
    def debug_set(self, name: str, value: int) -> None:
        _lib.check(self.lib.at_debug_set(self.ctx.handle, name.encode(), int(value)))

    def debug_get(self, name: str) -> int:
        v = ctypes.c_int(0)
        _lib.check(self.lib.at_debug_get(self.ctx.handle, name.encode(), ctypes.byref(v)))
        return int(v.value)

    def diag_errors(self, reset=False) -> dict:
        """What the library met and did not treat as its own failure (include/at_debug.h: at_diag_errors): HIP errors
        found pending in the calling thread in front of one of its launches, and failures it tolerates by design."""
        stale, tol = ctypes.c_int64(0), ctypes.c_int64(0)
        sc, tc = ctypes.c_int(0), ctypes.c_int(0)
        where = ctypes.create_string_buffer(512)
        _lib.check(self.lib.at_diag_errors(ctypes.byref(stale), ctypes.byref(sc), ctypes.byref(tol), ctypes.byref(tc),
                                           where, 512, 1 if reset else 0))
        return {"stale_seen": int(stale.value), "stale_last_code": int(sc.value), "tolerated": int(tol.value),
                "tolerated_last_code": int(tc.value), "where": where.value.decode()}

    # -- plumbing --------------------------------------------------------------------------
    def _stream(self) -> _vp:
        return _vp(torch.cuda.current_stream(self.device).cuda_stream)

    def _f32(self, t, name="tensor") -> torch.Tensor:
        if isinstance(t, np.ndarray):
            t = torch.from_numpy(np.ascontiguousarray(t, dtype=np.float32))
        if t.dtype != torch.float32:
            t = t.float()
        if t.device != self.device:
            # (asynchronous only from pinned memory: a pageable source -- often a temporary made two lines up -- must have
            # been read by the time this returns, and torch does not wait for a non_blocking copy of one)
            t = t.to(self.device, non_blocking=t.is_pinned() if t.device.type == "cpu" else True)
        if not t.is_contiguous():
            t = t.contiguous()
        return t

    def _trace_list(self, kind):
        if self.assign_trace is None or (self.assign_trace_only is not None and kind not in self.assign_trace_only):
            return None
        return self.assign_trace

    def empty(self, shape, dtype=torch.float32) -> torch.Tensor:
        return torch.empty(shape, dtype=dtype, device=self.device)

    def zeros(self, shape, dtype=torch.float32) -> torch.Tensor:
        return torch.zeros(shape, dtype=dtype, device=self.device)

    def from_host(self, a, dtype=None) -> torch.Tensor:
        t = torch.from_numpy(np.ascontiguousarray(a))
        if dtype is not None:
            t = t.to(dtype)
        return t.to(self.device, non_blocking=False)

    def from_host_async(self, a, dtype=None) -> torch.Tensor:
        """from_host through a pinned staging buffer: the copy is queued on the current stream and the caller goes
        on (a pageable copy would first wait for everything queued before it).  For small tables."""
        t = torch.from_numpy(np.ascontiguousarray(a))
        if dtype is not None:
            t = t.to(dtype)
        key = ("h2d", tuple(t.shape), t.dtype)
        pool = self.__dict__.setdefault("_pinned_pool", {})
        slot = pool.get(key)
        if slot is None:
            slot = pool[key] = [[torch.empty(t.shape, dtype=t.dtype).pin_memory() for _ in range(4)], 0, [None] * 4]
        i = slot[1] % 4
        if slot[2][i] is not None:
            slot[2][i].synchronize()          # (the copy that last used this staging buffer, four uploads ago)
        h = slot[0][i]
        slot[1] += 1
        h.copy_(t)
        out = h.to(self.device, non_blocking=True)
        slot[2][i] = self.record_event()
        return out

    def to_host(self, t: torch.Tensor) -> np.ndarray:
        return t.detach().cpu().numpy()

    def host_staging(self, shape, dtype) -> torch.Tensor:
        """Pinned host buffer for small asynchronous read-backs."""
        return torch.empty(shape, dtype=dtype).pin_memory()

    def synchronize(self) -> None:
        torch.cuda.current_stream(self.device).synchronize()

    def background_stream(self):
        """The context's lowest-priority stream as a torch stream (at_background_stream): `with torch.cuda.stream(s)`
        routes this backend's launches there."""
        if getattr(self, "_bg_stream", None) is None:
            h = ctypes.c_void_p()
            _lib.check(self.lib.at_background_stream(self.ctx.handle, ctypes.byref(h)))
            self._bg_stream = torch.cuda.ExternalStream(h.value, device=self.device)
        return self._bg_stream

    def record_event_timed(self):
        """record_event with timing: `a.elapsed_time(b)` between two of them once both have completed."""
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream(self.device))
        return ev

    def record_event(self):
        """Event on the current stream; `.synchronize()` on it waits for the work queued so far only."""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        return ev

    # -- operators ---------------------------------------------------------------------------
    def logmel(self, wave, sample_rate=22050, n_fft=512, hop=128, n_mels=64, fb=None,
               frame_major=False, l2norm=False, out=None) -> torch.Tensor:
        """wave [n_clips, L] (or [L]) -> [n_clips, n_mels, T], or [n_clips*T, n_mels] if frame_major."""
        wave = self._f32(wave)
        if wave.dim() == 1:
            wave = wave.unsqueeze(0)
        assert wave.dim() == 2
        n_clips, L = wave.shape
        T = self.num_frames(L, hop)
        fbt = self._f32(fb) if fb is not None else None
        if fbt is not None:
            assert tuple(fbt.shape) == (n_fft // 2 + 1, n_mels), "fb must be [n_fft/2+1, n_mels]"
        shape = (n_clips * T, n_mels) if frame_major else (n_clips, n_mels, T)
        if out is None:
            out = self.empty(shape)
        else:
            assert out.is_contiguous() and out.dtype == torch.float32 and out.numel() == n_clips * T * n_mels
        rec = self._trace_list("logmel")
        if rec is not None:  # bench.py: HIP events on the launch stream around the launch
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(self.device))
        with torch.cuda.device(self.device):
            _lib.check(self.lib.at_logmel_f32(
                self.ctx.handle, _ptr(wave), n_clips, L, wave.stride(0), sample_rate, n_fft, hop, n_mels,
                _ptr(fbt), _ptr(out), _lib.AT_LAYOUT_FRAME_MAJOR if frame_major else _lib.AT_LAYOUT_MEL_MAJOR,
                1 if l2norm else 0, self._stream()))
        if rec is not None:
            e1.record(torch.cuda.current_stream(self.device))
            rec.append(("logmel", n_clips * T, n_mels, hop, e0, e1))
        return out

    def logmel_minmax(self, wave, sample_rate=22050, n_fft=512, hop=128, n_mels=64, fb=None) -> torch.Tensor:
        """wave [n_clips, L] -> [n_clips, n_mels, T]: logmel() followed by minmax_scale_clips(), one call; the extremes are
        collected by the log-mel kernel (at_logmel_minmax_f32), so the scaling is a single pass."""
        wave = self._f32(wave)
        if wave.dim() == 1:
            wave = wave.unsqueeze(0)
        n_clips, L = wave.shape
        T = self.num_frames(L, hop)
        fbt = self._f32(fb) if fb is not None else None
        out = self.empty((n_clips, n_mels, T))
        with torch.cuda.device(self.device):
            for c0 in range(0, n_clips, 65535):
                c1 = min(n_clips, c0 + 65535)
                _lib.check(self.lib.at_logmel_minmax_f32(
                    self.ctx.handle, _ptr(wave[c0:c1]), c1 - c0, L, wave.stride(0), sample_rate, n_fft, hop, n_mels,
                    _ptr(fbt), _ptr(out[c0:c1]), _lib.AT_LAYOUT_MEL_MAJOR, self._stream()))
        return out

    def resample(self, wave, orig_freq: int, new_freq: int) -> torch.Tensor:
        """wave [n_clips, L] (or [L]) -> [n_clips, ceil(L*new/orig)] (torchaudio Resample defaults)."""
        wave = self._f32(wave)
        squeeze = wave.dim() == 1
        if squeeze:
            wave = wave.unsqueeze(0)
        n_clips, L = wave.shape
        out_len = self.resample_length(L, orig_freq, new_freq)
        out = self.empty((n_clips, out_len))
        with torch.cuda.device(self.device):
            for c0 in range(0, n_clips, 65535):
                c1 = min(n_clips, c0 + 65535)
                _lib.check(self.lib.at_resample_f32(self.ctx.handle, _ptr(wave[c0:c1]), c1 - c0, L, wave.stride(0),
                                                    orig_freq, new_freq, _ptr(out[c0:c1]), out.stride(0), self._stream()))
        return out[0] if squeeze else out

    def token_histogram(self, ids, k: int) -> torch.Tensor:
        """int64 [k] counts of the token ids (device tensor in, device tensor out)."""
        if isinstance(ids, np.ndarray):
            ids = torch.from_numpy(np.ascontiguousarray(ids, dtype=np.int64))
        ids = ids.to(self.device, dtype=torch.int64).contiguous().reshape(-1)
        counts = self.empty((k,), torch.int64)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.at_token_histogram_i64(self.ctx.handle, _ptr(ids), ids.numel(), k, _ptr(counts), self._stream()))
        return counts

    def token_stats(self, counts):
        """counts: int64 [k] device histogram -> (sorted_counts int64 [k] descending, sorted_tokens int32 [k],
        stats float64 [8]: total, unique, ranks below 80 % cumulative share, slope, intercept, r, points fitted), all on
        the device (at_token_stats_f64)."""
        assert counts.dtype == torch.int64 and counts.is_contiguous() and counts.device == self.device
        k = counts.numel()
        sc, stok = self.empty((k,), torch.int64), self.empty((k,), torch.int32)
        stats = self.empty((8,), torch.float64)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.at_token_stats_f64(self.ctx.handle, _ptr(counts), k, _ptr(sc), _ptr(stok), _ptr(stats), self._stream()))
        return sc, stok, stats

    def minmax_scale_clips(self, specs) -> torch.Tensor:
        """specs [n_clips, ...] float32 on the device: every clip becomes (x - min) / (max - min), in place."""
        assert specs.dtype == torch.float32 and specs.is_contiguous() and specs.device == self.device
        if specs.shape[0]:
            with torch.cuda.device(self.device):
                _lib.check(self.lib.at_minmax_scale_clips_f32(self.ctx.handle, _ptr(specs), specs.shape[0],
                                                              specs[0].numel(), self._stream()))
        return specs

    def conv1d_mel(self, frames, weight, bias=None, padding=1) -> torch.Tensor:
        """frames [n, n_mels] -> [n, n_mels * num_kernels]: nn.Conv1d(1, num_kernels, kernel_size, padding) along the mel
        axis, feature = mel * num_kernels + kernel (at_conv1d_mel_f32).  weight: the module's [num_kernels, 1, kernel_size]."""
        frames = self._f32(frames)
        n, n_mels = frames.shape
        w = self._f32(weight).reshape(weight.shape[0], -1).contiguous()
        b = self._f32(bias).contiguous() if bias is not None else None
        nk, ks = w.shape
        out = self.empty((n, n_mels * nk))
        with torch.cuda.device(self.device):
            _lib.check(self.lib.at_conv1d_mel_f32(self.ctx.handle, _ptr(frames), n, n_mels, _ptr(w), _ptr(b), nk, ks, int(padding),
                                                  _ptr(out), self._stream()))
        return out

    def l2norm_rows(self, x, out=None) -> torch.Tensor:
        x = self._f32(x)
        assert x.dim() == 2
        n, d = x.shape
        if out is None:
            out = torch.empty_like(x)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.at_l2norm_rows_f32(self.ctx.handle, _ptr(x), n, d, _ptr(out), self._stream()))
        return out

    def assign(self, x, c, want_dist=True):
        x, c = self._f32(x), self._f32(c)
        assert x.dim() == 2 and c.dim() == 2 and x.shape[1] == c.shape[1]
        n, d = x.shape
        k = c.shape[0]
        ids = self.empty((n,), torch.int64)
        dist = self.empty((n,), torch.float32) if want_dist else None
        rec = self._trace_list("plain")
        if rec is not None:  # bench.py: HIP events on the launch stream around every assign launch
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(self.device))
        with torch.cuda.device(self.device):
            _lib.check(self.lib.at_assign_f32(self.ctx.handle, _ptr(x), n, d, _ptr(c), k, _ptr(ids),
                                              _ptr(dist), self._stream()))
        if rec is not None:
            e1.record(torch.cuda.current_stream(self.device))
            rec.append(("plain", n, d, k, e0, e1))
        return ids, dist

    def assign_hinted(self, x, c, hint_ids, order=None, want_dist=True):
        """Same result as assign(), faster when the hints (the previous assignment) are mostly right.
        hint_ids: int64 [n] per row.  order: what centroid_accum(..., want_order=True) returned for
        those ids -- a pair (rows in member-list order, their ids in that order), both uint32 bit
        patterns in int32 tensors -- or None."""
        x, c = self._f32(x), self._f32(c)
        n, d = x.shape
        k = c.shape[0]
        assert hint_ids.dtype == torch.int64 and hint_ids.is_contiguous() and hint_ids.numel() == n
        hint_sorted = None
        if order is not None:
            order, hint_sorted = order
            for t in (order, hint_sorted):
                assert t.dtype == torch.int32 and t.is_contiguous() and t.numel() == n
        ids = self.empty((n,), torch.int64)
        dist = self.empty((n,), torch.float32) if want_dist else None
        rec = self._trace_list("hinted")
        if rec is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(self.device))
        with torch.cuda.device(self.device):
            _lib.check(self.lib.at_assign_hinted_f32(self.ctx.handle, _ptr(x), n, d, _ptr(c), k, _ptr(hint_ids),
                                                     _ptr(order), _ptr(hint_sorted), _ptr(ids), _ptr(dist),
                                                     self._stream()))
        if rec is not None:
            e1.record(torch.cuda.current_stream(self.device))
            rec.append(("hinted", n, d, k, e0, e1))
        return ids, dist

    # -- exact pruning for Lloyd iterations ---------------------------------------------------
    def group_min_dist(self, c, cperm) -> torch.Tensor:
        """cperm: int32 device tensor [ng*32] (-1 padded) -> dmin float32 [k, ng]."""
        c = self._f32(c)
        k, d = c.shape
        assert cperm.dtype == torch.int32 and cperm.is_contiguous() and cperm.numel() % 32 == 0
        ng = cperm.numel() // 32
        dmin = self.empty((k, ng))
        with torch.cuda.device(self.device):
            _lib.check(self.lib.at_group_min_dist_f32(self.ctx.handle, _ptr(c), k, d, _ptr(cperm), ng, _ptr(dmin),
                                                      self._stream()))
        return dmin

    def visit_order(self, ids, dis, k):
        """Rows sorted by (previous id, previous distance) -> (order, ids in that order), int32
        tensors holding uint32 bit patterns."""
        n = ids.numel()
        assert ids.dtype == torch.int64 and ids.is_contiguous()
        order, hs = self.empty((n,), torch.int32), self.empty((n,), torch.int32)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.at_visit_order_f32(self.ctx.handle, _ptr(ids), _ptr(dis), n, k, _ptr(order), _ptr(hs),
                                                   self._stream()))
        return order, hs

    def visit_order_beside(self, ids, dis, k, sum_out=None):
        """visit_order on a stream of its own (it has its own sort buffers): the caller's stream goes on at
        once and must call the returned function before it uses the result.  sum_out (float64 [1] view): the
        objective sum_i dis[i] is queued on that stream too -- nothing on the caller's stream needs it before the join."""
        main = torch.cuda.current_stream(self.device)
        side = getattr(self, "_order_stream", None)
        if side is None:
            side = self._order_stream = torch.cuda.Stream(self.device)
        side.wait_stream(main)                 # ids / dis are complete on the caller's stream
        with torch.cuda.stream(side):
            if sum_out is not None:
                self.sum_f64(dis, out=sum_out)
                sum_out.record_stream(side)
            out = self.visit_order(ids, dis, k)

        def join(out=out, ids=ids, dis=dis):   # (ids / dis are kept alive until the sort has been waited for)
            torch.cuda.current_stream(self.device).wait_stream(side)
            for t in out:
                t.record_stream(torch.cuda.current_stream(self.device))
            return out
        return join

    def sum_beside(self, dis, sum_out):
        """The objective sum_i dis[i] on the side stream of visit_order_beside; the returned function makes the caller's
        stream wait for it (iterations that sort nothing beside -- short shards, the last iteration)."""
        main = torch.cuda.current_stream(self.device)
        side = getattr(self, "_order_stream", None)
        if side is None:
            side = self._order_stream = torch.cuda.Stream(self.device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            self.sum_f64(dis, out=sum_out)
            sum_out.record_stream(side)

        def join(dis=dis):
            torch.cuda.current_stream(self.device).wait_stream(side)
        return join

    def prune_stats(self, reset=False):
        """(accumulators computed, accumulators of the dense sweep) over this context's exact pruned
        sweeps so far.  Synchronises."""
        a, t = ctypes.c_int64(0), ctypes.c_int64(0)
        _lib.check(self.lib.at_prune_stats(self.ctx.handle, ctypes.byref(a), ctypes.byref(t), 1 if reset else 0))
        return int(a.value), int(t.value)

    def group_means(self, c, cperm) -> torch.Tensor:
        c = self._f32(c)
        k, d = c.shape
        ng = cperm.numel() // 32
        means = self.empty((ng, d))
        with torch.cuda.device(self.device):
            _lib.check(self.lib.at_group_means_f32(self.ctx.handle, _ptr(c), k, d, _ptr(cperm), ng, _ptr(means),
                                                   self._stream()))
        return means

    def group_neighbours(self, means, nnb=4) -> torch.Tensor:
        """[ng, ceil(ng/32)] bit table: for every group the nnb groups with the nearest means
        (itself included).  Heuristic input of the coarse pass."""
        means = self._f32(means)
        ng, d = means.shape
        gnbr = torch.empty((ng, (ng + 31) // 32), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.at_group_neighbours_f32(self.ctx.handle, _ptr(means), ng, d, int(nnb), _ptr(gnbr),
                                                        self._stream()))
        return gnbr

    def _nearest_mean(self, x, means):
        """Group of the (approximately) nearest group mean -- a guess, so at d = 64 it is taken from the
        fp16 filter sweep over all groups of means instead of the dense fp32 sweep."""
        n, d = x.shape
        ngm = means.shape[0]
        if d not in (64, 128) or n < 20 or not self.switches["filter"]:
            return self.assign(x, means, want_dist=False)[0]
        key = (n, ngm)
        if getattr(self, "_ident", (None,))[0] != key:
            g8 = (ngm + 31) // 32
            perm = torch.full((g8 * 32,), -1, dtype=torch.int32, device=self.device)
            perm[:ngm] = torch.arange(ngm, dtype=torch.int32, device=self.device)
            every = torch.full((g8, (g8 + 31) // 32), -1, dtype=torch.int32, device=self.device)   # all bits set
            self._ident = (key, torch.arange(n, dtype=torch.int32, device=self.device),
                           torch.zeros(n, dtype=torch.int32, device=self.device), perm, every)
        _, order, zeros, perm, every = self._ident
        return self.assign_pruned(x, means, (order, zeros), perm, every, want_dist=False, mode=1, filter=True)[0]

    def assign_unguided(self, x, c, want_dist=True, cperm=None):
        """Exact nearest centroid by the fp16-split filter sweep with no guesses: every row starts without a running
        best, so every group is visited -- no pruning, but three fp16 MFMA products (and the fp32 redo of the rows the
        error bound cannot settle) instead of the fp32 MFMA sweep.  For tables too small to prune (k < 1024): 1.47 against
        2.53 ms per 4.3 M rows at k = 500 (tools/small_k_probe.py).  Same result as assign().  cperm: a spatial grouping
        of the centroids (group_rows_kd); without one they are grouped as they come -- every group is visited either
        way, but alike centroids in a group let the hi*hi screening drop more tiles (1.47 against 1.75 ms)."""
        x, c = self._f32(x), self._f32(c)
        n, k = x.shape[0], c.shape[0]
        cache = getattr(self, "_unguided", None)
        if cache is None or cache[0].numel() < n:
            cache = self._unguided = (torch.arange(n, dtype=torch.int32, device=self.device),
                                      torch.full((n,), -1, dtype=torch.int32, device=self.device), {})
        if cperm is None:
            cperm = cache[2].get(k)
        if cperm is None:
            ng = (k + 31) // 32
            cperm = torch.full((ng * 32,), -1, dtype=torch.int32, device=self.device)
            cperm[:k] = torch.arange(k, dtype=torch.int32, device=self.device)
            cache[2][k] = cperm
        dmin = self.group_min_dist(c, cperm)     # (also leaves the fp16 image of c in the context: image_current)
        return self.assign_pruned(x, c, (cache[0][:n], cache[1][:n]), cperm, dmin, want_dist=want_dist, filter=True,
                                  image_current=True)

    def assign_coarse(self, x, c, cperm, means, gnbr, want_dist=True):
        """Guesses for rows in their own coherent order (frames of clips): nearest group mean -> its
        neighbour groups -> best centroid among them, one launch (at_assign_coarse_f32)."""
        x, c, means = self._f32(x), self._f32(c), self._f32(means)
        n, d = x.shape
        ng = cperm.numel() // 32
        ids = self.empty((n,), torch.int64)
        dist = self.empty((n,), torch.float32) if want_dist else None
        with torch.cuda.device(self.device):
            _lib.check(self.lib.at_assign_coarse_f32(self.ctx.handle, _ptr(x), n, d, _ptr(c), c.shape[0], _ptr(cperm), ng,
                                                     _ptr(means), _ptr(gnbr), _ptr(ids), _ptr(dist), self._stream()))
        return ids, dist

    def assign_c2f(self, x, c, cperm, dmin, gnbr=None, want_dist=True, coherent=False):
        """Exact nearest centroid without guesses: nearest group mean -> best member of that group
        and its neighbour groups (a guess) -> pruned exact sweep.  Same result as assign().
        coherent: the rows are in an order in which neighbours are alike (frames of clips): the guess
        is then made in one launch without sorting the rows by mean first."""
        x, c = self._f32(x), self._f32(c)
        ng = cperm.numel() // 32
        means = self.group_means(c, cperm)
        if gnbr is None:
            gnbr = self.group_neighbours(means)
        if coherent and x.shape[1] in (64, 128) and self.switches["filter"] and self.switches["c2f_fused"]:
            guess, gdis = self.assign_coarse(x, c, cperm, means, gnbr)
            return self.assign_pruned(x, c, self.visit_order(guess, gdis, c.shape[0]), cperm, dmin, want_dist=want_dist)
        gx = self._nearest_mean(x, means)
        guess, gdis = self.assign_pruned(x, c, self.visit_order(gx, None, ng), cperm, gnbr, mode=1)
        return self.assign_pruned(x, c, self.visit_order(guess, gdis, c.shape[0]), cperm, dmin, want_dist=want_dist)

    def assign_pruned(self, x, c, order, cperm, dmin, want_dist=True, mode=0, filter=None, image_current=False):
        """mode 0: same result as assign(); `order` = visit_order(...) of the guesses.
        mode 1: best centroid among the groups named by each 32-row tile (order = visit_order of
        group ids) -- a guess generator.
        filter (default: on unless AT_FILTER=0): fp16-split first stage, same bits out.
        image_current: `dmin` was computed by group_min_dist(c, cperm) on these very tensors and nothing
        has used the filter since, so the fp16 image of the centroids need not be rebuilt."""
        x, c = self._f32(x), self._f32(c)
        n, d = x.shape
        if filter is None:
            filter = self.switches["filter"]
        use_filter = bool(filter) and d in (64, 128)
        k = c.shape[0]
        order, hint_sorted = order
        ng = cperm.numel() // 32
        ids = self.empty((n,), torch.int64)
        dist = self.empty((n,), torch.float32) if want_dist else None
        rec = self._trace_list("pruned" if mode == 0 else "coarse")
        # exact filtered calls do their pre-pass (guess distances + group masks) inside the sweep kernel
        fused = use_filter and mode == 0 and self.debug_get("filter_fused") != 0
        with torch.cuda.device(self.device):
            if not fused:  # pre-pass first, so that the events below bracket the sweep kernel only
                _lib.check(self.lib.at_prune_mask_f32(self.ctx.handle, _ptr(x), n, d, _ptr(c), k, _ptr(order),
                                                      _ptr(hint_sorted), ng, _ptr(dmin), mode, self._stream()))
            if rec is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(torch.cuda.current_stream(self.device))
            args = _lib.PrunedArgs(x=x.data_ptr(), n=n, d=d, c=c.data_ptr(), k=k, order=order.data_ptr(),
                                   hint_sorted=hint_sorted.data_ptr(), cperm=cperm.data_ptr(), ng=ng,
                                   bounds=dmin.data_ptr() if dmin is not None else None, guess_only=1 if mode else 0,
                                   use_filter=1 if use_filter else 0, prepass_done=0 if fused else 1,
                                   image_current=1 if image_current else 0, ids=ids.data_ptr(),
                                   dist_or_null=dist.data_ptr() if dist is not None else None)
            _lib.check(self.lib.at_assign_pruned_f32(self.ctx.handle, ctypes.byref(args), self._stream()))
        if rec is not None:
            e1.record(torch.cuda.current_stream(self.device))
            rec.append(("pruned" if mode == 0 else "coarse", n, d, k, e0, e1))
        return ids, dist

    def filter_stats(self, reset=True, timing=False):
        """(rows swept by the fp16-split filter, rows it handed to the fp32 sweep) since the last reset;
        with timing also (summed HIP-event ms of the stage-1 kernel, number of exact sweeps timed,
        32x32 tiles multiplied hi*hi, tiles refined with the lo products)."""
        a, b = ctypes.c_int64(0), ctypes.c_int64(0)
        ms, cnt = ctypes.c_double(0.0), ctypes.c_int64(0)
        tiles, refined = ctypes.c_int64(0), ctypes.c_int64(0)
        _lib.check(self.lib.at_filter_stats(self.ctx.handle, ctypes.byref(a), ctypes.byref(b), ctypes.byref(ms),
                                            ctypes.byref(cnt), ctypes.byref(tiles), ctypes.byref(refined),
                                            1 if reset else 0))
        return (a.value, b.value, ms.value, cnt.value, tiles.value, refined.value) if timing else (a.value, b.value)

    def filter_probe(self, x, c, order, cperm, dmin):
        """Test hook: stage 1 only -> (winner ids, approx [n, 2] = (P of the winner, gap to runner-up), listed)."""
        x, c = self._f32(x), self._f32(c)
        n, d = x.shape
        k = c.shape[0]
        order, hint_sorted = order
        ng = cperm.numel() // 32
        ids = self.empty((n,), torch.int64)
        approx = self.empty((n, 2), torch.float32)
        listed = ctypes.c_int64(0)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.at_filter_probe_f32(self.ctx.handle, _ptr(x), n, d, _ptr(c), k, _ptr(order),
                                                    _ptr(hint_sorted), _ptr(cperm), ng, _ptr(dmin), _ptr(ids),
                                                    _ptr(approx), ctypes.byref(listed), self._stream()))
        return ids, approx, listed.value

    def rand_perm_prefix_device(self, n: int, seed: int, m: int) -> torch.Tensor:
        """First m entries of faiss' rand_perm(n, seed) as a device int32 tensor, computed on the device
        (no host work, no upload); same bits as HostHelpers.rand_perm_prefix."""
        out = self.empty((m,), torch.int32)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.at_rand_perm_prefix_device(self.ctx.handle, n, seed, m, _ptr(out), self._stream()))
        return out

    def gather_rows(self, x, idx) -> torch.Tensor:
        x = self._f32(x)
        if isinstance(idx, np.ndarray):
            idx = self.from_host(idx.astype(np.int32, copy=False))
        assert idx.dtype == torch.int32 and idx.is_contiguous() and idx.device == self.device
        m, d = idx.numel(), x.shape[1]
        out = self.empty((m, d))
        with torch.cuda.device(self.device):
            _lib.check(self.lib.at_gather_rows_f32(self.ctx.handle, _ptr(x), d, _ptr(idx), m, _ptr(out),
                                                   self._stream()))
        return out

    def centroid_accum_join(self):
        """Makes the current stream wait for a long-list accumulation left pending by
        centroid_accum(..., defer_join=True).  Must precede any use of the packed partial."""
        with torch.cuda.device(self.device):
            _lib.check(self.lib.at_centroid_accum_join(self.ctx.handle, self._stream()))

    def centroid_accum(self, x, ids, k, out=None, want_order=False, defer_join=False):
        """Packed partial result [k*d + k]: sums [k, d] followed by counts [k].  With want_order also
        (rows sorted by (id, row), their ids in that order) as int32 tensors holding uint32 bit
        patterns -- the `order` argument of assign_hinted."""
        x = self._f32(x)
        n, d = x.shape
        assert ids.dtype == torch.int64 and ids.is_contiguous() and ids.numel() == n
        if out is None:
            out = self.empty((k * d + k,))
        order = self.empty((n,), torch.int32) if want_order else None
        sorted_ids = self.empty((n,), torch.int32) if want_order else None
        with torch.cuda.device(self.device):
            _lib.check(self.lib.at_centroid_accum_defer(self.ctx.handle, 1 if defer_join else 0))
            try:
                _lib.check(self.lib.at_centroid_accum_f32(
                    self.ctx.handle, _ptr(x), n, d, _ptr(ids), k, _ptr(out), _vp(out.data_ptr() + 4 * k * d),
                    _ptr(order), _ptr(sorted_ids), self._stream()))
            except Exception:
                # a failed call must not leave a half-deferred join behind: wait for whatever the side stream
                # was given and go back to the joined form
                self.lib.at_centroid_accum_join(self.ctx.handle, self._stream())
                raise
            finally:
                self.lib.at_centroid_accum_defer(self.ctx.handle, 0)
        return (out, (order, sorted_ids)) if want_order else out

    def centroid_finalize(self, parts, k, d):
        """parts [n_parts, k*d + k] packed partials (rank order) -> (centroids [k, d], hassign [k])."""
        if parts.dim() == 1:
            parts = parts.unsqueeze(0)
        assert parts.is_contiguous() and parts.shape[1] >= k * d + k   # (a packed partial may carry the objective behind)
        n_parts = parts.shape[0]
        cent = self.empty((k, d))
        hassign = self.empty((k,))
        stride = parts.stride(0)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.at_centroid_finalize_f32(
                self.ctx.handle, _ptr(parts), stride, _vp(parts.data_ptr() + 4 * k * d), stride, n_parts,
                k, d, _ptr(cent), _ptr(hassign), self._stream()))
        return cent, hassign

    def split_clusters_device(self, hassign, cent, n: int, nsplit_out) -> None:
        """faiss split_clusters in place on device tensors (hassign [k], cent [k, d]); nsplit_out: int32 [1]."""
        k, d = cent.shape
        assert hassign.dtype == torch.float32 and cent.dtype == torch.float32 and nsplit_out.dtype == torch.int32
        assert hassign.is_contiguous() and cent.is_contiguous() and hassign.numel() == k
        with torch.cuda.device(self.device):
            _lib.check(self.lib.at_split_clusters_f32(self.ctx.handle, d, k, n, _ptr(hassign), _ptr(cent), _ptr(nsplit_out),
                                                      self._stream()))

    def _objective_parts(self, parts, k, d, objs):
        if objs is not None:
            assert objs.dtype == torch.float64 and objs.is_contiguous()
            return _ptr(objs), 1, objs.numel()
        if parts.dim() == 1:
            parts = parts.unsqueeze(0)
        off, total = self.part_layout(k, d)
        assert parts.is_contiguous() and parts.shape[1] == total
        return _vp(parts.data_ptr() + 4 * off), total // 2, parts.shape[0]

    def lloyd_stats(self, hassign, parts, k: int, d: int, stats_row, objs=None) -> None:
        """stats_row (float64 [2]) <- (objective summed in rank order, imbalance).  The per-rank objectives are the
        doubles riding at the end of the packed partials, or -- objs given -- a float64 tensor [n_ranks]."""
        assert stats_row.dtype == torch.float64
        ptr, stride, n_parts = self._objective_parts(parts, k, d, objs)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.at_lloyd_stats_f64(self.ctx.handle, _ptr(hassign), k, ptr, stride, n_parts, _ptr(stats_row),
                                                   self._stream()))

    def lloyd_stats_split(self, hassign, cent, n: int, nsplit_out, parts, stats_row, objs=None) -> None:
        """lloyd_stats followed by split_clusters_device, one launch (at_lloyd_stats_split_f32)."""
        k, d = cent.shape
        assert stats_row.dtype == torch.float64 and nsplit_out.dtype == torch.int32
        assert hassign.dtype == torch.float32 and cent.dtype == torch.float32
        assert hassign.is_contiguous() and cent.is_contiguous() and hassign.numel() == k
        ptr, stride, n_parts = self._objective_parts(parts, k, d, objs)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.at_lloyd_stats_split_f32(self.ctx.handle, d, k, n, _ptr(hassign), _ptr(cent), _ptr(nsplit_out),
                                                         ptr, stride, n_parts, _ptr(stats_row), self._stream()))

    def sum_parts(self, parts) -> torch.Tensor:
        """parts [n_parts, m] float32 -> [m]: added in ascending part order (at_sum_parts_f32)."""
        assert parts.dim() == 2 and parts.is_contiguous() and parts.dtype == torch.float32
        out = self.empty((parts.shape[1],))
        with torch.cuda.device(self.device):
            _lib.check(self.lib.at_sum_parts_f32(self.ctx.handle, _ptr(parts), parts.stride(0), parts.shape[0], parts.shape[1],
                                                 _ptr(out), self._stream()))
        return out

    def to_host_async(self, t: torch.Tensor):
        """-> (pinned host tensor, event): the copy is queued on the current stream, the caller's thread goes on;
        whoever needs the values waits for the event."""
        # pinned buffers are kept (allocating one costs milliseconds and synchronises the device): four per
        # (shape, dtype), handed out in turn -- a reader is long done with a buffer four copies later
        key = (tuple(t.shape), t.dtype)
        pool = self.__dict__.setdefault("_pinned_pool", {})
        slot = pool.get(key)
        if slot is None:
            slot = pool[key] = [[torch.empty(t.shape, dtype=t.dtype).pin_memory() for _ in range(4)], 0]
        h = slot[0][slot[1] % 4]
        slot[1] += 1
        h.copy_(t, non_blocking=True)
        return h, self.record_event()

    def sum_f64(self, v, out=None) -> torch.Tensor:
        """Device double scalar (shape [1]); no synchronisation."""
        v = self._f32(v)
        if out is None:
            out = self.empty((1,), torch.float64)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.at_sum_f32(self.ctx.handle, _ptr(v), v.numel(), _ptr(out), self._stream()))
        return out

    def nonfinite_flag(self, v) -> torch.Tensor:
        """Device int32 [1]: 1 if any value of v is NaN/Inf.  No synchronisation."""
        v = self._f32(v)
        flag = self.empty((1,), torch.int32)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.at_any_nonfinite_f32(self.ctx.handle, _ptr(v), v.numel(), _ptr(flag),
                                                     self._stream()))
        return flag

    def logmel_nonfinite_take(self) -> torch.Tensor:
        """Device int32 [1]: 1 if a unit-row pass of logmel(..., l2norm=True) since the last call met a frame whose
        squared norm is not finite (a NaN / Inf in it, or squares that overflow: confirm with any_nonfinite).  Clears
        the context's flag.  No synchronisation, and no second read of the frames."""
        flag = self.empty((1,), torch.int32)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.at_logmel_nonfinite_take(self.ctx.handle, _ptr(flag), self._stream()))
        return flag

    def any_nonfinite(self, v) -> bool:
        return bool(self.nonfinite_flag(v).item())


_default: dict = {}


def default_backend(device=None) -> HipBackend:
    """Process-wide HipBackend per device (creates the at_ctx on first use)."""
    if device is None:
        if not torch.cuda.is_available():
            raise RuntimeError("audio_tokens_amd: no ROCm device visible and there is no CPU fallback")
        idx = torch.cuda.current_device()
    else:
        device = torch.device(device)
        idx = device.index if device.index is not None else torch.cuda.current_device()
    be = _default.get(idx)
    if be is None:
        be = _default[idx] = HipBackend(torch.device("cuda", idx))
    return be

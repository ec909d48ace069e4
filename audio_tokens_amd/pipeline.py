"""The reference's three hot-path stages chained on the device, without the .npy round trip.

run_pipeline.py of danavery/audio-tokens hands data from stage to stage through files
(spectrograms/*.npy -> centroids.npy -> tokenized_audio/*.npy).  DevicePipeline computes the same
quantities from waveforms that are already resident in HBM, which is what bench.py times:

    SpectrogramGenerator.generate_mel_spectrogram   (spectrogram_generator.py:123-126)
      + ClusterCreator._batch_generator's .T / concatenate and normalize_vectors
        (cluster_creator.py:52,64-66,83-102)          -> one fused launch: frame-major, unit rows
    ClusterCreator.run's per-batch kmeans.train loop (cluster_creator.py:42-59)
    SpecTokenizer.process_batch's search              (spec_tokenizer.py:66-78)

With torch.distributed initialised (`distributed=True`) every rank holds a contiguous block of the
clips of every file batch; log-mel and tokenise need no communication and k-means exchanges the
per-cluster partial sums/counts once per Lloyd iteration (ops.Kmeans).

`run_streaming` is the same computation for clip sets that do not fit in HBM (BASELINE.json's
configs[4]: 2 M clips = 1.76 TB of waveform per node): waveforms stay in host memory and cross to
the device in chunks through two pinned staging buffers on a copy stream, overlapped with the
kernels; frames exist only for the k-means batch being trained and are recomputed for tokenise
(log-mel costs less than keeping 3.5 G frames around), tokens return to the host chunk by chunk.
"""
from __future__ import annotations

import os
import time
from dataclasses import dataclass, field

import torch

from .backend import default_backend
from .ops import IndexFlatL2, Kmeans


@dataclass
class PipelineResult:
    centroids: torch.Tensor            # [k, d] unit-norm rows (what centroids.npy holds)
    tokens_train: torch.Tensor         # int64 [n_train_clips * T]
    tokens_val: torch.Tensor           # int64 [n_val_clips * T]
    frames_per_clip: int
    stage_seconds: dict = field(default_factory=dict)
    kmeans_stats: list = field(default_factory=list)


class DevicePipeline:
    def __init__(self, n_mels=64, vocab_size=500, niter=20, sample_rate=22050, n_fft=512, hop_length=128,
                 clustering_batch_size=10000, spectrogram_batch_size=5000, distributed=False,
                 process_group=None, backend=None, verbose=False, prune=True):
        self.n_mels, self.vocab_size, self.niter = n_mels, vocab_size, niter
        self.sample_rate, self.n_fft, self.hop_length = sample_rate, n_fft, hop_length
        self.clustering_batch_size = clustering_batch_size
        self.spectrogram_batch_size = spectrogram_batch_size
        self.distributed, self.process_group = distributed, process_group
        self.be = backend or default_backend()
        self.verbose = verbose
        self.prune = prune   # False: plain dense sweeps everywhere (bench.py's floor / verification run)
        self.overlap_logmel = os.environ.get("AT_OVERLAP_LOGMEL", "1") != "0"
        self.beside_clips = int(os.environ.get("AT_BESIDE_CLIPS", "50"))   # clips per side-stream log-mel launch
        self.world = 1
        if distributed:
            import torch.distributed as dist
            if dist.is_initialized():
                self.world = dist.get_world_size(process_group)

    def _frames(self, wave):
        """[n_clips, L] -> unit-norm frame-major rows [n_clips*T, n_mels]."""
        be = self.be
        n_clips, L = wave.shape
        T = be.num_frames(L, self.hop_length)
        out = be.empty((n_clips * T, self.n_mels))
        step = self.spectrogram_batch_size
        for c0 in range(0, n_clips, step):
            c1 = min(n_clips, c0 + step)
            be.logmel(wave[c0:c1], self.sample_rate, self.n_fft, self.hop_length, self.n_mels,
                      frame_major=True, l2norm=True, out=out[c0 * T:c1 * T])
        return out, T

    def _frames_beside(self, wave_tr, wave_va, per_rank):
        """Log-mel of everything but the first k-means batch on a side stream, in launches of a few hundred clips, so
        that it runs in the gaps of the Lloyd iterations of the batches before it (the tail of an iteration -- member
        lists, centroid sums, regrouping -- is a chain of small launches that leaves most of the chip idle, and the
        log-mel kernel is bound by latency, not by a unit the sweep saturates).  Returns the frame tensors, T, one event
        per k-means batch (its frames are complete), the event behind the validation frames, and faiss' input-check flag
        of the TRAINING frames (taken on the side stream between the two sets)."""
        be = self.be
        n_clips, L = wave_tr.shape
        T = be.num_frames(L, self.hop_length)
        main = torch.cuda.current_stream(be.device)
        if getattr(self, "_lm_stream", None) is None:
            # the lowest priority there is: the Lloyd iterations' own launches go first whenever both are ready
            # (torch's own stream pool offers "normal" and "high" only)
            self._lm_stream = be.background_stream() if os.environ.get("AT_BESIDE_PRIO", "1") != "0" else torch.cuda.Stream(device=be.device)
        side = self._lm_stream
        frames_tr = be.empty((n_clips * T, self.n_mels))
        n_va = 0 if wave_va is None else wave_va.shape[0]
        frames_va = be.empty((n_va * T, self.n_mels)) if n_va else None
        take_flag = hasattr(be, "logmel_nonfinite_take")
        if take_flag:
            be.logmel_nonfinite_take()                 # (whatever earlier log-mel passes of this context left behind)

        def logmel(wave, frames, c0, c1):
            be.logmel(wave[c0:c1], self.sample_rate, self.n_fft, self.hop_length, self.n_mels, frame_major=True, l2norm=True,
                      out=frames[c0 * T:c1 * T])

        first = min(n_clips, per_rank)
        for c0 in range(0, first, self.spectrogram_batch_size):
            logmel(wave_tr, frames_tr, c0, min(first, c0 + self.spectrogram_batch_size))
        ready = [torch.cuda.Event()]
        ready[0].record(main)
        piece = self.beside_clips
        bad = None
        with torch.cuda.stream(side):
            side.wait_event(ready[0])                  # (also orders the side stream behind the allocation of the frames)
            for b0 in range(first, n_clips, per_rank):
                b1 = min(n_clips, b0 + per_rank)
                for c0 in range(b0, b1, piece):
                    logmel(wave_tr, frames_tr, c0, min(b1, c0 + piece))
                ev = torch.cuda.Event()
                ev.record(side)
                ready.append(ev)
            if take_flag:
                bad = be.logmel_nonfinite_take()
            for c0 in range(0, n_va, piece):
                logmel(wave_va, frames_va, c0, min(n_va, c0 + piece))
            done = torch.cuda.Event()
            done.record(side)
        return frames_tr, frames_va, T, ready, done, bad

    def overlaps_logmel(self, n_train_clips) -> bool:
        """Whether run() computes the later batches' frames beside the training: only where a training is long enough
        to hide them -- the pruned path of large vocabularies (a Lloyd iteration of ~1.4 ms with a tail of small
        launches); at vocab_size 500 a whole training is 6 ms and the 50-clip launches cost more than they hide
        (configs[1]: 59 -> 64 ms per step when tried)."""
        per_rank = max(1, self.clustering_batch_size // self.world)
        return (self.overlap_logmel and self.be.device.type == "cuda" and n_train_clips > per_rank and self.prune
                and self.vocab_size >= 1024 and self.n_mels in (64, 128))

    def run(self, wave_train, wave_val=None, timing=False) -> PipelineResult:
        be = self.be
        sync = be.synchronize if timing else (lambda: None)
        secs = {}
        per_rank = max(1, self.clustering_batch_size // self.world)

        t0 = time.perf_counter()
        take_flag = hasattr(be, "logmel_nonfinite_take")
        # timing=True keeps the stages apart (stage_seconds are stand-alone costs); otherwise the log-mel of later
        # k-means batches and of the validation clips runs beside the training of the batches before them
        beside = not timing and self.overlaps_logmel(wave_train.shape[0])
        ready = done = None
        if beside:
            frames_tr, frames_va, T, ready, done, bad = self._frames_beside(
                be._f32(wave_train), be._f32(wave_val) if wave_val is not None and wave_val.shape[0] > 0 else None, per_rank)
        else:
            if take_flag:
                be.logmel_nonfinite_take()                 # (whatever earlier log-mel passes of this context left behind)
            frames_tr, T = self._frames(be._f32(wave_train))
            # faiss' input check (Clustering::train) on the training frames, from the unit-row pass that wrote them
            bad = be.logmel_nonfinite_take() if take_flag else None
            frames_va = None
            if wave_val is not None and wave_val.shape[0] > 0:
                frames_va, _ = self._frames(be._f32(wave_val))
        sync(); secs["logmel"] = time.perf_counter() - t0

        # k-means: one train() per batch of `clustering_batch_size` files (this rank's share of
        # each batch is clustering_batch_size / world clips), warm-started from the previous batch
        t0 = time.perf_counter()
        km = Kmeans(self.n_mels, self.vocab_size, niter=self.niter, verbose=self.verbose,
                    distributed=self.distributed, process_group=self.process_group, backend=be)
        km.prune = self.prune
        n_clips = wave_train.shape[0]
        pending = []
        if bad is None:
            if done is not None:
                torch.cuda.current_stream(be.device).wait_event(done)
            bad = be.nonfinite_flag(frames_tr)     # faiss' input check (Clustering::train), once for all batches
        for b, c0 in enumerate(range(0, n_clips, per_rank)):
            c1 = min(n_clips, c0 + per_rank)
            x = frames_tr[c0 * T:c1 * T]
            if ready is not None:
                torch.cuda.current_stream(be.device).wait_event(ready[b])
            # (no host round trip inside: the frames were scanned for NaN/Inf once, above; the statistics of
            # every batch are read back after the last one)
            km.train(x, init_centroids=None if b == 0 else km.centroids_device, sync=False, check_finite=False)
            pending.append(km._stats_pending)
        centroids = be.l2norm_rows(km.centroids_device)
        km.lend_grouping(centroids)
        sync(); secs["kmeans"] = time.perf_counter() - t0

        t0 = time.perf_counter()
        index = IndexFlatL2(self.n_mels, backend=be)          # SpecTokenizer.load_centroid_index
        index.prune = self.prune
        index.add(centroids)
        if done is not None:
            torch.cuda.current_stream(be.device).wait_event(done)
        tok_tr, _ = index.assign(frames_tr, want_dist=False)  # index.search(x, 1), ids only
        tok_va = be.empty((0,), torch.int64)
        if frames_va is not None:
            tok_va, _ = index.assign(frames_va, want_dist=False)
        sync(); secs["tokenize"] = time.perf_counter() - t0

        # the only host round trips of the pass, behind everything that was queued
        def over_ranks(flag):   # (every rank must take the same decision, or the others would wait in the next collective)
            if self.world > 1:
                import torch.distributed as dist
                flag = flag.cpu() if dist.get_backend(self.process_group) == "gloo" else flag
                dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.process_group)
            return bool(flag.item())

        if over_ranks(bad):
            # the unit-row pass flags a frame whose squared norm is not finite; squares that overflow look the same, so
            # the verdict is confirmed by the scan it replaced (an error path: never on the timed one)
            if not take_flag or over_ranks(be.nonfinite_flag(frames_tr)):
                raise RuntimeError("Error: 'std::isfinite(x_in[i])' failed: input contains NaN's or Inf's")
        stats = [km._read_stats(*p) for p in pending if p is not None]
        return PipelineResult(centroids, tok_tr, tok_va, T, secs, stats)

    # -- host-resident inputs ------------------------------------------------------------------
    class _Feeder:
        """Chunks of a host [n_clips, L] float32 tensor on the device, double buffered: while the
        kernels of chunk i run on the compute stream, chunk i+1 is copied host -> pinned staging ->
        device on a copy stream.  A staging/device buffer pair is reused only after the kernels
        that read it have been waited for (events)."""

        def __init__(self, be, wave_host, chunk_clips):
            self.be, self.wave, self.chunk = be, wave_host, int(chunk_clips)
            L = wave_host.shape[1]
            self.pinned = wave_host.is_pinned()
            self.copy_stream = torch.cuda.Stream(device=be.device)
            self.dev = [torch.empty((self.chunk, L), dtype=torch.float32, device=be.device) for _ in range(2)]
            self.stage = None if self.pinned else [torch.empty((self.chunk, L), dtype=torch.float32).pin_memory()
                                                   for _ in range(2)]
            self.copied = [torch.cuda.Event() for _ in range(2)]   # H2D of the slot finished
            self.consumed = [None, None]                             # kernels reading the slot finished

        def _issue(self, slot, c0, c1):
            n = c1 - c0
            if self.consumed[slot] is not None:
                self.copy_stream.wait_event(self.consumed[slot])
            with torch.cuda.stream(self.copy_stream):
                src = self.wave[c0:c1]
                if not self.pinned:
                    if self.consumed[slot] is not None:
                        self.copied[slot].synchronize()   # the previous H2D out of this staging buffer is done
                    self.stage[slot][:n].copy_(src)
                    src = self.stage[slot][:n]
                self.dev[slot][:n].copy_(src, non_blocking=True)
                self.copied[slot].record(self.copy_stream)

        def chunks(self, c_begin, c_end):
            """Yields (first clip, device view [n, L]) for the clips [c_begin, c_end)."""
            bounds = [(c, min(c_end, c + self.chunk)) for c in range(c_begin, c_end, self.chunk)]
            if not bounds:
                return
            self._issue(0, *bounds[0])
            for i, (c0, c1) in enumerate(bounds):
                slot = i & 1
                if i + 1 < len(bounds):
                    self._issue(slot ^ 1, *bounds[i + 1])
                torch.cuda.current_stream(self.be.device).wait_event(self.copied[slot])
                yield c0, self.dev[slot][:c1 - c0]
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.be.device))
                self.consumed[slot] = ev

    def run_streaming(self, wave_train_host, wave_val_host=None, chunk_clips=None, timing=False) -> PipelineResult:
        """run() for host-resident waveforms (torch CPU tensors [n_clips, L], pinned or not).  Same
        centroids and tokens as run() on the same clips; tokens are returned as host tensors."""
        be = self.be
        sync = be.synchronize if timing else (lambda: None)
        secs = {}
        chunk = int(chunk_clips or self.spectrogram_batch_size)
        n_clips, L = wave_train_host.shape
        T = be.num_frames(L, self.hop_length)
        feed = self._Feeder(be, wave_train_host, chunk)

        t0 = time.perf_counter()
        km = Kmeans(self.n_mels, self.vocab_size, niter=self.niter, verbose=self.verbose,
                    distributed=self.distributed, process_group=self.process_group, backend=be)
        km.prune = self.prune
        per_rank = max(1, self.clustering_batch_size // self.world)
        frames = be.empty((min(per_rank, n_clips) * T, self.n_mels))     # one k-means batch at a time
        stats = []
        for b, b0 in enumerate(range(0, n_clips, per_rank)):
            b1 = min(n_clips, b0 + per_rank)
            for c0, w in feed.chunks(b0, b1):
                be.logmel(w, self.sample_rate, self.n_fft, self.hop_length, self.n_mels, frame_major=True,
                          l2norm=True, out=frames[(c0 - b0) * T:(c0 - b0 + w.shape[0]) * T])
            x = frames[:(b1 - b0) * T]
            km.train(x) if b == 0 else km.train(x, init_centroids=km.centroids_device)
            stats.append(km.iteration_stats)
        centroids = be.l2norm_rows(km.centroids_device)
        km.lend_grouping(centroids)
        sync(); secs["logmel+kmeans"] = time.perf_counter() - t0

        t0 = time.perf_counter()
        index = IndexFlatL2(self.n_mels, backend=be)
        index.prune = self.prune
        index.add(centroids)

        def tokenise(wave_host, feeder):
            n = wave_host.shape[0]
            out = torch.empty((n * T,), dtype=torch.int64).pin_memory()
            for c0, w in feeder.chunks(0, n):
                fr = be.logmel(w, self.sample_rate, self.n_fft, self.hop_length, self.n_mels, frame_major=True,
                               l2norm=True, out=frames[:w.shape[0] * T] if w.shape[0] * T <= frames.shape[0] else None)
                tok, _ = index.assign(fr, want_dist=False)
                out[c0 * T:(c0 + w.shape[0]) * T].copy_(tok, non_blocking=True)
            be.synchronize()
            return out

        tok_tr = tokenise(wave_train_host, feed)
        tok_va = torch.empty((0,), dtype=torch.int64)
        if wave_val_host is not None and wave_val_host.shape[0] > 0:
            tok_va = tokenise(wave_val_host, self._Feeder(be, wave_val_host, chunk))
        sync(); secs["tokenize"] = time.perf_counter() - t0
        return PipelineResult(centroids, tok_tr, tok_va, T, secs, stats)

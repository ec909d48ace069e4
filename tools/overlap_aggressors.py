"""Which launch of the guess-mode search disturbs the producer kernels running beside it on the background stream.

Round 3: with packed-fp32 instructions the 512-point log-mel kernel returned wrong frames (lanes 48..63 of single registers)
beside the fp16 filter in guess mode; compiled without them (AT_NO_PACKED_FP32, csrc/at_internal.h) every line prints 0.
Logs of both builds: profiles/r03_packed_fp32_beside_mfma.txt."""
import os, sys, warnings, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_tokens_amd import _lib
from audio_tokens_amd.backend import default_backend, _ptr
from audio_tokens_amd.ops import Kmeans
from audio_tokens_amd.synth import synth_clips
be = default_backend()
T, k = 1723, 8192
wt = synth_clips(6000, L=220500, seed=4242, first_clip=0, device="cuda")
main = torch.cuda.current_stream()
bg = be.background_stream()
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    fr = be.logmel(wt[:3000], frame_major=True, l2norm=True)
    km = Kmeans(64, k, niter=5, backend=be)
    km.train(fr)
    C = km.centroids_device.clone()
    xs = fr[:2097152].contiguous()
    n = xs.shape[0]
    cperm = be.from_host(be.group_rows_kd(be.to_host(C)))
    ng = cperm.numel() // 32
    means = be.group_means(C, cperm)
    gnbr = be.group_neighbours(means, 8)
    gnbr4 = be.group_neighbours(means, 4)
    dmin = be.group_min_dist(C, cperm)
    gx = be._nearest_mean(xs, means)
    order, hs = be.visit_order(gx, None, ng)
    ids0, dis0 = be.assign(xs, C)
    order0, hs0 = be.visit_order(ids0, dis0, k)
    ids = be.empty((n,), torch.int64); dist = be.empty((n,), torch.float32)

    def mask_only(mode, od, h, bounds):
        _lib.check(be.lib.at_prune_mask_f32(be.ctx.handle, _ptr(xs), n, 64, _ptr(C), k, _ptr(od), _ptr(h), ng, _ptr(bounds), mode, be._stream()))

    def sweep_only(mode, od, h, bounds, flt):
        args = _lib.PrunedArgs(x=xs.data_ptr(), n=n, d=64, c=C.data_ptr(), k=k, order=od.data_ptr(), hint_sorted=h.data_ptr(),
                               cperm=cperm.data_ptr(), ng=ng, bounds=bounds.data_ptr(), guess_only=mode, use_filter=flt,
                               prepass_done=1, image_current=0, ids=ids.data_ptr(), dist_or_null=dist.data_ptr())
        _lib.check(be.lib.at_assign_pruned_f32(be.ctx.handle, ctypes.byref(args), be._stream()))

    mask_only(1, order, hs, gnbr); torch.cuda.synchronize()
    aggressors = {
        "guess-mode mask kernel only (group_only_mask_kernel)": (lambda: mask_only(1, order, hs, gnbr), 200),
        "guess-mode fp32 sweep only (masks of an earlier call)": (lambda: sweep_only(1, order, hs, gnbr, 0), 40),
        "guess-mode fp16 sweep only": (lambda: sweep_only(1, order, hs, gnbr, 1), 40),
        "exact fp16 sweep": (lambda: be.assign_pruned(xs, C, (order0, hs0), cperm, dmin), 30),
        "one-launch guess generator (assign_coarse)": (lambda: be.assign_coarse(xs, C, cperm, means, gnbr4), 20),
    }
    rows = torch.randn(6000 * 600, 64, device="cuda")
    victims = {
        "logmel n_fft=512 hop=128 (prefetching kernel)": lambda c0: be.logmel(wt[c0:c0 + 50], frame_major=True, l2norm=True),
        "logmel n_fft=512 hop=200 (staging kernel)": lambda c0: be.logmel(wt[c0:c0 + 50], hop=200, frame_major=True),
        "logmel n_fft=1024 (general kernel)": lambda c0: be.logmel(wt[c0:c0 + 50], n_fft=1024, hop=256, frame_major=True),
        "l2norm_rows": lambda c0: be.l2norm_rows(rows[c0 * 600:(c0 + 50) * 600]),
    }
    for name, (fn, reps) in aggressors.items():
      for vname, victim in victims.items():
        quiet = [victim(c0).clone() for c0 in range(0, 6000, 50)]
        torch.cuda.synchronize()
        wrong = 0
        for rep in range(3):
            ev = torch.cuda.Event(); ev.record(main)
            with torch.cuda.stream(bg):
                bg.wait_event(ev)
                got = [victim(c0) for c0 in range(0, 6000, 50)]
                done = torch.cuda.Event(); done.record(bg)
            for _ in range(reps):
                fn()
            busy = not done.query()
            main.wait_event(done); torch.cuda.synchronize()
            wrong += sum(int((a.view(torch.int32) != b.view(torch.int32)).any(1).sum()) for a, b in zip(quiet, got))
            del got
        print(f"{vname} | {name}: {wrong} wrong frames in 3 runs (log-mel still running when the last aggressor was queued: {busy})", flush=True)

"""Tokenise: what a frame's guess costs and what it is worth.

Today every frame's guess comes from the one-launch coarse generator (nearest group mean -> best member of that group and
its neighbour groups).  Alternative studied here: settle every s-th frame of a clip that way (key frames), and give the
frames in between the settled token of the nearest key frame (or the better of the two surrounding ones) as their guess.
Consecutive frames share 75 % of their samples.  Prints, per stride, how often the guess is the answer and what the exact
sweep costs behind it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_tokens_amd.backend import default_backend
from audio_tokens_amd.ops import Kmeans
from audio_tokens_amd.synth import synth_clips

be = default_backend()
k = 8192
noisy = len(sys.argv) > 1 and sys.argv[1] == "noisy"
clips = 2500
wave = synth_clips(clips, device="cuda", noisy=noisy)
frames = be.logmel(wave, frame_major=True, l2norm=True); del wave
n, d = frames.shape
T = n // clips
km = Kmeans(d, k, niter=20, backend=be); km.train(frames)
C = be.l2norm_rows(km.centroids_device)
cperm = be.from_host(be.group_rows_kd(be.to_host(C)))
dmin = be.group_min_dist(C, cperm)
means = be.group_means(C, cperm)
gnbr = be.group_neighbours(means, 4)


def t(fn, it=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3, r


ms_all, (truth, tdis) = t(lambda: be.assign_c2f(frames, C, cperm, dmin, gnbr, coherent=True))
ms_coarse, (g0, gd0) = t(lambda: be.assign_coarse(frames, C, cperm, means, gnbr))
ms_order, od0 = t(lambda: be.visit_order(g0, gd0, k))
ms_exact, _ = t(lambda: be.assign_pruned(frames, C, od0, cperm, dmin, want_dist=False))
print(f"n={n} T={T} noisy={noisy}: c2f {ms_all:.2f} ms = coarse {ms_coarse:.2f} + order {ms_order:.2f} + exact {ms_exact:.2f}; "
      f"coarse guess == truth {(g0 == truth).float().mean().item():.4f}")
ms_o, odt = t(lambda: be.visit_order(truth, tdis, k))
ms_e, _ = t(lambda: be.assign_pruned(frames, C, odt, cperm, dmin, want_dist=False))
print(f"  perfect guess: order {ms_o:.2f} + exact {ms_e:.2f}")

tok3 = truth.view(clips, T)
tt = torch.arange(T, device=frames.device)
for s in (2, 3, 4, 8):
    nkey = (T + s - 1) // s
    key_tok = tok3[:, ::s]                                    # settled tokens of the key frames
    near = torch.clamp((tt + s // 2) // s, max=nkey - 1)
    g_near = key_tok[:, near].reshape(-1).contiguous()
    lo_i = tt // s
    hi_i = torch.clamp(lo_i + 1, max=nkey - 1)
    g_lo, g_hi = key_tok[:, lo_i].reshape(-1), key_tok[:, hi_i].reshape(-1)
    d_lo = ((frames - C[g_lo]) ** 2).sum(1)
    d_hi = ((frames - C[g_hi]) ** 2).sum(1)
    g_best = torch.where(d_hi < d_lo, g_hi, g_lo).contiguous()
    for name, g in (("nearest key", g_near), ("better of two", g_best)):
        ms_o, od = t(lambda: be.visit_order(g, None, k))
        ms_e, (ids, _) = t(lambda: be.assign_pruned(frames, C, od, cperm, dmin, want_dist=False))
        assert torch.equal(ids, truth)
        rows, listed = be.filter_stats()
        frac = (s - 1) / s
        print(f"  stride {s} {name}: guess == truth {(g == truth).float().mean().item():.4f}; on all rows order {ms_o:.2f} + exact {ms_e:.2f} ms; "
              f"estimate = c2f/{s} + {frac:.2f} x (order + exact) = {ms_all / s + frac * (ms_o + ms_e):.2f} ms (today {ms_all:.2f})")

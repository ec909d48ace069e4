"""Tokenise: does the exact sweep need its rows sorted by guess when they arrive as consecutive frames of clips?
(development aid)  Same result either way; prints the cost of the sweep on rows in their own order."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_tokens_amd.backend import default_backend
from audio_tokens_amd.ops import Kmeans
from audio_tokens_amd.synth import synth_clips

be = default_backend()
k = 8192
noisy = len(sys.argv) > 1 and sys.argv[1] == "noisy"
wave = synth_clips(2500, device="cuda", noisy=noisy)
frames = be.logmel(wave, frame_major=True, l2norm=True); del wave
n, d = frames.shape
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


g0, gd0 = be.assign_coarse(frames, C, cperm, means, gnbr)
ms_order, od = t(lambda: be.visit_order(g0, gd0, k))
ms_sorted, (truth, _) = t(lambda: be.assign_pruned(frames, C, od, cperm, dmin, want_dist=False))
ident = torch.arange(n, device="cuda", dtype=torch.int32).view(od[0].dtype)
hs = g0.to(torch.int32).view(od[1].dtype).contiguous()
ms_own, (ids, _) = t(lambda: be.assign_pruned(frames, C, (ident, hs), cperm, dmin, want_dist=False))
print(f"noisy={noisy} n={n}: visiting order {ms_order:.2f} ms + exact sweep {ms_sorted:.2f} ms; rows in their own order: exact sweep {ms_own:.2f} ms; "
      f"same tokens {torch.equal(ids, truth)}")

"""Fraction of (32-row tile, 32-centroid group) accumulators the pruned sweep must compute at a late
Lloyd iteration, recomputed with torch from the product's own dmin / visiting order (analysis aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from audio_tokens_amd.backend import default_backend
from audio_tokens_amd.ops import Kmeans
from audio_tokens_amd.synth import synth_clips
be = default_backend()
k = 8192
wave = synth_clips(2250, device="cuda")
frames = be.logmel(wave, frame_major=True, l2norm=True); del wave
for early in (2, 12):
    km = Kmeans(64, k, niter=early, backend=be); km.train(frames)
    C0 = km.centroids_device
    perm = be.rand_perm_prefix(frames.shape[0], 1234, k * 256)
    xs = be.gather_rows(frames, perm)
    ids0, dis0 = be.assign(xs, C0)                       # "previous" assignment
    part = be.centroid_accum(xs, ids0, k)
    C1, h = be.centroid_finalize(part, k, 64)            # updated centroids
    for grouping in ("kd on C1", "kd on iteration-1 centroids"):
        if grouping == "kd on C1":
            cperm = be.from_host(be.group_rows_kd(be.to_host(C1)))
        else:
            km1 = Kmeans(64, k, niter=1, backend=be); km1.train(frames)
            cperm = be.from_host(be.group_rows_kd(km1.centroids))
        dmin = be.group_min_dist(C1, cperm)
        order, hs = be.visit_order(ids0, dis0, k)
        o = order.long() & 0xffffffff
        p = hs.long() & 0xffffffff
        dnew = ((xs[o] - C1[p]) ** 2).sum(1)
        tau = 2 * (dnew + 1.6e-5).sqrt()
        n32 = (o.numel() // 32) * 32
        need = torch.zeros(n32 // 32, dmin.shape[1], dtype=torch.bool, device="cuda")
        step = 4096 * 32
        for s in range(0, n32, step):
            e = min(n32, s + step)
            need[s // 32:e // 32] = (dmin[p[s:e]] <= tau[s:e, None]).view(-1, 32, dmin.shape[1]).any(1)
        frac = need.float().mean().item()
        wg = need.view(-1, 8, need.shape[1])[: (need.shape[0] // 8)].any(1)        # 256-row workgroups
        tiles = wg.view(wg.shape[0], -1, 4).any(2).float().mean().item()
        print(f"after {early} iters, {grouping}: jobs needed {frac*100:.1f}%  128-centroid tiles staged per WG {tiles*100:.1f}%  changed {(be.assign(xs, C1)[0] != ids0).float().mean().item()*100:.2f}%")

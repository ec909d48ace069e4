"""How many rows could skip the search in Lloyd iteration i+1 by a Hamerly-style test (second-best distance
minus the largest centroid movement still above the distance to the own centroid)?  Development aid."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_tokens_amd.backend import default_backend
from audio_tokens_amd.ops import Kmeans
from audio_tokens_amd.synth import synth_clips
be = default_backend()
wave = synth_clips(1300, seed=3, device="cuda")
x = be.logmel(wave, frame_major=True, l2norm=True)[:2097152].contiguous(); del wave
k = 8192
km = Kmeans(64, k, niter=1, backend=be)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    km.train(x)
    c = km.centroids_device.clone()
    sample = x[::16].contiguous()                      # 131072 rows: top-2 distances by torch
    prev_gap = None
    for it in range(2, 21):
        km.train(x, init_centroids=c)                  # one more Lloyd iteration
        c_new = km.centroids_device.clone()
        mv = (c_new - c).norm(dim=1)
        d = torch.cdist(sample, c_new)
        top2, idx = d.topk(2, dim=1, largest=False)
        if prev_gap is not None:
            l_old, own_old = prev_gap
            keep = own_old == idx[:, 0]
            safe_max = (l_old - mv.max() >= top2[:, 0]) & keep
            # per-centroid variant: the runner-up's own movement for it, the maximum for everybody else (third best on)
            print(f"it {it:2d}: movement max {mv.max():.2e} p99 {mv.quantile(0.99):.2e} median {mv.median():.2e}; "
                  f"rows keeping their centroid {keep.float().mean():.3f}; safe by max-movement test {safe_max.float().mean():.3f}")
        prev_gap = (top2[:, 1].clone(), idx[:, 0].clone())
        c = c_new

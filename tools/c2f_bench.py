"""Timing of the unguided exact search (coarse-to-fine pruned) vs the plain sweep on real frames."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_tokens_amd.backend import default_backend
from audio_tokens_amd.ops import Kmeans
from audio_tokens_amd.synth import synth_clips
be = default_backend()
k = 8192
wave = synth_clips(2500, device="cuda")
frames = be.logmel(wave, frame_major=True, l2norm=True); del wave
km = Kmeans(64, k, niter=20, backend=be); km.train(frames)
C = be.l2norm_rows(km.centroids_device)
def t(fn, it=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3, r
ms_plain, (ids_p, dis_p) = t(lambda: be.assign(frames, C))
cperm = be.from_host(be.group_rows_kd(be.to_host(C)))
dmin = be.group_min_dist(C, cperm)
ms_c2f, (ids_c, dis_c) = t(lambda: be.assign_c2f(frames, C, cperm, dmin))
print(f"n={frames.shape[0]}: plain {ms_plain:.1f} ms, coarse-to-fine {ms_c2f:.1f} ms, equal ids {torch.equal(ids_p, ids_c)} dis {torch.equal(dis_p.view(torch.int32), dis_c.view(torch.int32))}")
means = be.group_means(C, cperm)
gx, _ = be.assign(frames, means, want_dist=False)
for nnb in (1, 2, 4, 8, 16):
    gn = be.group_neighbours(means, nnb)
    ms1, (guess, gd) = t(lambda: be.assign_pruned(frames, C, be.visit_order(gx, None, 256), cperm, gn, mode=1))
    ms2, _ = t(lambda: be.assign_pruned(frames, C, be.visit_order(guess, gd, k), cperm, dmin))
    print("nnb=%d: guess == truth %.3f  coarse %.2f ms + exact %.2f ms" % (nnb, (guess == ids_p).float().mean().item(), ms1, ms2))
for name, fn in [("group assign", lambda: be.assign(frames, be.group_means(C, cperm), want_dist=False)),
                 ("visit_order(groups)", lambda: be.visit_order(gx, None, 256)),
                 ("coarse pass", lambda: be.assign_pruned(frames, C, be.visit_order(gx, None, 256), cperm, None, mode=1)),
                 ("visit_order(guess)", lambda: be.visit_order(guess, gd, k)),
                 ("exact pruned pass", lambda: be.assign_pruned(frames, C, be.visit_order(guess, gd, k), cperm, dmin))]:
    print(f"  {name}: {t(fn)[0]:.2f} ms")

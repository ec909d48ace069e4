"""Where the per-train() fixed time goes at the per-rank size of an 8-GPU run.  Development aid."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from audio_tokens_amd.backend import default_backend
from audio_tokens_amd.ops import Kmeans
from audio_tokens_amd.synth import synth_clips
be = default_backend()
wave = synth_clips(200, device="cuda")
x = be.logmel(wave, frame_major=True, l2norm=True)[:262144].contiguous()
km = Kmeans(64, 8192, niter=20, backend=be)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    km.train(x)
    cent = km.centroids_device
    def T(label, fn, reps=5):
        torch.cuda.synchronize(); fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = fn()
        torch.cuda.synchronize()
        print(f"{label:34s} {(time.perf_counter() - t0) / reps * 1e3:8.3f} ms")
        return out
    for niter in (20, 1, 0):
        km.niter = niter
        T(f"train(niter={niter}) warm", lambda: km.train(x, init_centroids=cent))
    km.niter = 20
    T("any_nonfinite(x)", lambda: be.any_nonfinite(x))
    ch = T("to_host(cent)", lambda: be.to_host(cent))
    cp = T("group_rows_kd (host)", lambda: be.group_rows_kd(ch))
    cperm = T("from_host(cperm)", lambda: be.from_host(cp))
    dmin = T("group_min_dist", lambda: be.group_min_dist(cent, cperm))
    gm = T("group_means", lambda: be.group_means(cent, cperm))
    gn = T("group_neighbours", lambda: be.group_neighbours(gm, 8))
    T("assign_c2f", lambda: be.assign_c2f(x, cent, cperm, dmin, gn))
    ids, dis = be.assign_c2f(x, cent, cperm, dmin, gn)
    vo = T("visit_order", lambda: be.visit_order(ids, dis, 8192))
    T("assign_pruned", lambda: be.assign_pruned(x, cent, vo, cperm, dmin, image_current=True))
    T("_finish", lambda: km._finish(cent))
    T("host_staging x2", lambda: (be.host_staging((8192,), torch.float32), be.host_staging((1,), torch.float64)))

"""Tokenise at a small vocabulary (configs[1]: k = 500): the dense fp32 MFMA sweep against the exact fp16-split filter
sweep with every group needed and no guesses (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_tokens_amd.backend import default_backend
from audio_tokens_amd.ops import Kmeans
from audio_tokens_amd.synth import synth_clips

be = default_backend()
k = int(sys.argv[1]) if len(sys.argv) > 1 else 500
wave = synth_clips(2500, device="cuda")
frames = be.logmel(wave, frame_major=True, l2norm=True); del wave
n, d = frames.shape
km = Kmeans(d, k, niter=20, backend=be); km.train(frames)
C = be.l2norm_rows(km.centroids_device)


def t(fn, it=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3, r


ms_dense, (truth, tdis) = t(lambda: be.assign(frames, C))
ms_f, (ids, dis) = t(lambda: be.assign_unguided(frames, C))
rows, listed = be.filter_stats()
print(f"k={k} n={n}: dense fp32 sweep {ms_dense:.2f} ms; exact fp16-split sweep without guesses {ms_f:.2f} ms (listed {listed / max(rows, 1):.4f}); "
      f"same ids {torch.equal(ids, truth)} same distances {torch.equal(dis.view(torch.int32), tdis.view(torch.int32))}")

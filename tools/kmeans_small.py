"""Lloyd iteration cost at the per-rank sizes of an N-GPU run (the global 2 097 152-row subsample of a
10 000-file batch split N ways): how much of an iteration is fixed cost.  Development aid."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_tokens_amd.backend import default_backend
from audio_tokens_amd.ops import Kmeans
from audio_tokens_amd.synth import synth_clips
be = default_backend()
wave = synth_clips(1300, device="cuda")
frames = be.logmel(wave, frame_major=True, l2norm=True)
del wave
worlds = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
for world in worlds:
    n = 2097152 // world
    x = frames[:n].contiguous()
    km = Kmeans(64, 8192, niter=20, backend=be)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        km.train(x)                                   # warm-up / centroids for the warm start
        torch.cuda.synchronize(); t0 = time.perf_counter()
        km.train(x, init_centroids=km.centroids_device)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"rows/rank {n:8d} (N={world}): {dt / 20 * 1e3:.3f} ms per Lloyd iteration")

"""Host-side cost of a Lloyd iteration: a training set so small that the device is never the limit.  Development aid."""
import os, sys, time, warnings, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_tokens_amd.backend import default_backend
from audio_tokens_amd.ops import Kmeans
from audio_tokens_amd.synth import synth_clips
be = default_backend()
wave = synth_clips(40, device="cuda")
x = be.logmel(wave, frame_major=True, l2norm=True)[:32768].contiguous()
km = Kmeans(64, 8192, niter=20, backend=be)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    km.train(x)
    cent = km.centroids_device
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        km.train(x, init_centroids=cent)
        torch.cuda.synchronize(); print(f"{(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per iteration at 32768 rows")
    pr = cProfile.Profile(); pr.enable()
    for _ in range(5):
        km.train(x, init_centroids=cent)
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)

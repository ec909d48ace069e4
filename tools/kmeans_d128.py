"""Lloyd iteration cost at d = 128, k = 8192 (BASELINE.json configs[2] shape).  Development aid."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_tokens_amd.backend import default_backend
from audio_tokens_amd.ops import Kmeans
from audio_tokens_amd.synth import synth_clips
be = default_backend()
be.debug_set("filter_timing", 1)
wave = synth_clips(1300, device="cuda")
x = be.logmel(wave, 22050, 512, 128, 128, frame_major=True, l2norm=True)[:2097152].contiguous()
del wave
km = Kmeans(128, 8192, niter=20, backend=be)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    km.train(x)
    km.phase_seconds = {}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    km.train(x, init_centroids=km.centroids_device)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"d=128: {dt / 20 * 1e3:.3f} ms per Lloyd iteration (phases synced): ", {k: round(v * 1e3 / 20, 3) for k, v in km.phase_seconds.items()})
rows, listed, ms, sweeps, tiles, refined = be.filter_stats(timing=True)
print(f"filter kernel avg {ms / max(sweeps, 1):.3f} ms; accumulators {tiles / (rows / 32 * 256):.3f}; listed {listed / rows:.4f}")

"""Per-phase wall time of one Kmeans.train on synthetic frames (development aid)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from audio_tokens_amd.backend import default_backend
from audio_tokens_amd.ops import Kmeans
from audio_tokens_amd.synth import synth_clips

clips = int(sys.argv[1]) if len(sys.argv) > 1 else 2250
k = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
be = default_backend()
wave = synth_clips(clips, device="cuda")
frames = be.logmel(wave, frame_major=True, l2norm=True)
del wave
print("frames", tuple(frames.shape), flush=True)
for rep in range(3):
    km = Kmeans(64, k, niter=20, backend=be)
    km.phase_seconds = {}
    t0 = time.perf_counter(); km.train(frames); torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"train {t1-t0:.3f}s phases:", {a: round(b, 4) for a, b in km.phase_seconds.items()})
    print(" nsplit", [s["nsplit"] for s in km.iteration_stats])
    print(" imbalance", [round(s["imbalance_factor"], 2) for s in km.iteration_stats][:5], " obj", [round(s["obj"], 1) for s in km.iteration_stats][:3], km.iteration_stats[-1]["obj"])

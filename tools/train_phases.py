"""Wall time of one Kmeans.train at the benchmark's batch size (10 000 clips -> 17.2 M rows, subsampled to
2 097 152), cold and warm, against the sum of its Lloyd iterations.  Development aid."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_tokens_amd.backend import default_backend
from audio_tokens_amd.ops import Kmeans
from audio_tokens_amd.synth import synth_clips
be = default_backend()
frames = []
for s in range(4):
    wave = synth_clips(2500, seed=s, device="cuda")
    frames.append(be.logmel(wave, frame_major=True, l2norm=True)); del wave
x = torch.cat(frames); del frames
print("rows", x.shape[0])
km = Kmeans(64, 8192, niter=20, backend=be)
def timed(label, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    print(f"{label:28s} {(time.perf_counter() - t0) * 1e3:8.2f} ms")
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    timed("cold train (first call)", lambda: km.train(x))
    cent = km.centroids_device
    for niter in (20, 20, 1, 0):
        km.niter = niter
        timed(f"warm train niter={niter}", lambda: km.train(x, init_centroids=cent))
    km.niter = 20
    timed("cold train", lambda: km.train(x))
    km.phase_seconds = {}
    timed("warm train, phases synced", lambda: km.train(x, init_centroids=cent))
    print({k: round(v * 1e3, 2) for k, v in km.phase_seconds.items()})
    print("nsplit per iteration (cold):")
    km.phase_seconds = None
    km.train(x); print([s["nsplit"] for s in km.iteration_stats])

"""One warm-started 20-iteration training at 2 097 152 rows, optional switches from the command line:
tools/kmeans_one.py [order_beside=0] [debug_name=value ...].  Development aid (profiling target)."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_tokens_amd.backend import default_backend
from audio_tokens_amd.ops import Kmeans
from audio_tokens_amd.synth import synth_clips
be = default_backend()
clips = int(os.environ.get("CLIPS", "1300"))
wave = synth_clips(clips, device="cuda")
frames = be.logmel(wave, frame_major=True, l2norm=True)
del wave
if clips > 1300:      # like the bench's subsample: rows drawn from all over a big batch
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    x = frames[torch.randperm(frames.shape[0], device="cuda", generator=g)[:2097152]].contiguous()
else:
    x = frames[:2097152].contiguous()
del frames
km = Kmeans(64, 8192, niter=20, backend=be)
for a in sys.argv[1:]:
    name, val = a.split("=")
    if name == "order_beside":
        km.order_beside = bool(int(val))
    else:
        be.debug_set(name, int(val))
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    km.train(x)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        km.train(x, init_centroids=km.centroids_device, check_finite=False)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{' '.join(sys.argv[1:]) or 'defaults'}: {dt / 20 * 1e3:.3f} ms per Lloyd iteration")

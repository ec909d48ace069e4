"""Does a high-priority main stream shorten the Lloyd iteration (the side streams keep normal priority)?  Development aid."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_tokens_amd.backend import default_backend
from audio_tokens_amd.ops import Kmeans
from audio_tokens_amd.synth import synth_clips
be = default_backend()
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else None)
wave = synth_clips(1300, device="cuda")
frames = be.logmel(wave, frame_major=True, l2norm=True)
del wave
x = frames[:2097152].contiguous()
km = Kmeans(64, 8192, niter=20, backend=be)
hi = torch.cuda.Stream(priority=-1)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    km.train(x)
    for name, ctxm in (("default stream", None), ("high-priority stream", hi), ("default stream", None), ("high-priority stream", hi)):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if ctxm is None:
            km.train(x, init_centroids=km.centroids_device, check_finite=False)
        else:
            with torch.cuda.stream(ctxm):
                km.train(x, init_centroids=km.centroids_device, check_finite=False)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{name}: {dt / 20 * 1e3:.3f} ms per Lloyd iteration")

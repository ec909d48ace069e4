"""Is the Lloyd loop bound by the host's launch rate?  Times Kmeans.train(sync=False) returning (host done queueing)
against the device finishing.  Development aid."""
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
for world in [int(a) for a in sys.argv[1:]] or [1, 8]:
    n = 2097152 // world
    x = frames[:n].contiguous()
    km = Kmeans(64, 8192, niter=20, backend=be)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        km.train(x)
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            km.train(x, init_centroids=km.centroids_device, sync=False, check_finite=False)
            t1 = time.perf_counter()
            torch.cuda.synchronize(); t2 = time.perf_counter()
            print(f"rows {n}: host queued 20 iterations in {(t1 - t0) * 1e3:.2f} ms, device done at {(t2 - t0) * 1e3:.2f} ms")

# where the host spends its time while queueing: wall time inside each backend method (no device waits added)
import collections, functools
acc = collections.defaultdict(float); cnt = collections.Counter()
def wrap(name, f):
    @functools.wraps(f)
    def g(*a, **kw):
        t = time.perf_counter()
        try:
            return f(*a, **kw)
        finally:
            acc[name] += time.perf_counter() - t; cnt[name] += 1
    return g
for name in dir(be):
    if name.startswith("__"): continue
    f = getattr(be, name)
    if callable(f) and not isinstance(f, type):
        try: setattr(be, name, wrap(name, f))
        except Exception: pass
x = frames[:int(os.environ.get('ROWS', '2097152'))].contiguous()
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    km.train(x, init_centroids=km.centroids_device, sync=False, check_finite=False)
    t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"host total {(t1 - t0) * 1e3:.2f} ms")
for k_, v in sorted(acc.items(), key=lambda kv: -kv[1])[:25]:
    print(f"{v * 1e3:8.2f} ms {cnt[k_]:5d} x  {k_}")

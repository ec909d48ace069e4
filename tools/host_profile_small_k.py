"""cProfile of the host side of a 20-iteration training at configs[1]'s size (k = 500, 128 000-row subsample of 17 M rows).
Development aid."""
import cProfile, os, pstats, sys, warnings, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_tokens_amd.backend import default_backend
from audio_tokens_amd.ops import Kmeans
from audio_tokens_amd.synth import synth_clips
be = default_backend()
wave = synth_clips(3000, device="cuda")
x = be.logmel(wave, frame_major=True, l2norm=True)
del wave
km = Kmeans(64, 500, niter=20, backend=be)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    km.train(x, sync=False, check_finite=False)
    km.train(x, init_centroids=km.centroids_device, sync=False, check_finite=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    km.train(x, init_centroids=km.centroids_device, sync=False, check_finite=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"host returned after {(t1 - t0) * 1e3:.2f} ms, device done after {(t2 - t0) * 1e3:.2f} ms")
    pr = cProfile.Profile()
    pr.enable()
    km.train(x, init_centroids=km.centroids_device, sync=False, check_finite=False)
    pr.disable()
    torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)

"""cProfile of the host side of a warm 20-iteration training at an N=8 shard's size (262 144 rows).  Development aid."""
import cProfile, os, pstats, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_tokens_amd.backend import default_backend
from audio_tokens_amd.ops import Kmeans
from audio_tokens_amd.synth import synth_clips
be = default_backend()
wave = synth_clips(400, device="cuda")
x = be.logmel(wave, frame_major=True, l2norm=True)[:int(os.environ.get("ROWS", "262144"))].contiguous()
del wave
km = Kmeans(64, 8192, niter=20, backend=be)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    km.train(x)
    km.train(x, init_centroids=km.centroids_device, sync=False, check_finite=False)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    if os.environ.get("SYNC") == "1":
        km.train(x, init_centroids=km.centroids_device)
    else:
        km.train(x, init_centroids=km.centroids_device, sync=False, check_finite=False)
    pr.disable()
    torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)

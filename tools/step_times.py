"""Per-step wall times of the bench's configs[3] shard (barrier + synchronize around every step).  Development aid."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_tokens_amd.backend import default_backend
from audio_tokens_amd.pipeline import DevicePipeline
from audio_tokens_amd.synth import synth_clips
be = default_backend()
wt = synth_clips(22500, L=220500, seed=4242, first_clip=0, device="cuda")
wv = synth_clips(2500, L=220500, seed=4242, first_clip=22500, device="cuda")
pipe = DevicePipeline(n_mels=64, vocab_size=8192, niter=20, sample_rate=22050, n_fft=512, hop_length=128,
                      clustering_batch_size=10000, backend=be)
ts = []
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pipe.run(wt, wv)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(" ".join(f"{t:.1f}" for t in ts))

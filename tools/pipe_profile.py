"""cProfile of the host side of one DevicePipeline.run(timing=True) at the bench's configs[3] shard.  Development aid."""
import cProfile, os, pstats, sys, warnings
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
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    pipe.run(wt, wv); pipe.run(wt, wv)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    res = pipe.run(wt, wv, timing=True)
    pr.disable()
print(res.stage_seconds)
pstats.Stats(pr).sort_stats("tottime").print_stats(14)

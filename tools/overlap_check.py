"""Development aid: DevicePipeline.run with the log-mel beside the training against the sequential form, bench-sized."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_tokens_amd.backend import default_backend
from audio_tokens_amd.pipeline import DevicePipeline
from audio_tokens_amd.synth import synth_clips
be = default_backend()
wt = synth_clips(22500, L=220500, seed=4242, first_clip=0, device="cuda")
wv = synth_clips(2500, L=220500, seed=4242, first_clip=22500, device="cuda")
mk = lambda prune: DevicePipeline(n_mels=64, vocab_size=8192, niter=20, sample_rate=22050, n_fft=512, hop_length=128,
                                  clustering_batch_size=10000, backend=be, prune=prune)
pipe = mk(True)
T = 1723
def diff(a, b, tag):
    c = torch.equal(a.centroids.view(torch.int32), b.centroids.view(torch.int32))
    dt = (a.tokens_train != b.tokens_train).nonzero().flatten()
    dv = (a.tokens_val != b.tokens_val).nonzero().flatten()
    msg = f"{tag}: centroids equal {c}; train tokens differ {dt.numel()}, val tokens differ {dv.numel()}"
    if dt.numel():
        clips = torch.unique(dt // T)
        msg += f"; train clips {clips[:8].tolist()}..{clips[-3:].tolist()} ({clips.numel()} clips)"
    if dv.numel():
        clips = torch.unique(dv // T)
        msg += f"; val clips {clips[:8].tolist()}..{clips[-3:].tolist()} ({clips.numel()} clips)"
    print(msg, flush=True)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    pipe.overlap_logmel = False
    ref = pipe.run(wt, wv)
    diff(ref, pipe.run(wt, wv), "sequential again")
    pipe.overlap_logmel = True
    for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
        diff(ref, pipe.run(wt, wv), f"beside #{i}")
    d = mk(False)
    diff(ref, d.run(wt, wv), "dense beside")
    d.overlap_logmel = False
    diff(ref, d.run(wt, wv), "dense sequential")

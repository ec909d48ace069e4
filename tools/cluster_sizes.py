"""Largest clusters of a converged training at the Lloyd shape (how long are the 'long lists'?).  Development aid."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_tokens_amd.backend import default_backend
from audio_tokens_amd.ops import Kmeans
from audio_tokens_amd.synth import synth_clips
be = default_backend()
wave = synth_clips(1300, device="cuda")
frames = be.logmel(wave, frame_major=True, l2norm=True)
for world in (1, 8):
    x = frames[:2097152 // world].contiguous()
    km = Kmeans(64, 8192, niter=20, backend=be)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        km.train(x)
    cnt = torch.bincount(km._last_assign, minlength=8192)
    top = torch.sort(cnt, descending=True).values[:8].tolist()
    print(f"rows {x.shape[0]}: largest clusters {top}; > 2048: {int((cnt > 2048).sum())}; median {int(cnt.median())}")

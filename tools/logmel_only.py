"""Runs a few at_logmel_f32 launches (development aid for rocprofv3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_tokens_amd.backend import default_backend
be = default_backend()
nm = int(sys.argv[1]) if len(sys.argv) > 1 else 64
g = torch.Generator(device="cuda").manual_seed(0)
w = torch.rand(2000, 220500, device="cuda", generator=g) * 0.2 - 0.1
for _ in range(4):
    be.logmel(w, n_mels=nm, frame_major=True, l2norm=True)
torch.cuda.synchronize()

"""Times at_logmel_f32 (development aid; also usable under rocprofv3)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_tokens_amd.backend import default_backend
be = default_backend()
nm = int(sys.argv[1]) if len(sys.argv) > 1 else 64
g = torch.Generator(device="cuda").manual_seed(0)
w = torch.rand(2000, 220500, device="cuda", generator=g) * 0.2 - 0.1
for _ in range(2):
    be.logmel(w, n_mels=nm, frame_major=True, l2norm=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    out = be.logmel(w, n_mels=nm, frame_major=True, l2norm=True)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 5 * 1e3
frames = out.shape[0]
print(f"n_mels={nm}: {ms:.3f} ms for {frames} frames = {frames / ms / 1e6:.3f} G frames/s, "
      f"{frames * (128 * 4 + nm * 4) / ms / 1e9:.2f} TB/s algorithmic")

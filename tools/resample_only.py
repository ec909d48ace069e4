"""Time at_resample_f32 for the two common AudioSet source rates (10 s clips)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_tokens_amd.backend import default_backend

be = default_backend()
for sr in (44100, 48000):
    n = 512
    w = torch.randn(n, sr * 10, device=be.device)
    for _ in range(2):
        y = be.resample(w, sr, 22050)
    be.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        y = be.resample(w, sr, 22050)
    e1.record(); be.synchronize()
    ms = e0.elapsed_time(e1) / 5
    taps = be.resample_taps(sr, 22050)[0]
    gb = (w.numel() + y.numel()) * 4 / 1e9
    print(f"{sr}->22050: {n} clips {ms:.2f} ms  {n / ms * 1e3:.0f} clips/s  {gb / ms * 1e3:.0f} GB/s algorithmic  "
          f"{2 * y.numel() * taps.shape[1] / ms / 1e9:.1f} TFLOP/s  taps {taps.shape}")

"""Times at_logmel_f32 at other transform sizes than the default (development aid): logmel_nfft.py [n_fft hop]..."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_tokens_amd.backend import default_backend
be = default_backend()
g = torch.Generator(device="cuda").manual_seed(0)
w = torch.rand(2000, 220500, device="cuda", generator=g) * 0.2 - 0.1
args = [int(a) for a in sys.argv[1:]] or [1024, 512, 1024, 256, 2048, 512, 256, 64, 512, 128]
for n_fft, hop in zip(args[::2], args[1::2]):
    for fm in (True, False):
        for _ in range(2):
            out = be.logmel(w, n_fft=n_fft, hop=hop, n_mels=64, frame_major=fm)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            out = be.logmel(w, n_fft=n_fft, hop=hop, n_mels=64, frame_major=fm)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        frames = out.numel() // 64
        print(f"n_fft={n_fft} hop={hop} {'frame' if fm else 'mel'}-major: {ms:.3f} ms for {frames} frames = {frames / ms / 1e6:.3f} G frames/s, "
              f"{frames * (hop * 4 + 64 * 4) / ms / 1e9:.2f} TB/s algorithmic", flush=True)

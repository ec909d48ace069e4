"""Ad-hoc timing of single operators on the GPU box (development aid, not the contract bench)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from audio_tokens_amd.backend import default_backend

be = default_backend()
g = torch.Generator(device="cuda").manual_seed(0)

def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

for (n, d, k) in [(2097152, 64, 8192), (2097152, 128, 8192), (128000, 64, 500), (4194304, 64, 8192)]:
    x = torch.nn.functional.normalize(torch.randn(n, d, device="cuda", generator=g), dim=1)
    c = torch.nn.functional.normalize(torch.randn(k, d, device="cuda", generator=g), dim=1)
    ms = timeit(lambda: be.assign(x, c))
    print(f"assign n={n} d={d} k={k}: {ms:.3f} ms  {2*n*d*k/ms/1e9:.1f} TFLOP/s", flush=True)
    ids, _ = be.assign(x, c)
    ms2 = timeit(lambda: be.centroid_accum(x, ids, k))
    print(f"  centroid_accum: {ms2:.3f} ms  ({n*d*4/ms2/1e6:.0f} GB/s of rows)", flush=True)

w = torch.rand(2000, 220500, device="cuda", generator=g) * 0.2 - 0.1
for nm in (64, 128):
    ms = timeit(lambda: be.logmel(w, n_mels=nm, frame_major=True, l2norm=True), 3)
    fr = 2000 * 1723
    print(f"logmel n_mels={nm}: {ms:.3f} ms  {fr/ms/1e6:.3f} Gframes/s  {fr*(512+4*nm)/ms/1e9:.3f} TB/s algorithmic", flush=True)
x = torch.randn(4000000, 64, device="cuda", generator=g)
ms = timeit(lambda: be.l2norm_rows(x))
print(f"l2norm 4M x 64: {ms:.3f} ms {x.numel()*8/ms/1e9:.3f} TB/s")

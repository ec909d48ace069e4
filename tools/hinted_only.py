"""Runs a few at_assign_hinted_f32 launches of the Lloyd shape with realistic hints (the previous
assignment against slightly moved centroids).  Development aid for rocprofv3 --pmc."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_tokens_amd.backend import default_backend
be = default_backend()
n, d, k = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (2097152, 64, 8192)))
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.nn.functional.normalize(torch.randn(n, d, device="cuda", generator=g), dim=1)
c = torch.nn.functional.normalize(torch.randn(k, d, device="cuda", generator=g), dim=1)
ids, _ = be.assign(x, c)
part, order = be.centroid_accum(x, ids, k, want_order=True)
c2 = torch.nn.functional.normalize(c + 0.02 * torch.randn(k, d, device="cuda", generator=g), dim=1)
for _ in range(5):
    ids2, _ = be.assign_hinted(x, c2, ids, order)
torch.cuda.synchronize()
print("changed", float((ids2 != ids).float().mean()))

"""fp16-split filter vs the fp32 pruned sweep on the Lloyd shape: bit equality, ambiguity rate,
measured approximation error against float64, and timing.  Development aid."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from audio_tokens_amd.backend import default_backend
from audio_tokens_amd.synth import synth_clips

be = default_backend()
n_clips = int(sys.argv[1]) if len(sys.argv) > 1 else 1218
k = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
wave = synth_clips(n_clips, L=220500, seed=4242, device=be.device)
x = be.logmel(wave, 22050, 512, 128, 64, frame_major=True, l2norm=True)
n, d = x.shape
print("rows", n)
g = torch.Generator(device="cuda").manual_seed(1)
c = x[torch.randperm(n, device="cuda", generator=g)[:k]].clone()
for it in range(3):                                   # a few Lloyd steps for realistic centroids
    ids, dis = be.assign(x, c)
    part = be.centroid_accum(x, ids, k)
    c2, h = be.centroid_finalize(part, k, d)
    c = torch.where(h[:, None] > 0, c2, c)
ids0, dis0 = be.assign(x, c)
part = be.centroid_accum(x, ids0, k)
c2, h = be.centroid_finalize(part, k, d)
c_new = torch.where(h[:, None] > 0, c2, c).contiguous()
ref_ids, ref_dis = be.assign(x, c_new)

cperm = be.from_host(be.group_rows_kd(be.to_host(c_new)))
dmin = be.group_min_dist(c_new, cperm)
order = be.visit_order(ids0, dis0, k)

def timed(fn, reps=3):
    fn(); be.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    be.synchronize()
    return out, (time.perf_counter() - t0) / reps * 1e3

(a_ids, a_dis), t_fp32 = timed(lambda: be.assign_pruned(x, c_new, order, cperm, dmin, filter=False))
be.filter_stats()
(b_ids, b_dis), t_filt = timed(lambda: be.assign_pruned(x, c_new, order, cperm, dmin, filter=True))
rows, listed = be.filter_stats()
print(f"fp32 pruned {t_fp32:.2f} ms   filter+redo {t_filt:.2f} ms   listed {listed / max(rows, 1):.4f}")
print("fp32 pruned == dense:", bool(torch.equal(a_ids, ref_ids)), bool(torch.equal(a_dis, ref_dis)))
print("filter      == dense:", bool(torch.equal(b_ids, ref_ids)), bool(torch.equal(b_dis, ref_dis)))

w_ids, approx, cnt = be.filter_probe(x, c_new, order, cperm, dmin)
be.synchronize()
ok = w_ids >= 0
xd, cd = x.double(), c_new.double()
P_true = (cd[w_ids.clamp(min=0)] ** 2).sum(1) - 2.0 * (xd * cd[w_ids.clamp(min=0)]).sum(1)
err = (approx[:, 0].double() - P_true).abs()[ok]
print(f"stage 1: listed {cnt} of {n} ({cnt / n:.4f}); winner differs from exact on {int((w_ids != ref_ids).sum())} rows "
      f"(all must be listed); |P16 - P64| max {err.max().item():.3e} mean {err.mean().item():.3e}")
gap = approx[:, 1]
print("gap percentiles", np.percentile(gap[ok].cpu().numpy(), [1, 2, 5, 10, 50]))

"""Exact search at d = 128, k = 8192 (the shape of BASELINE.json configs[2]): dense fp32 sweep vs the pruned
fp32 sweep vs the fp16-split filter in front of it.  Development aid."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_tokens_amd.backend import default_backend
from audio_tokens_amd.synth import synth_clips
be = default_backend()
wave = synth_clips(1218, L=220500, seed=4242, device=be.device)
x = be.logmel(wave, 22050, 512, 128, 128, frame_major=True, l2norm=True)
n, d = x.shape; k = 8192
g = torch.Generator(device="cuda").manual_seed(1)
c = x[torch.randperm(n, device="cuda", generator=g)[:k]].clone()
for it in range(4):
    ids, dis = be.assign(x, c)
    c2, h = be.centroid_finalize(be.centroid_accum(x, ids, k), k, d)
    c = torch.where(h[:, None] > 0, c2, c).contiguous()
ref_ids, ref_dis = be.assign(x, c)
cperm = be.from_host(be.group_rows_kd(be.to_host(c)))
dmin = be.group_min_dist(c, cperm)
order = be.visit_order(ids, dis, k)
def timed(fn, reps=3):
    fn(); be.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): out = fn()
    be.synchronize(); return out, (time.perf_counter() - t0) / reps * 1e3
(_, _), t_dense = timed(lambda: be.assign(x, c))
(a_ids, a_dis), t_fp32 = timed(lambda: be.assign_pruned(x, c, order, cperm, dmin, filter=False))
be.filter_stats()
(b_ids, b_dis), t_filt = timed(lambda: be.assign_pruned(x, c, order, cperm, dmin, filter=True))
rows, listed = be.filter_stats()
print(f"d={d}: dense {t_dense:.2f} ms  fp32 pruned {t_fp32:.2f} ms  filter+redo {t_filt:.2f} ms  listed {listed / max(rows, 1):.4f}")
print("fp32 pruned == dense:", bool(torch.equal(a_ids, ref_ids)), bool(torch.equal(a_dis, ref_dis)))
print("filter      == dense:", bool(torch.equal(b_ids, ref_ids)), bool(torch.equal(b_dis, ref_dis)))

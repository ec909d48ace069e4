"""Stage-1 kernel time of one Lloyd-shaped exact call under debug values given on the command line
(tools/walk_ablate.py name=value[,value...]).  Rows are drawn from all over a big batch, like the bench's subsample.
Development aid."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_tokens_amd.backend import default_backend
from audio_tokens_amd.synth import synth_clips

be = default_backend()
be.debug_set("filter_timing", 1)
k = 8192
wave = synth_clips(int(os.environ.get("CLIPS", "6000")), L=220500, seed=4242, device=be.device)
fr = be.logmel(wave, 22050, 512, 128, 64, frame_major=True, l2norm=True)
del wave
g = torch.Generator(device="cuda").manual_seed(1)
x = fr[torch.randperm(fr.shape[0], device="cuda", generator=g)[:2097152]].contiguous()
del fr
n, d = x.shape
c = x[torch.randperm(n, device="cuda", generator=g)[:k]].clone()
for it in range(int(os.environ.get("ITERS", "6"))):
    ids, dis = be.assign(x, c)
    part = be.centroid_accum(x, ids, k)
    c2, h = be.centroid_finalize(part, k, d)
    c = torch.where(h[:, None] > 0, c2, c).contiguous()
cperm = be.from_host(be.group_rows_kd(be.to_host(c)))
dmin = be.group_min_dist(c, cperm)
order = be.visit_order(ids, dis, k)
name, vals = sys.argv[1].split("=")
for v in vals.split(","):
    be.debug_set(name, int(v))
    be.assign_pruned(x, c, order, cperm, dmin, filter=True)
    be.synchronize()
    be.filter_stats(reset=True)
    for _ in range(3):
        be.assign_pruned(x, c, order, cperm, dmin, filter=True)
    be.synchronize()
    rows, listed, ms, sweeps, tiles, refined = be.filter_stats(timing=True)
    print(f"{name}={v}: stage-1 kernel {ms / max(sweeps, 1) * 1e3:.0f} us  (listed {listed / max(rows, 1):.4f})", flush=True)

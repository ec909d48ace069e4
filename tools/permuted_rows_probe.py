"""What would the sweep and the accumulation cost if the rows lay in memory in visiting order?  (development aid)
One Lloyd-shaped exact call and one centroid accumulation on bench-like rows, as shipped (rows gathered through the
visiting order / the member lists) and on a physically permuted copy (identity order)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_tokens_amd.backend import default_backend
from audio_tokens_amd.synth import synth_clips

be = default_backend()
be.debug_set("filter_timing", 1)
k = 8192
wave = synth_clips(6000, L=220500, seed=4242, device=be.device)
fr = be.logmel(wave, 22050, 512, 128, 64, frame_major=True, l2norm=True)
del wave
g = torch.Generator(device="cuda").manual_seed(1)
x = fr[torch.randperm(fr.shape[0], device="cuda", generator=g)[:2097152]].contiguous()
del fr
n, d = x.shape
c = x[torch.randperm(n, device="cuda", generator=g)[:k]].clone()
for it in range(6):
    ids, dis = be.assign(x, c)
    part = be.centroid_accum(x, ids, k)
    c2, h = be.centroid_finalize(part, k, d)
    c = torch.where(h[:, None] > 0, c2, c).contiguous()
cperm = be.from_host(be.group_rows_kd(be.to_host(c)))
dmin = be.group_min_dist(c, cperm)
order, hs = be.visit_order(ids, dis, k)
o = order.view(torch.int32).long()
xp = x[o].contiguous()                                    # rows in visiting order
ident = torch.arange(n, device="cuda", dtype=torch.int32)


def sweep(xx, od):
    be.assign_pruned(xx, c, od, cperm, dmin, filter=True); be.synchronize(); be.filter_stats(reset=True)
    for _ in range(3):
        r = be.assign_pruned(xx, c, od, cperm, dmin, filter=True)
    be.synchronize()
    rows, listed, ms, sweeps, tiles, refined = be.filter_stats(timing=True)
    return ms / sweeps * 1e3, r


def timed(fn, it=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e6


us0, (i0, d0) = sweep(x, (order, hs))
us1, (i1, d1) = sweep(xp, (ident.view(order.dtype) if order.dtype != torch.int32 else ident, hs))
print(f"sweep, rows gathered through the visiting order: {us0:.0f} us; rows lying in visiting order: {us1:.0f} us; same winners: {torch.equal(i0[o], i1)}")
ids_p = ids[o].contiguous()
print(f"centroid_accum as shipped: {timed(lambda: be.centroid_accum(x, ids, k)):.0f} us; on the permuted copy: {timed(lambda: be.centroid_accum(xp, ids_p, k)):.0f} us; "
      f"the permutation itself (gather_rows): {timed(lambda: be.gather_rows(x, o.to(torch.int32))):.0f} us")

"""How much of the centroid table can the triangle inequality rule out per point / per wave?
(analysis aid for the design of an exact pruned sweep; uses torch ops, not the product path)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from audio_tokens_amd.backend import default_backend
from audio_tokens_amd.ops import Kmeans
from audio_tokens_amd.synth import synth_clips

be = default_backend()
clips = int(sys.argv[1]) if len(sys.argv) > 1 else 2250
k = 8192
wave = synth_clips(clips, device="cuda")
frames = be.logmel(wave, frame_major=True, l2norm=True); del wave
km = Kmeans(64, k, niter=int(sys.argv[2]) if len(sys.argv) > 2 else 10, backend=be)
km.train(frames)
C = km.centroids_device
perm = be.rand_perm_prefix(frames.shape[0], 1234, k * 256)
xs = be.gather_rows(frames, perm)
ids, dis = be.assign(xs, C)
part, (order, sid) = be.centroid_accum(xs, ids, k, want_order=True)
CC = torch.cdist(C.double(), C.double()).float()          # [k, k] true centroid-centroid distances
r = dis.clamp_min(0).sqrt()
# per point: fraction of centroids that survive ||c - c_p|| <= 2 r
sel = torch.randint(0, xs.shape[0], (20000,), device="cuda")
surv = (CC[ids[sel]] <= 2 * r[sel, None]).float().mean(1)
print("per-point surviving fraction: mean %.4f median %.4f p90 %.4f" % (surv.mean(), surv.median(), surv.quantile(0.9)))
# group the centroids into 64 tiles of 128 by a crude locality order (k-means on centroids, 64 groups)
g = 64
cent_g = C[torch.randperm(k, device="cuda")[:g]].clone()
for _ in range(10):
    a = torch.cdist(C, cent_g).argmin(1)
    for j in range(g):
        m = a == j
        if m.any(): cent_g[j] = C[m].mean(0)
a = torch.cdist(C, cent_g).argmin(1)
ordc = torch.argsort(a, stable=True)                      # locality order; tiles = consecutive 128
tile_of = torch.empty(k, dtype=torch.long, device="cuda"); tile_of[ordc] = torch.arange(k, device="cuda") // 128
# D[p][T] = min over c in tile T of ||c - c_p||
D = torch.full((k, k // 128), 1e9, device="cuda")
D.scatter_reduce_(1, tile_of[None, :].expand(k, k), CC, reduce="amin")
# per wave (64 consecutive rows in member-list order): tile needed if any row has D[p][T] <= 2 r
o = order.long() & 0xffffffff
n = (o.numel() // 64) * 64
pw = ids[o[:n]].view(-1, 64); rw = r[o[:n]].view(-1, 64)
wsel = torch.randint(0, pw.shape[0], (4000,), device="cuda")
need = (D[pw[wsel]] <= 2 * rw[wsel][:, :, None]).any(1).float()   # [waves, tiles]
print("per-wave tiles needed (locality-ordered tiles): mean %.3f of %d tiles; median %.1f" % (need.sum(1).mean(), k // 128, need.sum(1).median()))
Dr = torch.full((k, k // 128), 1e9, device="cuda")
Dr.scatter_reduce_(1, (torch.arange(k, device="cuda") // 128)[None, :].expand(k, k), CC, reduce="amin")
need_r = (Dr[pw[wsel]] <= 2 * rw[wsel][:, :, None]).any(1).float()
print("per-wave tiles needed (original order tiles): mean %.3f" % need_r.sum(1).mean())
print("distinct previous centroids per wave: mean %.2f" % torch.tensor([len(torch.unique(pw[i])) for i in wsel[:500]]).float().mean())
print("rms r %.3f  median CC %.3f  nearest-centroid dist median %.3f" % (r.pow(2).mean().sqrt(), CC.median(), CC.fill_diagonal_(9).min(1).values.median()))

# ---- job-level (32 rows x 32 centroids) survival with balanced kd-style grouping of the centroids
def kd_groups(C, leaf=32):
    idx = [torch.arange(C.shape[0], device=C.device)]
    while idx[0].numel() > leaf:
        nxt = []
        for ix in idx:
            sub = C[ix]
            dim = sub.var(0).argmax()
            o = torch.argsort(sub[:, dim])
            h = ix.numel() // 2
            nxt += [ix[o[:h]], ix[o[h:]]]
        idx = nxt
    return torch.cat(idx)          # order; consecutive `leaf` entries form a group

for name, ordc in (("kd-tree", kd_groups(C)), ("original", torch.arange(k, device="cuda"))):
    grp_of = torch.empty(k, dtype=torch.long, device="cuda"); grp_of[ordc] = torch.arange(k, device="cuda") // 32
    Dg = torch.full((k, k // 32), 1e9, device="cuda")
    Dg.scatter_reduce_(1, grp_of[None, :].expand(k, k), CC, reduce="amin")
    n32 = (o.numel() // 32) * 32
    p32 = ids[o[:n32]].view(-1, 32); r32 = r[o[:n32]].view(-1, 32)
    ts = torch.randint(0, p32.shape[0], (8000,), device="cuda")
    tau = 2 * r32[ts] + 1e-2
    need = (Dg[p32[ts]] <= tau[:, :, None]).any(1).float()          # [row tiles, groups]
    print(f"{name}: jobs needed per 32-row tile: mean {need.sum(1).mean():.1f} of {k//32} (={need.mean()*100:.1f}%), median {need.sum(1).median():.0f}, p90 {need.sum(1).quantile(0.9):.0f}")
    # if rows inside a cluster were additionally ordered by r (outliers together)
    key = ids.double() * 4.0 + r.double().clamp(max=3.9)
    o2 = torch.argsort(key)
    p32b = ids[o2[:n32]].view(-1, 32); r32b = r[o2[:n32]].view(-1, 32)
    needb = (Dg[p32b[ts]] <= (2 * r32b[ts] + 1e-2)[:, :, None]).any(1).float()
    print(f"{name}: same with rows ordered by (cluster, r): mean {needb.sum(1).mean():.1f} (={needb.mean()*100:.1f}%)")

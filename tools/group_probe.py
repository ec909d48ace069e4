"""Compares centroid groupings (axis kd-tree vs principal-axis bisection) by the fraction of
accumulators the pruned sweep would still need.  Analysis aid (torch ops)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from audio_tokens_amd.backend import default_backend
from audio_tokens_amd.ops import Kmeans
from audio_tokens_amd.synth import synth_clips
be = default_backend()
k = 8192
wave = synth_clips(4000, device="cuda")
frames = be.logmel(wave, frame_major=True, l2norm=True); del wave
km = Kmeans(64, k, niter=8, backend=be); km.train(frames)
C0 = km.centroids_device
perm = be.rand_perm_prefix(frames.shape[0], 1234, k * 256)
xs = be.gather_rows(frames, perm)
ids0, dis0 = be.assign(xs, C0)
C1, h = be.centroid_finalize(be.centroid_accum(xs, ids0, k), k, 64)

def pca_groups(C, leaf=32):
    idx = [torch.arange(C.shape[0], device=C.device)]
    while idx[0].numel() > leaf:
        nxt = []
        for ix in idx:
            sub = C[ix] - C[ix].mean(0)
            v = torch.randn(sub.shape[1], device=C.device)
            for _ in range(8):
                v = sub.T @ (sub @ v); v = v / v.norm()
            o = torch.argsort(sub @ v)
            hh = ix.numel() // 2
            nxt += [ix[o[:hh]], ix[o[hh:]]]
        idx = nxt
    return torch.cat(idx)

def kmeans_balanced_groups(C, leaf=32):
    # recursive 2-means bisection balanced at the median of the projection onto the two centres' axis
    idx = [torch.arange(C.shape[0], device=C.device)]
    g = torch.Generator(device=C.device).manual_seed(0)
    while idx[0].numel() > leaf:
        nxt = []
        for ix in idx:
            sub = C[ix]
            a = sub[torch.randint(0, sub.shape[0], (1,), device=C.device, generator=g)]
            b = sub[((sub - a) ** 2).sum(1).argmax()][None]
            for _ in range(5):
                lab = ((sub - a) ** 2).sum(1) > ((sub - b) ** 2).sum(1)
                if lab.all() or (~lab).all(): break
                a, b = sub[~lab].mean(0, keepdim=True), sub[lab].mean(0, keepdim=True)
            o = torch.argsort((sub @ (b - a).T).squeeze(1))
            hh = ix.numel() // 2
            nxt += [ix[o[:hh]], ix[o[hh:]]]
        idx = nxt
    return torch.cat(idx)

order, hs = be.visit_order(ids0, dis0, k)
o = order.long() & 0xffffffff; p = hs.long() & 0xffffffff
dnew = ((xs[o] - C1[p]) ** 2).sum(1)
tau = 2 * (dnew + 1.6e-5).sqrt()
n32 = (o.numel() // 32) * 32
for name, ordc in (("axis kd (product)", torch.from_numpy(be.group_rows_kd(be.to_host(C1)).astype(np.int64)).cuda()),
                   ("principal-axis bisection", pca_groups(C1)), ("2-means bisection", kmeans_balanced_groups(C1))):
    cperm = ordc.int().contiguous()
    dmin = be.group_min_dist(C1, cperm)
    tot = 0.0; cnt = 0
    step = 4096 * 32
    for s in range(0, n32, step):
        e = min(n32, s + step)
        nd = (dmin[p[s:e]] <= tau[s:e, None]).view(-1, 32, dmin.shape[1]).any(1)
        tot += nd.float().sum().item(); cnt += nd.numel()
    print(f"{name}: accumulators needed {100*tot/cnt:.1f}%")

# ---- would a second, row-centred bound help?  |x - c| >= |x - m_g| - rad_g
ordc = torch.from_numpy(be.group_rows_kd(be.to_host(C1)).astype(np.int64)).cuda()
cperm = ordc.int().contiguous()
dmin = be.group_min_dist(C1, cperm)
G = C1[ordc].view(-1, 32, 64)
mg = G.mean(1)
rad = ((G - mg[:, None, :]) ** 2).sum(2).sqrt().max(1).values
xo = xs[o]
R = (dnew + 1.6e-5).sqrt()
tot1 = tot2 = cnt = 0
step = 2048 * 32
for s in range(0, n32, step):
    e = min(n32, s + step)
    n1 = dmin[p[s:e]] <= tau[s:e, None]
    dg = torch.cdist(xo[s:e], mg)
    n2 = n1 & ((dg - rad[None, :]) <= R[s:e, None] * 1.0001 + 1e-4)
    tot1 += n1.view(-1, 32, n1.shape[1]).any(1).float().sum().item()
    tot2 += n2.view(-1, 32, n2.shape[1]).any(1).float().sum().item()
    cnt += n1.numel() // 32
print(f"centroid bound only: {100*tot1/cnt:.1f}%   with the row-centred group bound too: {100*tot2/cnt:.1f}%")

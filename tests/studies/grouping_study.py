"""Grouping study (CPU only; not a test -- run by hand: python tests/studies/grouping_study.py [workdir]).

How many of the 32-centroid groups must a 32-row tile of the exact pruned sweep visit, as a function of how the
centroids are grouped?  Frames come from the oracle's log-mel of the synthetic clips (220 frames of each of 2400
clips), centroids from 12 plain Lloyd iterations (k = 2048), the grouping from the product's own host helper.
Result of round 2: principal-axis tree 29.7 % of the groups per tile, the same tree refined by
capacity-constrained k-means 29.0 %, random groups 55.7 %; the per-centroid Elkan set of a tile is 16.4 %."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
W = sys.argv[1] if len(sys.argv) > 1 else "/tmp/grouping_study"
os.makedirs(W, exist_ok=True)

# ---- 1. frames ----
if not os.path.exists(W + '/x.npy'):
    import oracle
    from audio_tokens_amd.synth import synth_clips
    oracle.build()
    t=time.time()
    wave = synth_clips(2400, L=220500, seed=4242, first_clip=0, device="cpu").numpy()
    rng = np.random.default_rng(0)
    rows=[]
    for w in wave:
        s = oracle.logmel(w, n_mels=64, hop=128).T.astype(np.float32)   # [T, 64]
        rows.append(s[rng.choice(s.shape[0], 220, replace=False)])
    x = oracle.l2norm_rows(np.concatenate(rows))
    print(x.shape, time.time()-t)
    np.save(W + "/x.npy", x)

# ---- 2. centroids ----
if not os.path.exists(W + '/c.npy'):
    import torch
    torch.set_num_threads(8)
    x = torch.from_numpy(np.load(W + "/x.npy"))
    n, d = x.shape
    k = 2048
    g = torch.Generator().manual_seed(1)
    c = x[torch.randperm(n, generator=g)[:k]].clone()
    def assign(x, c):
        cn = (c*c).sum(1)
        best = torch.empty(n, dtype=torch.long); bd = torch.empty(n)
        for s in range(0, n, 65536):
            D = cn[None,:] - 2*x[s:s+65536] @ c.T
            v, i = D.min(1); best[s:s+65536] = i; bd[s:s+65536] = v
        return best, (bd + (x*x).sum(1)).clamp_min(0)
    for it in range(12):
        ids, dis = assign(x, c)
        sums = torch.zeros(k, d).index_add_(0, ids, x); cnt = torch.bincount(ids, minlength=k).float()
        c = torch.where(cnt[:,None] > 0, sums / cnt[:,None].clamp_min(1), c)
    ids, dis = assign(x, c)
    np.save(W + "/c.npy", c.numpy()); np.save(W + "/ids.npy", ids.numpy()); np.save(W + "/dis.npy", dis.numpy())
    print("objective", float(dis.sum()), "mean R", float(dis.sqrt().mean()))

# ---- 3. groupings ----
import torch
from audio_tokens_amd.backend import HostHelpers
c = np.load(W + "/c.npy"); ids = np.load(W + "/ids.npy"); dis = np.load(W + "/dis.npy")
k, d = c.shape; ng = k // 32
R = np.sqrt(dis)
cc = torch.from_numpy(c)
D = torch.cdist(cc, cc)            # [k,k]
order = np.lexsort((dis, ids))     # by cluster, then distance
def evaluate(groups, name):       # groups: [ng,32] centroid indices
    gi = torch.from_numpy(groups.astype(np.int64))
    dmin = D[:, gi.reshape(-1)].reshape(k, ng, 32).min(2).values.numpy()      # [k, ng]
    need_row = dmin[ids] <= 2 * R[:, None]                                    # [n, ng]
    fr = need_row.mean()
    nt = len(order) // 32
    nt_rows = order[:nt*32].reshape(nt, 32)
    need_tile = need_row[nt_rows].any(1)
    # group radius statistic
    gm = c[groups].mean(1)
    rad = np.sqrt(((c[groups] - gm[:, None, :])**2).sum(2)).mean()
    print(f"{name:28s} per-row needed {fr:.4f}   per-tile needed {need_tile.mean():.4f}   mean member-to-group-mean {rad:.4f}")
hh = HostHelpers()
perm = hh.group_rows_kd(c).reshape(ng, 32)
assert (perm >= 0).all()
evaluate(perm, "PCA tree (current)")
rng = np.random.default_rng(0)
evaluate(rng.permutation(k).reshape(ng, 32), "random")
def balanced_refine(groups, iters):
    groups = groups.copy()
    for it in range(iters):
        gm = c[groups].mean(1)                                         # [ng, d]
        dist = ((c[:, None, :] - gm[None, :, :])**2).sum(2) if k*ng*d < 3e8 else None
        if dist is None:
            dist = (c*c).sum(1)[:, None] - 2 * c @ gm.T + (gm*gm).sum(1)[None, :]
        # greedy: pairs ascending
        flat = np.argsort(dist, axis=None)
        room = np.full(ng, 32); where = np.full(k, -1)
        left = k
        for f in flat:
            i, g = divmod(int(f), ng)
            if where[i] < 0 and room[g] > 0:
                where[i] = g; room[g] -= 1; left -= 1
                if left == 0: break
        groups = np.argsort(where, kind="stable").reshape(ng, 32)
    return groups
t = time.time()
for iters in (1, 3, 8):
    g2 = balanced_refine(perm, iters)
    evaluate(g2, f"PCA tree + {iters} balanced k-means")
print("time", time.time() - t)
# per-centroid Elkan fraction
Dn = D.numpy()
frs = []
for s in range(0, len(ids), 50000):
    sl = slice(s, s + 50000)
    frs.append((Dn[ids[sl]] <= 2 * R[sl, None]).mean())
print("per-row needed CENTROIDS fraction", np.mean(frs))
nt = len(order) // 32
rows_t = order[:nt*32].reshape(nt, 32)
Rmax = R[rows_t].max(1); pt = ids[rows_t]
same = (pt == pt[:, :1]).all(1)
fr_t = []
for s in range(0, nt, 2000):
    sl = slice(s, s + 2000)
    fr_t.append((Dn[pt[sl, 0]] <= 2 * Rmax[sl, None]).mean())
print("per-tile (first guess, max radius) needed centroids fraction", np.mean(fr_t), "tiles with one guess", same.mean())

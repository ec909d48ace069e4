"""How many rows could skip the exact sweep of a Lloyd iteration if every row kept ONE lower bound on its distance to
the second-nearest centroid (Hamerly), decayed per iteration only by the movement of the centroids Elkan's test admits
for the row's cluster?  (CPU study, run by hand; uses the frames of tests/studies/grouping_study.py.)

Per iteration: ub = distance to the (updated) own centroid; lb <- min(lb - M_p, E_p - ub) with M_p = the largest
movement among the centroids within 2 Rmax_p of c_p, E_p = the nearest centroid distance beyond that radius; a row is
settled when ub < lb.  Unsettled rows are searched exactly (numpy) and get fresh bounds.  Printed per iteration: the
fraction settled, the fraction that really changed cluster, and how many 32-row tiles of UNSETTLED rows remain."""
import os, sys
import numpy as np, torch
W = sys.argv[1] if len(sys.argv) > 1 else "/tmp/grouping_study"
torch.set_num_threads(8)
x = torch.from_numpy(np.load(W + "/x.npy"))
n, d = x.shape
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
g = torch.Generator().manual_seed(1)

def search2(xs, c):
    """nearest and second nearest (true distances)"""
    cn = (c * c).sum(1)
    b1 = torch.empty(len(xs)); b2 = torch.empty(len(xs)); i1 = torch.empty(len(xs), dtype=torch.long)
    for s in range(0, len(xs), 32768):
        D = (cn[None, :] - 2 * xs[s:s + 32768] @ c.T + (xs[s:s + 32768] ** 2).sum(1)[:, None]).clamp_min(0)
        v, i = D.topk(2, dim=1, largest=False)
        b1[s:s + 32768] = v[:, 0].sqrt(); b2[s:s + 32768] = v[:, 1].sqrt(); i1[s:s + 32768] = i[:, 0]
    return i1, b1, b2

def update(xs, ids, c):
    sums = torch.zeros(k, d).index_add_(0, ids, xs)
    cnt = torch.bincount(ids, minlength=k).float()
    return torch.where(cnt[:, None] > 0, sums / cnt[:, None].clamp_min(1), c)

def run(xs, c, niter, label):
    ids, ub, lb = search2(xs, c)
    for it in range(1, niter):
        c_new = update(xs, ids, c)
        move = (c_new - c).norm(dim=1)
        c = c_new
        ub = (xs - c[ids]).norm(dim=1)                                   # exact, every row (the sweep's pre-pass does it anyway)
        Dcc = torch.cdist(c, c)
        rmax = torch.zeros(k).scatter_reduce_(0, ids, ub, "amax")
        adm = Dcc <= 2 * rmax[:, None]
        M = torch.where(adm, move[None, :].expand(k, k), torch.zeros(())).amax(1)
        E = torch.where(adm, torch.full((), float("inf")), Dcc).amin(1)
        lb = torch.minimum(lb - M[ids], E[ids] - ub)
        settled = ub < lb
        uns = (~settled).nonzero().squeeze(1)
        i1, b1, b2 = search2(xs[uns], c)
        changed = float((i1 != ids[uns]).float().sum()) / len(xs)
        ids[uns] = i1; ub[uns] = b1; lb[uns] = b2
        print(f"{label} it {it:2d}: settled {float(settled.float().mean()):.3f}  changed cluster {changed:.4f}  "
              f"max move {float(move.max()):.3f} median move {float(move.median()):.4f}", flush=True)
    return c

half = n // 2
c0 = x[torch.randperm(half, generator=g)[:k]].clone()
c1 = run(x[:half], c0, 20, "cold ")
run(x[half:], c1, 20, "warm ")

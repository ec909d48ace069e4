"""Would a projected bounding-box test prune the exact sweep's group visits beyond Elkan's test?  (CPU study, run by hand.)

Elkan: a 32-row tile (rows of one cluster p, similar radius) needs group g iff min_{c in g}|c - c_p| <= 2 Rmax.
Box test: project rows and centroids on q orthonormal directions (the leading principal axes of the centroid table);
in that subspace distances only shrink, so the squared distance between any row of the tile and any member of g is at
least the squared gap between the tile's box and the group's box.  The tile needs g only if that gap <= Rmax.
Also a per-row variant (each row against the group box) to see what the tile granularity costs.
Printed: fraction of (tile, group) pairs needed under Elkan / box / both, and the ideal."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from audio_tokens_amd.backend import HostHelpers
W = sys.argv[1] if len(sys.argv) > 1 else "/tmp/grouping_study"
x = np.load(W + "/x.npy"); c = np.load(W + "/c.npy"); ids = np.load(W + "/ids.npy"); dis = np.load(W + "/dis.npy")
n, d = x.shape; k = c.shape[0]; ng = k // 32
torch.set_num_threads(8)
R = np.sqrt(dis)
groups = HostHelpers().group_rows_kd(c).reshape(ng, 32)
cc = torch.from_numpy(c)
D = torch.cdist(cc, cc)
dmin = D[:, torch.from_numpy(groups.reshape(-1).astype(np.int64))].reshape(k, ng, 32).min(2).values.numpy()
order = np.lexsort((dis, ids))
nt = n // 32
tiles = order[:nt * 32].reshape(nt, 32)
cm = c - c.mean(0)
U, S, Vt = np.linalg.svd(cm, full_matrices=False)
print("singular values (first 16):", np.round(S[:16], 2))
gidx = torch.from_numpy(groups.astype(np.int64))
for q in (4, 8, 16, 32, 64):
    P = Vt[:q].T.astype(np.float32)                     # [d, q] orthonormal
    xc = x @ P; cp = c @ P
    glo = cp[groups].min(1); ghi = cp[groups].max(1)    # [ng, q]
    tot = {"elkan": 0.0, "box": 0.0, "both": 0.0, "rowbox_both": 0.0}
    ns = 0
    for s in range(0, nt, 1024):
        t = tiles[s:s + 1024]
        Rmax = R[t].max(1)                              # [T]
        lo = xc[t].min(1); hi = xc[t].max(1)            # [T, q]
        gap = np.maximum(0, np.maximum(glo[None] - hi[:, None], lo[:, None] - ghi[None]))   # [T, ng, q]
        box = (gap ** 2).sum(2) <= (Rmax ** 2)[:, None]
        elk = (dmin[ids[t]] <= 2 * R[t][:, :, None]).any(1)                                  # [T, ng]
        # per-row box test, OR over the tile's rows
        xr = xc[t]                                                                            # [T, 32, q]
        gapr = np.maximum(0, np.maximum(glo[None, None] - xr[:, :, None], xr[:, :, None] - ghi[None, None]))
        rowbox = ((gapr ** 2).sum(3) <= (R[t] ** 2)[:, :, None])
        rb = (rowbox & (dmin[ids[t]] <= 2 * R[t][:, :, None])).any(1)
        tot["elkan"] += elk.mean() * len(t); tot["box"] += box.mean() * len(t); tot["both"] += (elk & box).mean() * len(t)
        tot["rowbox_both"] += rb.mean() * len(t)
        ns += len(t)
        if ns >= 4096: break
    print(f"q={q:2d}: " + "  ".join(f"{name} {v / ns:.4f}" for name, v in tot.items()))

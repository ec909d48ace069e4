"""Would a per-ROW test prune the exact sweep better than Elkan's per-cluster test?  (CPU study, run by hand:
python tests/studies/ball_test_study.py [workdir]; uses the frames / centroids tests/studies/grouping_study.py leaves.)

Elkan's lemma admits a group g for a row x with guess p when min_{c in g} |c - c_p| <= 2 |x - c_p|: it knows where x is
only up to the sphere around c_p.  A test that uses x itself: the group's members lie in the ball (m_g, r_g) around the
group mean, so g can hold a centroid within R = |x - c_p| of x only if |x - m_g| - r_g <= R -- one distance per row and
GROUP (k/32 of them: 1/32 of the dense work) instead of none.  Printed: the fraction of groups needed per row and per
32-row tile (rows in (cluster, distance) order) under Elkan's test, the ball test, both, and the ideal (some member
really is within R)."""
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
gm = c[groups].mean(1)                                                   # [ng, d]
rg = np.sqrt(((c[groups] - gm[:, None, :]) ** 2).sum(2)).max(1)          # [ng]
cc = torch.from_numpy(c)
D = torch.cdist(cc, cc)
dmin = D[:, torch.from_numpy(groups.reshape(-1).astype(np.int64))].reshape(k, ng, 32).min(2).values.numpy()
order = np.lexsort((dis, ids))
nt = n // 32
tiles = order[:nt * 32].reshape(nt, 32)
acc = {name: [0.0, 0.0] for name in ("elkan", "ball", "both", "ideal")}
xt = torch.from_numpy(x); gmt = torch.from_numpy(gm)
gidx = torch.from_numpy(groups.astype(np.int64))
CH = 32 * 512
for s in range(0, nt * 32, CH):
    rows = order[s:s + CH]
    xr = xt[rows]
    Rr = torch.from_numpy(R[rows])[:, None]
    elk = torch.from_numpy(dmin[ids[rows]]) <= 2 * Rr
    dm = torch.cdist(xr, gmt)                                            # [rows, ng]
    ball = dm - torch.from_numpy(rg)[None, :] <= Rr
    dall = torch.cdist(xr, cc)                                           # [rows, k]
    ideal = dall[:, gidx.reshape(-1)].reshape(len(rows), ng, 32).min(2).values <= Rr * (1 + 1e-6)
    for name, m in (("elkan", elk), ("ball", ball), ("both", elk & ball), ("ideal", ideal)):
        acc[name][0] += float(m.float().mean()) * len(rows)
        acc[name][1] += float(m.reshape(-1, 32, ng).any(1).float().mean()) * len(rows)
tot = nt * 32
print(f"n={n} k={k} groups={ng}; mean R {R.mean():.4f}, mean group radius {rg.mean():.4f}")
for name, (a, b) in acc.items():
    print(f"{name:6s}: per-row needed {a / tot:.4f}   per-tile needed {b / tot:.4f}")

"""Times stage 1 (pre-pass + filter sweep, at_filter_probe_f32) of the Lloyd shape; with
AT_FILTER_ABLATE set the sweep skips parts of its work (timing experiments, results are wrong)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_tokens_amd.backend import default_backend
from audio_tokens_amd.synth import synth_clips
be = default_backend()
wave = synth_clips(1218, L=220500, seed=4242, device=be.device)
x = be.logmel(wave, 22050, 512, 128, 64, frame_major=True, l2norm=True)
n, d = x.shape; k = 8192
g = torch.Generator(device="cuda").manual_seed(1)
c = x[torch.randperm(n, device="cuda", generator=g)[:k]].clone()
os.environ.pop("AT_FILTER_ABLATE_SAVED", None)
abl = os.environ.pop("AT_FILTER_ABLATE", None)
for it in range(4):
    ids, dis = be.assign(x, c)
    c2, h = be.centroid_finalize(be.centroid_accum(x, ids, k), k, d)
    c = torch.where(h[:, None] > 0, c2, c).contiguous()
cperm = be.from_host(be.group_rows_kd(be.to_host(c)))
dmin = be.group_min_dist(c, cperm)
order = be.visit_order(ids, dis, k)
def run():
    be.filter_probe(x, c, order, cperm, dmin)
modes = (("full", None), ("no loads", "1"), ("no screen", "2"), ("no mfma", "4"), ("no loads+screen", "3"),
                   ("no screen+mfma", "6"), ("nothing", "7"), ("nothing, contiguous rows", "15"), ("full, contiguous rows", "8"))
if len(sys.argv) > 1:
    modes = tuple(m for m in modes if m[1] == (None if sys.argv[1] == "full" else sys.argv[1]))
for label, val in modes:
    if val is None: os.environ.pop("AT_FILTER_ABLATE", None)
    else: os.environ["AT_FILTER_ABLATE"] = val
    run(); be.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): run()
    be.synchronize()
    print(f"{label:18s} {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms (pre-pass included)")

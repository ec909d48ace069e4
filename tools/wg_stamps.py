"""Where a wave of assign_f16filter_wg_kernel spends its cycles: runs one Lloyd-shaped exact sweep on the DIAGNOSTIC
build of the library (csrc/filter_wg.hip compiled with -DAT_WG_STAMPS into libaudio_tokens_amd_stamps.so; s_memtime
stamps leave through the statistics records, results unchanged).  Development aid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathlib import Path
from audio_tokens_amd import _lib
_lib.LIB_PATH = Path(_lib.LIB_PATH).with_name("libaudio_tokens_amd_stamps.so")
import torch
from audio_tokens_amd.backend import default_backend
from audio_tokens_amd.synth import synth_clips

be = default_backend()
be.debug_set("filter_timing", 1)
be.debug_set("filter_stats", 1)
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
order = be.visit_order(ids, dis, k)
for nb in (int(a) for a in (sys.argv[1:] or ["0"])):
    be.debug_set("filter_nb", nb)
    be.assign_pruned(x, c, order, cperm, dmin, filter=True)
    be.synchronize()
    be.filter_stats(reset=True); be.prune_stats(reset=True)
    reps = 3
    for _ in range(reps):
        be.assign_pruned(x, c, order, cperm, dmin, filter=True)
    be.synchronize()
    rows, listed, ms, sweeps, t_comp, t_ref = be.filter_stats(timing=True)
    t_pro, t_wait = be.prune_stats()
    waves = reps * (-(-n // (128 * (1 if nb == 1 else 2)))) * 4
    f = 16.0 / waves
    print(f"filter_nb={nb}: kernel {ms / sweeps * 1e3:.0f} us; per wave (cycles): prologue+list {t_pro * f:.0f}, walk waits {t_wait * f:.0f}, "
          f"walk work {t_comp * f:.0f}, refinement {t_ref * f:.0f}", flush=True)

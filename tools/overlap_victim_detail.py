"""Development aid: what is wrong in a log-mel frame computed beside the guess-mode sweep."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_tokens_amd.backend import default_backend
from audio_tokens_amd.ops import Kmeans
from audio_tokens_amd.synth import synth_clips
be = default_backend()
T, k = 1723, 8192
wt = synth_clips(6000, L=220500, seed=4242, first_clip=0, device="cuda")
main = torch.cuda.current_stream()
bg = be.background_stream() if os.environ.get("BG", "1") == "1" else torch.cuda.Stream()
torch.set_printoptions(linewidth=250, precision=3, sci_mode=False)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    fr = be.logmel(wt[:3000], frame_major=True, l2norm=True)
    km = Kmeans(64, k, niter=5, backend=be)
    km.train(fr)
    C = km.centroids_device.clone()
    xs = fr[:2097152].contiguous()
    cperm = be.from_host(be.group_rows_kd(be.to_host(C)))
    means = be.group_means(C, cperm)
    for name, fn in {"frame-major dB": lambda c0: be.logmel(wt[c0:c0 + 50], frame_major=True),
                     "mel-major dB": lambda c0: be.logmel(wt[c0:c0 + 50])}.items():
        quiet = [fn(c0).clone() for c0 in range(0, 6000, 50)]
        torch.cuda.synchronize()
        ev = torch.cuda.Event(); ev.record(main)
        with torch.cuda.stream(bg):
            bg.wait_event(ev)
            got = [fn(c0) for c0 in range(0, 6000, 50)]
            done = torch.cuda.Event(); done.record(bg)
        for _ in range(25):
            be._nearest_mean(xs, means)
        main.wait_event(done); torch.cuda.synchronize()
        shown = 0
        hist_t = torch.zeros(32, dtype=torch.long); hist_m = torch.zeros(64, dtype=torch.long); nbad = 0
        for li, (a, b) in enumerate(zip(quiet, got)):
            if name.startswith("frame"):
                a3, b3 = a.view(50, T, 64), b.view(50, T, 64)
            else:
                a3, b3 = a.transpose(1, 2), b.transpose(1, 2)         # [clip][t][mel]
            bad = (a3.view(torch.int32) != b3.view(torch.int32)) if a3.is_contiguous() else (a3 != b3)
            if not bad.any():
                continue
            idx = bad.nonzero()
            hist_t += torch.bincount((idx[:, 1] % 32).cpu(), minlength=32)
            hist_m += torch.bincount(idx[:, 2].cpu(), minlength=64)
            rows = torch.unique(idx[:, 0] * T + idx[:, 1])
            nbad += rows.numel()
            for r in rows[:2].tolist():
                if shown < 6:
                    c, t = r // T, r % T
                    print(f"{name}: launch {li} clip {c} t {t} (t%32 {t % 32}): {int(bad[c, t].sum())} of 64 mel values differ; diff (dB) =")
                    print("   ", (b3[c, t] - a3[c, t]))
                    shown += 1
        print(f"{name}: {nbad} wrong frames; t%32 histogram {hist_t.tolist()}; mel histogram {hist_m.tolist()}", flush=True)

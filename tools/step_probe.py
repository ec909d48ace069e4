"""Where does a pipeline step's time go: steps back to back vs separated by a synchronisation, with and
without bench.py's per-launch events.  python tools/step_probe.py [train_clips val_clips]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from audio_tokens_amd.backend import default_backend
from audio_tokens_amd.pipeline import DevicePipeline
from audio_tokens_amd.synth import synth_clips

n_tr = int(sys.argv[1]) if len(sys.argv) > 1 else 22500
n_va = int(sys.argv[2]) if len(sys.argv) > 2 else 2500
be = default_backend("cuda:0")
wt = synth_clips(n_tr, device="cuda:0"); wv = synth_clips(n_va, first_clip=n_tr, device="cuda:0")
pipe = DevicePipeline(n_mels=64, vocab_size=8192, niter=20, backend=be)
for _ in range(2):
    pipe.run(wt, wv); torch.cuda.synchronize()
for label, trace, sync in (("sync each, no trace", False, True), ("back to back, no trace", False, False),
                           ("sync each, trace", True, True), ("back to back, trace", True, False)):
    be.assign_trace = [] if trace else None
    torch.cuda.synchronize(); t0 = time.perf_counter(); host = []
    for _ in range(4):
        h0 = time.perf_counter(); pipe.run(wt, wv); host.append((time.perf_counter() - h0) * 1e3)
        if sync: torch.cuda.synchronize()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 4 * 1e3
    print(f"{label:26s}: {dt:7.1f} ms/step   host-side per step {[round(h, 1) for h in host]}", flush=True)
be.assign_trace = None
print("stage split:", pipe.run(wt, wv, timing=True).stage_seconds)

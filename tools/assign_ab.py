"""A/B of the at_assign_f32 kernel variants in one process (interleaved rounds), with a bit-exact
check of every variant against variant 0.  Development aid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_tokens_amd.backend import default_backend
be = default_backend()
variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,1,2,3,4".split(","))]
shapes = [(2097152, 64, 8192), (2097152, 128, 8192)] if len(sys.argv) < 3 else [tuple(int(v) for v in a.split("x")) for a in sys.argv[2:]]
g = torch.Generator(device="cuda").manual_seed(0)
for (n, d, k) in shapes:
    x = torch.nn.functional.normalize(torch.randn(n, d, device="cuda", generator=g), dim=1)
    c = torch.nn.functional.normalize(torch.randn(k, d, device="cuda", generator=g), dim=1)
    ref = None
    times = {v: [] for v in variants}
    for rnd in range(4):
        for v in variants:
            be.debug_set("assign_variant", v)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ids, dis = be.assign(x, c); e1.record(); torch.cuda.synchronize()
            if rnd > 0: times[v].append(e0.elapsed_time(e1))
            if ref is None: ref = (ids.clone(), dis.clone())
            elif rnd == 0:
                ok = torch.equal(ids, ref[0]) and torch.equal(dis.view(torch.int32), ref[1].view(torch.int32))
                print(f"  variant {v} bit-exact vs variant {variants[0]}: {ok}", flush=True)
    for v in variants:
        t = sorted(times[v]); med = t[len(t)//2]
        print(f"n={n} d={d} k={k} variant {v}: median {med:.3f} ms min {t[0]:.3f}  {2*n*d*k/med/1e9:.1f} TFLOP/s", flush=True)

"""Which packed-fp32 instruction returns wrong values beside the fp16 filter in guess mode?
Victims: tools/probes/pk_probe.hip (one packed instruction each, checked in place against the unpacked one) on the
background stream; aggressor: HipBackend._nearest_mean (the library's guess-mode sweep) on the main stream.
Build first: tools/probes/build_pk_probe.sh."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from audio_tokens_amd.backend import default_backend

be = default_backend()
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes", "libpkprobe.so"))
lib.pk_probe_launch.restype = ctypes.c_int
lib.pk_probe_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p]
rng = np.random.default_rng(5)
k, n = 8192, 1 << 20
c = rng.standard_normal((k, 64)).astype(np.float32)
c /= np.linalg.norm(c, axis=1, keepdims=True)
C = be._f32(c)
xs = torch.nn.functional.normalize(torch.randn(n, 64, device="cuda"), dim=1)
cperm = be.from_host(be.group_rows_kd(c))
means = be.group_means(C, cperm)
be._nearest_mean(xs, means)
main, bg = torch.cuda.current_stream(), be.background_stream()
names = ["v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32", "v_pk_mov_b32", "pk_add op_sel swap", "pk_add bcast+neg", "pk_mul V,V,S", "pk_mov op_sel:[1,0]", "pk_add neg only", "pk_fma V,V,0 op_sel_hi"]
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
for lds in (0,):
    for vregs in (6, 24, 100):          # 100: the operands pass through LDS (written by the neighbour lane, 128-bit read back)
        for op in range(10):
            res = []
            for busy in (False, True):
                out = torch.zeros(5, dtype=torch.int64, device="cuda")
                torch.cuda.synchronize()
                ev0 = torch.cuda.Event(enable_timing=True); ev0.record(main)
                if busy:
                    for _ in range(80):
                        be._nearest_mean(xs, means)
                main_end = torch.cuda.Event(enable_timing=True); main_end.record(main)
                with torch.cuda.stream(bg):
                    bg.wait_event(ev0)
                    rc = lib.pk_probe_launch(op, iters, vregs, lds, out.data_ptr(), bg.cuda_stream)
                    assert rc == 0, rc
                    bg_end = torch.cuda.Event(enable_timing=True); bg_end.record(bg)
                torch.cuda.synchronize()
                o = out.cpu().tolist()
                res.append((o, ev0.elapsed_time(bg_end), ev0.elapsed_time(main_end)))
            (q, tq, _), (b_, tb, tm) = res
            print(f"{names[op]:23s} operands/lane {'6 via LDS' if vregs == 100 else vregs:>9} LDS {lds // 1024:2d} KB: quiet mismatches {q[:4]} ({tq:.1f} ms); beside the guess-mode sweeps "
                  f"{b_[:4]} of {b_[4] * 64 * 2048:.2e} checked (victim {tb:.1f} ms, sweeps {tm:.1f} ms)", flush=True)

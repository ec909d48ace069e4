#!/usr/bin/env python3
"""Per-kernel table of ONE pipeline step from a rocprofv3 kernel trace (csv): tools/step_trace.py <kernel_trace.csv> [step]
The trace is of `bench.py --steps 1 --warmup 1 --no-dense-floor --no-cpu-baseline`: the steps are delimited by the
logmel launches (6 per step).  Prints total time per kernel inside the chosen step and the durations of the
split_clusters launches in order."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
step = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"]
lm = [i for i, r in enumerate(rows) if "logmel_kernel" in name(r)]
per = 6
b, e = lm[step * per], (lm[(step + 1) * per] if len(lm) > (step + 1) * per else len(rows))
sel = rows[b:e]
t0, t1 = int(sel[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in sel)
print(f"step {step}: {len(sel)} launches, wall {(t1 - t0) / 1e6:.2f} ms")
tot, cnt = defaultdict(float), defaultdict(int)
for r in sel:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot[name(r)[:110]] += d
    cnt[name(r)[:110]] += 1
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:40]:
    print(f"{v / 1e3:9.3f} ms {cnt[k]:5d} x {v / cnt[k]:9.1f} us  {k}")
print("sum of kernel time %.2f ms" % (sum(tot.values()) / 1e3))
sp = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in sel if "split_clusters" in name(r)]
print("split_clusters us:", " ".join(f"{x:.0f}" for x in sp))
# idle gaps on the device: time no kernel was running
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in sel)
busy, cur_s, cur_e = 0, ev[0][0], ev[0][1]
for s, e_ in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e_
    else:
        cur_e = max(cur_e, e_)
busy += cur_e - cur_s
print(f"device busy {busy / 1e6:.2f} ms of {(t1 - t0) / 1e6:.2f} ms")
for pat in ("assign_f16filter_kernel<64, 2, false, true", "centroid_accum_kernel", "exact_rows_kernel"):
    sp = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in sel if pat in name(r)]
    print(pat, "us:", " ".join(f"{x:.0f}" for x in sp))

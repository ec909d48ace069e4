"""Reads a rocprofv3 --kernel-trace of tools/kmeans_small.py and prints, for the last Lloyd iterations,
each kernel's start offset, duration and the idle gap before it (per stream).  Development aid."""
import csv, glob, sys
trace = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
tr = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
# an iteration starts at each exact filter sweep
starts = [i for i, r in enumerate(tr) if "f16filter" in r["Kernel_Name"] and "false, true" in r["Kernel_Name"]]
a, b = starts[-4], starts[-2]
t0 = int(tr[a]["Start_Timestamp"])
busy_end = t0
busy = 0
for r in tr[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - busy_end) / 1e3
    name = r["Kernel_Name"].replace("void ", "").replace("rocprim::ROCPRIM_400200_NS::detail::", "")[:90]
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  gap {gap:7.1f}  q{r.get('Queue_Id', '?')}  {name}")
    if e > busy_end:
        busy += (e - max(s, busy_end))
        busy_end = e
print(f"two iterations: {(int(tr[b]['Start_Timestamp']) - t0) / 1e3:.1f} us, busy {busy / 1e3:.1f} us")
ts = [int(tr[i]["Start_Timestamp"]) for i in starts[-21:]]
print("sweep-to-sweep us:", " ".join(f"{(y - x) / 1e3:.0f}" for x, y in zip(ts, ts[1:])))

#!/usr/bin/env python3
"""Timeline of ONE Lloyd iteration from a rocprofv3 kernel trace csv: tools/iter_timeline.py <kernel_trace.csv> [which]
Prints every launch between the start of the `which`-th Lloyd filter sweep and the start of the next one:
offset of start and end in us (relative to the sweep's start), queue id, kernel name."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
which = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sw = [i for i, r in enumerate(rows) if "assign_f16filter_kernel<64, " in r["Kernel_Name"] and "false, true" in r["Kernel_Name"]]
a, b = sw[which], sw[which + 1]
t0 = int(rows[a]["Start_Timestamp"])
lo = a
while lo > 0 and int(rows[lo - 1]["End_Timestamp"]) > t0:
    lo -= 1
print(f"iteration wall: {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us, {b - a} launches")
qs = {}
for r in rows[lo:b + 1]:
    q = qs.setdefault(r.get("Queue_Id", "?"), len(qs))
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    if "rocprim" in nm:
        nm = "rocprim:" + nm.split("detail::")[-1][:60]
    print(f"{s:9.1f} {e:9.1f} ({e - s:7.1f})  q{q}  {nm[:90]}")

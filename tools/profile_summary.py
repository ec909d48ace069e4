"""Writes profiles/r01_bench_kernel_stats{.csv,_top.txt} from a rocprofv3 --kernel-trace --stats run of
bench.py (directory with *_kernel_stats.csv and *_kernel_trace.csv) and that run's bench JSON line."""
import csv, glob, json, os, shutil, sys, collections
src, bench_json = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
stats = glob.glob(os.path.join(src, "*kernel_stats.csv"))[0]
trace = glob.glob(os.path.join(src, "*kernel_trace.csv"))[0]
shutil.copy(stats, os.path.join(root, "profiles", "r01_bench_kernel_stats.csv"))
d = json.loads(open(bench_json).read().strip().splitlines()[-1])
rows = list(csv.DictReader(open(stats)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
tr = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
shapes = collections.defaultdict(list)
exact = []
for i, r in enumerate(tr):
    if "f16filter" not in r["Kernel_Name"]:
        continue
    if "false, true" in r["Kernel_Name"]:
        shapes[int(r["Grid_Size_X"])].append(dur(r))
    j = i + 1
    while j < len(tr) and ("rocclr" in tr[j]["Kernel_Name"]):   # skip the runtime's copy / fill kernels
        j += 1
    nxt = tr[j]["Kernel_Name"] if j < len(tr) else ""
    if "exact_dist_todo" in nxt or "exact_dist_visit" in nxt or "exact_rows" in nxt:
        exact.append(dur(r))
out = os.path.join(root, "profiles", "r01_bench_kernel_stats_top.txt")
with open(out, "w") as f:
    f.write("rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline   (MI355X, 1 GPU)\n")
    f.write("5 pipeline runs in the process (1 warm-up + 3 timed + 1 per-stage split); the synthetic-data generation "
            "(at::native kernels) is part of the totals.\n")
    f.write(f"bench line of this run: {d['value']:.4g} frames/s, {d['ms_per_step']:.1f} ms/step\n\n")
    f.write("assign_f16filter_kernel<D,NB,GUESS=false,FUSED=true[,WPS]>, the roofline kernel of bench.py (<64,2,false,true,3> for the Lloyd sweeps, <64,4,false,true> for the long tokenise sweeps):\n")
    for k, v in sorted(shapes.items()):
        f.write(f"  launches of {k:>9d} threads (one 64-lane wave per 64 rows in the <64,2,..,3> form, per 128 rows in <64,4,..>): {len(v):4d}, avg {sum(v) / len(v):9.1f} us\n")
    f.write(f"  exact-mode launches (followed by exact_dist_todo_kernel / exact_rows_kernel; 62 per pipeline run: 60 Lloyd sweeps of\n"
            f"  2 097 152 rows + the tokenise sweeps of 38.8 M and 4.3 M rows): {len(exact)} in this trace, avg {sum(exact) / max(1, len(exact)):.1f} us\n"
            f"  bench.py roofline.avg_launch_ms (HIP events around the same kernel, exact launches of the 3 timed steps): "
            f"{d['roofline']['avg_launch_ms'] * 1e3:.1f} us over {d['roofline']['launches']} launches\n\n")
    f.write(f"{'kernel':100s} {'calls':>6s} {'avg_us':>10s} {'total_ms':>10s} {'%':>6s}\n")
    for r in rows[:40]:
        f.write(f"{r['Name'][:100]:100s} {r['Calls']:>6s} {float(r['AverageNs']) / 1e3:10.1f} {float(r['TotalDurationNs']) / 1e6:10.1f} "
                f"{float(r['TotalDurationNs']) / tot * 100:6.2f}\n")
print(open(out).read()[:1800])

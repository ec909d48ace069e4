"""Builds profiles/ (the tracked, judged summaries) from what tools/collect_profiles.sh left under gpurun_out/.
python tools/make_profiles.py gpurun_out/r02a_profiles gpurun_out/r2_lm_pmc|- r02     ("-": no new log-mel counters; the
committed log-mel entries are kept)"""
import csv, glob, json, os, shutil, sys, collections
src, lm, tag = sys.argv[1], sys.argv[2], sys.argv[3]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(root, "profiles")
last = lambda f: json.loads(open(f).read().strip().splitlines()[-1])

for c in (1, 2, 3):
    f = os.path.join(src, f"bench_config{c}.json")
    if os.path.exists(f):
        json.dump(last(f), open(os.path.join(P, f"{tag}_bench_config{c}.json"), "w"), indent=1)

# ---- kernel trace of the driver command -----------------------------------------------------------------
stats = glob.glob(os.path.join(src, "trace", "*kernel_stats.csv"))[0]
trace = glob.glob(os.path.join(src, "trace", "*kernel_trace.csv"))[0]
shutil.copy(stats, os.path.join(P, f"{tag}_bench_kernel_stats.csv"))
d = last(os.path.join(src, "bench_traced.json"))
rows = [r for r in csv.DictReader(open(stats))]
mine = [r for r in rows if "at::native" not in r["Name"] and "rocclr" not in r["Name"]]
tot = sum(float(r["TotalDurationNs"]) for r in mine)
tr = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
filt = collections.defaultdict(list)
for r in tr:
    if "f16filter" in r["Kernel_Name"] and "false, true" in r["Kernel_Name"]:
        filt[r["Kernel_Name"][r["Kernel_Name"].index("assign_f16filter_kernel"):].split("(")[0]].append(dur(r))
with open(os.path.join(P, f"{tag}_bench_kernel_stats_top.txt"), "w") as f:
    f.write("rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline   (MI355X, 1 GPU)\n")
    f.write("7 pipeline runs in the process (1 warm-up + 1 traced + 3 timed + 1 per-stage split + 1 dense floor); the torch kernels that generate\n"
            "the synthetic waveforms are left out of the table below (they are in the .csv).\n")
    f.write(f"bench line of this run: {d['value']:.4g} frames/s, {d['ms_per_step']:.1f} ms/step, verified={d['verified']}\n\n")
    f.write("exact filter sweeps (the roofline kernel of bench.py), by instantiation:\n")
    for k, v in sorted(filt.items()):
        f.write(f"  {k}: {len(v)} launches, avg {sum(v) / len(v):.1f} us\n")
    allf = [x for v in filt.values() for x in v]
    f.write(f"  all exact-mode launches: {len(allf)}, avg {sum(allf) / len(allf):.1f} us;  bench.py roofline.avg_launch_ms (HIP events the library records\n"
            f"  around the same kernel, timed steps only): {d['roofline']['avg_launch_ms'] * 1e3:.1f} us over {d['roofline']['launches']} launches\n\n")
    f.write(f"{'kernel':104s} {'calls':>6s} {'avg_us':>10s} {'total_ms':>10s} {'%':>6s}\n")
    for r in mine[:36]:
        f.write(f"{r['Name'][:104]:104s} {r['Calls']:>6s} {float(r['AverageNs']) / 1e3:10.1f} {float(r['TotalDurationNs']) / 1e6:10.1f} "
                f"{float(r['TotalDurationNs']) / tot * 100:6.2f}\n")

# ---- PMC: filter sweep (Lloyd form) and log-mel -----------------------------------------------------------
def pmc_mean(dirs, pat):
    out = {}
    for dd in dirs:
        for fcsv in glob.glob(dd + "/**/*counter_collection.csv", recursive=True):
            acc = collections.defaultdict(list)
            for r in csv.DictReader(open(fcsv)):
                if pat in r["Kernel_Name"]:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            for k, v in acc.items():
                v = v[len(v) // 2:]
                out[k] = sum(v) / len(v)
    return out

def trace_avg_us(dirs, pat):
    v = []
    for dd in dirs:
        for ft in glob.glob(dd + "/**/*kernel_trace.csv", recursive=True):
            x = [dur(r) for r in csv.DictReader(open(ft)) if pat in r["Kernel_Name"]]
            v += x[len(x) // 2:]
    return sum(v) / len(v) if v else None

traffic = {}
fd = glob.glob(os.path.join(src, "pmc_filter", "*"))
pf = pmc_mean(fd, "f16filter_kernel<64, 2, false, true, 3")
if pf:
    us = trace_avg_us(fd, "f16filter_kernel<64, 2, false, true, 3")
    rd, wr = 2 * pf.get("FETCH_SIZE", 0) * 1024, pf.get("WRITE_SIZE", 0) * 1024
    pf["launch_us_under_pmc"] = us
    json.dump(pf, open(os.path.join(P, f"{tag}_pmc_filter_lloyd.json"), "w"), indent=1)
    traffic["filter_d64"] = {"kernel": "assign_f16filter_kernel<64,2,false,true,3>, warm-start Lloyd sweeps of 2 097 152 rows (tools/kmeans_small.py 1)",
                             "hbm_bytes_per_launch": rd + wr, "read_bytes": rd, "write_bytes": wr, "launch_ms": us / 1e3,
                             "algorithmic_bytes": 2097152 * (256 + 4 + 4 + 8 + 4),
                             "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes, mean of the later half of the launches; "
                                       "FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B: MI355X_MICROARCH.md, HBM section), WRITE_SIZE as reported"}
lp = pmc_mean(glob.glob(os.path.join(lm, "*")), "logmel_kernel") if lm != "-" else None
if not lp and os.path.exists(os.path.join(P, "kernel_traffic.json")):      # keep what is committed
    prev = json.load(open(os.path.join(P, "kernel_traffic.json")))
    if "logmel_d64" in prev:
        traffic["logmel_d64"] = prev["logmel_d64"]
if lp:
    us = trace_avg_us(glob.glob(os.path.join(lm, "*")), "logmel_kernel")
    rd, wr = 2 * lp.get("FETCH_SIZE", 0) * 1024, lp.get("WRITE_SIZE", 0) * 1024
    lp["launch_us_under_pmc"] = us
    lp["frames_per_launch"] = 3446000
    json.dump(lp, open(os.path.join(P, f"{tag}_pmc_logmel.json"), "w"), indent=1)
    traffic["logmel_d64"] = {"kernel": "logmel_kernel<true>, 2000 ten-second clips, n_mels=64, frame-major unit rows (tools/logmel_only.py 64)",
                             "hbm_bytes_per_launch": rd + wr, "read_bytes": rd, "write_bytes": wr, "launch_ms": us / 1e3,
                             "algorithmic_bytes": 3446000 * 768,
                             "method": "as above"}
json.dump(traffic, open(os.path.join(P, "kernel_traffic.json"), "w"), indent=1)
print(open(os.path.join(P, f"{tag}_bench_kernel_stats_top.txt")).read()[:2500])
print(json.dumps(traffic, indent=1)[:1500])

"""Reads a rocprofv3 --pmc counter_collection.csv and prints, for the Lloyd filter sweeps, the average number
of resident waves (SQ_WAVE_CYCLES x 4 / kernel cycles).  Development aid."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if "f16filter" in r["Kernel_Name"] and "false, true" in r["Kernel_Name"]:
        acc[r["Kernel_Name"].split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    m = {n: sum(v[-10:]) / len(v[-10:]) for n, v in c.items()}   # the last ten launches (converged iterations)
    print(k, {n: f"{v:.4g}" for n, v in m.items()})
    if "GRBM_GUI_ACTIVE" in m and "SQ_WAVE_CYCLES" in m:
        cyc = m["GRBM_GUI_ACTIVE"] / 8
        print(f"  kernel cycles {cyc:.4g}; resident waves on average {m['SQ_WAVE_CYCLES'] * 4 / cyc:.0f} of {256 * 4 * 3} slots"
              f" ({m['SQ_WAVE_CYCLES'] * 4 / cyc / (256 * 12) * 100:.0f} %)")

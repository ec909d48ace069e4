"""Reads rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs (two directories) of tools/kmeans_small.py and prints the
mean HBM bytes per launch of the Lloyd filter sweep over its last ten launches, corrected as
MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE counts 128-byte requests at 64 bytes: doubled; both
counters are in KB).  Development aid."""
import csv, glob, sys
def mean_last(dirname, counter, match):
    f = glob.glob(dirname + "/**/*counter_collection.csv", recursive=True)[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
            if r["Counter_Name"] == counter and match in r["Kernel_Name"] and "false, true" in r["Kernel_Name"]]
    return sum(vals[-10:]) / len(vals[-10:]), len(vals)
fetch, nf = mean_last(sys.argv[1], "FETCH_SIZE", "f16filter")
write, nw = mean_last(sys.argv[2], "WRITE_SIZE", "f16filter")
print(f"launches seen {nf}/{nw}; FETCH_SIZE raw {fetch:.0f} KB -> reads {2 * fetch * 1024 / 1e6:.1f} MB; WRITE_SIZE {write:.0f} KB -> writes {write * 1024 / 1e6:.1f} MB; "
      f"total {(2 * fetch + write) * 1024 / 1e6:.1f} MB per launch")

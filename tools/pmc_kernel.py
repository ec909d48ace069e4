"""Mean of every counter of rocprofv3 --pmc runs over the dispatches of one kernel -> JSON on stdout.
python tools/pmc_kernel.py <kernel-name substring> <dir> [<dir> ...]   (development aid)"""
import csv, glob, json, sys, collections
pat, out = sys.argv[1], {}
for d in sys.argv[2:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            v = v[len(v) // 2:]            # (the later dispatches: warm)
            out[k] = sum(v) / len(v)
            out.setdefault("dispatches", len(v))
print(json.dumps(out))

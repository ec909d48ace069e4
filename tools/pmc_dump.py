"""Mean of every counter of a rocprofv3 --pmc run over the last ten Lloyd filter sweeps -> JSON on stdout.  Development aid."""
import csv, glob, json, sys, collections
out = {}
for d in sys.argv[1:]:
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "f16filter" in r["Kernel_Name"] and "false, true" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    out.update({k: round(sum(v[-10:]) / len(v[-10:])) for k, v in acc.items()})
print(json.dumps(out))

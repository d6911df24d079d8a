"""dev tool: per-kernel sums of rocprofv3 PMC counters of a run directory under gpurun_out/ (per launch)."""
import collections, csv, glob, sys
run, keys = sys.argv[1], sys.argv[2:]
f = glob.glob(f"gpurun_out/{run}/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0]
    if any(x in k for x in keys):
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k, c in acc.items():
    m = len(n[k])
    print(k[:40], "launches", m, {a: round(b / m) for a, b in sorted(c.items())})

#!/usr/bin/env python3
"""Dev tool: per-kernel averages of a rocprofv3 --pmc run under gpurun_out/<name> (kernels whose name contains argv[2])."""
import collections, csv, glob, sys
f = glob.glob(f"gpurun_out/{sys.argv[1]}/**/*counter_collection.csv", recursive=True)[0]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if pat in r["Kernel_Name"]:
        d[r["Kernel_Name"].split("(")[0][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in d.items():
    print(k)
    for c, x in sorted(v.items()):
        print(f"   {c:32s} {sum(x) / len(x):16.0f}   (n={len(x)})")

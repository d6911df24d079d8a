#!/bin/bash
cd $GRAFT_REPO_ROOT
VARIANTS=("" "-DTF_AB_NO_READAHEAD" "-DTS_AB_NO_BOX" "${@}")
for defs in "${VARIANTS[@]}"; do
  NF_EXTRA_DEFS="$defs" python -m neurofluid_amd.build > /dev/null 2>&1 || { echo "build failed: $defs"; continue; }
  tag=$(echo "base$defs" | tr -d ' ' | tr -c 'A-Za-z0-9_\n' '_')
  bash tools/prof.sh ab_$tag python $GRAFT_REPO_ROOT/tools/trans_perf.py 30 > /dev/null 2>&1
  echo "== $defs"
  python - "$tag" <<'PY'
import csv, sys, glob
f = glob.glob(f"gpurun_out/ab_{sys.argv[1]}/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n = r["Name"].split("(")[0]
    if any(k in n for k in ("k_trans_",)):
        print(f"   {n[:34]:34s} {float(r['AverageNs'])/1e3:8.1f} us")
PY
done
NF_EXTRA_DEFS="" python -m neurofluid_amd.build > /dev/null 2>&1
rm -rf gpurun_out/ab_*

#!/bin/bash
# A/B of a kernel inside the end-to-end training step (one box): tools/ab_e2e.sh <kernel-substring> "<defs>" ...
cd $GRAFT_REPO_ROOT
K=$1; shift
for defs in "${@}"; do
  NF_EXTRA_DEFS="$defs" python -m neurofluid_amd.build > /dev/null 2>&1 || { echo "build failed: $defs"; continue; }
  tag=$(echo "e2e$defs" | tr -d ' ' | tr -c 'A-Za-z0-9_\n' '_')
  bash tools/prof.sh ab_$tag python $GRAFT_REPO_ROOT/tools/e2e_perf.py 20 > /dev/null 2>&1
  echo "== $defs"; grep train_e2e gpurun_out/ab_$tag/run.log | tail -1
  grep "$K" gpurun_out/ab_$tag/p_kernel_stats.csv | awk -F'",' '{print $1}' | head -0
  python - "$K" gpurun_out/ab_$tag/p_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[2])):
    if sys.argv[1] in r["Name"]:
        print("  %-50s calls %5s avg %8.1f us" % (r["Name"][:50], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
NF_EXTRA_DEFS="" python -m neurofluid_amd.build > /dev/null 2>&1
rm -rf gpurun_out/ab_*

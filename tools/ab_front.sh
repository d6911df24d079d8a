#!/bin/bash
# A/B builds of the front kernels of the transition step (one box): tools/ab_front.sh "<defs>" ...   (AB_FROZEN=frozen: every step from the same state —
# mandatory for switches that corrupt the output; AB_MODE: trans_perf.py's mode)
cd $GRAFT_REPO_ROOT
VARIANTS=("${@}")
for defs in "${VARIANTS[@]}"; do
  NF_EXTRA_DEFS="$defs" python -m neurofluid_amd.build > /dev/null 2>&1 || { echo "build failed: $defs"; continue; }
  tag=$(echo "base$defs" | tr -d ' ' | tr -c 'A-Za-z0-9_\n' '_')
  bash tools/prof.sh ab_$tag python $GRAFT_REPO_ROOT/tools/trans_perf.py 30 ${AB_MODE:-fp32} ${AB_FROZEN:-} > /dev/null 2>&1
  echo "== $defs"; grep iter gpurun_out/ab_$tag/run.log | tail -1
  python tools/kstats.py gpurun_out/ab_$tag/p_kernel_stats.csv 90 12 | grep "k_trans_\|kernel ms"
done
NF_EXTRA_DEFS="" python -m neurofluid_amd.build > /dev/null 2>&1
rm -rf gpurun_out/ab_*

#!/bin/bash
# A/B builds of the G-free conv kernels (one box): tools/ab_gf.sh "<defs>" ...
cd $GRAFT_REPO_ROOT
for defs in "${@}"; do
  NF_EXTRA_DEFS="$defs" python -m neurofluid_amd.build > /dev/null 2>&1 || { echo "build failed: $defs"; continue; }
  tag=$(echo "base$defs" | tr -d ' ' | tr -c 'A-Za-z0-9_\n' '_')
  bash tools/prof.sh ab_$tag python $GRAFT_REPO_ROOT/tools/trans_perf.py 30 ${AB_MODE:-fp32} ${AB_FROZEN:-} > /dev/null 2>&1
  echo "== $defs"; grep iter gpurun_out/ab_$tag/run.log | tail -1
  python tools/kstats.py gpurun_out/ab_$tag/p_kernel_stats.csv 90 12 | grep "k_cconv_gf\|kernel ms"
done
NF_EXTRA_DEFS="" python -m neurofluid_amd.build > /dev/null 2>&1
rm -rf gpurun_out/ab_*

#!/bin/bash
# SQ counters (the TCP / TA counter pass hung the profiler on this pool: not collected) of the G-free conv kernels (dev): tools/pmc_gf.sh "<defs>"
cd $GRAFT_REPO_ROOT
NF_EXTRA_DEFS="$1" python -m neurofluid_amd.build > /dev/null 2>&1 || { echo "build failed"; exit 1; }
TR="python $GRAFT_REPO_ROOT/tools/trans_perf.py 10"
timeout 120 bash tools/pmc.sh pmc_gf_a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" $TR > /dev/null
timeout 120 bash tools/pmc.sh pmc_gf_b "SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_INSTS_MFMA SQ_INSTS_VMEM_WR" $TR > /dev/null
for r in pmc_gf_a pmc_gf_b; do python tools/pmc_quick.py $r k_cconv_gf; done
find gpurun_out/pmc_gf_a gpurun_out/pmc_gf_b -name "*kernel_trace.csv" -delete
NF_EXTRA_DEFS="" python -m neurofluid_amd.build > /dev/null 2>&1

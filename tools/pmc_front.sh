#!/bin/bash
# SQ counters of the front kernels of the fused step (dev)
cd $GRAFT_REPO_ROOT
TR="python $GRAFT_REPO_ROOT/tools/trans_perf.py 10 ${1:-all_pairs}"
bash tools/pmc.sh pmc_front_a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE" $TR > /dev/null
bash tools/pmc.sh pmc_front_b "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU" $TR > /dev/null
python tools/pmc_quick.py pmc_front_a k_trans_front k_trans_stage1
python tools/pmc_quick.py pmc_front_b k_trans_front k_trans_stage1
find gpurun_out/pmc_front_a gpurun_out/pmc_front_b -name "*kernel_trace.csv" -delete

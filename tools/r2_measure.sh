#!/bin/bash
# Round-2 measurement pass on the GPU box: gpu tests (full log), default bench line, strong-scaling control-flow runs
# on one device (world 2/4/8 over gloo), kernel stats and PMC passes for the non-MLP kernels.
# usage (from the repo root on the GPU box): bash tools/r2_measure.sh [tests] [bench] [strong] [pmc] [pmct] [pmcs] [pmch]
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out
WHAT="${@:-tests bench strong pmc}"
for w in $WHAT; do case $w in
tests)
  python -m pytest tests -m gpu -q -s --durations=8 > $O/r2_gputest_full.log 2>&1; tail -4 $O/r2_gputest_full.log ;;
bench)
  python bench.py > $O/r2_bench_default.json 2> $O/r2_bench_default.err; tail -c 600 $O/r2_bench_default.json ;;
strong)
  for img in 400 800; do for n in 1 2 4 8; do
    if [ $n = 1 ]; then
      python bench.py --scaling strong --image $img --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>> $O/r2_strong.err | tail -1 > $O/r2_strong_${img}_w1.json
    else
      NF_BENCH_SINGLE_DEVICE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500+n)) \
        bench.py --gpus $n --scaling strong --image $img --steps 2 --warmup 1 --no-extras 2>> $O/r2_strong.err | tail -1 > $O/r2_strong_${img}_w${n}.json
    fi
  done; done; ls -la $O/r2_strong_* ;;
pmc)
  RB="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1"
  TR="python $GRAFT_REPO_ROOT/tools/trans_perf.py 10"
  bash tools/prof.sh r2_stats_render $RB > /dev/null
  bash tools/prof.sh r2_stats_trans $TR > /dev/null
  bash tools/pmc.sh r2_pmc_fetch_render "FETCH_SIZE" $RB > /dev/null
  bash tools/pmc.sh r2_pmc_write_render "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" $RB > /dev/null
  bash tools/pmc.sh r2_pmc_sq_render "SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" $RB > /dev/null
  bash tools/pmc.sh r2_pmc_fetch_trans "FETCH_SIZE" $TR > /dev/null
  bash tools/pmc.sh r2_pmc_write_trans "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" $TR > /dev/null
  bash tools/pmc.sh r2_pmc_sq_trans "SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" $TR > /dev/null
  # the heavy csv files stay on the box; keep what the summariser needs (counter collections + stats) under 64 MiB
  find $O -name "*kernel_trace.csv" -path "*r2_pmc*" -delete
  du -sh $O/r2_* | tail -12 ;;
pmct)
  TS="python $GRAFT_REPO_ROOT/tools/train_hostprof.py"
  bash tools/prof.sh r2_stats_train $TS > /dev/null
  bash tools/pmc.sh r2_pmc_fetch_train "FETCH_SIZE" $TS > /dev/null
  bash tools/pmc.sh r2_pmc_write_train "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" $TS > /dev/null
  bash tools/pmc.sh r2_pmc_sq_train "SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" $TS > /dev/null
  find $O -name "*kernel_trace.csv" -path "*r2_pmc*" -delete
  grep step $O/r2_stats_train/run.log ;;
pmcs)
  MB="python $GRAFT_REPO_ROOT/tools/mlp_bench.py 3276800 3 split"
  bash tools/prof.sh r2_stats_mlps $MB > /dev/null
  bash tools/pmc.sh r2_pmc_sq_mlps "SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" $MB > /dev/null
  bash tools/pmc.sh r2_pmc_fetch_mlps "FETCH_SIZE" $MB > /dev/null
  bash tools/pmc.sh r2_pmc_write_mlps "WRITE_SIZE" $MB > /dev/null
  find $O -name "*kernel_trace.csv" -path "*r2_pmc*" -delete ;;
pmch)
  MB="python $GRAFT_REPO_ROOT/tools/mlp_bench.py 3276800 3 fp16v3"
  bash tools/prof.sh r2_stats_mlph $MB > /dev/null
  bash tools/pmc.sh r2_pmc_sq_mlph "SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" $MB > /dev/null
  bash tools/pmc.sh r2_pmc_sq2_mlph "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" $MB > /dev/null
  bash tools/pmc.sh r2_pmc_fetch_mlph "FETCH_SIZE" $MB > /dev/null
  bash tools/pmc.sh r2_pmc_write_mlph "WRITE_SIZE" $MB > /dev/null
  find $O -name "*kernel_trace.csv" -path "*r2_pmc*" -delete
  grep -h "fp16 v3" $O/r2_stats_mlph/run.log $O/r2_pmc_sq_mlph/run.log | tail -6 ;;
esac; done

#!/bin/bash
# Round-6 measurement pass on the GPU box.
# usage (from the repo root on the GPU box): bash tools/r6_measure.sh [tests] [bench] [strong] [pmc] [pmct] [e2e] [micro]
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out
WHAT="${@:-tests bench strong pmc}"
SQ="SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE"
for w in $WHAT; do case $w in
tests)
  python -m pytest tests -m gpu -q -s --durations=8 > $O/r6_gputest_full.log 2>&1; tail -4 $O/r6_gputest_full.log ;;
bench)
  python bench.py > $O/r6_bench_default.json 2> $O/r6_bench_default.err; tail -c 400 $O/r6_bench_default.json ;;
strong)
  for img in 400 800; do for n in 2 8; do
      NF_BENCH_SINGLE_DEVICE=1 python bench.py --gpus $n --image $img --steps 2 --warmup 1 --no-extras 2>> $O/r6_strong.err | tail -1 > $O/r6_strong_${img}_w${n}.json
  done; done; ls -la $O/r6_strong_* ;;
pmc)
  RB="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --steps 8 --warmup 3"
  for mode in fp32 split; do
    TR="python $GRAFT_REPO_ROOT/tools/trans_perf.py 30 $mode"
    bash tools/prof.sh r6_stats_trans_$mode $TR > /dev/null
    bash tools/pmc.sh r6_pmc_fetch_trans_$mode "FETCH_SIZE" $TR > /dev/null
    bash tools/pmc.sh r6_pmc_write_trans_$mode "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" $TR > /dev/null
    bash tools/pmc.sh r6_pmc_sq_trans_$mode "$SQ" $TR > /dev/null
  done
  bash tools/prof.sh r6_stats_render $RB > /dev/null
  bash tools/pmc.sh r6_pmc_fetch_render "FETCH_SIZE" $RB > /dev/null
  bash tools/pmc.sh r6_pmc_write_render "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" $RB > /dev/null
  bash tools/pmc.sh r6_pmc_sq_render "$SQ" $RB > /dev/null
  find $O -name "*kernel_trace.csv" -path "*r6_pmc*" -delete
  du -sh $O/r6_* | tail -14 ;;
e2e)
  bash tools/prof.sh r6_stats_e2e python $GRAFT_REPO_ROOT/tools/e2e_perf.py 20 > /dev/null
  grep -E "train_e2e|blocks" $O/r6_stats_e2e/run.log ;;
micro)
  (cd tools/micro && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak mfma_peak.hip && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak_f32 mfma_peak_f32.hip) 2>&1 | tail -2
  (/tmp/mfma_peak; /tmp/mfma_peak_f32) > $O/r6_mfma_peak.txt 2>&1; cat $O/r6_mfma_peak.txt
  (cd tools/micro && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_clock mfma_clock.hip) 2>/dev/null; /tmp/mfma_clock > $O/r6_mfma_clock.txt 2>&1; cat $O/r6_mfma_clock.txt ;;
pmct)
  TS="python $GRAFT_REPO_ROOT/tools/train_hostprof.py"
  bash tools/prof.sh r6_stats_train $TS > /dev/null
  bash tools/pmc.sh r6_pmc_fetch_train "FETCH_SIZE" $TS > /dev/null
  bash tools/pmc.sh r6_pmc_write_train "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" $TS > /dev/null
  bash tools/pmc.sh r6_pmc_sq_train "$SQ" $TS > /dev/null
  find $O -name "*kernel_trace.csv" -path "*r6_pmc*" -delete
  grep step $O/r6_stats_train/run.log ;;
esac; done

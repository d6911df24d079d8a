#!/bin/bash
# A/B of the fused step's search: all pairs vs the cell grid (one box)
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_trans.py -x -q 2>&1 | tail -5
for m in grid all_pairs grid all_pairs; do python tools/trans_perf.py 30 $m | tail -1; done
bash tools/prof.sh stats_ab_search python $GRAFT_REPO_ROOT/tools/trans_perf.py 30 all_pairs 2>&1 | grep -v "^$" | head -14

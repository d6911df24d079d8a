#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_trans.py -x -q 2>&1 | tail -3
bash tools/ab_search2.sh "" "-DTS_AB_NO_BOX"

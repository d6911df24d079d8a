#!/bin/bash
# usage: tools/pmc.sh <name> "<counters>" <cmd...>  — rocprofv3 PMC pass (own run, no kernel-trace mix with sys traces)
set -u
NAME=$1; shift
CTRS=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT -o p -- "$@" > $OUT/run.log 2>&1
grep -E "iter|Error|error" $OUT/run.log | head -8
ls $OUT

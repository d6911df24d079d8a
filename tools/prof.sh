#!/bin/bash
# usage: tools/prof.sh <name> <cmd...>   — rocprofv3 kernel-trace + stats into gpurun_out/<name>
set -u
NAME=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- "$@" > $OUT/run.log 2>&1
grep -E "iter|metric|Error|error" $OUT/run.log | head -20
find $OUT -name "*kernel_stats.csv" | head -3
F=$(find $OUT -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && head -25 "$F"

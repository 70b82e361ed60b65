#!/bin/bash
# Kernel stats of the cfg-4 training step (run on the GPU box: gpurun -- 'bash tools/profile_training_cfg4.sh r03_a [rows]').
TAG=${1:-r03_x}
ROWS=${2:-262144}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_train_cfg4_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o train --output-format csv -- python3 $R/tools/bench_graph_training.py cfg4 $ROWS > $OUT/train_stats.log 2>&1
tail -3 $OUT/train_stats.log
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $OUT/training_cfg4_kernel_stats.csv && head -30 $f | cut -c1-180
find $OUT -name "*kernel_trace.csv" -delete

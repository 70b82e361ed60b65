#!/bin/bash
# Kernel stats of the cfg-3 (spline) training step at 2^18 rows (run on the GPU box:
#   gpurun -- 'bash tools/profile_training_cfg3.sh r02_a').
TAG=${1:-r02_x}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_train3_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o train --output-format csv -- python3 $R/tools/bench_configs.py --train cfg3 --rows 262144 > $OUT/train_stats.log 2>&1
tail -3 $OUT/train_stats.log
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $OUT/training_cfg3_kernel_stats.csv && head -24 $f | cut -c1-220
find $OUT -name "*kernel_trace.csv" -delete

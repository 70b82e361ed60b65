#!/bin/bash
# Kernel stats of the cfg-2 training step (run on the GPU box: gpurun -- 'bash tools/profile_training.sh r02_a').
TAG=${1:-r02_x}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_train_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o train --output-format csv -- python3 $R/tools/bench_configs.py --train cfg2_f32 > $OUT/train_stats.log 2>&1
tail -3 $OUT/train_stats.log
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $OUT/training_cfg2_kernel_stats.csv && head -14 $f | cut -c1-200
find $OUT -name "*kernel_trace.csv" -delete

#!/bin/bash
# A/B timing of the cfg-4 backward program: shipped library vs build_variants/lib_noside.so (no factor stores)
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for v in base noside; do
  if [ $v = noside ]; then export STRIBOR_HIP_LIB=$R/build_variants/lib_noside.so; fi
  OUT=$R/gpurun_out/exp_cfg4_$v; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats -d $OUT -o t --output-format csv -- python3 $R/tools/bench_graph_training.py cfg4 262144 > $OUT/log.txt 2>&1
  f=$(find $OUT -name "*kernel_stats.csv" | head -1)
  echo "== $v"; grep -E "flow_fused_kernel<1, 8|wgrad_kernel" $f | cut -d, -f1-4 | cut -c1-120
  find $OUT -name "*kernel_trace.csv" -delete
done

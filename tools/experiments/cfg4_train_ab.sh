#!/bin/bash
# A/B of the cfg-4 TRAINING step on one box: variant libraries in build_variants/, interleaved rounds (2^20 rows and 2^18 rows)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for round in 1 2 3; do
  for f in build_variants/libstribor_hip_*.so; do
    [ -f "$f" ] || continue
    echo "== round $round: $f"
    STRIBOR_HIP_LIB=$R/$f python tools/bench_configs.py --train --train-only cfg4 2>&1 | grep "forward+backward" | cut -c1-40,75-140
    STRIBOR_HIP_LIB=$R/$f python tools/bench_configs.py --train --train-only cfg4 --rows 262144 2>&1 | grep "forward+backward" | cut -c1-40,75-140
  done
done

#!/bin/bash
# A/B of the cfg-4 kernel on ONE box: variant libraries in build_variants/ (STRIBOR_HIP_LIB), interleaved rounds, with the oracle check
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for round in 1 2 3; do
  for f in build_variants/libstribor_hip_*.so; do
    [ -f "$f" ] || continue
    case "$f" in *dbg*) continue;; esac
    echo "== round $round: $f"; STRIBOR_HIP_LIB=$R/$f python tools/bench_configs.py ${CFG:-cfg4} 2>&1 | grep rows_per_s | cut -c1-30,118-200
  done
done

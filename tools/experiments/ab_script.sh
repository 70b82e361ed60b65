#!/bin/bash
# Any benchmark script under every variant library of build_variants/ on ONE box, interleaved: ROUNDS=3 bash tools/experiments/ab_script.sh tools/bench_elementwise.py [args]
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for round in $(seq 1 ${ROUNDS:-3}); do
  for f in build_variants/libstribor_hip_*.so; do
    [ -f "$f" ] || continue
    n=$(basename $f .so); n=${n#libstribor_hip_}
    echo "== round $round: $n"
    STRIBOR_HIP_LIB=$R/$f python "$@" 2>&1 | grep -E "${GREP:-.}"
  done
done

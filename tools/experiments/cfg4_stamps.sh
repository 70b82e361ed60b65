#!/bin/bash
# in-kernel phase stamps of the cfg-4 kernel (tools/build_variant.sh dbg1 4_2 "-DSX_EXPERIMENTS -DSX_DEBUG_KNOBS -DSX_ONLY_MODE=7"; dbg5: + -DSX_PROF_THREAD=320): wave 1 (early half) and wave 5 (late half)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for f in build_variants/libstribor_hip_*dbg*.so; do
  [ -f "$f" ] || continue
  echo "== $f"; STRIBOR_HIP_LIB=$R/$f SX_PROF=1 python tools/bench_configs.py cfg4 2>&1 | grep -E "sx prof|rows_per_s" | tail -4 | cut -c1-900
done

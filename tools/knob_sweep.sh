#!/bin/bash
# Timing experiments on the fused kernel (cfg 2): SX_DBG bits switch parts of the kernel off in the
# -DSX_DEBUG_KNOBS build (libstribor_hip_dbg.so; results are wrong, only the time matters):
#   1 no weight re-staging   2 no per-step wait+barrier   4 no tanh   8 no exp   16 no MFMA
cd "$(dirname "$0")/.."
export STRIBOR_HIP_LIB=$PWD/stribor_amd/libstribor_hip_dbg.so
run() { echo "== $*"; env "$@" python tools/bench_configs.py cfg2_f32 2>&1 | grep rows_per_s | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   ', d['config'], '%.4g rows/s' % d['rows_per_s'], '%.4f ms' % d['ms_per_batch'])"; }
for B in 1 2; do
  for D in 0 1 2 3 4 8 12 16 28 31 19; do
    run SX_BLOCKS_PER_CU=$B SX_DBG=$D
  done
done

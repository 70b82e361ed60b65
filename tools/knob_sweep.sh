#!/bin/bash
# Timing experiments on the fused kernel (cfg 2).  Two kinds of builds (results are wrong, only the time matters):
#   * run-time knobs: a -DSX_EXPERIMENTS -DSX_DEBUG_KNOBS build (libstribor_hip_dbg.so; the experiment code lives in
#     csrc/sx_flow_experiments.h, which the product build never includes); SX_DBG bits switch parts of the kernel off
#       1 no weight re-staging   2 no per-step wait+barrier   16 no MFMA
#     and SX_PROF=1 prints in-kernel phase stamps, the shader clock and the spread of workgroup end times;
#   * compile-time ablations that keep the code straight-line: -DSX_EXPERIMENTS -DSX_X=<bits> builds (libstribor_hip_x<bits>.so)
#       4 no hidden transcendentals   8 no scale exp2   16 no MFMA   32 no weight ds_read   64 no fp16 split
# Build a variant:  tools/build_variant.sh x16 2_2 "-DSX_EXPERIMENTS -DSX_X=16"   (copies csrc to /tmp, rebuilds one object) and copy
# the .so next to the shipped one; STRIBOR_HIP_LIB selects it.
cd "$(dirname "$0")/.."
run() { echo "== $*"; env "$@" python tools/bench_configs.py cfg2_f32 2>&1 | grep rows_per_s | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   ', d['config'], '%.4g rows/s' % d['rows_per_s'], '%.4f ms' % d['ms_per_batch'])"; }
if [ -f stribor_amd/libstribor_hip_dbg.so ]; then
  for D in 0 1 2 3 16 19; do run STRIBOR_HIP_LIB=$PWD/stribor_amd/libstribor_hip_dbg.so SX_DBG=$D; done
  run STRIBOR_HIP_LIB=$PWD/stribor_amd/libstribor_hip_dbg.so SX_PROF=1
fi
for f in stribor_amd/libstribor_hip_x*.so; do [ -f "$f" ] && run STRIBOR_HIP_LIB=$PWD/$f; done
run SX_STATIC_CHUNKS=1
run SX_NO_PURE_MODE=1
run SX_BLOCKS_PER_CU=1

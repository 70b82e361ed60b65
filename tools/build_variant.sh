#!/bin/bash
# Variant build of the library for A/B timing on ONE box (boxes differ by 3 - 8 %: never compare across gpurun calls).
#   tools/build_variant.sh <name> <TX_HT pair, e.g. 4_2, or pair + MODE family, e.g. 4_2_f0> "<extra hipcc flags>"
# copies csrc + include to /tmp/v_<name>, rebuilds only sx_flow_x_<pair>_f*.o (or the one family object) and the host dispatcher with the extra flags (e.g.
# "-DSX_ONLY_MODE=7" for a seconds-long single-kernel build, "-DSX_EXPERIMENTS -DSX_DEBUG_KNOBS" for the in-kernel stamps,
# "-DSX_EXPERIMENTS -DSX_X=32" for an ablation) and leaves build_variants/libstribor_hip_<name>.so (git-ignored; it travels with gpurun).
# STRIBOR_HIP_LIB=$PWD/build_variants/libstribor_hip_<name>.so selects it; tools/experiments/cfg4_ab.sh runs all of them interleaved.
set -e
NAME=$1; PAIR=$2; EXTRA=$3
R=$(cd "$(dirname "$0")/.." && pwd)
V=/tmp/v_$NAME
rm -rf $V && mkdir -p $V/stribor_amd && cp -a $R/stribor_amd/csrc $V/stribor_amd/ && cp -a $R/include $V/
cd $V/stribor_amd/csrc
# every other object counts as up to date (a header edited since the last full build would otherwise rebuild all of them with $EXTRA)
touch *.o sx_build_id.inc 2>/dev/null
case $PAIR in *_f?) rm -f sx_flow_x_$PAIR.o ;; *) rm -f sx_flow_x_${PAIR}_f?.o ;; esac
rm -f sx_flow_fused.o ../libstribor_hip.so
make -j8 FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $EXTRA" 2>&1 | grep -v hipcc | tail -3
mkdir -p $R/build_variants && cp $V/stribor_amd/libstribor_hip.so $R/build_variants/libstribor_hip_$NAME.so

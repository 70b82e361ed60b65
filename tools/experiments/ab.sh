#!/bin/bash
# A/B of variant libraries (build_variants/libstribor_hip_*.so, tools/build_variant.sh) on ONE box, interleaved rounds:
#   CFG="cfg3" ROUNDS=3 bash tools/experiments/ab.sh [extra bench_configs.py args]
# prints ms per batch and the mean log_prob (equal between variants = same arithmetic) per variant and round
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for round in $(seq 1 ${ROUNDS:-3}); do
  for f in build_variants/libstribor_hip_*.so; do
    [ -f "$f" ] || continue
    case "$f" in *dbg*) continue;; esac
    n=$(basename $f .so); n=${n#libstribor_hip_}
    STRIBOR_HIP_LIB=$R/$f python tools/bench_configs.py ${CFG:-cfg3} "$@" 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if not l.startswith('{'):
        if 'Error' in l or 'error' in l: print('   ', l[:200])
        continue
    d = json.loads(l)
    print('round $round %-14s %-34s %8.4f ms  mean_lp %s' % ('$n', d['config'][:34], d['ms_per_batch'], d.get('mean_log_prob')))
"
  done
done

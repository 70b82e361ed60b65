#!/bin/bash
# This tree against the tree of the last commit on ONE box (boxes differ by 3 - 8 %), interleaved rounds:
#   build_variants/head_tree = `git archive HEAD` + its built libstribor_hip.so (the two trees' Python and C ABI may differ, so a
#   library swap -- tools/experiments/ab.sh -- is not enough).   CFG="cfg2 cfg3 cfg4" ROUNDS=3 bash tools/experiments/ab_head.sh [args]
R=${GRAFT_REPO_ROOT:-$PWD}
for round in $(seq 1 ${ROUNDS:-3}); do
  for t in head new; do
    d=$R; [ $t = head ] && d=$R/build_variants/head_tree
    (cd $d && python tools/bench_configs.py ${CFG:-cfg2 cfg3 cfg4} "$@" 2>&1) | python3 -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if not l.startswith('{'):
        if 'Error' in l: print('   ', l[:200])
        continue
    d = json.loads(l)
    print('round $round %-5s %-40s %9.4f ms  mean_lp %s' % ('$t', d['config'][:40], d['ms_per_batch'], d.get('mean_log_prob')))
"
  done
done

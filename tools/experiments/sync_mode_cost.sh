#!/bin/bash
# What the default error-reporting mode costs a training step: STRIBOR_SYNC_ERRORS=grad (one stream synchronisation at the end of every
# graph-building flow call, so a data-dependent error leaves the failing call) against =0 (deferred report), interleaved on ONE box.
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
for round in 1 2 3; do
  for mode in 0 grad; do
    echo "== round $round STRIBOR_SYNC_ERRORS=$mode"
    STRIBOR_SYNC_ERRORS=$mode python3 tools/bench_configs.py cfg2_f32 --train --train-only 2>/dev/null | grep forward
    STRIBOR_SYNC_ERRORS=$mode python3 tools/bench_configs.py cfg3 cfg4 --rows 262144 --train --train-only 2>/dev/null | grep forward
  done
done

#!/bin/bash
# The round-5 additions to a round's profiles/ refresh (gpurun -- 'bash tools/profile_round_extra.sh r05'): the slab forward tier
# (kernel stats + counters of its kernel at hidden 160), the training-step cliffs, and the --fat fuzz batches.  Outputs under gpurun_out/.
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_wide_$TAG -- python3 tools/experiments/wide_spline.py 160 > gpurun_out/${TAG}_wide_spline_prof.log 2>&1
cp $(find gpurun_out/prof_wide_$TAG -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_wide_spline_kernel_stats.csv
python3 tools/experiments/wide_spline.py 64 160 256 2>/dev/null > gpurun_out/${TAG}_wide_spline.jsonl
bash tools/pmc_kernel.sh rqs_slab_fwd slabf160 tools/experiments/wide_spline.py 160 > gpurun_out/${TAG}_pmck_slabf.log 2>&1
bash tools/pmc_kernel.sh rqs_slab_hidden slabh160 tools/experiments/wide_spline.py 160 > gpurun_out/${TAG}_pmck_slabh.log 2>&1
cd $R
python3 tools/bench_train_cliffs.py 2>/dev/null > gpurun_out/${TAG}_train_cliffs.jsonl
STRIBOR_SPLINE_UNFUSED=1 python3 tools/bench_train_cliffs.py 64 128 2>/dev/null >> gpurun_out/${TAG}_train_cliffs.jsonl
(timeout 400 python3 tools/fuzz_train.py 120 601 --fat 2>&1 | grep -v amdgpu | tail -4) > gpurun_out/${TAG}_fuzz_fat_train.log
(timeout 300 python3 tools/fuzz_train.py 120 602 --fat --infer 2>&1 | grep -v amdgpu | tail -4) > gpurun_out/${TAG}_fuzz_fat_infer.log
(timeout 300 python3 tools/fuzz_train.py 100 603 --fat --forward 2>&1 | grep -v amdgpu | tail -4) > gpurun_out/${TAG}_fuzz_fat_forward.log
(timeout 300 python3 tools/fuzz_train.py 100 604 --fat --k16 --infer 2>&1 | grep -v amdgpu | tail -4) > gpurun_out/${TAG}_fuzz_fat_k16_infer.log
(timeout 300 python3 tools/fuzz_train.py 100 605 --fat --k16 2>&1 | grep -v amdgpu | tail -4) > gpurun_out/${TAG}_fuzz_fat_k16_train.log
tail -n 2 gpurun_out/${TAG}_fuzz_fat_*.log

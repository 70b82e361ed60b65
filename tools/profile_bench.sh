#!/bin/bash
# Runs on the GPU box (gpurun -- 'bash tools/profile_bench.sh r01_g'): rocprofv3 kernel stats of bench.py and of every
# config, HBM-traffic PMC passes (separate --pmc runs, no tracing domains besides --kernel-trace) and SQ counter
# passes of the fused kernel.  Outputs land in gpurun_out/prof_<tag>/; tools/pmc_summary.py condenses them.
TAG=${1:-r01_x}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_stats.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/stats_all -o all --output-format csv -- python3 $R/tools/bench_configs.py cfg1 cfg2 cfg3 cfg4 > $OUT/all_stats.log 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
  "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_VALU SQ_INSTS_MFMA" \
  "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_VALU_TRANS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $OUT/pmc$i -o p --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/pmc$i.log 2>&1
done
python3 $R/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt

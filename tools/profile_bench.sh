#!/bin/bash
# Runs on the GPU box (gpurun -- 'bash tools/profile_bench.sh r02_a <commit>'): rocprofv3 kernel stats of bench.py (cfg 2
# headline + cfg 3 / cfg 4 lines + the stand-alone affine kernel), HBM-traffic PMC passes (separate --pmc runs, no tracing
# domains besides --kernel-trace) and SQ counter passes of the fused kernels.  Outputs land in gpurun_out/prof_<tag>/;
# tools/pmc_summary.py condenses them into pmc_cfg{2,3,4}.json / sq_cfg{2,3,4}.json.
TAG=${1:-r02_x}
COMMIT=${2:-unknown}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-training > $OUT/bench_stats.log 2>&1
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/bench_kernel_stats.csv
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
  "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_VALU SQ_INSTS_MFMA" \
  "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_VALU_TRANS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
  "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_VALU_MFMA_COEXEC_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $OUT/pmc$i -o p --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-training > $OUT/pmc$i.log 2>&1
done
BID=$(python3 -c "import sys; sys.path.insert(0, '$R'); from stribor_amd import _hip; print(_hip.build_id())")
python3 $R/tools/pmc_summary.py $OUT $COMMIT $BID > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -delete

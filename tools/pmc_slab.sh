# SQ counters of sx_rqs_slab_bwd's main kernel (GPU box): gpurun -- 'bash tools/pmc_slab.sh'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_slab; mkdir -p $O
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_TRANS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $O/p$i -o p --output-format csv -- python3 $R/tools/bench_slab.py "$@" > $O/log$i.txt 2>&1
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob('$O/p*/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'rqs_slab_bwd' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
a={k:sum(v)/len(v) for k,v in agg.items()}
for k,v in sorted(a.items()): print('%-28s %.4g'%(k,v))
cyc=a['GRBM_GUI_ACTIVE']/8; simd=cyc*1024
print('cycles',cyc,'valu busy',4*a['SQ_ACTIVE_INST_VALU']/simd,'mfma busy',a['SQ_VALU_MFMA_BUSY_CYCLES']/simd,'waves/simd',4*a['SQ_WAVE_CYCLES']/simd)
import json
chunks=131072.0
json.dump({'kernel':'rqs_slab_bwd_kernel<2,16,true,2> (2^18 rows, D=64, 32 transformed columns, H=64, K=16)','counters':a,'gpu_cycles':cyc,
 'valu_issue_busy_frac':4*a['SQ_ACTIVE_INST_VALU']/simd,'mfma_pipe_busy_frac':a['SQ_VALU_MFMA_BUSY_CYCLES']/simd,'waves_per_simd':4*a['SQ_WAVE_CYCLES']/simd,
 'valu_instructions_per_chunk_and_slab':a['SQ_INSTS_VALU']/chunks,'mfma_per_chunk_and_slab':a['SQ_INSTS_MFMA']/chunks,'wait_frac_of_wave_cycles':a['SQ_WAIT_ANY']/a['SQ_WAVE_CYCLES']},open('$O/sq_slab_bwd.json','w'),indent=1)
PY

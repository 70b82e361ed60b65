# SQ counters of the stand-alone spline kernels (GPU box): gpurun -- 'bash tools/pmc_spline_kernels.sh'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_spline; mkdir -p $O
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVES" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $O/p$i -o p --output-format csv -- python3 $R/tools/bench_spline_kernels.py "$@" > $O/log$i.txt 2>&1
done
python3 - <<PY
import csv,glob,collections,json,re
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$O/p*/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name']
        if 'rqs_kernel' in n or 'cubic_kernel' in n:
            m=re.search(r'(rqs_kernel|cubic_kernel)<([^>]*)>',n)
            agg[m.group(0) if m else n][r['Counter_Name']].append(float(r['Counter_Value']))
out={}
for k,c in sorted(agg.items()):
    a={n:sum(v)/len(v) for n,v in c.items()}
    cyc=a['GRBM_GUI_ACTIVE']/8; simd=cyc*1024
    waves=a.get('SQ_WAVES',0)
    out[k]={'gpu_cycles':cyc,'valu_issue_busy_frac':4*a['SQ_ACTIVE_INST_VALU']/simd,'waves_per_simd':4*a['SQ_WAVE_CYCLES']/simd,
            'wait_frac_of_wave_cycles':a['SQ_WAIT_ANY']/a['SQ_WAVE_CYCLES'],'counters':a}
    elems=(1<<18)*32
    print(k,'cycles %.0f valu busy %.2f waves/simd %.2f wait %.2f | per element: VALU %.0f LDS %.1f VMEM rd %.1f wr %.1f | bank-conflict frac %.3f'%(cyc,out[k]['valu_issue_busy_frac'],out[k]['waves_per_simd'],out[k]['wait_frac_of_wave_cycles'],a['SQ_INSTS_VALU']*64/elems,a['SQ_INSTS_LDS']*64/elems,a['SQ_INSTS_VMEM_RD']*64/elems,a['SQ_INSTS_VMEM_WR']*64/elems,a.get('SQ_LDS_BANK_CONFLICT',0)/(cyc*256)))
    out[k]['valu_instructions_per_element']=a['SQ_INSTS_VALU']*64/elems
json.dump(out,open('$O/sq_spline_kernels.json','w'),indent=1)
PY
cat $O/log1.txt | tail -6

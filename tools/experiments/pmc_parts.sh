#!/bin/bash
# VALU / MFMA instruction counts per launch of the cfg-4 kernel on dense-only, coupling-only and the mixed stack (tools/experiments/cfg4_parts.py)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/pmc_parts; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for case in dense16 coupling16 cfg4; do
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 -d $O/$case -o p --output-format csv -- python3 $R/tools/experiments/cfg4_parts.py $case > $O/$case.log 2>&1
  python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob('$O/$case/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'flow_fused' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
a={k:sum(v)/len(v) for k,v in agg.items()}
w=(1<<20)/32
print('$case', {k: round(v/w/16,1) for k,v in sorted(a.items())}, '(per wave and step, 16 steps)')
PY
done
rm -rf $O

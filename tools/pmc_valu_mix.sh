#!/bin/bash
# VALU instruction mix of bench.py's fused kernels (GPU box: gpurun -- 'bash tools/pmc_valu_mix.sh [tag]'): per-class wave-instruction
# counts (SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F32, _CVT, _INT32, _MFMA_MOPS_F16) beside SQ_INSTS_VALU / SQ_INSTS_MFMA / SQ_INSTS_SALU /
# SQ_INSTS_LDS, averaged per launch; the remainder of SQ_INSTS_VALU is moves, selects, compares, min / max, bit operations and
# accumulator-register moves.  Writes gpurun_out/valu_mix_<tag>/valu_mix.json.
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/valu_mix_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_ADD_F16 SQ_INSTS_VALU_FMA_F16 SQ_INSTS_VALU_MUL_F16 SQ_INSTS_VALU_INT64 SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $O/p$i -o p --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-training > $O/p$i.log 2>&1
done
BID=$(python3 -c "import sys; sys.path.insert(0, '$R'); from stribor_amd import _hip; print(_hip.build_id())")
python3 - <<PY
import csv, glob, collections, json, re
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$O/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r'flow_fused_kernel<\s*(\d+),\s*(\d+),\s*(\d+),\s*(\d+)\s*>', r['Kernel_Name'])
        if m:
            per['flow_fused_kernel<%s,%s,%s,%s>' % m.groups()][r['Counter_Name']].append(float(r['Counter_Value']))
out = {'build_id': '$BID', 'note': 'wave-instructions per launch (bench.py: 2^20 rows per launch); classes as the SQ counters define them', 'kernels': {}}
for k, cs in per.items():
    a = {c: sum(v) / len(v) for c, v in cs.items()}
    valu = a.get('SQ_INSTS_VALU', 0.0)
    named = sum(a.get(c, 0.0) for c in ('SQ_INSTS_VALU_ADD_F32', 'SQ_INSTS_VALU_MUL_F32', 'SQ_INSTS_VALU_FMA_F32', 'SQ_INSTS_VALU_TRANS_F32',
                                        'SQ_INSTS_VALU_CVT', 'SQ_INSTS_VALU_INT32', 'SQ_INSTS_VALU_ADD_F16', 'SQ_INSTS_VALU_FMA_F16',
                                        'SQ_INSTS_VALU_MUL_F16', 'SQ_INSTS_VALU_INT64'))
    a['other_valu (moves, selects, compares, min/max, bit ops, accvgpr)'] = valu - named - a.get('SQ_INSTS_MFMA', 0.0)
    out['kernels'][k] = a
json.dump(out, open('$O/valu_mix.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
PY
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -delete

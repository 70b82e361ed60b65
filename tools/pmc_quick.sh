#!/bin/bash
# Counters of ONE config's fused kernel in three rocprofv3 --pmc passes over tools/bench_configs.py (GPU box:
#   gpurun -- 'bash tools/pmc_quick.sh cfg3 tag'): wave-cycle accounting (SQ_WAVE_CYCLES / WAIT / ACTIVE, matrix-pipe busy) and the
# per-class VALU wave-instruction mix, averaged per launch; per wave-layer figures for the BASELINE shapes (2^20 rows, 8 layers,
# one wave = 32 rows).  Writes gpurun_out/pmcq_<cfg>_<tag>.json.
CFG=${1:-cfg3}; TAG=${2:-x}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/pmcq_${CFG}_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_INT64 SQ_WAVES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $O/p$i -o p --output-format csv -- python3 $R/tools/bench_configs.py $CFG > $O/log$i.txt 2>&1
done
BID=$(python3 -c "import sys; sys.path.insert(0, '$R'); from stribor_amd import _hip; print(_hip.build_id())")
python3 - <<PY
import csv, glob, collections, json, re
per = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob('$O/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r'flow_fused_kernel<\s*(\d+),\s*(\d+),\s*(\d+),\s*(\d+)\s*>', r['Kernel_Name'])
        if m:
            per['flow_fused_kernel<%s,%s,%s,%s>' % m.groups()][r['Counter_Name']].append(float(r['Counter_Value']))
for f in glob.glob('$O/p*/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r'flow_fused_kernel<\s*(\d+),\s*(\d+),\s*(\d+),\s*(\d+)\s*>', r['Kernel_Name'])
        if m:
            dur['flow_fused_kernel<%s,%s,%s,%s>' % m.groups()].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
out = {'build_id': '$BID', 'config': '$CFG', 'kernels': {}}
for k, cs in per.items():
    a = {c: sum(v) / len(v) for c, v in cs.items()}
    cyc = a.get('GRBM_GUI_ACTIVE', 0.0) / 8
    simd = cyc * 1024
    d = {'counters_avg_per_launch': a, 'shader_cycles_per_launch': cyc}
    if dur[k]:
        d['kernel_ns_profiled'] = sum(dur[k]) / len(dur[k])
        d['effective_clock_ghz'] = cyc / d['kernel_ns_profiled'] if d['kernel_ns_profiled'] else None
    if simd:
        d['valu_issue_busy_frac'] = 4 * a.get('SQ_ACTIVE_INST_VALU', 0) / simd
        d['mfma_pipe_busy_frac'] = a.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / simd
        d['valu_mfma_coexec_frac'] = a.get('SQ_VALU_MFMA_COEXEC_CYCLES', 0) / simd
        d['waves_per_simd_avg'] = 4 * a.get('SQ_WAVE_CYCLES', 0) / simd
        wc = a.get('SQ_WAVE_CYCLES', 0) or 1
        d['sq_wait_any_frac_of_wave_cycles'] = a.get('SQ_WAIT_ANY', 0) / wc
        d['sq_wait_inst_any_frac_of_wave_cycles'] = a.get('SQ_WAIT_INST_ANY', 0) / wc
        d['sq_active_inst_any_frac_of_wave_cycles'] = a.get('SQ_ACTIVE_INST_ANY', 0) / wc
    wl = (1 << 20) / 32 * 8
    valu = a.get('SQ_INSTS_VALU', 0.0)
    named = sum(a.get(c, 0.0) for c in ('SQ_INSTS_VALU_ADD_F32', 'SQ_INSTS_VALU_MUL_F32', 'SQ_INSTS_VALU_FMA_F32', 'SQ_INSTS_VALU_TRANS_F32',
                                        'SQ_INSTS_VALU_CVT', 'SQ_INSTS_VALU_INT32', 'SQ_INSTS_VALU_INT64'))
    d['valu_mix_per_wave_layer (2^20 rows, 8 layers)'] = {
        'add_f32': a.get('SQ_INSTS_VALU_ADD_F32', 0) / wl, 'mul_f32': a.get('SQ_INSTS_VALU_MUL_F32', 0) / wl,
        'fma_f32': a.get('SQ_INSTS_VALU_FMA_F32', 0) / wl, 'trans_f32': a.get('SQ_INSTS_VALU_TRANS_F32', 0) / wl,
        'cvt': a.get('SQ_INSTS_VALU_CVT', 0) / wl, 'int32': a.get('SQ_INSTS_VALU_INT32', 0) / wl, 'int64': a.get('SQ_INSTS_VALU_INT64', 0) / wl,
        'mfma': a.get('SQ_INSTS_MFMA', 0) / wl, 'other': (valu - named - a.get('SQ_INSTS_MFMA', 0)) / wl, 'valu_total': valu / wl,
        'salu': a.get('SQ_INSTS_SALU', 0) / wl, 'lds': a.get('SQ_INSTS_LDS', 0) / wl}
    out['kernels'][k] = d
json.dump(out, open('$R/gpurun_out/pmcq_${CFG}_$TAG.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $O

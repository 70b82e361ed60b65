#!/bin/bash
# Counters of ONE kernel (name substring) in separate rocprofv3 --pmc passes over a python script:
#   gpurun -- 'bash tools/pmc_kernel.sh rqs_slab_fwd tag tools/experiments/wide_spline.py 160'
# wave-cycle accounting, matrix / vector pipe busy, L2 hits / misses, HBM bytes, LDS; averaged per launch.
# Writes gpurun_out/pmck_<tag>.json.
PAT=$1; TAG=$2; shift 2
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/pmck_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_VALU_TRANS_F32" \
           "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  ( cd $R && rocprofv3 --kernel-trace --pmc $set -d $O/p$i -o p --output-format csv -- python3 "$@" > $O/log$i.txt 2>&1 )
done
python3 - <<PY
import csv, glob, collections, json
per = collections.defaultdict(list); dur = []
for f in glob.glob('$O/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if '$PAT' in r['Kernel_Name']:
            per[r['Counter_Name']].append(float(r['Counter_Value']))
for f in glob.glob('$O/p1/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if '$PAT' in r['Kernel_Name']:
            dur.append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
a = {c: sum(v) / len(v) for c, v in per.items()}
out = {'kernel': '$PAT', 'launches': len(dur), 'counters_avg_per_launch': a}
cyc = a.get('GRBM_GUI_ACTIVE', 0.0) / 8
simd = cyc * 1024
if dur:
    out['kernel_ns_profiled'] = sum(dur) / len(dur)
if simd:
    out['valu_issue_busy_frac'] = 4 * a.get('SQ_ACTIVE_INST_VALU', 0) / simd
    out['mfma_pipe_busy_frac'] = a.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / simd
    out['valu_mfma_coexec_frac'] = a.get('SQ_VALU_MFMA_COEXEC_CYCLES', 0) / simd
    out['waves_per_simd_avg'] = 4 * a.get('SQ_WAVE_CYCLES', 0) / simd
    wc = a.get('SQ_WAVE_CYCLES', 0) or 1
    out['sq_wait_any_frac_of_wave_cycles'] = a.get('SQ_WAIT_ANY', 0) / wc
    out['sq_wait_inst_any_frac_of_wave_cycles'] = a.get('SQ_WAIT_INST_ANY', 0) / wc
    out['lds_active_frac_of_wave_cycles'] = a.get('SQ_ACTIVE_INST_LDS', 0) / wc
    out['lds_wait_frac_of_wave_cycles'] = a.get('SQ_WAIT_INST_LDS', 0) / wc
    out['effective_clock_ghz'] = cyc / out['kernel_ns_profiled'] if dur else None
if a.get('TCC_REQ_sum'):
    out['l2_hit_frac'] = a.get('TCC_HIT_sum', 0) / (a.get('TCC_HIT_sum', 0) + a.get('TCC_MISS_sum', 0) or 1)
out['hbm_read_MB (FETCH_SIZE KiB x 2: gfx950 streaming-read correction)'] = a.get('FETCH_SIZE', 0) * 2 * 1024 / 1e6
out['hbm_written_MB'] = a.get('WRITE_SIZE', 0) * 1024 / 1e6
json.dump(out, open('$R/gpurun_out/pmck_$TAG.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $O

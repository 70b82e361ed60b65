"""Condense the rocprofv3 outputs of tools/profile_bench.sh: per-kernel averages of every counter, HBM bytes per
launch (FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE is doubled for streaming reads per MI355X_MICROARCH.md's gfx950
note) and the SQ utilisation figures of the fused kernel.  Writes <dir>/pmc_cfg2.json and <dir>/sq_cfg2.json."""
import collections
import csv
import glob
import json
import os
import sys


def main(d):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, 'pmc*', '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            k = 'flow_fused_kernel' if 'flow_fused' in r['Kernel_Name'] else (
                'affine_coupling_vec_kernel' if 'affine_coupling_vec' in r['Kernel_Name'] else None)
            if k:
                per[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for f in glob.glob(os.path.join(d, 'pmc3', '**', '*kernel_trace.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if 'flow_fused' in r['Kernel_Name']:
                dur['flow_fused_kernel'].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
    avg = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in per.items()}
    note = ('rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `python3 bench.py --steps 5 --warmup 2 '
            '--no-cpu-baseline`; KiB units; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of '
            'a coalesced streaming read); per launch of 2^20 rows')
    traffic = {}
    for k, a in avg.items():
        if 'FETCH_SIZE' in a and 'WRITE_SIZE' in a:
            traffic[k] = {'FETCH_SIZE_KiB_raw': a['FETCH_SIZE'], 'WRITE_SIZE_KiB_raw': a['WRITE_SIZE'],
                          'hbm_bytes_per_launch': int(2 * a['FETCH_SIZE'] * 1024 + a['WRITE_SIZE'] * 1024), 'note': note}
    json.dump(traffic, open(os.path.join(d, 'pmc_cfg2.json'), 'w'), indent=1)
    a = avg.get('flow_fused_kernel', {})
    sq = {'counters_avg_per_launch': a}
    if 'GRBM_GUI_ACTIVE' in a:
        cyc = a['GRBM_GUI_ACTIVE'] / 8.0            # summed over the 8 XCDs
        simd_cycles = cyc * 1024                    # 256 CUs x 4 SIMDs
        sq['shader_cycles_per_launch'] = cyc
        if dur['flow_fused_kernel']:
            ns = sum(dur['flow_fused_kernel']) / len(dur['flow_fused_kernel'])
            sq['kernel_ns_profiled'] = ns
            sq['effective_clock_ghz'] = cyc / ns
        # SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles; SQ_VALU_MFMA_BUSY_CYCLES counts cycles
        if 'SQ_ACTIVE_INST_VALU' in a:
            sq['valu_issue_busy_frac'] = 4 * a['SQ_ACTIVE_INST_VALU'] / simd_cycles
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in a:
            sq['mfma_pipe_busy_frac'] = a['SQ_VALU_MFMA_BUSY_CYCLES'] / simd_cycles
        if 'SQ_WAVE_CYCLES' in a:
            sq['waves_per_simd_avg'] = 4 * a['SQ_WAVE_CYCLES'] / simd_cycles
        for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_SCA'):
            if c in a and 'SQ_WAVE_CYCLES' in a:
                sq[c.lower() + '_frac_of_wave_cycles'] = a[c] / a['SQ_WAVE_CYCLES']
    json.dump(sq, open(os.path.join(d, 'sq_cfg2.json'), 'w'), indent=1)
    print(json.dumps({'traffic': {k: v['hbm_bytes_per_launch'] for k, v in traffic.items()},
                      'sq': {k: v for k, v in sq.items() if k != 'counters_avg_per_launch'}}, indent=1))
    for c, v in sorted(a.items()):
        print('%-32s %.5g' % (c, v))


if __name__ == '__main__':
    main(sys.argv[1])

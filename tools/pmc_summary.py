"""Condense the rocprofv3 outputs of tools/profile_bench.sh: per-kernel averages of every counter, HBM bytes per
launch (FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE is doubled for streaming reads per MI355X_MICROARCH.md's gfx950
note) and the SQ utilisation figures of the fused kernels.  bench.py launches three fused-kernel variants (cfg 2:
<1,2,2,5>, cfg 3: <1,2,2,3>, cfg 4: <1,4,2,7>) and the stand-alone affine kernel; every one gets its own record.
Writes <dir>/pmc_cfg{2,3,4}.json and <dir>/sq_cfg{2,3,4}.json; `python tools/pmc_summary.py <dir> <commit> <build_id>` (build_id =
sx_build_id() of the library profiled: bench.py quotes a profile's figures only for the same build)."""
import collections
import csv
import glob
import json
import os
import re
import sys

VARIANTS = {'cfg2': (1, 2, 2, 5), 'cfg3': (1, 2, 2, 3), 'cfg4': (1, 4, 2, 7)}


def kernel_key(name):
    if 'affine_coupling_vec' in name:
        return 'affine_coupling_vec_kernel'
    # (round 6: every no-graph call is followed by the exact redo pass of sx_flow_run2 -- the SAME template instance in the sx_f32x
    #  namespace, on ordinary data an empty launch -- and bench.py's cfg2_exact entry runs that namespace's kernels too: the counters of
    #  a config are those of its fp16 x 3 kernel only)
    if 'sx_f32x' in name:
        return None
    m = re.search(r'flow_fused_kernel<\s*(\d+),\s*(\d+),\s*(\d+),\s*(\d+)\s*>', name)
    if not m:
        m = re.search(r'flow_fused_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E', name)
    if m:
        t = tuple(int(v) for v in m.groups())
        for cfg, want in VARIANTS.items():
            if t == want:
                return cfg
    return None


def main(d, commit, build_id=None):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, 'pmc*', '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            k = kernel_key(r['Kernel_Name'])
            if k:
                per[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for f in glob.glob(os.path.join(d, 'pmc3', '**', '*kernel_trace.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            k = kernel_key(r['Kernel_Name'])
            if k:
                dur[k].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
    avg = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in per.items()}
    note = ('rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `python3 bench.py --steps 5 --warmup 2 '
            '--no-cpu-baseline`; KiB units; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of '
            'a coalesced streaming read); per launch of 2^20 rows')

    def traffic(a):
        if 'FETCH_SIZE' not in a or 'WRITE_SIZE' not in a:
            return None
        return {'FETCH_SIZE_KiB_raw': a['FETCH_SIZE'], 'WRITE_SIZE_KiB_raw': a['WRITE_SIZE'],
                'hbm_bytes_per_launch': int(2 * a['FETCH_SIZE'] * 1024 + a['WRITE_SIZE'] * 1024), 'note': note}

    out = {}
    for cfg in VARIANTS:
        a = avg.get(cfg, {})
        t = traffic(a)
        if cfg == 'cfg2':
            rec = {'commit': commit, 'build_id': build_id}
            if t:
                rec['flow_fused_kernel'] = t
            ta = traffic(avg.get('affine_coupling_vec_kernel', {}))
            if ta:
                rec['affine_coupling_vec_kernel'] = ta
        else:
            rec = dict(t or {}, commit=commit, build_id=build_id, kernel='flow_fused_kernel<%d,%d,%d,%d>' % VARIANTS[cfg])
        json.dump(rec, open(os.path.join(d, 'pmc_%s.json' % cfg), 'w'), indent=1)
        sq = {'commit': commit, 'build_id': build_id, 'kernel': 'flow_fused_kernel<%d,%d,%d,%d>' % VARIANTS[cfg], 'counters_avg_per_launch': a}
        if 'GRBM_GUI_ACTIVE' in a:
            cyc = a['GRBM_GUI_ACTIVE'] / 8.0            # summed over the 8 XCDs
            simd_cycles = cyc * 1024                    # 256 CUs x 4 SIMDs
            sq['shader_cycles_per_launch'] = cyc
            if dur[cfg]:
                ns = sum(dur[cfg]) / len(dur[cfg])
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
        if 'SQ_INSTS_VALU_FMA_F32' in a and 'SQ_INSTS_VALU' in a:
            # VALU instruction mix per wave and layer (a wave = 32 rows; bench.py launches 2^20 rows; cfg 2 / 3: 8 layers, cfg 4: 16):
            # the classes are the SQ counters'; `other` = SQ_INSTS_VALU minus the named classes and the MFMAs: moves, selects,
            # compares, min / max, bit operations.  cfg 3: a lane evaluates 16 spline elements per layer.
            wl = (1 << 20) / 32 * {'cfg2': 8, 'cfg3': 8, 'cfg4': 16}[cfg]
            named = ['SQ_INSTS_VALU_ADD_F32', 'SQ_INSTS_VALU_MUL_F32', 'SQ_INSTS_VALU_FMA_F32', 'SQ_INSTS_VALU_TRANS_F32', 'SQ_INSTS_VALU_CVT',
                     'SQ_INSTS_VALU_INT32', 'SQ_INSTS_VALU_INT64']
            mix = {c[len('SQ_INSTS_VALU_'):].lower(): a.get(c, 0.0) / wl for c in named}
            mix['mfma'] = a.get('SQ_INSTS_MFMA', 0.0) / wl
            mix['other'] = (a['SQ_INSTS_VALU'] - sum(a.get(c, 0.0) for c in named) - a.get('SQ_INSTS_MFMA', 0.0)) / wl
            mix['valu_total'] = a['SQ_INSTS_VALU'] / wl
            mix['salu'] = a.get('SQ_INSTS_SALU', 0.0) / wl
            mix['lds'] = a.get('SQ_INSTS_LDS', 0.0) / wl
            sq['valu_mix_per_wave_layer'] = {k_: round(v, 1) for k_, v in mix.items()}
        json.dump(sq, open(os.path.join(d, 'sq_%s.json' % cfg), 'w'), indent=1)
        out[cfg] = {'traffic': (t or {}).get('hbm_bytes_per_launch'),
                    'sq': {k: v for k, v in sq.items() if k not in ('counters_avg_per_launch',)}}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else 'unknown', sys.argv[3] if len(sys.argv) > 3 else None)

#!/bin/bash
# HBM traffic of the cfg-3 (spline) training step's kernels at 2^18 rows (GPU box: gpurun -- 'bash tools/pmc_training_cfg3.sh [unfused]'):
# FETCH_SIZE and WRITE_SIZE in separate --pmc passes (KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 note), summed
# over ALL kernels of one step and listed for the large ones.  "unfused" measures the per-row parameter path it replaced.
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=fused
if [ "$1" = "unfused" ]; then export STRIBOR_SPLINE_UNFUSED=1; TAG=unfused; fi
O=$R/gpurun_out/pmc_train3_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $O/$c -o p --output-format csv -- python3 $R/tools/bench_configs.py --train cfg3 --rows 262144 --train-only > $O/$c.log 2>&1
done
BID=$(python3 -c "import sys; sys.path.insert(0, '$R'); from stribor_amd import _hip; print(_hip.build_id())")
python3 - <<PY
import csv, glob, collections, json
per = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob('$O/*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        k = ('rqs_slab_bwd_kernel' if 'rqs_slab_bwd' in n else 'rqs_slab_dh_reduce_kernel' if 'dh_reduce' in n else
             'exact redo pass (sx_f32x; empty list)' if 'sx_f32x' in n else
             'flow_fused_kernel (per-layer forward)' if 'flow_fused' in n else 'rqs_inverse_bwd_kernel' if 'rqs_inverse_bwd' in n else
             'rqs_kernel (forward)' if 'rqs_kernel' in n else 'library GEMM (Cijk)' if n.startswith('Cijk') else 'other')
        per[k][r['Counter_Name']] += float(r['Counter_Value'])
        calls[k][r['Counter_Name']] += 1
steps = 11.0          # bench_configs.py --train: 1 warm-up + 5 x 2 timed steps
out = {'rows': 262144, 'steps_profiled': steps, 'path': '$TAG', 'build_id': '$BID', 'kernels': {}}
tot_r = tot_w = 0.0
for k, a in per.items():
    rd = 2 * a.get('FETCH_SIZE', 0) * 1024 / 1e6 / steps
    wr = a.get('WRITE_SIZE', 0) * 1024 / 1e6 / steps
    tot_r += rd; tot_w += wr
    out['kernels'][k] = {'launches_per_step': round(calls[k].get('FETCH_SIZE', 0) / steps, 1), 'hbm_read_MB_per_step': round(rd, 1),
                         'hbm_write_MB_per_step': round(wr, 1)}
out['hbm_read_MB_per_step'] = round(tot_r, 1)
out['hbm_write_MB_per_step'] = round(tot_w, 1)
json.dump(out, open('$O/pmc_training_cfg3_$TAG.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
PY
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -delete

#!/bin/bash
# HBM traffic of the cfg-4 (dense-linear + affine couplings) training step's kernels at 2^18 rows (GPU box: gpurun -- 'bash tools/pmc_training_cfg4.sh'):
# FETCH_SIZE and WRITE_SIZE in separate --pmc passes (KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 note), summed
# over ALL kernels of one step and listed for the large ones.
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=fused
O=$R/gpurun_out/pmc_train4_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $O/$c -o p --output-format csv -- python3 $R/tools/bench_configs.py --train cfg4 --rows 262144 --train-only > $O/$c.log 2>&1
done
BID=$(python3 -c "import sys; sys.path.insert(0, '$R'); from stribor_amd import _hip; print(_hip.build_id())")
python3 - <<PY
import csv, glob, collections, json
per = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob('$O/*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        k = ('flow_fused_kernel<1,8,2,4> (backward program)' if 'flow_fused_kernel<1, 8' in n else
             'exact redo pass (sx_f32x; empty list)' if 'sx_f32x' in n else
             'flow_fused_kernel (forward)' if 'flow_fused' in n else 'wgrad_kernel<4,4>' if 'wgrad_kernel<4, 4' in n else
             'wgrad_kernel<4,2>' if 'wgrad_kernel<4, 2' in n else 'wgrad_kernel<2,2>' if 'wgrad_kernel<2, 2' in n else
             'wgrad_reduce_kernel' if 'wgrad_reduce' in n else 'tri_inverse_kernel' if 'tri_inverse' in n else
             'library GEMM (Cijk, D x D)' if n.startswith('Cijk') else 'other')
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
json.dump(out, open('$O/pmc_training_cfg4_$TAG.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
PY
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -delete

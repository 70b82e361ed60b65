#!/bin/bash
# HBM traffic of the cfg-2 training step's kernels (run on the GPU box: gpurun -- 'bash tools/pmc_training.sh'):
# FETCH_SIZE and WRITE_SIZE in separate --pmc passes (KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 note).
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/pmc_train; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $O/$c -o p --output-format csv -- python3 $R/tools/bench_configs.py --train cfg2_f32 > $O/$c.log 2>&1
done
python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$O/*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        k = ('layer-major backward flow_fused_kernel<1,4,2,11>' if 'Li4ELi2ELi11E' in n or '<1, 4, 2, 11>' in n else
             'backward program flow_fused_kernel<1,4,2,4>' if 'Li4ELi2ELi4E' in n or '<1, 4, 2, 4>' in n else
             'exact redo pass (sx_f32x; empty list)' if 'sx_f32x' in n else
             'forward flow_fused_kernel<1,2,2,5>' if 'flow_fused' in n else
             'wgrad_layer_kernel<1,2,1,8>' if 'wgrad_layer' in n else
             'wgrad_reduce_kernel' if 'wgrad_reduce' in n else None)
        if k: agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for k, cs in agg.items():
    a = {c: sum(v) / len(v) for c, v in cs.items()}
    out[k] = {'FETCH_SIZE_KiB_raw': a.get('FETCH_SIZE'), 'WRITE_SIZE_KiB_raw': a.get('WRITE_SIZE'),
              'hbm_read_MB_per_launch': round(2 * a.get('FETCH_SIZE', 0) * 1024 / 1e6, 1),
              'hbm_write_MB_per_launch': round(a.get('WRITE_SIZE', 0) * 1024 / 1e6, 1)}
json.dump(out, open('$O/pmc_training_cfg2.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
PY
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -delete

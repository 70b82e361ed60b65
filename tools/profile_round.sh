#!/bin/bash
# Everything a round's profiles/ refresh needs, in one GPU-box call (gpurun -- 'bash tools/profile_round.sh r05 <commit>'):
# bench kernel stats + HBM / SQ counters (profile_bench.sh), the three training steps' kernel stats and HBM traffic, the stand-alone
# element-wise kernels, the shape cliffs, and one un-profiled bench.py line.  Outputs under gpurun_out/; copy what is judged to profiles/.
TAG=${1:-r05}; COMMIT=${2:-unknown}
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
bash tools/profile_bench.sh $TAG $COMMIT > gpurun_out/profile_bench_$TAG.log 2>&1
bash tools/profile_training.sh $TAG > gpurun_out/profile_training_$TAG.log 2>&1
bash tools/profile_training_cfg3.sh $TAG > gpurun_out/profile_training_cfg3_$TAG.log 2>&1
bash tools/profile_training_cfg4.sh $TAG > gpurun_out/profile_training_cfg4_$TAG.log 2>&1
bash tools/pmc_training.sh > gpurun_out/pmc_training_$TAG.log 2>&1
bash tools/pmc_training_cfg3.sh > gpurun_out/pmc_training_cfg3_$TAG.log 2>&1
bash tools/pmc_training_cfg4.sh > gpurun_out/pmc_training_cfg4_$TAG.log 2>&1
cd $R
python3 tools/bench_elementwise.py > gpurun_out/${TAG}_elementwise.jsonl 2> gpurun_out/${TAG}_elementwise.err
python3 tools/bench_cliffs.py > gpurun_out/${TAG}_cliffs.jsonl 2> gpurun_out/${TAG}_cliffs.err
python3 bench.py --steps 200 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python3 -c "import sys; sys.path.insert(0, '.'); from stribor_amd import _hip; print(_hip.build_id())" > gpurun_out/${TAG}_build_id.txt
ls gpurun_out | head -50

#!/bin/bash
# The fuzz campaign of a round's FINAL build in one GPU-box call (gpurun -- 'bash tools/fuzz_round.sh r06 900'): every batch's
# command line and its last lines (the last case and `worst`) -> gpurun_out/<TAG>_fuzz_final.txt; then the memory-safety pass (NaN-filled
# torch.empty / library scratch).  SEED0 + k are the seeds.
TAG=${1:-r06}; S=${2:-900}
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
OUT=gpurun_out/${TAG}_fuzz_final.txt
: > $OUT
run() { local t=$1; shift; echo "== $*" >> $OUT; (timeout $t python3 tools/"$@" 2>&1 | grep -v amdgpu | tail -2 | cut -c1-260) >> $OUT; }
run 500 fuzz_train.py 150 $((S+1)) --mix
run 400 fuzz_train.py 150 $((S+2)) --mix --infer
run 300 fuzz_train.py 100 $((S+3)) --wide --infer
run 300 fuzz_train.py 80 $((S+4)) --xwide --infer
run 300 fuzz_train.py 40 $((S+5)) --long --infer
run 300 fuzz_train.py 100 $((S+6)) --infer --bf16
run 300 fuzz_train.py 100 $((S+7)) --forward
run 300 fuzz_train.py 80 $((S+8)) --mix --sets
run 400 fuzz_train.py 100 $((S+9)) --fat
run 300 fuzz_train.py 100 $((S+10)) --fat --infer
run 300 fuzz_train.py 80 $((S+11)) --fat --k16 --infer
run 300 fuzz_train.py 60 $((S+12)) --time
run 300 fuzz_dense.py 120 $((S+13))
run 300 fuzz_dense.py 120 $((S+14)) --big
run 300 fuzz_dense.py 120 $((S+18)) --mid
run 300 fuzz_dense.py 60 $((S+15)) --bf16
run 300 fuzz_dense.py 60 $((S+16)) --exact
run 300 fuzz_dense.py 40 $((S+17)) --big --bf16
echo "== STRIBOR_TEST_POISON=1 STRIBOR_POISON_SCRATCH=1 python -m pytest tests -m gpu -k 'not graph and not stay_inside'" >> $OUT
(STRIBOR_TEST_POISON=1 STRIBOR_POISON_SCRATCH=1 timeout 900 python3 -m pytest tests -m gpu -q -k 'not graph and not stay_inside' 2>&1 | tail -3) >> $OUT
cat $OUT

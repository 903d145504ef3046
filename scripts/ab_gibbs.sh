#!/bin/bash
# A/B of the Gibbs kernel: build/libquilt_amd_base.so (the commit before) against the tree's library.
#   gpurun --timeout 1200 -- 'bash scripts/ab_gibbs.sh [base]'
export QA_DEV=1   # (quilt_amd.native loads only the tree's library otherwise)
OUT=$PWD/gpurun_out/ab; mkdir -p $OUT
python -m pytest tests/test_gibbs_gpu.py tests/test_rtwin_gpu.py tests/test_headline_gpu.py -x -q -m gpu -k "gibbs or rtwin or shard" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for LIB in $1 new; do
  if [ $LIB = base ]; then export QUILT_AMD_LIB=$PWD/build/libquilt_amd_base.so; else unset QUILT_AMD_LIB; fi
  for CFG in "2048 20000" "2048 5000" "1024 20000"; do
    set -- $CFG
    echo "== $LIB chains $1 reads $2"
    python scripts/perf_gibbs.py --chains $1 --reads $2 --init-iter --reps 2 2>&1 | grep "rep 1"
  done
done 2>&1 | tee $OUT/ab.log

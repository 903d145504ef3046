#!/bin/bash
# Developer aid: where a 2 048-chain launch of the sampler's 256-register build spends its time -- the tree's library against
# builds with one part switched off (build/libquilt_amd_dbg_<V>.so; QA_DEV=1 lets quilt_amd.native load them: -DQA_DBG_SKIP_BWD, -DQA_DBG_SKIP_SHARD,
# -DQA_DBG_NO_LATE_LOADS; EARLY_E = -DQA_LEAN_EARLY_E, a candidate, whose results must be right: the Gibbs tests run on it first; results of those builds are wrong, only their times mean something), at 20 000 and 5 000 reads.
#   gpurun --timeout 1500 -- 'bash scripts/decompose_gibbs.sh'
export QA_DEV=1
QUILT_AMD_LIB=$PWD/build/libquilt_amd_dbg_EARLY_E.so python -m pytest tests/test_gibbs_gpu.py tests/test_rtwin_gpu.py tests/test_headline_gpu.py -x -q -m gpu -k 'gibbs or rtwin or shard' 2>&1 | tail -2
for V in full SKIP_BWD SKIP_SHARD NO_LATE_LOADS EARLY_E; do
  if [ $V = full ]; then unset QUILT_AMD_LIB; else export QUILT_AMD_LIB=$PWD/build/libquilt_amd_dbg_$V.so; fi
  for R in 20000 5000; do
    echo -n "$V reads $R: "
    python scripts/perf_gibbs.py --chains 2048 --reads $R --init-iter --reps 2 2>&1 | grep "rep 1" | sed 's/.*gibbs \([0-9.]*\) ms.*/\1 ms/'
  done
done

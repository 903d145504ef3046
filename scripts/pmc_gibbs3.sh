#!/bin/bash
# Developer aid: SQ instruction counters of the three-label sampler alone (scripts/perf_gibbs.py --nipt), one chain per SIMD.
#   gpurun --timeout 900 -- 'bash scripts/pmc_gibbs3.sh 896 20000'
OUT=$PWD/gpurun_out/r05; mkdir -p $OUT; export TMPDIR=/tmp
CH=${1:-896}; RD=${2:-20000}
CMD="python $GRAFT_REPO_ROOT/scripts/perf_gibbs.py --nipt --chains $CH --reads $RD --samples 16 --reps 1"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_g3 -o pmc -- $CMD > $OUT/pmc_g3.out 2> $OUT/pmc_g3.err)
find $OUT/pmc_g3 -name '*counter_collection.csv' -exec cp {} $OUT/pmc_g3_counters.csv \;
rm -rf $OUT/pmc_g3
python scripts/pmc_insts.py $OUT/pmc_g3_counters.csv > $OUT/pmc_g3_${CH}_${RD}.json
python - <<PY
import json
d=json.load(open("$OUT/pmc_g3_${CH}_${RD}.json"))["per_launch"]
for k,v in d.items():
    if "gibbs3" in k or "block" in k: print(k, {a:(round(b/1e6,1) if isinstance(b,float) else b) for a,b in v.items()})
PY
rm -f $OUT/pmc_g3_counters.csv

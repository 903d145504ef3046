#!/bin/bash
# PMC evidence for the Gibbs kernel build that dominates the headline (2 048-chain launches of k_gibbs<10, 1, true>):
#   gpurun --timeout 1500 -- 'bash scripts/pmc_round.sh r04'
# Launch sets of two steps (the default --fuse 2) with --steps 6 --warmup 2 (three launch sets: one whole set per host thread), so every main-round Gibbs launch is the lean build.
# One counter group per pass, --kernel-trace only (no --stats, no other trace domain).  Summaries per template instantiation.
TAG=${1:-r05}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --r2-vs-cpu 0 --no-alone --precision fp64 --dotcall 0"
pass() {   # name, counters...
    local N=$1; shift
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$N -o pmc -- $CMD > $OUT/pmc_$N.json 2> $OUT/pmc_$N.err)
    find $OUT/pmc_$N -name '*counter_collection.csv' -exec cp {} $OUT/pmc_${N}_counters.csv \;
    rm -rf $OUT/pmc_$N
    ls -la $OUT/pmc_${N}_counters.csv 2>/dev/null || tail -5 $OUT/pmc_$N.err
}
pass FETCH FETCH_SIZE
pass WRITE WRITE_SIZE
pass INSTS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAVES
pass ACTIVE2 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_WAVE_CYCLES
python scripts/pmc_summary.py $OUT/pmc_FETCH_counters.csv $OUT/pmc_WRITE_counters.csv $OUT/pmc_traffic.json \
    "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --r2-vs-cpu 0 --no-alone --precision fp64 --dotcall 0"
python scripts/pmc_insts.py $(ls $OUT/pmc_INSTS_counters.csv $OUT/pmc_ACTIVE_counters.csv $OUT/pmc_ACTIVE2_counters.csv 2>/dev/null) > $OUT/pmc_insts.json
python - <<PY
import json
d=json.load(open("$OUT/pmc_insts.json"))["per_launch"]
for k,v in d.items():
    if "gibbs" in k: print(k, json.dumps(v))
PY
rm -f $OUT/pmc_*_counters.csv

#!/bin/bash
# One secondary workload, profiled like the headline: its bench line, `rocprofv3 --kernel-trace --stats` of the same command, and the
# HBM traffic of its kernels from two PMC passes (FETCH_SIZE, WRITE_SIZE: one counter per pass, --kernel-trace only), summarised by
# scripts/pmc_summary.py with the calibrated factors.  Run on the GPU box:
#   gpurun --timeout 2400 -- 'bash scripts/profile_mode.sh r05 quilt2_default --mspbwt --rare-common 2 --steps 8 --warmup 2'
# Writes gpurun_out/<tag>/{bench_line,kernel_stats,pmc_traffic}_<name>.*  (copy to profiles/<tag>_*).
TAG=$1; NAME=$2; shift 2
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="$* --no-alone --precision fp64 --dotcall 0 --r2-vs-cpu 0 --no-cpu-baseline"
python bench.py $ARGS > $OUT/bench_line_$NAME.json 2> $OUT/bench_$NAME.err; tail -c 300 $OUT/bench_line_$NAME.json; echo
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$NAME -o stats -- \
    python $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/bench_line_${NAME}_under_rocprof.json 2> $OUT/rocprof_$NAME.err)
find $OUT/stats_$NAME -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats_$NAME.csv \;
rm -rf $OUT/stats_$NAME
head -14 $OUT/kernel_stats_$NAME.csv
# PMC passes over a shorter run of the same workload (one launch set per host thread)
PARGS=$(echo "$ARGS" | sed -E 's/--steps [0-9]+/--steps 4/; s/--warmup [0-9]+/--warmup 1/')
for C in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_${NAME}_$C -o pmc -- \
        python $GRAFT_REPO_ROOT/bench.py $PARGS > $OUT/pmc_${NAME}_$C.json 2> $OUT/pmc_${NAME}_$C.err)
    find $OUT/pmc_${NAME}_$C -name '*counter_collection.csv' -exec cp {} $OUT/pmc_${NAME}_${C}.csv \;
    rm -rf $OUT/pmc_${NAME}_$C
done
python scripts/pmc_summary.py $OUT/pmc_${NAME}_FETCH_SIZE.csv $OUT/pmc_${NAME}_WRITE_SIZE.csv $OUT/pmc_traffic_$NAME.json \
    "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python bench.py $PARGS"
rm -f $OUT/pmc_${NAME}_*.csv

#!/bin/bash
# rocprofv3 --kernel-trace --stats of one bench command (developer aid): gpurun -- 'bash scripts/stats_only.sh r05 NAME <bench flags>'
TAG=$1; NAME=$2; shift 2
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="$* --no-alone --precision fp64 --dotcall 0 --r2-vs-cpu 0 --no-cpu-baseline"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$NAME -o stats -- \
    python $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/bench_line_${NAME}_under_rocprof.json 2> $OUT/rocprof_$NAME.err)
find $OUT/stats_$NAME -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats_$NAME.csv \;
find $OUT/stats_$NAME -name '*kernel_trace.csv' -exec cp {} $OUT/kernel_trace_$NAME.csv \;
rm -rf $OUT/stats_$NAME
head -14 $OUT/kernel_stats_$NAME.csv | cut -c1-200

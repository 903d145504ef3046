#!/bin/bash
# Produces the files profiles/<tag>_* are copied from.  Run on the GPU box:
#   gpurun --timeout 3000 -- 'bash scripts/profile_round.sh r04'        (the PMC passes of the build that dominates: scripts/pmc_round.sh)
# 1. the -m gpu tests; 2. the bench line of the driver's own command (--steps 20 --warmup 5); 3. rocprofv3 --kernel-trace
# --stats of the same command (CPU baseline leg off: it forks 128 host processes the profiler would follow); 4. PMC passes
# (each alone with --kernel-trace: FETCH_SIZE, WRITE_SIZE, instruction counters); 5. the ONT / NIPT / fp64-dosage / msPBWT-mode lines.
TAG=${1:-r04}
WHAT=${2:-all}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
set -x
if [[ $WHAT == all || $WHAT == tests ]]; then
python -m pytest tests -m gpu -x -q --durations=8 > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
fi
if [[ $WHAT == all || $WHAT == bench ]]; then
python bench.py --steps 20 --warmup 5 > $OUT/bench_line_steps20.json 2> $OUT/bench_steps20.err; tail -c 600 $OUT/bench_line_steps20.json
fi
if [[ $WHAT == all || $WHAT == stats ]]; then
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --r2-vs-cpu 0 --dotcall 0 > $OUT/bench_line_under_rocprof.json 2> $OUT/rocprof_stats.err)
find $OUT/stats -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/stats -name '*agent_info.csv' -exec cp {} $OUT/agent_info.csv \;
head -12 $OUT/kernel_stats.csv
rm -rf $OUT/stats
fi
if [[ $WHAT == all || $WHAT == pmc ]]; then
for C in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    N=$(echo $C | cut -d' ' -f1)
    (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$N -o pmc -- \
        python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --r2-vs-cpu 0 --no-alone > $OUT/pmc_$N.json 2> $OUT/pmc_$N.err)
    find $OUT/pmc_$N -name '*counter_collection.csv' -exec cp {} $OUT/pmc_${N}_counters.csv \;
    rm -rf $OUT/pmc_$N
done
python scripts/pmc_summary.py $OUT/pmc_FETCH_SIZE_counters.csv $OUT/pmc_WRITE_SIZE_counters.csv $OUT/pmc_traffic.json \
    "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --r2-vs-cpu 0 --no-alone"
python scripts/pmc_insts.py $OUT/pmc_SQ_INSTS_VALU_counters.csv $OUT/pmc_SQ_INSTS_VMEM_counters.csv > $OUT/pmc_insts.json
cat $OUT/pmc_insts.json | head -40
# the raw per-dispatch counter files are large; keep the summaries only
rm -f $OUT/pmc_*_counters.csv
fi
if [[ $WHAT == all || $WHAT == lines ]]; then
python bench.py --mode ont --steps 12 --warmup 4 --r2-vs-cpu 0 --dotcall 0 > $OUT/bench_line_ont.json 2> $OUT/bench_ont.err; tail -c 300 $OUT/bench_line_ont.json
python bench.py --mode nipt --steps 12 --warmup 4 --r2-vs-cpu 0 > $OUT/bench_line_nipt.json 2> $OUT/bench_nipt.err; tail -c 300 $OUT/bench_line_nipt.json
python bench.py --mspbwt --steps 12 --warmup 4 --dotcall 0 > $OUT/bench_line_mspbwt.json 2> $OUT/bench_mspbwt.err; tail -c 300 $OUT/bench_line_mspbwt.json
python bench.py --mspbwt --steps 20 --warmup 5 --no-alone --precision fp64 --no-cpu-baseline --r2-vs-cpu 0 > $OUT/bench_line_mspbwt_steps20.json 2> $OUT/bench_mspbwt20.err; tail -c 300 $OUT/bench_line_mspbwt_steps20.json
python bench.py --bam --steps 12 --warmup 4 --no-cpu-baseline --r2-vs-cpu 0 > $OUT/bench_line_from_bam.json 2> $OUT/bench_from_bam.err; tail -c 300 $OUT/bench_line_from_bam.json
python bench.py --K 5000 --batch 32 --steps 80 --warmup 16 --dotcall 0 > $OUT/bench_line_configs1_K5000_b32.json 2> $OUT/bench_configs1.err; tail -c 300 $OUT/bench_line_configs1_K5000_b32.json
python bench.py --K 64976 --steps 8 --warmup 2 --no-alone --precision fp64 --no-cpu-baseline --r2-vs-cpu 0 --dotcall 0 > $OUT/bench_line_K64976.json 2> $OUT/bench_K64976.err; tail -c 300 $OUT/bench_line_K64976.json
python bench.py --mode nipt --rare-common 2 --steps 4 --warmup 2 --no-alone --no-cpu-baseline --r2-vs-cpu 0 --dotcall 0 > $OUT/bench_line_nipt_rare_common.json 2> $OUT/bench_nipt_rc.err; tail -c 300 $OUT/bench_line_nipt_rare_common.json
python bench.py --mspbwt --rare-common 2 --steps 8 --warmup 2 --no-alone --precision fp64 --dotcall 0 --r2-vs-cpu 0 > $OUT/bench_line_quilt2_default.json 2> $OUT/bench_quilt2.err; tail -c 300 $OUT/bench_line_quilt2_default.json
fi

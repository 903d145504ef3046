#!/bin/bash
# Round 6's measurement legs (each can be run alone):  gpurun --timeout 3000 -- 'bash scripts/round6.sh <leg>'
#   headline   the driver's own command (CPU baseline measured whole), then rocprofv3 --kernel-trace --stats of the same command
#   pmc        FETCH_SIZE / WRITE_SIZE passes over launches of the dominant kernel (scripts/pmc_round.sh's first two passes)
#   cpu_ont | cpu_nipt | cpu_mspbwt | cpu_quilt2   one secondary workload WITH its whole-sample CPU baseline (128 samples, all cores)
#   hostshare  the headline workload confined to one rank's share of the host at 8 ranks (bench.py --host-share 8)
TAG=r06
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
C="--no-alone --precision fp64 --dotcall 0"
for LEG in "$@"; do
case $LEG in
headline)
    python bench.py --steps 20 --warmup 5 > $OUT/bench_line_steps20.json 2> $OUT/bench_steps20.err; tail -c 400 $OUT/bench_line_steps20.json; echo
    (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- \
        python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --r2-vs-cpu 0 --dotcall 0 > $OUT/bench_line_under_rocprof.json 2> $OUT/rocprof_stats.err)
    find $OUT/stats -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats_bench_steps20.csv \;
    find $OUT/stats -name '*agent_info.csv' -exec cp {} $OUT/agent_info.csv \;
    head -12 $OUT/kernel_stats_bench_steps20.csv
    rm -rf $OUT/stats ;;
pmc)
    CMD="python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --r2-vs-cpu 0 --no-alone --precision fp64 --dotcall 0"
    for N in FETCH_SIZE WRITE_SIZE; do
        (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $N --output-format csv -d $OUT/pmc_$N -o pmc -- $CMD > $OUT/pmc_$N.json 2> $OUT/pmc_$N.err)
        find $OUT/pmc_$N -name '*counter_collection.csv' -exec cp {} $OUT/pmc_${N}_counters.csv \;
        rm -rf $OUT/pmc_$N
    done
    python scripts/pmc_summary.py $OUT/pmc_FETCH_SIZE_counters.csv $OUT/pmc_WRITE_SIZE_counters.csv $OUT/pmc_traffic.json \
        "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --r2-vs-cpu 0 --no-alone --precision fp64 --dotcall 0"
    rm -f $OUT/pmc_*_counters.csv
    python - <<PY
import json
d=json.load(open("$OUT/pmc_traffic.json"))
for k,v in (d.get("per_instantiation") or d.get("kernels") or {}).items():
    if "gibbs" in k: print(k, json.dumps(v)[:300])
PY
    ;;
cpu_ont)    python bench.py --mode ont --steps 12 --warmup 4 --r2-vs-cpu 1 $C > $OUT/bench_line_ont_with_cpu_baseline.json 2> $OUT/bench_ont.err; tail -c 300 $OUT/bench_line_ont_with_cpu_baseline.json; echo ;;
cpu_nipt)   python bench.py --mode nipt --steps 12 --warmup 4 --r2-vs-cpu 1 --cpu-baseline-budget 700 $C > $OUT/bench_line_nipt_with_cpu_baseline.json 2> $OUT/bench_nipt.err; tail -c 300 $OUT/bench_line_nipt_with_cpu_baseline.json; echo ;;
cpu_mspbwt) python bench.py --mspbwt --steps 20 --warmup 5 --r2-vs-cpu 1 $C > $OUT/bench_line_mspbwt_with_cpu_baseline.json 2> $OUT/bench_mspbwt.err; tail -c 300 $OUT/bench_line_mspbwt_with_cpu_baseline.json; echo ;;
cpu_quilt2) python bench.py --mspbwt --rare-common 2 --steps 20 --warmup 4 --r2-vs-cpu 1 --cpu-baseline-budget 700 $C > $OUT/bench_line_quilt2_default_with_cpu_baseline.json 2> $OUT/bench_quilt2.err; tail -c 300 $OUT/bench_line_quilt2_default_with_cpu_baseline.json; echo ;;
hostshare)  python bench.py --steps 20 --warmup 5 --host-share 8 --no-cpu-baseline --r2-vs-cpu 0 $C > $OUT/bench_line_host_share_at_8.json 2> $OUT/bench_hostshare.err; tail -c 300 $OUT/bench_line_host_share_at_8.json; echo ;;
esac
done

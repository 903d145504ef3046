#!/bin/bash
# The round's secondary bench lines, each priced on its own committed counter summary (profiles/r05_pmc_traffic_*.json).
#   gpurun --timeout 3400 -- 'bash scripts/final_lines.sh r05'
TAG=${1:-r05}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
C="--no-alone --precision fp64 --dotcall 0 --r2-vs-cpu 0 --no-cpu-baseline"
run() { local N=$1; shift; python bench.py $* > $OUT/bench_line_$N.json 2> $OUT/bench_$N.err; python - <<PY
import json
try:
    b=json.load(open("$OUT/bench_line_$N.json")); r=b["roofline"]
    print("$N", round(b["value"],2), r["kernel"], "frac", round(r["frac"],3), r["priced_against"][:28], "busy", (b.get("device_phases") or {}).get("busy_frac"))
except Exception as e: print("$N", "FAILED", e)
PY
}
run nipt --mode nipt --steps 12 --warmup 4 $C
run configs1 --K 5000 --batch 32 --steps 80 --warmup 16 $C
run quilt2_default --mspbwt --rare-common 2 --steps 8 --warmup 2 $C
run quilt2_default_steps20 --mspbwt --rare-common 2 --steps 20 --warmup 4 $C
run ont --mode ont --steps 12 --warmup 4 $C
run mspbwt_steps20 --mspbwt --steps 20 --warmup 5 $C
run K64976 --K 64976 --steps 8 --warmup 2 $C
run from_bam --bam --steps 12 --warmup 4 $C
run nipt_rare_common --mode nipt --rare-common 2 --steps 4 --warmup 2 $C
run mixed_and_dotcall --steps 20 --warmup 5 --no-alone --precision both --dotcall 16 --r2-vs-cpu 0 --no-cpu-baseline

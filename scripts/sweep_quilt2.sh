#!/bin/bash
# QUILT2's default mode (use_mspbwt + impute_rare_common): host threads x steps per launch set.  usage: sweep_quilt2.sh "W F" "W F" ...
OUT=$PWD/gpurun_out/r05; mkdir -p $OUT
for CFG in "$@"; do
  set -- $CFG
  python bench.py --mspbwt --rare-common 2 --steps 8 --warmup 2 --no-alone --precision fp64 --dotcall 0 --r2-vs-cpu 0 --no-cpu-baseline --workers $1 --fuse $2 --gate-trace $OUT/q2_gate_w$1_f$2.npy > $OUT/q2_w$1_f$2.json 2> $OUT/q2_w$1_f$2.err
  python - <<PY
import json
b=json.load(open("$OUT/q2_w$1_f$2.json"))
print("workers $1 fuse $2:", round(b["value"],2), "samples/s; chains/launch", b.get("gibbs_chains_per_launch"), "busy", b["device_phases"]["busy_frac"], "slots", b["device_phases"]["gibbs_mean_simd_slots"], [ (k["kernel"], k["launches"], k["avg_launch_ms"]) for k in b["kernels"]], b["host_seconds"])
PY
  python scripts/gate_timeline.py $OUT/q2_gate_w$1_f$2.npy v | tail -60
done

#!/bin/bash
# Developer aid: the three-label sampler alone on the device, one chain per SIMD (896 chains) against the 256-register build
# (1 792 chains), at 20 000 and 5 000 reads (slope = read visits, intercept = grid steps).  gpurun -- 'bash scripts/perf_gibbs3.sh'
for CFG in "${@:-896 20000}" ; do
  set -- $CFG
  echo "== chains $1 reads $2"
  python scripts/perf_gibbs.py --nipt --chains $1 --reads $2 --samples 16 --reps 2 2>&1 | grep "rep 1"
done

#!/bin/bash
# Developer aid: the two-label sampler alone on the device (2 048 chains: the 256-register build).  gpurun -- 'bash scripts/perf_gibbs2.sh'
python scripts/perf_gibbs.py --chains ${1:-2048} --reads ${2:-20000} --samples 16 --reps 2 2>&1 | grep "rep 1"

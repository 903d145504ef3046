#!/bin/bash
OUT=$PWD/gpurun_out/probe; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q --durations=8 > $OUT/pytest_gpu.log 2>&1; tail -14 $OUT/pytest_gpu.log
timeout 600 python bench.py --mspbwt --steps 20 --warmup 5 --no-alone --precision fp64 --no-cpu-baseline --no-scan-check --r2-vs-cpu 0 --gate-trace $OUT/gate_mspbwt.npy > $OUT/bench_mspbwt.json 2> $OUT/bench_mspbwt.err
python -c "import json;d=json.load(open('$OUT/bench_mspbwt.json'));print('mspbwt',d['value'],d['ms_per_step'])"
timeout 600 python bench.py --mode ont --steps 12 --warmup 4 --no-alone --no-cpu-baseline --r2-vs-cpu 0 --fuse 2 > $OUT/bench_ont_fuse2.json 2> $OUT/bench_ont_fuse2.err
python -c "import json;d=json.load(open('$OUT/bench_ont_fuse2.json'));print('ont fuse2',d['value'],d['ms_per_step'])"
timeout 600 python bench.py --mode ont --steps 12 --warmup 4 --no-alone --no-cpu-baseline --r2-vs-cpu 0 > $OUT/bench_ont.json 2> $OUT/bench_ont.err
python -c "import json;d=json.load(open('$OUT/bench_ont.json'));print('ont',d['value'],d['ms_per_step'])"

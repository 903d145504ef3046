#!/bin/bash
OUT=$PWD/gpurun_out/probe; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_pipeline_gpu.py -m gpu -x -q -k "threads" > $OUT/pytest_pipeline.log 2>&1; tail -3 $OUT/pytest_pipeline.log
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --r2-vs-cpu 0 --no-alone --precision fp64 --gate-trace $OUT/gate_tail2.npy > $OUT/bench_tail2.json 2> $OUT/bench_tail2.err
python -c "import json;d=json.load(open('$OUT/bench_tail2.json'));print('split_remainder=1',d['value'],d['ms_per_step'])"

set -x
OUT=$PWD/gpurun_out/exp12
mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
QA_TIMING=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_timing.log 2>&1
python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; tail -c 700 $OUT/bench.json
python bench.py --steps 4 --warmup 1 --no-cpu-baseline --workers 3 > $OUT/bench_w3.json 2> $OUT/bench_w3.err; tail -c 300 $OUT/bench_w3.json

set -x
OUT=$PWD/gpurun_out/exp15
mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
python bench.py --mspbwt --steps 4 --warmup 1 > $OUT/bench_m2.json 2> $OUT/bench_m2.err; tail -c 300 $OUT/bench_m2.json
python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; tail -c 300 $OUT/bench.json

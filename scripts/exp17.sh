set -x
OUT=$PWD/gpurun_out/exp17
mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
python scripts/perf_fullpass.py --P 256 --thin-frac 1 > $OUT/perf_fullpass.log 2>&1; tail -12 $OUT/perf_fullpass.log
python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; tail -c 900 $OUT/bench.json

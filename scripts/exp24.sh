set -x
OUT=$PWD/gpurun_out/r02
mkdir -p $OUT
python bench.py --rare-common 2 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_line_rare_common.json 2> $OUT/bench_rare_common.err; tail -c 400 $OUT/bench_line_rare_common.json; tail -3 $OUT/bench_rare_common.err

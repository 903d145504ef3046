set -x
OUT=$PWD/gpurun_out/r02
mkdir -p $OUT
python bench.py --bam --steps 4 --warmup 1 --no-cpu-baseline > $OUT/bench_line_from_bam.json 2> $OUT/bench_from_bam.err; tail -c 500 $OUT/bench_line_from_bam.json
python bench.py --mode nipt --steps 4 --warmup 1 > $OUT/bench_line_nipt.json 2> $OUT/bench_nipt.err; tail -c 300 $OUT/bench_line_nipt.json

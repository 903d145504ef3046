set -x
OUT=$PWD/gpurun_out/exp19
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --steps 8 --warmup 2 --no-cpu-baseline --split alternate > $OUT/bench_alt.json 2> $OUT/bench_alt.err; tail -c 400 $OUT/bench_alt.json
python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $OUT/bench_halves.json 2> $OUT/bench_halves.err; tail -c 400 $OUT/bench_halves.json
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st -o st -- python $GRAFT_REPO_ROOT/bench.py --mode nipt --steps 2 --warmup 1 --no-cpu-baseline > $OUT/nipt.json 2> $OUT/nipt.err)
find $OUT/st -name '*kernel_stats.csv' -exec cp {} $OUT/nipt_kernel_stats.csv \;
rm -rf $OUT/st
head -12 $OUT/nipt_kernel_stats.csv | cut -c1-150

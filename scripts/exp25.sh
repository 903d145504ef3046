set -x
OUT=$PWD/gpurun_out/exp25
mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -o tr -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench.json 2> $OUT/err.log)
find $OUT/tr -name '*kernel_trace.csv' -exec cp {} $OUT/kernel_trace.csv \;
rm -rf $OUT/tr
ls -la $OUT

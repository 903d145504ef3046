set -x
mkdir -p gpurun_out/exp9
for v in SKIP_BWD SKIP_SHARD; do
  export QUILT_AMD_LIB=$PWD/quilt_amd/csrc/libquilt_amd_$v.so
  python scripts/perf_gibbs.py --chains 512 --samples 64 --reps 2 --reads 20000 > gpurun_out/exp9/g_$v.log 2>&1; tail -2 gpurun_out/exp9/g_$v.log
done
unset QUILT_AMD_LIB
python -m pytest tests -m gpu -x -q --durations=5 > gpurun_out/exp9/pytest.log 2>&1; tail -8 gpurun_out/exp9/pytest.log
python bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/exp9/bench.json 2> gpurun_out/exp9/bench.err; tail -c 1200 gpurun_out/exp9/bench.json

mkdir -p gpurun_out/exp42
timeout 900 python -m pytest tests/test_fullpass_gpu.py tests/test_headline_gpu.py tests/test_golden_gpu.py tests/test_pipeline_gpu.py tests/test_select_gpu.py -m gpu -x -q > gpurun_out/exp42/tests.log 2>&1; tail -3 gpurun_out/exp42/tests.log
python scripts/perf_rank.py > gpurun_out/exp42/rank.log 2>&1; tail -4 gpurun_out/exp42/rank.log

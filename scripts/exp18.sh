set -x
OUT=$PWD/gpurun_out/exp18
mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -6 $OUT/pytest.log

#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of rocprofv3 against known byte counts, per access shape (scripts/micro/fetch_calibrate.hip):
#   hipcc -O3 --offload-arch=gfx950 scripts/micro/fetch_calibrate.hip -o build/fetch_calibrate     (here: cross-compiles)
#   gpurun --timeout 600 -- 'bash scripts/calibrate_fetch.sh r05'
# One counter per pass, --kernel-trace only.  Writes gpurun_out/<tag>/fetch_calibration.json (copy to profiles/).
TAG=${1:-r05}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BIN=$GRAFT_REPO_ROOT/build/fetch_calibrate
$BIN > $OUT/cal_plain.txt
for C in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/cal_$C -o cal -- $BIN > $OUT/cal_$C.txt 2> $OUT/cal_$C.err)
    find $OUT/cal_$C -name '*counter_collection.csv' -exec cp {} $OUT/cal_${C}.csv \;
    rm -rf $OUT/cal_$C
done
python scripts/calibrate_fetch_summary.py $OUT/cal_plain.txt $OUT/cal_FETCH_SIZE.csv $OUT/cal_WRITE_SIZE.csv > $OUT/fetch_calibration.json
cat $OUT/fetch_calibration.json

mkdir -p gpurun_out/exp40
for cfg in "nipt 4 alternate" "nipt 2 halves" "ont 4 alternate" "ont 2 halves"; do
  set -- $cfg
  timeout 900 python bench.py --mode $1 --steps 12 --warmup 4 --no-cpu-baseline --workers $2 --split $3 > gpurun_out/exp40/$1_w$2_$3.log 2>&1
  python - "$1" "$2" "$3" <<'PY'
import json,sys
try:
    d=json.loads(open(f"gpurun_out/exp40/{sys.argv[1]}_w{sys.argv[2]}_{sys.argv[3]}.log").read().strip().splitlines()[-1])
    print(sys.argv[1:], round(d['value'],2), d['host_seconds'])
except Exception as e: print(sys.argv[1:], 'failed', e)
PY
done

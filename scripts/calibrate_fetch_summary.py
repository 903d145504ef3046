"""scripts/calibrate_fetch.sh: requested bytes of every calibration kernel (the binary's own CAL lines) over what rocprofv3's
FETCH_SIZE / WRITE_SIZE report for it (units of 1 024 bytes).  `factor` = requested bytes / (counter x 1024): what a counter
reading has to be multiplied by to give bytes in that access shape."""
import csv
import json
import sys

plain, fetch_csv, write_csv = sys.argv[1:4]
req, ms = {}, {}
for line in open(plain):
    if line.startswith("CAL "):
        f = line.split()
        req[f[1]] = float(f[3])
        ms[f[1]] = float(f[5])


def counter(path, name):
    out = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name:
            k = r["Kernel_Name"].split("(")[0].strip()
            out[k] = out.get(k, 0.0) + float(r["Counter_Value"])
    return out


fs, ws = counter(fetch_csv, "FETCH_SIZE"), counter(write_csv, "WRITE_SIZE")
res = {"what": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over scripts/micro/fetch_calibrate.hip: every kernel moves a "
               "known byte count through a 4 GiB buffer (16 x the Infinity Cache) once; factor = requested bytes / (counter x 1024)",
       "kernels": {}}
for k in req:
    cnt = fs.get(k) if k.startswith("rd") else ws.get(k)
    other = ws.get(k) if k.startswith("rd") else fs.get(k)
    res["kernels"][k] = {"requested_bytes": req[k], "counter": "FETCH_SIZE" if k.startswith("rd") else "WRITE_SIZE",
                         "counter_value_x1024": None if cnt is None else cnt * 1024,
                         "factor": None if not cnt else round(req[k] / (cnt * 1024), 4),
                         "other_counter_x1024": None if other is None else other * 1024,
                         "ms_unprofiled": ms[k], "GBps_unprofiled": round(req[k] / 1e6 / ms[k], 1)}
print(json.dumps(res, indent=1))

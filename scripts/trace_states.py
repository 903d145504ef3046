"""GPU state fractions from a rocprofv3 kernel trace (csv): how long no kernel, only Gibbs launches, only full-panel /
other kernels, or both were in flight, and the number of Gibbs waves in flight.  Usage: trace_states.py <kernel_trace.csv>
[skip_seconds]  (skip: seconds after the first Gibbs launch to leave out -- warm-up)."""
import collections
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
    iv = []
    for r in rows:
        n = r["Kernel_Name"]
        kind = "g" if "k_gibbs" in n else "f"
        waves = int(r["Grid_Size_X"]) // 64
        iv.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind, waves))
    g = [x for x in iv if x[2] == "g"]
    lo = min(x[0] for x in g) + int(skip * 1e9)
    hi = max(x[1] for x in g)
    pts = []
    for s, e, k, w in iv:
        if e < lo or s > hi:
            continue
        pts += [(max(s, lo), 1, k, w), (min(e, hi), -1, k, -w)]
    pts.sort()
    cur = collections.Counter()
    gw = 0
    last = lo
    acc, wacc = collections.Counter(), collections.Counter()
    for t, d, k, w in pts:
        key = ("G" if cur["g"] else "-") + ("F" if cur["f"] else "-")
        acc[key] += t - last
        if cur["g"]:
            wacc[min(gw, 4096) // 256 * 256] += t - last
        cur[k] += d
        if k == "g":
            gw += w
        last = t
    tot = (hi - lo) / 1e9
    print(f"window {tot:.1f} s")
    for k, v in sorted(acc.items(), key=lambda x: -x[1]):
        print(f"  {k}: {v / 1e9:7.2f} s  {100 * v / 1e9 / tot:5.1f} %")
    print("  Gibbs waves of the launches in flight (launched, not necessarily resident):",
          {k: round(v / 1e9, 1) for k, v in sorted(wacc.items())})


if __name__ == "__main__":
    main()

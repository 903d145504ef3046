"""Per-kernel instruction counters from rocprofv3 --pmc passes (counter_collection.csv files) -> JSON on stdout.

usage: python scripts/pmc_insts.py <counter_collection.csv> [...]

Sums every counter per kernel family (k_gibbs, k_fwd64, ...) over all dispatches of a run and divides by the dispatch
count.  SQ_INSTS_* count wave-level instructions; SQ_WAVE_CYCLES counts cycles of resident waves (4 per quad-cycle on
gfx9: the guide's note on SQ_*_CYCLES applies), so ratios between kernels are meaningful, absolute cycles need that factor.
"""
import collections
import csv
import json
import re
import sys


def main():
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for path in sys.argv[1:]:
        for r in csv.DictReader(open(path)):
            m = re.search(r"(k_\w+)", r["Kernel_Name"])
            if not m:
                continue
            k = m.group(1)
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[(k, path)].add(r["Dispatch_Id"])
    out = {}
    for k in sorted(acc):
        n = max(len(v) for (kk, _), v in disp.items() if kk == k)
        out[k] = {"launches": n, **{c: v / n for c, v in sorted(acc[k].items())}}
    json.dump({"per_launch": out}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()

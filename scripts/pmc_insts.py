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
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))   # kernel, counter, file
    disp = collections.defaultdict(set)
    wgs = collections.defaultdict(float)
    for path in sys.argv[1:]:
        seen = set()
        for r in csv.DictReader(open(path)):
            m = re.search(r"(k_\w+(?:<[^>]*>)?)", r["Kernel_Name"])   # per template instantiation
            if not m:
                continue
            k = m.group(1)
            acc[k][r["Counter_Name"]][path] += float(r["Counter_Value"])
            disp[(k, path)].add(r["Dispatch_Id"])
            if (k, r["Dispatch_Id"]) not in seen:
                seen.add((k, r["Dispatch_Id"]))
                try:
                    wgs[(k, path)] += max(1, int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])))
                except (KeyError, ValueError):
                    pass
    out = {}
    for k in sorted(acc):
        n = max(len(v) for (kk, _), v in disp.items() if kk == k)
        w = max([v for (kk, _), v in wgs.items() if kk == k] or [0])
        # a counter collected in several passes (SQ_WAVE_CYCLES) is averaged over them, not summed
        out[k] = {"launches": n, "workgroups_per_launch": w / n if n else 0,
                  **{c: sum(v.values()) / len(v) / n for c, v in sorted(acc[k].items())}}
    json.dump({"per_launch": out}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()

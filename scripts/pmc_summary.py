"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/<tag>_pmc_traffic.json.

usage: python scripts/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> "<command>"

Units and corrections (MI355X_MICROARCH.md, "HBM"): rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB-like units of 1024
bytes... the counters count 64-byte requests; on gfx950 FETCH_SIZE tallies the 128-byte requests of wide coalesced reads
at 64 bytes, so the read side is doubled.  WRITE_SIZE is uncalibrated in the guide; it is reported as is.
"""
import collections
import csv
import json
import re
import sys


def per_kernel(path, counter, pattern=r"(k_\w+)"):
    tot = collections.defaultdict(float)
    n = collections.Counter()
    ms = collections.defaultdict(float)
    wgs = collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        m = re.search(pattern, r["Kernel_Name"])
        if not m:
            continue
        k = m.group(1)
        tot[k] += float(r["Counter_Value"])
        n[k] += 1
        ms[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        # workgroups of the launch (a Gibbs launch: one per chain), from whichever geometry columns the csv carries
        try:
            wgs[k] += max(1, int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])))
        except (KeyError, ValueError):
            pass
    return tot, n, ms, wgs


def main():
    fetch_csv, write_csv, out, cmd = sys.argv[1:5]
    f, nf, msf, wg = per_kernel(fetch_csv, "FETCH_SIZE")
    w, nw, _, _ = per_kernel(write_csv, "WRITE_SIZE")
    res = {"command": cmd, "units": "bytes; FETCH_SIZE (KB) x 1024 x 2 (gfx950 wide-read correction), WRITE_SIZE (KB) x 1024",
           "kernels": {}, "instantiations": {}}
    # the same per template instantiation (k_gibbs<10, 1, true> -- two chains per SIMD -- and k_gibbs<10, 1, false> are
    # different code: bench.py prices a launch with the figures of the build it actually ran)
    INST = r"(k_\w+(?:<[^>]*>)?)"
    fi, nfi, msfi, wgi = per_kernel(fetch_csv, "FETCH_SIZE", INST)
    wi, _, _, _ = per_kernel(write_csv, "WRITE_SIZE", INST)
    for k in sorted(fi):
        if "<" not in k:
            continue
        fb, wb = fi[k] * 1024 * 2, wi.get(k, 0.0) * 1024
        res["instantiations"][k] = {"launches": nfi[k], "hbm_bytes_per_launch": (fb + wb) / max(nfi[k], 1), "total_ms": msfi[k],
                                    "workgroups": wgi.get(k, 0),
                                    "hbm_bytes_per_workgroup": (fb + wb) / wgi[k] if wgi.get(k) else None}
    for k in sorted(f):
        fb = f[k] * 1024 * 2
        wb = w.get(k, 0.0) * 1024
        res["kernels"][k] = {"launches": nf[k], "fetch_bytes": fb, "write_bytes": wb,
                             "hbm_bytes_per_launch": (fb + wb) / max(nf[k], 1), "total_ms": msf[k]}
        if wg.get(k):   # launches of different sizes in one run (a Gibbs launch of a whole batch, of the phasing chains alone)
            res["kernels"][k]["workgroups"] = wg[k]
            res["kernels"][k]["hbm_bytes_per_workgroup"] = (fb + wb) / wg[k]
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res["kernels"].items():
        print(k, v["launches"], f"{v['hbm_bytes_per_launch'] / 1e9:.2f} GB/launch", f"{v['total_ms']:.1f} ms")


if __name__ == "__main__":
    main()

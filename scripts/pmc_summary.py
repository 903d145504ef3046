"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/<tag>_pmc_traffic.json.

usage: python scripts/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> "<command>"

Units and corrections (MI355X_MICROARCH.md, "HBM"): rocprofv3 reports FETCH_SIZE / WRITE_SIZE in units of 1024 bytes; on gfx950
FETCH_SIZE tallies the 128-byte requests of coalesced reads at 64 bytes, so the read side is doubled.  The guide calibrates that
for 16-byte-per-lane streaming only and leaves other widths and WRITE_SIZE open; CALIBRATED HERE (round 5) on this library's own
access shapes -- scripts/micro/fetch_calibrate.hip under the same two counter passes, profiles/r05_fetch_calibration.json: known
byte counts through a 4 GiB buffer give FETCH_SIZE x 1024 x 2.000 for 16-, 8- and 4-byte-per-lane contiguous loads alike, x 2
on the 128-byte lines touched for the sampler's state columns (raw_buffer_load_b64, 600 of 640 rows of a 5 120-byte pitch:
4 864 bytes fetched for 4 800 requested), and WRITE_SIZE x 1024 x 1.000 for 16- and 8-byte-per-lane stores and the column
stores.  The factors below are read from that file when it is there (FETCH: rd8buf's line-level 2.0; WRITE: 1.0).
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


def calibration():
    """(read factor, write factor, source) from the committed calibration, else the guide's 2 / 1."""
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r05_fetch_calibration.json")
    try:
        k = json.load(open(path))["kernels"]
        rf = round(k["rd16"]["factor"], 2)   # (= rd8 = rd4; rd8buf fetches whole lines: 2.0 x the lines touched)
        wf = round(k["wr8buf"]["factor"], 2)
        return rf, wf, "profiles/r05_fetch_calibration.json (scripts/micro/fetch_calibrate.hip)"
    except (OSError, KeyError, TypeError, ValueError):
        return 2.0, 1.0, "MI355X_MICROARCH.md (16 B/lane streaming reads); WRITE_SIZE as reported"


def main():
    fetch_csv, write_csv, out, cmd = sys.argv[1:5]
    RF, WF, cal_src = calibration()
    f, nf, msf, wg = per_kernel(fetch_csv, "FETCH_SIZE")
    w, nw, _, _ = per_kernel(write_csv, "WRITE_SIZE")
    res = {"command": cmd, "units": f"bytes; FETCH_SIZE x 1024 x {RF}, WRITE_SIZE x 1024 x {WF}", "calibration": cal_src,
           "kernels": {}, "instantiations": {}}
    # the same per template instantiation (k_gibbs<10, 1, true> -- two chains per SIMD -- and k_gibbs<10, 1, false> are
    # different code: bench.py prices a launch with the figures of the build it actually ran)
    INST = r"(k_\w+(?:<[^>]*>)?)"
    fi, nfi, msfi, wgi = per_kernel(fetch_csv, "FETCH_SIZE", INST)
    wi, _, _, _ = per_kernel(write_csv, "WRITE_SIZE", INST)
    for k in sorted(fi):
        if "<" not in k:
            continue
        fb, wb = fi[k] * 1024 * RF, wi.get(k, 0.0) * 1024 * WF
        res["instantiations"][k] = {"launches": nfi[k], "hbm_bytes_per_launch": (fb + wb) / max(nfi[k], 1), "total_ms": msfi[k],
                                    "workgroups": wgi.get(k, 0),
                                    "hbm_bytes_per_workgroup": (fb + wb) / wgi[k] if wgi.get(k) else None}
    for k in sorted(f):
        fb = f[k] * 1024 * RF
        wb = w.get(k, 0.0) * 1024 * WF
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

"""Fractions of time with 0 / 1 / 2 ... host threads inside a native (device) call, from a QUILT_AMD_TRACE dump, and what the
threads were doing while none was.  Usage: host_trace_states.py <trace.json> [skip_seconds]"""
import collections
import json
import sys


def main():
    sp = json.load(open(sys.argv[1]))
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
    dev = [x for x in sp if x[1].startswith("device:")]
    lo = min(x[2] for x in dev) + skip
    hi = max(x[3] for x in dev)
    pts = []
    for th, name, t0, t1 in dev:
        if t1 < lo or t0 > hi:
            continue
        pts += [(max(t0, lo), 1, name), (min(t1, hi), -1, name)]
    pts.sort()
    cur = 0
    last = lo
    acc = collections.Counter()
    idle = []
    for t, d, name in pts:
        acc[cur] += t - last
        if cur == 0 and t > last:
            idle.append((last, t))
        cur += d
        last = t
    tot = hi - lo
    print(f"window {tot:.1f} s; threads inside a native call:")
    for k in sorted(acc):
        print(f"  {k}: {acc[k]:7.2f} s  {100 * acc[k] / tot:5.1f} %")
    by = collections.Counter()
    for th, name, t0, t1 in dev:
        by[name] += max(0.0, min(t1, hi) - max(t0, lo))
    print("  thread-seconds per native call:", {k: round(v, 1) for k, v in by.items()})
    # host phases overlapping the idle intervals
    host = [x for x in sp if not x[1].startswith("device:")]
    over = collections.Counter()
    for a, b in idle:
        for th, name, t0, t1 in host:
            o = min(b, t1) - max(a, t0)
            if o > 0:
                over[name] += o
    print("  host phases open during the idle time (thread-seconds):", {k: round(v, 2) for k, v in over.most_common()})
    idle.sort(key=lambda x: x[0] - x[1])
    print("  longest idle intervals (s):", [round(b - a, 3) for a, b in idle[:12]])


if __name__ == "__main__":
    main()

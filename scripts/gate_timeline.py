"""Summarise a device-gate hold trace (bench.py --gate-trace FILE.npy): where the device's time goes between launch sets.

Rows: request, admit, kernels done, release [ms since the timed region began], SIMD slots (0: exclusive), thread."""
import sys

import numpy as np


def union(iv):
    iv = sorted(iv)
    tot, cur_a, cur_b = 0.0, None, None
    for a, b in iv:
        if cur_b is None or a > cur_b:
            if cur_b is not None:
                tot += cur_b - cur_a
            cur_a, cur_b = a, b
        else:
            cur_b = max(cur_b, b)
    if cur_b is not None:
        tot += cur_b - cur_a
    return tot


def main(path, verbose=False):
    tr = np.load(path)
    tr = tr[np.argsort(tr[:, 1])]
    t_lo, t_hi = tr[:, 0].min(), tr[:, 3].max()
    span = t_hi - t_lo
    ex = tr[tr[:, 4] == 0]
    gb = tr[tr[:, 4] > 0]
    held = union([(r[1], r[3]) for r in tr])
    print(f"{len(tr)} holds over {span / 1e3:.2f} s: held {held / span:.3f}; exclusive {len(ex)} holds "
          f"{union([(r[1], r[3]) for r in ex]) / span:.3f}; Gibbs {len(gb)} holds {union([(r[1], r[3]) for r in gb]) / span:.3f}")
    if len(gb):
        d = gb[:, 3] - gb[:, 1]
        k = gb[:, 2] - gb[:, 1]
        print(f"  Gibbs holds: mean {d.mean():.0f} ms (kernels + labels back {k.mean():.0f} ms, outputs back {(d - k).mean():.0f} ms); "
              f"slots mean {gb[:, 4].mean():.0f}; slot-time / (1024 x span) = {(gb[:, 4] * d).sum() / 1024 / span:.3f}")
        for s in np.unique(gb[:, 4]):
            m = gb[:, 4] == s
            print(f"    {int(s):5d} slots: {m.sum():3d} holds, mean {d[m].mean():.0f} ms, queued {(gb[m, 1] - gb[m, 0]).mean():.0f} ms")
    if len(ex):
        d = ex[:, 3] - ex[:, 1]
        print(f"  exclusive holds: mean {d.mean():.0f} ms, queued {(ex[:, 1] - ex[:, 0]).mean():.0f} ms")
    # idle gaps: time with no holder
    ev = sorted([(r[1], 1) for r in tr] + [(r[3], -1) for r in tr])
    act, last, gaps = 0, t_lo, []
    for t, s in ev:
        if act == 0 and t > last:
            gaps.append((last, t))
        act += s
        last = t
    g = np.array([b - a for a, b in gaps]) if gaps else np.zeros(0)
    print(f"  idle gaps: {len(g)} totalling {g.sum() / span:.3f} of the span; the 5 longest [ms]: {np.sort(g)[-5:][::-1].round(0)}")
    # threads: share of time queued / holding / elsewhere (host)
    for th in np.unique(tr[:, 5]):
        m = tr[:, 5] == th
        q = (tr[m, 1] - tr[m, 0]).sum()
        h = (tr[m, 3] - tr[m, 1]).sum()
        print(f"  thread {int(th):8d}: {m.sum():3d} holds, queued {q / span:.2f}, holding {h / span:.2f}, elsewhere {1 - (q + h) / span:.2f}")
    if verbose:
        for r in tr:
            print(f"    {r[0]:9.0f} {r[1]:9.0f} {r[2]:9.0f} {r[3]:9.0f}  slots {int(r[4]):5d}  thread {int(r[5])}")


if __name__ == "__main__":
    main(sys.argv[1], verbose=len(sys.argv) > 2)

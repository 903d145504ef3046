"""Developer aid: time the full-panel pass kernels at a given (K, nSNPs, P) on the GPU."""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quilt_amd.native import DevicePanel, check, last_fullpass_timing_ms, lib, ptr  # noqa: E402
from quilt_amd.synth import make_synthetic_panel  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--K", type=int, default=50000)
ap.add_argument("--T", type=int, default=64000)
ap.add_argument("--P", type=int, default=256)
ap.add_argument("--thin-frac", type=float, default=0.0, help="fraction of thin passes")
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--ktop", type=int, default=5)
ap.add_argument("--fp64", action="store_true", help="dosage passes with fp64 state (k_fwd64 + k_bwd64d)")
ap.add_argument("--driver", action="store_true", help="time the driver path (qa_fullpass_reads_batch: fused top-K)")
a = ap.parse_args()

t0 = time.time()
panel = make_synthetic_panel(K=a.K, nSNPs=a.T, seed=4916, keep_rhb_t=True)
print(f"panel built in {time.time() - t0:.1f}s: K={panel.K} G={panel.nGrids}", flush=True)
dev = DevicePanel(panel)
if a.fp64:
    dev.set_dosage_precision(64)
G, T = panel.nGrids, panel.nSNPs
rng = np.random.default_rng(1)
# synthetic gl: ~10 % of SNPs informative per label
gl = np.ones((a.P, T, 2))
for p in range(a.P):
    idx = rng.choice(T, size=T // 10, replace=False)
    ref = rng.random(len(idx)) < 0.7
    e = 10.0 ** (-rng.integers(20, 41, size=len(idx)) / 10.0)
    gl[p, idx, 0] = np.where(ref, 1 - e, e / 3)
    gl[p, idx, 1] = np.where(ref, e / 3, 1 - e)
cols = np.full(G, -1, dtype=np.int32)
w = np.sort(rng.choice(np.arange(1, G), size=max(1, G // 10), replace=False))
cols[w] = np.arange(len(w))
n_thin = len(w)
want = (np.arange(a.P) >= int(a.P * a.thin_frac)).astype(np.int32)
dosage = np.zeros((a.P, T))
bptr = np.zeros(a.P * n_thin + 1, dtype=np.int32)
cap = a.P * n_thin * 64
bidx = np.zeros(cap, dtype=np.int32)
bval = np.zeros(cap)
lib().qa_profile_name.restype = C.c_char_p
NAMES = [lib().qa_profile_name(C.c_int32(k)).decode() for k in range(lib().qa_profile_count())]
for r in range(a.reps):
    t0 = time.time()
    lib().qa_profile_reset()
    check(lib().qa_fullpass_batch(dev.handle, C.c_int32(a.P), ptr(gl), ptr(want), ptr(cols), C.c_int32(a.ktop),
                                  ptr(dosage), ptr(bptr), ptr(bidx), ptr(bval), C.c_int64(cap)))
    wall = time.time() - t0
    tm = last_fullpass_timing_ms()
    nd = int(want.sum()); nt = a.P - nd
    alg = (nd * 10.0 + nt * 2.8) * panel.K * G
    line = []
    for k, nm in enumerate(NAMES):
        ms, n, b = C.c_double(), C.c_int64(), C.c_double()
        lib().qa_profile_get(C.c_int32(k), C.byref(ms), C.byref(n), C.byref(b))
        if n.value:
            line.append(f"{nm} {ms.value:.1f} ms ({b.value / 1e6 / max(ms.value, 1e-9):.0f} GB/s)")
    print(f"rep {r}: wall {wall:.3f}s  " + "  ".join(line), flush=True)
print("dosage range", dosage[want == 1].min() if want.any() else None, dosage.max())

"""Developer aid: the fp64 ranking passes of the driver path (fused top-K picker) at production scale: kernel time against the
number of thinned grids, i.e. what one thinned grid's picker costs."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from quilt_amd import native
from quilt_amd.driver import HipBackend, thinned_grid_columns
from quilt_amd.native import DevicePanel
from quilt_amd.synth import make_synthetic_panel, make_synthetic_sample

panel = make_synthetic_panel(K=50000, nSNPs=64000, seed=4916)
dev = DevicePanel(panel)
be = HipBackend(dev)
ns, nch = 16, 128          # 128 chains x 2 labels = 256 passes: one round of compute units
samples = [make_synthetic_sample(panel, seed=1000 + i, n_reads=20000) for i in range(ns)]
cs = [c % ns for c in range(nch)]
labels = [(np.arange(samples[s].nReads) % 2 + 1).astype(np.int32) if False else samples[s].truth_label.astype(np.int32) for s in cs]
names = [native.lib().qa_profile_name(i).decode() for i in range(native.lib().qa_profile_count())]
for frac in (0.1, 0.05, 0.02, 0.005):
    cols = thinned_grid_columns(panel.nGrids, frac)
    for rep in range(2):
        native.lib().qa_profile_reset()
        t0 = time.time()
        be.fullpass_reads_batch(samples, cs, labels, [0] * nch, [1] * nch, cols, 5, 1e-10, 5)
        wall = time.time() - t0
        out = {}
        for i, n in enumerate(names):
            ms, cnt, b = C.c_double(), C.c_int64(), C.c_double()
            native.lib().qa_profile_get(i, C.byref(ms), C.byref(cnt), C.byref(b))
            if cnt.value: out[n] = (round(ms.value, 2), cnt.value)
    print(f"thin {frac}: {int((cols >= 0).sum())} grids, wall {wall:.2f} s, {out}", flush=True)
dev.close()

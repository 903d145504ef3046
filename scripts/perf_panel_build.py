"""Developer aid: time the device panel builder against the host restatement at production size."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quilt_amd.native import DevicePanel
from quilt_amd.panel import make_rhb_t_equality
from quilt_amd.synth import make_synthetic_panel

panel = make_synthetic_panel(K=50000, nSNPs=64000, seed=4916)
t0 = time.time(); ref = make_rhb_t_equality(panel.rhb_t, 255, panel.nSNPs, panel.ref_error); t_host = time.time() - t0
for rep in range(3):
    t0 = time.time(); dev = DevicePanel.from_rhb(panel); t_dev = time.time() - t0
    if rep < 2: dev.close()
hm, B, off, sk, sw = dev.export_tables()
print(f"host (numpy) {t_host:.2f} s; device incl. upload of rhb_t ({panel.rhb_t.nbytes / 1e6:.0f} MB) {t_dev:.3f} s; "
      f"equal: {np.array_equal(hm, ref['hapMatcherR']) and np.array_equal(B, ref['distinctHapsB'])}; specials {len(sk)}")

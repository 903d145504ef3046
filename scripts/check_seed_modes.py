"""Developer aid: one seed / parameter set of scripts/check_pipeline_seeds.py through four device variants (native loop or the Python
driver, fp64 or fp32-state dosage passes) against the CPU oracle: which of them follows the oracle's labels?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from quilt_amd.driver import Driver, DriverParams, HipBackend
from quilt_amd.impute import impute_samples
from quilt_amd.native import DevicePanel
from quilt_amd.synth import make_synthetic_panel, make_synthetic_sample
from tests.oracle_backend import OracleBackend

sd = int(sys.argv[1]) if len(sys.argv) > 1 else 0
kw = dict(Ksubset=int(sys.argv[2]) if len(sys.argv) > 2 else 128, Knew=int(sys.argv[3]) if len(sys.argv) > 3 else 128)
panel = make_synthetic_panel(K=5000, nSNPs=3200, seed=11, ref_error=1e-3, nGen=100, expRate=1.0)
samples = [make_synthetic_sample(panel, seed=5000 + 10 * sd + i, n_reads=800) for i in range(2)]
prm = DriverParams(nGibbsSamples=3, seed=100 + sd, **kw)
ref = Driver(panel, OracleBackend(panel), prm).run(samples)
for prec in (64, 32):
    dev = DevicePanel(panel)
    dev.set_dosage_precision(prec)
    for name, run in (("native", lambda: impute_samples([dev], samples, prm)), ("python", lambda: Driver(panel, HipBackend(dev), prm).run(samples))):
        got = run()
        print(prec, name, [(bool(np.array_equal(g.read_labels, r.read_labels)), float(np.abs(g.dosage - r.dosage).max())) for g, r in zip(got, ref)], flush=True)
    dev.close()

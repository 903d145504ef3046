"""Developer aid: the whole pipeline on the device -- the native loop, qa_impute_samples, fp64 passes -- vs the same loop on the
CPU oracle over several seeds / parameter sets (medium panel): are the consensus labels identical, how far are the dosages apart?
Usage: python scripts/check_pipeline_seeds.py [n_seeds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from quilt_amd.driver import Driver, DriverParams
from quilt_amd.impute import impute_samples
from quilt_amd.native import DevicePanel
from quilt_amd.synth import make_synthetic_panel, make_synthetic_sample
from tests.oracle_backend import OracleBackend

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
panel = make_synthetic_panel(K=5000, nSNPs=3200, seed=11, ref_error=1e-3, nGen=100, expRate=1.0)
dev = DevicePanel(panel)
dev.set_dosage_precision(64)
bad = 0
for sd in range(n_seeds):
    for tag, kw in (("K200/80", dict(Ksubset=200, Knew=80)), ("K128/128", dict(Ksubset=128, Knew=128)),
                    ("m2", dict(Ksubset=100, Knew=100, use_mspbwt=True, mspbwt_nindices=2)),
                    ("nipt", dict(Ksubset=128, Knew=128, method="nipt"))):
        ff = 0.2 if tag == "nipt" else 0.0
        samples = [make_synthetic_sample(panel, seed=5000 + 10 * sd + i, n_reads=800, ff=ff) for i in range(2)]
        prm = DriverParams(nGibbsSamples=3, seed=100 + sd, **kw)
        t0 = time.time()
        got = impute_samples([dev], samples, prm)
        ref = Driver(panel, OracleBackend(panel), prm).run(samples)
        for i, (g, r) in enumerate(zip(got, ref)):
            same = np.array_equal(g.read_labels, r.read_labels)
            dd = np.abs(g.dosage - r.dosage).max()
            bad += (not same) or dd > 1e-8
            print(f"seed {sd} {tag} sample {i}: labels identical {same}, max|d dosage| {dd:.2e}, {time.time() - t0:.1f} s", flush=True)
print("MISMATCHES", bad)
dev.close()

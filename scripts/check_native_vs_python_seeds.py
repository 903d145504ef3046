"""Developer aid: the native loop (qa_impute_samples) against the Python statement of it (quilt_amd/driver.py) on the SAME device,
over the seeds / parameter sets of scripts/check_pipeline_seeds.py: every output must be equal bit for bit.
Usage: python scripts/check_native_vs_python_seeds.py [n_seeds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from quilt_amd.driver import Driver, DriverParams, HipBackend
from quilt_amd.impute import impute_samples
from quilt_amd.native import DevicePanel
from quilt_amd.synth import make_synthetic_panel, make_synthetic_sample

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
panel = make_synthetic_panel(K=5000, nSNPs=3200, seed=11, ref_error=1e-3, nGen=100, expRate=1.0)
dev = DevicePanel(panel)
dev.set_dosage_precision(64)
bad = n = refetch = 0
for sd in range(n_seeds):
    for tag, kw in (("K200/80", dict(Ksubset=200, Knew=80)), ("K128/128", dict(Ksubset=128, Knew=128)),
                    ("m2", dict(Ksubset=100, Knew=100, use_mspbwt=True, mspbwt_nindices=2)),
                    ("nipt", dict(Ksubset=128, Knew=128, method="nipt"))):
        ff = 0.2 if tag == "nipt" else 0.0
        samples = [make_synthetic_sample(panel, seed=5000 + 10 * sd + i, n_reads=800, ff=ff) for i in range(2)]
        prm = DriverParams(nGibbsSamples=3, seed=100 + sd, **kw)
        got, st = impute_samples([dev], samples, prm, return_stats=True)
        want = Driver(panel, HipBackend(dev), prm).run(samples)
        refetch += st["full_list_refetches"]
        for g, w in zip(got, want):
            n += 1
            same = (np.array_equal(g.read_labels, w.read_labels) and np.array_equal(g.dosage, w.dosage) and np.array_equal(g.gp_t, w.gp_t)
                    and np.array_equal(g.phasing_haps, w.phasing_haps))
            if not same:
                bad += 1
                print(f"seed {sd} {tag}: native and Python loop differ", flush=True)
print(f"{n} sample runs, {refetch} complete-list fetches, MISMATCHES {bad}")
dev.close()

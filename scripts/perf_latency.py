"""One sample alone through qa_impute_samples at the headline panel size (the quick-start's shape): seconds, for the sampler's
waves-per-chain geometries (QA_GIBBS_NW).   gpurun -- 'for NW in 0 2 5; do QA_NW=$NW python scripts/perf_latency.py; done'"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
nw = os.environ.get("QA_NW", "0")
if nw != "0":
    os.environ["QA_GIBBS_NW"] = nw
import numpy as np
from quilt_amd.driver import DriverParams
from quilt_amd.impute import impute_samples
from quilt_amd.native import DevicePanel
from quilt_amd.synth import make_synthetic_panel, make_synthetic_sample
panel = make_synthetic_panel(K=50000, nSNPs=64000, seed=4916)
dev = DevicePanel(panel)
dev.set_dosage_precision(64)
dev.set_exclusive(True)
s = [make_synthetic_sample(panel, seed=1000, n_reads=20000)]
prm = DriverParams(seed=1)
impute_samples([dev], s, prm)
t = time.perf_counter()
r = impute_samples([dev], s, prm)
tp = time.perf_counter() - t
print("QA_GIBBS_NW", nw, "one sample: %.3f s" % tp, "labels checksum", int(np.sum(r[0].read_labels * np.arange(1, len(r[0].read_labels) + 1) % 1000003)))
if os.environ.get("QA_VALIDATION", "0") != "0":
    # the same sample in VALIDATION MODE (qa_panel_set_sum_order: every K-wide sum of the full-panel passes in the order the reference's
    # code adds it, one lane adding K values per grid) -- what the bit-identity with the CPU path costs at the headline panel size
    import json
    dev.set_sum_order(1)
    impute_samples([dev], s, prm)
    t = time.perf_counter()
    v = impute_samples([dev], s, prm)
    tv = time.perf_counter() - t
    print(json.dumps({"workload": "one 1x sample (20 000 reads), K = 50 000 x 64 000 SNPs, QUILT defaults, qa_impute_samples, fp64 dosage passes",
                      "production_mode_s": round(tp, 3), "validation_mode_s": round(tv, 3),
                      "labels_identical_to_production_mode": bool(np.array_equal(v[0].read_labels, r[0].read_labels)),
                      "max_abs_dosage_diff_vs_production_mode": float(np.abs(v[0].dosage - r[0].dosage).max())}))
dev.close()

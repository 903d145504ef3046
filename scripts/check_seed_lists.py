"""Developer aid: where do the device pipeline and the CPU-oracle pipeline part?  One seed of scripts/check_pipeline_seeds.py with the
selection made on the HOST from the lists either backend returns (HipBackend.select_on_device off): the first selection whose lists
differ is printed list by list."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import quilt_amd.driver as drv
from quilt_amd.driver import Driver, DriverParams, HipBackend
from quilt_amd.native import DevicePanel
from quilt_amd.synth import make_synthetic_panel, make_synthetic_sample
from tests.oracle_backend import OracleBackend

sd = int(sys.argv[1]) if len(sys.argv) > 1 else 3
kw = dict(Ksubset=int(sys.argv[2]) if len(sys.argv) > 2 else 128, Knew=int(sys.argv[3]) if len(sys.argv) > 3 else 128)
if os.environ.get("QA_CHECK_QUICK_START"):   # the panel of test_quick_start_shaped_run_bam_to_vcf, QUILT's defaults
    from quilt_amd.synth import make_1000g_like_panel
    panel = make_1000g_like_panel(K=5008, nSNPs=3200, seed=2504)
    samples = [make_synthetic_sample(panel, seed=77, n_reads=1000)]
    prm = DriverParams(seed=sd)
else:
    panel = make_synthetic_panel(K=5000, nSNPs=3200, seed=11, ref_error=1e-3, nGen=100, expRate=1.0)
    samples = [make_synthetic_sample(panel, seed=5000 + 10 * sd + i, n_reads=800) for i in range(1)]
    prm = DriverParams(nGibbsSamples=3, seed=100 + sd, **kw)
log = {}
orig = drv.everything_select_good_haps_dense
def logged(Knew, K_top, top, prev, K, seed, truncated=False):
    log.setdefault(cur[0], []).append((int(seed), np.array(top), bool(truncated)))
    return orig(Knew, K_top, top, prev, K, seed, truncated)
drv.everything_select_good_haps_dense = logged
cur = ["oracle"]
ref = Driver(panel, OracleBackend(panel), prm).run(samples)
dev = DevicePanel(panel)
dev.set_dosage_precision(64)
n_dup = len(panel.hapMatcherR) - len(np.unique(np.asarray(panel.hapMatcherR), axis=0))
print("haplotypes that repeat another one over the whole region:", n_dup, "of", panel.K)
be = HipBackend(dev)
be.select_on_device = False
cur[0] = "device"
got = Driver(panel, be, prm).run(samples)
print("labels identical", np.array_equal(got[0].read_labels, ref[0].read_labels), len(log["oracle"]), len(log["device"]))
for (s1, t1, tr1), (s2, t2, tr2) in zip(log["oracle"], log["device"]):
    if s1 != s2:
        print("seeds part", s1, s2); break
    w = max(t1.shape[2], t2.shape[2])
    a = np.zeros(t1.shape[:2] + (w,), dtype=np.int64); a[:, :, :t1.shape[2]] = t1
    b = np.zeros(t2.shape[:2] + (w,), dtype=np.int64); b[:, :, :t2.shape[2]] = t2
    if not np.array_equal(a, b):
        print("first differing selection: seed", s1, "truncated", tr1, tr2)
        for l in range(a.shape[0]):
            for g in range(a.shape[1]):
                if not np.array_equal(a[l, g], b[l, g]):
                    print(" label", l, "thinned grid", g, "oracle", a[l, g].tolist(), "device", b[l, g].tolist())
                    ks = sorted(set(a[l, g].tolist()) ^ set(b[l, g].tolist()))
                    hm = np.asarray(panel.hapMatcherR)
                    for k in ks:
                        if k > 0:
                            same = [int(j) for j in set(a[l, g].tolist()) | set(b[l, g].tolist()) if j > 0 and j != k and np.array_equal(hm[j - 1], hm[k - 1])]
                            print("   haplotype", k, "has the same codes at every grid as", same)
        break
else:
    print("all selections saw the same lists")
dev.close()

import numpy as np
import quilt_amd.driver as D
from quilt_amd.driver import Driver, DriverParams, HipBackend
from quilt_amd.native import DevicePanel, DeviceRareCommon
from quilt_amd.synth import make_synthetic_panel, make_rare_common, make_synthetic_sample_rare_common
from tests.oracle_backend import OracleBackend

panel = make_synthetic_panel(K=5000, nSNPs=3200, seed=11)
rc = make_rare_common(panel, 4)
samples = [make_synthetic_sample_rare_common(panel, rc, 2500 + i, n_reads=800, ff=0.15 + 0.1 * i)[0] for i in range(2)]
prm = DriverParams(nGibbsSamples=2, Ksubset=128, Knew=128, seed=5, impute_rare_common=True, method="nipt")
rec = {}
orig = D.get_initial_read_labels_nipt
def make(tag):
    def f(e, ff, rng):
        H = orig(e, ff, rng)
        rec.setdefault(tag, []).append((e.copy(), H.copy()))
        return H
    return f
dev = DevicePanel(panel); drc = DeviceRareCommon(dev, rc)
D.get_initial_read_labels_nipt = make("gpu")
got = Driver(panel, HipBackend(dev, drc), prm, rare_common=rc).run(samples)
D.get_initial_read_labels_nipt = make("cpu")
ref = Driver(panel, OracleBackend(panel, rc), prm, rare_common=rc).run(samples)
for i, ((eg, Hg), (ec, Hc)) in enumerate(zip(rec["gpu"], rec["cpu"])):
    d = np.abs(eg - ec)
    flips = ((eg > 0.5) != (ec > 0.5)).sum()
    print(i, "max|de|", d.max(), "threshold flips", int(flips), "labels differ", int((Hg != Hc).sum()))
    if flips:
        w = np.nonzero(((eg > 0.5) != (ec > 0.5)).any(axis=0))[0][:3]
        print("   e.g.", eg[:, w].T, ec[:, w].T)

# per-chain all-SNP results
caps = {}
orig_round = Driver._rare_common_round
def wrap(tag):
    def f(self, chains):
        orig_round(self, chains)
        caps.setdefault(tag, []).extend([(ch.i_sample, ch.i_chain, [h.copy() for h in ch.hap_all], ch.which_haps_to_use.copy()) for ch in chains])
    return f
Driver._rare_common_round = wrap("gpu")
got = Driver(panel, HipBackend(dev, drc), prm, rare_common=rc).run(samples)
Driver._rare_common_round = wrap("cpu")
ref = Driver(panel, OracleBackend(panel, rc), prm, rare_common=rc).run(samples)
for a, b in zip(caps["gpu"], caps["cpu"]):
    d = max(np.abs(x - y).max() for x, y in zip(a[2], b[2]))
    print(a[0], a[1], "which equal", np.array_equal(a[3], b[3]), "max|d hap_all|", d)
for g, r in zip(got, ref):
    print("final max|d|", np.abs(g.dosage - r.dosage).max(), np.abs(g.fet_dosage - r.fet_dosage).max(), g.nDosage, r.nDosage)

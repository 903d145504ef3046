"""Developer aid: tests/test_pipeline_gpu.py's medium run on the HIP backend and on the CPU oracle with every round's
selection recorded: where do the two pipelines first part, and is it a tie?  (Panel haplotypes that coincide over the
region have posteriors equal up to their last bits; which of them a list ranks first is rounding noise -- DESIGN.md 4.4.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from quilt_amd.driver import Driver, DriverParams, HipBackend
from quilt_amd.native import DevicePanel
from quilt_amd.synth import make_synthetic_panel, make_synthetic_sample
from tests.oracle_backend import OracleBackend

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 5
panel = make_synthetic_panel(K=5000, nSNPs=3200, seed=11)
samples = [make_synthetic_sample(panel, seed=1000 + i, n_reads=1000) for i in range(3)]
prm = DriverParams(nGibbsSamples=3, Ksubset=200, Knew=200, seed=seed)
log = {}
orig = Driver._round
def wrap(tag):
    def f(self, chains, i_it):
        r = orig(self, chains, i_it)
        for ch in chains:
            log.setdefault(tag, []).append((ch.i_sample, ch.i_chain, i_it, ch.which_haps_to_use.copy(), ch.read_labels.copy()))
        return r
    return f
dev = DevicePanel(panel)
Driver._round = wrap("gpu")
got = Driver(panel, HipBackend(dev), prm).run(samples)
Driver._round = wrap("cpu")
ref = Driver(panel, OracleBackend(panel), prm).run(samples)
Driver._round = orig
first = None
for a, b in zip(log["gpu"], log["cpu"]):
    assert a[:3] == b[:3]
    lab, sel = np.array_equal(a[4], b[4]), np.array_equal(np.sort(a[3]), np.sort(b[3]))
    if not (lab and sel) and first is None:
        first = (a, b)
    print("sample %d chain %d it %d: labels equal %s, selection equal %s" % (a[0], a[1], a[2], lab, sel))
if first:
    a, b = first
    only_g, only_c = np.setdiff1d(a[3], b[3]), np.setdiff1d(b[3], a[3])
    print("first divergence: sample %d chain %d it %d; only on the GPU side %s, only on the CPU side %s" % (a[0], a[1], a[2], only_g[:8], only_c[:8]))
    # are the haplotypes that differ duplicates of each other over the whole region?
    hm = panel.hapMatcherR
    for x in only_g[:4]:
        twins = [int(y) for y in only_c if np.array_equal(hm[x - 1], hm[y - 1])]
        same_as = np.nonzero((hm == hm[x - 1]).all(axis=1))[0] + 1
        print("  haplotype %d: identical codes over all grids with %s of the other side's; %d copies in the panel" % (x, twins, len(same_as)))
for i, (g, r) in enumerate(zip(got, ref)):
    print("sample %d: labels identical %s, max|d dosage| %.3e" % (i, np.array_equal(g.read_labels, r.read_labels), np.abs(g.dosage - r.dosage).max()))
dev.close()

"""Developer aid: the first round of one chain of tests/test_pipeline_gpu.py's medium run on both backends -- Gibbs labels, then
the COMPLETE best-haplotype lists of the thin passes: where do the HIP library's lists differ from the oracle's, and by how much
in value?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from quilt_amd.driver import ChainState, Driver, DriverParams, HipBackend, chain_rng
from quilt_amd.native import DevicePanel
from quilt_amd.synth import make_synthetic_panel, make_synthetic_sample
from tests.oracle_backend import OracleBackend

i_sample, i_chain, seed = 2, 1, 5
panel = make_synthetic_panel(K=5000, nSNPs=3200, seed=11)
smp = make_synthetic_sample(panel, seed=1000 + i_sample, n_reads=1000)
prm = DriverParams(nGibbsSamples=3, Ksubset=200, Knew=200, seed=seed)
dev = DevicePanel(panel)
out = {}
for tag, be in (("gpu", HipBackend(dev)), ("cpu", OracleBackend(panel))):
    be.select_on_device = False
    d = Driver(panel, be, prm)
    ch = ChainState(smp, i_sample, i_chain, chain_rng(prm.seed, i_sample, i_chain))
    # the first half of Driver._round: draws + the Gibbs call
    P = d.params
    R = smp.nReads
    ch.which_haps_to_use = np.sort(ch.rng.choice(panel.K, size=P.Ksubset, replace=False) + 1).astype(np.int32)
    H0 = ch.rng.integers(1, 3, size=R).astype(np.int32)
    sr, fr, ss = int(ch.rng.integers(0, 2 ** 63)), int(ch.rng.integers(0, R)), int(ch.rng.integers(0, 2 ** 63))
    res = d._gibbs_with_retry([ch], [smp], [H0], [sr], [fr], [ss], gibbs_initialize_iteratively=True)
    ch.read_labels = res[0]["double_list_of_ending_read_labels"][0][0].astype(np.int32)
    lists = d._full_lists(ch)   # [label][thinned grid] -> 1-based haplotypes, best first
    # and the values: a thin pass again, raw
    per_base = np.repeat(ch.read_labels, np.diff(smp.read_ptr))
    from quilt_amd.driver import make_gl_from_u_bq
    gls = [make_gl_from_u_bq(smp.u[(per_base == l) & (smp.bq != 0)], smp.bq[(per_base == l) & (smp.bq != 0)], panel.nSNPs, P.minGLValue, be.make_gl_bound) for l in (1, 2)]
    _, best = be.fullpass_batch(gls, [0, 0], d.cols, P.K_top_matches)
    out[tag] = (ch.read_labels.copy(), lists, best)
print("Gibbs labels identical:", np.array_equal(out["gpu"][0], out["cpu"][0]))
for l in range(2):
    for j in range(len(out["gpu"][1][l])):
        a, b = out["gpu"][1][l][j], out["cpu"][1][l][j]
        if not np.array_equal(a, b):
            bg, bc = out["gpu"][2][l][j], out["cpu"][2][l][j]
            print("label %d thinned grid %d: gpu %d entries, cpu %d entries" % (l + 1, j, len(a), len(b)))
            vg = dict(zip(np.asarray(bg["top_matches"]).tolist(), np.asarray(bg["top_matches_values"]).tolist()))
            vc = dict(zip(np.asarray(bc["top_matches"]).tolist(), np.asarray(bc["top_matches_values"]).tolist()))
            for k in sorted(set(vg) ^ set(vc))[:6]:
                print("   haplotype %d: gpu %r cpu %r" % (k, vg.get(k), vc.get(k)))
            both = sorted(set(vg) & set(vc))
            if both:
                print("   common entries: max rel diff %.3e; smallest kept value gpu %.17g cpu %.17g" % (
                    max(abs(vg[k] - vc[k]) / max(abs(vc[k]), 1e-300) for k in both), min(vg.values()), min(vc.values())))
dev.close()

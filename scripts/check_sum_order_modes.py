import numpy as np, sys, time
sys.path.insert(0,'.')
from oracle import oracle as O
from quilt_amd.native import DevicePanel
from quilt_amd.synth import make_synthetic_panel, make_synthetic_sample
from tests.util import label_gl, thin_cols
from tests.test_sum_order_gpu import _run_gpu
panel = make_synthetic_panel(K=5000, nSNPs=3200, seed=11, ref_error=1e-3, nGen=100, expRate=1.0)
sample = make_synthetic_sample(panel, seed=1001, n_reads=800)
cols = thin_cols(panel.nGrids)
gl = label_gl(panel, sample, 1, O)
ref = O.haploid_dosage_versus_refs(panel, gl, cols, return_gamma_t=True, return_betaHat_t=True, always_normalize=False, get_best_haps_from_thinned_sites=True)
for mode in (False, True):
    dev = DevicePanel(panel); dev.set_sum_order(mode); dev.set_dosage_precision(64)
    t=time.time()
    got = _run_gpu(dev, gl, cols, return_dosage=True, return_gamma_t=True, return_betaHat_t=True, get_best_haps_from_thinned_sites=True, always_normalize=False)
    print(mode, time.time()-t, [ (k, bool(np.array_equal(got[k], ref[k])), float(np.abs(got[k]-ref[k]).max())) for k in ("c","alphaHat_t","betaHat_t","gamma_t","dosage")])
    dev.close()

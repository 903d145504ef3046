"""Gibbs-level check of the rare + common call in NIPT mode (HIP vs oracle); prints the first differences."""
import numpy as np
from quilt_amd.synth import make_synthetic_panel, make_rare_common, make_synthetic_sample_rare_common
from quilt_amd.native import DevicePanel, DeviceRareCommon
from quilt_amd.gibbs_nipt import rcpp_forwardBackwardGibbsNIPT
from oracle import oracle as O

panel = make_synthetic_panel(K=5000, nSNPs=3200, seed=11)
rc = make_rare_common(panel, 4)
_, s = make_synthetic_sample_rare_common(panel, rc, 2500, n_reads=800, ff=0.15)
rng = np.random.default_rng(3)
which = np.sort(rng.choice(panel.K, 128, replace=False)).astype(np.int32) + 1
R, G = s.nReads, rc.nGrids_all
H0 = rng.choice([1, 2, 3], p=[0.5, 0.4, 0.1], size=R).astype(np.int32)
ru, rb, rr = rng.random(R * 21), rng.random(3 * R), rng.random(3 * R)
dev = DevicePanel(panel)
drc = DeviceRareCommon(dev, rc)
for blk in (False, True):
    ref = O.forwardBackwardGibbsNIPT(panel, s, which, H0, ru, 0, np.zeros(3 * G), ff=0.15, perform_block_gibbs=blk,
                                     disable_read_category_usage=True, rare_common=rc, runif_block=rb, runif_resample=rr,
                                     L_grid=rc.L_grid_all)
    got = rcpp_forwardBackwardGibbsNIPT(dev, s, which, H0, ru, 0, None, ff=0.15, perform_block_gibbs=blk,
                                        disable_read_category_usage=True, rare_common=drc, runif_block=rb, runif_resample=rr)
    print("block", blk, "labels differ:", int((got["H"] != ref["H"]).sum()), "classes differ:", int((got["H_class"] != ref["H_class"]).sum()),
          "hapProbs max|d|:", float(np.abs(got["hapProbs_t"] - ref["hapProbs_t"]).max()),
          "genProbsF max|d|:", float(np.abs(got["genProbsF_t"] - ref["genProbsF_t"]).max()))

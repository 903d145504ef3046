"""Developer aid: print GPU-vs-oracle error summaries for the full-panel pass (needs a GPU)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402
from quilt_amd.native import DevicePanel  # noqa: E402
from quilt_amd.reference_single import Rcpp_haploid_dosage_versus_refs  # noqa: E402
from quilt_amd.synth import make_synthetic_panel, make_synthetic_sample  # noqa: E402
from tests.util import label_gl, thin_cols  # noqa: E402


def run(panel, symbols, seed=1001, n_reads=None):
    dev = DevicePanel(panel, use_eMatDH_special_symbols=symbols)
    sample = make_synthetic_sample(panel, seed=seed, n_reads=n_reads or max(40, panel.nSNPs // 4))
    cols = thin_cols(panel.nGrids)
    K, G, T = panel.K, panel.nGrids, panel.nSNPs
    gl = label_gl(panel, sample, 1, O)
    ref = O.haploid_dosage_versus_refs(panel, gl, cols, return_gamma_t=True, return_betaHat_t=True,
                                       always_normalize=True, get_best_haps_from_thinned_sites=True,
                                       use_eMatDH_special_symbols=symbols)
    out = dict(alphaHat_t=np.zeros((K, G), order="F"), c=np.ones(G), dosage=np.zeros(T),
               gamma_t=np.zeros((K, G), order="F"), betaHat_t=np.zeros((K, G), order="F"),
               best_haps_stuff_list=[None] * int((cols >= 0).sum()))
    Rcpp_haploid_dosage_versus_refs(dev, gl, gammaSmall_cols_to_get=cols, return_dosage=True,
                                    return_gamma_t=True, return_betaHat_t=True,
                                    get_best_haps_from_thinned_sites=True, **out)
    print(f"K={K} G={G} T={T} symbols={symbols} specials={len(panel.eMatDH_special_values_list)}")
    ea = np.abs(out["alphaHat_t"] - ref["alphaHat_t"]).max(axis=0)
    print(" alpha max err per grid (first 12):", np.array2string(ea[:12], precision=2))
    print(" alpha worst grid", int(ea.argmax()), ea.max())
    print(" c rel err per grid (first 12):", np.array2string(np.abs(out["c"] / ref["c"] - 1)[:12], precision=2))
    eb = np.abs(out["betaHat_t"] / np.maximum(ref["betaHat_t"], 1e-300) - 1).max(axis=0)
    print(" beta rel err per grid (last 12):", np.array2string(eb[-12:], precision=2))
    eg = np.abs(out["gamma_t"] - ref["gamma_t"]).max(axis=0)
    print(" gamma max err per grid (last 12):", np.array2string(eg[-12:], precision=2), "worst", int(eg.argmax()), eg.max())
    print(" gamma colsum range", out["gamma_t"].sum(0).min(), out["gamma_t"].sum(0).max())
    ed = np.abs(out["dosage"] - ref["dosage"])
    print(" dosage max err", ed.max(), "at SNP", int(ed.argmax()), "grid", int(ed.argmax()) // 32)
    per_grid = np.array([ed[32 * g:32 * g + 32].max() for g in range(G)])
    print(" dosage err per grid:", np.array2string(per_grid[:16], precision=2))
    print(" best[0] gpu", out["best_haps_stuff_list"][0], "\n best[0] ref", ref["best_haps"][0])
    dev.close()


if __name__ == "__main__":
    p = make_synthetic_panel(K=1000, nSNPs=500, seed=4916, ref_error=0.01, nGen=10, expRate=100, region_bp=5000)
    run(p, False)
    p2 = make_synthetic_panel(K=5000, nSNPs=3200, seed=11, ref_error=1e-3)
    run(p2, False, n_reads=1000)

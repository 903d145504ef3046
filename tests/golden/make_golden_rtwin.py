"""Generates tests/golden/rtwin_*.npz from the R-twin restatement (oracle/rtwin.py, an independent reading of the reference's
R implementations) and cross-checks the C oracle (oracle/*.c, a reading of the reference's C++) against it.

The fixtures hold INPUTS (packed panel rhb_t, transition rates, reads, labels, uniforms) and the R-twin's OUTPUTS; the
tests rebuild the panel tables from rhb_t and compare (a) the C oracle on the CPU and (b) the HIP path on the GPU with
them.  Two independent restatements of two different reference sources agreeing is the pin this repository can offer for
the bulk arithmetic without R.

Run:  python tests/golden/make_golden_rtwin.py     (a few seconds; asserts the cross-check before writing)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import oracle as O          # noqa: E402
from oracle import rtwin                # noqa: E402
from quilt_amd.synth import make_synthetic_panel, make_synthetic_sample   # noqa: E402


def thin_cols(G, every=2):
    cols = np.full(G, -1, dtype=np.int32)
    w = np.arange(1, G, every)
    cols[w] = np.arange(len(w), dtype=np.int32)
    return cols


def fullpass_case(seed, K, T, nMaxDH, n_reads):
    panel = make_synthetic_panel(K=K, nSNPs=T, seed=seed, nMaxDH=nMaxDH, ref_error=1e-3, stress_grids=(1, 3))
    s = make_synthetic_sample(panel, seed=seed + 1, n_reads=n_reads)
    per_base = np.repeat(s.truth_label, np.diff(s.read_ptr))
    sel = (per_base == 1) & (s.bq != 0)
    gl = rtwin.make_gl_from_u_bq(s.u[sel], s.bq[sel], T)
    gl_c = O.make_gl_from_u_bq(s.u[sel], s.bq[sel], T)
    np.testing.assert_allclose(gl, gl_c, rtol=4e-16)   # (numpy's pow and libm's may differ in the last bit)
    cols = thin_cols(panel.nGrids)
    out = {}
    for always in (True, False):
        tw = rtwin.R_haploid_dosage_versus_refs(panel, gl, cols, always_normalize=always)
        # the C++ never normalises emissions in the R twin's sense: compare with normalize_emissions on AND off
        for norm_e in (True, False):
            oc = O.haploid_dosage_versus_refs(panel, gl, cols, return_gamma_t=True, return_betaHat_t=True, always_normalize=always,
                                              normalize_emissions=norm_e, get_best_haps_from_thinned_sites=True)
            d = np.abs(oc["dosage"] - tw["dosage"]).max()
            assert d < 1e-12, ("dosage", always, norm_e, d)
            np.testing.assert_allclose(oc["gamma_t"], tw["gamma_t"], rtol=1e-9, atol=1e-300)
            # c: the C++ keeps alpha scaled by 1 / sigma and records c = 1 / (sigma A); the R twin's textbook recursion has
            # column sum sigma A: the same number.  (With emission normalisation, or the lazy schedule, the two differ by
            # known per-grid factors and only gamma / dosage / lists are comparable.)
            if always and not norm_e:
                np.testing.assert_allclose(oc["c"], tw["c"], rtol=1e-10)
                np.testing.assert_allclose(oc["alphaHat_t"], tw["alphaHat_t"], rtol=1e-9, atol=1e-300)
            assert len(oc["best_haps"]) == len(tw["best_haps"])
            for (oi, ov), (ti, tv) in zip(oc["best_haps"], tw["best_haps"]):
                assert np.array_equal(oi, ti), "best-haplotype lists: C oracle vs R twin"
        out[always] = tw
    tw = out[True]
    return dict(rhb_t=panel.rhb_t, transMatRate_t=panel.transMatRate_t, nSNPs=T, nMaxDH=nMaxDH, ref_error=panel.ref_error,
                gl=gl, cols=cols, dosage=tw["dosage"], gamma_t=tw["gamma_t"], alphaHat_t=tw["alphaHat_t"], c_R=tw["c"],
                best_idx=np.concatenate([b[0] for b in tw["best_haps"]]),
                best_ptr=np.cumsum([0] + [len(b[0]) for b in tw["best_haps"]]).astype(np.int32),
                c_R_lazy=out[False]["c"])


def gibbs_case(seed, K, T, Ks, n_reads, init_iter):
    panel = make_synthetic_panel(K=K, nSNPs=T, seed=seed, nMaxDH=255, ref_error=1e-3, stress_grids=())
    s = make_synthetic_sample(panel, seed=seed + 1, n_reads=n_reads)
    rng = np.random.default_rng(seed + 2)
    which = np.sort(rng.choice(K, Ks, replace=False)).astype(np.int32) + 1
    H0 = rng.integers(1, 3, size=s.nReads).astype(np.int32)
    ru = rng.random(s.nReads * 21)
    fr = int(rng.integers(0, s.nReads))
    tw = rtwin.forwardBackwardGibbsNIPT(panel, s, which, H0, ru, fr, gibbs_initialize_iteratively=init_iter)
    oc = O.forwardBackwardGibbsNIPT(panel, s, which, H0, ru, fr, np.zeros(3 * (panel.nGrids - 1)), perform_block_gibbs=False,
                                    gibbs_initialize_iteratively=init_iter)
    assert np.array_equal(oc["H"], tw["H"]), f"Gibbs labels: C oracle vs R twin ({(oc['H'] != tw['H']).sum()} differ)"
    assert np.array_equal(oc["H_class"], tw["H_class"]), "H_class: C oracle vs R twin"
    for h in range(2):
        np.testing.assert_allclose(oc["alphaHat_t"][h], tw["alphaHat_t"][h], rtol=1e-8, atol=1e-300)
        np.testing.assert_allclose(oc["betaHat_t"][h], tw["betaHat_t"][h], rtol=1e-8, atol=1e-300)
        np.testing.assert_allclose(oc["c"][h], tw["c"][h], rtol=1e-8)
        np.testing.assert_allclose(oc["eMatGrid_t"][h], tw["eMatGrid_t"][h], rtol=1e-8)
    np.testing.assert_allclose(oc["hapProbs_t"][:2], tw["hapProbs_t"][:2], rtol=1e-8, atol=1e-14)
    return dict(rhb_t=panel.rhb_t, transMatRate_t=panel.transMatRate_t, nSNPs=T, ref_error=panel.ref_error,
                read_ptr=s.read_ptr, u=s.u, bq=s.bq, wif=s.wif, which=which, H0=H0, runif_reads=ru, first_read=fr,
                init_iter=int(init_iter), H=tw["H"], H_class=tw["H_class"], hapProbs_t=tw["hapProbs_t"],
                alphaHat_t1=tw["alphaHat_t"][0], betaHat_t2=tw["betaHat_t"][1], c1=tw["c"][0], c2=tw["c"][1])


def shard_case(seed, K, T, Ks, n_reads):
    """Diploid Gibbs call WITH its shard passes (after sweeps 3, 6, 9): R_shard_block_gibbs_resampler restated
    (oracle/rtwin.py) against the C oracle's reading of Rcpp_shard_block_gibbs_resampler."""
    panel = make_synthetic_panel(K=K, nSNPs=T, seed=seed, nMaxDH=255, ref_error=1e-3, stress_grids=())
    s = make_synthetic_sample(panel, seed=seed + 1, n_reads=n_reads)
    rng = np.random.default_rng(seed + 2)
    which = np.sort(rng.choice(K, Ks, replace=False)).astype(np.int32) + 1
    H0 = rng.integers(1, 3, size=s.nReads).astype(np.int32)
    ru = rng.random(s.nReads * 21)
    rs = rng.random(3 * (panel.nGrids - 1))
    fr = int(rng.integers(0, s.nReads))
    tr = []
    tw = rtwin.forwardBackwardGibbsNIPT(panel, s, which, H0, ru, fr, block_gibbs_iterations=(3, 6, 9), runif_shard=rs, trace=tr)
    oc = O.forwardBackwardGibbsNIPT(panel, s, which, H0, ru, fr, rs)
    assert np.array_equal(oc["H"], tw["H"]) and np.array_equal(oc["H_class"], tw["H_class"]), "shard passes: C oracle vs R twin"
    flips = np.stack([t["flip_mode"] for t in tr])
    assert flips.sum() >= 2, "the fixture must contain flips"
    for h in range(2):
        for nm in ("alphaHat_t", "betaHat_t", "c", "eMatGrid_t"):
            np.testing.assert_allclose(oc[nm][h], tw[nm][h], rtol=1e-9, atol=1e-300)
    return dict(rhb_t=panel.rhb_t, transMatRate_t=panel.transMatRate_t, nSNPs=T, ref_error=panel.ref_error,
                read_ptr=s.read_ptr, u=s.u, bq=s.bq, wif=s.wif, which=which, H0=H0, runif_reads=ru, runif_shard=rs, first_read=fr,
                H=tw["H"], H_class=tw["H_class"], flip_mode=flips, p_stay=np.stack([t["p_stay"] for t in tr]),
                hapProbs_t=tw["hapProbs_t"], alphaHat_t1=tw["alphaHat_t"][0], betaHat_t2=tw["betaHat_t"][1],
                eMatGrid_t1=tw["eMatGrid_t"][0], c1=tw["c"][0], c2=tw["c"][1])


def block_case(seed, ff, K, T, Ks, n_reads, q, radius):
    """NIPT Gibbs call WITH block definition and block passes (block_approach = 6): R_define_blocked_snps_using_gamma_on_the_fly,
    R_make_gibbs_considers and R_block_gibbs_resampler restated (oracle/rtwin.py) against the C oracle's reading of the C++."""
    panel = make_synthetic_panel(K=K, nSNPs=T, seed=seed, nMaxDH=255, ref_error=1e-3, stress_grids=())
    s = make_synthetic_sample(panel, seed=seed + 1, n_reads=n_reads, ff=ff)
    rng = np.random.default_rng(seed + 2)
    which = np.sort(rng.choice(K, Ks, replace=False)).astype(np.int32) + 1
    R = s.nReads
    H0 = (rng.choice(3, size=R, p=[0.5, 0.5 - ff / 2, ff / 2]) + 1).astype(np.int32)
    ru = rng.random(R * 21)
    rb, rr = rng.random(3 * R), rng.random(3 * R)
    fr = int(rng.integers(0, R))
    tr = []
    kw = dict(ff=ff, runif_block=rb, runif_resample=rr, block_gibbs_quantile_prob=q, shuffle_bin_radius=radius)
    tw = rtwin.forwardBackwardGibbsNIPT(panel, s, which, H0, ru, fr, block_gibbs_iterations=(3, 6, 9), trace=tr, **kw)
    oc = O.forwardBackwardGibbsNIPT(panel, s, which, H0, ru, fr, np.zeros(3 * (panel.nGrids - 1)), **kw)
    assert np.array_equal(oc["H"], tw["H"]) and np.array_equal(oc["H_class"], tw["H_class"]), "block passes: C oracle vs R twin"
    chosen = np.concatenate([t["ir_chosen"] for t in tr])
    assert all(t["available_rules_agree"] for t in tr), "pick a case on which the R and the C++ availability rules coincide"
    assert len(chosen) >= 12 and len(set(chosen.tolist())) >= 3, "the fixture must hold several blocks and several relabellings"
    for h in range(3):
        for nm in ("alphaHat_t", "betaHat_t", "c", "eMatGrid_t"):
            np.testing.assert_allclose(oc[nm][h], tw[nm][h], rtol=1e-9, atol=1e-300)
    return dict(rhb_t=panel.rhb_t, transMatRate_t=panel.transMatRate_t, nSNPs=T, ref_error=panel.ref_error, L_grid=panel.L_grid,
                read_ptr=s.read_ptr, u=s.u, bq=s.bq, wif=s.wif, which=which, H0=H0, runif_reads=ru, runif_block=rb,
                runif_resample=rr, first_read=fr, ff=ff, quantile_prob=q, shuffle_bin_radius=radius,
                H=tw["H"], H_class=tw["H_class"], ir_chosen=chosen,
                n_blocks=np.array([len(t["ir_chosen"]) for t in tr], dtype=np.int32),
                block_grid_end=np.concatenate([t["grid_end"] for t in tr]),
                hapProbs_t=tw["hapProbs_t"], alphaHat_t1=tw["alphaHat_t"][0], betaHat_t3=tw["betaHat_t"][2],
                eMatGrid_t2=tw["eMatGrid_t"][1], c1=tw["c"][0], c3=tw["c"][2])


if __name__ == "__main__":
    for i, sd in enumerate((51, 52)):
        g = shard_case(seed=sd, K=400, T=320, Ks=48, n_reads=120)
        np.savez_compressed(os.path.join(HERE, f"rtwin_shard_{i}.npz"), **g)
        print(f"shard passes (case {i}): C oracle == R twin (labels, H_class identical; state 1e-9; {int(g['flip_mode'].sum())} flips)")
    for i, sd in enumerate((61, 63)):
        g = block_case(seed=sd, ff=0.2, K=300, T=1920, Ks=32, n_reads=500, q=0.8, radius=2000)
        np.savez_compressed(os.path.join(HERE, f"rtwin_block_{i}.npz"), **g)
        print(f"NIPT block passes (case {i}): C oracle == R twin (labels, H_class identical; state 1e-9; relabellings "
              f"{g['ir_chosen'].tolist()})")
    fp = fullpass_case(seed=31, K=160, T=150, nMaxDH=6, n_reads=60)
    np.savez_compressed(os.path.join(HERE, "rtwin_fullpass.npz"), **fp)
    print("full-panel pass: C oracle == R twin (dosage 1e-12, gamma 1e-9, lists identical)")
    for i, init in enumerate((False, True)):
        g = gibbs_case(seed=41 + i, K=400, T=320, Ks=48, n_reads=120, init_iter=init)
        np.savez_compressed(os.path.join(HERE, f"rtwin_gibbs_{'init' if init else 'labels'}.npz"), **g)
        print(f"Gibbs (gibbs_initialize_iteratively = {init}): C oracle == R twin (labels, H_class identical; state 1e-8)")

"""GPU parity of the final all-SNP ("rare + common") Gibbs call of QUILT2 through the C ABI vs the fp64 CPU oracle
(make_eMatRead_t_rare_common = TRUE: QUILT/R/rare_common.R:325-398; gibbs-small.cpp:270-460 and :711-867).

Bar as for the ordinary call: read labels and H_class IDENTICAL under the same uniforms; fp64 state and
hapProbs / genProbs over all SNPs within 1e-9 relative.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL = 1e-9


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as O
    return O


def _setup(panel, seed, Ks, n_reads, carriers=(0, 4)):
    from quilt_amd.synth import make_rare_common, make_synthetic_sample_rare_common
    rc = make_rare_common(panel, seed, carriers=carriers)
    _, s_all = make_synthetic_sample_rare_common(panel, rc, seed + 1, n_reads=n_reads)
    rng = np.random.default_rng(seed + 17)
    which = np.sort(rng.choice(panel.K, Ks, replace=False)).astype(np.int32) + 1
    H0 = rng.integers(1, 3, size=s_all.nReads).astype(np.int32)
    ru = rng.random(s_all.nReads * 21)
    rs = rng.random(3 * (rc.nGrids_all - 1))
    return rc, s_all, which, H0, ru, rs


def _compare(got, ref):
    assert not got["underflow_problem"] and ref["status"] == 0
    assert np.array_equal(got["H"], ref["H"]), f"{(got['H'] != ref['H']).sum()} labels differ"
    assert np.array_equal(got["H_class"], ref["H_class"])
    for h in range(2):
        np.testing.assert_allclose(got[f"eMatGrid_t{h + 1}"], ref["eMatGrid_t"][h], rtol=RTOL)
        np.testing.assert_allclose(got[f"alphaHat_t{h + 1}"], ref["alphaHat_t"][h], rtol=RTOL, atol=1e-300)
        np.testing.assert_allclose(got[f"betaHat_t{h + 1}"], ref["betaHat_t"][h], rtol=RTOL, atol=1e-300)
        np.testing.assert_allclose(got[f"c{h + 1}"], ref["c"][h], rtol=RTOL)
    for n in ("hapProbs_t", "genProbsM_t"):
        np.testing.assert_allclose(got[n][:2] if n == "hapProbs_t" else got[n], ref[n][:2] if n == "hapProbs_t" else ref[n],
                                   rtol=RTOL, atol=1e-14)


@pytest.mark.parametrize("panel_name,Ks,n_reads,carriers", [("small_panel", 100, 120, (0, 4)),
                                                            ("ragged_panel", 77, 250, (0, 40)),
                                                            ("medium_panel", 600, 1200, (0, 6))])
def test_rare_common_gibbs_matches_oracle(request, oracle, panel_name, Ks, n_reads, carriers):
    from quilt_amd.gibbs_nipt import rcpp_forwardBackwardGibbsNIPT
    from quilt_amd.native import DevicePanel, DeviceRareCommon
    panel = request.getfixturevalue(panel_name)
    dev = DevicePanel(panel)
    rc, s_all, which, H0, ru, rs = _setup(panel, 23, Ks, n_reads, carriers)
    drc = DeviceRareCommon(dev, rc)
    # the reference's arguments for this call: starting labels given, read categories off (impute_one_sample defaults)
    for dis in (True, False):
        ref = oracle.forwardBackwardGibbsNIPT(panel, s_all, which, H0, ru, 0, rs, disable_read_category_usage=dis,
                                              rare_common=rc)
        got = rcpp_forwardBackwardGibbsNIPT(dev, s_all, which, H0, ru, 0, rs, disable_read_category_usage=dis,
                                            return_state=True, rare_common=drc)
        _compare(got, ref)
    drc.close()
    dev.close()


def test_rare_common_batch_and_wave_geometries(medium_panel, oracle, monkeypatch):
    """Several chains in one launch (different samples and haplotype subsets) under every chain geometry."""
    from quilt_amd.gibbs_nipt import forwardBackwardGibbsNIPT_batch
    from quilt_amd.native import DevicePanel, DeviceRareCommon
    from quilt_amd.synth import make_rare_common, make_synthetic_sample_rare_common
    panel = medium_panel
    rc = make_rare_common(panel, 5)
    dev = DevicePanel(panel)
    drc = DeviceRareCommon(dev, rc)
    rng = np.random.default_rng(3)
    G = rc.nGrids_all
    samples, whichs, H0s, rus, rss, refs = [], [], [], [], [], []
    for c in range(3):
        _, s_all = make_synthetic_sample_rare_common(panel, rc, 100 + c, n_reads=400 + 50 * c)
        which = np.sort(rng.choice(panel.K, 600, replace=False)).astype(np.int32) + 1
        H0 = rng.integers(1, 3, size=s_all.nReads).astype(np.int32)
        ru, rs = rng.random(s_all.nReads * 21), rng.random(3 * (G - 1))
        samples.append(s_all); whichs.append(which); H0s.append(H0); rus.append(ru); rss.append(rs)
        refs.append(oracle.forwardBackwardGibbsNIPT(panel, s_all, which, H0, ru, 0, rs, disable_read_category_usage=True,
                                                    rare_common=rc))
    for nw in ("1", "2", "5"):
        monkeypatch.setenv("QA_GIBBS_NW", nw)
        got = forwardBackwardGibbsNIPT_batch(dev, samples, whichs, H0s, rus, [0, 0, 0], rss,
                                             disable_read_category_usage=True, rare_common=drc)
        for g, r in zip(got, refs):
            assert np.array_equal(g["H"], r["H"])
            np.testing.assert_allclose(g["hapProbs_t"][:2], r["hapProbs_t"][:2], rtol=RTOL, atol=1e-14)
            np.testing.assert_allclose(g["genProbsM_t"], r["genProbsM_t"], rtol=RTOL, atol=1e-14)
    drc.close()
    dev.close()


def test_rare_common_rejects_bad_tables(small_panel):
    from quilt_amd.native import DevicePanel, DeviceRareCommon, QuiltAmdError
    from quilt_amd.synth import make_rare_common
    import dataclasses
    dev = DevicePanel(small_panel)
    rc = make_rare_common(small_panel, 9)
    bad = dataclasses.replace(rc, snp_is_common=np.ones_like(rc.snp_is_common))
    with pytest.raises(QuiltAmdError):
        DeviceRareCommon(dev, bad)
    dev.close()


def test_rare_common_pipeline_matches_oracle(medium_panel):
    """The whole driver with impute_rare_common on the HIP backend vs on the oracle: consensus labels identical, all-SNP
    dosages within the fp32 rounding of the dosage passes that seed the all-SNP starting labels."""
    from quilt_amd.driver import Driver, DriverParams, HipBackend
    from quilt_amd.native import DevicePanel, DeviceRareCommon
    from quilt_amd.synth import make_rare_common, make_synthetic_sample_rare_common
    from tests.oracle_backend import OracleBackend
    from tests.util import r2
    panel = medium_panel
    rc = make_rare_common(panel, 4)
    samples = [make_synthetic_sample_rare_common(panel, rc, 2000 + i, n_reads=1000)[0] for i in range(3)]
    prm = DriverParams(nGibbsSamples=3, Ksubset=200, Knew=200, seed=5, impute_rare_common=True)
    dev = DevicePanel(panel)
    drc = DeviceRareCommon(dev, rc)
    got = Driver(panel, HipBackend(dev, drc), prm, rare_common=rc).run(samples)
    ref = Driver(panel, OracleBackend(panel, rc), prm, rare_common=rc).run(samples)
    for i, (g, r) in enumerate(zip(got, ref)):
        assert g.nDosage == r.nDosage == 3 and g.dosage.shape == (rc.nSNPs_all,)
        assert np.array_equal(g.read_labels, r.read_labels)
        np.testing.assert_allclose(g.gp_t.sum(axis=0), 1.0, atol=2e-3)
        print(f"sample {i}: all-SNP r2(gpu, oracle) = {r2(g.dosage, r.dosage):.6f}, max|d| = {np.abs(g.dosage - r.dosage).max():.2e}")
        assert r2(g.dosage, r.dosage) >= 0.999
        truth = samples[i].all_snp.truth_haps.sum(axis=0)
        assert abs(r2(g.dosage, truth) - r2(r.dosage, truth)) < 0.02
    drc.close()
    dev.close()


def test_rare_common_nipt_pipeline_matches_oracle(medium_panel):
    """impute_rare_common with method = "nipt": the all-SNP call with three labels and its block Gibbs on the all-SNP
    grid, fetal fractions differing between samples; HIP backend vs oracle backend.

    (With impute_rare_common EVERY chain's final haplotype selection is used.  Panel haplotypes that coincide over the
    region have posteriors equal up to the last bits, and which of two such candidates ranks first is then decided by
    rounding noise that differs between any two implementations -- R's included; when that happens the random subsample
    of candidates differs and so does that one chain.  The seeds below do not hit such a near-tie)"""
    from quilt_amd.driver import Driver, DriverParams, HipBackend
    from quilt_amd.native import DevicePanel, DeviceRareCommon
    from quilt_amd.synth import make_rare_common, make_synthetic_sample_rare_common
    from tests.oracle_backend import OracleBackend
    from tests.util import r2
    panel = medium_panel
    rc = make_rare_common(panel, 4)
    samples = [make_synthetic_sample_rare_common(panel, rc, 2500 + i, n_reads=800, ff=0.15 + 0.1 * i)[0] for i in range(2)]
    prm = DriverParams(nGibbsSamples=2, Ksubset=128, Knew=128, seed=6, impute_rare_common=True, method="nipt")
    dev = DevicePanel(panel)
    drc = DeviceRareCommon(dev, rc)
    got = Driver(panel, HipBackend(dev, drc), prm, rare_common=rc).run(samples)
    ref = Driver(panel, OracleBackend(panel, rc), prm, rare_common=rc).run(samples)
    for g, r in zip(got, ref):
        assert np.array_equal(g.read_labels, r.read_labels)
        assert g.dosage.shape == (rc.nSNPs_all,) and g.phasing_haps.shape == (rc.nSNPs_all, 3)
        print(f"r2 mother {r2(g.dosage, r.dosage):.6f} fetus {r2(g.fet_dosage, r.fet_dosage):.6f} max|d| "
              f"{np.abs(g.dosage - r.dosage).max():.2e} {np.abs(g.fet_dosage - r.fet_dosage).max():.2e}")
        assert r2(g.dosage, r.dosage) >= 0.999 and r2(g.fet_dosage, r.fet_dosage) >= 0.999
        assert np.abs(g.dosage - r.dosage).max() <= 1e-4 and np.abs(g.fet_dosage - r.fet_dosage).max() <= 1e-4
    drc.close()
    dev.close()


def test_rare_common_degenerate_tables(small_panel, oracle):
    """No rare SNP at all (every SNP common: the call must equal the ordinary one), and rare SNPs nobody carries."""
    from quilt_amd.gibbs_nipt import rcpp_forwardBackwardGibbsNIPT
    from quilt_amd.native import DevicePanel, DeviceRareCommon
    from quilt_amd.synth import make_rare_common, make_synthetic_sample, make_synthetic_sample_rare_common
    panel = small_panel
    dev = DevicePanel(panel)
    rng = np.random.default_rng(4)
    which = np.sort(rng.choice(panel.K, 90, replace=False)).astype(np.int32) + 1
    # (a) all SNPs common
    rc0 = make_rare_common(panel, 1, n_rare=0)
    assert rc0.nSNPs_all == panel.nSNPs
    s = make_synthetic_sample(panel, seed=8, n_reads=150)
    R = s.nReads
    H0 = rng.integers(1, 3, size=R).astype(np.int32)
    ru, rs = rng.random(R * 21), rng.random(3 * (panel.nGrids - 1))
    d0 = DeviceRareCommon(dev, rc0)
    a = rcpp_forwardBackwardGibbsNIPT(dev, s, which, H0, ru, 0, rs, rare_common=d0)
    b = rcpp_forwardBackwardGibbsNIPT(dev, s, which, H0, ru, 0, rs)
    assert np.array_equal(a["H"], b["H"])
    np.testing.assert_allclose(a["hapProbs_t"], b["hapProbs_t"], rtol=RTOL, atol=1e-14)
    d0.close()
    # (b) rare SNPs without carriers: common factors that rescaling removes; hapProbs = ref_error there
    rc1 = make_rare_common(panel, 2, carriers=(0, 0))
    _, s_all = make_synthetic_sample_rare_common(panel, rc1, 9, n_reads=150)
    R = s_all.nReads
    H0 = rng.integers(1, 3, size=R).astype(np.int32)
    ru, rs = rng.random(R * 21), rng.random(3 * (rc1.nGrids_all - 1))
    d1 = DeviceRareCommon(dev, rc1)
    got = rcpp_forwardBackwardGibbsNIPT(dev, s_all, which, H0, ru, 0, rs, rare_common=d1, disable_read_category_usage=True)
    ref = oracle.forwardBackwardGibbsNIPT(panel, s_all, which, H0, ru, 0, rs, rare_common=rc1, disable_read_category_usage=True)
    assert np.array_equal(got["H"], ref["H"])
    np.testing.assert_allclose(got["hapProbs_t"][:2], ref["hapProbs_t"][:2], rtol=RTOL, atol=1e-14)
    assert np.allclose(got["hapProbs_t"][:2][:, rc1.snp_is_common == 0], panel.ref_error)
    d1.close()
    dev.close()

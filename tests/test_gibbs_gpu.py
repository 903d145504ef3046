"""GPU parity: HIP Gibbs read-label sampler (through the C ABI) vs the fp64 CPU oracle.

Bar: sampled read labels and H_class IDENTICAL under the same uniforms; fp64 state (alpha, beta,
eMatGrid, c) and hapProbs / genProbs within 1e-9 relative (only the order of the Ks-wide sums differs).
"""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL = 1e-9


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as O
    return O


def _setup(panel, seed, Ks, n_reads, mode="short"):
    from quilt_amd.synth import make_synthetic_sample
    s = make_synthetic_sample(panel, seed=seed, n_reads=n_reads, mode=mode)
    rng = np.random.default_rng(seed + 17)
    which = np.sort(rng.choice(panel.K, Ks, replace=False)).astype(np.int32) + 1
    H0 = rng.integers(1, 3, size=s.nReads).astype(np.int32)
    ru = rng.random(s.nReads * 21)
    rs = rng.random(3 * (panel.nGrids - 1))
    fr = int(rng.integers(0, s.nReads))
    return s, which, H0, ru, rs, fr


def _compare(got, ref, Ks):
    assert not got["underflow_problem"] and ref["status"] == 0
    assert np.array_equal(got["H"], ref["H"]), f"{(got['H'] != ref['H']).sum()} labels differ"
    assert np.array_equal(got["H_class"], ref["H_class"])
    for h in range(2):
        np.testing.assert_allclose(got[f"eMatGrid_t{h + 1}"], ref["eMatGrid_t"][h], rtol=RTOL)
        np.testing.assert_allclose(got[f"alphaHat_t{h + 1}"], ref["alphaHat_t"][h], rtol=RTOL, atol=1e-300)
        np.testing.assert_allclose(got[f"betaHat_t{h + 1}"], ref["betaHat_t"][h], rtol=RTOL, atol=1e-300)
        np.testing.assert_allclose(got[f"c{h + 1}"], ref["c"][h], rtol=RTOL)
    np.testing.assert_allclose(got["hapProbs_t"], ref["hapProbs_t"], rtol=RTOL, atol=1e-14)
    np.testing.assert_allclose(got["genProbsM_t"], ref["genProbsM_t"], rtol=RTOL, atol=1e-14)
    np.testing.assert_allclose(got["genProbsF_t"], ref["genProbsF_t"], rtol=RTOL, atol=1e-14)


@pytest.mark.parametrize("init_iter", [False, True])
@pytest.mark.parametrize("panel_name,Ks,n_reads", [("small_panel", 100, 100), ("ragged_panel", 77, 250),
                                                   ("medium_panel", 600, 1500)])
def test_gibbs_matches_oracle(request, oracle, panel_name, Ks, n_reads, init_iter):
    from quilt_amd.gibbs_nipt import rcpp_forwardBackwardGibbsNIPT
    from quilt_amd.native import DevicePanel
    panel = request.getfixturevalue(panel_name)
    dev = DevicePanel(panel)
    s, which, H0, ru, rs, fr = _setup(panel, 11, Ks, n_reads)
    ref = oracle.forwardBackwardGibbsNIPT(panel, s, which, H0, ru, fr, rs, gibbs_initialize_iteratively=init_iter)
    got = rcpp_forwardBackwardGibbsNIPT(dev, s, which, H0, ru, fr, rs, gibbs_initialize_iteratively=init_iter,
                                        return_state=True)
    _compare(got, ref, Ks)
    dev.close()


def test_gibbs_ont_reads_and_category_switch(medium_panel, oracle):
    """Long noisy reads (ONT-like) and disable_read_category_usage (rare/common default)."""
    from quilt_amd.gibbs_nipt import rcpp_forwardBackwardGibbsNIPT
    from quilt_amd.native import DevicePanel
    panel = medium_panel
    dev = DevicePanel(panel)
    s, which, H0, ru, rs, fr = _setup(panel, 5, 200, 40, mode="ont")
    for dis in (False, True):
        ref = oracle.forwardBackwardGibbsNIPT(panel, s, which, H0, ru, fr, rs, disable_read_category_usage=dis)
        got = rcpp_forwardBackwardGibbsNIPT(dev, s, which, H0, ru, fr, rs, disable_read_category_usage=dis,
                                            return_state=True)
        _compare(got, ref, 200)


def test_gibbs_batch_of_chains(small_panel, oracle):
    """Several chains (different samples, haplotype subsets, uniforms) in one launch == one at a time."""
    from quilt_amd.gibbs_nipt import forwardBackwardGibbsNIPT_batch
    from quilt_amd.native import DevicePanel
    panel = small_panel
    dev = DevicePanel(panel)
    setups = [_setup(panel, 100 + i, 64, 60 + 13 * i) for i in range(5)]
    got = forwardBackwardGibbsNIPT_batch(dev, [x[0] for x in setups], [x[1] for x in setups], [x[2] for x in setups],
                                         [x[3] for x in setups], [x[5] for x in setups], [x[4] for x in setups])
    for g, (s, which, H0, ru, rs, fr) in zip(got, setups):
        ref = oracle.forwardBackwardGibbsNIPT(panel, s, which, H0, ru, fr, rs)
        assert np.array_equal(g["H"], ref["H"])
        np.testing.assert_allclose(g["hapProbs_t"], ref["hapProbs_t"], rtol=RTOL, atol=1e-14)


def test_gibbs_no_block_no_burn(small_panel, oracle):
    """n_its = 0: initialisation only leaves the labels untouched."""
    from quilt_amd.gibbs_nipt import rcpp_forwardBackwardGibbsNIPT
    from quilt_amd.native import DevicePanel
    panel = small_panel
    dev = DevicePanel(panel)
    s, which, H0, ru, rs, fr = _setup(panel, 3, 50, 80)
    got = rcpp_forwardBackwardGibbsNIPT(dev, s, which, H0, ru, fr, rs, n_gibbs_burn_in_its=0, n_gibbs_sample_its=0,
                                        perform_block_gibbs=False, return_state=True)
    ref = oracle.forwardBackwardGibbsNIPT(panel, s, which, H0, ru, fr, rs, n_gibbs_burn_in_its=0,
                                          n_gibbs_sample_its=0, perform_block_gibbs=False)
    assert np.array_equal(got["H"], H0)
    for h in range(2):
        np.testing.assert_allclose(got[f"alphaHat_t{h + 1}"], ref["alphaHat_t"][h], rtol=RTOL, atol=1e-300)
        np.testing.assert_allclose(got[f"betaHat_t{h + 1}"], ref["betaHat_t"][h], rtol=RTOL, atol=1e-300)


def test_shard_uniforms_are_checked_when_read_and_ignored_when_not(small_panel, oracle):
    """runif_shard is read only when a shard pass draws from it: with the passes switched off a one-element array is fine (the
    library read 45 doubles of it anyway until round 6 -- found by AddressSanitizer on the host code, scripts/asan_host.sh); with
    the passes on, the wrapper refuses an array shorter than block iterations x (nGrids - 1)."""
    from quilt_amd.gibbs_nipt import rcpp_forwardBackwardGibbsNIPT
    from quilt_amd.native import DevicePanel
    panel = small_panel
    dev = DevicePanel(panel)
    s, which, H0, ru, rs, fr = _setup(panel, 3, 50, 80)
    short = np.zeros(1)
    got = rcpp_forwardBackwardGibbsNIPT(dev, s, which, H0, ru, fr, short, n_gibbs_burn_in_its=2, n_gibbs_sample_its=1, perform_block_gibbs=False)
    ref = oracle.forwardBackwardGibbsNIPT(panel, s, which, H0, ru, fr, rs, n_gibbs_burn_in_its=2, n_gibbs_sample_its=1, perform_block_gibbs=False)
    assert np.array_equal(got["H"], ref["H"])
    with pytest.raises(ValueError, match="runif_shard"):
        rcpp_forwardBackwardGibbsNIPT(dev, s, which, H0, ru, fr, short, n_gibbs_burn_in_its=2, n_gibbs_sample_its=1)
    dev.close()


def test_gibbs_seeded_streams_match_explicit_uniforms(small_panel, oracle):
    """The device counter-based stream == quilt_amd.rng.stream_uniform fed to the oracle as arrays."""
    from quilt_amd.gibbs_nipt import forwardBackwardGibbsNIPT_batch
    from quilt_amd.native import DevicePanel
    from quilt_amd.rng import stream_uniform
    panel = small_panel
    dev = DevicePanel(panel)
    s, which, H0, _, _, fr = _setup(panel, 21, 64, 120)
    sr, ss = 0x1234567890ABCDEF, 0x0FEDCBA987654321
    got = forwardBackwardGibbsNIPT_batch(dev, [s], [which], [H0], None, [fr], None, seed_reads=[sr], seed_shard=[ss])[0]
    ru = stream_uniform(sr, s.nReads * 21)
    rs = stream_uniform(ss, 3 * (panel.nGrids - 1))
    ref = oracle.forwardBackwardGibbsNIPT(panel, s, which, H0, ru, fr, rs)
    assert np.array_equal(got["H"], ref["H"])
    np.testing.assert_allclose(got["hapProbs_t"], ref["hapProbs_t"], rtol=RTOL, atol=1e-14)


@pytest.mark.parametrize("nw", ["1", "2", "5", "10"])
def test_gibbs_every_chain_geometry(medium_panel, oracle, nw, monkeypatch):
    """Ksubset = 600 can run as 1, 2, 5 or 10 wavefronts per chain; all must give the oracle's labels."""
    from quilt_amd.gibbs_nipt import rcpp_forwardBackwardGibbsNIPT
    from quilt_amd.native import DevicePanel
    monkeypatch.setenv("QA_GIBBS_NW", nw)
    panel = medium_panel
    dev = DevicePanel(panel)
    s, which, H0, ru, rs, fr = _setup(panel, 31, 600, 900)
    for init_iter in (False, True):
        ref = oracle.forwardBackwardGibbsNIPT(panel, s, which, H0, ru, fr, rs, gibbs_initialize_iteratively=init_iter)
        got = rcpp_forwardBackwardGibbsNIPT(dev, s, which, H0, ru, fr, rs, gibbs_initialize_iteratively=init_iter,
                                            return_state=True)
        _compare(got, ref, 600)


def test_gibbs_two_chains_per_simd_build(medium_panel, oracle, monkeypatch):
    """A launch with more chains than the device has SIMDs runs the 256-register build of the sampler (two chains per SIMD,
    no columns fetched a grid ahead); QA_GIBBS_LEAN = 1 forces it for a small launch: the oracle's labels, classes and state,
    with and without iterative initialisation, shard passes included."""
    from quilt_amd.gibbs_nipt import rcpp_forwardBackwardGibbsNIPT
    from quilt_amd.native import DevicePanel
    monkeypatch.setenv("QA_GIBBS_LEAN", "1")
    panel = medium_panel
    dev = DevicePanel(panel)
    s, which, H0, ru, rs, fr = _setup(panel, 37, 600, 900)
    for init_iter in (False, True):
        ref = oracle.forwardBackwardGibbsNIPT(panel, s, which, H0, ru, fr, rs, gibbs_initialize_iteratively=init_iter)
        got = rcpp_forwardBackwardGibbsNIPT(dev, s, which, H0, ru, fr, rs, gibbs_initialize_iteratively=init_iter,
                                            return_state=True)
        _compare(got, ref, 600)
    dev.close()


def test_gibbs_edge_inputs(small_panel, oracle):
    """Ragged and degenerate chains in one launch: a single read; all reads in one grid; reads of 6-8 SNPs (dense
    emission columns next to the compact table form); a read whose bases all have zero quality."""
    import copy
    from quilt_amd.gibbs_nipt import forwardBackwardGibbsNIPT_batch
    from quilt_amd.native import DevicePanel
    panel = small_panel
    dev = DevicePanel(panel)
    setups = []
    # (a) one read
    setups.append(_setup(panel, 301, 64, 1))
    # (b) every read in the same grid
    s, which, H0, ru, rs, fr = _setup(panel, 302, 64, 30)
    s = copy.deepcopy(s)
    g0 = int(s.wif[0])
    s.wif[:] = g0
    s.u[:] = 32 * g0 + (s.u % 32)
    setups.append((s, which, H0, ru, rs, fr))
    # (c) many long-ish reads: more than 5 informative SNPs -> dense columns
    s, which, H0, ru, rs, fr = _setup(panel, 303, 64, 120)
    assert (np.diff(s.read_ptr) > 5).any() and (np.diff(s.read_ptr) <= 5).any()
    setups.append((s, which, H0, ru, rs, fr))
    # (d) first read without any base quality (factor 1 everywhere: gibbs-small.cpp:139-181 carry-over from nothing)
    s, which, H0, ru, rs, fr = _setup(panel, 304, 64, 40)
    s = copy.deepcopy(s)
    s.bq[s.read_ptr[0]:s.read_ptr[1]] = 0
    setups.append((s, which, H0, ru, rs, fr))
    got = forwardBackwardGibbsNIPT_batch(dev, [x[0] for x in setups], [x[1] for x in setups], [x[2] for x in setups],
                                         [x[3] for x in setups], [x[5] for x in setups], [x[4] for x in setups])
    for g, (s, which, H0, ru, rs, fr) in zip(got, setups):
        ref = oracle.forwardBackwardGibbsNIPT(panel, s, which, H0, ru, fr, rs)
        assert np.array_equal(g["H"], ref["H"])
        assert bool(g["underflow_problem"]) == bool(ref["status"] == 1)
        np.testing.assert_allclose(g["hapProbs_t"], ref["hapProbs_t"], rtol=RTOL, atol=1e-14)
    dev.close()


def test_gibbs_underflow_is_reported(small_panel, oracle):
    """Thousands of reads piled on one grid: the grid's emission product underflows for every haplotype, the forward
    sum is 0 and c infinite -- status QA_UNDERFLOW / underflow_problem on both sides (the driver then retries with a
    smaller maxDifferenceBetweenReads, functions.R:2704-2715)."""
    import copy
    from quilt_amd.gibbs_nipt import rcpp_forwardBackwardGibbsNIPT
    from quilt_amd.native import DevicePanel
    panel = small_panel
    dev = DevicePanel(panel)
    s, which, H0, ru, rs, fr = _setup(panel, 78, 64, 4000)
    s = copy.deepcopy(s)
    g0 = int(s.wif[len(s.wif) // 2])
    s.wif[:] = g0
    s.u[:] = 32 * g0 + (s.u % 32)
    ref = oracle.forwardBackwardGibbsNIPT(panel, s, which, H0, ru, fr, rs)
    got = rcpp_forwardBackwardGibbsNIPT(dev, s, which, H0, ru, fr, rs)
    assert ref["status"] == 1, "the test input is meant to underflow"
    assert got["underflow_problem"]
    dev.close()


@pytest.mark.parametrize("init_iter", [False, True])
@pytest.mark.parametrize("panel_name,Ks,n_reads", [("small_panel", 100, 150), ("medium_panel", 600, 900)])
def test_nipt_three_label_sampler(request, oracle, panel_name, Ks, n_reads, init_iter):
    """NIPT mode (BASELINE configs[4] in small): ff > 0, three read labels, the sampler without the block resampler.
    Labels and classes identical, hapProbs / genProbs (all three haplotypes) to 1e-9."""
    from quilt_amd.gibbs_nipt import rcpp_forwardBackwardGibbsNIPT
    from quilt_amd.native import DevicePanel, QuiltAmdError
    from quilt_amd.synth import make_synthetic_sample
    panel = request.getfixturevalue(panel_name)
    dev = DevicePanel(panel)
    ff = 0.2
    s = make_synthetic_sample(panel, seed=21, n_reads=n_reads, ff=ff)
    rng = np.random.default_rng(5)
    which = np.sort(rng.choice(panel.K, Ks, replace=False)).astype(np.int32) + 1
    H0 = rng.integers(1, 4, size=s.nReads).astype(np.int32)
    ru = rng.random(s.nReads * 21)
    rs = rng.random(3 * (panel.nGrids - 1))
    fr = int(rng.integers(0, s.nReads))
    ref = oracle.forwardBackwardGibbsNIPT(panel, s, which, H0, ru, fr, rs, ff=ff, perform_block_gibbs=False,
                                          gibbs_initialize_iteratively=init_iter)
    got = rcpp_forwardBackwardGibbsNIPT(dev, s, which, H0, ru, fr, rs, ff=ff, perform_block_gibbs=False,
                                        gibbs_initialize_iteratively=init_iter)
    assert ref["status"] == 0 and not got["underflow_problem"]
    assert set(np.unique(ref["H"])) <= {1, 2, 3} and (ref["H"] == 3).any()
    assert np.array_equal(got["H"], ref["H"]), f"{(got['H'] != ref['H']).sum()} labels differ"
    assert np.array_equal(got["H_class"], ref["H_class"])
    np.testing.assert_allclose(got["hapProbs_t"], ref["hapProbs_t"], rtol=RTOL, atol=1e-14)
    np.testing.assert_allclose(got["genProbsM_t"], ref["genProbsM_t"], rtol=RTOL, atol=1e-14)
    np.testing.assert_allclose(got["genProbsF_t"], ref["genProbsF_t"], rtol=RTOL, atol=1e-14)
    with pytest.raises((QuiltAmdError, ValueError)):   # block passes without their uniforms: refused, not silently skipped
        rcpp_forwardBackwardGibbsNIPT(dev, s, which, H0, ru, fr, None, ff=ff)
    dev.close()


@pytest.mark.parametrize("init_iter", [False, True])
@pytest.mark.parametrize("panel_name,Ks,n_reads", [("small_panel", 100, 150), ("ragged_panel", 77, 400),
                                                   ("medium_panel", 600, 900)])
def test_nipt_block_gibbs(request, oracle, panel_name, Ks, n_reads, init_iter):
    """NIPT with the block resampler (gibbs-nipt-block.cpp: block definition from the switch rate, six relabellings per
    block, labels re-drawn from their classes), as the reference runs it by default (block passes after sweeps 3, 6, 9):
    labels and classes identical to the oracle under the same uniforms, hapProbs / genProbs to 1e-9."""
    from quilt_amd.gibbs_nipt import rcpp_forwardBackwardGibbsNIPT
    from quilt_amd.native import DevicePanel
    from quilt_amd.synth import make_synthetic_sample
    panel = request.getfixturevalue(panel_name)
    dev = DevicePanel(panel)
    ff = 0.2
    s = make_synthetic_sample(panel, seed=21, n_reads=n_reads, ff=ff)
    rng = np.random.default_rng(5)
    which = np.sort(rng.choice(panel.K, Ks, replace=False)).astype(np.int32) + 1
    R = s.nReads
    H0 = rng.choice([1, 2, 3], p=[0.5, 0.4, 0.1], size=R).astype(np.int32)
    ru = rng.random(R * 21)
    rb, rr = rng.random(3 * R), rng.random(3 * R)
    fr = int(rng.integers(0, R))
    for n_burn in (20, 9):   # 9: the call ends right after the last block pass
        ref = oracle.forwardBackwardGibbsNIPT(panel, s, which, H0, ru, fr, np.zeros(3 * panel.nGrids), ff=ff,
                                              gibbs_initialize_iteratively=init_iter, n_gibbs_burn_in_its=n_burn,
                                              runif_block=rb, runif_resample=rr)
        got = rcpp_forwardBackwardGibbsNIPT(dev, s, which, H0, ru, fr, None, ff=ff, gibbs_initialize_iteratively=init_iter,
                                            n_gibbs_burn_in_its=n_burn, runif_block=rb, runif_resample=rr)
        assert ref["status"] == 0 and not got["underflow_problem"]
        assert np.array_equal(got["H"], ref["H"]), f"{(got['H'] != ref['H']).sum()} labels differ (n_burn = {n_burn})"
        assert np.array_equal(got["H_class"], ref["H_class"])
        np.testing.assert_allclose(got["hapProbs_t"], ref["hapProbs_t"], rtol=RTOL, atol=1e-14)
        np.testing.assert_allclose(got["genProbsM_t"], ref["genProbsM_t"], rtol=RTOL, atol=1e-14)
        np.testing.assert_allclose(got["genProbsF_t"], ref["genProbsF_t"], rtol=RTOL, atol=1e-14)
    # the block passes do something: without them the labels differ
    plain = rcpp_forwardBackwardGibbsNIPT(dev, s, which, H0, ru, fr, None, ff=ff, perform_block_gibbs=False,
                                          gibbs_initialize_iteratively=init_iter)
    assert (plain["H"] != got["H"]).any()
    dev.close()


def test_nipt_block_gibbs_seeded_batch(medium_panel, oracle):
    """Several NIPT chains in one launch set with the counter-based uniform streams."""
    from quilt_amd.gibbs_nipt import forwardBackwardGibbsNIPT_batch
    from quilt_amd.native import DevicePanel
    from quilt_amd.rng import stream_uniform
    from quilt_amd.synth import make_synthetic_sample
    panel = medium_panel
    dev = DevicePanel(panel)
    rng = np.random.default_rng(9)
    samples, whichs, H0s, srs, sss, refs = [], [], [], [], [], []
    ffs = [0.1, 0.15, 0.25]   # one launch set, a fetal fraction per chain
    for c in range(3):
        ff = ffs[c]
        s = make_synthetic_sample(panel, seed=40 + c, n_reads=500 + 100 * c, ff=ff)
        which = np.sort(rng.choice(panel.K, 600, replace=False)).astype(np.int32) + 1
        H0 = rng.choice([1, 2, 3], p=[0.5, 0.4, 0.1], size=s.nReads).astype(np.int32)
        sr, ss = int(rng.integers(0, 2 ** 62)), int(rng.integers(0, 2 ** 62))
        R = s.nReads
        ru = stream_uniform(sr, R * 21)
        blk = stream_uniform(ss, 3 * 2 * R).reshape(3, 2, R)
        refs.append(oracle.forwardBackwardGibbsNIPT(panel, s, which, H0, ru, 0, np.zeros(3 * panel.nGrids), ff=ff,
                                                    runif_block=blk[:, 0, :].copy(), runif_resample=blk[:, 1, :].copy()))
        samples.append(s); whichs.append(which); H0s.append(H0); srs.append(sr); sss.append(ss)
    got = forwardBackwardGibbsNIPT_batch(dev, samples, whichs, H0s, None, [0, 0, 0], None, ff=ffs, seed_reads=srs,
                                         seed_shard=sss)
    for g, r in zip(got, refs):
        assert np.array_equal(g["H"], r["H"]) and np.array_equal(g["H_class"], r["H_class"])
        np.testing.assert_allclose(g["hapProbs_t"], r["hapProbs_t"], rtol=RTOL, atol=1e-14)
    dev.close()
    dev.close()



@pytest.mark.parametrize("nw", ["1", "2"])
def test_nipt_every_chain_geometry(medium_panel, oracle, nw, monkeypatch):
    """Ksubset = 600 in NIPT mode: the sampler with one wave per chain (10 rows per lane) or two; the block kernel always
    runs two waves and reads the compact emissions in whichever layout the sampler's geometry produced."""
    from quilt_amd.gibbs_nipt import rcpp_forwardBackwardGibbsNIPT
    from quilt_amd.native import DevicePanel
    from quilt_amd.synth import make_synthetic_sample
    monkeypatch.setenv("QA_GIBBS_NW", nw)
    panel = medium_panel
    dev = DevicePanel(panel)
    ff = 0.2
    s = make_synthetic_sample(panel, seed=33, n_reads=700, ff=ff)
    rng = np.random.default_rng(8)
    which = np.sort(rng.choice(panel.K, 600, replace=False)).astype(np.int32) + 1
    R = s.nReads
    H0 = rng.choice([1, 2, 3], p=[0.5, 0.4, 0.1], size=R).astype(np.int32)
    ru, rb, rr = rng.random(R * 21), rng.random(3 * R), rng.random(3 * R)
    ref = oracle.forwardBackwardGibbsNIPT(panel, s, which, H0, ru, 5, np.zeros(3 * panel.nGrids), ff=ff,
                                          gibbs_initialize_iteratively=True, runif_block=rb, runif_resample=rr)
    got = rcpp_forwardBackwardGibbsNIPT(dev, s, which, H0, ru, 5, None, ff=ff, gibbs_initialize_iteratively=True,
                                        runif_block=rb, runif_resample=rr)
    assert np.array_equal(got["H"], ref["H"]) and np.array_equal(got["H_class"], ref["H_class"])
    np.testing.assert_allclose(got["hapProbs_t"], ref["hapProbs_t"], rtol=RTOL, atol=1e-14)
    dev.close()



@pytest.mark.parametrize("init_iter", [False, True])
def test_nipt_two_chains_per_simd_build(medium_panel, oracle, init_iter, monkeypatch):
    """The three-label sampler's 256-register build (two chains per SIMD: eMatGrid's three columns in LDS during a grid's
    reads, alpha * beta formed in beta's registers, no columns fetched a grid ahead; not the default -- it measured no faster
    than two launches at one chain per SIMD -- selected by QA_GIBBS3_LEAN = 1): labels and classes identical to the oracle,
    block passes included, with both initialisations."""
    from quilt_amd.gibbs_nipt import forwardBackwardGibbsNIPT_batch
    from quilt_amd.native import DevicePanel
    from quilt_amd.synth import make_synthetic_sample
    monkeypatch.setenv("QA_GIBBS_NW", "1")
    monkeypatch.setenv("QA_GIBBS3_LEAN", "1")
    panel = medium_panel
    dev = DevicePanel(panel)
    rng = np.random.default_rng(18)
    samples, whichs, H0s, rus, rbs, rrs, frs, refs = [], [], [], [], [], [], [], []
    ffs = [0.2, 0.1, 0.3]
    for c, ff in enumerate(ffs):
        s = make_synthetic_sample(panel, seed=60 + c, n_reads=600 + 150 * c, ff=ff)
        which = np.sort(rng.choice(panel.K, 600, replace=False)).astype(np.int32) + 1
        R = s.nReads
        H0 = rng.choice([1, 2, 3], p=[0.5, 0.4, 0.1], size=R).astype(np.int32)
        ru, rb, rr = rng.random(R * 21), rng.random(3 * R), rng.random(3 * R)
        fr = int(rng.integers(0, R))
        refs.append(oracle.forwardBackwardGibbsNIPT(panel, s, which, H0, ru, fr, np.zeros(3 * panel.nGrids), ff=ff,
                                                    gibbs_initialize_iteratively=init_iter, runif_block=rb, runif_resample=rr))
        samples.append(s); whichs.append(which); H0s.append(H0); rus.append(ru); rbs.append(rb); rrs.append(rr); frs.append(fr)
    from quilt_amd.gibbs_nipt import rcpp_forwardBackwardGibbsNIPT
    for c in range(3):
        got = rcpp_forwardBackwardGibbsNIPT(dev, samples[c], whichs[c], H0s[c], rus[c], frs[c], None, ff=ffs[c],
                                            gibbs_initialize_iteratively=init_iter, runif_block=rbs[c], runif_resample=rrs[c])
        r = refs[c]
        assert r["status"] == 0 and not got["underflow_problem"]
        assert np.array_equal(got["H"], r["H"]), f"{(got['H'] != r['H']).sum()} labels differ"
        assert np.array_equal(got["H_class"], r["H_class"])
        np.testing.assert_allclose(got["hapProbs_t"], r["hapProbs_t"], rtol=RTOL, atol=1e-14)
        np.testing.assert_allclose(got["genProbsM_t"], r["genProbsM_t"], rtol=RTOL, atol=1e-14)
        np.testing.assert_allclose(got["genProbsF_t"], r["genProbsF_t"], rtol=RTOL, atol=1e-14)
    dev.close()


def test_nipt_edge_inputs(small_panel, oracle):
    """NIPT block Gibbs on awkward inputs: a handful of reads (blocks without reads are merged away, possibly all but
    one), reads piled on the first / last grid, a single block."""
    import copy
    from quilt_amd.gibbs_nipt import rcpp_forwardBackwardGibbsNIPT
    from quilt_amd.native import DevicePanel
    from quilt_amd.synth import make_synthetic_sample
    panel = small_panel
    dev = DevicePanel(panel)
    ff = 0.3
    rng = np.random.default_rng(12)
    which = np.sort(rng.choice(panel.K, 70, replace=False)).astype(np.int32) + 1
    G = panel.nGrids
    cases = []
    for n_reads in (2, 3, 7):
        cases.append(make_synthetic_sample(panel, seed=60 + n_reads, n_reads=n_reads, ff=ff))
    s = copy.deepcopy(make_synthetic_sample(panel, seed=70, n_reads=40, ff=ff))
    s.wif[:20] = 0; s.u[: s.read_ptr[20]] %= 32
    s.wif[20:] = G - 1; s.u[s.read_ptr[20]:] = 32 * (G - 1) + s.u[s.read_ptr[20]:] % (panel.nSNPs - 32 * (G - 1))
    cases.append(s)
    for s in cases:
        R = s.nReads
        H0 = rng.choice([1, 2, 3], p=[0.5, 0.35, 0.15], size=R).astype(np.int32)
        ru, rb, rr = rng.random(R * 21), rng.random(3 * R), rng.random(3 * R)
        ref = oracle.forwardBackwardGibbsNIPT(panel, s, which, H0, ru, 0, np.zeros(3 * G), ff=ff, runif_block=rb, runif_resample=rr)
        got = rcpp_forwardBackwardGibbsNIPT(dev, s, which, H0, ru, 0, None, ff=ff, runif_block=rb, runif_resample=rr)
        assert ref["status"] == 0 and not got["underflow_problem"]
        assert np.array_equal(got["H"], ref["H"]) and np.array_equal(got["H_class"], ref["H_class"]), R
        np.testing.assert_allclose(got["hapProbs_t"], ref["hapProbs_t"], rtol=RTOL, atol=1e-14)
    dev.close()



def test_nipt_underflow_is_reported(small_panel, oracle):
    """NIPT with block passes when the sweeps underflow: the chain stops, the block definition is skipped for it, the call
    reports underflow_problem (the driver then retries with a smaller maxDifferenceBetweenReads)."""
    import copy
    from quilt_amd.gibbs_nipt import rcpp_forwardBackwardGibbsNIPT
    from quilt_amd.native import DevicePanel
    from quilt_amd.synth import make_synthetic_sample
    panel = small_panel
    dev = DevicePanel(panel)
    s = copy.deepcopy(make_synthetic_sample(panel, seed=78, n_reads=4000, ff=0.2))
    g0 = int(s.wif[len(s.wif) // 2])
    s.wif[:] = g0
    s.u[:] = 32 * g0 + (s.u % 32)
    rng = np.random.default_rng(2)
    which = np.sort(rng.choice(panel.K, 64, replace=False)).astype(np.int32) + 1
    R = s.nReads
    H0 = rng.choice([1, 2, 3], p=[0.5, 0.4, 0.1], size=R).astype(np.int32)
    ru, rb, rr = rng.random(R * 21), rng.random(3 * R), rng.random(3 * R)
    ref = oracle.forwardBackwardGibbsNIPT(panel, s, which, H0, ru, 0, np.zeros(3 * panel.nGrids), ff=0.2, runif_block=rb,
                                          runif_resample=rr)
    got = rcpp_forwardBackwardGibbsNIPT(dev, s, which, H0, ru, 0, None, ff=0.2, runif_block=rb, runif_resample=rr)
    assert ref["status"] == 1, "the test input is meant to underflow"
    assert got["underflow_problem"]
    dev.close()


def test_wave_sum_lane_maps_on_the_device(tmp_path):
    """The samplers' two- and four-value wave sums rest on what v_permlane32_swap / v_permlane16_swap do with the lanes
    (gibbs_dev.hpp: wsum2, wsum3, wsum4); scripts/micro/permlane_sum2.hip / permlane_sum4.hip state the expected lane of every
    total and print what the device delivers."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for name in ("permlane_sum2", "permlane_sum4"):
        exe = str(tmp_path / name)
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", os.path.join(root, "scripts", "micro", name + ".hip"), "-o", exe],
                       check=True, capture_output=True, timeout=300)
        out = subprocess.run([exe], check=True, capture_output=True, text=True, timeout=60).stdout
        if name == "permlane_sum4":
            got = [float(v) for v in re.search(r"lanes 15 31 47 63 = (.*)", out).group(1).split()]
            a, b, c, d = [float(v) for v in re.search(r"expected a b c d\s*= (.*)", out).group(1).split()]
            assert got == [a, c, b, d], out   # rows 0..3 of the butterfly hold a, c, b, d
        else:
            m = re.search(r"lane31 (\S+) lane63 (\S+) ; expected sum\(a\) (\S+) sum\(b\) (\S+)", out)
            assert float(m.group(1)) == float(m.group(3)) and float(m.group(2)) == float(m.group(4)), out

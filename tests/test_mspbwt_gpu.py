"""use_mspbwt = TRUE on the device: the haplotype search (csrc/match.hip) against its numpy
statement -- integer work, identical -- and the mode end to end against the CPU path."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sparse_panel():
    """Mostly reference alleles: the all-zero word is a real symbol (usually the first) of every grid's dictionary, next to the
    zero padding of distinctHapsB's unused rows."""
    from quilt_amd.synth import make_synthetic_panel
    from tests.util import panel_from_rhb
    base = make_synthetic_panel(K=600, nSNPs=640, seed=3)
    rng = np.random.default_rng(4)
    rhb = np.asfortranarray(np.where(rng.random(base.rhb_t.shape) < 0.7, 0, base.rhb_t).astype(np.int32))
    p = panel_from_rhb(rhb, base.transMatRate_t, 640, 255, base.ref_error)
    p.L_grid = base.L_grid
    return p


@pytest.mark.parametrize("which", ["small", "ragged", "medium", "sparse"])
def test_device_search_equals_its_definition(small_panel, ragged_panel, medium_panel, sparse_panel, which):
    from quilt_amd.mspbwt import find_good_matches, match_tables_as_lists, rcpp_int_contract
    from quilt_amd.native import DevicePanel
    from quilt_amd.synth import make_truth_haplotype, panel_hap_bits
    from tests.oracle_backend import find_good_matches_bruteforce
    panel = dict(small=small_panel, ragged=ragged_panel, medium=medium_panel, sparse=sparse_panel)[which]
    rng = np.random.default_rng(5)
    queries = [panel_hap_bits(panel, 3)]                               # a panel haplotype: matches itself end to end
    queries += [make_truth_haplotype(panel, rng) for _ in range(3)]    # mosaics of panel haplotypes
    noisy = queries[1].copy()
    flip = rng.random(len(noisy)) < 0.02
    noisy[flip] = 1 - noisy[flip]
    queries.append(noisy)                                              # words that are in no dictionary
    queries.append(np.zeros(panel.nSNPs, dtype=np.int8))
    Zs = np.stack([rcpp_int_contract(q) for q in queries])
    dev = DevicePanel(panel)
    for nind, min_len, n_max in ((1, 1, 7), (2, 1, 40), (4, 2, 150), (3, 1, 1), (4, 1, 2000)):
        got = match_tables_as_lists(*find_good_matches(dev, Zs, nind, min_len, n_max))
        ref = find_good_matches_bruteforce(panel, Zs, nind, min_len, n_max)
        for q in range(len(Zs)):
            for i in range(nind):
                assert np.array_equal(got[q][i], ref[q][i]), (which, nind, min_len, n_max, q, i)
    dev.close()


@pytest.mark.parametrize("search", ["scan", "exhaustive"])
def test_pipeline_mspbwt_matches_the_cpu_path(medium_panel, search):
    """``scan``: the product's msPBWT neighbour scan (csrc/mspbwt.cpp) against the numpy restatement the oracle backend runs;
    ``exhaustive``: the device search (csrc/match.hip) against its numpy definition."""
    from quilt_amd.driver import Driver, DriverParams, HipBackend
    from quilt_amd.native import DevicePanel
    from quilt_amd.synth import make_synthetic_sample
    from tests.oracle_backend import OracleBackend
    from tests.util import r2
    panel = medium_panel
    samples = [make_synthetic_sample(panel, seed=900 + i, n_reads=600) for i in range(2)]
    prm = DriverParams(nGibbsSamples=2, Ksubset=100, Knew=100, seed=3, use_mspbwt=True, mspbwt_nindices=2, mspbwt_search=search)
    dev = DevicePanel(panel)
    got = Driver(panel, HipBackend(dev), prm).run(samples)
    dev.close()
    ref = Driver(panel, OracleBackend(panel), prm).run(samples)
    for s, g, r in zip(samples, got, ref):
        assert np.array_equal(g.read_labels, r.read_labels)
        assert np.abs(g.dosage - r.dosage).max() <= 1e-6      # the dosages are the Gibbs call's fp64 hapProbs
        assert r2(g.dosage, s.truth_haps[:2].sum(axis=0)) > 0.8


def test_hap_words_formed_on_the_device(medium_panel):
    """qa_gibbs_opts_t.hap_words_out: rcpp_int_contract(round(hapProbs_t)) per chain and haplotype (mspbwt.R:271-272), formed on
    the device from the hapProbs the same call returns."""
    from quilt_amd.gibbs_nipt import forwardBackwardGibbsNIPT_batch
    from quilt_amd.mspbwt import int_contract_rows
    from quilt_amd.native import DevicePanel
    from quilt_amd.synth import make_synthetic_sample
    panel = medium_panel
    rng = np.random.default_rng(2)
    samples = [make_synthetic_sample(panel, seed=700 + i, n_reads=400) for i in range(3)]
    which = [np.sort(rng.choice(panel.K, 100, replace=False)).astype(np.int32) + 1 for _ in samples]
    H0 = [rng.integers(1, 3, size=s.nReads).astype(np.int32) for s in samples]
    seeds = rng.integers(0, 2 ** 63, size=3).astype(np.uint64)
    dev = DevicePanel(panel)
    out = forwardBackwardGibbsNIPT_batch(dev, samples, which, H0, None, [0, 5, 9], None, seed_reads=seeds, seed_shard=seeds + 1,
                                         return_hapProbs=True, return_genProbs=False, return_hap_words=True)
    only = forwardBackwardGibbsNIPT_batch(dev, samples, which, H0, None, [0, 5, 9], None, seed_reads=seeds, seed_shard=seeds + 1,
                                          return_hapProbs=False, return_genProbs=False, return_hap_words=True)
    dev.close()
    for o, w in zip(out, only):
        assert np.array_equal(o["hap_words"], int_contract_rows(np.asarray(o["hapProbs_t"])))
        assert np.array_equal(o["hap_words"], w["hap_words"]) and "hapProbs_t" not in w


def test_selection_from_the_device_search_and_from_the_neighbour_scan(medium_panel):
    """The device search's definition (every haplotype's longest run, the longest first) against the restated msPBWT neighbour
    scan (tests/mspbwt_scan.py; parity with the mspbwt package itself is unpinned, the package is not in the reference tree):
    share of the next small panel chosen from both (bar 0.55 on clean mosaic queries, 0.45 with 0.5 % of the alleles flipped) and
    share of the scan's longest matches the device search reports (0.95 / 0.9).  bench.py --mspbwt reports the same statistic on
    the K = 50 000 panel."""
    from quilt_amd.mspbwt import find_good_matches, match_tables_as_lists
    from quilt_amd.native import DevicePanel
    from tests.mspbwt_scan import find_good_matches_scan, selection_agreement
    from tests.test_mspbwt_cpu import _scan_queries
    panel = medium_panel
    dev = DevicePanel(panel)
    for err, bar_sel, bar_long in ((0.0, 0.55, 0.95), (0.005, 0.45, 0.9)):
        Zs = _scan_queries(panel, 2, err, 11)
        sc = find_good_matches_scan(panel, Zs, 4, L=3, M=1)
        got = match_tables_as_lists(*find_good_matches(dev, Zs, 4, 1, 150))
        a = selection_agreement(sc, got, 100, panel.K, panel.nGrids)
        assert a["selected"] >= bar_sel and a["longest"] >= bar_long, (err, a)
    dev.close()

"""use_mspbwt = TRUE: the selection logic of QUILT/R/mspbwt.R:225-474 restated in
quilt_amd/mspbwt.py, on hand-made match tables, and the mode end to end on the CPU path (oracle Gibbs + the numpy statement
of the device search)."""
import numpy as np
import pytest


def test_int_contract_and_mtm_table():
    from quilt_amd.mspbwt import matches_to_mtm, rcpp_int_contract
    hap = np.zeros(70, dtype=int)
    hap[[0, 3, 31, 32, 69]] = 1
    assert rcpp_int_contract(hap).tolist() == [np.int32(1 | 8 | (1 << 31) - (1 << 32)).item(), 1, 1 << 5]
    # index 1: haplotype 4 twice from the same start (the longer one stays), haplotype 2 once; index 2: haplotype 7
    idx1 = np.array([[4, 2, 5], [4, 2, 3], [2, 0, 9]])       # (index0, start0, len1)
    idx2 = np.array([[7, 1, 9]])
    mtm = matches_to_mtm([idx1, idx2], nGrids=40)
    # columns index1, start1, end1, len1, key, n; ordered by length (stable: index 1's rows before index 2's at a tie)
    assert mtm[:, :4].tolist() == [[3, 1, 9, 9], [8, 2, 10, 9], [5, 3, 7, 5]]
    assert mtm[:, 4].tolist() == [40 * 1 + 9, 40 * 2 + 10, 40 * 3 + 7] and mtm[:, 5].tolist() == [1, 2, 1]
    assert matches_to_mtm([np.zeros((0, 3), dtype=int), None], 40).shape == (0, 6)


def test_select_new_haps_branches():
    from quilt_amd.mspbwt import select_new_haps_mspbwt_v3
    none = [[np.zeros((0, 3), dtype=int)], [np.zeros((0, 3), dtype=int)]]
    out = select_new_haps_mspbwt_v3(none, Knew=10, Kfull=100, nGrids=50, seed_select=3)
    assert len(set(out.tolist())) == 10 and out.min() >= 1 and out.max() <= 100            # sample(1:Kfull, Knew)
    few = [[np.array([[4, 0, 9], [9, 3, 4]])], [np.array([[4, 5, 2], [11, 0, 7]])]]
    out = select_new_haps_mspbwt_v3(few, Knew=6, Kfull=100, nGrids=50, seed_select=3)
    assert out[:3].tolist() == [5, 10, 12] and len(set(out.tolist())) == 6                 # found ones first, then a top-up
    # more candidates than Knew: weights = length / coverage so far, interleaved between the two haplotypes
    h1 = np.array([[0, 0, 10], [1, 0, 10], [2, 20, 5], [3, 30, 4]])     # 1 and 2 cover the same stretch: the second weighs half
    h2 = np.array([[10, 0, 8], [11, 10, 8], [12, 20, 8]])
    out = select_new_haps_mspbwt_v3([[h1], [h2]], Knew=4, Kfull=100, nGrids=50, seed_select=3)
    # h1 by length: 1 (w 1), 2 (w 0.5), 3 (w 1), 4 (w 1) -> order by weight, stable: 1, 3, 4, 2; h2: 11, 12, 13 (all 1)
    assert out.tolist() == [1, 11, 3, 12]
    out = select_new_haps_mspbwt_v3([[h1], [h2]], Knew=6, Kfull=100, nGrids=50, seed_select=3)
    assert out.tolist() == [1, 11, 3, 12, 4, 13]
    # exactly as many as wanted: the identified haplotypes as they come (haplotype 1's by length, then haplotype 2's)
    out = select_new_haps_mspbwt_v3([[h1], [h2]], Knew=7, Kfull=100, nGrids=50, seed_select=3)
    assert out.tolist() == [1, 2, 3, 4, 11, 12, 13]


def test_bruteforce_search_definition(small_panel):
    """The numpy statement of the search on a query copied from a panel haplotype: that haplotype matches end to end."""
    from quilt_amd.mspbwt import rcpp_int_contract
    from quilt_amd.synth import panel_hap_bits
    from tests.oracle_backend import find_good_matches_bruteforce
    panel = small_panel
    k = 17
    Z = rcpp_int_contract(panel_hap_bits(panel, k))
    found = find_good_matches_bruteforce(panel, Z[None, :], 2, 1, 20)[0]
    for i, m in enumerate(found):
        n_pos = len(range(i, panel.nGrids, 2))
        row = m[m[:, 0] == k]
        special = np.asarray(panel.hapMatcherR)[k, i::2] == 0
        if not special.any():
            assert len(row) == 1 and row[0, 1] == 0 and row[0, 2] == n_pos
        assert (np.diff(m[:, 0]) > 0).all() and (m[:, 2] >= 1).all() and len(m) <= 20


def test_pipeline_mspbwt_on_the_cpu_path(medium_panel):
    from quilt_amd.driver import Driver, DriverParams
    from quilt_amd.synth import make_synthetic_sample
    from tests.oracle_backend import OracleBackend
    from tests.util import r2
    import pytest
    panel = medium_panel
    samples = [make_synthetic_sample(panel, seed=900 + i, n_reads=600) for i in range(2)]
    prm = DriverParams(nGibbsSamples=2, Ksubset=100, Knew=100, seed=3, use_mspbwt=True, mspbwt_nindices=2)
    res = Driver(panel, OracleBackend(panel), prm).run(samples)
    for s, r in zip(samples, res):
        assert r.nDosage == 2 and np.isfinite(r.dosage).all() and r.dosage.min() >= 0 and r.dosage.max() <= 2 + 1e-9
        np.testing.assert_allclose(r.gp_t.sum(axis=0), 1.0, atol=1e-9)
        assert r2(r.dosage, s.truth_haps[:2].sum(axis=0)) > 0.8
    with pytest.raises(ValueError, match="Knew must equal"):
        Driver(panel, OracleBackend(panel), DriverParams(Ksubset=100, Knew=50, use_mspbwt=True)).run(samples)


def test_native_batch_selection_equals_the_numpy_text():
    """csrc/hostio.cpp qa_select_new_haps_mspbwt against quilt_amd/mspbwt.py select_new_haps_mspbwt_v3 on random match tables
    that reach all three branches (no match, fewer than Knew, more than Knew), two and three haplotypes per chain."""
    from quilt_amd.mspbwt import match_tables_as_lists, select_new_haps_mspbwt_batch, select_new_haps_mspbwt_v3
    rng = np.random.default_rng(12)
    Kfull, nGrids = 300, 200
    for n_label in (2, 3):
        for Knew, n_avail in ((20, 0), (40, 12), (15, 60), (50, 200), (7, 7)):
            n_chain, ni, mm = 6, 3, 25
            match = np.zeros((n_chain * n_label, ni, mm, 3), dtype=np.int32)
            n = np.zeros((n_chain * n_label, ni), dtype=np.int32)
            haps_pool = rng.choice(Kfull, size=max(n_avail, 1), replace=False)
            for q in range(n_chain * n_label):
                for i in range(ni):
                    cnt = 0 if n_avail == 0 else int(rng.integers(0, mm + 1))
                    hp = np.sort(rng.choice(haps_pool, size=min(cnt, len(haps_pool)), replace=False))
                    n[q, i] = len(hp)
                    match[q, i, :len(hp), 0] = hp
                    match[q, i, :len(hp), 1] = rng.integers(0, 30, size=len(hp))
                    match[q, i, :len(hp), 2] = rng.integers(1, 20, size=len(hp))
            seeds = [int(x) for x in rng.integers(0, 2 ** 63, size=n_chain)]
            got = select_new_haps_mspbwt_batch(match, n, n_label, Knew, Kfull, nGrids, seeds)
            lists = match_tables_as_lists(match, n)
            for c in range(n_chain):
                ref = select_new_haps_mspbwt_v3(lists[c * n_label:(c + 1) * n_label], Knew, Kfull, nGrids, seeds[c])
                assert np.array_equal(got[c], ref), (n_label, Knew, n_avail, c)


def _scan_queries(panel, n, err, seed):
    from quilt_amd.mspbwt import rcpp_int_contract
    from quilt_amd.synth import make_truth_haplotype
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        h = make_truth_haplotype(panel, rng).copy()
        flip = rng.random(len(h)) < err
        h[flip] = 1 - h[flip]
        out.append(rcpp_int_contract(h))
    return np.stack(out)


def test_neighbour_scan_holds_the_prefix_order_property(small_panel, ragged_panel):
    """tests/mspbwt_scan.py, the restated msPBWT query (parity unpinned: the package is not in the reference tree): with
    check=True every position asserts that the haplotypes next to the query's insertion point carry the longest matches ending
    there and that the lengths fall away on both sides -- the property of the positional prefix order the scan relies on."""
    from tests.mspbwt_scan import find_good_matches_scan
    for panel, nind in ((small_panel, 2), (ragged_panel, 3)):
        Zs = _scan_queries(panel, 3, 0.01, 7)
        found = find_good_matches_scan(panel, Zs, nind, L=3, M=1, check=True)
        for per_index in found:
            for i, m in enumerate(per_index):
                n_pos = len(range(i, panel.nGrids, nind))
                assert len(m) and (m[:, 2] >= 1).all() and (m[:, 1] + m[:, 2] <= n_pos).all()
                assert len({(int(k), int(s)) for k, s, _ in m}) == len(m)      # one row per (haplotype, start)
        # M: only matches at least that long are reported
        for per_index in find_good_matches_scan(panel, Zs, nind, L=3, M=3):
            assert all((m[:, 2] >= 3).all() for m in per_index)


def test_selection_from_the_search_and_from_the_neighbour_scan(medium_panel):
    """How far the next small panel chosen from this library's search definition (every haplotype's longest run, the longest
    first) agrees with the one chosen from the msPBWT neighbour scan, on mosaic queries: stated bars, the same ones the GPU
    test and bench.py --mspbwt report against."""
    from tests.mspbwt_scan import find_good_matches_scan, selection_agreement
    from tests.oracle_backend import find_good_matches_bruteforce
    panel = medium_panel
    for err, bar_sel, bar_long in ((0.0, 0.55, 0.95), (0.005, 0.45, 0.9)):
        Zs = _scan_queries(panel, 2, err, 11)
        sc = find_good_matches_scan(panel, Zs, 4, L=3, M=1)
        bf = find_good_matches_bruteforce(panel, Zs, 4, 1, 150)
        a = selection_agreement(sc, bf, 100, panel.K, panel.nGrids)
        assert a["selected"] >= bar_sel and a["longest"] >= bar_long, (err, a)


@pytest.mark.parametrize("which,nind,L,M", [("small", 4, 3, 1), ("ragged", 3, 5, 2), ("medium", 1, 2, 1), ("medium", 4, 3, 1),
                                            ("small", 2, 64, 1)])
def test_native_neighbour_scan_equals_the_restatement(small_panel, ragged_panel, medium_panel, which, nind, L, M):
    """csrc/mspbwt.cpp (the panel's msPBWT indices + the neighbour scan: the product's query behind select_new_haps_mspbwt_v3,
    host code) against tests/mspbwt_scan.py, written apart from it: the same (haplotype, start, length) rows for every query and
    index -- mosaic queries, noisy ones (words in no dictionary: no symbol), a panel haplotype itself, the all-reference
    haplotype -- and the same next small panel from the fused scan + selection call."""
    from quilt_amd.mspbwt import MsPbwtIndex, rcpp_int_contract, select_new_haps_mspbwt_v3
    from quilt_amd.synth import panel_hap_bits
    from tests.mspbwt_scan import find_good_matches_scan
    panel = dict(small=small_panel, ragged=ragged_panel, medium=medium_panel)[which]
    Zs = np.concatenate([_scan_queries(panel, 2, 0.0, 3), _scan_queries(panel, 2, 0.02, 4),
                         np.stack([rcpp_int_contract(panel_hap_bits(panel, 5)), rcpp_int_contract(np.zeros(panel.nSNPs))])])
    idx = MsPbwtIndex(panel, nind)
    got = idx.find_good_matches(Zs, L, M)
    want = find_good_matches_scan(panel, Zs, nind, L, M)
    for q in range(len(Zs)):
        for i in range(nind):
            assert np.array_equal(got[q][i], want[q][i]), (q, i)
    assert sum(len(m) for per in got for m in per) > 0
    Knew = min(60, panel.K // 2)
    sel = idx.select_new_haps(Zs, 2, L, M, Knew, [5, 6, 7])
    for c in range(3):
        assert np.array_equal(sel[c], select_new_haps_mspbwt_v3(want[2 * c:2 * c + 2], Knew, panel.K, panel.nGrids, [5, 6, 7][c]))
    idx.close()


def test_native_scan_refuses_what_it_cannot_index(small_panel):
    from quilt_amd.mspbwt import MsPbwtIndex
    with pytest.raises(ValueError):
        MsPbwtIndex(small_panel, 0)
    idx = MsPbwtIndex(small_panel, 2)
    with pytest.raises(ValueError):
        idx.find_good_matches(np.zeros((1, small_panel.nGrids), dtype=np.int32), 65, 1)
    idx.close()


def test_pipeline_mspbwt_scan_equals_exhaustive_interface(medium_panel):
    """Both queries drive the same driver: the scan (default, the reference's semantics) and the exhaustive device-search
    definition give complete, valid small panels; on the oracle backend the scan path is the numpy restatement."""
    from quilt_amd.driver import Driver, DriverParams
    from quilt_amd.synth import make_synthetic_sample
    from tests.oracle_backend import OracleBackend
    panel = medium_panel
    samples = [make_synthetic_sample(panel, seed=900 + i, n_reads=300) for i in range(1)]
    for search in ("scan", "exhaustive"):
        prm = DriverParams(nGibbsSamples=2, Ksubset=100, Knew=100, seed=3, use_mspbwt=True, mspbwt_nindices=2, mspbwt_search=search)
        res = Driver(panel, OracleBackend(panel), prm).run(samples)
        assert res[0].nDosage > 0 and np.isfinite(res[0].dosage).all()
    with pytest.raises(ValueError):
        DriverParams(use_mspbwt=True, Ksubset=100, Knew=100, mspbwt_search="index").resolved(panel.K)

"""CPU tests of the host-side driver logic restated from QUILT/R/functions.R."""
import numpy as np
import pytest

from quilt_amd import driver as D


def test_thinned_columns_default():
    cols = D.thinned_grid_columns(2000, 0.1)
    idx = np.nonzero(cols >= 0)[0]
    assert len(idx) == 200 and idx[0] == 0 and idx[-1] == 1999
    assert np.array_equal(cols[idx], np.arange(200))
    assert np.array_equal(np.nonzero(D.thinned_grid_columns(5, 0.1) >= 0)[0], [0])


def test_select_good_haps_takes_rank_by_rank():
    rng = np.random.default_rng(0)
    # two labels, three grids; rank-1 haplotypes first, then rank-2, ...
    new_haps = [[np.array([5, 9, 1, 2, 3]), np.array([7, 5, 4, 6, 8]), np.array([5, 7, 10, 11, 12])],
                [np.array([20, 21, 22, 23, 24]), np.array([20, 5, 25, 26, 27]), np.array([30, 31, 32, 33, 34])]]
    out = D.everything_select_good_haps(4, 5, new_haps, np.zeros(0, dtype=np.int64), 100, 11)
    assert set(out.tolist()) == {5, 7, 20, 30}           # exactly the distinct rank-1 entries
    out = D.everything_select_good_haps(6, 5, new_haps, np.array([7]), 100, 12)
    assert {5, 20, 30} <= set(out.tolist()) and 7 not in out and len(set(out.tolist())) == 6
    # the subsample of the overshooting rank depends on the seed only (the device draws the same keys)
    a = D.everything_select_good_haps(5, 5, new_haps, np.zeros(0, dtype=np.int64), 100, 13)
    b = D.everything_select_good_haps(5, 5, new_haps, np.zeros(0, dtype=np.int64), 100, 13)
    c = [D.everything_select_good_haps(5, 5, new_haps, np.zeros(0, dtype=np.int64), 100, sd) for sd in range(20, 40)]
    assert np.array_equal(a, b) and len({tuple(x.tolist()) for x in c}) == 3   # rank 2 offers 9, 21, 31
    # not enough candidates: filled at random from the rest of the panel
    out = D.everything_select_good_haps(40, 5, new_haps, np.zeros(0, dtype=np.int64), 100, 14)
    assert len(set(out.tolist())) == 40 and out.min() >= 1 and out.max() <= 100
    # previously selected haplotypes: a keyed subset of the current small panel, no repeats
    w = np.arange(1, 601, dtype=np.int32)[::-1].copy()
    p1, p2 = D.previously_selected(w, 100, 5), D.previously_selected(w, 100, 6)
    assert len(set(p1.tolist())) == 100 and set(p1.tolist()) <= set(w.tolist()) and not np.array_equal(p1, p2)


def test_recast_haps():
    hd1 = np.array([0.9, 0.2, 0.6, 0.4])
    hd2 = np.array([0.8, 0.1, 0.3, 0.45])
    gp = np.array([[0.0, 0.1, 0.9], [0.8, 0.2, 0.0], [0.1, 0.8, 0.1], [0.2, 0.7, 0.1]])
    h1, h2 = D.recast_haps(hd1, hd2, gp)
    assert np.allclose(h1, [0.9, 0.2, 0.6, 0.0]) and np.allclose(h2, [0.8, 0.1, 0.3, 1.0])


def test_best_read_labels_majority_flip():
    # 7 runs that agree except that runs 0 and 1 are phase-flipped after read 20
    R, n = 60, 7
    rng = np.random.default_rng(1)
    base = rng.integers(1, 3, size=R)
    m = np.tile(base[:, None], (1, n))
    m[20:, 0] = 3 - m[20:, 0]
    m[20:, 1] = 3 - m[20:, 1]
    conf = np.ones((R, n), dtype=bool)
    out = D.determine_best_read_label_so_far(m, conf, R, n, can_hap=n)
    assert np.array_equal(out, base)                       # canonical run is trusted, nothing to change in it
    # canonical run itself is the odd one out: it gets flipped back
    m2 = np.tile(base[:, None], (1, n))
    m2[30:, n - 1] = 3 - m2[30:, n - 1]
    out2 = D.determine_best_read_label_so_far(m2, conf, R, n, can_hap=n)
    assert np.array_equal(out2, base)
    # too few confident reads: canonical labels unchanged
    out3 = D.determine_best_read_label_so_far(m2, np.zeros((R, n), dtype=bool), R, n, can_hap=n)
    assert np.array_equal(out3, m2[:, n - 1])


def test_params_small_panel_reset():
    p = D.DriverParams().resolved(K=100)
    assert (p.n_seek_its, p.n_burn_in_seek_its, p.Ksubset, p.Knew) == (1, 0, 100, 100)
    assert D.DriverParams().resolved(K=5000).n_burn_in_seek_its == 2


def test_pipelined_batches_equal_separate_runs():
    """run_stream fuses the phasing rounds of a batch with the main rounds of the next one (mixed launches, first-round
    chains next to later-round chains): per-sample results must not depend on it."""
    from quilt_amd.synth import make_synthetic_panel, make_synthetic_sample
    from tests.oracle_backend import OracleBackend
    panel = make_synthetic_panel(K=400, nSNPs=320, seed=21)
    samples = [make_synthetic_sample(panel, seed=50 + i, n_reads=60) for i in range(5)]
    prm = D.DriverParams(nGibbsSamples=2, Ksubset=64, Knew=64, seed=9)
    batches = [(samples[0:2], 0), (samples[2:3], 2), (samples[3:5], 3)]
    streamed = list(D.Driver(panel, OracleBackend(panel), prm).run_stream(batches))
    assert [len(b) for b in streamed] == [2, 1, 2]
    for (smp, off), got in zip(batches, streamed):
        ref = D.Driver(panel, OracleBackend(panel), prm).run(smp, sample_offset=off)
        for g, r in zip(got, ref):
            assert np.array_equal(g.read_labels, r.read_labels)
            assert np.array_equal(g.dosage, r.dosage) and np.array_equal(g.phasing_haps, r.phasing_haps)


def test_best_read_labels_lazy_form_equals_literal_form():
    """determine_best_read_label_so_far keeps the reference's suffix rewrites as flip parities; the line-by-line
    restatement of functions.R:1680-1784 is the check."""
    rng = np.random.default_rng(3)
    for _ in range(200):
        R = int(rng.integers(5, 300))
        n = int(rng.choice([1, 2, 3, 7]))
        m = np.tile(rng.integers(1, 3, size=R)[:, None], (1, n))
        for c in range(n):
            for pos in rng.integers(0, R, size=rng.integers(0, 4)):
                m[pos:, c] = 3 - m[pos:, c]
            noise = rng.random(R) < rng.choice([0, 0.02, 0.2])
            m[noise, c] = 3 - m[noise, c]
        conf = rng.random((R, n)) < rng.choice([0.3, 0.8, 1.0])
        can = int(rng.integers(1, n + 1))
        assert np.array_equal(D.determine_best_read_label_so_far(m, conf, R, n, can_hap=can),
                              D._determine_best_read_label_so_far_literal(m, conf, R, n, can_hap=can))


def test_rare_common_pipeline_on_the_oracle():
    """impute_rare_common (functions.R:1042-1123, rare_common.R:109-420): every Gibbs sample ends with a Gibbs call over
    ALL SNPs; results switch over to all SNPs (functions.R:1305-1307) and stay independent of the batching."""
    from quilt_amd.synth import make_rare_common, make_synthetic_panel, make_synthetic_sample_rare_common
    from tests.oracle_backend import OracleBackend
    from tests.util import r2
    panel = make_synthetic_panel(K=400, nSNPs=320, seed=21)
    rc = make_rare_common(panel, 3)
    samples = [make_synthetic_sample_rare_common(panel, rc, 50 + i, n_reads=150)[0] for i in range(3)]
    prm = D.DriverParams(nGibbsSamples=2, Ksubset=64, Knew=64, seed=9, impute_rare_common=True)
    one = D.Driver(panel, OracleBackend(panel, rc), prm, rare_common=rc).run(samples)
    for smp, r in zip(samples, one):
        assert r.dosage.shape == (rc.nSNPs_all,) and r.gp_t.shape == (3, rc.nSNPs_all) and r.nDosage == 2
        assert r.phasing_haps.shape == (rc.nSNPs_all, 2)
        np.testing.assert_allclose(r.gp_t.sum(axis=0), 1.0, atol=1e-9)
        truth = smp.all_snp.truth_haps.sum(axis=0)
        assert r2(r.dosage, truth) > 0.5
        assert r.dosage[rc.snp_is_common == 0].min() >= 2 * panel.ref_error - 1e-12   # rare SNPs: ref_error per haplotype at least
    streamed = list(D.Driver(panel, OracleBackend(panel, rc), prm, rare_common=rc).run_stream([(samples[:2], 0), (samples[2:], 2)]))
    for g, r in zip(streamed[0] + streamed[1], one):
        assert np.array_equal(g.dosage, r.dosage) and np.array_equal(g.phasing_haps, r.phasing_haps)
    with pytest.raises(ValueError):
        D.Driver(panel, OracleBackend(panel), prm)


def test_get_initial_read_labels():
    e = np.array([[0.9, 0.1, 0.5], [0.1, 0.9, 0.5]])
    assert D.get_initial_read_labels(e, np.array([0.5, 0.5, 0.7])).tolist() == [2, 1, 1]


def test_nipt_pipeline_on_the_oracle():
    """method = "nipt" (functions.R:586, 1009-1016, 1188-1199, 1218-1231): three read labels, mother and fetus
    accumulators, the NIPT consensus and recast; batching does not change results."""
    from quilt_amd.synth import make_synthetic_panel, make_synthetic_sample
    from tests.oracle_backend import OracleBackend
    from tests.util import r2
    panel = make_synthetic_panel(K=400, nSNPs=640, seed=21)
    samples = [make_synthetic_sample(panel, seed=70 + i, n_reads=300, ff=0.15 + 0.05 * i) for i in range(3)]
    prm = D.DriverParams(nGibbsSamples=2, Ksubset=64, Knew=64, seed=9, method="nipt")
    one = D.Driver(panel, OracleBackend(panel), prm).run(samples)
    for smp, r in zip(samples, one):
        assert r.phasing_haps.shape == (panel.nSNPs, 3) and set(np.unique(r.phasing_haps)) <= {0.0, 1.0}
        assert set(np.unique(r.read_labels)) <= {1, 2, 3}
        np.testing.assert_allclose(r.gp_t.sum(axis=0), 1.0, atol=1e-9)
        np.testing.assert_allclose(r.fet_gp_t.sum(axis=0), 1.0, atol=1e-9)
        mat = smp.truth_haps[0] + smp.truth_haps[1]
        fet = smp.truth_haps[0] + smp.truth_haps[2]
        assert r2(r.dosage, mat) > 0.3 and r2(r.fet_dosage, fet) > 0.2
    streamed = list(D.Driver(panel, OracleBackend(panel), prm).run_stream([(samples[:2], 0), (samples[2:], 2)]))
    for g, r in zip(streamed[0] + streamed[1], one):
        assert np.array_equal(g.dosage, r.dosage) and np.array_equal(g.fet_dosage, r.fet_dosage)
        assert np.array_equal(g.read_labels, r.read_labels)


def test_nipt_consensus_and_recast():
    rl = np.array([[1, 1], [3, 2], [2, 2], [1, 3]] * 3)
    conf = np.ones_like(rl, dtype=bool)
    out = D.determine_best_read_label_so_far_nipt(rl, conf, len(rl), 2, can_hap=2)
    assert np.array_equal(out, rl[:, 1])        # too few confident rows to flip anything; the 3s survive
    h1, h2, h3 = np.array([0.9, 0.2, 0.6, 0.4]), np.array([0.1, 0.7, 0.6, 0.4]), np.array([0.2, 0.9, 0.1, 0.6])
    mat = np.array([[0.1, 0.1, 0.0, 0.8], [0.8, 0.8, 0.1, 0.1], [0.1, 0.1, 0.9, 0.1]])
    fet = np.array([[0.1, 0.7, 0.1, 0.1], [0.8, 0.2, 0.1, 0.8], [0.1, 0.1, 0.8, 0.1]])
    a, b, c = D.recast_nipt_haps(h1, h2, h3, mat, fet)
    # site 0: (1, 1) with rounded haps (1, 0, 0) kept; site 1: (1, 0) -> (0, 1, 0); site 2: (2, 2) -> all 1; site 3: (0, 1) -> (0, 0, 1)
    assert (a.tolist(), b.tolist(), c.tolist()) == ([1, 0, 1, 0], [0, 1, 1, 0], [0, 0, 1, 1])


def test_nipt_rare_common_pipeline_on_the_oracle():
    """impute_rare_common with method = "nipt": all-SNP starting labels by read grouping (gibbs-nipt.R:1655-1849), the
    all-SNP Gibbs call with three labels and its block Gibbs on the all-SNP grid, mother / fetus outputs over all SNPs."""
    from quilt_amd.synth import make_rare_common, make_synthetic_panel, make_synthetic_sample_rare_common
    from tests.oracle_backend import OracleBackend
    panel = make_synthetic_panel(K=400, nSNPs=640, seed=21)
    rc = make_rare_common(panel, 3)
    samples = [make_synthetic_sample_rare_common(panel, rc, 90 + i, n_reads=300, ff=0.2)[0] for i in range(2)]
    prm = D.DriverParams(nGibbsSamples=2, Ksubset=64, Knew=64, seed=9, method="nipt", impute_rare_common=True)
    res = D.Driver(panel, OracleBackend(panel, rc), prm, rare_common=rc).run(samples)
    for r in res:
        assert r.dosage.shape == r.fet_dosage.shape == (rc.nSNPs_all,) and r.phasing_haps.shape == (rc.nSNPs_all, 3)
        np.testing.assert_allclose(r.gp_t.sum(axis=0), 1.0, atol=1e-9)
        np.testing.assert_allclose(r.fet_gp_t.sum(axis=0), 1.0, atol=1e-9)


def test_nipt_initial_labels_by_grouping():
    assert D.preserve_round(np.array([1.5, 2.5, 3.0])).tolist() == [1, 3, 3] and D.preserve_round(np.array([0.2, 0.8])).tolist() == [0, 1]
    rng = np.random.default_rng(1)
    e = np.array([[1.0, 0.1, 1.0, 0.2, 1.0], [0.1, 1.0, 1.0, 0.1, 0.9], [0.2, 0.2, 0.1, 1.0, 0.8]])
    H = D.get_initial_read_labels_nipt(e, 0.2, rng)
    assert H[0] == 1 and H[1] == 2 and H[3] == 3 and H[2] in (1, 2) and H[4] in (1, 2, 3)


def test_parameter_validation():
    """validate_n_seek_its_and_n_burn_in_seek_its (quilt.R): the burn-in must leave at least one counting iteration."""
    import pytest
    from quilt_amd.driver import DriverParams
    with pytest.raises(ValueError):
        DriverParams(n_seek_its=3, n_burn_in_seek_its=3).resolved(5000)
    with pytest.raises(ValueError):
        DriverParams(n_seek_its=0).resolved(5000)
    assert DriverParams(n_seek_its=1).resolved(5000).n_burn_in_seek_its == 0


def test_sample_without_reads_is_rejected():
    import pytest
    from quilt_amd.driver import Driver, DriverParams
    from quilt_amd.synth import SampleReads, make_synthetic_panel
    from tests.oracle_backend import OracleBackend
    panel = make_synthetic_panel(K=300, nSNPs=320, seed=3)
    empty = SampleReads(read_ptr=np.zeros(1, dtype=np.int32), u=np.zeros(0, dtype=np.int32), bq=np.zeros(0, dtype=np.int32),
                        wif=np.zeros(0, dtype=np.int32))
    with pytest.raises(ValueError, match="without reads"):
        Driver(panel, OracleBackend(panel), DriverParams(nGibbsSamples=1, Ksubset=32, Knew=32)).run([empty])


def test_bq_zero_bases_carry_nothing():
    """A base with bq == 0 is neither ref nor alt: host mirror, oracle and device kernel all skip it (functions.R:2018-2020)."""
    from oracle import oracle as O
    from quilt_amd.driver import make_gl_from_u_bq
    u = np.array([3, 3, 7, 9], dtype=np.int32)
    bq = np.array([30, 0, -25, 0], dtype=np.int32)
    gl = make_gl_from_u_bq(u, bq, 12, 1e-10, O.make_gl_bound)
    ref = O.make_gl_from_u_bq(u, bq, 12)
    assert np.array_equal(gl, ref)
    assert gl[0, 9] == 1 and gl[1, 9] == 1
    only = make_gl_from_u_bq(u[[0, 2]], bq[[0, 2]], 12, 1e-10, O.make_gl_bound)
    assert np.array_equal(gl, only)


def test_underflow_retry_reruns_only_the_failed_chains():
    """impute_one_sample's loop (functions.R:2612-2716): a chain whose Gibbs call reports underflow is re-run with
    maxDifferenceBetweenReads / 10, the others keep their result; more than ten consecutive failures is an error."""
    import pytest
    from quilt_amd.driver import Driver, DriverParams
    from quilt_amd.synth import make_synthetic_panel, make_synthetic_sample
    from tests.oracle_backend import OracleBackend
    panel = make_synthetic_panel(K=400, nSNPs=320, seed=3)
    samples = [make_synthetic_sample(panel, seed=50 + i, n_reads=60) for i in range(2)]

    class Flaky(OracleBackend):
        def __init__(self, panel, fail_above):
            super().__init__(panel)
            self.fail_above, self.calls = fail_above, []

        def gibbs_batch(self, samples, which, *a, maxDifferenceBetweenReads, **kw):
            self.calls.append((len(samples), maxDifferenceBetweenReads))
            out = super().gibbs_batch(samples, which, *a, maxDifferenceBetweenReads=maxDifferenceBetweenReads, **kw)
            if maxDifferenceBetweenReads > self.fail_above:
                out[0] = dict(out[0], underflow_problem=True)   # the first chain of every launch "underflows"
            return out

    prm = DriverParams(nGibbsSamples=2, Ksubset=48, Knew=48, seed=2)
    be = Flaky(panel, fail_above=1e8)
    drv = Driver(panel, be, prm)
    res = drv.run(samples)
    assert drv.n_underflow_retries > 0 and len(res) == 2
    first = be.calls[:3]
    assert first[0] == (4, 1e10) and first[1] == (1, 1e9) and first[2] == (1, 1e8)   # 2 samples x 2 chains, then the failed one alone
    assert all(np.isfinite(r.dosage).all() for r in res)
    with pytest.raises(RuntimeError, match="underflow"):
        Driver(panel, Flaky(panel, fail_above=0.5), prm).run(samples)


def test_truncated_lists_are_refetched():
    """A panel with duplicated haplotypes: gamma ties make best-haplotype lists longer than the batched call returns; when the
    selection runs out of ranked candidates it must draw from the FULL lists (functions.R:2278-2281), as the reference does."""
    from quilt_amd.driver import Driver, DriverParams
    from quilt_amd.synth import make_synthetic_panel, make_synthetic_sample
    from tests.oracle_backend import OracleBackend
    from tests.util import panel_from_rhb
    base = make_synthetic_panel(K=40, nSNPs=320, seed=8)
    rhb = np.asfortranarray(np.tile(base.rhb_t, (12, 1)))          # every haplotype 12 times: 480 haplotypes
    panel = panel_from_rhb(rhb, base.transMatRate_t, 320, 255, base.ref_error)
    panel.L_grid = base.L_grid
    samples = [make_synthetic_sample(panel, seed=70, n_reads=80)]
    prm = DriverParams(nGibbsSamples=1, Ksubset=64, Knew=64, seed=4)

    class Narrow(OracleBackend):
        pass

    drv = Driver(panel, Narrow(panel), prm)
    res = drv.run(samples)
    assert drv.n_full_list_refetches > 0, "the test panel is meant to produce truncated lists and an exhausted selection"
    # the same run with lists wide enough never to truncate gives the same result
    wide = Driver(panel, OracleBackend(panel), prm)
    wide.top_width = 600
    res_wide = wide.run(samples)
    assert wide.n_full_list_refetches == 0
    assert np.array_equal(res[0].read_labels, res_wide[0].read_labels)
    assert np.array_equal(res[0].dosage, res_wide[0].dosage)


def test_underflowing_input_is_retried_by_the_cpu_path(small_panel):
    """The input tests/test_pipeline_gpu.py::test_underflow_retry_on_the_device runs on the device does underflow in the
    oracle at 1e10, and the retry loop (functions.R:2704-2715) brings it through."""
    from quilt_amd.driver import Driver, DriverParams
    from tests.oracle_backend import OracleBackend
    from tests.util import underflowing_sample
    s = underflowing_sample(small_panel)
    drv = Driver(small_panel, OracleBackend(small_panel), DriverParams(nGibbsSamples=1, Ksubset=64, Knew=64, seed=6))
    res = drv.run([s])
    assert drv.n_underflow_retries > 0
    assert np.isfinite(res[0].dosage).all()


def _bam_to_vcf(tmp_path, panel, backend, method="diploid", n_samples=3, n_reads=300, ff=None, prm=None):
    """Synthetic samples -> BAM files -> loader -> driver -> VCF; returns (parsed VCF rows, run record, truth dosages)."""
    import gzip
    from quilt_amd.driver import DriverParams
    from quilt_amd.io import impute_bams_to_vcf
    from quilt_amd.synth import make_synthetic_sample
    from tests import bamutil
    rng = np.random.default_rng(21)
    T = panel.nSNPs
    alleles = [tuple(rng.choice(list("ACGT"), size=2, replace=False)) for _ in range(T)]
    ref, alt = [a for a, _ in alleles], [b for _, b in alleles]
    bams, truth = [], []
    for i in range(n_samples):
        s = make_synthetic_sample(panel, seed=300 + i, n_reads=n_reads, ff=(ff or 0.0))
        path = str(tmp_path / f"s{i}.bam")
        bamutil.write_bam(path, [("chr20", int(panel.L[-1]) + 1000)], bamutil.sample_to_alignments(s, panel.L, ref, alt, rng))
        bams.append(path)
        truth.append(s.truth_haps[:2].sum(axis=0).astype(float))
    empty = str(tmp_path / "empty.bam")                     # a sample without reads in the region: written as missing
    bamutil.write_bam(empty, [("chr20", int(panel.L[-1]) + 1000)], [])
    bams.insert(1, empty)
    names = [f"NA{i}" for i in range(len(bams))]
    if prm is None:
        prm = DriverParams(nGibbsSamples=2, Ksubset=64, Knew=64, seed=9, method=method)
    out = str(tmp_path / "quilt.vcf.gz")
    rec = impute_bams_to_vcf(panel, backend, bams, names, "chr20", ref, alt, out, params=prm,
                             ff=None if ff is None else [ff] * len(bams))
    lines = [l for l in gzip.open(out, "rt").read().split("\n") if l and not l.startswith("##")]
    assert lines[0].split("\t")[9:] == names
    rows = [l.split("\t") for l in lines[1:]]
    assert len(rows) == T
    return rows, rec, truth


def test_bam_to_vcf_end_to_end_on_the_cpu_path(tmp_path, small_panel):
    """f3 -> the driver (on the oracle backend) -> f4: dosages in the file are the driver's, rounded to three decimals; the
    sample without reads is '.'; imputed dosages track the truth."""
    from tests.oracle_backend import OracleBackend
    from tests.util import r2
    rows, rec, truth = _bam_to_vcf(tmp_path, small_panel, OracleBackend(small_panel))
    assert set(rec["results"]) == {0, 2, 3} and rec["columns"][1] is None
    for col, i_truth in ((0, 0), (2, 1), (3, 2)):
        ds = np.array([float(r[9 + col].split(":")[2]) for r in rows])
        res = rec["results"][col]
        assert np.abs(ds - (res.gp_t[1] + 2 * res.gp_t[2])).max() <= 5.1e-4
        assert r2(ds, truth[i_truth]) > 0.8
        gt = rows[0][9 + col].split(":")[0]
        assert len(gt) == 3 and gt[1] == "|"
    assert all(r[9 + 1] == "./.:.,.,.:.:.,." for r in rows)
    assert rows[5][8] == "GT:GP:DS:HD" and rows[5][7].startswith("EAF=")


def test_native_consensus_equals_the_numpy_text():
    """csrc/hostio.cpp qa_consensus_read_labels against assess_ability_of_reads_to_be_confident + determine_best_read_label_so_far
    (and its NIPT wrapper) of quilt_amd/driver.py, themselves tested against the line-by-line form: random label matrices with
    switch points, flipped runs, unconfident reads, NaN likelihoods, fewer than 10 confident rows."""
    from quilt_amd.io import consensus_read_labels
    rng = np.random.default_rng(17)
    for trial in range(60):
        n = int(rng.choice([1, 2, 3, 7]))
        R = int(rng.choice([5, 40, 600]))
        K = int(rng.choice([2, 3]))
        base = rng.integers(1, 3, size=R)
        lab = np.empty((n, R), dtype=np.int32)
        for c in range(n):
            x = base.copy()
            for cut in rng.integers(0, R, size=int(rng.integers(0, 4))):      # runs with flipped labels
                x[cut:] = 3 - x[cut:]
            noise = rng.random(R) < 0.03
            x[noise] = 3 - x[noise]
            lab[c] = x
        if K == 3:
            lab[rng.random((n, R)) < 0.1] = 3
        p = rng.random((n, K, R)) ** 8
        p[:, :, rng.random(R) < 0.02] = 0.0                                     # 0 / 0 -> NaN -> 0.5 (1 / 3)
        if trial % 7 == 0:
            p[:] = 0.5                                                           # nothing confident
        can = int(rng.integers(1, n + 1))
        got = consensus_read_labels(lab, p, can_hap=can)
        conf = np.stack([D.assess_ability_of_reads_to_be_confident(p[c]) for c in range(n)], axis=1)
        fn = D.determine_best_read_label_so_far_nipt if K == 3 else D.determine_best_read_label_so_far
        ref = fn(lab.T.copy(), conf, R, n, can_hap=can)
        assert np.array_equal(got, ref), (trial, n, R, K, can)


def test_host_span_trace_records_the_phases_of_a_run(tmp_path, monkeypatch):
    """quilt_amd/trace.py: off by default; with a path set the driver's phases come out as (thread, name, start, end)."""
    import json
    from quilt_amd import trace
    assert not trace.enabled()
    with trace.span("nothing"):
        pass
    assert trace._spans == []
    out = tmp_path / "trace.json"
    monkeypatch.setattr(trace, "_PATH", str(out))
    monkeypatch.setattr(trace, "_spans", [])
    with trace.span("device:fake"):
        trace.add("inner", 1.0, 2.0)
    trace.mark("instant")
    trace.dump()
    got = json.load(open(out))
    names = [x[1] for x in got]
    assert names == ["inner", "device:fake", "instant"]
    assert all(x[3] >= x[2] for x in got)


def test_diploid_block_gibbs_is_a_named_choice():
    """the reference's diploid block pass never relabels; that is the only behaviour on offer and it has
    a name."""
    from quilt_amd.driver import DriverParams
    assert DriverParams().resolved(1000).diploid_block_gibbs == "reference_noop"
    with pytest.raises(ValueError, match="reference_noop"):
        DriverParams(diploid_block_gibbs="active").resolved(1000)


def test_pair_gate_lets_a_finished_thread_leave():
    """workers.PairGate: threads meet before their Gibbs launches; one that has no launches left (uneven number of batches)
    leaves, and the remaining thread no longer waits out the timeout."""
    import threading
    import time
    from quilt_amd.workers import PairGate
    gate = PairGate(2, timeout=5.0)
    met = []
    th = threading.Thread(target=lambda: met.append(gate.wait()))
    th.start()
    assert gate.wait() is True
    th.join()
    assert met == [True]
    gate.leave()                      # the other thread is done
    t0 = time.perf_counter()
    assert gate.wait() is True        # alone now: no waiting
    assert time.perf_counter() - t0 < 1.0
    gate.reset(2)
    t0 = time.perf_counter()
    short = PairGate(2, timeout=0.2)
    assert short.wait() is False and time.perf_counter() - t0 >= 0.19   # nobody came: goes alone after the timeout


def _threads_over_streams(panel, prm, streams, tail, fail_in=None, rare_common=None, fail_in_tail_rounds=False):
    """One Driver (oracle backend) per stream, each on its own host thread, sharing ``tail`` -- what DeviceWorkers does.
    ``fail_in_tail_rounds``: whichever thread runs a phasing-only round (the fused tail) fails inside it."""
    import threading
    from tests.oracle_backend import OracleBackend
    out, err = [None] * len(streams), [None] * len(streams)

    def work(w):
        try:
            if rare_common is not None:
                drv = D.Driver(panel, OracleBackend(panel, rare_common), prm, rare_common=rare_common)
            else:
                drv = D.Driver(panel, OracleBackend(panel), prm)
            drv.phasing_tail = tail
            if fail_in == w:
                def boom(*a, **k):
                    raise RuntimeError("injected")
                drv._round = boom
            if fail_in_tail_rounds:
                orig = drv._round

                def boom_in_tail(chains, i_it):
                    if all(ch.phasing for ch in chains):
                        raise RuntimeError("injected in the tail rounds")
                    return orig(chains, i_it)
                drv._round = boom_in_tail
            out[w] = list(drv.run_stream(streams[w]))
        except BaseException as e:
            err[w] = e

    th = [threading.Thread(target=work, args=(w,), daemon=True) for w in range(len(streams))]
    for t in th:
        t.start()
    for t in th:
        t.join(120)
        assert not t.is_alive(), "a host thread is still waiting at the tail"
    return out, err


@pytest.mark.parametrize("shape", ["4 batches over 3 threads", "2 batches over 3 threads", "even"])
def test_last_batches_phasing_rounds_fused_across_threads(shape):
    """PhasingTail: the threads' last batches leave their phasing rounds to the thread that drains last; per-sample results
    are those of separate runs, every batch comes back through its own thread's stream."""
    from quilt_amd.synth import make_synthetic_panel, make_synthetic_sample
    from tests.oracle_backend import OracleBackend
    panel = make_synthetic_panel(K=400, nSNPs=320, seed=22)
    samples = [make_synthetic_sample(panel, seed=70 + i, n_reads=50) for i in range(6)]
    prm = D.DriverParams(nGibbsSamples=2, Ksubset=64, Knew=64, seed=4)
    if shape == "4 batches over 3 threads":
        streams = [[(samples[0:2], 0), (samples[4:5], 4)], [(samples[2:3], 2)], [(samples[3:4], 3)]]
    elif shape == "2 batches over 3 threads":
        streams = [[(samples[0:2], 0)], [(samples[2:5], 2)], []]
    else:
        streams = [[(samples[0:3], 0)], [(samples[3:6], 3)]]
    tail = D.PhasingTail(len(streams))
    seen_rounds = []
    orig = D.Driver._round

    def spy(self, chains, i_it):
        seen_rounds.append((i_it, sum(ch.phasing for ch in chains), len(chains)))
        return orig(self, chains, i_it)
    D.Driver._round = spy
    try:
        out, err = _threads_over_streams(panel, prm, streams, tail)
    finally:
        D.Driver._round = orig
    assert all(e is None for e in err), err
    for st, got in zip(streams, out):
        assert [len(b) for b in got] == [len(s) for s, _ in st]
        for (smp, off), res in zip(st, got):
            ref = D.Driver(panel, OracleBackend(panel), prm).run(smp, sample_offset=off)
            for g, r in zip(res, ref):
                assert np.array_equal(g.read_labels, r.read_labels) and g.nDosage == r.nDosage
                assert np.array_equal(g.dosage, r.dosage) and np.array_equal(g.phasing_haps, r.phasing_haps)
                assert np.array_equal(g.gp_t, r.gp_t)
    # exactly one set of phasing-only rounds, carrying the last batch of every thread that had one
    only = [r for r in seen_rounds if r[1] == r[2]]
    n_last = sum(len(st[-1][0]) for st in streams if st)
    assert sorted(only) == [(i, n_last, n_last) for i in (1, 2, 3)]


def test_tail_failure_reaches_the_waiting_threads():
    """A thread that fails must not leave the others waiting for rounds nobody will run."""
    from quilt_amd.synth import make_synthetic_panel, make_synthetic_sample
    panel = make_synthetic_panel(K=400, nSNPs=320, seed=22)
    samples = [make_synthetic_sample(panel, seed=70 + i, n_reads=50) for i in range(3)]
    prm = D.DriverParams(nGibbsSamples=2, Ksubset=64, Knew=64, seed=4)
    streams = [[(samples[0:1], 0)], [(samples[1:2], 1), (samples[2:3], 2)]]
    out, err = _threads_over_streams(panel, prm, streams, D.PhasingTail(2), fail_in=1)
    assert isinstance(err[1], RuntimeError) and isinstance(err[0], RuntimeError)
    # ... and a consumer that stops early counts as drained
    tail = D.PhasingTail(2)
    from tests.oracle_backend import OracleBackend
    drv = D.Driver(panel, OracleBackend(panel), prm)
    drv.phasing_tail = tail
    g = drv.run_stream([(samples[0:1], 0), (samples[1:2], 1)])
    next(g)
    g.close()
    assert tail.n_active == 1


def test_failure_inside_the_fused_tail_rounds_reaches_the_owners():
    """The thread that drains last holds the other threads' last batches only in its local list: when it fails INSIDE the
    fused phasing rounds, the owners (blocked on their futures) must get the failure, not wait for ever."""
    from quilt_amd.synth import make_synthetic_panel, make_synthetic_sample
    panel = make_synthetic_panel(K=400, nSNPs=320, seed=22)
    samples = [make_synthetic_sample(panel, seed=70 + i, n_reads=50) for i in range(4)]
    prm = D.DriverParams(nGibbsSamples=2, Ksubset=64, Knew=64, seed=4)
    streams = [[(samples[0:1], 0)], [(samples[1:2], 1), (samples[2:3], 2)], [(samples[3:4], 3)]]
    out, err = _threads_over_streams(panel, prm, streams, D.PhasingTail(3), fail_in_tail_rounds=True)
    assert all(isinstance(e, RuntimeError) for e in err), err


@pytest.mark.parametrize("n_batches", [4, 5, 2, -5])
def test_workers_cut_the_left_over_batches_into_parts(n_batches):
    """DeviceWorkers.run_stream, split = "alternate": whole batches in turn, the batches left over by the thread count cut
    into one part per thread (a part may be empty); every batch comes back whole, in order, with the results of separate
    runs.  (The workers object is built around oracle-backed drivers: the threading logic needs no device.)"""
    from quilt_amd.synth import make_synthetic_panel, make_synthetic_sample
    from quilt_amd.workers import DeviceWorkers
    from tests.oracle_backend import OracleBackend
    panel = make_synthetic_panel(K=400, nSNPs=320, seed=23)
    samples = [make_synthetic_sample(panel, seed=90 + i, n_reads=40) for i in range(9)]
    prm = D.DriverParams(nGibbsSamples=2, Ksubset=64, Knew=64, seed=6)
    cuts = {4: [0, 2, 3, 5, 9], 5: [0, 2, 3, 5, 7, 9], 2: [0, 4, 9]}[abs(n_batches)]
    batches = [(samples[a:b], a) for a, b in zip(cuts[:-1], cuts[1:])]
    if n_batches < 0:       # the last batch's samples do not follow the one before it: the left-overs are cut one by one
        batches[-1] = (batches[-1][0], 100)
    wk = object.__new__(DeviceWorkers)
    wk.n, wk.split, wk.fuse_tails, wk.split_remainder = 3, "alternate", True, True
    if n_batches == 2:      # (this shape also through split = "halves": every batch cut into parts)
        wh = object.__new__(DeviceWorkers)
        wh.n, wh.split, wh.fuse_tails, wh.split_remainder = 3, "halves", True, True
        wh.drivers = [D.Driver(panel, OracleBackend(panel), prm) for _ in range(3)]
        halves = list(wh.run_stream(batches))
        assert [len(g) for g in halves] == [len(s) for s, _ in batches]
    wk.drivers = [D.Driver(panel, OracleBackend(panel), prm) for _ in range(3)]
    taken = [[] for _ in range(3)]
    for w, d in enumerate(wk.drivers):
        orig = d._new_batch
        d._new_batch = (lambda samples, offset, _o=orig, _w=w: (taken[_w].append((offset, len(samples))), _o(samples, offset))[1])
    got = list(wk.run_stream(batches))
    assert [len(g) for g in got] == [len(s) for s, _ in batches]
    if n_batches == 4:      # three whole batches in turn, the fourth (4 samples at offset 5) cut 1 + 2 + 1 (sharding.get_sample_range)
        assert taken == [[(0, 2), (5, 1)], [(2, 1), (6, 2)], [(3, 2), (8, 1)]]
    elif n_batches == 5:    # two left over, consecutive samples: cut as one run of four samples, 1 + 2 + 1
        assert taken == [[(0, 2), (5, 1)], [(2, 1), (6, 2)], [(3, 2), (8, 1)]]
    elif n_batches == -5:   # two left over, not consecutive: 2 samples each over three threads, 1 + 1 + 0
        assert taken == [[(0, 2), (5, 1), (100, 1)], [(2, 1), (6, 1), (101, 1)], [(3, 2)]]
    else:                   # fewer batches than threads: nothing to cut
        assert taken == [[(0, 4)], [(4, 5)], []]
    for bi, ((smp, off), res) in enumerate(zip(batches, got)):
        ref = D.Driver(panel, OracleBackend(panel), prm).run(smp, sample_offset=off)
        for si, (g, r) in enumerate(zip(res, ref)):
            assert np.array_equal(g.read_labels, r.read_labels)
            assert np.array_equal(g.dosage, r.dosage) and np.array_equal(g.phasing_haps, r.phasing_haps)
            if n_batches == 2:
                assert np.array_equal(halves[bi][si].dosage, r.dosage) and np.array_equal(halves[bi][si].read_labels, r.read_labels)


@pytest.mark.parametrize("mode", ["nipt", "rare_common"])
def test_fused_phasing_tail_in_the_other_modes(mode):
    """The threads' last batches run their phasing rounds together also with three read labels (mother / fetus accumulators,
    NIPT recast) and with impute_rare_common (the all-SNP Gibbs call after the rounds takes the other threads' phasing chains
    too): results of separate runs."""
    from quilt_amd.synth import (make_rare_common, make_synthetic_panel, make_synthetic_sample,
                                 make_synthetic_sample_rare_common)
    from tests.oracle_backend import OracleBackend
    rc = None
    if mode == "nipt":
        panel = make_synthetic_panel(K=400, nSNPs=640, seed=21)
        samples = [make_synthetic_sample(panel, seed=70 + i, n_reads=120, ff=0.15 + 0.05 * i) for i in range(4)]
        prm = D.DriverParams(nGibbsSamples=2, Ksubset=64, Knew=64, seed=9, method="nipt")
    else:
        panel = make_synthetic_panel(K=400, nSNPs=320, seed=21)
        rc = make_rare_common(panel, 3)
        samples = [make_synthetic_sample_rare_common(panel, rc, 50 + i, n_reads=100)[0] for i in range(4)]
        prm = D.DriverParams(nGibbsSamples=2, Ksubset=64, Knew=64, seed=9, impute_rare_common=True)
    streams = [[(samples[0:1], 0), (samples[3:4], 3)], [(samples[1:3], 1)]]
    out, err = _threads_over_streams(panel, prm, streams, D.PhasingTail(2), rare_common=rc)
    assert all(e is None for e in err), err
    for st, got in zip(streams, out):
        for (smp, off), res in zip(st, got):
            if rc is not None:
                ref = D.Driver(panel, OracleBackend(panel, rc), prm, rare_common=rc).run(smp, sample_offset=off)
            else:
                ref = D.Driver(panel, OracleBackend(panel), prm).run(smp, sample_offset=off)
            assert len(res) == len(ref)
            for g, r in zip(res, ref):
                assert np.array_equal(g.read_labels, r.read_labels) and g.nDosage == r.nDosage
                assert np.array_equal(g.dosage, r.dosage) and np.array_equal(g.phasing_haps, r.phasing_haps)
                if mode == "nipt":
                    assert np.array_equal(g.fet_dosage, r.fet_dosage) and np.array_equal(g.fet_gp_t, r.fet_gp_t)


def test_phred_eps_is_the_c_librarys_pow():
    import math
    from quilt_amd.driver import phred_eps
    q = np.arange(-93, 94)
    assert all(float(e) == math.pow(10.0, -abs(int(v)) / 10.0) for e, v in zip(phred_eps(q), q))

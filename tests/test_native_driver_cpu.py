"""The native per-sample driver loop (csrc/impute.cpp behind qa_impute_samples) without a device: the product's host code run
through qa_impute_samples_backend over the CPU oracle's entry points (tests/native_driver_backend.py) must equal, bit for bit,
(i) quilt_amd/driver.py on the oracle backend and (ii) the literal restatement of the R loop nest (tests/r_driver_twin.py) --
whatever the launch-set size, the number of host threads or the handling of the stream's end."""
import numpy as np
import pytest

from tests.oracle_backend import OracleBackend


@pytest.fixture(scope="module")
def twin_panel():
    from quilt_amd.synth import make_synthetic_panel
    return make_synthetic_panel(K=400, nSNPs=3200, seed=77, ref_error=1e-3)


def _same(a, b):
    assert a.nDosage == b.nDosage
    assert np.array_equal(a.read_labels, b.read_labels), "consensus read labels"
    assert np.array_equal(a.dosage, b.dosage), "dosage"
    assert np.array_equal(a.gp_t, b.gp_t), "genotype posteriors"
    assert np.array_equal(a.phasing_haps, b.phasing_haps), "phased haplotypes"


def test_chain_stream_matches_numpy_text():
    """csrc/impute.cpp's ChainStream == quilt_amd/rng.py::ChainStream: seen through the first-round draws (the small panel's
    rows and the starting labels decide everything downstream), here directly on the rule for subsets."""
    from quilt_amd.rng import ChainStream, keyed_subset, stream_u64
    s = ChainStream(7, 3, 2)
    a = s.choice(1000, 50, replace=False)
    assert len(set(a.tolist())) == 50 and s.ctr == 1000
    assert np.array_equal(a, keyed_subset(s.key, 1000, 50, 0))
    x = s.integers(0, 2 ** 63)
    assert 0 <= x < 2 ** 63 and x == int(np.floor((int(stream_u64(s.key, 1, 1000)[0]) >> 11) / 2.0 ** 53 * 2.0 ** 63))
    lab = s.integers(1, 3, size=4000)
    assert set(lab.tolist()) == {1, 2} and abs(lab.mean() - 1.5) < 0.05


@pytest.mark.parametrize("kw", [
    dict(nGibbsSamples=7, n_seek_its=3, Ksubset=64, Knew=64),
    dict(nGibbsSamples=3, n_seek_its=3, Ksubset=64, Knew=24),
    dict(nGibbsSamples=4, n_seek_its=2, n_burn_in_seek_its=0, Ksubset=48, Knew=48),
    dict(nGibbsSamples=2, n_seek_its=3, Ksubset=200, Knew=200, K_top_matches=1, heuristic_match_thin=0.03),
], ids=["defaults", "Knew<Ksubset", "no-burn-in", "exhausted-ranks"])
def test_native_loop_equals_python_driver_and_r_twin(twin_panel, kw):
    from quilt_amd.driver import Driver, DriverParams
    from quilt_amd.synth import make_synthetic_sample
    from tests.native_driver_backend import impute_samples_on_oracle
    from tests.r_driver_twin import get_and_impute_one_sample
    panel = twin_panel
    samples = [make_synthetic_sample(panel, seed=300 + i, n_reads=260) for i in range(3)]
    common = dict(small_ref_panel_gibbs_iterations=6, small_ref_panel_block_gibbs_iterations=(1, 3), seed=11)
    P = DriverParams(**common, **kw)
    want = Driver(panel, OracleBackend(panel), P).run(samples, sample_offset=5)
    got, stats, tab = impute_samples_on_oracle(panel, samples, P, sample_offset=5, samples_per_launch_set=2)
    for a, b in zip(got, want):
        _same(a, b)
    tw = get_and_impute_one_sample(panel, samples[1], 6, **common, **kw)
    np.testing.assert_allclose(got[1].dosage, tw["dosage"], rtol=0, atol=1e-13)
    assert np.array_equal(got[1].read_labels, tw["read_labels"])
    np.testing.assert_allclose(got[1].phasing_haps, tw["phasing_haps"], rtol=0, atol=1e-12)
    if kw.get("K_top_matches") == 1:   # the complete-lists branch (functions.R:2276-2302) ran in the native loop
        assert stats["full_list_refetches"] > 0 and tab.calls["fullpass"] == stats["full_list_refetches"]
    assert stats["gibbs_chain_calls"] >= len(samples) * (P.nGibbsSamples + 1) * P.n_seek_its


@pytest.mark.parametrize("n_threads,per_set,fuse", [(1, 256, True), (1, 2, True), (3, 2, True), (3, 2, False), (2, 3, True), (4, 1, True)])
def test_native_loop_is_independent_of_threads_and_launch_sets(twin_panel, n_threads, per_set, fuse):
    from quilt_amd.driver import Driver, DriverParams
    from quilt_amd.synth import make_synthetic_sample
    from tests.native_driver_backend import impute_samples_on_oracle
    panel = twin_panel
    samples = [make_synthetic_sample(panel, seed=500 + i, n_reads=150 + 10 * i) for i in range(7)]
    P = DriverParams(nGibbsSamples=3, n_seek_its=2, Ksubset=64, Knew=40, small_ref_panel_gibbs_iterations=4,
                     small_ref_panel_block_gibbs_iterations=(2,), seed=3)
    want = Driver(panel, OracleBackend(panel), P).run(samples, sample_offset=10)
    got, stats, _ = impute_samples_on_oracle(panel, samples, P, sample_offset=10, samples_per_launch_set=per_set,
                                             n_threads=n_threads, fuse_tails=fuse)
    for a, b in zip(got, want):
        _same(a, b)
    # pipelining: a set's phasing rounds share launches with the next set's main rounds
    if n_threads == 1 and per_set == 2:
        assert stats["gibbs_launches"] == (4 + 1) * P.n_seek_its


def test_native_loop_mspbwt_mode(twin_panel):
    """use_mspbwt = TRUE: no full-panel pass; the next small panel from the neighbour scan of the Gibbs call's rounded haploid
    dosages (the library's host index), the dosages from the Gibbs call itself."""
    from quilt_amd.driver import Driver, DriverParams
    from quilt_amd.synth import make_synthetic_sample
    from tests.native_driver_backend import impute_samples_on_oracle
    panel = twin_panel
    samples = [make_synthetic_sample(panel, seed=700 + i, n_reads=220) for i in range(4)]
    P = DriverParams(nGibbsSamples=3, n_seek_its=3, Ksubset=64, Knew=64, small_ref_panel_gibbs_iterations=5,
                     small_ref_panel_block_gibbs_iterations=(2,), seed=9, use_mspbwt=True, mspbwt_nindices=2)
    want = Driver(panel, OracleBackend(panel), P).run(samples, sample_offset=0)
    got, stats, tab = impute_samples_on_oracle(panel, samples, P, samples_per_launch_set=3, n_threads=2)
    assert tab.calls["select"] == 0 and tab.calls["fullpass"] == 0
    for a, b in zip(got, want):
        _same(a, b)


def test_native_loop_reports_failures(twin_panel):
    """A hard error of an entry point inside a worker thread ends the call with that status and its text; the other
    threads -- including one waiting for the fused tail rounds -- are released."""
    from quilt_amd.driver import DriverParams
    from quilt_amd.synth import make_synthetic_sample
    from tests.native_driver_backend import impute_samples_on_oracle
    panel = twin_panel
    samples = [make_synthetic_sample(panel, seed=800 + i, n_reads=120) for i in range(6)]
    P = DriverParams(nGibbsSamples=2, n_seek_its=2, Ksubset=48, Knew=48, small_ref_panel_gibbs_iterations=3,
                     small_ref_panel_block_gibbs_iterations=(1,), seed=5)
    for at in (1, 4, 7):
        with pytest.raises(RuntimeError, match="status -4"):
            impute_samples_on_oracle(panel, samples, P, samples_per_launch_set=1, n_threads=3, fail_at_call=("gibbs", at))
    with pytest.raises(RuntimeError, match="no reads"):
        from tests.native_driver_backend import _Reads
        z = np.zeros(0, dtype=np.int32)
        empty = _Reads(np.zeros(1, dtype=np.int32), z, z, z)
        impute_samples_on_oracle(panel, [samples[0], empty], P)


@pytest.mark.parametrize("mspbwt,n_threads", [(False, 1), (True, 2)])
def test_native_loop_rare_common(mspbwt, n_threads):
    """impute_rare_common (functions.R:1042-1123, rare_common.R:61-420) in the native loop: every Gibbs sample and the phasing
    iteration end with the all-SNP Gibbs call, started from labels drawn against the latest haploid dosages spread over all SNPs;
    accumulators, phased haplotypes and nDosage cover all SNPs -- equal to quilt_amd/driver.py on the oracle."""
    from quilt_amd.driver import Driver, DriverParams
    from quilt_amd.synth import make_rare_common, make_synthetic_panel, make_synthetic_sample_rare_common
    from tests.native_driver_backend import impute_samples_on_oracle
    panel = make_synthetic_panel(K=300, nSNPs=640, seed=5)
    rc = make_rare_common(panel, 3)
    samples = [make_synthetic_sample_rare_common(panel, rc, 50 + i, n_reads=150)[0] for i in range(4)]
    P = DriverParams(nGibbsSamples=2, Ksubset=64, Knew=64, seed=9, impute_rare_common=True, small_ref_panel_gibbs_iterations=5,
                     small_ref_panel_block_gibbs_iterations=(2,), use_mspbwt=mspbwt, mspbwt_nindices=2)
    want = Driver(panel, OracleBackend(panel, rc), P, rare_common=rc).run(samples, sample_offset=3)
    got, stats, tab = impute_samples_on_oracle(panel, samples, P, sample_offset=3, samples_per_launch_set=2, n_threads=n_threads,
                                               rare_common=rc)
    assert tab.calls["gibbs_rc"] > 0 and tab.calls["emat_all"] == tab.calls["gibbs_rc"]
    for a, b in zip(got, want):
        assert a.dosage.shape == (rc.nSNPs_all,) and a.nDosage == b.nDosage == P.nGibbsSamples
        _same(a, b)


@pytest.mark.parametrize("n_threads", [1, 2])
def test_native_loop_nipt_rare_common(n_threads):
    """method = "nipt" with impute_rare_common: the all-SNP call's starting labels come from the read grouping by which of
    (hap1, hap2, hap3) a read fits (get_read_groupings_given_fetal_fraction_and_cov / sample_H_for_NIPT_given_groupings,
    gibbs-nipt.R:1655-1849), its block passes are cut on the all-SNP grid, the mother's and the fetus' accumulators and the three
    phased haplotypes cover all SNPs -- equal to quilt_amd/driver.py on the oracle."""
    from quilt_amd.driver import Driver, DriverParams
    from quilt_amd.synth import make_rare_common, make_synthetic_panel, make_synthetic_sample_rare_common
    from tests.native_driver_backend import impute_samples_on_oracle
    panel = make_synthetic_panel(K=300, nSNPs=640, seed=5)
    rc = make_rare_common(panel, 3)
    samples = []
    for i in range(3):
        s = make_synthetic_sample_rare_common(panel, rc, 70 + i, n_reads=160)[0]
        s.ff = 0.1 + 0.07 * i
        samples.append(s)
    P = DriverParams(nGibbsSamples=2, n_seek_its=2, Ksubset=64, Knew=64, seed=11, method="nipt", impute_rare_common=True,
                     small_ref_panel_gibbs_iterations=5, small_ref_panel_block_gibbs_iterations=(2,))
    want = Driver(panel, OracleBackend(panel, rc), P, rare_common=rc).run(samples, sample_offset=1)
    got, stats, tab = impute_samples_on_oracle(panel, samples, P, sample_offset=1, samples_per_launch_set=2, n_threads=n_threads,
                                               rare_common=rc)
    assert tab.calls["gibbs_rc"] > 0
    for a, b in zip(got, want):
        assert a.phasing_haps.shape == (rc.nSNPs_all, 3) and a.fet_dosage.shape == (rc.nSNPs_all,)
        _same(a, b)
        assert np.array_equal(a.fet_dosage, b.fet_dosage) and np.array_equal(a.fet_gp_t, b.fet_gp_t)
    assert len({tuple(np.unique(a.read_labels)) for a in got} & {(1, 2, 3)}) == 1


@pytest.mark.parametrize("n_threads,per_set", [(1, 256), (2, 1)])
def test_native_loop_nipt(twin_panel, n_threads, per_set):
    """method = "nipt" in the native loop: three read labels drawn with the sample's fetal fraction (functions.R:586), one ff per
    chain in the Gibbs call with its block passes, three full-panel passes per chain, mother / fetus accumulators
    (:1009-1016), three-way read confidence and consensus labels (:1788-1829), recast_nipt_haps (:3214-3287) -- equal to
    quilt_amd/driver.py on the oracle, samples with different fetal fractions in one launch set."""
    from quilt_amd.driver import Driver, DriverParams
    from quilt_amd.synth import make_synthetic_sample
    from tests.native_driver_backend import impute_samples_on_oracle
    panel = twin_panel
    samples = [make_synthetic_sample(panel, seed=900 + i, n_reads=220, ff=0.1 + 0.05 * i) for i in range(3)]
    P = DriverParams(nGibbsSamples=2, n_seek_its=2, Ksubset=64, Knew=64, seed=4, method="nipt", small_ref_panel_gibbs_iterations=5,
                     small_ref_panel_block_gibbs_iterations=(2,))
    want = Driver(panel, OracleBackend(panel), P).run(samples, sample_offset=2)
    got, stats, _ = impute_samples_on_oracle(panel, samples, P, sample_offset=2, samples_per_launch_set=per_set, n_threads=n_threads)
    for a, b in zip(got, want):
        assert a.phasing_haps.shape == (panel.nSNPs, 3) and (a.read_labels == 3).any()
        _same(a, b)
        assert np.array_equal(a.fet_dosage, b.fet_dosage) and np.array_equal(a.fet_gp_t, b.fet_gp_t)


@pytest.mark.parametrize("n_threads,per_set,n_upper", [(1, 256, 7), (1, 2, 7), (3, 2, 10), (2, 3, 9), (3, 1, 16)])
def test_samples_handed_over_one_by_one_equal_the_flat_call(twin_panel, n_threads, per_set, n_upper):
    """params->sample_source (qa_sample_source_t): the reads handed over when the launch set holding the sample is taken, the call's
    n_sample an UPPER BOUND the source ends early (a caller that learns which samples it keeps while loading them).  Same bytes as
    the flat call, whatever the threads and the launch sets; every sample is acquired once, in ascending order within its set."""
    from quilt_amd.driver import DriverParams
    from quilt_amd.synth import make_synthetic_sample
    from tests.native_driver_backend import impute_samples_on_oracle
    panel = twin_panel
    samples = [make_synthetic_sample(panel, seed=500 + i, n_reads=150 + 10 * i) for i in range(7)]
    P = DriverParams(nGibbsSamples=3, n_seek_its=2, Ksubset=64, Knew=40, small_ref_panel_gibbs_iterations=4,
                     small_ref_panel_block_gibbs_iterations=(2,), seed=3)
    want, _, _ = impute_samples_on_oracle(panel, samples, P, sample_offset=10, samples_per_launch_set=per_set, n_threads=n_threads)
    log = []
    got, stats, _ = impute_samples_on_oracle(panel, samples, P, sample_offset=10, samples_per_launch_set=per_set, n_threads=n_threads,
                                             source=dict(n_upper=n_upper, order_log=log))
    for a, b in zip(got, want):
        _same(a, b)
    served = [s for s in log if s < len(samples)]
    assert sorted(served) == list(range(len(samples))), log      # once each
    for lo in range(0, len(samples), per_set):                     # ascending within a launch set
        mine = [s for s in served if lo <= s < lo + per_set]
        assert mine == sorted(mine)
    assert all(s <= n_upper - 1 for s in log)
    assert stats["gibbs_chain_calls"] >= len(samples) * (P.nGibbsSamples + 1) * P.n_seek_its


def test_sample_source_modes_and_failures(twin_panel):
    """The source with method = "nipt" (the fetal fraction of a sample read after it was acquired) and with impute_rare_common (the
    all-SNP reads in the same view); a source that fails ends the call with its status; a range that ends before its first
    sample is an empty, successful call."""
    from quilt_amd.driver import DriverParams
    from quilt_amd.synth import make_rare_common, make_synthetic_panel, make_synthetic_sample, make_synthetic_sample_rare_common
    from tests.native_driver_backend import impute_samples_on_oracle
    panel = twin_panel
    samples = [make_synthetic_sample(panel, seed=900 + i, n_reads=200, ff=0.1 + 0.05 * i) for i in range(3)]
    P = DriverParams(nGibbsSamples=2, n_seek_its=2, Ksubset=64, Knew=64, seed=4, method="nipt", small_ref_panel_gibbs_iterations=4,
                     small_ref_panel_block_gibbs_iterations=(2,))
    want, _, _ = impute_samples_on_oracle(panel, samples, P, sample_offset=2, samples_per_launch_set=2, n_threads=2)
    got, _, _ = impute_samples_on_oracle(panel, samples, P, sample_offset=2, samples_per_launch_set=2, n_threads=2, source=dict(n_upper=5))
    for a, b in zip(got, want):
        _same(a, b)
        assert np.array_equal(a.fet_dosage, b.fet_dosage) and np.array_equal(a.fet_gp_t, b.fet_gp_t)
    P2 = DriverParams(nGibbsSamples=2, n_seek_its=2, Ksubset=48, Knew=48, small_ref_panel_gibbs_iterations=3,
                      small_ref_panel_block_gibbs_iterations=(1,), seed=5)
    with pytest.raises(RuntimeError, match="status -2.*sample source failed at sample 2"):
        impute_samples_on_oracle(panel, samples, P2, samples_per_launch_set=1, n_threads=3, source=dict(fail_at=2))
    none, stats, tab = impute_samples_on_oracle(panel, [], P2, n_threads=2, source=dict(n_upper=4))
    assert none == [] and tab.calls["gibbs"] == 0
    small = make_synthetic_panel(K=300, nSNPs=640, seed=5)
    rc = make_rare_common(small, 3)
    rs = [make_synthetic_sample_rare_common(small, rc, 50 + i, n_reads=150)[0] for i in range(3)]
    P3 = DriverParams(nGibbsSamples=2, Ksubset=64, Knew=64, seed=9, impute_rare_common=True, small_ref_panel_gibbs_iterations=4,
                      small_ref_panel_block_gibbs_iterations=(2,))
    want, _, _ = impute_samples_on_oracle(small, rs, P3, sample_offset=3, samples_per_launch_set=2, n_threads=2, rare_common=rc)
    got, _, _ = impute_samples_on_oracle(small, rs, P3, sample_offset=3, samples_per_launch_set=2, n_threads=2, rare_common=rc,
                                         source=dict(n_upper=4))
    for a, b in zip(got, want):
        _same(a, b)


def test_result_does_not_depend_on_the_plan_for_left_over_sets():
    """Seven launch sets of three samples over three host threads leave one over: it goes whole to the first thread (the default) or is cut across
    the threads (QA_IMPUTE_CUT_LEFTOVERS=1, read once per process: two child processes).  Same bytes either way."""
    import hashlib
    import os
    import subprocess
    import sys
    code = r'''
import hashlib, sys
sys.path.insert(0, %r)
import numpy as np
from quilt_amd.driver import DriverParams
from quilt_amd.synth import make_synthetic_panel, make_synthetic_sample
from tests.native_driver_backend import impute_samples_on_oracle
panel = make_synthetic_panel(K=200, nSNPs=320, seed=3)
samples = [make_synthetic_sample(panel, seed=700 + i, n_reads=60) for i in range(21)]
P = DriverParams(nGibbsSamples=2, n_seek_its=2, Ksubset=32, Knew=32, seed=5, small_ref_panel_gibbs_iterations=3,
                 small_ref_panel_block_gibbs_iterations=(1,))
got, stats, _ = impute_samples_on_oracle(panel, samples, P, samples_per_launch_set=3, n_threads=3)
h = hashlib.sha256()
for g in got:
    for a in (g.dosage, g.gp_t, np.ascontiguousarray(g.phasing_haps), g.read_labels):
        h.update(np.ascontiguousarray(a).tobytes())
print("HASH", h.hexdigest(), stats["gibbs_launches"])
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for cut in ("0", "1"):
        env = dict(os.environ, QA_IMPUTE_CUT_LEFTOVERS=cut)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        line = [x for x in r.stdout.splitlines() if x.startswith("HASH")][0].split()
        out[cut] = (line[1], int(line[2]))
    assert out["0"][0] == out["1"][0]
    assert out["1"][1] > out["0"][1]   # (the cut plan makes more, smaller launches)


@pytest.mark.parametrize("method", ["diploid", "nipt"])
def test_bam_range_call_equals_the_python_bam_to_vcf_path(tmp_path, small_panel, method):
    """qa_impute_bam_range (csrc/bamrange.cpp: BAM paths in, VCF columns and the range's count arrays out, one native call --
    what shim/quilt-amd.R's fast path calls) against quilt_amd.io.impute_bams_to_vcf (loader -> Python driver -> column writers ->
    SummaryCounts), both on the oracle's entry points: the same column TEXT for every sample, the unimputed sample reported as
    such, the same read labels, and the four count arrays (quilt.R:955-961) equal bit for bit.  The dropped file sits in the
    middle of the range: the kept samples' global indices are handed over one by one (qa_impute_params_t.sample_index), so its
    absence does not shift anyone's random streams."""
    from quilt_amd.driver import DriverParams
    from tests.native_driver_backend import impute_bam_range_on_oracle
    from tests.oracle_backend import OracleBackend
    from tests.test_driver_host import _bam_to_vcf
    ff = 0.2 if method == "nipt" else None
    prm = DriverParams(nGibbsSamples=2, Ksubset=64, Knew=64, seed=9, method=method)
    rows, rec, _ = _bam_to_vcf(tmp_path, small_panel, OracleBackend(small_panel), method=method, ff=ff, prm=prm)
    # the same files, alleles and parameters (test_driver_host._bam_to_vcf draws the alleles from this stream)
    rng = np.random.default_rng(21)
    alleles = [tuple(rng.choice(list("ACGT"), size=2, replace=False)) for _ in range(small_panel.nSNPs)]
    ref, alt = [a for a, _ in alleles], [b for _, b in alleles]
    bams = [str(tmp_path / n) for n in ("s0.bam", "empty.bam", "s1.bam", "s2.bam")]
    # impute_bams_to_vcf numbers the samples it keeps 0, 1, 2: the global indices of the kept files here (the dropped one's is unused)
    got = impute_bam_range_on_oracle(small_panel, bams, "chr20", ref, alt, prm, sample_index=[0, 99, 1, 2],
                                     ff=None if ff is None else [ff] * 4, n_io_threads=3, samples_per_launch_set=2)
    assert got["imputed"] == [True, False, True, True] and got["n_reads"][1] == 0 and got["columns"][1] is None
    for i in (0, 2, 3):
        assert got["columns"][i].tolist() == rec["columns"][i].tolist() == [r[9 + i] for r in rows]
        assert np.array_equal(got["results"][i].read_labels, rec["results"][i].read_labels)
        assert np.array_equal(got["results"][i].gp_t, rec["results"][i].gp_t)
    for name in ("infoCount", "afCount", "hweCount", "alleleCount"):
        assert np.array_equal(getattr(got["counts"], name), getattr(rec["counts"], name)), name
    # and with other global indices the draws differ (the indices are what keys them)
    other = impute_bam_range_on_oracle(small_panel, bams, "chr20", ref, alt, prm, sample_index=[5, 99, 6, 7],
                                       ff=None if ff is None else [ff] * 4)
    assert not all(np.array_equal(other["results"][i].read_labels, got["results"][i].read_labels) for i in (0, 2, 3))


def test_bam_range_call_at_its_edges(tmp_path, small_panel):
    """The range call where its bookkeeping could go wrong: dropped files first, last, all of them, next to each other; the same
    file twice; no file at all; one host thread and more threads than files; an unreadable file in the middle (the call fails
    naming it, whichever thread met it) -- every kept sample equal to itself in the plain four-file range."""
    from quilt_amd.driver import DriverParams
    from tests.native_driver_backend import impute_bam_range_on_oracle
    from tests.oracle_backend import OracleBackend
    from tests.test_driver_host import _bam_to_vcf
    prm = DriverParams(nGibbsSamples=2, Ksubset=64, Knew=64, seed=9)
    _bam_to_vcf(tmp_path, small_panel, OracleBackend(small_panel), prm=prm)   # (writes s0 / empty / s1 / s2 .bam)
    rng = np.random.default_rng(21)
    alleles = [tuple(rng.choice(list("ACGT"), size=2, replace=False)) for _ in range(small_panel.nSNPs)]
    ref, alt = [a for a, _ in alleles], [b for _, b in alleles]
    path = lambda n: str(tmp_path / (n + ".bam"))
    run = lambda names, idx, **kw: impute_bam_range_on_oracle(small_panel, [path(n) for n in names], "chr20", ref, alt, prm,
                                                              sample_index=idx, samples_per_launch_set=2, **kw)
    base = run(["s0", "empty", "s1", "s2"], [0, 1, 2, 3], n_io_threads=3)
    col = {"s0": base["columns"][0].tolist(), "s1": base["columns"][2].tolist(), "s2": base["columns"][3].tolist()}
    gidx = {"s0": 0, "s1": 2, "s2": 3, "empty": 1}

    def check(names, **kw):
        got = run(names, [gidx[n] for n in names], **kw)
        assert got["imputed"] == [n != "empty" for n in names]
        for i, n in enumerate(names):
            if n == "empty":
                assert got["columns"][i] is None and got["n_reads"][i] == 0 and i not in got["results"]
            else:
                assert got["columns"][i].tolist() == col[n], (names, i)
        assert got["counts"].hweCount.sum() == small_panel.nSNPs * sum(n != "empty" for n in names)
        return got

    check(["empty", "s0", "s1", "s2"], n_io_threads=1)             # dropped first; one loader
    check(["s0", "s1", "s2", "empty"], n_io_threads=8)             # dropped last; more threads than files
    check(["empty", "empty", "s1", "empty", "s0", "empty"], n_io_threads=2)
    none = check(["empty", "empty"], n_io_threads=2)               # nothing to impute: a successful call
    assert none["stats"]["gibbs_chain_calls"] == 0
    # a caller that wants columns, labels and counts only: the same text and counts, no per-SNP arrays (their pages went back)
    lean = run(["s0", "empty", "s1", "s2"], [0, 1, 2, 3], n_io_threads=3, discard_sample_arrays=True)
    for i, n in ((0, "s0"), (2, "s1"), (3, "s2")):
        assert lean["columns"][i].tolist() == col[n]
        r = lean["results"][i]
        assert r.dosage is None and r.gp_t is None and r.phasing_haps is None
        assert np.array_equal(r.read_labels, base["results"][i].read_labels) and r.nDosage == base["results"][i].nDosage
    for name in ("infoCount", "afCount", "hweCount", "alleleCount"):
        assert np.array_equal(getattr(lean["counts"], name), getattr(base["counts"], name)), name
    nothing = run([], [])
    assert nothing["imputed"] == [] and nothing["counts"].afCount.sum() == 0
    twice = check(["s1", "s1"], n_io_threads=2)                    # the same file (and global index) twice: the same column twice
    assert twice["columns"][0].tolist() == twice["columns"][1].tolist()
    for n_io in (1, 4):
        with pytest.raises(RuntimeError, match="missing.bam"):
            run(["s0", "missing", "s1", "s2"], [0, 1, 2, 3], n_io_threads=n_io)


@pytest.mark.parametrize("mspbwt,method", [(False, "diploid"), (True, "diploid"), (False, "nipt")])
def test_bam_range_call_with_rare_and_common_snps(tmp_path, mspbwt, method):
    """impute_rare_common = TRUE through the range call (QUILT2's default mode with use_mspbwt): every file is piled up TWICE -- at
    the common SNPs and at all SNPs (functions.R:132-172) -- the all-SNP reads ride in the same per-sample view
    (qa_sample_view_t), columns and count arrays cover ALL SNPs, the allele counts come from the all-SNP pile-up.  Against the same
    steps taken one by one in Python: loader twice per file, quilt_amd/driver.py on the oracle, the column writer, SummaryCounts."""
    from quilt_amd.driver import Driver, DriverParams
    from quilt_amd.io import SummaryCounts, loadBamAndConvert, make_per_sample_vcf_col, make_per_sample_vcf_col_nipt, per_sample_counts
    from quilt_amd.synth import make_rare_common, make_synthetic_panel, make_synthetic_sample_rare_common
    from tests import bamutil
    from tests.native_driver_backend import impute_bam_range_on_oracle
    panel = make_synthetic_panel(K=300, nSNPs=640, seed=5)
    rc = make_rare_common(panel, 3)
    Ta = rc.nSNPs_all
    rng = np.random.default_rng(8)
    alleles = [tuple(rng.choice(list("ACGT"), size=2, replace=False)) for _ in range(Ta)]
    ref_all, alt_all = [a for a, _ in alleles], [b for _, b in alleles]
    common = np.flatnonzero(rc.snp_is_common == 1)
    ref, alt = [ref_all[i] for i in common], [alt_all[i] for i in common]
    grid_all = (np.arange(Ta) // 32).astype(np.int32)
    header = [("chr20", int(rc.L_all[-1]) + 1000)]
    files = []
    for i in range(3):
        s_all = make_synthetic_sample_rare_common(panel, rc, 60 + i, n_reads=160)[0].all_snp
        f = str(tmp_path / f"r{i}.bam")
        bamutil.write_bam(f, header, bamutil.sample_to_alignments(s_all, rc.L_all, ref_all, alt_all, rng))
        files.append(f)
    empty = str(tmp_path / "none.bam")
    bamutil.write_bam(empty, header, [])
    files.insert(2, empty)
    P = DriverParams(nGibbsSamples=2, Ksubset=64, Knew=64, seed=9, impute_rare_common=True, small_ref_panel_gibbs_iterations=4,
                     small_ref_panel_block_gibbs_iterations=(2,), use_mspbwt=mspbwt, mspbwt_nindices=2, method=method)
    ffs = [0.12, 0.2, 0.3, 0.17] if method == "nipt" else None
    opts = dict(downsampleToCov=0, bqFilter=1)
    # step by step in Python
    grid = panel.grid if panel.grid is not None else np.arange(panel.nSNPs, dtype=np.int32) // 32
    samples, kept = [], []
    for i, f in enumerate(files):
        s = loadBamAndConvert(f, "chr20", panel.L, ref, alt, grid, **opts)
        if s.nReads < 2:
            continue
        s.all_snp = loadBamAndConvert(f, "chr20", rc.L_all, ref_all, alt_all, grid_all, **opts)
        if ffs:
            s.ff = ffs[i]
        samples.append(s)
        kept.append(i)
    assert kept == [0, 1, 3]
    drv = Driver(panel, OracleBackend(panel, rc), P, rare_common=rc)
    want = [drv.run([s], sample_offset=10 + i)[0] for i, s in zip(kept, samples)]
    counts = SummaryCounts(Ta)
    cols = []
    for s, r in zip(samples, want):
        counts.add_sample(*per_sample_counts(r.gp_t, s.all_snp, Ta))
        cols.append(make_per_sample_vcf_col_nipt(r.gp_t, r.fet_gp_t, r.phasing_haps, r.dosage, r.fet_dosage) if ffs
                    else make_per_sample_vcf_col(r.gp_t, r.phasing_haps, True))
    # the one native call
    got = impute_bam_range_on_oracle(panel, files, "chr20", ref, alt, P, n_threads=2, rare_common=rc, sample_index=[10, 11, 12, 13],
                                     all_sites=(rc.L_all, ref_all, alt_all, grid_all), n_io_threads=3, samples_per_launch_set=2, ff=ffs, **opts)
    assert got["imputed"] == [True, True, False, True]
    for j, i in enumerate(kept):
        g = got["results"][i]
        assert g.dosage.shape == (Ta,) and np.array_equal(g.dosage, want[j].dosage) and np.array_equal(g.gp_t, want[j].gp_t)
        assert np.array_equal(g.read_labels, want[j].read_labels) and np.array_equal(g.phasing_haps, want[j].phasing_haps)
        assert got["columns"][i].tolist() == cols[j].tolist()
    for name in ("infoCount", "afCount", "hweCount", "alleleCount"):
        assert np.array_equal(getattr(got["counts"], name), getattr(counts, name)), name

"""GPU end-to-end: the whole per-sample driver (Gibbs + full-panel passes + selection + consensus
phasing) on the HIP backend vs the same driver on the fp64 CPU oracle, same seeds.

Bar (BASELINE.json): dosage r2 >= 0.999 against the CPU path; genotype probabilities sum to 1.
The chains themselves must coincide: the Gibbs sampler is fp64 with the same uniforms, and the best-haplotype lists that
choose every next small panel come from fp64-state ranking passes, so the consensus read labels are identical and the
dosages differ only by the fp32 rounding of the dosage passes (|diff| <= 1e-4 asserted, ~2e-6 observed).
"""
import numpy as np
import pytest

from tests.util import r2

pytestmark = pytest.mark.gpu


def _run_both(panel, samples, prm):
    from quilt_amd.driver import Driver, HipBackend
    from quilt_amd.native import DevicePanel
    from tests.oracle_backend import OracleBackend
    dev = DevicePanel(panel)
    got = Driver(panel, HipBackend(dev), prm).run(samples)
    ref = Driver(panel, OracleBackend(panel), prm).run(samples)
    dev.close()
    return got, ref


# The medium runs: EVERY driver seed of a range, none chosen.  Round 4 ran one seed here and had to move it off a last-bit tie
# (seed 5: one chain's first round meets several haplotypes whose gamma equals the K_top-th largest up to the last bits; device
# list 7 entries, oracle list 5, both valid by reference-single.cpp:129-194).  Now: (i) in VALIDATION mode (K-wide sums in the
# reference's order, csrc/fullpass_ref.hip) every seed -- seed 5 included -- ends bit-identical to the CPU pipeline; (ii) in
# PRODUCTION mode a run either coincides with the CPU pipeline (labels identical, dosage to the fp32 dosage passes' rounding) or,
# where it met such a tie, lies within the CPU pipeline's own seed-to-seed spread (tests/test_sum_order_gpu.py for the sweeps).
MEDIUM_SEEDS = list(range(5, 13))
_CACHE = {}


def _medium_run(panel, seed):
    import dataclasses
    from quilt_amd.driver import Driver, DriverParams, HipBackend
    from quilt_amd.native import DevicePanel
    from quilt_amd.synth import make_synthetic_sample
    from tests.oracle_backend import OracleBackend
    if seed not in _CACHE:
        samples = [make_synthetic_sample(panel, seed=1000 + i, n_reads=1000) for i in range(3)]
        prm = DriverParams(nGibbsSamples=3, Ksubset=200, Knew=200, seed=seed)
        got, ref = _run_both(panel, samples, prm)
        dev = DevicePanel(panel)
        dev.set_sum_order(True)
        val = Driver(panel, HipBackend(dev), prm).run(samples)
        dev.close()
        ref2 = Driver(panel, OracleBackend(panel), dataclasses.replace(prm, seed=seed + 1000)).run(samples)
        _CACHE[seed] = dict(samples=samples, got=got, ref=ref, val=val, ref2=ref2)
    return _CACHE[seed]


def test_pipeline_r2_bar_vs_cpu_path(medium_panel):
    """BASELINE.json: dosage r2 vs the CPU path >= 0.999 -- for every sample of every seed whose chains coincide with the CPU
    path's, and exactly 1 in validation mode; a production-mode run that met a last-bit tie (another realisation of the same
    sampler) is held to the CPU path's own r2 between two driver seeds on the same reads instead."""
    n = parted = 0
    for seed in MEDIUM_SEEDS:
        run = _medium_run(medium_panel, seed)
        for i, (g, r, v, r2nd) in enumerate(zip(run["got"], run["ref"], run["val"], run["ref2"])):
            n += 1
            assert np.array_equal(v.read_labels, r.read_labels) and np.array_equal(v.dosage, r.dosage), (seed, i)
            if np.array_equal(g.read_labels, r.read_labels) and np.abs(g.dosage - r.dosage).max() <= 1e-4:
                assert r2(g.dosage, r.dosage) >= 0.999, (seed, i, r2(g.dosage, r.dosage))
            else:
                parted += 1
                print(f"seed {seed} sample {i}: parted, r2(GPU, CPU) = {r2(g.dosage, r.dosage):.5f}, CPU seed-to-seed r2 = {r2(r2nd.dosage, r.dosage):.5f}")
                assert r2(g.dosage, r.dosage) >= r2(r2nd.dosage, r.dosage), (seed, i)
    print(f"{parted} of {n} production-mode sample runs parted from the CPU path")
    assert parted <= n // 4


@pytest.mark.parametrize("seed", MEDIUM_SEEDS)
def test_pipeline_matches_oracle(medium_panel, seed):
    run = _medium_run(medium_panel, seed)
    samples = run["samples"]
    for i, (g, r, v) in enumerate(zip(run["got"], run["ref"], run["val"])):
        # validation mode: the CPU pipeline's results bit for bit
        assert v.nDosage == r.nDosage == 3
        assert np.array_equal(v.read_labels, r.read_labels)
        assert np.array_equal(v.dosage, r.dosage) and np.array_equal(v.gp_t, r.gp_t) and np.array_equal(v.phasing_haps, r.phasing_haps)
        # production mode
        assert g.nDosage == 3
        np.testing.assert_allclose(g.gp_t.sum(axis=0), 1.0, atol=2e-3)   # check_quilt_output (test-drivers.R:38-61)
        truth = samples[i].truth_haps.sum(axis=0)
        assert r2(g.dosage, truth) >= 0.9 and abs(r2(g.dosage, truth) - r2(r.dosage, truth)) < 0.02
        same = np.array_equal(g.read_labels, r.read_labels)
        if same and np.abs(g.dosage - r.dosage).max() <= 1e-4:   # (fp32 dosage passes: ~2e-6 observed)
            # recast_haps (functions.R:1207-1217) takes argmax decisions on the genotype posteriors: an fp32-rounding-sized
            # difference can flip one at a near-tie, so a handful of sites may differ
            assert np.mean(np.abs(g.phasing_haps - r.phasing_haps) > 1e-4) <= 2e-3
        print(f"seed {seed} sample {i}: r2(gpu, oracle) = {r2(g.dosage, r.dosage):.6f}, max|d| = {np.abs(g.dosage - r.dosage).max():.2e}, "
              f"consensus labels identical: {same}")


def test_pipeline_default_parameters_small(small_panel):
    """K < Ksubset path (quilt.R:453-463): n_seek_its = 1, Ksubset = K."""
    from quilt_amd.driver import DriverParams
    from quilt_amd.synth import make_synthetic_sample
    panel = small_panel
    samples = [make_synthetic_sample(panel, seed=7, n_reads=150)]
    prm = DriverParams(nGibbsSamples=2, Ksubset=2000, Knew=2000, seed=3)
    got, ref = _run_both(panel, samples, prm)
    assert r2(got[0].dosage, ref[0].dosage) >= 0.999


def test_pipelined_batches_gpu(medium_panel):
    """Driver.run_stream on the HIP backend: launches that mix first-round main chains with phasing chains of the
    previous batch (first_read < 0 marks the latter) give each sample the result of a run of its own."""
    from quilt_amd.driver import Driver, DriverParams, HipBackend
    from quilt_amd.native import DevicePanel
    from quilt_amd.synth import make_synthetic_sample
    panel = medium_panel
    samples = [make_synthetic_sample(panel, seed=2000 + i, n_reads=600) for i in range(4)]
    prm = DriverParams(nGibbsSamples=2, Ksubset=200, Knew=200, seed=11)
    dev = DevicePanel(panel)
    batches = [(samples[0:2], 0), (samples[2:4], 2)]
    streamed = list(Driver(panel, HipBackend(dev), prm).run_stream(batches))
    for (smp, off), got in zip(batches, streamed):
        ref = Driver(panel, HipBackend(dev), prm).run(smp, sample_offset=off)
        for g, r in zip(got, ref):
            assert np.array_equal(g.read_labels, r.read_labels)
            assert np.abs(g.dosage - r.dosage).max() <= 1e-6
    dev.close()


def test_two_host_threads_share_the_device(medium_panel):
    """DeviceWorkers: two threads, each with its own panel handle / stream / arena, split every batch; per-sample
    results equal the single-threaded driver's."""
    from quilt_amd.driver import Driver, DriverParams, HipBackend
    from quilt_amd.native import DevicePanel
    from quilt_amd.synth import make_synthetic_sample
    from quilt_amd.workers import DeviceWorkers
    panel = medium_panel
    samples = [make_synthetic_sample(panel, seed=3000 + i, n_reads=500) for i in range(5)]
    prm = DriverParams(nGibbsSamples=2, Ksubset=200, Knew=200, seed=13)
    batches = [(samples[0:3], 0), (samples[3:5], 3)]
    wk = DeviceWorkers(panel, prm, n_workers=2)
    got = list(wk.run_stream(batches))
    wk.close()
    dev = DevicePanel(panel)
    for (smp, off), g_batch in zip(batches, got):
        ref = Driver(panel, HipBackend(dev), prm).run(smp, sample_offset=off)
        assert len(g_batch) == len(ref)
        for g, r in zip(g_batch, ref):
            assert np.array_equal(g.read_labels, r.read_labels)
            assert np.abs(g.dosage - r.dosage).max() <= 1e-6
    dev.close()


@pytest.mark.parametrize("mspbwt", [False, True])
def test_host_threads_take_the_device_in_phases(medium_panel, mspbwt):
    """DeviceWorkers(exclusive=True), the bench's configuration: three threads, whole batches in turn, launch sets through the
    device gate (full-panel sets exclusive, Gibbs launches side by side, the msPBWT search as an express hold), scratch from
    the one device-wide arena.  Results equal the single-threaded, ungated driver's; the gate saw holds of both kinds."""
    from quilt_amd import native
    from quilt_amd.driver import Driver, DriverParams, HipBackend
    from quilt_amd.native import DevicePanel
    from quilt_amd.synth import make_synthetic_sample
    from quilt_amd.workers import DeviceWorkers
    panel = medium_panel
    samples = [make_synthetic_sample(panel, seed=3100 + i, n_reads=500) for i in range(8)]
    prm = DriverParams(nGibbsSamples=2, Ksubset=200, Knew=200, seed=14, use_mspbwt=mspbwt, mspbwt_nindices=2)
    batches = [(samples[0:3], 0), (samples[3:5], 3), (samples[5:6], 5), (samples[6:8], 6)]
    native.gate_stats(0, reset=True)
    wk = DeviceWorkers(panel, prm, n_workers=3, exclusive=True, split="alternate")
    got = list(wk.run_stream(batches))
    wk.close()
    st = native.gate_stats(0)
    # (msPBWT mode with the default neighbour scan has no device work besides the Gibbs launches: the query is host code)
    assert st["gibbs_holds"] > 0 and st["held_ms"] > 0 and (st["holds"] == st["gibbs_holds"] if mspbwt else st["holds"] > st["gibbs_holds"])
    dev = DevicePanel(panel)
    for (smp, off), g_batch in zip(batches, got):
        ref = Driver(panel, HipBackend(dev), prm).run(smp, sample_offset=off)
        assert len(g_batch) == len(ref)
        for g, r in zip(g_batch, ref):
            assert np.array_equal(g.read_labels, r.read_labels)
            assert np.abs(g.dosage - r.dosage).max() <= 1e-6
    dev.close()


def test_pipeline_ont_reads(medium_panel):
    """BASELINE configs[3] in small: long noisy reads (hundreds of SNPs each, Jmax path, reads spanning many grids)
    through the whole driver, GPU vs the same driver on the oracle."""
    from quilt_amd.driver import DriverParams
    from quilt_amd.synth import make_synthetic_sample
    panel = medium_panel
    samples = [make_synthetic_sample(panel, seed=4000 + i, mode="ont", n_reads=40) for i in range(2)]
    prm = DriverParams(nGibbsSamples=2, Ksubset=200, Knew=200, seed=17)
    got, ref = _run_both(panel, samples, prm)
    for g, r in zip(got, ref):
        assert np.array_equal(g.read_labels, r.read_labels)
        assert np.abs(g.dosage - r.dosage).max() <= 1e-4
        assert r2(g.dosage, r.dosage) >= 0.999


@pytest.fixture(scope="module")
def full_size():
    """BASELINE.json's headline panel (K = 50 000 haplotypes, 64 000 SNPs / 2 000 grids), built on the device as a production
    run would build it."""
    from quilt_amd.native import DevicePanel
    from quilt_amd.synth import make_synthetic_panel
    panel = make_synthetic_panel(K=50000, nSNPs=64000, seed=4916)
    dev = DevicePanel.from_rhb(panel)
    yield panel, dev
    dev.close()


def _check_quilt_output(r, n_dosage):
    """check_quilt_output (test-drivers.R:1-89) on one result: posteriors sum to 1 +- 0.002, dosages within range and
    consistent with them, phased haplotypes are probabilities."""
    assert r.nDosage == n_dosage
    np.testing.assert_allclose(r.gp_t.sum(axis=0), 1.0, atol=2e-3)
    assert r.dosage.min() >= -1e-9 and r.dosage.max() <= 2 + 1e-9
    np.testing.assert_allclose(r.dosage, r.gp_t[1] + 2 * r.gp_t[2], atol=1e-9)
    assert r.phasing_haps.min() >= 0 and r.phasing_haps.max() <= 1


def test_full_size_invariants(full_size):
    """BASELINE.json's headline sizes (K = 50 000 haplotypes, 64 000 SNPs / 2 000 grids, 20 000 reads): the CPU path
    takes ~20 minutes per sample there, so the whole driver is checked through size-independent properties -- the
    acceptance criteria of the reference's own end-to-end tests (check_quilt_output, test-drivers.R:1-89): genotype
    probabilities sum to 1 +- 0.002, dosages within [0, 2] and consistent with them, imputed dosage close to the
    simulated truth; plus: every Gibbs label is 1 or 2, phased haplotypes in [0, 1], results independent of batching."""
    from quilt_amd.driver import Driver, DriverParams, HipBackend
    from quilt_amd.synth import make_synthetic_sample
    panel, dev = full_size
    samples = [make_synthetic_sample(panel, seed=1000 + i, n_reads=20000) for i in range(2)]
    prm = DriverParams(nGibbsSamples=7, n_seek_its=3, Ksubset=600, Knew=600, seed=1)
    res = Driver(panel, HipBackend(dev), prm).run(samples)
    for s, r in zip(samples, res):
        _check_quilt_output(r, 7)
        assert set(np.unique(r.read_labels)) <= {1, 2}
        truth = s.truth_haps.sum(axis=0)
        assert r2(r.dosage, truth) >= 0.99
        assert np.mean(np.abs(r.dosage - truth) > 0.1) < 0.02    # "DS within 0.1 of truth" for nearly every site
    one = Driver(panel, HipBackend(dev), prm).run(samples[1:], sample_offset=1)[0]
    assert np.array_equal(one.read_labels, res[1].read_labels) and np.abs(one.dosage - res[1].dosage).max() <= 1e-6


def test_full_size_invariants_ont(full_size):
    """BASELINE configs[3] at the headline sizes: 300 long noisy reads per sample (hundreds of SNPs each, the Jmax path, reads
    spanning many grids); same properties as above."""
    from quilt_amd.driver import Driver, DriverParams, HipBackend
    from quilt_amd.synth import make_synthetic_sample
    panel, dev = full_size
    samples = [make_synthetic_sample(panel, seed=2000 + i, mode="ont", n_reads=300) for i in range(2)]
    prm = DriverParams(nGibbsSamples=7, n_seek_its=3, Ksubset=600, Knew=600, seed=2)
    res = Driver(panel, HipBackend(dev), prm).run(samples)
    for s, r in zip(samples, res):
        _check_quilt_output(r, 7)
        assert set(np.unique(r.read_labels)) <= {1, 2}
        truth = s.truth_haps.sum(axis=0)
        print("ont: r2 vs truth", r2(r.dosage, truth), "off by > 0.1:", np.mean(np.abs(r.dosage - truth) > 0.1))
        assert r2(r.dosage, truth) >= 0.99
    one = Driver(panel, HipBackend(dev), prm).run(samples[1:], sample_offset=1)[0]
    assert np.array_equal(one.read_labels, res[1].read_labels) and np.abs(one.dosage - res[1].dosage).max() <= 1e-6


def test_full_size_invariants_nipt(full_size):
    """BASELINE configs[4] at the headline sizes: method = "nipt", mother + fetus from one read mixture (ff = 0.2), three read
    labels, block Gibbs.  check_quilt_output's properties for the maternal AND the fetal output; all three labels in use; the
    maternal dosage close to the simulated truth, the fetal one (a fifth of the reads carry it) correlated with it; results
    independent of batching."""
    from quilt_amd.driver import Driver, DriverParams, HipBackend
    from quilt_amd.synth import make_synthetic_sample
    panel, dev = full_size
    samples = [make_synthetic_sample(panel, seed=3000 + i, n_reads=20000, ff=0.2) for i in range(2)]
    prm = DriverParams(nGibbsSamples=7, n_seek_its=3, Ksubset=600, Knew=600, seed=3, method="nipt")
    res = Driver(panel, HipBackend(dev), prm).run(samples)
    for s, r in zip(samples, res):
        _check_quilt_output(r, 7)
        np.testing.assert_allclose(r.fet_gp_t.sum(axis=0), 1.0, atol=2e-3)
        assert r.fet_dosage.min() >= -1e-9 and r.fet_dosage.max() <= 2 + 1e-9
        np.testing.assert_allclose(r.fet_dosage, r.fet_gp_t[1] + 2 * r.fet_gp_t[2], atol=1e-9)
        assert set(np.unique(r.read_labels)) == {1, 2, 3} and r.phasing_haps.shape[1] == 3
        mat = s.truth_haps[0] + s.truth_haps[1]      # maternal transmitted + untransmitted
        fet = s.truth_haps[0] + s.truth_haps[2]      # maternal transmitted + paternal transmitted (functions.R:1009-1016)
        print("nipt: r2 vs truth, mother", r2(r.dosage, mat), "fetus", r2(r.fet_dosage, fet))
        assert r2(r.dosage, mat) >= 0.95        # (0.981 / 0.978 observed)
        assert r2(r.fet_dosage, fet) >= 0.8     # (0.888 / 0.893 observed)
    one = Driver(panel, HipBackend(dev), prm).run(samples[1:], sample_offset=1)[0]
    assert np.array_equal(one.read_labels, res[1].read_labels) and np.abs(one.dosage - res[1].dosage).max() <= 1e-6
    assert np.abs(one.fet_dosage - res[1].fet_dosage).max() <= 1e-6


def test_pipeline_nipt(medium_panel):
    """BASELINE configs[4] in small: method = "nipt" end to end (three-label sampler with its block Gibbs, three
    full-panel passes per chain, NIPT consensus and recast) on the HIP backend vs the oracle backend."""
    from quilt_amd.driver import Driver, DriverParams, HipBackend
    from quilt_amd.native import DevicePanel
    from quilt_amd.synth import make_synthetic_sample
    from tests.oracle_backend import OracleBackend
    panel = medium_panel
    samples = [make_synthetic_sample(panel, seed=3000 + i, n_reads=1000, ff=0.15 + 0.05 * i) for i in range(3)]   # one launch, three fetal fractions
    prm = DriverParams(nGibbsSamples=3, Ksubset=200, Knew=200, seed=5, method="nipt")
    dev = DevicePanel(panel)
    got = Driver(panel, HipBackend(dev), prm).run(samples)
    ref = Driver(panel, OracleBackend(panel), prm).run(samples)
    dev.close()
    n_same_phase = 0
    for i, (g, r) in enumerate(zip(got, ref)):
        assert g.nDosage == r.nDosage == 3
        assert np.array_equal(g.read_labels, r.read_labels) and (g.read_labels == 3).any()
        np.testing.assert_allclose(g.gp_t.sum(axis=0), 1.0, atol=2e-3)
        np.testing.assert_allclose(g.fet_gp_t.sum(axis=0), 1.0, atol=2e-3)
        assert np.abs(g.dosage - r.dosage).max() <= 1e-4 and np.abs(g.fet_dosage - r.fet_dosage).max() <= 1e-4
        assert r2(g.dosage, r.dosage) >= 0.999 and r2(g.fet_dosage, r.fet_dosage) >= 0.999
        # The phasing chain runs on the last chain's final small panel.  Two haplotypes whose gamma differ in the last bit can
        # come out tied on one side and ordered on the other (the K-wide normalising sums are formed in a different order),
        # which permutes that panel and, through the order of the Gibbs sums, may move a phasing label: the phased
        # haplotypes then agree as dosages, not entry by entry.  Entry-by-entry agreement is required of most samples.
        n_same_phase += np.mean(np.abs(g.phasing_haps - r.phasing_haps) > 1e-4) <= 2e-3
        assert r2(g.phasing_haps[:, :2].sum(axis=1), r.phasing_haps[:, :2].sum(axis=1)) >= 0.98
        mat = samples[i].truth_haps[0] + samples[i].truth_haps[1]
        print(f"sample {i}: r2(gpu, oracle) mother {r2(g.dosage, r.dosage):.6f} fetus {r2(g.fet_dosage, r.fet_dosage):.6f}; "
              f"mother vs truth {r2(g.dosage, mat):.3f}")
    assert n_same_phase >= 2


def test_cu_partition_does_not_change_results(medium_panel):
    """qa_panel_set_cu_partition only moves the Gibbs launches to a CU-masked stream."""
    from quilt_amd.driver import Driver, DriverParams, HipBackend
    from quilt_amd.native import DevicePanel
    from quilt_amd.synth import make_synthetic_sample
    panel = medium_panel
    samples = [make_synthetic_sample(panel, seed=1000 + i, n_reads=600) for i in range(2)]
    prm = DriverParams(nGibbsSamples=2, Ksubset=128, Knew=128, seed=5)
    outs = []
    for part in (None, (1, 2)):
        dev = DevicePanel(panel)
        if part:
            dev.set_cu_partition(*part)
        outs.append(Driver(panel, HipBackend(dev), prm).run(samples))
        dev.close()
    for a, b in zip(*outs):
        assert np.array_equal(a.read_labels, b.read_labels) and np.array_equal(a.dosage, b.dosage)


def test_truncated_lists_refetched_on_the_device():
    """Duplicated panel haplotypes: exact gamma ties overflow the fused picker's candidate list (the grid is handed to k_topk)
    and the lists the driver receives are truncated; the selection that runs out of ranked candidates re-fetches the full
    lists (qa_fullpass_batch) -- result identical to the CPU path's, which always holds full lists."""
    from quilt_amd.driver import Driver, DriverParams, HipBackend
    from quilt_amd.native import DevicePanel
    from quilt_amd.synth import make_synthetic_panel, make_synthetic_sample
    from tests.oracle_backend import OracleBackend
    from tests.util import panel_from_rhb
    base = make_synthetic_panel(K=40, nSNPs=320, seed=8)
    rhb = np.asfortranarray(np.tile(base.rhb_t, (12, 1)))
    panel = panel_from_rhb(rhb, base.transMatRate_t, 320, 255, base.ref_error)
    panel.L_grid = base.L_grid
    samples = [make_synthetic_sample(panel, seed=70, n_reads=80)]
    prm = DriverParams(nGibbsSamples=1, Ksubset=64, Knew=64, seed=4)
    dev = DevicePanel(panel)
    drv = Driver(panel, HipBackend(dev), prm)
    got = drv.run(samples)
    ref = Driver(panel, OracleBackend(panel), prm).run(samples)
    assert drv.n_full_list_refetches > 0
    assert np.array_equal(got[0].read_labels, ref[0].read_labels)
    assert np.abs(got[0].dosage - ref[0].dosage).max() <= 1e-4
    dev.close()


def test_underflow_retry_on_the_device(small_panel):
    """Contradictory reads piled on one grid underflow the small-panel forward at maxDifferenceBetweenReads = 1e10; the driver
    repeats the call with a tenth of it until the call succeeds (functions.R:2704-2715).
    One call, far from the edge: at 1e10 both paths report the underflow; at 3 neither does and the labels are identical.
    Through the driver: both paths retry and finish.  (How many retries a chain needs is decided where a product of ~100
    emissions crosses 1e-308 -- in the denormal range x * (1 / e), the device's form, and the reference's x / e round
    differently, so at the crossing the two may report a step apart; the retry loop makes either outcome valid.)"""
    from quilt_amd.driver import Driver, DriverParams, HipBackend
    from quilt_amd.gibbs_nipt import rcpp_forwardBackwardGibbsNIPT
    from quilt_amd.native import DevicePanel
    from oracle import oracle as O
    from tests.oracle_backend import OracleBackend
    from tests.util import underflowing_sample
    panel = small_panel
    s = underflowing_sample(panel)
    dev = DevicePanel(panel)
    rng = np.random.default_rng(4)
    which = np.sort(rng.choice(panel.K, 64, replace=False)).astype(np.int32) + 1
    R = s.nReads
    H0 = rng.integers(1, 3, size=R).astype(np.int32)
    ru, rs = rng.random(R * 21), rng.random(3 * (panel.nGrids - 1))
    for md, under in ((1e10, True), (3.0, False)):
        ref = O.forwardBackwardGibbsNIPT(panel, s, which, H0, ru, 0, rs, maxDifferenceBetweenReads=md)
        got = rcpp_forwardBackwardGibbsNIPT(dev, s, which, H0, ru, 0, rs, maxDifferenceBetweenReads=md)
        assert bool(ref["underflow_problem"]) == under and bool(got["underflow_problem"]) == under
        if not under:
            assert np.array_equal(got["H"], ref["H"])
    prm = DriverParams(nGibbsSamples=1, Ksubset=64, Knew=64, seed=6)
    d_gpu = Driver(panel, HipBackend(dev), prm)
    got = d_gpu.run([s])
    d_cpu = Driver(panel, OracleBackend(panel), prm)
    ref = d_cpu.run([s])
    dev.close()
    assert d_cpu.n_underflow_retries > 0 and d_gpu.n_underflow_retries > 0
    assert abs(d_gpu.n_underflow_retries - d_cpu.n_underflow_retries) <= 0.5 * d_cpu.n_underflow_retries
    assert np.isfinite(got[0].dosage).all() and np.isfinite(ref[0].dosage).all()


@pytest.mark.parametrize("method", ["diploid", "nipt"])
def test_bam_to_vcf_end_to_end_on_the_device(tmp_path, small_panel, method):
    """The formats either side of the path with the device in the middle: BAM files -> loader ->
    driver on the HIP backend -> VCF.  In validation mode (the K-wide sums in the reference's order) the file is the CPU path's,
    row for row; in production mode this small panel (K = 1 000, Ksubset = 64), full of duplicated haplotypes, lets a last-bit
    tie send a chain down another (equally valid) Gibbs path, so the file agrees closely, not entry by entry."""
    from quilt_amd.driver import HipBackend
    from quilt_amd.native import DevicePanel
    from tests.oracle_backend import OracleBackend
    from tests.test_driver_host import _bam_to_vcf
    ff = 0.2 if method == "nipt" else None
    dev = DevicePanel(small_panel)
    for d in ("gpu", "val", "cpu"):
        (tmp_path / d).mkdir()
    rows_g, rec_g, truth = _bam_to_vcf(tmp_path / "gpu", small_panel, HipBackend(dev), method=method, ff=ff)
    dev.set_sum_order(True)
    rows_v, rec_v, _ = _bam_to_vcf(tmp_path / "val", small_panel, HipBackend(dev), method=method, ff=ff)
    rows_c, rec_c, _ = _bam_to_vcf(tmp_path / "cpu", small_panel, OracleBackend(small_panel), method=method, ff=ff)
    dev.close()
    assert rows_v == rows_c   # validation mode: the same text
    for k in rec_c["results"]:
        assert np.array_equal(rec_v["results"][k].read_labels, rec_c["results"][k].read_labels)
    assert len(rows_g) == len(rows_c)
    # production mode: same sites, same columns, well-formed entries, dosages that agree as a whole
    n_post = 2 if method == "diploid" else 4      # GP, DS | MGP, MDS, FGP, FDS
    ds_g, ds_c = [], []
    for a, b in zip(rows_g, rows_c):
        assert a[:7] == b[:7] and a[8] == b[8] and len(a) == len(b)
        for x, y in zip(a[9:], b[9:]):
            px, py = x.split(":"), y.split(":")
            if px[1].startswith("."):   # the sample without reads: the reference's fixed string, whatever the method
                assert x == y == "./.:.,.,.:.:.,."
                continue
            assert len(px) == len(py) == (4 if method == "diploid" else 5)
            gp = [float(v) for v in px[1].split(",")]
            assert abs(sum(gp) - 1) <= 2.1e-3 and abs(float(px[2]) - (gp[1] + 2 * gp[2])) <= 2.1e-3
            ds_g.append(float(px[2]))
            ds_c.append(float(py[2]))
    from tests.util import r2
    assert r2(np.array(ds_g), np.array(ds_c)) >= 0.98
    assert np.mean(np.abs(np.array(ds_g) - np.array(ds_c)) <= 1.001e-3) >= 0.8


@pytest.fixture(scope="module")
def quick_start_panel():
    from quilt_amd.native import DevicePanel
    from quilt_amd.synth import make_1000g_like_panel
    panel = make_1000g_like_panel(K=5008, nSNPs=3200, seed=2504)
    dev = DevicePanel.from_rhb(panel)
    yield panel, dev
    dev.close()


@pytest.mark.parametrize("seed", list(range(1, 9)))
def test_quick_start_shaped_run_bam_to_vcf(tmp_path, quick_start_panel, seed):
    """BASELINE configs[0]'s SHAPE (the quick-start: one 1x sample against the 1000 Genomes panel, ~5 000 haplotypes) with
    stand-ins for the data that cannot be had here: a K = 5 008 panel with a rare-variant-dominated (1 / i) frequency spectrum
    (quilt_amd.synth.make_1000g_like_panel) compressed ON THE DEVICE from its packed form (qa_panel_create_from_rhb: the step
    quilt-prepare-reference.R:416-428 does with STITCH), one synthetic 1x sample through a BAM file, QUILT's defaults
    (nGibbsSamples = 7, n_seek_its = 3, Ksubset = 600), fp64 dosage passes, BAM -> loader -> driver -> VCF.
    EVERY driver seed 1..8 (27 % of this panel's haplotypes repeat another one over the whole region: every best-haplotype list
    is a long exact tie, the panel on which the device and the CPU path part most often).  Validation mode (the K-wide sums in the
    reference's order): the file equals the one the CPU path writes from the same BAM -- same text, on every seed.  Production
    mode: either that, or (a last-bit tie on the way) dosages within the CPU path's own spread between two driver seeds."""
    import dataclasses
    from quilt_amd.driver import DriverParams, HipBackend
    from tests.oracle_backend import OracleBackend
    from tests.test_driver_host import _bam_to_vcf
    from tests.util import r2
    panel, dev = quick_start_panel
    prm = DriverParams(seed=seed)
    for d in ("gpu", "val", "cpu", "cpu2"):
        (tmp_path / d).mkdir()
    dev.set_dosage_precision(64)
    dev.set_sum_order(False)
    rows_g, rec_g, truth = _bam_to_vcf(tmp_path / "gpu", panel, HipBackend(dev), n_samples=1, n_reads=1000, prm=prm)
    dev.set_sum_order(True)
    rows_v, rec_v, _ = _bam_to_vcf(tmp_path / "val", panel, HipBackend(dev), n_samples=1, n_reads=1000, prm=prm)
    dev.set_sum_order(False)
    cpu = OracleBackend(panel, n_threads=8)
    rows_c, rec_c, _ = _bam_to_vcf(tmp_path / "cpu", panel, cpu, n_samples=1, n_reads=1000, prm=prm)
    _, rec_c2, _ = _bam_to_vcf(tmp_path / "cpu2", panel, cpu, n_samples=1, n_reads=1000, prm=dataclasses.replace(prm, seed=seed + 1000))
    g, v, c, c2 = (rec["results"][0] for rec in (rec_g, rec_v, rec_c, rec_c2))
    # validation mode: bit for bit, and the same file
    assert np.array_equal(v.read_labels, c.read_labels) and np.array_equal(v.dosage, c.dosage)
    assert rows_v == rows_c
    # production mode
    ds = np.array([float(r[9].split(":")[2]) for r in rows_g])
    assert r2(ds, truth[0]) > 0.9
    if np.array_equal(g.read_labels, c.read_labels) and np.abs(g.dosage - c.dosage).max() <= 1e-9:
        diff = [(a, b) for a, b in zip(rows_g, rows_c) if a != b]
        # (three-decimal strings of numbers that agree to 1e-9: a value within 1e-9 of a rounding boundary may print differently)
        assert len(diff) <= 2, diff[:3]
    else:
        print(f"seed {seed}: parted from the CPU path, r2(GPU, CPU) = {r2(g.dosage, c.dosage):.6f}, CPU seed-to-seed r2 = {r2(c2.dosage, c.dosage):.6f}")
        assert r2(g.dosage, c.dosage) >= r2(c2.dosage, c.dosage)
        assert abs(r2(g.dosage, truth[0]) - r2(c.dosage, truth[0])) <= 0.01

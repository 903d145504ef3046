"""qa_impute_samples on the device: the native loop over the library's own entry points equals quilt_amd/driver.py over the
same entry points bit for bit (every chain owns its random stream; the kernels are deterministic), for one and for three
host threads, M1 and msPBWT mode, and agrees with the whole pipeline on the CPU oracle as the Python driver does."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def medium_panel():
    from quilt_amd.synth import make_synthetic_panel
    return make_synthetic_panel(K=5000, nSNPs=3200, seed=11)


def _same(a, b):
    assert a.nDosage == b.nDosage
    assert np.array_equal(a.read_labels, b.read_labels)
    assert np.array_equal(a.dosage, b.dosage)
    assert np.array_equal(a.gp_t, b.gp_t)
    assert np.array_equal(a.phasing_haps, b.phasing_haps)


@pytest.mark.parametrize("mspbwt", [False, True])
@pytest.mark.parametrize("n_threads", [1, 3])
def test_native_driver_equals_python_driver_on_the_device(medium_panel, mspbwt, n_threads):
    from quilt_amd.driver import Driver, DriverParams, HipBackend
    from quilt_amd.impute import impute_samples
    from quilt_amd.native import DevicePanel
    from quilt_amd.synth import make_synthetic_sample
    panel = medium_panel
    samples = [make_synthetic_sample(panel, seed=4100 + i, n_reads=500) for i in range(7)]
    prm = DriverParams(nGibbsSamples=3, Ksubset=200, Knew=200, seed=21, use_mspbwt=mspbwt, mspbwt_nindices=2)
    dev = DevicePanel(panel)
    dev.set_dosage_precision(64)
    want = Driver(panel, HipBackend(dev), prm).run(samples, sample_offset=40)
    devs = [DevicePanel(panel) for _ in range(n_threads)]
    for d in devs:
        d.set_device_share(n_threads)
        d.set_dosage_precision(64)
        if n_threads > 1:
            d.set_exclusive(True)
    got, st = impute_samples(devs, samples, prm, sample_offset=40, samples_per_launch_set=2, return_stats=True)
    for d in devs:
        d.close()
    dev.close()
    for a, b in zip(got, want):
        _same(a, b)
    assert st["gibbs_chain_calls"] >= 7 * 4 * 3
    if not mspbwt:   # (10 thinned grids x 2 labels x 5 ranks < Knew: these selections go on to the complete lists)
        assert st["device_selections"] + st["full_list_refetches"] > 0


def test_native_driver_agrees_with_the_cpu_pipeline(medium_panel):
    """The metric's `dosage r2 vs CPU ref` for the native path: labels identical, dosages to 1e-9 (fp64 dosage passes)."""
    from quilt_amd.driver import Driver, DriverParams
    from quilt_amd.impute import impute_samples
    from quilt_amd.native import DevicePanel
    from quilt_amd.synth import make_synthetic_sample
    from tests.oracle_backend import OracleBackend
    panel = medium_panel
    samples = [make_synthetic_sample(panel, seed=4200 + i, n_reads=400) for i in range(2)]
    prm = DriverParams(nGibbsSamples=2, Ksubset=128, Knew=128, seed=5)
    ref = Driver(panel, OracleBackend(panel, n_threads=4), prm).run(samples)
    dev = DevicePanel(panel)
    dev.set_dosage_precision(64)
    got = impute_samples([dev], samples, prm)
    dev.close()
    for g, r in zip(got, ref):
        assert np.array_equal(g.read_labels, r.read_labels)
        assert np.abs(g.dosage - r.dosage).max() <= 1e-9
        assert np.corrcoef(g.dosage, r.dosage)[0, 1] ** 2 >= 0.999999


def test_native_driver_refuses_bad_input(medium_panel):
    from quilt_amd.driver import DriverParams
    from quilt_amd.impute import impute_samples
    from quilt_amd.native import DevicePanel, QuiltAmdError
    from quilt_amd.synth import make_synthetic_sample
    from tests.native_driver_backend import _Reads
    dev = DevicePanel(medium_panel)
    z = np.zeros(0, dtype=np.int32)
    with pytest.raises(QuiltAmdError, match="no reads"):
        impute_samples([dev], [make_synthetic_sample(medium_panel, seed=1, n_reads=50), _Reads(np.zeros(1, dtype=np.int32), z, z, z)],
                       DriverParams(nGibbsSamples=2, Ksubset=64, Knew=64))
    dev.close()


def test_native_driver_rare_common_on_the_device(medium_panel):
    """impute_rare_common through qa_impute_samples on the device == quilt_amd/driver.py on the device, bit for bit (all-SNP
    dosages, phased haplotypes, labels), with two host threads."""
    from quilt_amd.driver import Driver, DriverParams, HipBackend
    from quilt_amd.impute import impute_samples
    from quilt_amd.native import DevicePanel, DeviceRareCommon
    from quilt_amd.synth import make_rare_common, make_synthetic_sample_rare_common
    panel = medium_panel
    rc = make_rare_common(panel, 4)
    samples = [make_synthetic_sample_rare_common(panel, rc, 2600 + i, n_reads=500)[0] for i in range(5)]
    prm = DriverParams(nGibbsSamples=2, Ksubset=128, Knew=128, seed=8, impute_rare_common=True)
    dev = DevicePanel(panel)
    dev.set_dosage_precision(64)
    drc = DeviceRareCommon(dev, rc)
    want = Driver(panel, HipBackend(dev, drc), prm, rare_common=rc).run(samples, sample_offset=7)
    devs = [DevicePanel(panel) for _ in range(2)]
    for d in devs:
        d.set_device_share(2)
        d.set_dosage_precision(64)
        d.set_exclusive(True)
    drcs = [DeviceRareCommon(d, rc) for d in devs]
    got = impute_samples(devs, samples, prm, sample_offset=7, samples_per_launch_set=2, drcs=drcs)
    for x in drcs + [drc]:
        x.close()
    for d in devs + [dev]:
        d.close()
    for a, b in zip(got, want):
        assert a.dosage.shape == (rc.nSNPs_all,)
        _same(a, b)


def test_native_driver_nipt_rare_common_on_the_device(medium_panel):
    """method = "nipt" together with impute_rare_common through qa_impute_samples on the device == quilt_amd/driver.py on the
    device, bit for bit (mother and fetus over all SNPs, three phased haplotypes, labels)."""
    from quilt_amd.driver import Driver, DriverParams, HipBackend
    from quilt_amd.impute import impute_samples
    from quilt_amd.native import DevicePanel, DeviceRareCommon
    from quilt_amd.synth import make_rare_common, make_synthetic_sample_rare_common
    panel = medium_panel
    rc = make_rare_common(panel, 4)
    samples = []
    for i in range(3):
        s = make_synthetic_sample_rare_common(panel, rc, 2700 + i, n_reads=500)[0]
        s.ff = 0.12 + 0.05 * i
        samples.append(s)
    prm = DriverParams(nGibbsSamples=2, Ksubset=128, Knew=128, seed=10, method="nipt", impute_rare_common=True)
    dev = DevicePanel(panel)
    dev.set_dosage_precision(64)
    drc = DeviceRareCommon(dev, rc)
    want = Driver(panel, HipBackend(dev, drc), prm, rare_common=rc).run(samples, sample_offset=2)
    got = impute_samples([dev], samples, prm, sample_offset=2, samples_per_launch_set=2, drcs=[drc])
    drc.close()
    dev.close()
    for a, b in zip(got, want):
        assert a.dosage.shape == (rc.nSNPs_all,) and a.phasing_haps.shape == (rc.nSNPs_all, 3)
        _same(a, b)
        assert np.array_equal(a.fet_dosage, b.fet_dosage) and np.array_equal(a.fet_gp_t, b.fet_gp_t)


def test_native_driver_nipt_on_the_device(medium_panel):
    """method = "nipt" through qa_impute_samples on the device == quilt_amd/driver.py on the device, bit for bit (mother and fetus,
    three phased haplotypes, consensus labels), samples with different fetal fractions, two host threads."""
    from quilt_amd.driver import Driver, DriverParams, HipBackend
    from quilt_amd.impute import impute_samples
    from quilt_amd.native import DevicePanel
    from quilt_amd.synth import make_synthetic_sample
    panel = medium_panel
    samples = [make_synthetic_sample(panel, seed=4300 + i, n_reads=500, ff=0.1 + 0.04 * i) for i in range(5)]
    prm = DriverParams(nGibbsSamples=2, Ksubset=128, Knew=128, seed=12, method="nipt")
    dev = DevicePanel(panel)
    dev.set_dosage_precision(64)
    want = Driver(panel, HipBackend(dev), prm).run(samples, sample_offset=3)
    devs = [DevicePanel(panel) for _ in range(2)]
    for d in devs:
        d.set_device_share(2)
        d.set_dosage_precision(64)
        d.set_exclusive(True)
    got = impute_samples(devs, samples, prm, sample_offset=3, samples_per_launch_set=2)
    for d in devs + [dev]:
        d.close()
    for a, b in zip(got, want):
        assert a.phasing_haps.shape == (panel.nSNPs, 3)
        _same(a, b)
        assert np.array_equal(a.fet_dosage, b.fet_dosage) and np.array_equal(a.fet_gp_t, b.fet_gp_t)


def test_prepared_range_runs_again_with_the_same_result(medium_panel):
    """quilt_amd.impute.prepare_range + run_prepared (what bench.py times: the range already in the C ABI's flat form) ==
    impute_samples, and a second run over the same prepared range returns the same bytes -- the library zeroes a set's
    accumulator rows itself, whatever the caller's arrays held."""
    from quilt_amd.driver import DriverParams
    from quilt_amd.impute import impute_samples, prepare_range, run_prepared
    from quilt_amd.native import DevicePanel
    from quilt_amd.synth import make_synthetic_sample
    panel = medium_panel
    samples = [make_synthetic_sample(panel, seed=4400 + i, n_reads=300) for i in range(5)]
    prm = DriverParams(nGibbsSamples=2, Ksubset=128, Knew=128, seed=21)
    devs = [DevicePanel(panel) for _ in range(2)]
    for d in devs:
        d.set_device_share(2)
        d.set_dosage_precision(64)
        d.set_exclusive(True)
    want = impute_samples(devs, samples, prm, sample_offset=9, samples_per_launch_set=2)
    r = prepare_range(devs, samples, prm, sample_offset=9, samples_per_launch_set=2)
    first, stats = run_prepared(r, return_stats=True)
    again = run_prepared(r)
    for d in devs:
        d.close()
    assert stats["gibbs_launches"] > 0
    for a, b, c in zip(first, again, want):
        _same(a, c)
        _same(b, c)
    assert not first[0].phasing_haps.flags.owndata and first[0].phasing_haps.shape == (panel.nSNPs, 2)


@pytest.mark.parametrize("method", ["diploid", "nipt"])
def test_samples_handed_over_one_by_one_on_the_device(medium_panel, method):
    """params->sample_source on the device: qa_impute_samples handed each sample when its launch set is taken (what
    qa_impute_bam_range does while its loader threads are still reading later files) returns the bytes of the flat call."""
    from quilt_amd.driver import DriverParams
    from quilt_amd.impute import impute_samples
    from quilt_amd.native import DevicePanel
    from quilt_amd.synth import make_synthetic_sample
    panel = medium_panel
    samples = [make_synthetic_sample(panel, seed=4500 + i, n_reads=300 + 20 * i, ff=(0.1 + 0.03 * i) if method == "nipt" else 0.0)
               for i in range(7)]
    prm = DriverParams(nGibbsSamples=2, Ksubset=128, Knew=128, seed=23, method=method)
    devs = [DevicePanel(panel) for _ in range(3)]
    for d in devs:
        d.set_device_share(3)
        d.set_dosage_precision(64)
        d.set_exclusive(True)
    want = impute_samples(devs, samples, prm, sample_offset=5, samples_per_launch_set=2)
    got = impute_samples(devs, samples, prm, sample_offset=5, samples_per_launch_set=2, one_by_one=True)
    for d in devs:
        d.close()
    for a, b in zip(got, want):
        _same(a, b)
        if method == "nipt":
            assert np.array_equal(a.fet_dosage, b.fet_dosage) and np.array_equal(a.fet_gp_t, b.fet_gp_t)


def test_native_loop_equals_python_loop_over_seeds_that_reach_the_complete_lists_branch():
    """The panel and parameters of scripts/check_pipeline_seeds.py (K = 5 000 over 100 grids, Knew = Ksubset: the device selection
    regularly runs out of ranked candidates and both host loops fetch complete lists from genotype likelihoods THEY build): the
    native loop and the Python loop on the same device, seed after seed, bit for bit.  (Seeds 0 and 3 parted until the Python side
    took 10^(-q/10) from the C library's pow() like the native side: driver.phred_eps.)"""
    from quilt_amd.driver import Driver, DriverParams, HipBackend
    from quilt_amd.impute import impute_samples
    from quilt_amd.native import DevicePanel
    from quilt_amd.synth import make_synthetic_panel, make_synthetic_sample
    panel = make_synthetic_panel(K=5000, nSNPs=3200, seed=11, ref_error=1e-3, nGen=100, expRate=1.0)
    dev = DevicePanel(panel)
    dev.set_dosage_precision(64)
    refetches = 0
    for sd in range(8):
        samples = [make_synthetic_sample(panel, seed=5000 + 10 * sd + i, n_reads=800) for i in range(2)]
        prm = DriverParams(nGibbsSamples=3, seed=100 + sd, Ksubset=128, Knew=128)
        got, stats = impute_samples([dev], samples, prm, return_stats=True)
        want = Driver(panel, HipBackend(dev), prm).run(samples)
        refetches += stats["full_list_refetches"]
        for a, b in zip(got, want):
            _same(a, b)
    dev.close()
    assert refetches > 0


def test_handle_buffers_go_with_the_handle(medium_panel):
    """The pinned transfer buffers qa_impute_samples keeps per panel handle are dropped by qa_panel_destroy (a caller that creates and
    destroys its handles per call -- the shim's range routine -- used to leak gigabytes per call, and a later handle at a recycled
    address inherited the stale entry)."""
    from quilt_amd.driver import DriverParams
    from quilt_amd.impute import impute_samples
    from quilt_amd.native import DevicePanel, lib
    from quilt_amd.synth import make_synthetic_sample
    L = lib()
    import ctypes
    L.qa_impute_kept_buffers.restype = ctypes.c_int
    L.qa_impute_release_buffers()
    assert L.qa_impute_kept_buffers() == 0
    samples = [make_synthetic_sample(medium_panel, seed=4300 + i, n_reads=300) for i in range(2)]
    prm = DriverParams(nGibbsSamples=2, Ksubset=64, Knew=64, seed=3)
    for _ in range(2):
        dev = DevicePanel(medium_panel)
        impute_samples([dev], samples, prm)
        assert L.qa_impute_kept_buffers() == 1
        dev.close()
        assert L.qa_impute_kept_buffers() == 0


_SPLIT_SCRIPT = r"""
import hashlib, sys
sys.path.insert(0, %r)
import numpy as np
from quilt_amd.driver import DriverParams
from quilt_amd.impute import impute_samples
from quilt_amd.native import DevicePanel
from quilt_amd.synth import make_synthetic_panel, make_synthetic_sample
panel = make_synthetic_panel(K=3000, nSNPs=64000, seed=12)
samples = [make_synthetic_sample(panel, seed=5200 + i, n_reads=3000) for i in range(80)]
dev = DevicePanel(panel)
dev.set_dosage_precision(64)
dev.set_exclusive(True)
res = impute_samples([dev], samples, DriverParams(nGibbsSamples=7, n_seek_its=2, Ksubset=600, Knew=600, seed=4), samples_per_launch_set=80)
h = hashlib.sha256()
for r in res:
    for a in (r.dosage, r.gp_t, r.phasing_haps, r.read_labels):
        h.update(np.ascontiguousarray(a).tobytes())
print("DIGEST", h.hexdigest())
"""


def test_launches_cut_by_memory_give_the_same_results():
    """A Gibbs call that does not fit the arena is cut into equal launches; with qa_gibbs_opts_t.reads_same_as a chain's bases may
    then lie in ANOTHER launch's part of the caller's arrays (its sample's first chain went with the launch before).  Same samples
    with the arena at 10 %% of the device (the 560-chain calls cut in two: 61 MB of state per chain against 28 GB) and at the default:
    identical results."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for frac in ("0.1", None):
        env = dict(os.environ)
        env.pop("QA_ARENA_FRACTION", None)
        env["QA_TIMING"] = "1"   # ([qa_gibbs C=<chains of the launch>] ... on stderr)
        if frac:
            env["QA_ARENA_FRACTION"] = frac
        r = subprocess.run([sys.executable, "-c", _SPLIT_SCRIPT % root], capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("DIGEST")][0].split()
        import re
        chains = [int(x) for x in re.findall(r"\[qa_gibbs C=(\d+)\]", r.stderr)]
        out[frac] = (line[1], max(chains), len(chains))
    assert out["0.1"][0] == out[None][0]
    assert out[None][1] == 560 and out["0.1"][1] < 560 and out["0.1"][2] > out[None][2], out


def test_bam_range_call_with_rare_and_common_snps_on_the_device(tmp_path, medium_panel):
    """qa_impute_bam_range with impute_rare_common (+ use_mspbwt: QUILT2's default mode) on the device: every file piled up at the
    common SNPs and at all SNPs by the native loader, the all-SNP reads handed over in the same per-sample view -- equal to
    qa_impute_samples on the same files loaded through the Python binding of that loader, flat; columns cover all SNPs."""
    from quilt_amd.driver import DriverParams
    from quilt_amd.impute import impute_bam_range, impute_samples
    from quilt_amd.io import loadBamAndConvert, make_per_sample_vcf_col
    from quilt_amd.native import DevicePanel, DeviceRareCommon
    from quilt_amd.synth import make_rare_common, make_synthetic_sample_rare_common
    from tests import bamutil
    panel = medium_panel
    rc = make_rare_common(panel, 4)
    Ta = rc.nSNPs_all
    rng = np.random.default_rng(8)
    alleles = [tuple(rng.choice(list("ACGT"), size=2, replace=False)) for _ in range(Ta)]
    ref_all, alt_all = [a for a, _ in alleles], [b for _, b in alleles]
    common = np.flatnonzero(rc.snp_is_common == 1)
    ref, alt = [ref_all[i] for i in common], [alt_all[i] for i in common]
    grid_all = (np.arange(Ta) // 32).astype(np.int32)
    grid = panel.grid if panel.grid is not None else np.arange(panel.nSNPs, dtype=np.int32) // 32
    header = [("chr20", int(rc.L_all[-1]) + 1000)]
    files, samples = [], []
    opts = dict(downsampleToCov=0, bqFilter=1)
    for i in range(5):
        s_all = make_synthetic_sample_rare_common(panel, rc, 2800 + i, n_reads=400)[0].all_snp
        f = str(tmp_path / f"r{i}.bam")
        bamutil.write_bam(f, header, bamutil.sample_to_alignments(s_all, rc.L_all, ref_all, alt_all, rng))
        files.append(f)
        s = loadBamAndConvert(f, "chr20", panel.L, ref, alt, grid, **opts)
        s.all_snp = loadBamAndConvert(f, "chr20", rc.L_all, ref_all, alt_all, grid_all, **opts)
        samples.append(s)
    prm = DriverParams(nGibbsSamples=2, Ksubset=128, Knew=128, seed=8, impute_rare_common=True, use_mspbwt=True, mspbwt_nindices=2)
    devs = [DevicePanel(panel) for _ in range(2)]
    for d in devs:
        d.set_device_share(2)
        d.set_dosage_precision(64)
        d.set_exclusive(True)
    drcs = [DeviceRareCommon(d, rc) for d in devs]
    want = impute_samples(devs, samples, prm, sample_offset=20, samples_per_launch_set=2, drcs=drcs)
    got = impute_bam_range(devs, files, "chr20", ref, alt, prm, sample_index=list(range(20, 25)), drcs=drcs,
                           all_sites=(rc.L_all, ref_all, alt_all, grid_all), samples_per_launch_set=2, n_io_threads=3, **opts)
    for x in drcs:
        x.close()
    for d in devs:
        d.close()
    assert all(got["imputed"])
    for i, w in enumerate(want):
        g = got["results"][i]
        assert g.dosage.shape == (Ta,)
        _same(g, w)
        assert got["columns"][i].tolist() == make_per_sample_vcf_col(w.gp_t, w.phasing_haps, True).tolist()
    assert got["counts"].hweCount.sum() == 5 * Ta

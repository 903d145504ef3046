"""BASELINE.json's configs[0] and configs[1] AT THEIR STATED SIZE, from BAM files to VCF columns through the product's fast path
(qa_impute_bam_range: what shim/quilt-amd.R calls), on stand-ins for the data that cannot be had in this image.

  configs[0]  "Quick-start example: NA12878 1x BAM, chr20:2000001-4000000, 1000G panel (~5k haps)": ONE 1x sample against a
              K = 5 008 panel over the WHOLE 2 Mb region -- 64 000 SNPs, 2 000 grids (rounds 3-5 tested 100 grids) -- with QUILT's
              defaults.  The panel is quilt_amd.synth.make_1000g_like_panel (1 / i site-frequency spectrum, 27 % of the haplotypes
              repeat another one: the tie-richest panel there is), compressed ON THE DEVICE from its packed form.
              Validation mode (every K-wide sum in the order the reference's code adds it) must write the SAME TEXT as the CPU path
              from the same BAM; production mode must agree with the truth as well as the CPU path does; the wall time of the one
              sample -- what a new user's first run costs -- is recorded.
  configs[1]  "32 synthetic 1x short-read BAMs, chr20 2 Mb, K = 5 000 haps, 1 MI355X": the 32-sample job as ONE call, I/O inside
              the clock.  One launch set: all fill and drain -- the number a single small job achieves, recorded beside the
              steady-state bench line.  Parity at this size: size-independent properties on every sample, and two of the 32
              against the CPU path (read labels identical, dosage r2 >= 0.999: the metric's bar).

The measured numbers go to gpurun_out/configs_*.json when that directory exists (copied to profiles/ by hand).
"""
import json
import os
import time

import numpy as np
import pytest

from tests.util import r2

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _record(name, d):
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        json.dump(d, open(os.path.join(out, name), "w"), indent=1)
    print(name, json.dumps(d))


def _bams(tmp_path, panel, seeds, n_reads):
    from quilt_amd.synth import make_synthetic_sample, synthetic_alleles, write_synthetic_bam
    ref, alt = synthetic_alleles(panel.nSNPs, 1)
    samples, files = [], []
    for sd in seeds:
        s = make_synthetic_sample(panel, seed=sd, n_reads=n_reads)
        f = str(tmp_path / f"s{sd}.bam")
        write_synthetic_bam(f, s, panel.L, ref, alt, seed=sd)
        samples.append(s)
        files.append(f)
    return samples, files, ref, alt


def _cpu_columns(tmp_path, panel, files, names, ref, alt, prm, n_threads=8):
    """The CPU path from the same BAM files: loader -> Python driver over the oracle -> column writers."""
    from quilt_amd.io import impute_bams_to_vcf
    from tests.oracle_backend import OracleBackend
    rec = impute_bams_to_vcf(panel, OracleBackend(panel, n_threads=n_threads), files, names, "chr20", list(ref), list(alt),
                             str(tmp_path / "cpu.vcf.gz"), params=prm, downsampleToCov=0, bqFilter=1)
    return rec


def test_configs0_quick_start_over_the_whole_region(tmp_path):
    from quilt_amd.driver import DriverParams
    from quilt_amd.impute import impute_bam_range
    from quilt_amd.native import DevicePanel
    from quilt_amd.synth import make_1000g_like_panel
    panel = make_1000g_like_panel(K=5008, nSNPs=64000, seed=2504)
    assert panel.nGrids == 2000
    samples, files, ref, alt = _bams(tmp_path, panel, [4001], 20000)
    prm = DriverParams(seed=3)   # QUILT's defaults: nGibbsSamples = 7, n_seek_its = 3, Ksubset = Knew = 600
    dev = DevicePanel.from_rhb(panel)
    dev.set_dosage_precision(64)
    kw = dict(downsampleToCov=0, bqFilter=1, samples_per_launch_set=1)
    impute_bam_range([dev], files, "chr20", ref, alt, prm, **kw)   # (first call: module load, arenas, the msPBWT-free warm-up)
    t0 = time.perf_counter()
    got = impute_bam_range([dev], files, "chr20", ref, alt, prm, **kw)
    wall = time.perf_counter() - t0
    dev.set_sum_order(1)
    t0 = time.perf_counter()
    val = impute_bam_range([dev], files, "chr20", ref, alt, prm, **kw)
    wall_val = time.perf_counter() - t0
    dev.close()
    t0 = time.perf_counter()
    cpu = _cpu_columns(tmp_path, panel, files, ["NA0"], ref, alt, prm)
    wall_cpu = time.perf_counter() - t0
    # validation mode: the CPU path's text, entry for entry, and its read labels
    assert val["columns"][0].tolist() == cpu["columns"][0].tolist()
    assert np.array_equal(val["results"][0].read_labels, cpu["results"][0].read_labels)
    assert np.array_equal(val["results"][0].dosage, cpu["results"][0].dosage)
    for name in ("infoCount", "afCount", "hweCount", "alleleCount"):
        assert np.array_equal(getattr(val["counts"], name), getattr(cpu["counts"], name)), name
    # production mode: identical to the CPU path, or (a last-bit tie on this duplicate-rich panel) as close to the truth as it is
    truth = samples[0].truth_haps[:2].sum(axis=0).astype(float)
    g, c = got["results"][0], cpu["results"][0]
    same = np.array_equal(g.read_labels, c.read_labels) and np.abs(g.dosage - c.dosage).max() <= 1e-9
    assert np.isfinite(g.dosage).all() and np.abs(g.gp_t.sum(axis=0) - 1).max() <= 1e-9
    assert r2(g.dosage, truth) > 0.9 and abs(r2(g.dosage, truth) - r2(c.dosage, truth)) <= 0.01
    assert wall < 30, wall   # (3-4 s expected: the chains' serial time; a bound against a silent fallback, not a target)
    _record("configs0_quick_start.json", dict(
        workload="configs[0] stand-in at the region's full size: K = 5 008 (make_1000g_like_panel) x 64 000 SNPs / 2 000 grids, one 1x sample "
                 "(20 000 reads) from a BAM file to its VCF column, QUILT defaults, qa_impute_bam_range, one host thread",
        wall_s_production=round(wall, 3), seconds=got["seconds"], wall_s_validation_mode=round(wall_val, 3),
        wall_s_cpu_path_8_threads=round(wall_cpu, 3), validation_mode_text_equals_cpu_path=True,
        production_mode_identical_to_cpu_path=bool(same), r2_vs_truth=dict(gpu=r2(g.dosage, truth), cpu=r2(c.dosage, truth)),
        r2_gpu_vs_cpu=r2(g.dosage, c.dosage)))


def test_configs1_thirty_two_samples_one_call(tmp_path):
    from quilt_amd.driver import Driver, DriverParams
    from quilt_amd.impute import impute_bam_range
    from quilt_amd.native import DevicePanel
    from quilt_amd.synth import make_synthetic_panel
    from tests.oracle_backend import OracleBackend
    panel = make_synthetic_panel(K=5000, nSNPs=64000, seed=77)
    seeds = list(range(6000, 6032))
    samples, files, ref, alt = _bams(tmp_path, panel, seeds, 20000)
    prm = DriverParams(seed=11)
    devs = [DevicePanel(panel) for _ in range(3)]
    for d in devs:
        d.set_device_share(3)
        d.set_exclusive(True)
        d.set_dosage_precision(64)
    kw = dict(downsampleToCov=0, bqFilter=1, samples_per_launch_set=32)
    impute_bam_range(devs, files[:2], "chr20", ref, alt, prm, **kw)   # (warm-up: arenas, pinned buffers)
    t0 = time.perf_counter()
    got = impute_bam_range(devs, files, "chr20", ref, alt, prm, **kw)
    wall = time.perf_counter() - t0
    for d in devs:
        d.close()
    assert all(got["imputed"]) and len(got["results"]) == 32
    r2s = []
    for i, s in enumerate(samples):
        r = got["results"][i]
        truth = s.truth_haps[:2].sum(axis=0).astype(float)
        assert np.isfinite(r.dosage).all() and np.abs(r.gp_t.sum(axis=0) - 1).max() <= 1e-9
        assert np.abs(r.dosage - (r.gp_t[1] + 2 * r.gp_t[2])).max() <= 1e-9        # the dosage IS the posteriors' mean
        assert set(np.unique(r.read_labels)) <= {1, 2} and len(r.read_labels) == s.nReads
        col = got["columns"][i]
        assert len(col) == panel.nSNPs and col[0].count(":") == 3
        r2s.append(r2(r.dosage, truth))
    assert min(r2s) > 0.97, min(r2s)
    # two of the 32 against the CPU path (same reads, same global indices 0 and 31)
    cpu = Driver(panel, OracleBackend(panel, n_threads=8), prm)
    worst = 1.0
    for i in (0, 31):
        from quilt_amd.io import loadBamAndConvert
        s = loadBamAndConvert(files[i], "chr20", panel.L, list(ref), list(alt), panel.grid, downsampleToCov=0, bqFilter=1)
        c = cpu.run([s], sample_offset=i)[0]
        g = got["results"][i]
        assert np.array_equal(g.read_labels, c.read_labels)
        worst = min(worst, r2(g.dosage, c.dosage))
        assert np.abs(g.dosage - c.dosage).max() <= 1e-9
    assert worst >= 0.999
    _record("configs1_32_samples_one_call.json", dict(
        workload="configs[1] as BASELINE states it: 32 synthetic 1x BAMs (20 000 reads), K = 5 000 x 64 000 SNPs / 2 000 grids, ONE "
                 "qa_impute_bam_range call (three host threads, one launch set of 32 samples = 224 chains), load and format inside the clock",
        samples_per_s_incl_io=32 / wall, wall_s=round(wall, 3), seconds=got["seconds"],
        r2_vs_truth_min=min(r2s), r2_vs_truth_mean=float(np.mean(r2s)), r2_vs_cpu_path_worst_of_2=worst,
        note="a single 32-sample job is one launch set: every Gibbs launch costs a chain's serial time with a quarter of the SIMDs "
             "in use; the steady state of many such jobs is the configs[1] bench line"))

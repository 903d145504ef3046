"""Shared by tests/test_sum_order_gpu.py and scripts/check_sum_order_seeds.py (test infrastructure): the two seed sweeps on which
the device pipeline and the CPU pipeline were seen to part (round 4: scripts/check_quick_start_seeds.py, ten driver seeds on the
quick-start-shaped panel; scripts/check_seed_lists.py, 24 seeds on the K = 5 000 panel), run through the native loop
(qa_impute_samples) in both modes of the full-panel passes and through the CPU pipeline (the oracle backend) twice -- the second
time under another driver seed, which gives the sampler's own run-to-run spread for the same reads."""
import dataclasses

import numpy as np


def _r2(a, b):
    return float(np.corrcoef(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64))[0, 1] ** 2)


def _quick_start():
    from quilt_amd.driver import DriverParams
    from quilt_amd.synth import make_1000g_like_panel, make_synthetic_sample
    panel = make_1000g_like_panel(K=5008, nSNPs=3200, seed=2504)   # 1 / i spectrum; 27 % of the haplotypes repeat another one
    sample = make_synthetic_sample(panel, seed=77, n_reads=1000)
    return panel, True, [(sd, sample, DriverParams(seed=sd)) for sd in range(1, 11)]   # QUILT's defaults: 7 x 3 rounds, Ksubset 600


def _medium():
    from quilt_amd.driver import DriverParams
    from quilt_amd.synth import make_synthetic_panel, make_synthetic_sample
    panel = make_synthetic_panel(K=5000, nSNPs=3200, seed=11, ref_error=1e-3, nGen=100, expRate=1.0)
    return panel, False, [(sd, make_synthetic_sample(panel, seed=5000 + 10 * sd, n_reads=800),
                           DriverParams(nGibbsSamples=3, seed=100 + sd, Ksubset=128, Knew=128)) for sd in range(1, 25)]


def _medium_nipt():
    """method = "nipt" (mother + fetus, three full-panel passes per chain, block Gibbs) on the K = 5 000 panel: eight seeds."""
    from quilt_amd.driver import DriverParams
    from quilt_amd.synth import make_synthetic_panel, make_synthetic_sample
    panel = make_synthetic_panel(K=5000, nSNPs=3200, seed=11, ref_error=1e-3, nGen=100, expRate=1.0)
    return panel, False, [(sd, make_synthetic_sample(panel, seed=7000 + 10 * sd, n_reads=1000, ff=0.2),
                           DriverParams(nGibbsSamples=3, seed=300 + sd, Ksubset=128, Knew=128, method="nipt")) for sd in range(1, 9)]


CASES = {"quick_start": _quick_start, "medium": _medium, "medium_nipt": _medium_nipt}


def run_case(name, verbose=False):
    """One row per seed: validation mode and production mode against the CPU pipeline, the CPU pipeline's own spread, r2 vs truth."""
    from quilt_amd.driver import Driver
    from quilt_amd.impute import impute_samples
    from quilt_amd.native import DevicePanel
    from tests.oracle_backend import OracleBackend
    panel, from_rhb, seeds = CASES[name]()
    make = DevicePanel.from_rhb if from_rhb else DevicePanel   # (from_rhb: the panel compressed on the device, as the quick start's)
    prod, val = make(panel), make(panel)
    for d in (prod, val):
        d.set_dosage_precision(64)
    val.set_sum_order(True)
    cpu_be = OracleBackend(panel, n_threads=8)
    rows = []
    for sd, sample, prm in seeds:
        truth = sample.truth_haps[:2].sum(axis=0).astype(float)
        cpu = Driver(panel, cpu_be, prm).run([sample])[0]
        cpu2 = Driver(panel, cpu_be, dataclasses.replace(prm, seed=prm.seed + 1000)).run([sample])[0]
        gv = impute_samples([val], [sample], prm)[0]
        gp = impute_samples([prod], [sample], prm)[0]
        row = dict(seed=sd,
                   val_labels_identical=bool(np.array_equal(gv.read_labels, cpu.read_labels)),
                   val_dosage_identical=bool(np.array_equal(gv.dosage, cpu.dosage)),
                   val_dosage_maxdiff=float(np.abs(gv.dosage - cpu.dosage).max()),
                   val_gp_identical=bool(np.array_equal(gv.gp_t, cpu.gp_t)),
                   val_phase_identical=bool(np.array_equal(gv.phasing_haps, cpu.phasing_haps)),
                   prod_labels_identical=bool(np.array_equal(gp.read_labels, cpu.read_labels)),
                   prod_dosage_maxdiff=float(np.abs(gp.dosage - cpu.dosage).max()),
                   prod_r2_vs_cpu=_r2(gp.dosage, cpu.dosage), floor_r2=_r2(cpu2.dosage, cpu.dosage),
                   cpu_r2_truth=_r2(cpu.dosage, truth), cpu2_r2_truth=_r2(cpu2.dosage, truth), prod_r2_truth=_r2(gp.dosage, truth))
        rows.append(row)
        if verbose:
            print(name, row, flush=True)
    prod.close()
    val.close()
    return rows

#!/usr/bin/env python
"""bench.py -- samples/sec of the QUILT per-sample hot path on MI355X ("Measurement").

One "step" = one batch of synthetic 1x samples taken through the whole per-sample driver (7 + 1 Gibbs
chains x 3 rounds of [small-panel Gibbs -> full-panel forward/backward per read label -> haplotype
re-selection], reference defaults) against a synthetic K-haplotype panel.  Samples are independent:
with N GPUs every rank imputes its own batch (weak scaling), no collective on the data path.

`--gpus N` (N > 1) re-executes itself under torch.distributed.run with N ranks when it was not launched
that way already (the driver launches it as `python -m torch.distributed.run ... bench.py --gpus N`).

Prints ONE JSON line (rank 0) with the contract fields plus `roofline` (dominant kernel, HIP-event
timed inside the library on its launch stream), `cpu_baseline` (the fp64 C oracle, one sample per
physical host core, on a bounded share of the same workload; N = 1 only) and `parity_vs_cpu_path`
(the GPU re-running the baseline's own calls: labels, dosage r2).
"""
import argparse
import json
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md; 6.3 TB/s is what a streaming copy achieves)
_CPU_PANEL = None
PMC_INSTS_FILE = os.path.join("profiles", "r05_pmc_insts.json")


def pmc_file_for(mode, K, batch, mspbwt, rare_common):
    """The committed counter summary (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, scripts/pmc_summary.py) of THIS workload, newest
    round first -- every bench line prices its dominant kernel on counter traffic of its own workload, or says it has none."""
    if mspbwt and rare_common:
        names = ["r05_pmc_traffic_quilt2_default.json"]
    elif mspbwt and mode == "short" and K == 50000:
        names = ["r05_pmc_traffic_mspbwt.json"]
    elif mspbwt or rare_common:
        names = []
    elif mode == "ont" and K == 50000:
        names = ["r05_pmc_traffic_ont.json"]
    elif mode == "short" and batch == 128 and K == 64976:
        names = ["r05_pmc_traffic_K64976.json"]
    elif mode == "nipt" and K == 50000:
        names = ["r05_pmc_traffic_nipt.json"]
    elif mode == "short" and K == 5000 and batch == 32:
        names = ["r05_pmc_traffic_configs1.json"]
    elif mode == "short" and K == 50000 and batch == 128:
        names = ["r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json"]
    else:
        names = []
    for n in names:
        if os.path.exists(os.path.join(ROOT, "profiles", n)):
            return os.path.join("profiles", n)
    return None

def workload_label(mode, K, batch, mspbwt=False, rare_common=False):
    """Which BASELINE.json configuration a run is (or that it is none): keyed on the read model AND the panel size / batch."""
    if mode == "ont":
        base = "BASELINE.json configs[3] (ONT-style long reads, noisy base qualities)" if K == 50000 else None
    elif mode == "nipt":
        base = "BASELINE.json configs[4] (NIPT: mother + fetus, three read labels, block Gibbs; ff = 0.2)" if K == 50000 else None
    elif K == 50000:
        base = "BASELINE.json configs[2] (the K=50k configuration the metric is quoted on)"
    elif K == 5000 and batch == 32:
        base = "BASELINE.json configs[1] (32 synthetic 1x short-read samples, K=5 000, one MI355X)"
    elif K == 64976:
        base = "not a BASELINE.json configuration: the HRC panel's size (K=64 976; the reference developers' profiling workload, scripts/profile.R:70-107)"
    else:
        base = None
    if base is None:
        base = f"not a BASELINE.json configuration (mode {mode}, K={K}, {batch} samples per step)"
    if mspbwt and rare_common:
        base += " in QUILT2's default mode (use_mspbwt=TRUE, impute_rare_common=TRUE)"
    elif mspbwt:
        base += " in mode M2 (use_mspbwt=TRUE)"
    elif rare_common:
        base += " with impute_rare_common=TRUE"
    return base + ", per-GPU share"


def physical_cores():
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:
        pass
    return os.cpu_count() or 1


def _cpu_inputs(panel, n_reads, i, Ksubset, ff):
    """The inputs of one host core's baseline calls (deterministic in i: the GPU re-runs core 0's for the parity check)."""
    from quilt_amd.driver import thinned_grid_columns
    from quilt_amd.synth import make_synthetic_sample
    s = make_synthetic_sample(panel, seed=1000 + i, n_reads=n_reads, ff=ff)
    rng = np.random.default_rng(i)
    which = np.sort(rng.choice(panel.K, min(Ksubset, panel.K), replace=False)).astype(np.int32) + 1
    R = s.nReads
    H0 = ((rng.choice(3, size=R, p=[0.5, 0.5 - ff / 2, ff / 2]) + 1) if ff > 0 else rng.integers(1, 3, size=R)).astype(np.int32)
    ru, rs = rng.random(R * 21), rng.random(3 * (panel.nGrids - 1))
    nipt = dict(ff=ff, runif_block=rng.random(3 * R), runif_resample=rng.random(3 * R)) if ff > 0 else {}
    cols = thinned_grid_columns(panel.nGrids, 0.1)
    return s, which, H0, ru, rs, nipt, cols


def _cpu_worker(args):
    """One host core: the three native calls of the per-sample loop on the fp64 oracle, once each, for
    this core's own synthetic sample (so that all cores contend for memory as forked R workers do)."""
    n_reads, i, Ksubset, ff = args
    from oracle import oracle as O
    panel = _CPU_PANEL   # built once in the parent, shared copy-on-write by the forked workers
    s, which, H0, ru, rs, nipt, cols = _cpu_inputs(panel, n_reads, i, Ksubset, ff)
    t0 = time.perf_counter()
    g = O.forwardBackwardGibbsNIPT(panel, s, which, H0, ru, 0, rs, **nipt)
    t1 = time.perf_counter()
    per_base = np.repeat(g["H"], np.diff(s.read_ptr))
    sel = (per_base == 1) & (s.bq != 0)
    gl = O.make_gl_from_u_bq(s.u[sel], s.bq[sel], panel.nSNPs)
    t2 = time.perf_counter()
    O.haploid_dosage_versus_refs(panel, gl, cols, return_dosage=False, get_best_haps_from_thinned_sites=True)
    t3 = time.perf_counter()
    d = O.haploid_dosage_versus_refs(panel, gl, cols, return_dosage=True, get_best_haps_from_thinned_sites=True)
    t4 = time.perf_counter()
    keep = (g["H"], d["dosage"], d["best_haps"]) if i == 0 else None
    return (t1 - t0, t3 - t2, t4 - t3), keep


_CPU_RC = None


_BAM_DIR = None


def _sample_worker(args):
    from quilt_amd.synth import make_synthetic_sample, make_synthetic_sample_rare_common
    seed, n_reads, mode = args
    if mode == "nipt" and _CPU_RC is not None:   # NIPT with impute_rare_common: the mixture read over all SNPs as well
        s = make_synthetic_sample_rare_common(_CPU_PANEL, _CPU_RC, seed, n_reads=n_reads, ff=0.2)[0]
        s.ff = 0.2
    elif mode == "nipt":   # BASELINE configs[4]: mother + fetus, one fetal fraction for the batch
        s = make_synthetic_sample(_CPU_PANEL, seed=seed, n_reads=n_reads, ff=0.2)
    elif _CPU_RC is not None:   # the sample read over the common SNPs, with its all-SNP reads attached
        s = make_synthetic_sample_rare_common(_CPU_PANEL, _CPU_RC, seed, n_reads=n_reads)[0]
    else:
        s = make_synthetic_sample(_CPU_PANEL, seed=seed, n_reads=n_reads, mode=mode)
    if _BAM_DIR is not None:   # --bam: the sample also as a BAM file, which the run then reads back through the loader
        from quilt_amd.synth import synthetic_alleles, write_synthetic_bam
        ref, alt = synthetic_alleles(_CPU_PANEL.nSNPs, 1)
        write_synthetic_bam(os.path.join(_BAM_DIR, f"s{seed}.bam"), s, _CPU_PANEL.L, ref, alt, seed=seed)
    return s


def make_samples(panel, seeds, n_reads, n_proc, mode="short", rare_common=None, bam_dir=None):
    """Synthetic samples, generated by forked workers (before any HIP context exists in this process)."""
    import multiprocessing as mp
    global _CPU_PANEL, _CPU_RC, _BAM_DIR
    _CPU_PANEL = panel
    _CPU_RC = rare_common
    _BAM_DIR = bam_dir
    with mp.get_context("fork").Pool(max(1, n_proc)) as pool:
        return pool.map(_sample_worker, [(sd, n_reads, mode) for sd in seeds], chunksize=4)


def reload_from_bams(panel, flat, seeds, bam_dir):
    """--bam: every sample is read back from its BAM file by the native loader (qa_bam_load_sample_reads)
    and THAT is what the timed run imputes.  Returns the loaded samples and the loader's time per sample."""
    from quilt_amd.io import loadBamAndConvert
    from quilt_amd.synth import synthetic_alleles
    ref, alt = synthetic_alleles(panel.nSNPs, 1)
    out, t0 = [], time.perf_counter()
    for s, sd in zip(flat, seeds):
        # (bqFilter = 1, no coverage cap: the synthetic base qualities -- 5-15 in ONT mode -- and depths are the workload itself)
        g = loadBamAndConvert(os.path.join(bam_dir, f"s{sd}.bam"), "chr20", panel.L, ref, alt, panel.grid, downsampleToCov=0,
                              bqFilter=1)
        if g.nReads != s.nReads or len(g.u) != len(s.u):
            raise RuntimeError("the BAM loader did not give back the sample's reads")
        g.truth_haps, g.ff, g.all_snp = s.truth_haps, s.ff, s.all_snp
        out.append(g)
    return out, (time.perf_counter() - t0) / max(len(flat), 1)


def cpu_baseline(panel, n_reads, params, full_chains, ff=0.0, mspbwt=False):
    """CPU baseline on a BOUNDED SAMPLE (no whole sample is imputed on the CPU: one costs ~20 core-minutes): every
    physical host core (one synthetic sample each, mirroring the reference's mclapply sharding, quilt.R:691-692) runs ONE
    small-panel Gibbs call, ONE thin full-panel pass and ONE dosage full-panel pass on the fp64 oracle; a sample costs
    chains x n_seek_its Gibbs calls, and per Gibbs call one full-panel pass per read label of which 1 in n_seek_its
    computes dosages.  The per-sample cost is composed from the mean call times over the cores
    (the max is reported too).  The R interpreter overhead of the real driver is not included, so this baseline is
    faster than the reference."""
    import multiprocessing as mp
    global _CPU_PANEL
    from oracle import oracle as O
    O.lib()
    _CPU_PANEL = panel
    cores = physical_cores()
    ctx = mp.get_context("fork")
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(n_reads, i, params["Ksubset"], ff) for i in range(cores)])
    wall = time.perf_counter() - t0
    times = np.array([r[0] for r in res])
    keep = res[0][1]
    n_calls = full_chains * params["n_seek_its"]
    n_lab = 3 if ff > 0 else 2   # one full-panel pass per read label and Gibbs call (impute_using_everything)
    def per_sample(t):
        tg, tt, td = t
        if mspbwt:   # use_mspbwt = TRUE: no full-panel pass; the indexed query of the reference costs milliseconds: priced at 0
            return n_calls * tg
        return n_calls * tg + n_lab * (n_calls - full_chains) * tt + n_lab * full_chains * td
    ps_mean, ps_max = per_sample(times.mean(axis=0)), per_sample(times.max(axis=0))
    tg, tt, td = times.mean(axis=0)
    out = dict(value=float(cores / ps_mean), unit="samples/sec", cores=cores, logical_cpus=os.cpu_count(), kind="port",
               value_1_core=float(1.0 / ps_mean), value_slowest_core=float(cores / ps_max),
               sample=f"bounded sample, composed (not a whole-sample run): per physical core 1 Gibbs call ({tg:.2f} s mean), "
                      f"1 thin pass ({tt:.2f} s), 1 dosage pass ({td:.2f} s) of one synthetic sample, all {cores} cores "
                      f"concurrently; per-sample cost = {n_calls} Gibbs calls + {n_lab * (n_calls - full_chains)} thin + "
                      f"{n_lab * full_chains} dosage passes = {ps_mean:.1f} s/core mean ({ps_max:.1f} s slowest core); "
                      f"{wall:.0f} s of wall time")
    if mspbwt:
        out["sample"] = (f"bounded sample, composed (not a whole-sample run): per physical core 1 Gibbs call ({tg:.2f} s mean) of one "
                         f"synthetic sample, all {cores} cores concurrently; use_mspbwt = TRUE: per-sample cost = {n_calls} Gibbs calls "
                         f"= {ps_mean:.1f} s/core mean ({ps_max:.1f} s slowest core); the msPBWT queries between them (milliseconds "
                         f"each with the reference's index) are priced at 0, so this baseline is faster than the reference; "
                         f"{wall:.0f} s of wall time")
    return out, keep


_WHOLE = None


def _whole_worker(i):
    """One host core: ONE WHOLE SAMPLE through the entire per-sample pipeline on the CPU path (the driver loop over the fp64
    oracle's entry points, single-threaded) -- what one mclapply worker of the reference does for one sample."""
    from quilt_amd.driver import Driver, DriverParams
    from tests.oracle_backend import OracleBackend

    class Backend(OracleBackend):
        # use_mspbwt = TRUE: the reference answers its msPBWT queries with the mspbwt package's C++ on the host; the CPU path here
        # uses this library's host C++ for the same query (csrc/mspbwt.cpp: no device involved) -- the numpy restatement the tests
        # check it against would understate the CPU reference by an order of magnitude.  The panel's indices were built once in
        # the parent (copy-on-write after the fork), as QUILT loads `ms_indices` once per run.
        def mspbwt_select(self, Zs, n_label, nindices, L, M, Knew, seeds):
            from quilt_amd.mspbwt import panel_mspbwt_index
            return panel_mspbwt_index(self.panel, nindices).select_new_haps(Zs, n_label, L, M, Knew, seeds)

    smp, off, params, keep = _WHOLE[i]
    os.environ["QA_HOST_THREADS"] = "1"   # (this worker process only: a query's scan runs on the worker's own core, like the rest of its sample)
    t0 = time.perf_counter()
    drv = Driver(_CPU_PANEL, Backend(_CPU_PANEL, _CPU_RC, n_threads=1), DriverParams(**params), rare_common=_CPU_RC)
    res = drv.run([smp], sample_offset=off)
    return i, time.perf_counter() - t0, (res[0] if keep else None)


def cpu_baseline_whole(panel, params, work, cores, budget_s, n_keep, rare_common=None):
    """THE CPU baseline: `cores` whole samples, one per physical core, all at once (the reference's
    mclapply(mc.cores = nCores), quilt.R:691-692), each through the whole per-sample pipeline on the CPU path.  `work` =
    [(sample, global sample index)]: the samples of the last timed batch under the seeds the GPU run gives them, so that the
    first n_keep results are also the metric's `dosage r2 vs CPU ref` reference.  Bounded by `budget_s` of wall time: workers
    still running then are stopped and the figure comes from the samples that finished (flagged)."""
    import multiprocessing as mp
    global _CPU_PANEL, _WHOLE
    from oracle import oracle as O
    global _CPU_RC
    O.lib()
    _CPU_PANEL = panel
    _CPU_RC = rare_common
    if params.get("use_mspbwt"):
        from quilt_amd.mspbwt import panel_mspbwt_index
        panel_mspbwt_index(panel, params.get("mspbwt_nindices", 4))   # built before the fork: shared by the workers
    n = min(cores, len(work))
    _WHOLE = [(smp, off, params, i < n_keep) for i, (smp, off) in enumerate(work[:n])]
    t0 = time.perf_counter()
    secs, kept = {}, {}
    pool = mp.get_context("fork").Pool(n)
    try:
        it = pool.imap_unordered(_whole_worker, range(n))
        for _ in range(n):
            left = budget_s - (time.perf_counter() - t0)
            if left <= 0:
                break
            try:
                i, t, res = it.next(timeout=left)
            except mp.TimeoutError:
                break
            secs[i] = t
            if res is not None:
                kept[i] = res
    finally:
        pool.terminate()
        pool.join()
    wall = time.perf_counter() - t0
    if not secs:
        return None, None
    t = np.array([secs[i] for i in sorted(secs)])
    out = dict(value=float((1.0 / t).sum() * n / len(t)), unit="samples/sec", cores=n, logical_cpus=os.cpu_count(), kind="port",
               whole_samples=len(t), seconds_per_sample_mean=float(t.mean()), seconds_per_sample_max=float(t.max()),
               seconds_per_sample_min=float(t.min()), value_batch_wall=float(len(t) / wall), wall_s=round(wall, 1),
               sample=f"WHOLE samples: {n} synthetic samples of the last timed batch, one per physical core, all {n} cores at once "
                      f"(mclapply's sharding), each through the entire per-sample pipeline on the CPU path (every chain, every round, "
                      f"the driver's host logic; fp64 C oracle, one thread per sample); {len(t)} finished: {t.mean():.1f} s per sample "
                      f"mean ({t.min():.1f} .. {t.max():.1f}); value = sum over cores of 1 / seconds (each core's own rate); "
                      f"{wall:.0f} s of wall time")
    if len(t) < n:
        out["incomplete"] = f"{n - len(t)} of {n} workers were stopped at the {budget_s:.0f} s budget; value scales the finished ones' rates to {n} cores"
    ref = None
    if len(kept) == n_keep and n_keep > 0:
        ref = dict(ref=[kept[i] for i in range(n_keep)], n=n_keep, cpu_seconds=round(max(secs[i] for i in range(n_keep)), 1), threads=1)
    return out, ref


def parity_vs_cpu(dev, panel, n_reads, params, ff, keep):
    """The GPU re-runs core 0's baseline calls on the same inputs (same uniforms): read labels must be identical, the
    dosage of the following full-panel pass is compared by r2 and max |diff|, the best-haplotype lists by identity."""
    from quilt_amd.gibbs_nipt import rcpp_forwardBackwardGibbsNIPT
    from quilt_amd.reference_single import Rcpp_haploid_dosage_versus_refs
    from oracle import oracle as O
    H_cpu, dos_cpu, best_cpu = keep
    s, which, H0, ru, rs, nipt, cols = _cpu_inputs(panel, n_reads, 0, params["Ksubset"], ff)
    if ff > 0:
        got = rcpp_forwardBackwardGibbsNIPT(dev, s, which, H0, ru, 0, rs, ff=ff, runif_block=nipt["runif_block"],
                                            runif_resample=nipt["runif_resample"])
    else:
        got = rcpp_forwardBackwardGibbsNIPT(dev, s, which, H0, ru, 0, rs)
    labels_equal = bool(np.array_equal(got["H"], H_cpu))
    per_base = np.repeat(H_cpu, np.diff(s.read_ptr))
    sel = (per_base == 1) & (s.bq != 0)
    gl = O.make_gl_from_u_bq(s.u[sel], s.bq[sel], panel.nSNPs)   # (host marshalling of the shared input)
    n_thin = int((cols >= 0).sum())
    dosage = np.zeros(panel.nSNPs)
    best = [None] * n_thin
    Rcpp_haploid_dosage_versus_refs(dev, gl, gammaSmall_cols_to_get=cols, dosage=dosage, best_haps_stuff_list=best,
                                    return_dosage=True, return_betaHat_t=False, return_gamma_t=False,
                                    get_best_haps_from_thinned_sites=True)
    lists_equal = all(np.array_equal(b["top_matches"], o[0]) for b, o in zip(best, best_cpu))
    r2 = float(np.corrcoef(dosage, dos_cpu)[0, 1] ** 2)
    return dict(scope="core 0's cpu_baseline calls re-run on the GPU with the same inputs and uniforms: 1 Gibbs call "
                      f"(Ks={params['Ksubset']}, {s.nReads} reads, {panel.nGrids} grids), 1 dosage pass (K={panel.K})",
                gibbs_labels_identical=labels_equal, n_reads=int(s.nReads),
                dosage_r2=r2, dosage_max_abs_diff=float(np.abs(dosage - dos_cpu).max()),
                best_haps_lists_identical=bool(lists_equal))


def maybe_spawn(a, argv):
    """`--gpus N` without a launcher: become `python -m torch.distributed.run --nproc-per-node N bench.py ...`."""
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
        os.execv(sys.executable, cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--K", type=int, default=50000)
    ap.add_argument("--nsnps", type=int, default=64000)
    ap.add_argument("--batch", type=int, default=128, help="samples per step per GPU")
    ap.add_argument("--fuse", type=int, default=None, metavar="N",
                    help="steps a host thread takes per launch set (default 2; 1 with --mode ont / nipt): 2 x 128 samples are 2 048 "
                         "Gibbs chains, two per SIMD, which the sampler's 256-register build keeps resident together")
    ap.add_argument("--reads", type=int, default=None, help="reads per sample (default 20000; 300 with --mode ont)")
    ap.add_argument("--mode", choices=["short", "ont", "nipt"], default="short",
                    help="read model (ont: BASELINE configs[3]; nipt: configs[4], method = nipt with ff = 0.2)")
    ap.add_argument("--workers", type=int, default=None,
                    help="host threads per GPU (each with its own stream); default 3, 4 with --mspbwt (measured: DESIGN.md 5)")
    ap.add_argument("--rare-common", type=float, default=0.0, metavar="F",
                    help="impute_rare_common with F x nsnps rare SNPs: every Gibbs sample ends with a Gibbs call over all SNPs "
                         "(QUILT2; not the headline workload, no CPU baseline)")
    ap.add_argument("--precision", choices=["both", "fp64", "mixed"], default="fp64",
                    help="fp64: every kernel of the path computes in double, as the reference (the headline `value`); mixed: dosage "
                         "passes with fp32 state, fp64 emissions and sums (SURVEY 8(d)'s fp32 alpha checkpoint); both: the "
                         "K timed steps run at fp64 and are reported as `value`, then the same K steps run again with the mixed "
                         "dosage passes in a second timed region of their own, reported under `mixed_precision` (the default until round 4; "
                         "since the CPU baseline is measured on whole samples -- six minutes of wall -- the default run is the fp64 region alone)")
    ap.add_argument("--fp64-dosage", action="store_true", help="(older spelling of --precision fp64)")
    ap.add_argument("--host-share", type=int, default=1,
                    help="N > 1: confine this run to 1 / N of the host's logical CPUs and give it the host-thread budget of one rank of an "
                         "N-rank run -- measures, on one GPU, what a rank's share of the host costs (the 8-GPU node's host side)")
    ap.add_argument("--io-threads", type=int, default=0,
                    help="--bam: host threads of qa_impute_bam_range's loading and of its formatting (0 = the library's default: min(16, "
                         "hardware threads) per rank; more get in each other's and the imputation's way: csrc/bamrange.cpp); divided by "
                         "the number of ranks")
    ap.add_argument("--bam-lean", action="store_true",
                    help="--bam as the R fast path calls it: discard_sample_arrays (columns, labels and counts come back; a sample's "
                         "per-SNP arrays go back to the system once its column is formatted); the check against the timed region is then on "
                         "the last step's column TEXT")
    ap.add_argument("--bam", action="store_true",
                    help="the synthetic samples go through BAM files: written before the run, read back by the native loader "
                         "(qa_bam_load_sample_reads) outside the timed region; the loader's time per sample is reported")
    ap.add_argument("--no-alone", action="store_true",
                    help="skip the stand-alone kernel timings taken before the warm-up (`kernels_alone` in the output)")
    ap.add_argument("--pass-priority", type=int, default=0, choices=(0, 1),
                    help="full-panel calls on a highest-priority stream (qa_panel_set_pass_priority)")
    ap.add_argument("--exclusive", type=int, default=1, choices=(0, 1),
                    help="exclusive device phases (qa_panel_set_exclusive): every Gibbs launch / full-panel launch set gets the whole "
                         "device and the device-wide arena, the host threads take turns")
    ap.add_argument("--fuse-tails", type=int, default=1, choices=(0, 1),
                    help="1: when the stream drains, the host threads' last batches run their phasing rounds together "
                         "(driver.PhasingTail); 0: each thread runs its own, one after the other")
    ap.add_argument("--split-remainder", type=int, default=1, choices=(0, 1),
                    help="1: with --split alternate the launch sets left over when their number is not a multiple of the "
                         "thread count are cut into one part per thread; 0: they go to the threads in turn like the others")
    ap.add_argument("--gibbs-gate", type=float, default=0.0, metavar="SEC",
                    help="host threads wait up to SEC for each other before a Gibbs launch, so that the two launches overlap fully")
    ap.add_argument("--split", choices=["halves", "alternate"], default="alternate",
                    help="how the host threads share the work: whole batches in turn (default: a thread's Gibbs launch then carries "
                         "a whole batch's chains, one per SIMD at the defaults), or every batch cut into one part per thread")
    ap.add_argument("--mspbwt", action="store_true",
                    help="use_mspbwt = TRUE (mode M2): no full-panel pass; the small panel is re-selected from long matches of the "
                         "Gibbs call's haploid dosages against the panel (--mspbwt-search: the msPBWT neighbour scan by default)")
    ap.add_argument("--mspbwt-search", choices=["scan", "exhaustive"], default="scan",
                    help="--mspbwt: the query behind select_new_haps_mspbwt_v3 -- the msPBWT neighbour scan of the panel's indices "
                         "(the reference's semantics; host, csrc/mspbwt.cpp) or the exhaustive device search (csrc/match.hip)")
    ap.add_argument("--cu-partition", action="store_true",
                    help="confine each host thread's Gibbs launches to its own half of the CUs (measured slower, DESIGN.md 5)")
    ap.add_argument("--gate-trace", default=None, metavar="NPY",
                    help="write the device gate's hold trace of the (first) timed region (request, admit, kernels done, release, "
                         "SIMD slots, thread) to this .npy file")
    ap.add_argument("--pageable", action="store_true",
                    help="the driver's large transfer buffers as ordinary (pageable) memory instead of qa_host_alloc: the staged path "
                         "a caller that cannot allocate through the library gets (the R shim: R owns its vectors)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline", choices=["whole", "composed"], default="whole",
                    help="whole (default): cpu_baseline.value is MEASURED on whole samples, one per physical core, all cores at once "
                         "(minutes of wall; the composed figure is kept beside it); composed: only the composed figure (one Gibbs call "
                         "+ one thin + one dosage pass per core, priced up to a sample)")
    ap.add_argument("--cpu-baseline-budget", type=float, default=420.0, metavar="SEC",
                    help="wall-time bound of the whole-sample CPU baseline (measured: 355-389 s on 128 cores); workers still running then are "
                         "stopped, and when none has finished the composed figure stands alone")
    ap.add_argument("--no-scan-check", action="store_true",
                    help="--mspbwt: skip the comparison of the device search with the msPBWT neighbour scan (CPU, ~20 s)")
    ap.add_argument("--r2-vs-cpu", type=int, default=4, metavar="N",
                    help="also impute the first N samples of the last batch with the whole pipeline on the CPU oracle (its chains on "
                         "a thread pool: about a minute per sample on a many-core host; before any HIP context exists, rank 0 at "
                         "N = 1 only) and report the metric's `dosage r2 vs CPU ref`; 0 switches it off")
    ap.add_argument("--driver", choices=["native", "python"], default=None,
                    help="the per-sample loop between the native compute calls: native = qa_impute_samples (csrc/impute.cpp: C++ host "
                         "threads, one call per sample range; the default where it applies: diploid samples without --rare-common), "
                         "python = quilt_amd/driver.py + workers.py (Python threads over the same batched entry points)")
    ap.add_argument("--dotcall", type=int, default=None, metavar="W",
                    help="after the timed regions, measure the `.Call` shim's path as well (scripts/dotcall_path.py: per sample, per "
                         "Gibbs sample, one qa_gibbs_batch(n_chain = 1) and one qa_Rcpp_haploid_dosage_versus_refs per label, from W "
                         "worker processes sharing the GPU; one timed sample per worker) and report it as `dotcall_path`; default 0 "
                         "= off (16 is what the committed lines used)")
    ap.add_argument("--one-device", action="store_true",
                    help="rehearsal of the N-rank run on ONE GPU: every rank uses device 0 (set QA_ARENA_FRACTION so that N arenas fit) and "
                         "the ranks synchronise over gloo (RCCL refuses several ranks on one device).  Exercises N processes' panel "
                         "uploads, arenas, pinned buffers and gates at once; its samples/s is NOT a scaling figure")
    ap.add_argument("--digest", default=None, metavar="DIR",
                    help="write DIR/step_<global step>.sha256 for every timed step: a digest of the step's results (dosage, gp_t, phased "
                         "haplotypes, read labels, sample by sample), keyed by the GLOBAL step index rank * (warmup + steps) + step -- "
                         "an N-rank run and a 1-rank run over the same global steps must write the same files")
    ap.add_argument("--stub", action="store_true",
                    help="test hook: no device work at all (gloo rendezvous, a driver that returns zeros); value is 0")
    a = ap.parse_args()
    if a.fp64_dosage:
        a.precision = "fp64"
    maybe_spawn(a, sys.argv[1:])
    if a.reads is None:
        a.reads = 300 if a.mode == "ont" else 20000

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # the native calls' own host threads (per-chain tables, validation): the machine's cores divided between the ranks of this
    # node and the host threads of each rank, so that 8 ranks x 4 threads do not start 16 helpers each at the same moment
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if a.host_share > 1 and world == 1:
        # one rank's share of the host at N ranks: the first 1 / N of the logical CPUs, before any worker process or thread exists
        cpus = sorted(os.sched_getaffinity(0))
        os.sched_setaffinity(0, set(cpus[:max(1, len(cpus) // a.host_share)]))
        local_world = a.host_share
    if a.workers is None:
        a.workers = 4 if a.mspbwt else 3
    if a.pageable:
        os.environ["QUILT_AMD_PAGEABLE"] = "1"
    if a.fuse is None:
        # a launch set should carry enough Gibbs chains to fill the device whatever the batch: two chains per SIMD (2 048; the
        # sampler's 256-register build) for short reads -- 2 steps of 128 samples, 8 of 32 (configs[1]); ONT: short Gibbs
        # launches, NIPT: three labels, no 256-register build -- one chain per SIMD
        per_step = a.batch * 8
        # (impute_rare_common: the all-SNP Gibbs call's state is three times as long -- 1 024 chains per launch set)
        a.fuse = (max(1, min(16, round(2048 / per_step))) if (a.mode == "short" and a.rare_common <= 0)
                  else max(1, min(16, round(1024 / per_step))))
    a.fuse = max(1, a.fuse)
    native_ok = not (a.mspbwt and a.mspbwt_search != "scan") and not a.stub
    if a.driver is None:
        a.driver = "native" if native_ok and a.split == "alternate" and a.exclusive and not a.gibbs_gate and not a.cu_partition else "python"
    if a.driver == "native" and not native_ok:
        raise SystemExit("--driver native: msPBWT mode runs the neighbour scan (--mspbwt-search scan); not with --stub")
    os.environ.setdefault("QA_HOST_THREADS", str(max(2, (os.cpu_count() or 16) // max(1, local_world * a.workers))))
    params = dict(nGibbsSamples=7, n_seek_its=3, Ksubset=600, Knew=600, seed=1)
    full_chains = params["nGibbsSamples"] + 1
    ff = 0.2 if a.mode == "nipt" else 0.0

    # everything that forks happens before any HIP context exists in this process
    from quilt_amd.synth import make_synthetic_panel
    panel = make_synthetic_panel(K=a.K, nSNPs=a.nsnps, seed=4916)
    if a.mode == "nipt":
        params["method"] = "nipt"
    rc = None
    if a.rare_common > 0:
        from quilt_amd.synth import make_rare_common
        rc = make_rare_common(panel, 7, n_rare=int(a.rare_common * a.nsnps))
        params["impute_rare_common"] = True
    if a.mspbwt:
        params["use_mspbwt"] = True
        params["mspbwt_search"] = a.mspbwt_search
    cpu, keep = None, None
    # The CPU legs run on rank 0 while the host cores are otherwise idle and before any HIP context exists in this process
    # (they fork / start threads).  With several ranks the others wait for rank 0, so that the baseline is measured "in the
    # same run" at every N without the ranks' own sample generation competing for the cores (a flag file in /tmp keyed by the
    # launcher's pid and port: no process group -- and so no HIP context -- is needed yet)
    cpu_flag = None
    if world > 1:
        cpu_flag = f"/tmp/quilt_amd_bench_{os.getppid()}_{os.environ.get('MASTER_PORT', '0')}.cpu_done"
    if rank == 0 and not a.no_cpu_baseline and rc is None and not a.stub:
        cpu, keep = cpu_baseline(panel, a.reads, params, full_chains, ff=ff, mspbwt=a.mspbwt)
    if cpu_flag is not None:
        if rank == 0:
            open(cpu_flag, "w").close()
        else:
            t_wait = time.time()
            while not os.path.exists(cpu_flag) and time.time() - t_wait < 600:
                time.sleep(0.2)
            a.waited_for_rank0_s = time.time() - t_wait
    n_steps = a.warmup + a.steps
    seeds = [1000 + (rank * n_steps + st) * a.batch + i for st in range(n_steps) for i in range(a.batch)]
    bam_dir, bam_load_s = None, None
    a.bam = a.bam or a.bam_lean
    if a.bam:
        import tempfile
        if rc is not None:
            raise SystemExit("--bam is not combined with --rare-common")
        bam_dir = tempfile.mkdtemp(prefix="quilt_amd_bams_")
    flat = make_samples(panel, seeds, a.reads, min(32, max(1, physical_cores() // max(world, 1))), a.mode, rc, bam_dir)
    if a.bam:
        import torch  # noqa: F401  (before libquilt_amd.so is loaded: both must resolve the HIP runtime torch ships, see main())
        flat, bam_load_s = reload_from_bams(panel, flat, seeds, bam_dir)
        a.bam_load_s = bam_load_s
        a.bam_dir = bam_dir   # (kept for the I/O-inclusive leg after the timed region; removed there)
    samples = [flat[st * a.batch:(st + 1) * a.batch] for st in range(n_steps)]
    cpu_pipeline = None
    if (rank == 0 and world == 1 and not a.no_cpu_baseline and not a.stub and a.cpu_baseline == "whole" and
            (not a.mspbwt or a.mspbwt_search == "scan")):
        # the baseline proper: whole samples, one per physical core, all cores at once; the composed figure stays beside it
        cores = physical_cores()
        work, st = [], n_steps - 1
        while len(work) < cores and st >= 0:   # the last timed batch first (its first samples double as the r2 reference)
            work += [(smp, st * a.batch + i) for i, smp in enumerate(samples[st])]
            st -= 1
        whole, ref = cpu_baseline_whole(panel, params, work, cores, a.cpu_baseline_budget, min(a.r2_vs_cpu, len(samples[-1])), rc)
        if whole is not None:
            if cpu is not None:   # (no composed figure with impute_rare_common: its all-SNP calls are not in the composition)
                whole["composed"] = cpu
                whole["composed_over_whole"] = round(cpu["value"] / whole["value"], 3)
            cpu = whole
        if ref is not None:
            cpu_pipeline = ref
    if cpu_pipeline is None and rank == 0 and world == 1 and a.r2_vs_cpu > 0 and rc is None and not a.stub and not a.mspbwt:
        cpu_pipeline = cpu_pipeline_reference(a, panel, params, samples)
    import torch
    import torch.distributed as dist
    if a.one_device:
        local_rank_dev = 0
    else:
        local_rank_dev = local_rank
    if world > 1:
        if a.stub or a.one_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if not a.stub:
        torch.cuda.set_device(local_rank_dev)

    from quilt_amd.driver import DriverParams
    if a.stub:
        from quilt_amd.workers import StubWorkers
        native = None
        drv = StubWorkers(panel, DriverParams(**params))
    else:
        from quilt_amd import native
        from quilt_amd.workers import DeviceWorkers
        native.check(native.lib().qa_set_device(local_rank_dev))
        if a.driver == "native":
            from quilt_amd.workers import NativeWorkers
            drv = NativeWorkers(panel, DriverParams(**params), n_workers=a.workers, fp64_dosage=a.precision != "mixed",
                                exclusive=bool(a.exclusive), fuse_tails=bool(a.fuse_tails), rare_common=rc)
        else:
          drv = DeviceWorkers(panel, DriverParams(**params), n_workers=a.workers, rare_common=rc,
                            cu_partition=a.cu_partition, fp64_dosage=a.precision != "mixed", split=a.split, gibbs_gate=a.gibbs_gate,
                            pass_priority=bool(a.pass_priority), exclusive=bool(a.exclusive),
                            fuse_tails=bool(a.fuse_tails), split_remainder=bool(a.split_remainder))

    def barrier():
        if not a.stub:
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        if not a.stub:
            torch.cuda.synchronize()

    def stream(lo, hi):
        # a host thread takes `fuse` consecutive steps per launch set (their samples' global indices are consecutive)
        for st in range(lo, hi, a.fuse):
            top = min(st + a.fuse, hi)
            yield [smp for q in range(st, top) for smp in samples[q]], (rank * n_steps + st) * a.batch

    alone = None
    if native is not None and rank == 0 and not a.no_alone and hasattr(drv, "drivers"):
        alone = kernels_alone(native, drv, samples)

    def timed_region(n_warm):
        """n_warm untimed batches, then the K timed batches bracketed by barriers; returns what report() needs."""
        for _ in drv.run_stream(stream(0, n_warm)):
            pass
        if native is not None:
            native.lib().qa_profile_reset()
            native.gate_stats(local_rank_dev, reset=True)
            if a.gate_trace:
                native.gate_trace(local_rank_dev, on=True)
        drv.reset_timing()
        # the native loop's caller holds the range's reads in the C ABI's flat form (include/quilt_amd.h: the samples' reads back
        # to back) before the clock starts -- host buffers, still to cross PCIe; concatenating 2 560 Python objects' arrays is
        # this harness's business, not the path's (the R shim flattens R's lists in C)
        timed_input = drv.prepare(stream(a.warmup, n_steps)) if hasattr(drv, "prepare") else None
        barrier()
        t0 = time.perf_counter()
        t0_gate = time.monotonic() * 1e3   # (std::chrono::steady_clock on Linux: the gate trace's clock)
        last = None
        # the K timed steps are K whole batches: the driver pipelines consecutive batches (phasing rounds of one fused
        # with the main rounds of the next), and the pipeline is filled and drained inside the timed region
        n_seen = 0
        for res in drv.run_stream(timed_input if timed_input is not None else stream(a.warmup, n_steps)):
            last = res[-a.batch:]   # (the results of the last STEP: the tail of the last launch set)
            if a.digest:   # (launch sets come back in order, a.fuse steps each)
                import hashlib
                os.makedirs(a.digest, exist_ok=True)
                for q in range(0, len(res), a.batch):
                    h = hashlib.sha256()
                    for r in res[q:q + a.batch]:
                        for arr in (r.dosage, r.gp_t, r.phasing_haps, r.read_labels):
                            h.update(np.ascontiguousarray(arr).tobytes())
                    g_step = rank * n_steps + a.warmup + n_seen
                    open(os.path.join(a.digest, f"step_{g_step}.sha256"), "w").write(h.hexdigest() + "\n")
                    n_seen += 1
        barrier()
        elapsed = time.perf_counter() - t0
        per_rank = None
        if world > 1:
            # every rank's own clock around its K steps (the barrier on either side makes them nearly equal; what differs is a
            # rank that finished early and waited), its host-thread budget and how long it waited for rank 0's CPU legs
            mine = torch.tensor([elapsed, float(os.environ.get("QA_HOST_THREADS", "0")), float(getattr(a, "waited_for_rank0_s", 0.0))],
                                dtype=torch.float64, device="cpu" if (a.stub or a.one_device) else "cuda")
            allr = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allr, mine)
            per_rank = [[float(x) for x in r.cpu()] for r in allr]
            t = torch.tensor([elapsed], device="cpu" if (a.stub or a.one_device) else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        reg = dict(elapsed=elapsed, last=last, timing=dict(drv.timing), chains=getattr(drv, "n_gibbs_chain_calls", 0),
                   per_rank=per_rank)
        if native is not None:
            reg["prof"] = read_profile(native)
            reg["gate"] = native.gate_stats(local_rank_dev)
            if a.gate_trace and rank == 0:
                tr = native.gate_trace(local_rank_dev, on=False)
                if len(tr):
                    np.save(a.gate_trace, np.column_stack([tr[:, :4] - t0_gate, tr[:, 4:]]))
                a.gate_trace = None
        return reg

    main_reg = timed_region(a.warmup)
    mixed_reg = None
    if a.precision == "both" and native is not None:
        for d in drv.devs:   # the same K batches again with the mixed dosage passes, in a timed region of their own
            d.set_dosage_precision(32)
        mixed_reg = timed_region(min(a.warmup, 1))

    from quilt_amd import trace
    trace.dump()   # host span trace, only with QUILT_AMD_TRACE=<file>
    if rank == 0:
        out = report(a, panel, params, native, drv, samples, main_reg, world, rc, cpu, keep, ff, full_chains, alone,
                     fp64=a.precision != "mixed")
        if a.host_share > 1:
            out["host_share"] = {"of": a.host_share, "logical_cpus_used": len(os.sched_getaffinity(0)),
                                 "QA_HOST_THREADS": int(os.environ.get("QA_HOST_THREADS", "0")),
                                 "what": f"this whole run was confined to 1 / {a.host_share} of the host's logical CPUs (sched_setaffinity before anything "
                                         f"else started) with the host-thread budget a rank has when {a.host_share} ranks share the node: `value` is "
                                         f"what ONE rank of such a run achieves on its share of the host (the other {a.host_share - 1} ranks' memory "
                                         "traffic is absent)"}
        if a.one_device and world > 1:
            out["one_device_rehearsal"] = (f"{world} ranks on ONE GPU (device 0, gloo rendezvous, QA_ARENA_FRACTION="
                                           f"{os.environ.get('QA_ARENA_FRACTION', 'default')}): a rehearsal of the N-rank run's host side "
                                           "and of N processes sharing a device -- `value` is NOT a scaling figure")
        if mixed_reg is not None:
            mx = report(a, panel, params, native, drv, samples, mixed_reg, world, rc, None, None, ff, full_chains, None, fp64=False)
            out["mixed_precision"] = {k: mx[k] for k in ("value", "unit", "ms_per_step", "dtype", "roofline", "kernels", "host_seconds",
                                                          "device_phases", "dosage_r2_vs_truth_sample0") if k in mx}
            out["mixed_precision"]["what"] = ("the same K batches in a second timed region with the dosage passes at fp32 state (fp64 "
                                              "emissions and sums): SURVEY 8(d)'s fp32 alpha checkpoint, 10 instead of 18 bytes per cell")
            a_, b_ = main_reg["last"][0].dosage, mixed_reg["last"][0].dosage
            out["mixed_precision"]["dosage_vs_fp64_run_sample0"] = {"r2": float(np.corrcoef(a_, b_)[0, 1] ** 2),
                                                                    "max_abs_diff": float(np.abs(a_ - b_).max())}
        if a.dotcall is None:
            a.dotcall = 0   # (80 s; off by default since round 5 -- `--dotcall 16` measures it: profiles/r05_bench_line_dotcall.json)
        if native is not None and world == 1 and a.driver == "native" and hasattr(drv, "devs"):
            # one sample alone on the device (the quick-start's shape): the latency of the whole per-sample pipeline
            from quilt_amd.impute import impute_samples
            t_l = time.perf_counter()
            impute_samples(drv.devs[:1], samples[-1][:1], DriverParams(**params), sample_offset=10 ** 6, drcs=drv.drcs[:1])
            out["one_sample_latency_s"] = round(time.perf_counter() - t_l, 3)
        if getattr(a, "bam_dir", None) and native is not None and world == 1 and a.driver == "native" and hasattr(drv, "devs"):
            # --bam: the SAME K timed steps once more through qa_impute_bam_range (csrc/bamrange.cpp: what shim/quilt-amd.R's fast path
            # calls) -- BAM paths in, VCF columns and count arrays out, loading and formatting on host threads INSIDE the clock.
            # `value` stays the compute-only rate (inputs resident in host memory); this is the rate with the I/O either side.
            from quilt_amd.impute import impute_bam_range
            from quilt_amd.synth import synthetic_alleles
            ref, alt = synthetic_alleles(panel.nSNPs, 1)
            t_seeds = seeds[a.warmup * a.batch:n_steps * a.batch]
            files = [os.path.join(a.bam_dir, f"s{sd}.bam") for sd in t_seeds]
            t_io = time.perf_counter()
            r_io = impute_bam_range(drv.devs, files, "chr20", ref, alt, DriverParams(**params),
                                    sample_index=[(rank * n_steps + a.warmup) * a.batch + i for i in range(len(files))],
                                    ff=[0.2] * len(files) if a.mode == "nipt" else None, samples_per_launch_set=a.batch * a.fuse,
                                    fuse_tails=bool(a.fuse_tails), downsampleToCov=0, bqFilter=1, n_io_threads=max(1, a.io_threads // max(world, 1)) if a.io_threads else 0,
                                    copy_out=range(len(files) - a.batch, len(files)), discard_sample_arrays=bool(a.bam_lean))
            t_io = time.perf_counter() - t_io
            if a.bam_lean:
                from quilt_amd.io import make_per_sample_vcf_col
                same = all(r_io["columns"][len(files) - a.batch + i].tolist() ==
                           make_per_sample_vcf_col(main_reg["last"][i].gp_t, main_reg["last"][i].phasing_haps, True).tolist() for i in range(a.batch))
            else:
                same = all(np.array_equal(r_io["results"][len(files) - a.batch + i].dosage, main_reg["last"][i].dosage) for i in range(a.batch))
            sec = r_io["seconds"]
            out["from_bam_files"] = {
                "value": len(files) / t_io, "unit": "samples/sec", "samples": len(files), "wall_s": round(t_io, 3),
                "seconds": {k: round(v, 3) for k, v in sec.items()},
                "io_threads": (max(1, a.io_threads // max(world, 1)) if a.io_threads else "min(16, hardware threads)"),
                "share_of_wall": {"impute_with_loading_and_formatting_beside_it": round(sec["impute"] / t_io, 4),
                                  "format_and_counts_left_at_the_end": round(sec["format"] / t_io, 4),
                                  "outside_the_native_call": round(1 - sec["total"] / t_io, 4)},
                "last_file_loaded_at_s": round(sec["load"], 3),
                "compute_only_value": out["value"],
                "last_step_dosages_equal_the_timed_region": bool(same),
                "vcf_bytes_per_sample": int(np.mean([len(c.buf) for c in r_io["columns"] if c is not None])),
                "results_copied_into_python": a.batch, "discard_sample_arrays": bool(a.bam_lean),
                "what": "qa_impute_bam_range over the timed steps' BAM files, one native call, everything inside the clock: the files are "
                        "loaded on host threads in file order BESIDE the imputation (qa_impute_samples is handed each sample when its "
                        "launch set is taken; seconds.load = when the last file was in), the columns of finished launch sets are "
                        "formatted on host threads beside it too; seconds.format = the last sets' columns + the range's count arrays, "
                        "after the device work ends"}
            import shutil
            shutil.rmtree(a.bam_dir, ignore_errors=True)
        if a.dotcall > 0 and native is not None and world == 1:
            # the device-wide arena goes with the last handle: the workers are processes of their own and need the memory
            drv.close()
            import subprocess
            cmd = [sys.executable, os.path.join(ROOT, "scripts", "dotcall_path.py"), "--workers", str(a.dotcall), "--samples", "1",
                   "--K", str(a.K), "--nsnps", str(a.nsnps), "--reads", str(a.reads), "--device", str(local_rank)]
            if a.precision == "mixed":
                cmd.append("--mixed")
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
                out["dotcall_path"] = json.loads(r.stdout.strip().splitlines()[-1])
                out["dotcall_path"]["batched_over_dotcall"] = out["value"] / max(out["dotcall_path"]["value"], 1e-12)
            except Exception as e:   # noqa: BLE001 -- a reported leg, never the reason a bench line is lost
                out["dotcall_path"] = {"error": repr(e)[:300]}
        if cpu_pipeline is not None:
            out["dosage_r2_vs_cpu_pipeline"] = r2_vs_cpu_pipeline(cpu_pipeline, main_reg["last"])
            if out.get("cpu_baseline") and "composed" not in out["cpu_baseline"]:
                # the composed figure checked against WHOLE samples: the r2 leg runs n samples through the whole per-sample
                # pipeline on the CPU path (all chains, all rounds) on `threads` cores
                n_, sec, thr = cpu_pipeline["n"], cpu_pipeline["cpu_seconds"], cpu_pipeline["threads"]
                cores = out["cpu_baseline"]["cores"]
                out["cpu_baseline"]["whole_sample_check"] = {
                    "what": f"{n_} whole sample(s) through the entire per-sample pipeline on the CPU path (the dosage_r2_vs_cpu_pipeline "
                            f"leg), its chains on {thr} threads: seconds, and the samples/sec that rate would give on all {cores} cores "
                            "if it scaled with the cores (it has the device-free host logic of the driver in it, which the composed "
                            "figure leaves out)",
                    "samples": n_, "seconds": sec, "threads": thr,
                    "samples_per_sec_scaled_to_cores": round(n_ / max(sec, 1e-9) * cores / thr, 3)}
            if mixed_reg is not None:
                out["mixed_precision"]["dosage_r2_vs_cpu_pipeline"] = r2_vs_cpu_pipeline(cpu_pipeline, mixed_reg["last"])
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        if rank == 0 and cpu_flag is not None and os.path.exists(cpu_flag):
            os.remove(cpu_flag)
        dist.destroy_process_group()


def read_profile(native):
    """The library's per-kernel accumulators since the last reset (HIP events on the launch streams)."""
    import ctypes as C
    L = native.lib()
    L.qa_profile_name.restype = C.c_char_p
    prof = []
    for k in range(L.qa_profile_count()):
        ms, n, b, busy, units, serial = C.c_double(), C.c_int64(), C.c_double(), C.c_double(), C.c_double(), C.c_double()
        wgs = C.c_double()
        L.qa_profile_get(k, C.byref(ms), C.byref(n), C.byref(b))
        L.qa_profile_get_busy(k, C.byref(busy))
        L.qa_profile_get_work(k, C.byref(units), C.byref(serial))
        L.qa_profile_get_workgroups(k, C.byref(wgs))
        if n.value:
            prof.append(dict(kernel=L.qa_profile_name(k).decode(), ms=ms.value, launches=n.value, alg_bytes=b.value,
                             busy_ms=busy.value, units=units.value, serial=serial.value, workgroups=wgs.value))
    return prof


def kernels_alone(native, drv, samples):
    """Every kernel of the path with the device to itself: one host thread takes 512 chains through a first and a last round
    of the driver while the others wait (before the warm-up, outside the timed region).  Per kernel the mean launch time and the algorithmic
    GB/s of a launch -- the kernels' own rates, next to the in-run ones that include the waiting for compute units other
    threads' launches hold."""
    from quilt_amd.driver import ChainState, chain_rng
    native.lib().qa_profile_reset()
    d = drv.drivers[0]
    P = d.params
    half = samples[0][:max(1, len(samples[0]) // 2)]
    # 64 samples x (nGibbsSamples + 1) chains = 512: one first round (Gibbs launch + ranking passes, selection) and one last
    # round (Gibbs launch + dosage passes), the two kinds of round a batch goes through
    chains = [ChainState(smp, i, c, chain_rng(P.seed, i, c)) for i, smp in enumerate(half) for c in range(1, P.nGibbsSamples + 2)]
    d._round(chains, 1)
    d._round(chains, P.n_seek_its)
    d._round_dosages, d._round_dosage_chains = None, []
    d.timing = {k: 0.0 for k in d.timing}
    d.n_gibbs_chain_calls = 0
    d.n_device_selections = 0
    out = []
    for p in read_profile(native):
        per = p["alg_bytes"] / max(p["launches"], 1)
        avg = p["ms"] / max(p["launches"], 1)
        out.append({"kernel": p["kernel"], "launches": p["launches"], "avg_launch_ms": round(avg, 3),
                    "GBps_per_launch": round(per / 1e6 / avg, 1) if avg > 0 else 0.0,
                    "frac_of_hbm_peak": round(per / 1e6 / avg / HBM_PEAK_GBS, 3) if avg > 0 else 0.0})
    native.lib().qa_profile_reset()
    return out


def report(a, panel, params, native, drv, samples, reg, world, rc, cpu, keep, ff, full_chains, alone=None, fp64=True):
    import ctypes as C
    last, elapsed = reg["last"], reg["elapsed"]
    value = 0.0 if a.stub else a.batch * world * a.steps / elapsed
    out = {
        "metric": "samples/sec on 2Mb region, K=50k haps, 1x coverage; dosage r2 vs CPU ref",
        "value": value, "unit": "samples/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64" if fp64 else "f64 (Gibbs sampler, ranking passes) / f32 state with f64 emissions and sums (dosage passes)",
        "data": "stub (no device work)" if a.stub else
                ("synthetic, through BAM files read by the native loader before the timed region" if a.bam else "synthetic"),
        "config": {"workload": f"{workload_label(a.mode, a.K, a.batch, a.mspbwt, rc is not None)}: {a.batch} synthetic 1x {a.mode}-read samples per GPU per step, "
                               f"{a.nsnps} SNPs ({panel.nGrids} grids, 2 Mb + buffers), K={a.K} haplotypes, {a.reads} reads/sample, "
                               "QUILT defaults (nGibbsSamples=7, n_seek_its=3, Ksubset=600), "
                               + (("use_mspbwt=TRUE (mode M2: no full-panel pass; the next small panel comes from the msPBWT neighbour scan of the "
                                   "panel's indices, host C++, csrc/mspbwt.cpp)" if a.mspbwt_search == "scan" else
                                   "use_mspbwt=TRUE (mode M2: no full-panel pass; the next small panel comes from the exhaustive device "
                                   "search of csrc/match.hip -- this library's own definition, not the reference's query)") if a.mspbwt
                                  else "use_mspbwt=FALSE")
                               + (f", impute_rare_common=TRUE with {rc.nSNPs_all} SNPs in all ({rc.nGrids_all} grids)" if rc is not None else ""),
                   "mode": a.mode, "K": a.K, "nSNPs": a.nsnps, "samples_per_step_per_gpu": a.batch, "steps_per_launch_set": a.fuse,
                   "inputs": "host buffers cross PCIe inside the timed region (reads per call, labels, seeds; dosages and top "
                             "lists back): value is the PCIe-inclusive rate"
                             + ("; the range's reads are handed over in the C ABI's flat form (the samples' reads back to back, "
                                "include/quilt_amd.h), flattened before the clock starts" if a.driver == "native" else "")
                             + ("; --pageable: the dosage rounds come back into pageable memory through the library's staging copy"
                                if a.pageable else ""),
                   "driver": ("native: qa_impute_samples (csrc/impute.cpp), one call per sample range, C++ host threads" if a.driver == "native"
                              else "python: quilt_amd/driver.py + workers.py over the batched entry points"),
                   "parallelism": f"samples sharded over {world} GPU(s), no collective; {a.workers} host threads per "
                                  "GPU (" + ("whole batches in turn" if a.split == "alternate" else "every batch cut into one part per thread") +
                                  "), consecutive batches pipelined"
                                  + (f"; a host thread's launch set carries {a.fuse} steps ({a.fuse * a.batch} samples: "
                                     f"{a.fuse * a.batch * (params['nGibbsSamples'] + 1)} Gibbs chains with the phasing chains "
                                     "of the set before it)" if a.fuse > 1 else "")
                                  + ("; device phases: full-panel launch sets exclusive, Gibbs launches that fit run together" if a.exclusive else "")
                                  + ("; when the stream drains the threads' last batches run their phasing rounds in one launch per round"
                                     if a.fuse_tails and a.workers > 1 else "")
                                  + ("; launch sets left over by the thread count go whole to the first threads"
                                     if a.driver == "native" and a.workers > 1 else
                                     "; launch sets left over by the thread count are cut into one part per thread"
                                     if a.split_remainder and a.split == "alternate" and a.workers > 1 else "")},
    }
    if reg.get("per_rank"):
        el = [r[0] for r in reg["per_rank"]]
        rate = [0.0 if a.stub else a.batch * a.steps / max(e, 1e-12) for e in el]
        out["ranks"] = {"n": world, "elapsed_s": [round(e, 4) for e in el],
                        "samples_per_sec_min": min(rate), "samples_per_sec_max": max(rate),
                        "QA_HOST_THREADS": [int(r[1]) for r in reg["per_rank"]],
                        "host_threads_per_rank": a.workers, "logical_cpus": os.cpu_count(),
                        "waited_for_rank0_cpu_legs_s": [round(r[2], 2) for r in reg["per_rank"]],
                        "what": "per rank: its own clock around the K timed steps (value uses the MAX), the host threads one native "
                                "call may start (the node's cores divided between the ranks and their host threads), and how long "
                                "the rank waited at the flag file for rank 0's CPU baseline legs before creating its HIP context"}
    if getattr(a, "bam_load_s", None) is not None:
        out["bam_load_ms_per_sample"] = 1e3 * a.bam_load_s
    if a.stub:
        out["stub"] = True
        out["roofline"] = None
        out["cpu_baseline"] = None
        return out
    prof = reg["prof"]
    dom = max(prof, key=lambda p: p["ms"])
    # per-launch figure (the recipe): algorithmic bytes of one launch / its mean duration.  Launches of the host threads
    # overlap on the device, so the aggregate rate (bytes / time during which at least one launch of the kernel ran) is given
    # beside it; it may exceed what HBM can stream because the bytes are algorithmic, not traffic.
    per_launch = dom["alg_bytes"] / max(dom["launches"], 1)
    avg_ms = dom["ms"] / max(dom["launches"], 1)
    ach = per_launch / 1e9 / (avg_ms / 1e3) if avg_ms > 0 else 0.0
    agg = dom["alg_bytes"] / 1e9 / (dom["busy_ms"] / 1e3) if dom["busy_ms"] > 0 else 0.0
    traffic, traffic_source = None, None
    pmc_path = pmc_file_for(a.mode, a.K, a.batch, a.mspbwt, rc is not None)
    try:   # HBM bytes per launch of the dominant kernel from the committed PMC passes of this same workload
        pmc = json.load(open(os.path.join(ROOT, pmc_path)))
        # per template instantiation where the summary has it (the sampler's builds are different code)
        inst = pmc.get("instantiations", {})
        key = dom["kernel"] if dom["kernel"] in inst else ("k_gibbs<10, 1, false>" if dom["kernel"] == "k_gibbs" else None)
        pk = inst[key] if key in inst else pmc["kernels"][dom["kernel"].split("<")[0]]
        if dom["kernel"] == "k_gibbs3" and all(k in pmc["kernels"] for k in ("k_gibbs3", "k_ematread")) and dom.get("workgroups"):
            # NIPT: the library's k_gibbs3 entry is a whole CALL -- the sampler's segments (cut at the block-Gibbs iterations), the
            # switch-rate kernel and the block kernel between them, the host's block definition -- while rocprof sees every launch:
            # bytes per chain and call = the three kernels' bytes / (chains x calls), calls = k_ematread's launches (one per call)
            kk = pmc["kernels"]
            seg = kk["k_gibbs3"]["launches"] / max(kk["k_ematread"]["launches"], 1)
            tot = sum(kk[k]["fetch_bytes"] + kk[k]["write_bytes"] for k in ("k_gibbs3", "k_block3", "k_block_rate3") if k in kk)
            per_chain_call = tot / (kk["k_gibbs3"]["workgroups"] / seg)
            traffic = per_chain_call * dom["workgroups"] / max(dom["launches"], 1)
            how = (f"k_gibbs3 + k_block3 + k_block_rate3 bytes per chain and call ({per_chain_call / 1e9:.2f} GB: {seg:.0f} sampler segments "
                   f"with the block passes between them) x this run's {dom['workgroups'] / max(dom['launches'], 1):.0f} chains per call")
        elif pk.get("hbm_bytes_per_workgroup") and dom.get("workgroups") and dom["kernel"].startswith("k_gibbs"):
            # launches come in sizes: the counters' bytes per workgroup (= per chain) times this run's chains per launch
            traffic = pk["hbm_bytes_per_workgroup"] * dom["workgroups"] / max(dom["launches"], 1)
            how = (f"bytes per workgroup (one per chain) of {key or dom['kernel']} x this run's "
                   f"{dom['workgroups'] / max(dom['launches'], 1):.0f} chains per launch")
        else:
            traffic = pk["hbm_bytes_per_launch"]
            how = "bytes per launch"
        traffic_source = (f"{pmc_path}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes; factors calibrated, "
                          f"profiles/r05_fetch_calibration.json) of `{pmc['command']}`, {how}; not measured in this run")
    except (OSError, KeyError, ValueError, TypeError):
        pass
    ratio = (traffic / per_launch) if traffic else None
    # What the fraction is priced on.  The algorithmic bytes of SURVEY 8(d) are a contract, not a measurement: where the
    # counters show that a kernel moves FEWER bytes than that (the sampler carries alpha in registers and reads compact
    # emissions: ratio ~0.6) the algorithmic rate can exceed what HBM carries -- even the chip's peak -- and says nothing about
    # the memory system.  `achieved` / `frac` are then the counters' bytes over the launch time (what HBM actually carried);
    # the algorithmic figure stays beside it, flagged.
    alg = {"bytes_per_launch": per_launch, "achieved": ach, "frac": ach / HBM_PEAK_GBS,
           "note": "SURVEY 8(d) contract bytes / launch time; exceeds the traffic-based figure where bytes stay on chip"}
    if ratio is not None and ratio < 1.0:
        ach_t = traffic / 1e9 / (avg_ms / 1e3) if avg_ms > 0 else 0.0
        alg["flag"] = "not_all_moved: counter traffic / algorithmic bytes = %.2f" % ratio
        priced = "hbm traffic (PMC counters, scaled to this run's launches)"
    else:
        ach_t = ach
        priced = "algorithmic bytes" + ("" if traffic else " (no counter summary of this workload is committed: an upper bound of what HBM carried -- the "
                                        "sampler keeps alpha in registers and reads compact emissions)")
    issue = None
    try:   # the instruction-issue side of the same kernel build, from the committed SQ counter passes
        pi = json.load(open(os.path.join(ROOT, PMC_INSTS_FILE)))["per_launch"]
        ki = pi.get(dom["kernel"]) or pi.get("k_gibbs<10, 1, false>" if dom["kernel"] == "k_gibbs" else dom["kernel"])
        if ki and ki.get("SQ_WAVE_CYCLES") and ki.get("SQ_WAVES"):
            waves_per_simd = 2 if dom["kernel"] == "k_gibbs<10, 1, true>" else 1
            wc = ki["SQ_WAVE_CYCLES"]
            visits = (dom["units"] / dom["workgroups"]) if dom.get("workgroups") else None   # read visits + grid steps per chain
            issue = {"what": "SQ counters per launch (quad-cycles), summed over the launch's waves; a SIMD holds waves_per_simd of them",
                     "source": PMC_INSTS_FILE, "waves_per_simd": waves_per_simd,
                     "valu_busy_frac_of_simd_cycles": round(min(1.0, waves_per_simd * ki.get("SQ_ACTIVE_INST_VALU", 0) / wc), 3),
                     "any_inst_busy_frac_of_simd_cycles": round(min(1.0, waves_per_simd * ki.get("SQ_ACTIVE_INST_ANY", 0) / wc), 3),
                     "wave_wait_frac": round(ki.get("SQ_WAIT_ANY", 0) / wc, 3),
                     "wave_issue_stall_frac": round(ki.get("SQ_WAIT_INST_ANY", 0) / wc, 3),
                     "salu_per_valu": round(ki.get("SQ_INSTS_SALU", 0) / max(ki.get("SQ_INSTS_VALU", 1), 1), 3)}
            if visits:
                per_wave = lambda c: ki.get(c, 0) / ki["SQ_WAVES"] / visits
                issue["insts_per_read_visit_or_grid_step"] = {k.replace("SQ_INSTS_", "").lower(): round(per_wave(k), 1)
                                                             for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM")}
    except (OSError, KeyError, ValueError, ZeroDivisionError):
        pass
    hbm_frac = ach_t / HBM_PEAK_GBS
    bound = "hbm"
    if issue is not None and issue["any_inst_busy_frac_of_simd_cycles"] > hbm_frac and ratio is not None and ratio < 1.0:
        bound = "issue"
    elif ratio is not None and ratio < 0.5:
        bound = "latency"
    roof = {"bound": bound, "priced_against": priced, "kernel": dom["kernel"], "achieved": ach_t, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": hbm_frac, "traffic": traffic, "traffic_source": traffic_source,
            "traffic_over_algorithmic": ratio, "algorithmic": alg, "issue": issue,
            "avg_launch_ms": avg_ms, "launches": dom["launches"], "alg_bytes_per_launch": per_launch,
            "aggregate": {"achieved": agg, "frac": agg / HBM_PEAK_GBS, "busy_ms": dom["busy_ms"],
                          "note": "all overlapping launches of the kernel together: ALGORITHMIC bytes / time with >= 1 launch running"}}
    if traffic and dom["busy_ms"] > 0:   # what HBM actually carries while >= 1 launch of the kernel runs (counter bytes)
        hbm = traffic * dom["launches"] / 1e9 / (dom["busy_ms"] / 1e3)
        roof["aggregate"]["hbm_traffic_GBps"] = hbm
        roof["aggregate"]["hbm_traffic_frac_of_peak"] = hbm / HBM_PEAK_GBS
    if dom["kernel"].startswith("k_gibbs"):
        roof["limited_by"] = ("a chain is a serial string of dependent fp64 instructions per read (sums over the small panel, a wave "
                              "reduction, a reciprocal, the draw); two chains per SIMD (the 256-register build) fill each other's "
                              "waits; `issue` says how busy the SIMDs' issue slots are, `frac` how much of the HBM peak the launch's "
                              "counter traffic amounts to -- the larger of the two names the bound")
    if dom["serial"] > 0:   # the serial chain's step time and the rate of read visits
        roof["us_per_grid_step"] = 1e3 * dom["ms"] / dom["serial"]
        roof["read_visits_and_grid_steps_per_s"] = dom["units"] / (dom["busy_ms"] / 1e3) if dom["busy_ms"] > 0 else None
    out["roofline"] = roof
    out["kernels"] = [{"kernel": p["kernel"], "ms": round(p["ms"], 2), "busy_ms": round(p["busy_ms"], 2), "launches": p["launches"],
                       "avg_launch_ms": round(p["ms"] / max(p["launches"], 1), 3),
                       "GBps_per_launch": round(p["alg_bytes"] / 1e6 / p["ms"], 1) if p["ms"] > 0 else 0.0} for p in prof]
    if alone is not None:
        out["kernels_alone"] = {"what": "one host thread, 512 chains through a first and a last round of the driver with the device "
                                        "to itself, before the warm-up (outside the timed region): the kernels' own launch times, without the waiting "
                                        "for compute units that the in-run figures of `kernels` include",
                                "kernels": alone}
        da = [k for k in alone if k["kernel"] == dom["kernel"]]
        if da:
            roof["alone"] = {"avg_launch_ms": da[0]["avg_launch_ms"], "achieved": da[0]["GBps_per_launch"],
                             "frac": da[0]["frac_of_hbm_peak"]}
    out["host_seconds"] = {k: round(v, 3) for k, v in reg["timing"].items()}
    gib = [p for p in prof if p["kernel"].startswith("k_gibbs")]
    if gib and reg["chains"]:
        out["gibbs_chains_per_launch"] = round(reg["chains"] / max(sum(p["launches"] for p in gib), 1), 1)
    if a.exclusive and reg.get("gate"):
        gs = reg["gate"]
        g_ms = max(gs["held_ms"] - gs["exclusive_ms"], 1e-9)
        out["device_phases"] = {
            "what": "qa_gate_stats over the timed region: time with a launch set on the device, split into full-panel phases "
                    "(exclusive) and Gibbs phases; SIMD slots (of 1 024) the Gibbs launches held, averaged over the Gibbs phases",
            "busy_frac": round(gs["held_ms"] / 1e3 / elapsed, 3), "fullpass_frac": round(gs["exclusive_ms"] / 1e3 / elapsed, 3),
            "gibbs_frac": round(g_ms / 1e3 / elapsed, 3), "gibbs_mean_simd_slots": round(gs["gibbs_slot_ms"] / g_ms, 1),
            "mean_simd_slots_over_the_region": round(gs["gibbs_slot_ms"] / 1e3 / elapsed, 1),
            "queued_s": round(gs["queued_ms"] / 1e3, 2), "holds": gs["holds"]}
    truth = (samples[-1][0].all_snp if rc is not None else samples[-1][0]).truth_haps[:2].sum(axis=0)
    out["dosage_r2_vs_truth_sample0"] = float(np.corrcoef(last[0].dosage, truth)[0, 1] ** 2)
    out["cpu_baseline"] = cpu
    if a.mspbwt and fp64 and native is not None and rc is None and not getattr(a, "no_scan_check", False):
        out["mspbwt_search_vs_neighbour_scan"] = search_vs_scan(drv.devs[0], panel, params, last[0])
    if keep is not None:
        dev = drv.devs[0]
        out["parity_vs_cpu_path"] = parity_vs_cpu(dev, panel, a.reads, params, ff, keep)
    return out


def search_vs_scan(dev, panel, params, res):
    """use_mspbwt = TRUE: the product's query (``mspbwt_search``: the msPBWT neighbour scan of csrc/mspbwt.cpp by default, or the
    exhaustive device search of csrc/match.hip) against the neighbour scan restated test-side (tests/mspbwt_scan.py, numpy; the
    mspbwt package is not in the reference tree, so parity with the package itself is unpinned), on the two phased haplotypes of
    one result of the last batch; the exhaustive device search's agreement is reported beside it either way."""
    from quilt_amd.driver import DriverParams
    from quilt_amd.mspbwt import find_good_matches, int_contract_rows, match_tables_as_lists, panel_mspbwt_index
    from tests.mspbwt_scan import find_good_matches_scan, selection_agreement
    P = DriverParams(**params)
    Zs = int_contract_rows(np.ascontiguousarray(res.phasing_haps.T[:2], dtype=np.float64))
    n_max = P.mspbwt_max_matches or 50 * P.mspbwtL
    t0 = time.perf_counter()
    exh = match_tables_as_lists(*find_good_matches(dev, Zs, P.mspbwt_nindices, P.mspbwtM, n_max))
    t1 = time.perf_counter()
    idx = panel_mspbwt_index(panel, P.mspbwt_nindices)
    t1b = time.perf_counter()
    nat = idx.find_good_matches(Zs, P.mspbwtL, P.mspbwtM)
    t2 = time.perf_counter()
    scan = find_good_matches_scan(panel, Zs, P.mspbwt_nindices, P.mspbwtL, P.mspbwtM)
    t3 = time.perf_counter()
    rows_equal = all(np.array_equal(nat[q][i], scan[q][i]) for q in range(len(Zs)) for i in range(P.mspbwt_nindices))
    got = nat if P.mspbwt_search == "scan" else exh
    agree = selection_agreement(scan, got, P.Knew, panel.K, panel.nGrids)
    other = selection_agreement(scan, exh, P.Knew, panel.K, panel.nGrids)
    return dict(what="the next small panel (select_new_haps_mspbwt_v3, Knew haplotypes) chosen from the product's query "
                     f"(mspbwt_search = {P.mspbwt_search}) and from the msPBWT neighbour scan restated test-side (mspbwtL up and down "
                     "per grid, tests/mspbwt_scan.py) for the two phased haplotypes of sample 0 of the last batch: `selected` = "
                     "share chosen by both; `longest` = share of the scan's Knew longest-matching haplotypes the product's query "
                     "reports at all; `length` = the same weighted by match length; `rows_identical`: the native scan's "
                     "(haplotype, start, length) rows equal the restatement's.  Parity with the mspbwt package itself is unpinned "
                     "(not in the reference tree)",
                search=P.mspbwt_search, rows_identical=bool(rows_equal),
                **{k: (round(v, 4) if isinstance(v, float) else v) for k, v in agree.items()},
                exhaustive_device_search={k: (round(v, 4) if isinstance(v, float) else v) for k, v in other.items()},
                device_search_ms=round((t1 - t0) * 1e3, 1), native_scan_ms=round((t2 - t1b) * 1e3, 2),
                index_bytes=idx.bytes, restated_scan_cpu_s=round(t3 - t2, 1))


def cpu_pipeline_reference(a, panel, params, samples):
    """The metric's "dosage r2 vs CPU ref": the first N samples of the LAST timed batch through the whole per-sample pipeline
    on the CPU oracle -- the same driver, the same seeds (every chain's stream is keyed by the global sample index), its
    chains on a thread pool.  Runs before any HIP context exists; compared with the GPU run's results afterwards."""
    from quilt_amd.driver import Driver, DriverParams
    from tests.oracle_backend import OracleBackend
    n_steps = a.warmup + a.steps
    n = min(a.r2_vs_cpu, len(samples[-1]))
    t0 = time.perf_counter()
    n_thr = min(max(32, 8 * n), physical_cores())   # (n samples x 7 chains advance in lock-step: a thread per chain)
    ref = Driver(panel, OracleBackend(panel, n_threads=n_thr), DriverParams(**params)).run(samples[-1][:n],
                                                                                            sample_offset=(n_steps - 1) * a.batch)
    return dict(ref=ref, n=n, cpu_seconds=round(time.perf_counter() - t0, 1), threads=n_thr)


def r2_vs_cpu_pipeline(cpu_pipeline, last):
    ref, n = cpu_pipeline["ref"], cpu_pipeline["n"]
    r2 = [float(np.corrcoef(last[i].dosage, ref[i].dosage)[0, 1] ** 2) for i in range(n)]
    return dict(what="whole pipeline, GPU vs the same driver on the CPU oracle (fp64), same seeds: the metric's `dosage r2 vs CPU ref`",
                samples=n, r2=r2, max_abs_diff=[float(np.abs(last[i].dosage - ref[i].dosage).max()) for i in range(n)],
                labels_identical=[bool(np.array_equal(last[i].read_labels, ref[i].read_labels)) for i in range(n)],
                cpu_seconds=cpu_pipeline["cpu_seconds"], cpu_threads=cpu_pipeline["threads"])


if __name__ == "__main__":
    main()

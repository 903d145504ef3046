#!/usr/bin/env python
"""bench.py -- samples/sec of the QUILT per-sample hot path on MI355X ("Measurement").

One "step" = one batch of synthetic 1x samples taken through the whole per-sample driver (7 + 1 Gibbs
chains x 3 rounds of [small-panel Gibbs -> full-panel forward/backward per read label -> haplotype
re-selection], reference defaults) against a synthetic K-haplotype panel.  Samples are independent:
with N GPUs every rank imputes its own batch (weak scaling), no collective on the data path.

Prints ONE JSON line (rank 0) with the contract fields plus `roofline` (dominant kernel, HIP-event
timed inside the library on its launch stream) and `cpu_baseline` (the fp64 C oracle pipeline, one
sample per host core, on a bounded share of the same workload; N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
_CPU_PANEL = None
KERNEL_NAMES = ["k_emat", "k_fwd", "k_bwd", "k_dosage+k_topk", "k_ematread", "k_gibbs", "k_happrobs", "k_fwd64", "k_bwd64"]


def _cpu_worker(args):
    """One host core: the three native calls of the per-sample loop on the fp64 oracle, once each, for
    this core's own synthetic sample (so that all cores contend for memory as forked R workers do)."""
    n_reads, i, Ksubset = args
    from oracle import oracle as O
    from quilt_amd.driver import thinned_grid_columns
    from quilt_amd.synth import make_synthetic_sample
    panel = _CPU_PANEL   # built once in the parent, shared copy-on-write by the forked workers
    s = make_synthetic_sample(panel, seed=1000 + i, n_reads=n_reads)
    rng = np.random.default_rng(i)
    which = np.sort(rng.choice(panel.K, min(Ksubset, panel.K), replace=False)).astype(np.int32) + 1
    H0 = rng.integers(1, 3, size=s.nReads).astype(np.int32)
    ru, rs = rng.random(s.nReads * 21), rng.random(3 * (panel.nGrids - 1))
    cols = thinned_grid_columns(panel.nGrids, 0.1)
    t0 = time.perf_counter()
    g = O.forwardBackwardGibbsNIPT(panel, s, which, H0, ru, 0, rs)
    t1 = time.perf_counter()
    per_base = np.repeat(g["H"], np.diff(s.read_ptr))
    sel = (per_base == 1) & (s.bq != 0)
    gl = O.make_gl_from_u_bq(s.u[sel], s.bq[sel], panel.nSNPs)
    t2 = time.perf_counter()
    O.haploid_dosage_versus_refs(panel, gl, cols, return_dosage=False, get_best_haps_from_thinned_sites=True)
    t3 = time.perf_counter()
    O.haploid_dosage_versus_refs(panel, gl, cols, return_dosage=True, get_best_haps_from_thinned_sites=True)
    t4 = time.perf_counter()
    return t1 - t0, t3 - t2, t4 - t3


def cpu_baseline(K, T, n_reads, params, full_chains):
    """CPU baseline on a bounded sample: every host core (one synthetic sample each, mirroring the reference's
    mclapply sharding, quilt.R:691-692) runs ONE small-panel Gibbs call, ONE thin full-panel pass and ONE dosage
    full-panel pass on the fp64 oracle; a sample costs chains x n_seek_its Gibbs calls, and per Gibbs call two
    full-panel passes of which 1 in n_seek_its computes dosages.  The R interpreter
    overhead of the real driver is not included, so this baseline is faster than the reference."""
    import multiprocessing as mp
    global _CPU_PANEL
    from oracle import oracle as O
    from quilt_amd.synth import make_synthetic_panel
    O.lib()
    _CPU_PANEL = make_synthetic_panel(K=K, nSNPs=T, seed=4916)
    cores = os.cpu_count() or 1
    ctx = mp.get_context("fork")
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        times = np.array(pool.map(_cpu_worker, [(n_reads, i, params["Ksubset"]) for i in range(cores)]))
    wall = time.perf_counter() - t0
    tg, tt, td = times.max(axis=0)            # slowest core, as a fork-join over cores would see
    n_calls = full_chains * params["n_seek_its"]
    per_sample = n_calls * tg + 2 * (n_calls - full_chains) * tt + 2 * full_chains * td
    return dict(value=float(cores / per_sample), unit="samples/sec", cores=cores, kind="port",
                sample=f"per core: 1 Gibbs call ({tg:.2f} s), 1 thin pass ({tt:.2f} s), 1 dosage pass ({td:.2f} s) of one "
                       f"synthetic sample, all {cores} cores concurrently; per-sample cost composed as {n_calls} Gibbs calls + "
                       f"{2 * (n_calls - full_chains)} thin + {2 * full_chains} dosage passes = {per_sample:.1f} s/core "
                       f"({wall:.0f} s of wall time)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--K", type=int, default=50000)
    ap.add_argument("--nsnps", type=int, default=64000)
    ap.add_argument("--batch", type=int, default=128, help="samples per step per GPU")
    ap.add_argument("--reads", type=int, default=20000)
    ap.add_argument("--workers", type=int, default=2, help="host threads per GPU (each with its own stream and arena)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    params = dict(nGibbsSamples=7, n_seek_its=3, Ksubset=600, Knew=600, seed=1)
    full_chains = params["nGibbsSamples"] + 1

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(a.K, a.nsnps, a.reads, params, full_chains)   # before any HIP context exists (fork)

    import torch
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)

    from quilt_amd import native
    from quilt_amd.driver import DriverParams
    from quilt_amd.synth import make_synthetic_panel, make_synthetic_sample
    from quilt_amd.workers import DeviceWorkers
    native.check(native.lib().qa_set_device(local_rank))
    panel = make_synthetic_panel(K=a.K, nSNPs=a.nsnps, seed=4916)
    n_steps = a.warmup + a.steps
    samples = [[make_synthetic_sample(panel, seed=1000 + (rank * n_steps + st) * a.batch + i, n_reads=a.reads)
                for i in range(a.batch)] for st in range(n_steps)]
    drv = DeviceWorkers(panel, DriverParams(**params), n_workers=a.workers)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def stream(lo, hi):
        return ((samples[st], (rank * n_steps + st) * a.batch) for st in range(lo, hi))

    for _ in drv.run_stream(stream(0, a.warmup)):
        pass
    native.lib().qa_profile_reset()
    drv.reset_timing()
    barrier()
    t0 = time.perf_counter()
    last = None
    # the K timed steps are K whole batches: the driver pipelines consecutive batches (phasing rounds of one fused
    # with the main rounds of the next), and the pipeline is filled and drained inside the timed region
    for res in drv.run_stream(stream(a.warmup, n_steps)):
        last = res
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        import ctypes as C
        prof = []
        for k in range(len(KERNEL_NAMES)):
            ms, n, b = C.c_double(), C.c_int64(), C.c_double()
            native.lib().qa_profile_get(k, C.byref(ms), C.byref(n), C.byref(b))
            busy = C.c_double()
            native.lib().qa_profile_get_busy(k, C.byref(busy))
            prof.append(dict(kernel=KERNEL_NAMES[k], ms=ms.value, launches=n.value, alg_bytes=b.value, busy_ms=busy.value))
        dom = max(prof, key=lambda p: p["ms"])
        # launches of the two host threads overlap on the device: the kernel's rate is its bytes over the time during
        # which at least one of its launches ran (union of the HIP-event intervals); avg_launch_ms is the per-launch mean
        ach = dom["alg_bytes"] / 1e9 / (dom["busy_ms"] / 1e3) if dom["busy_ms"] > 0 else 0.0
        # HBM bytes per launch of the dominant kernel from the committed PMC passes of this same workload
        # (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, scripts/pmc_summary.py), else null
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            if f"--batch {a.batch} " in pmc["command"] + " " and f"--workers {a.workers} " in pmc["command"] + " ":
                traffic = pmc["kernels"][dom["kernel"]]["hbm_bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            pass
        truth = samples[-1][0].truth_haps.sum(axis=0)
        r2_truth = float(np.corrcoef(last[0].dosage, truth)[0, 1] ** 2)
        value = a.batch * world * a.steps / elapsed
        out = {
            "metric": "samples/sec on 2Mb region, K=50k haps, 1x coverage; dosage r2 vs CPU ref",
            "value": value, "unit": "samples/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64 (Gibbs sampler, ranking passes) / f32 state with f64 emissions and sums (dosage passes)",
            "data": "synthetic",
            "config": {"workload": f"{a.batch} synthetic 1x short-read samples per GPU per step, {a.nsnps} SNPs "
                                   f"({panel.nGrids} grids, 2 Mb + buffers), K={a.K} haplotypes, {a.reads} reads/sample, "
                                   "QUILT defaults (nGibbsSamples=7, n_seek_its=3, Ksubset=600), use_mspbwt=FALSE",
                       "K": a.K, "nSNPs": a.nsnps, "samples_per_step_per_gpu": a.batch,
                       "parallelism": f"samples sharded over {world} GPU(s), no collective; {a.workers} host threads per "
                                      "GPU, consecutive batches pipelined"},
            "roofline": {"bound": "hbm", "kernel": dom["kernel"], "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                         "avg_launch_ms": dom["ms"] / max(dom["launches"], 1), "launches": dom["launches"],
                         "busy_ms": dom["busy_ms"],
                         "alg_bytes_per_launch": dom["alg_bytes"] / max(dom["launches"], 1)},
            "kernels": [{"kernel": p["kernel"], "ms": round(p["ms"], 2), "busy_ms": round(p["busy_ms"], 2), "launches": p["launches"],
                         "GBps": (p["alg_bytes"] / 1e9 / (p["busy_ms"] / 1e3)) if p["busy_ms"] > 0 else 0.0} for p in prof],
            "host_seconds": {k: round(v, 3) for k, v in drv.timing.items()},
            "dosage_r2_vs_truth_sample0": r2_truth,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

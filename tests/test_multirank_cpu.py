"""N > 1 path on CPU: two gloo ranks each impute their contiguous sample range (no collective on the data
path); the concatenation equals the single-process result bit for bit, and the timing reduction is a MAX."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from quilt_amd.driver import Driver, DriverParams
    from quilt_amd.sharding import get_sample_range
    from quilt_amd.synth import make_synthetic_panel, make_synthetic_sample
    from tests.oracle_backend import OracleBackend
    panel = make_synthetic_panel(K=600, nSNPs=640, seed=3)
    N = 5
    lo, hi = get_sample_range(N, world)[rank]
    samples = [make_synthetic_sample(panel, seed=100 + i, n_reads=150) for i in range(N)]
    prm = DriverParams(nGibbsSamples=2, Ksubset=64, Knew=64, seed=9)
    drv = Driver(panel, OracleBackend(panel), prm)
    res = drv.run(samples[lo:hi], sample_offset=lo)
    t = torch.tensor([float(rank + 1)])
    dist.barrier()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    gathered = [None] * world
    dist.all_gather_object(gathered, [(lo + i, r.dosage) for i, r in enumerate(res)])
    # the run's only cross-shard reduction: the per-SNP count arrays behind the VCF's INFO column (writers.R:38-47)
    from quilt_amd.io import SummaryCounts, per_sample_counts
    from quilt_amd.sharding import reduce_counts
    counts = SummaryCounts(panel.nSNPs)
    for s, r in zip(samples[lo:hi], res):
        counts.add_sample(*per_sample_counts(r.gp_t, s, panel.nSNPs))
    reduce_counts(counts)
    if rank == 0:
        q.put((float(t.item()), [x for part in gathered for x in part], counts.as_vector()))
    dist.destroy_process_group()


def test_two_ranks_equal_one_process():
    sys.path.insert(0, ROOT)
    from quilt_amd.driver import Driver, DriverParams
    from quilt_amd.sharding import get_sample_range
    from quilt_amd.synth import make_synthetic_panel, make_synthetic_sample
    from tests.oracle_backend import OracleBackend
    assert get_sample_range(5, 2) == [(0, 2), (2, 5)] and get_sample_range(2, 4) == [(0, 1), (1, 2), (2, 2), (2, 2)]
    assert get_sample_range(1024, 8)[3] == (384, 512)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    tmax, parts, count_vec = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert tmax == 2.0
    panel = make_synthetic_panel(K=600, nSNPs=640, seed=3)
    samples = [make_synthetic_sample(panel, seed=100 + i, n_reads=150) for i in range(5)]
    ref = Driver(panel, OracleBackend(panel), DriverParams(nGibbsSamples=2, Ksubset=64, Knew=64, seed=9)).run(samples)
    assert [i for i, _ in parts] == [0, 1, 2, 3, 4]
    for (i, d), r in zip(parts, ref):
        assert np.array_equal(d, r.dosage)
    # the summed count arrays equal the single-process sums
    from quilt_amd.io import SummaryCounts, per_sample_counts
    one = SummaryCounts(panel.nSNPs)
    for s, r in zip(samples, ref):
        one.add_sample(*per_sample_counts(r.gp_t, s, panel.nSNPs))
    np.testing.assert_allclose(count_vec, one.as_vector(), rtol=1e-12, atol=1e-12)


def test_bench_gpus_flag_creates_the_ranks():
    """`bench.py --gpus 2` launched plainly re-executes itself under torch.distributed.run with two ranks (the driver's
    own launch line sets WORLD_SIZE and is taken as it is); with the stub driver (no device work, gloo) the line
    reports n_gpus = 2.  Reference analogue: mclapply(mc.cores = nCores), quilt.R:691-692."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub", "--gpus", "2", "--K", "300", "--nsnps", "640",
                          "--batch", "2", "--reads", "40", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line"
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["warmup"] == 1
    assert rec["stub"] is True and rec["value"] == 0.0 and rec["scaling"] == "weak"
    # and the launcher's own form: WORLD_SIZE already set, no re-exec
    env1 = dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub", "--gpus", "1", "--K", "300", "--nsnps", "640",
                          "--batch", "2", "--reads", "40", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=600, env=env1, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    assert json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])["n_gpus"] == 1


def test_eight_ranks_plumbing_on_the_stub_backend():
    """What the first real 8-GPU run must not fail on: eight ranks (gloo, stub driver), the host cores divided between ranks and
    host threads (QA_HOST_THREADS), rank 0's CPU-leg flag file honoured by the other seven and removed afterwards, one JSON
    line with per-rank rates.  Reference analogue: mclapply(mc.cores = nCores) over sampleRanges (quilt.R:688-692)."""
    import glob
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "QA_HOST_THREADS")}
    before = set(glob.glob("/tmp/quilt_amd_bench_*.cpu_done"))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub", "--gpus", "8", "--K", "300", "--nsnps", "640",
                          "--batch", "2", "--reads", "40", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line"
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 8 and rec["scaling"] == "weak"
    rk = rec["ranks"]
    assert rk["n"] == 8 and len(rk["elapsed_s"]) == 8 and len(rk["QA_HOST_THREADS"]) == 8
    assert "samples_per_sec_min" in rk and "samples_per_sec_max" in rk
    # the cores are divided: every rank got the same budget, at least 2, and 8 ranks x host threads x budget fits the machine
    # (or sits at the floor of 2 on a small one)
    b = set(rk["QA_HOST_THREADS"])
    assert len(b) == 1 and min(b) >= 2
    budget = min(b)
    assert budget == max(2, (os.cpu_count() or 16) // (8 * rk["host_threads_per_rank"]))
    assert rk["waited_for_rank0_cpu_legs_s"][0] == 0.0 and all(w < 300 for w in rk["waited_for_rank0_cpu_legs_s"])
    assert set(glob.glob("/tmp/quilt_amd_bench_*.cpu_done")) <= before, "rank 0 removes the flag file"

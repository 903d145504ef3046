"""How the native loader scales with host threads on the GPU box: qa_impute_bam_range with a minimum read count nobody meets runs its
load phase only.   gpurun -- 'python scripts/perf_bam_load.py'"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiprocessing as mp
import numpy as np
from quilt_amd.synth import make_synthetic_panel, make_synthetic_sample, synthetic_alleles, write_synthetic_bam

N = int(os.environ.get("N_FILES", "512"))
panel = make_synthetic_panel(K=2000, nSNPs=64000, seed=4916)
ref, alt = synthetic_alleles(panel.nSNPs, 1)
d = tempfile.mkdtemp(prefix="qa_load_")


def mk(i):
    s = make_synthetic_sample(panel, seed=i, n_reads=20000)
    write_synthetic_bam(os.path.join(d, f"s{i}.bam"), s, panel.L, ref, alt, seed=i)
    return i


with mp.get_context("fork").Pool(32) as pool:
    pool.map(mk, range(N), chunksize=4)
files = [os.path.join(d, f"s{i}.bam") for i in range(N)]
print("files", N, "bytes each", os.path.getsize(files[0]))
from quilt_amd.driver import DriverParams
from quilt_amd.impute import impute_bam_range
from quilt_amd.native import DevicePanel
dev = DevicePanel(panel)
for nt in (1, 8, 16, 32, 64, 128):
    t = time.perf_counter()
    r = impute_bam_range([dev], files, "chr20", ref, alt, DriverParams(), minimum_number_of_sample_reads=10 ** 9, n_io_threads=nt,
                         downsampleToCov=0, bqFilter=1)
    w = time.perf_counter() - t
    print(f"{nt:4d} threads: load {r['seconds']['load']:.3f} s  ({1e3 * r['seconds']['load'] * nt / N:.1f} thread-ms per file), call {w:.3f} s")
dev.close()

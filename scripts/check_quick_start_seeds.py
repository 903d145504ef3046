"""Developer aid: tests/test_pipeline_gpu.py::test_quick_start_shaped_run_bam_to_vcf for several driver seeds: which of them give the
device and the CPU path the same labels (no last-bit tie at a best-haplotype list's threshold on the way)?"""
import os, sys, tempfile, pathlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from quilt_amd.driver import DriverParams, HipBackend
from quilt_amd.native import DevicePanel
from quilt_amd.synth import make_1000g_like_panel
from tests.oracle_backend import OracleBackend
from tests.test_driver_host import _bam_to_vcf
if os.environ.get("QA_CHECK_NUMPY_POW"):   # the Python side as it was before driver.phred_eps: numpy's vectorised power
    import quilt_amd.driver as _d
    _d.phred_eps = lambda bq: 10.0 ** (-np.abs(np.asarray(bq)) / 10.0)
panel = make_1000g_like_panel(K=5008, nSNPs=3200, seed=2504)
dev = DevicePanel.from_rhb(panel)
dev.set_dosage_precision(64)
for sd in range(1, 11):
    prm = DriverParams(seed=sd)
    with tempfile.TemporaryDirectory() as t:
        t = pathlib.Path(t); (t / "g").mkdir(); (t / "c").mkdir()
        rg = _bam_to_vcf(t / "g", panel, HipBackend(dev), n_samples=1, n_reads=1000, prm=prm)[1]
        rc = _bam_to_vcf(t / "c", panel, OracleBackend(panel, n_threads=8), n_samples=1, n_reads=1000, prm=prm)[1]
    a, b = rg["results"][0], rc["results"][0]
    print(sd, bool(np.array_equal(a.read_labels, b.read_labels)), float(np.abs(a.dosage - b.dosage).max()), flush=True)
dev.close()

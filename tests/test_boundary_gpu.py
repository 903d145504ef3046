"""The C ABI driven from plain C on the GPU (tests/c/c_harness.c): panel upload, one Gibbs call, one full-panel pass,
invariants checked in C.  The same sequence the R shim (shim/quilt_amd_shim.c) performs."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_harness_on_device():
    out = subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "c")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-3000:]
    run = subprocess.run([os.path.join(ROOT, "tests", "c", "c_harness")], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "HARNESS_OK" in run.stdout and "IMPUTE_SAMPLES_OK" in run.stdout, run.stdout + run.stderr


def test_pinned_host_buffers_take_the_same_bytes_as_staged_ones():
    """qa_host_alloc buffers skip the staging copy (include/quilt_amd.h): same results either way, both directions."""
    import ctypes as C
    import numpy as np
    from quilt_amd.gibbs_nipt import calculate_eMatRead_t_vs_haplotypes_batch
    from quilt_amd.native import DevicePanel, QuiltAmdError, lib, pinned_empty
    from quilt_amd.synth import make_synthetic_panel, make_synthetic_sample
    from quilt_amd.driver import HipBackend, thinned_grid_columns
    panel = make_synthetic_panel(K=512, nSNPs=96 * 32, seed=3)
    dev = DevicePanel(panel)
    samples = [make_synthetic_sample(panel, seed=40 + i, n_reads=300) for i in range(3)]
    rng = np.random.default_rng(5)
    # upload direction: the read-confidence call with its haplotypes in a pinned buffer and in plain memory
    haps = rng.random((3, 2, panel.nSNPs))
    pinned = pinned_empty(haps.shape)
    pinned[...] = haps
    a = calculate_eMatRead_t_vs_haplotypes_batch(dev, samples, haps, 1000.0, hap_major=True)
    b = calculate_eMatRead_t_vs_haplotypes_batch(dev, samples, pinned, 1000.0, hap_major=True)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    # download direction: the backend's dosage rounds land in its pinned buffer; the same call through the raw entry point
    # with a plain numpy buffer returns the same bytes
    be = HipBackend(dev)
    labels = [rng.integers(1, 3, size=s.nReads).astype(np.int32) for s in samples]
    cols = thinned_grid_columns(panel.nGrids, 0.1)
    d1 = be.fullpass_reads_batch(samples, [0, 1, 2], labels, [True] * 3, [False] * 3, cols, 5, 1e-10, 8)[0].copy()
    assert be._dosage_buf is not None
    be2 = HipBackend(dev)
    import quilt_amd.native as native
    keep = native.pinned_empty
    try:   # a backend whose "pinned" buffer is ordinary memory: the staged path
        native.pinned_empty = lambda shape, dtype=np.float64: np.empty(shape, dtype=dtype)
        d2 = be2.fullpass_reads_batch(samples, [0, 1, 2], labels, [True] * 3, [False] * 3, cols, 5, 1e-10, 8)[0].copy()
    finally:
        native.pinned_empty = keep
    assert np.array_equal(d1, d2)
    assert d1.min() >= 0 and d1.max() <= 1 and d1.std() > 0
    # errors: a pointer the library did not hand out
    lib().qa_host_free.argtypes = [C.c_void_p]
    assert lib().qa_host_free(C.c_void_p(haps.ctypes.data)) < 0
    del pinned, b

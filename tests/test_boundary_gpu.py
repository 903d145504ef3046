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
    assert "HARNESS_OK" in run.stdout, run.stdout + run.stderr

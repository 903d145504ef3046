"""The two statements of the per-sample loop over EVERY case of tests/mode_matrix.py ON THE DEVICE: quilt_amd/driver.py over the
library's entry points (HipBackend) and qa_impute_samples (csrc/impute.cpp), two host threads, launch sets of two samples -- the
same bytes (every chain owns its random stream; the kernels are deterministic); and once more with the samples handed over one
by one (params->sample_source)."""
import numpy as np
import pytest

from tests import mode_matrix as MM

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,kw", MM.cases(), ids=[n for n, _ in MM.cases()])
def test_both_loops_return_the_same_bytes_on_the_device(name, kw):
    from quilt_amd.driver import Driver, HipBackend
    from quilt_amd.impute import impute_samples
    from quilt_amd.native import DevicePanel, DeviceRareCommon
    panel, rc, samples, P = MM.make_case(kw, reads=300, seed0=70)   # (K = 300 x 20 grids: 'panel_smaller_than_Ksubset' needs K < 600)
    dev = DevicePanel(panel)
    dev.set_dosage_precision(64)
    drc = DeviceRareCommon(dev, rc) if rc is not None else None
    want = Driver(panel, HipBackend(dev, drc), P, rare_common=rc).run(samples, sample_offset=3)
    devs = [DevicePanel(panel) for _ in range(2)]
    for d in devs:
        d.set_device_share(2)
        d.set_dosage_precision(64)
        d.set_exclusive(True)
    drcs = [DeviceRareCommon(d, rc) for d in devs] if rc is not None else ()
    got = impute_samples(devs, samples, P, sample_offset=3, samples_per_launch_set=2, drcs=drcs)
    again = impute_samples(devs, samples, P, sample_offset=3, samples_per_launch_set=2, drcs=drcs, one_by_one=True)
    for x in list(drcs) + ([drc] if drc is not None else []):
        x.close()
    for d in devs + [dev]:
        d.close()
    for res in (got, again):
        assert len(res) == len(want)
        for a, b in zip(res, want):
            assert a.nDosage == b.nDosage
            for f in ("read_labels", "dosage", "gp_t", "phasing_haps"):
                assert np.array_equal(getattr(a, f), getattr(b, f)), f"{name}: {f}"
            if P.method == "nipt":
                assert np.array_equal(a.fet_dosage, b.fet_dosage) and np.array_equal(a.fet_gp_t, b.fet_gp_t), name

"""The two statements of the per-sample loop over EVERY case of tests/mode_matrix.py ON THE DEVICE: quilt_amd/driver.py over the
library's entry points (HipBackend) and qa_impute_samples (csrc/impute.cpp), two host threads, launch sets of two samples -- the
same bytes (every chain owns its random stream; the kernels are deterministic); and once more with the samples handed over one
by one (params->sample_source)."""
import numpy as np
import pytest

from tests import mode_matrix as MM

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rig():
    """One panel (and its all-SNP side) on the device for all cases: a handle for the Python loop, two for the native loop's threads."""
    from quilt_amd.native import DevicePanel, DeviceRareCommon
    panel, rc, _, _ = MM.make_case(dict(MM.BASE, impute_rare_common=True), n_samples=0)
    dev = DevicePanel(panel)
    dev.set_dosage_precision(64)
    devs = [DevicePanel(panel) for _ in range(2)]
    for d in devs:
        d.set_device_share(2)
        d.set_dosage_precision(64)
        d.set_exclusive(True)
    drc = DeviceRareCommon(dev, rc)
    drcs = [DeviceRareCommon(d, rc) for d in devs]
    yield dict(panel=panel, rc=rc, dev=dev, devs=devs, drc=drc, drcs=drcs)
    for x in drcs + [drc]:
        x.close()
    for d in devs + [dev]:
        d.close()


@pytest.mark.parametrize("name,kw", MM.cases(), ids=[n for n, _ in MM.cases()])
def test_both_loops_return_the_same_bytes_on_the_device(rig, name, kw):
    from quilt_amd.driver import Driver, HipBackend
    from quilt_amd.impute import impute_samples
    panel, rc, samples, P = MM.make_case(kw, reads=300, seed0=70)   # (K = 300 x 20 grids: 'panel_smaller_than_Ksubset' needs K < 600)
    assert panel.K == rig["panel"].K and np.array_equal(panel.hapMatcher, rig["panel"].hapMatcher)   # (the rig's panel, rebuilt from its seed)
    rare = rc is not None
    want = Driver(rig["panel"], HipBackend(rig["dev"], rig["drc"] if rare else None), P, rare_common=rig["rc"] if rare else None).run(samples, sample_offset=3)
    drcs = rig["drcs"] if rare else ()
    got = impute_samples(rig["devs"], samples, P, sample_offset=3, samples_per_launch_set=2, drcs=drcs)
    again = impute_samples(rig["devs"], samples, P, sample_offset=3, samples_per_launch_set=2, drcs=drcs, one_by_one=True)
    for res in (got, again):
        assert len(res) == len(want)
        for a, b in zip(res, want):
            assert a.nDosage == b.nDosage
            for f in ("read_labels", "dosage", "gp_t", "phasing_haps"):
                assert np.array_equal(getattr(a, f), getattr(b, f)), f"{name}: {f}"
            if P.method == "nipt":
                assert np.array_equal(a.fet_dosage, b.fet_dosage) and np.array_equal(a.fet_gp_t, b.fet_gp_t), name

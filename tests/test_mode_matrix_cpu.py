"""The two statements of the per-sample loop -- quilt_amd/driver.py and csrc/impute.cpp -- over EVERY case of tests/mode_matrix.py
on the oracle backend: the same bytes.  Plus the guards that keep the table complete."""
import numpy as np
import pytest

from tests import mode_matrix as MM
from tests.oracle_backend import OracleBackend


def test_every_parameter_of_the_loop_is_classified():
    have, known = MM.driver_fields(), MM.classified()
    assert have - known == set(), f"DriverParams fields not entered in tests/mode_matrix.py: {sorted(have - known)}"
    assert known - have == set(), f"tests/mode_matrix.py names fields DriverParams does not have: {sorted(known - have)}"
    assert set(MM.SWITCHES) & MM.KNOBS == set()
    names = [n for n, _ in MM.cases()]
    assert len(names) == len(set(names)) and len(names) >= 8 + 2 * len(MM.BRANCHES)


def test_the_python_only_parameters_are_refused_or_inert_in_the_native_loop():
    from quilt_amd.driver import DriverParams
    from quilt_amd.impute import make_params
    with pytest.raises(ValueError, match="scan"):
        make_params(DriverParams(use_mspbwt=True, mspbwt_search="exhaustive"), 256)
    assert DriverParams().diploid_block_gibbs == "reference_noop"


@pytest.mark.parametrize("name,kw", MM.cases(), ids=[n for n, _ in MM.cases()])
def test_both_loops_return_the_same_bytes(name, kw):
    from quilt_amd.driver import Driver
    from tests.native_driver_backend import impute_samples_on_oracle
    panel, rc, samples, P = MM.make_case(kw)
    want = Driver(panel, OracleBackend(panel, rc), P, rare_common=rc).run(samples, sample_offset=3)
    got, stats, tab = impute_samples_on_oracle(panel, samples, P, sample_offset=3, samples_per_launch_set=2, n_threads=2, rare_common=rc)
    assert len(got) == len(want) == len(samples)
    for a, b in zip(got, want):
        assert a.nDosage == b.nDosage
        for f in ("read_labels", "dosage", "gp_t", "phasing_haps"):
            assert np.array_equal(getattr(a, f), getattr(b, f)), f"{name}: {f}"
        if P.method == "nipt":
            assert np.array_equal(a.fet_dosage, b.fet_dosage) and np.array_equal(a.fet_gp_t, b.fet_gp_t), name
    # the case did run the path it names
    if P.use_mspbwt:
        assert tab.calls["select"] == 0 and tab.calls["fullpass"] == 0
    if P.impute_rare_common:
        assert tab.calls["gibbs_rc"] > 0
    if name.endswith("complete_lists"):
        assert stats["full_list_refetches"] > 0

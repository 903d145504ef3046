"""everything_select_good_haps on the device (csrc/select.hip, behind qa_fullpass_reads_select_batch)
against the host function (quilt_amd/driver.py), which draws with the same keyed rule: identical which_haps_to_use.

The lists the device selects from never leave it, so the host side of the comparison takes them from the plain
qa_fullpass_reads_batch call on the same inputs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _both(panel, n_samples, n_chains, Ksubset, Knew, K_top_matches, thin=0.1, top_width=None, seed=3):
    from quilt_amd.driver import HipBackend, thinned_grid_columns
    from quilt_amd.native import DevicePanel
    from quilt_amd.synth import make_synthetic_sample
    rng = np.random.default_rng(seed)
    samples = [make_synthetic_sample(panel, seed=500 + i, n_reads=200) for i in range(n_samples)]
    chain_sample = [c % n_samples for c in range(n_chains)]
    labels = [rng.integers(1, 3, size=samples[s].nReads).astype(np.int32) for s in chain_sample]
    want_top = [1] * n_chains
    want_top[1] = 0                      # a chain without lists keeps its small panel: status -1
    cols = thinned_grid_columns(panel.nGrids, thin)
    which = [np.sort(rng.choice(panel.K, Ksubset, replace=False) + 1).astype(np.int32) for _ in range(n_chains)]
    seeds = [int(rng.integers(0, 2 ** 63)) for _ in range(n_chains)]
    top_width = top_width or K_top_matches
    dev = DevicePanel(panel)
    be = HipBackend(dev)
    _, top, cnt = be.fullpass_reads_batch(samples, chain_sample, labels, [0] * n_chains, want_top, cols, K_top_matches, 1e-10,
                                          top_width)
    _, none, cnt2, nxt, status = be.fullpass_reads_batch(samples, chain_sample, labels, [0] * n_chains, want_top, cols,
                                                         K_top_matches, 1e-10, top_width,
                                                         select=dict(Ksubset=Ksubset, Knew=Knew, which=which, seeds=seeds))
    dev.close()
    assert none is None and np.array_equal(cnt, cnt2)
    return top, cnt, which, seeds, nxt, status, want_top


@pytest.mark.parametrize("Ksubset,Knew,K_top", [(64, 64, 5), (100, 40, 5), (200, 150, 8), (400, 100, 3)])
def test_device_selection_matches_host(small_panel, Ksubset, Knew, K_top):
    from quilt_amd.driver import ListsTruncated, everything_select_good_haps_dense, previously_selected
    panel = small_panel
    top, cnt, which, seeds, nxt, status, want_top = _both(panel, 3, 7, Ksubset, Knew, K_top)
    n_done = 0
    for c in range(len(which)):
        if not want_top[c]:
            assert status[c] == -1
            continue
        prev = previously_selected(which[c], Ksubset - Knew, seeds[c])
        # the host function on a table cut to the K_top ranks the device looks at: raising where it would go on to "all entries"
        try:
            sel = everything_select_good_haps_dense(Knew, K_top, top[c].astype(np.int64) + 1, prev, panel.K, seeds[c],
                                                    truncated=True)
        except ListsTruncated:
            assert status[c] == 1, "ranks exhausted on the host: the device must hand the chain back"
            continue
        assert status[c] == 0
        assert np.array_equal(nxt[c], np.concatenate([prev, sel]))
        assert len(set(nxt[c].tolist())) == Ksubset and nxt[c].min() >= 1 and nxt[c].max() <= panel.K
        n_done += 1
    assert n_done >= 1 or (status[np.array(want_top) == 1] == 1).all()


def test_device_selection_hands_back_when_ranks_run_out(small_panel):
    """Knew larger than the distinct candidates of K_top ranks: every chain comes back with status 1."""
    top, cnt, which, seeds, nxt, status, want_top = _both(small_panel, 2, 4, 900, 900, 1, thin=0.5)
    assert all(status[c] == 1 for c in range(4) if want_top[c])


def test_driver_uses_the_device_selection(medium_panel):
    """Driver on the HIP backend with and without the device selection, and on the CPU path: the same small panels, hence the
    same labels and dosages."""
    from quilt_amd.driver import Driver, DriverParams, HipBackend
    from quilt_amd.native import DevicePanel
    from quilt_amd.synth import make_synthetic_sample
    from tests.oracle_backend import OracleBackend
    panel = medium_panel
    samples = [make_synthetic_sample(panel, seed=40 + i, n_reads=600) for i in range(2)]
    prm = DriverParams(nGibbsSamples=2, n_seek_its=3, Ksubset=200, Knew=40, seed=5)   # 2 labels x 10 thinned grids x 5 ranks >= Knew
    dev = DevicePanel(panel)
    d_on = Driver(panel, HipBackend(dev), prm)
    got = d_on.run(samples)
    be_off = HipBackend(dev)
    be_off.select_on_device = False
    d_off = Driver(panel, be_off, prm)
    got_off = d_off.run(samples)
    dev.close()
    assert d_on.n_device_selections > 0 and d_off.n_device_selections == 0
    for a, b in zip(got, got_off):
        assert np.array_equal(a.read_labels, b.read_labels) and np.array_equal(a.dosage, b.dosage)
    ref = Driver(panel, OracleBackend(panel), prm).run(samples)
    for a, r in zip(got, ref):
        assert np.array_equal(a.read_labels, r.read_labels) and np.abs(a.dosage - r.dosage).max() <= 1e-4

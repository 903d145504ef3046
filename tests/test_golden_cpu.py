"""The oracle against the committed fixtures of tests/golden (see make_golden.py for what they are and are not)."""
import json
import os

import numpy as np

from oracle import oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_reference_known_answers():
    ka = json.load(open(os.path.join(GOLD, "known_answers.json")))
    for c in ka["rcpp_simple_binary_search"]["cases"]:
        assert O.simple_binary_search(c["val"], np.array(c["vec"], dtype=np.int32)) == c["expect_0based"]
    for c in ka["rcpp_simple_binary_matrix_search"]["cases"]:
        keys = np.array(c["keys"], dtype=np.int32)
        mat = np.stack([keys, keys + 10], axis=1).astype(np.int32)
        assert O.simple_binary_matrix_search(c["val"], mat, c["s1"], c["e1"]) == c["expect"]
    g = ka["Rcpp_make_gl_bound"]
    gl = np.asfortranarray(np.array(g["gl_in"], dtype=np.float64))
    O.make_gl_bound(gl, g["minGLValue"], np.arange(gl.shape[1], dtype=np.int32))
    np.testing.assert_allclose(gl, np.array(g["gl_out"]), rtol=1e-15)
    for c in ka["get_top_K_or_more_matches"]["cases"]:
        idx, _, _ = O.get_top_K_or_more_matches(np.array(c["alpha"], dtype=float), np.array(c["beta"], dtype=float), c["K_top"])
        assert idx.tolist() == c["expect_k_0based"]


def _problem():
    from quilt_amd.synth import make_synthetic_panel, make_synthetic_sample
    from tests.golden.make_golden import PANEL, SAMPLE
    panel = make_synthetic_panel(**PANEL)
    return panel, make_synthetic_sample(panel, **SAMPLE)


def test_oracle_reproduces_fullpass_fixture():
    panel, sample = _problem()
    z = np.load(os.path.join(GOLD, "fullpass_small.npz"))
    for label in (1, 2):
        r = O.haploid_dosage_versus_refs(panel, z[f"gl{label}"], z["cols"], get_best_haps_from_thinned_sites=True,
                                         always_normalize=True)
        np.testing.assert_allclose(r["dosage"], z[f"dosage{label}"], rtol=0, atol=1e-14)
        np.testing.assert_allclose(r["c"], z[f"c{label}"], rtol=1e-13)
        for j, (idx, val) in enumerate(r["best_haps"]):
            assert np.array_equal(idx, z[f"best_idx{label}_{j}"])
            np.testing.assert_allclose(val, z[f"best_val{label}_{j}"], rtol=1e-13)


def test_oracle_reproduces_gibbs_fixture():
    from quilt_amd.rng import stream_uniform
    panel, sample = _problem()
    z = np.load(os.path.join(GOLD, "gibbs_small.npz"))
    ru = stream_uniform(int(z["seed_reads"]), sample.nReads * 21)
    rs = stream_uniform(int(z["seed_shard"]), 3 * (panel.nGrids - 1))
    g = O.forwardBackwardGibbsNIPT(panel, sample, z["which"], z["H0"], ru, int(z["first_read"]), rs,
                                   gibbs_initialize_iteratively=True)
    assert np.array_equal(g["H"], z["H"]) and int(g["underflow_problem"]) == int(z["underflow"])
    np.testing.assert_allclose(g["hapProbs_t"], z["hapProbs_t"], rtol=0, atol=1e-13)


def test_oracle_reproduces_nipt_and_rare_common_fixtures():
    from quilt_amd.rng import stream_uniform
    from quilt_amd.synth import make_rare_common, make_synthetic_sample, make_synthetic_sample_rare_common
    from tests.golden.make_golden import RC_SEED, SAMPLE
    panel, _ = _problem()
    z = np.load(os.path.join(GOLD, "nipt_small.npz"))
    ff = float(z["ff"])
    s3 = make_synthetic_sample(panel, seed=SAMPLE["seed"] + 1, n_reads=60, ff=ff)
    R = s3.nReads
    blk = stream_uniform(int(z["seed_shard"]), 3 * 2 * R).reshape(3, 2, R)
    g = O.forwardBackwardGibbsNIPT(panel, s3, z["which"], z["H0"], stream_uniform(int(z["seed_reads"]), R * 21),
                                   int(z["first_read"]), np.zeros(3 * panel.nGrids), ff=ff, gibbs_initialize_iteratively=True,
                                   runif_block=blk[:, 0, :].copy(), runif_resample=blk[:, 1, :].copy())
    assert np.array_equal(g["H"], z["H"]) and np.array_equal(g["H_class"], z["H_class"])
    np.testing.assert_allclose(g["hapProbs_t"], z["hapProbs_t"], rtol=0, atol=1e-13)
    z = np.load(os.path.join(GOLD, "rare_common_small.npz"))
    rc = make_rare_common(panel, RC_SEED)
    _, s_all = make_synthetic_sample_rare_common(panel, rc, SAMPLE["seed"] + 2, n_reads=60)
    R = s_all.nReads
    g = O.forwardBackwardGibbsNIPT(panel, s_all, z["which"], z["H0"], stream_uniform(int(z["seed_reads"]), R * 21), 0,
                                   stream_uniform(int(z["seed_shard"]), 3 * (rc.nGrids_all - 1)),
                                   disable_read_category_usage=True, rare_common=rc)
    assert np.array_equal(g["H"], z["H"])
    np.testing.assert_allclose(g["hapProbs_t"], z["hapProbs_t"], rtol=0, atol=1e-13)

"""Generates the fixtures in this directory.  Run from the repo root:  python tests/golden/make_golden.py

The reference (R + Rcpp) cannot be built or run in this environment and stores no golden vectors of its own
(SURVEY.md 8(c)); what can be pinned is:
  * known_answers.json -- the RNG-independent known answers asserted by the reference's own tests, transcribed by hand
    (inputs and expected outputs only; the file:line of each is in the entry);
  * fullpass_small.npz / gibbs_small.npz / nipt_small.npz / rare_common_small.npz -- outputs of this repo's fp64 oracle
    (oracle/) on small seeded problems (diploid full pass and Gibbs; NIPT Gibbs with its block passes; the rare + common
    all-SNP Gibbs call), so
    that a change of the oracle or of the HIP path shows up as a diff against committed data.  They are NOT outputs of
    the reference ("parity unpinned", see oracle/quilt_oracle.h).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import oracle as O                                   # noqa: E402
from quilt_amd.rng import stream_uniform                          # noqa: E402
from quilt_amd.synth import (make_rare_common, make_synthetic_panel, make_synthetic_sample,   # noqa: E402
                             make_synthetic_sample_rare_common)
from tests.util import label_gl, thin_cols                        # noqa: E402

PANEL = dict(K=300, nSNPs=160, seed=77, nMaxDH=20)
SAMPLE = dict(seed=78, n_reads=40)
RC_SEED = 79


def known_answers():
    return {
        "rcpp_simple_binary_search": {
            "source": "QUILT/tests/testthat/test-unit-reference-single.R:150-166",
            "cases": [{"val": 7, "vec": [7], "expect_0based": 0},
                      {"val": 30, "vec": [10, 20, 30, 40, 50], "expect_0based": 2},
                      {"val": 10, "vec": [10, 20, 30, 40, 50], "expect_0based": 0},
                      {"val": 50, "vec": [10, 20, 30, 40, 50], "expect_0based": 4}]},
        "rcpp_simple_binary_matrix_search": {
            "source": "QUILT/tests/testthat/test-unit-reference-single.R:168-206 (value column = key + 10; the search "
                      "returns the value of the matching row within rows s1..e1, 1-based inclusive)",
            "cases": [{"val": 40, "keys": [5, 6, 10, 20, 30, 40, 50, 7, 8], "s1": 3, "e1": 7, "expect": 50},
                      {"val": 10, "keys": [5, 6, 10, 20, 30, 40, 50, 7, 8], "s1": 3, "e1": 7, "expect": 20},
                      {"val": 50, "keys": [10, 20, 30, 40, 50], "s1": 1, "e1": 5, "expect": 60}]},
        "Rcpp_make_gl_bound": {
            "source": "QUILT/tests/testthat/test-unit-reference-single.R:31-59 (larger member -> 1, other floored)",
            "minGLValue": 1e-10,
            "gl_in": [[1e-30, 0.2, 0.5], [1e-3, 1e-40, 0.25]],
            "gl_out": [[1e-10, 1.0, 1.0], [1.0, 1e-10, 0.5]]},
        "get_top_K_or_more_matches": {
            "source": "QUILT/tests/testthat/test-unit-reference-single.R:102-145 (distinct values: the K largest in "
                      "ascending k; an all-tied vector returns every k)",
            "cases": [{"alpha": [0.1, 0.9, 0.3, 0.8, 0.2, 0.7], "beta": [1, 1, 1, 1, 1, 1], "K_top": 3,
                       "expect_k_0based": [1, 3, 5]},
                      {"alpha": [1, 1, 1, 1], "beta": [1, 1, 1, 1], "K_top": 2, "expect_k_0based": [0, 1, 2, 3]}]},
    }


def main():
    with open(os.path.join(HERE, "known_answers.json"), "w") as f:
        json.dump(known_answers(), f, indent=1)
    panel = make_synthetic_panel(**PANEL)
    sample = make_synthetic_sample(panel, **SAMPLE)
    cols = thin_cols(panel.nGrids, every=2)
    out = {}
    for label in (1, 2):
        gl = label_gl(panel, sample, label, O)
        r = O.haploid_dosage_versus_refs(panel, gl, cols, get_best_haps_from_thinned_sites=True, always_normalize=True)
        out[f"gl{label}"] = gl
        out[f"dosage{label}"] = r["dosage"]
        out[f"c{label}"] = r["c"]
        for j, (idx, val) in enumerate(r["best_haps"]):
            out[f"best_idx{label}_{j}"] = idx
            out[f"best_val{label}_{j}"] = val
    out["cols"] = cols
    np.savez_compressed(os.path.join(HERE, "fullpass_small.npz"), **out)

    rng = np.random.default_rng(5)
    Ks = 64
    which = np.sort(rng.choice(panel.K, Ks, replace=False)).astype(np.int32) + 1
    H0 = rng.integers(1, 3, size=sample.nReads).astype(np.int32)
    seed_reads, seed_shard, first_read = 123456789, 987654321, 7
    n_its, nb = 21, 3
    ru = stream_uniform(seed_reads, sample.nReads * n_its)
    rs = stream_uniform(seed_shard, nb * (panel.nGrids - 1))
    g = O.forwardBackwardGibbsNIPT(panel, sample, which, H0, ru, first_read, rs, gibbs_initialize_iteratively=True)
    np.savez_compressed(os.path.join(HERE, "gibbs_small.npz"), which=which, H0=H0, seed_reads=np.uint64(seed_reads),
                        seed_shard=np.uint64(seed_shard), first_read=np.int32(first_read), H=g["H"],
                        hapProbs_t=g["hapProbs_t"], underflow=np.int32(g["underflow_problem"]))
    # NIPT: three labels, block passes after sweeps 3, 6, 9; uniforms from the two counter-based streams
    ff = 0.2
    s3 = make_synthetic_sample(panel, seed=SAMPLE["seed"] + 1, n_reads=60, ff=ff)
    R = s3.nReads
    H0 = (rng.choice(3, size=R, p=[0.5, 0.4, 0.1]) + 1).astype(np.int32)
    blk = stream_uniform(seed_shard, nb * 2 * R).reshape(nb, 2, R)
    g = O.forwardBackwardGibbsNIPT(panel, s3, which, H0, stream_uniform(seed_reads, R * n_its), first_read,
                                   np.zeros(nb * panel.nGrids), ff=ff, gibbs_initialize_iteratively=True,
                                   runif_block=blk[:, 0, :].copy(), runif_resample=blk[:, 1, :].copy())
    np.savez_compressed(os.path.join(HERE, "nipt_small.npz"), which=which, H0=H0, ff=np.float64(ff),
                        seed_reads=np.uint64(seed_reads), seed_shard=np.uint64(seed_shard), first_read=np.int32(first_read),
                        H=g["H"], H_class=g["H_class"], hapProbs_t=g["hapProbs_t"], genProbsF_t=g["genProbsF_t"],
                        underflow=np.int32(g["underflow_problem"]))
    # rare + common: the all-SNP Gibbs call (diploid), starting labels given, read categories off
    rc = make_rare_common(panel, RC_SEED)
    _, s_all = make_synthetic_sample_rare_common(panel, rc, SAMPLE["seed"] + 2, n_reads=60)
    R = s_all.nReads
    H0 = rng.integers(1, 3, size=R).astype(np.int32)
    g = O.forwardBackwardGibbsNIPT(panel, s_all, which, H0, stream_uniform(seed_reads, R * n_its), 0,
                                   stream_uniform(seed_shard, nb * (rc.nGrids_all - 1)), disable_read_category_usage=True,
                                   rare_common=rc)
    np.savez_compressed(os.path.join(HERE, "rare_common_small.npz"), which=which, H0=H0,
                        seed_reads=np.uint64(seed_reads), seed_shard=np.uint64(seed_shard), H=g["H"],
                        hapProbs_t=g["hapProbs_t"], underflow=np.int32(g["underflow_problem"]))
    print("written:", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()

"""CPU-oracle backend for :class:`quilt_amd.driver.Driver` (test infrastructure: lets the tests run the
whole per-sample pipeline on the fp64 oracle and compare it with the HIP pipeline)."""
import numpy as np

from oracle import oracle as O


class OracleBackend:
    def __init__(self, panel):
        self.panel = panel

    def make_gl_bound(self, gl, minGLValue, to_fix):
        O.make_gl_bound(gl, minGLValue, to_fix)

    def gibbs_batch(self, samples, which, starts, runif_reads, first_reads, runif_shards, *,
                    n_gibbs_burn_in_its, n_gibbs_sample_its, block_gibbs_iterations, gibbs_initialize_iteratively,
                    maxDifferenceBetweenReads, Jmax_local):
        out = []
        for s, w, h, ru, fr, rs in zip(samples, which, starts, runif_reads, first_reads, runif_shards):
            r = O.forwardBackwardGibbsNIPT(self.panel, s, w, h, ru, fr, rs,
                                           n_gibbs_burn_in_its=n_gibbs_burn_in_its,
                                           n_gibbs_sample_its=n_gibbs_sample_its,
                                           block_gibbs_iterations=block_gibbs_iterations,
                                           gibbs_initialize_iteratively=gibbs_initialize_iteratively,
                                           maxDifferenceBetweenReads=maxDifferenceBetweenReads, Jmax=Jmax_local)
            r["double_list_of_ending_read_labels"] = [[r["H"]]]
            out.append(r)
        return out

    def fullpass_batch(self, gls, want_dosage, cols, K_top_matches):
        dosages, best = [], []
        for gl, wd in zip(gls, want_dosage):
            r = O.haploid_dosage_versus_refs(self.panel, gl, cols, K_top_matches=K_top_matches,
                                             return_dosage=bool(wd), get_best_haps_from_thinned_sites=True)
            dosages.append(r["dosage"])
            best.append([dict(top_matches=i, top_matches_values=v) for i, v in r["best_haps"]])
        return dosages, best

    def read_confidence_batch(self, samples, haps, maxDifferenceBetweenReads):
        return [O.calculate_eMatRead_t_vs_haplotypes(s, h, maxDifferenceBetweenReads) for s, h in zip(samples, haps)]

"""CPU-oracle backend for :class:`quilt_amd.driver.Driver` (test infrastructure: lets the tests run the
whole per-sample pipeline on the fp64 oracle and compare it with the HIP pipeline)."""
import numpy as np

from oracle import oracle as O


class OracleBackend:
    def __init__(self, panel, rare_common=None, n_threads=1):
        """``n_threads`` > 1: the chains of a call run on a thread pool (the oracle's C calls release the GIL) -- the chains
        are independent, so nothing but the wall time changes."""
        self.panel = panel
        self.rare_common = rare_common
        self.n_threads = n_threads

    def _map(self, f, items):
        if self.n_threads <= 1 or len(items) <= 1:
            return [f(x) for x in items]
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(self.n_threads) as ex:
            return list(ex.map(f, items))

    def make_gl_bound(self, gl, minGLValue, to_fix):
        O.make_gl_bound(gl, minGLValue, to_fix)

    def gibbs_batch(self, samples, which, starts, seed_reads, first_reads, seed_shards, *,
                    n_gibbs_burn_in_its, n_gibbs_sample_its, block_gibbs_iterations, gibbs_initialize_iteratively,
                    maxDifferenceBetweenReads, Jmax_local, rare_common=False, ff=0.0, shuffle_bin_radius=5000,
                    return_hapProbs=False, return_hap_words=False):   # (the oracle always returns hapProbs_t; the driver packs)
        from quilt_amd.rng import stream_uniform
        n_its = n_gibbs_burn_in_its + n_gibbs_sample_its
        nb = len(block_gibbs_iterations)
        G = self.rare_common.nGrids_all if rare_common else self.panel.nGrids
        extra = dict(rare_common=self.rare_common, disable_read_category_usage=True) if rare_common else {}
        ffs = list(ff) if np.ndim(ff) > 0 else [ff] * len(samples)
        def one(args):
            s, w, h, sr, fr, ss, ff = args
            ex = dict(extra)
            ru = stream_uniform(sr, s.nReads * n_its)
            if ff != 0:   # NIPT: the block passes' uniforms, [pass][block choice | label re-draw][read] of the same stream
                blk = stream_uniform(ss, nb * 2 * s.nReads).reshape(nb, 2, s.nReads)
                ex = dict(ex, ff=ff, runif_block=blk[:, 0, :].copy(), runif_resample=blk[:, 1, :].copy(),
                          shuffle_bin_radius=shuffle_bin_radius,
                          L_grid=self.rare_common.L_grid_all if rare_common else None)
                rs = np.zeros(nb * G)
            else:
                rs = stream_uniform(ss, nb * (G - 1))
            init = bool(gibbs_initialize_iteratively) and fr >= 0   # per chain (include/quilt_amd.h: first_read < 0)
            r = O.forwardBackwardGibbsNIPT(self.panel, s, w, h, ru, max(fr, 0), rs,
                                           n_gibbs_burn_in_its=n_gibbs_burn_in_its,
                                           n_gibbs_sample_its=n_gibbs_sample_its,
                                           block_gibbs_iterations=block_gibbs_iterations,
                                           gibbs_initialize_iteratively=init,
                                           maxDifferenceBetweenReads=maxDifferenceBetweenReads, Jmax=Jmax_local, **ex)
            r["double_list_of_ending_read_labels"] = [[r["H"]]]
            return r
        return self._map(one, list(zip(samples, which, starts, seed_reads, first_reads, seed_shards, ffs)))

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

    def read_likelihood_all_snps_batch(self, samples_all, haps, maxDifferenceBetweenReads):
        return [O.calculate_eMatRead_t_vs_haplotypes(s, list(h.T), maxDifferenceBetweenReads, rescale_eMatRead_t=True, Jmax=100)
                for s, h in zip(samples_all, haps)]

    def fullpass_reads_batch(self, samples, chain_sample, labels, want_dosage, want_top, cols, K_top_matches, minGLValue,
                             top_width, n_label=2):
        from quilt_amd.driver import make_gl_from_u_bq
        T = self.panel.nSNPs
        n_chain = len(chain_sample)
        n_thin = int((np.asarray(cols) >= 0).sum())
        dosage = np.zeros((n_chain, n_label, T))
        top = np.full((n_chain, n_label, n_thin, top_width), -1, dtype=np.int32)
        cnt = np.zeros((n_chain, n_label, n_thin), dtype=np.int32)
        def one(c):
            s = samples[chain_sample[c]]
            per_base = np.repeat(labels[c], np.diff(s.read_ptr))
            for l in range(1, n_label + 1):
                sel = (per_base == l) & (s.bq != 0)
                gl = make_gl_from_u_bq(s.u[sel], s.bq[sel], T, minGLValue, self.make_gl_bound)
                r = O.haploid_dosage_versus_refs(self.panel, gl, cols, K_top_matches=K_top_matches,
                                                 return_dosage=bool(want_dosage[c]),
                                                 get_best_haps_from_thinned_sites=bool(want_top[c]))
                dosage[c, l - 1] = r["dosage"]
                for j, (idx, v) in enumerate(r["best_haps"] if want_top[c] else []):
                    order = np.argsort(-v, kind="stable")      # everything_per_hap_rejig_haps (functions.R:2161-2170)
                    k = idx[order][:top_width]
                    top[c, l - 1, j, : len(k)] = k
                    cnt[c, l - 1, j] = len(idx)
        self._map(one, list(range(n_chain)))
        return dosage, top, cnt

    def find_good_matches(self, Zs, nindices, min_len, max_matches):
        from quilt_amd.mspbwt import match_lists_as_tables
        return match_lists_as_tables(find_good_matches_bruteforce(self.panel, Zs, nindices, min_len, max_matches), max_matches)


    def mspbwt_select(self, Zs, n_label, nindices, L, M, Knew, seeds):
        """DriverParams.mspbwt_search = "scan": the restated neighbour scan (tests/mspbwt_scan.py, numpy, written apart from
        csrc/mspbwt.cpp) followed by the numpy text of select_new_haps_mspbwt_v3."""
        from quilt_amd.mspbwt import select_new_haps_mspbwt_v3
        from tests.mspbwt_scan import find_good_matches_scan
        found = find_good_matches_scan(self.panel, np.asarray(Zs), nindices, L, M)
        n_chain = len(found) // n_label
        return np.stack([select_new_haps_mspbwt_v3(found[c * n_label:(c + 1) * n_label], Knew, self.panel.K, self.panel.nGrids,
                                                   int(seeds[c])) for c in range(n_chain)]).astype(np.int32)


def find_good_matches_bruteforce(panel, Zs, nindices, min_len, max_matches):
    """The definition in csrc/match.hip / include/quilt_amd.h (qa_find_good_matches), written out with numpy: per haplotype its
    longest (earliest) run of matching positions within each interleaved index; the max_matches longest of those with at least
    min_len positions, ties at the cut to the lower haplotype; reported in haplotype order."""
    hm = np.asarray(panel.hapMatcherR if panel.hapMatcherR is not None else panel.hapMatcher).astype(np.int64)   # K x G
    B = np.asarray(panel.distinctHapsB)                                                                           # nMaxDH x G
    K, G = hm.shape
    out = []
    for Z in np.asarray(Zs):
        qc = np.zeros(G, dtype=np.int64)
        for g in range(G):
            w = np.nonzero(B[:, g] == Z[g])[0]
            qc[g] = w[0] + 1 if len(w) else 0
        per_index = []
        for i in range(nindices):
            grids = np.arange(i, G, nindices)
            m = (hm[:, grids] == qc[grids][None, :]) & (hm[:, grids] != 0)            # K x positions
            best_len = np.zeros(K, dtype=np.int64)
            best_start = np.zeros(K, dtype=np.int64)
            run = np.zeros(K, dtype=np.int64)
            start = np.zeros(K, dtype=np.int64)
            for j in range(len(grids)):
                start = np.where(m[:, j] & (run == 0), j, start)
                run = np.where(m[:, j], run + 1, 0)
                better = run > best_len
                best_len = np.where(better, run, best_len)
                best_start = np.where(better, start, best_start)
            ok = np.nonzero(best_len >= min_len)[0]
            order = ok[np.lexsort((ok, -best_len[ok]))][:max_matches]     # length descending, haplotype ascending
            order = np.sort(order)                                         # reported in haplotype order
            per_index.append(np.column_stack([order, best_start[order], best_len[order]]).astype(np.int32))
        out.append(per_index)
    return out

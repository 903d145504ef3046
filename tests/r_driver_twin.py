"""A SECOND statement of the per-sample driver loop, for the tests only: ``get_and_impute_one_sample`` (QUILT/R/functions.R:3-1500)
read line by line from the R and written as the R is written -- one sample at a time, one Gibbs sample after the other, one
seek iteration after the other -- with none of quilt_amd/driver.py's structure (lock-step chains, batched calls, pipelined
batches, truncated lists, flip parities, device-side selection).  It shares NO code with quilt_amd/driver.py: every piece
of host logic is restated here from the R text it cites --

  * the loop nest, starting labels, hand-over of ``which_haps_to_use`` / ``read_labels``      functions.R:400-1282
  * ``full_gammaSmall_cols_to_get``                                                             quilt.R:719-721
  * impute_one_sample's underflow retry                                                         functions.R:2612-2716
  * impute_using_everything (labels -> per-label gl -> full pass -> lists)                      functions.R:2014-2115
  * everything_per_hap_rejig_haps / everything_select_good_haps                                 functions.R:2162-2170, :2262-2310
  * the accumulation after the burn-in seek iterations and the final division                  functions.R:999-1020, :1304-1325
  * assess_ability_of_reads_to_be_confident                                                     functions.R:1610-1655
  * determine_best_read_label_so_far                                                            functions.R:1680-1784
  * recast_haps                                                                                 functions.R:3180-3209

The arithmetic (Gibbs call, full-panel pass, read likelihoods against two haplotypes) is the CPU oracle's, called one chain
at a time.  R's Mersenne-Twister cannot be reproduced without R, so the random draws are the ONE convention shared with the
product: the chain's stream (quilt_amd.rng.ChainStream) and the counter streams of quilt_amd/rng.py, consumed in the order the R code draws
(functions.R:580, :584, gibbs-nipt.cpp:2845-2848, gibbs-nipt-block.cpp:2054, functions.R:746, :2294, :2301).

tests/test_driver_twin_cpu.py compares :class:`quilt_amd.driver.Driver` on the oracle backend with this, sample by sample.
"""
import numpy as np

from oracle import oracle as O
from quilt_amd.rng import SELECT_OFFSET_POOL, SELECT_OFFSET_PREV, SELECT_OFFSET_RANK, keyed_subset, stream_uniform


def R_seq_length_out(a, b, n):
    """R's seq(a, b, length.out = n)."""
    if n == 1:
        return np.array([float(a)])
    return a + (b - a) * np.arange(n) / (n - 1)


def R_round(x):
    return float(np.round(x))   # R rounds half to even, as numpy does


def full_gammaSmall_cols_to_get(nGrids, heuristic_match_thin):
    """quilt.R:719-721: ``ww <- seq(1, nGrids, length.out = max(1, round(thin * nGrids)))``, ``cols[ww] <- 0:(length(ww) - 1)``
    (a fractional index truncates)."""
    n = int(max(1, R_round(heuristic_match_thin * nGrids)))
    ww = R_seq_length_out(1, nGrids, n)
    cols = np.full(nGrids, -1, dtype=np.int32)
    for i, w in enumerate(ww):
        cols[int(np.floor(w + 1e-9)) - 1] = i
    used = np.nonzero(cols >= 0)[0]
    assert len(used) == n, "two thinned grids fell on one index: pick nGrids / heuristic_match_thin where they do not"
    return cols


def chain_rng(seed, i_sample, i_gibbs_sample):
    """The chain's draws: the counter stream keyed by (seed, sample, Gibbs sample) -- the stand-in for R's stream."""
    from quilt_amd.rng import ChainStream
    return ChainStream(seed, i_sample, i_gibbs_sample)


def everything_per_hap_rejig_haps(best):
    """functions.R:2162-2170: ``top_matches[order(-top_matches_values)]``, 1-based (R's order() is stable)."""
    return [(np.asarray(idx) + 1)[np.argsort(-np.asarray(val), kind="stable")] for idx, val in best]


def R_unique(x):
    seen, out = set(), []
    for v in x:
        if v not in seen:
            seen.add(v)
            out.append(v)
    return out


N_EXHAUSTED = [0]   # how often the selection went past the ranked candidates (for the tests to know what they covered)


def everything_select_good_haps(Knew, K_top_matches, new_haps, previously_selected_haplotypes, K, seed_select):
    """functions.R:2262-2310.  ``new_haps[[label]][[thinned grid]]``: 1-based haplotypes, best first.  R's ``sample`` becomes
    "the smallest keys of the selection stream" (quilt_amd/rng.py)."""
    i = 1
    to_keep = []
    prev = set(int(x) for x in previously_selected_haplotypes)
    done = False
    while not done:
        if i <= K_top_matches:
            # unique(unlist(sapply(new_haps, function(x) lapply(x, function(y) y[i])))): labels outermost, grids in order
            new = R_unique([int(y[i - 1]) for x in new_haps for y in x if len(y) >= i])
        else:
            new = R_unique([int(v) for x in new_haps for y in x for v in y])
            done = True
            N_EXHAUSTED[0] += 1
        new = [v for v in new if v not in prev]               # setdiff(new, previously_selected_haplotypes)
        kept = set(to_keep)
        new = [v for v in new if v not in kept]               # setdiff(new, to_keep)
        if len(new) < Knew - len(to_keep):
            to_keep = to_keep + new
            i += 1
        else:
            toadd = Knew - len(to_keep)
            pick = keyed_subset(seed_select, len(new), toadd, SELECT_OFFSET_RANK)   # new[sample(1:length(new), toadd)]
            to_keep = to_keep + [new[j] for j in pick]
            done = True
    if len(to_keep) < Knew:
        taken = set(to_keep) | prev
        pool = [k for k in range(1, K + 1) if k not in taken]   # setdiff(1:K, c(to_keep, previously_selected_haplotypes))
        pick = keyed_subset(seed_select, len(pool), Knew - len(to_keep), SELECT_OFFSET_POOL)
        to_keep = to_keep + [pool[j] for j in pick]
    if len(to_keep) != Knew:
        raise RuntimeError("Have returned too many haps")
    return np.array(to_keep, dtype=np.int32)


def assess_ability_of_reads_to_be_confident(hap1, hap2, sample, maxDifferenceBetweenReads, minrp=0.95):
    """functions.R:1610-1655 (diploid): read likelihoods against (hap1, hap2), unscaled; confident when the larger share
    exceeds minrp."""
    p = O.calculate_eMatRead_t_vs_haplotypes(sample, [hap1, hap2], maxDifferenceBetweenReads)
    p1, p2 = np.asarray(p[0], dtype=np.float64), np.asarray(p[1], dtype=np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        mp = p1 / (p1 + p2)
    mp[np.isnan(mp)] = 0.5
    low = mp < 0.5
    mp[low] = 1 - mp[low]
    return mp > minrp


def determine_best_read_label_so_far(read_label_matrix_all, read_label_matrix_conf, nReads, nGibbsSamples, can_hap):
    """functions.R:1680-1784, statement by statement (1-based loop variables kept; the final flips use the loop's leftover
    ``i``, as the R does)."""
    rlm = np.array(read_label_matrix_all, dtype=np.float64)
    read_labels = rlm[:, can_hap - 1].astype(np.int32)
    a = rlm.copy()
    a[~np.asarray(read_label_matrix_conf, dtype=bool)] = np.nan
    a = a[np.isnan(a).sum(axis=1) == 0]
    if a.shape[0] < 10:
        return read_labels
    can = a[:, can_hap - 1].copy()
    a = a - can[:, None]
    s = np.nonzero(np.diff(np.abs(a).sum(axis=1)) != 0)[0] + 1     # which(): 1-based
    if len(s) == 0:
        return read_labels
    s1 = np.concatenate([[1], s + 1])
    nrow = a.shape[0]
    flip_matrix = np.zeros((len(s1), nGibbsSamples), dtype=bool)
    i = 1
    for i in range(2, len(s1) + 1):
        cur = a[s1[i - 1] - 1, :].copy()
        changed = [c for c in range(1, nGibbsSamples + 1) if cur[c - 1] != 0]
        w = np.arange(s1[i - 1], nrow + 1) - 1          # s1[i]:nrow(a), as 0-based rows
        if len(changed) > 0:
            if len(changed) <= nGibbsSamples / 2:
                for c1 in changed:
                    reverted = a[w, c1 - 1] + can[w]
                    reverted = 3 - reverted
                    a[w, c1 - 1] = reverted - can[w]
                assert can_hap not in changed
            else:
                changed = [c for c in range(1, nGibbsSamples + 1) if cur[c - 1] == 0]
                assert can_hap in changed
                for c1 in changed:
                    reverted = a[w, c1 - 1] + can[w]
                    reverted = 3 - reverted
                    a[w, c1 - 1] = reverted - can[w]
                reverted = a[w, :] + can[w][:, None]
                can[w] = 3 - can[w]
                a[w, :] = reverted - can[w][:, None]
        for c in changed:
            flip_matrix[i - 1, c - 1] = True
    for i_col in range(1, nGibbsSamples + 1):
        if flip_matrix[:, i_col - 1].any():
            lo = s1[i - 1]                                # w <- s1[i]:nReads with the loop's last i
            if lo <= nReads:
                rlm[lo - 1:nReads, i_col - 1] = 3 - rlm[lo - 1:nReads, i_col - 1]
    return rlm[:, can_hap - 1].astype(np.int32)


def recast_haps(hd1, hd2, gp):
    """functions.R:3180-3209; ``gp``: nSNPs x 3."""
    hd1, hd2 = np.array(hd1, dtype=np.float64), np.array(hd2, dtype=np.float64)
    gt1 = np.round(hd1) + np.round(hd2)
    max_val = gp[:, 0].copy()
    gt3 = np.zeros(gp.shape[0])
    for i in (2, 3):
        w = gp[:, i - 1] > max_val
        gt3[w] = i - 1
        max_val[w] = gp[w, i - 1]
    for t in np.nonzero(gt3 != gt1)[0]:
        if gt3[t] == 0:
            hd1[t], hd2[t] = 0, 0
        elif gt3[t] == 2:
            hd1[t], hd2[t] = 1, 1
        else:
            a1, a2 = hd1[t], hd2[t]
            if a1 > a2:
                hd1[t], hd2[t] = 1, 0
            else:
                hd1[t], hd2[t] = 0, 1
    return hd1, hd2


def get_and_impute_one_sample(panel, sample, i_sample, *, nGibbsSamples=7, n_seek_its=3, n_burn_in_seek_its=None, Ksubset=600,
                              Knew=600, K_top_matches=5, heuristic_match_thin=0.1, small_ref_panel_gibbs_iterations=20,
                              small_ref_panel_block_gibbs_iterations=(3, 6, 9), maxDifferenceBetweenReads=1e10,
                              minGLValue=1e-10, seed=1, log=None):
    """functions.R:3-1500 for method = "diploid", use_mspbwt = FALSE, impute_rare_common = FALSE.  Returns dosage, gp_t,
    phasing_haps (nSNPs x 2) and the consensus read labels the phasing iterations started from."""
    K, T, G = panel.K, panel.nSNPs, panel.nGrids
    R = sample.nReads
    if n_burn_in_seek_its is None:                 # quilt.R:248-250
        n_burn_in_seek_its = n_seek_its - 1
    if K < Ksubset:                                # quilt.R:453-463
        n_seek_its, n_burn_in_seek_its, Ksubset, Knew = 1, 0, K, K
    if Knew > Ksubset:                             # quilt.R:467-471
        Knew = Ksubset
    cols = full_gammaSmall_cols_to_get(G, heuristic_match_thin)
    n_its = small_ref_panel_gibbs_iterations + 1   # n_gibbs_sample_its = 1 (functions.R:654)
    nb = len(small_ref_panel_block_gibbs_iterations)
    dosage, gp_t, nDosage = np.zeros(T), np.zeros((3, T)), 0
    read_label_matrix_all = np.zeros((R, nGibbsSamples), dtype=np.int64)
    read_label_matrix_conf = np.zeros((R, nGibbsSamples), dtype=bool)
    per_base_read = np.repeat(np.arange(R), np.diff(sample.read_ptr))
    which_haps_to_use = read_labels = hap1 = hap2 = phasing_haps = consensus = None
    for i_gibbs_sample in range(1, nGibbsSamples + 2):
        phasing_it = i_gibbs_sample == nGibbsSamples + 1
        rng = chain_rng(seed, i_sample, i_gibbs_sample)
        for i_it in range(1, n_seek_its + 1):
            if i_it == 1 and not phasing_it:       # functions.R:579-591
                which_haps_to_use = np.sort(rng.choice(K, size=Ksubset, replace=False) + 1).astype(np.int32)
                H = rng.integers(1, 3, size=R).astype(np.int32)
                gibbs_initialize_iteratively = True
            else:                                  # :593-596 (a phasing iteration takes over read_labels and which_haps_to_use)
                H = read_labels
                gibbs_initialize_iteratively = False
            seed_reads = int(rng.integers(0, 2 ** 63))          # runif(nReads * n_its)            gibbs-nipt.cpp:2845
            first_read = int(rng.integers(0, R))                # sample(nReads, 1) - 1                          :2848
            seed_shard = int(rng.integers(0, 2 ** 63))          # runif(nGrids - 1) per shard pass  gibbs-nipt-block.cpp:2054
            runif_reads = stream_uniform(seed_reads, R * n_its)
            runif_shard = stream_uniform(seed_shard, nb * (G - 1))
            mdbr, n_imputing = maxDifferenceBetweenReads, 0
            while True:                            # impute_one_sample's retry (functions.R:2612-2716)
                g = O.forwardBackwardGibbsNIPT(panel, sample, which_haps_to_use, H, runif_reads, first_read, runif_shard,
                                               n_gibbs_burn_in_its=small_ref_panel_gibbs_iterations, n_gibbs_sample_its=1,
                                               block_gibbs_iterations=small_ref_panel_block_gibbs_iterations,
                                               gibbs_initialize_iteratively=gibbs_initialize_iteratively,
                                               maxDifferenceBetweenReads=mdbr)
                if not g["underflow_problem"]:
                    break
                mdbr = max(1.0, mdbr / 10)
                n_imputing += 1
                if n_imputing > 10:
                    raise RuntimeError("consecutive underflow problems")
            read_labels = np.asarray(g["H"], dtype=np.int32)                                       # :745
            seed_select = int(rng.integers(0, 2 ** 63))
            previously_selected_haplotypes = which_haps_to_use[                                    # :746
                keyed_subset(seed_select, len(which_haps_to_use), Ksubset - Knew, SELECT_OFFSET_PREV)]
            return_dosage = i_it > n_burn_in_seek_its                                              # :748
            # ---- impute_using_everything (functions.R:2014-2115)
            new_haps, dos = [], []
            for i_hap in (1, 2):
                sel = (read_labels[per_base_read] == i_hap) & (np.asarray(sample.bq) != 0)
                gl = O.make_gl_from_u_bq(np.asarray(sample.u)[sel], np.asarray(sample.bq)[sel], T, minGLValue)
                fp = O.haploid_dosage_versus_refs(panel, gl, cols, K_top_matches=K_top_matches, return_dosage=return_dosage,
                                                  get_best_haps_from_thinned_sites=True, always_normalize=False,
                                                  normalize_emissions=True)
                if fp["dosage"].min() < -1e-5 or fp["dosage"].max() > 1 + 1e-5:
                    raise RuntimeError("Dosage observed outside of range of 0 to 1")
                dos.append(fp["dosage"].copy())
                new_haps.append(everything_per_hap_rejig_haps(fp["best_haps"]))
            new = everything_select_good_haps(Knew, K_top_matches, new_haps, previously_selected_haplotypes, K, seed_select)
            which_haps_to_use = np.concatenate([previously_selected_haplotypes, new]).astype(np.int32)   # :951
            hap1, hap2 = dos
            if log is not None:
                log.append(dict(i_gibbs_sample=i_gibbs_sample, i_it=i_it, read_labels=read_labels.copy(),
                                which_haps_to_use=which_haps_to_use.copy()))
            if not phasing_it and i_it > n_burn_in_seek_its:                                        # :999-1006
                dosage = dosage + hap1 + hap2
                gp_t = gp_t + np.stack([(1 - hap1) * (1 - hap2), (1 - hap1) * hap2 + hap1 * (1 - hap2), hap1 * hap2])
                nDosage += 1
        if not phasing_it:                                                                          # :1144-1157
            read_label_matrix_all[:, i_gibbs_sample - 1] = read_labels
            read_label_matrix_conf[:, i_gibbs_sample - 1] = assess_ability_of_reads_to_be_confident(
                hap1, hap2, sample, maxDifferenceBetweenReads)
        if i_gibbs_sample == nGibbsSamples:                                                         # :1170-1182
            read_labels = determine_best_read_label_so_far(read_label_matrix_all, read_label_matrix_conf, R, nGibbsSamples,
                                                           can_hap=nGibbsSamples)
            consensus = read_labels.copy()
        if phasing_it:                                                                              # :1207-1217
            h1, h2 = recast_haps(hap1, hap2, gp_t.T)
            phasing_haps = np.stack([h1, h2], axis=1)
    return dict(dosage=dosage / nDosage, gp_t=gp_t / nDosage, phasing_haps=phasing_haps, read_labels=consensus,
                nDosage=nDosage)                                                                    # :1304-1311

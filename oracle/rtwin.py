"""oracle/rtwin.py -- SECOND, INDEPENDENT restatement of the hot path, from the reference's R twins (TEST INFRASTRUCTURE).

`oracle/*.c` restates the reference's C++ (QUILT/src/*.cpp).  The reference also carries readable R implementations of
the same algorithms, which its own test-suite compares with the C++ (test-unit-reference-single.R, test-unit-gibbs-*.R).
This file restates THOSE -- different source files, different parameterisation, written without looking at oracle/*.c:

  * R_haploid_dosage_versus_refs            QUILT/R/reference-single.R:94-372  (textbook scaling: alpha_g = (jump + sigma *
    alpha_{g-1}) * e, c_g = 1 / sum; the C++ keeps alpha scaled by 1 / sigma and folds sigma into c; no emission
    normalisation; specials straight from rhb_t), build_eMatDH :45-73, make_gl_from_u_bq :19-42
  * R_get_top_K_or_more_matches_while_building_gamma   QUILT/R/functions.R:2207-2258
  * forwardBackwardGibbsNIPT / gibbs_nipt_one_iteration / evaluate_read_probabilities / evaluate_read_variability /
    alpha_forward_one / make_rlc / initialize_gibbs_forward_backward   QUILT/R/gibbs-nipt.R:4-350, :508-997, :1859-1975,
    :2016-2160 (three labels always, label 3 carrying prior 0 for a diploid sample; dense eMatRead from expanded
    haplotypes), hapProbs by the dense gamma x haplotype product (save_various_gammas :358-487 with the neutral label
    probabilities)

  * R_block_gibbs_resampler with block_approach = 6 (the production value, gibbs-nipt.cpp:239) and its parts
    R_gibbs_block_forward_one / R_consider_block_relabelling / calculate_block_read_label_probabilities_using_H_class /
    R_reset_local_variables / R_make_gibbs_considers / get_log_p_H_class2   QUILT/R/gibbs-nipt-block.R:401-973, :1227-1697,
    :1921-1992, :2414-2553, :2658-2862, :3629-3660; R_define_blocked_snps_using_gamma_on_the_fly :978-1149 (with the three
    C++ helpers the R twin itself calls: rcpp_make_smoothed_rate / rcpp_determine_where_to_stop, copied-from-stitch.cpp:446-567,
    rcpp_simple_quantile, gibbs-nipt-block.cpp:81-85); sample_H_using_H_class :3715-3723
  * R_shard_block_gibbs_resampler   QUILT/R/gibbs-nipt-block.R:2874-3581 (ff == 0, shard_check_every_pair)

Two independent readings of two different reference sources agreeing -- on every output the two parameterisations share
(dosage, gamma, best-haplotype lists, read labels under the same uniforms, hapProbs, sum(log c) up to the known sigma
terms) -- is the strongest pin available in a container without R.  tests/golden/make_golden_rtwin.py
generates the committed fixtures from THIS file and cross-checks oracle/*.c against it.

Plain numpy, loops over grids and reads (small cases only).  Never imported by the product.

SUMS.  Every sum()/colSums()/rowSums() of the R text is `_rsum` here: R adds left to right in a long double and rounds once
(src/main/summary.c rsum / array.c, LDOUBLE).  That is a THIRD order beside the C++'s two -- explicit loops (left to right in
double) and Armadillo's sum() (two accumulators over even / odd elements: oracle/quilt_oracle.h) -- so the twins agree with
oracle/*.c to rounding (the tolerances of tests/golden/make_golden_rtwin.py), never bit for bit, exactly as R's twins agree
with the package's C++ in the reference's own tests (expect_equal, tolerance 1.5e-8).
"""
from __future__ import annotations

import numpy as np


def _rsum(x, axis=None):
    """R's sum() / colSums() / rowSums(): NOT numpy's pairwise tree and NOT Armadillo's two accumulators (oracle/quilt_oracle.h)
    -- R adds left to right in a long double (src/main/summary.c: rsum, LDOUBLE; src/main/array.c for colSums / rowSums) and
    rounds to double once at the end.  np.cumsum is a sequential recurrence, and np.longdouble is the x87 80-bit type on this
    platform, which is what LDOUBLE is on x86-64 Linux builds of R."""
    x = np.asarray(x)
    if x.dtype.kind in "biu":
        return x.sum(axis=axis)
    if axis is None:
        x = x.ravel(order="F")
        return np.float64(np.cumsum(x, dtype=np.longdouble)[-1]) if x.size else np.float64(0.0)
    return np.cumsum(x, axis=axis, dtype=np.longdouble).take(-1, axis=axis).astype(np.float64)


# --------------------------------------------------------------------------------------------------------------------
# shared: base qualities, haplotype expansion
# --------------------------------------------------------------------------------------------------------------------

def bq_to_probs(bq):
    """STITCH convertScaledBQtoProbs as QUILT uses it (reference-single.R:29; the same convention spelled out in
    make_eMatRead_t_using_binary, reference-single.R:451-459): column 0 = P(base | ref), column 1 = P(base | alt)."""
    import math
    bq = np.asarray(bq, dtype=np.float64)
    out = np.ones((len(bq), 2))
    # R's 10^x is the C library's pow(); numpy's vectorised power differs from it in the last bit for some qualities (22, 50, ...)
    pw = lambda x: np.array([math.pow(10.0, float(v)) for v in x])
    w = bq < 0
    eps = pw(bq[w] / 10.0)
    out[w, 0], out[w, 1] = 1 - eps, eps / 3
    w = bq > 0
    eps = pw(-bq[w] / 10.0)
    out[w, 0], out[w, 1] = eps / 3, 1 - eps
    return out


def expand_words(words, n_bits=32):
    """STITCH::int_expand: bit b of a 32-bit word = allele at the b-th SNP of the grid (LSB first)."""
    w = np.asarray(words).astype(np.int64) & 0xFFFFFFFF
    return ((w[..., None] >> np.arange(n_bits)) & 1).astype(np.int64)


def make_gl_from_u_bq(u, bq, nSNPs, minGLValue=1e-10):
    """reference-single.R:19-42 (u 0-based here), with Rcpp_make_gl_bound's rule (largest member 1, smallest >= minGLValue)."""
    gl = np.ones((2, nSNPs))
    probs = bq_to_probs(bq)
    for i in range(len(u)):
        gl[:, u[i]] = gl[:, u[i]] * probs[i]
    if minGLValue > 0:
        for t in np.nonzero(_rsum((gl < minGLValue), axis=0) > 0)[0]:
            a, b = gl[0, t], gl[1, t]
            if a > b:
                gl[0, t], gl[1, t] = 1.0, max(b / a, minGLValue)
            else:
                gl[0, t], gl[1, t] = max(a / b, minGLValue), 1.0
    return gl


# --------------------------------------------------------------------------------------------------------------------
# full-panel pass (reference-single.R:94-372)
# --------------------------------------------------------------------------------------------------------------------

def _word_prob(bits, gl_local, ref_error):
    """get_prob_for_k / build_eMatDH inner loop (reference-single.R:56-67, 76-91): product over the grid's SNPs."""
    prob = 1.0
    for b in range(gl_local.shape[1]):
        dR, dA = gl_local[0, b], gl_local[1, b]
        prob = prob * ((dR * (1 - ref_error) + dA * ref_error) if bits[b] == 0 else (dR * ref_error + dA * (1 - ref_error)))
    return prob


def _emission_column(panel, gl, g):
    """P(reads | haplotype k) at grid g for every k: through eMatDH for coded haplotypes, from rhb_t for code 0."""
    T = panel.nSNPs
    s, e = 32 * g, min(32 * (g + 1), T)
    gl_local = gl[:, s:e]
    hm = panel.hapMatcherR if panel.hapMatcherR is not None else panel.hapMatcher
    codes = np.asarray(hm[:, g]).astype(np.int64)
    tab = np.array([_word_prob(expand_words(panel.distinctHapsB[d, g], e - s), gl_local, panel.ref_error)
                    for d in range(panel.nMaxDH)])
    col = np.where(codes > 0, tab[np.maximum(codes, 1) - 1], 0.0)
    for k in np.nonzero(codes == 0)[0]:
        col[k] = _word_prob(expand_words(panel.rhb_t[k, g], e - s), gl_local, panel.ref_error)
    return col


def get_top_K_or_more_matches(alpha_col, beta_col, K_top_matches):
    """functions.R:2207-2258: running K_top largest values (ascending array), then every k at or above the smallest kept."""
    K = len(alpha_col)
    gamma = alpha_col * beta_col
    top = np.zeros(K_top_matches)
    for k in range(K):
        g = gamma[k]
        if g == top[0]:
            pass
        elif g > top[0]:
            beats = 0
            for j in range(K_top_matches):
                if g > top[j]:
                    beats = j
            if beats > 0:
                for i in range(beats):
                    top[i] = top[i + 1]
            top[beats] = g
    idx = np.nonzero(gamma >= top[0])[0]
    return idx.astype(np.int32), gamma[idx]


def R_haploid_dosage_versus_refs(panel, gl, gammaSmall_cols_to_get=None, K_top_matches=5, always_normalize=True,
                                 min_emission_prob_normalization_threshold=1e-100):
    """reference-single.R:94-372.  Returns alphaHat_t, betaHat_t (after the c factor), c, gamma_t, dosage, best_haps."""
    K, G, T = panel.K, panel.nGrids, panel.nSNPs
    tm = panel.transMatRate_t
    hm = panel.hapMatcherR if panel.hapMatcherR is not None else panel.hapMatcher
    alpha = np.zeros((K, G))
    c = np.ones(G)
    emis = [_emission_column(panel, gl, g) for g in range(G)]
    alpha[:, 0] = emis[0] * (1 / K)
    c[0] = 1 / _rsum(alpha[:, 0])
    alpha[:, 0] *= c[0]
    running = 1.0
    for g in range(1, G):
        jump_prob = tm[1, g - 1] / K
        jump_prob_plus = jump_prob if always_normalize else jump_prob * _rsum(alpha[:, g - 1])
        not_jump_prob = tm[0, g - 1]
        alpha[:, g] = (jump_prob_plus + not_jump_prob * alpha[:, g - 1]) * emis[g]
        if always_normalize:
            c[g] = 1 / _rsum(alpha[:, g])
            alpha[:, g] *= c[g]
        else:
            running *= min(1.0, emis[g].min())
            if g == G - 1 or running < min_emission_prob_normalization_threshold:
                c[g] = 1 / _rsum(alpha[:, g])
                alpha[:, g] *= c[g]
                running = 1.0
    beta = np.zeros((K, G))
    gamma = np.zeros((K, G))
    dosage = np.zeros(T)
    best = {}
    bcol = np.ones(K)
    for g in range(G - 1, -1, -1):
        if g < G - 1:
            jump_prob = tm[1, g] / K
            not_jump_prob = tm[0, g]
            e_times_b = bcol * emis[g + 1]
            bcol = not_jump_prob * e_times_b + jump_prob * _rsum(e_times_b)
        if gammaSmall_cols_to_get is not None and gammaSmall_cols_to_get[g] >= 0:
            best[int(gammaSmall_cols_to_get[g])] = get_top_K_or_more_matches(alpha[:, g], bcol, K_top_matches)
        gcol = alpha[:, g] * bcol
        s, e = 32 * g, min(32 * (g + 1), T)
        dosageL = np.zeros(e - s)
        matched = np.zeros(panel.nMaxDH)
        codes = np.asarray(hm[:, g]).astype(np.int64)
        for k in range(K):
            if codes[k] > 0:
                matched[codes[k] - 1] += gcol[k]
            else:
                bits = expand_words(panel.rhb_t[k, g], e - s).astype(np.float64)
                dosageL += gcol[k] * np.where(bits == 0, panel.ref_error, 1 - panel.ref_error)
        for b in range(e - s):
            for dh in range(panel.nMaxDH):
                dosageL[b] += matched[dh] * panel.distinctHapsIE[dh, s + b]
        dosage[s:e] = dosageL
        bcol = bcol * c[g]
        beta[:, g] = bcol
        gamma[:, g] = gcol
    n_thin = len(best)
    return dict(alphaHat_t=alpha, betaHat_t=beta, c=c, gamma_t=gamma, dosage=dosage,
                best_haps=[best[i] for i in range(n_thin)])


# --------------------------------------------------------------------------------------------------------------------
# small-panel Gibbs sampler (gibbs-nipt.R)
# --------------------------------------------------------------------------------------------------------------------

def make_eMatRead_t(panel, sample, which_haps_to_use_1based, maxDifferenceBetweenReads=1e10, Jmax=10000, rescale=True):
    """Read likelihoods against the Ks selected haplotypes, the dense form the R twin calls (rcpp_make_eMatRead_t with
    eHapsCurrent_tc = the haplotypes' allele probabilities, gibbs-nipt.R:128-146): product over the read's SNPs of
    e * pA + (1 - e) * pR with e = 1 - ref_error for an alt allele, ref_error for a ref allele; divided by the column
    maximum and floored at 1 / maxDifferenceBetweenReads; a degenerate column becomes all 1."""
    which0 = np.asarray(which_haps_to_use_1based, dtype=np.int64) - 1
    Ks, R = len(which0), sample.nReads
    bits = np.zeros((Ks, panel.nSNPs), dtype=np.int64)
    for g in range(panel.nGrids):
        s, e = 32 * g, min(32 * (g + 1), panel.nSNPs)
        bits[:, s:e] = expand_words(panel.rhb_t[which0, g], 32)[:, : e - s]
    eh = np.where(bits == 1, 1 - panel.ref_error, panel.ref_error)
    out = np.ones((Ks, R))
    probs = bq_to_probs(sample.bq)
    pR_prev, pA_prev = 1.0, 1.0
    for r in range(R):
        s, e = sample.read_ptr[r], sample.read_ptr[r + 1]
        n = min(e - s - 1, Jmax) + 1
        col = np.ones(Ks)
        for j in range(s, s + n):
            if sample.bq[j] != 0:
                pR_prev, pA_prev = probs[j, 0], probs[j, 1]
            col = col * (eh[:, sample.u[j]] * pA_prev + (1 - eh[:, sample.u[j]]) * pR_prev)
        if rescale:
            x = col.max()
            with np.errstate(divide="ignore"):
                d1 = 1 / x if x != 0 else np.inf
            if not np.isfinite(x) or x == 0 or not np.isfinite(d1):
                col[:] = 1.0
            else:
                col = np.maximum(col * d1, 1 / maxDifferenceBetweenReads)
        out[:, r] = col
    return out


def evaluate_read_variability(eMatRead_t):
    """gibbs-nipt.R:2016-2066: category 1 none below 1 - 1e-12, 2 all such entries equal, 3 fewer than floor(0.2 K), else 0."""
    K, R = eMatRead_t.shape
    cat = np.zeros(R, dtype=np.int64)
    idx = []
    thresh, thresh2 = 1 - 1e-12, int(np.floor(K * 0.20))
    for r in range(R):
        w = np.nonzero(eMatRead_t[:, r] < thresh)[0]
        idx.append(w)
        if len(w) == 0:
            cat[r] = 1
        elif np.all(eMatRead_t[w, r] == eMatRead_t[w[0], r]):
            cat[r] = 2
        elif len(w) < thresh2:
            cat[r] = 3
    return cat, idx


def make_rlc(ff):
    """gibbs-nipt.R:1960-1975"""
    p = np.array([0.5, (1 - ff) / 2, ff / 2])
    with np.errstate(invalid="ignore", divide="ignore"):
        return np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [p[0] / (p[0] + p[1]), p[1] / (p[0] + p[1]), 0],
                         [p[0] / (p[0] + p[2]), 0, p[2] / (p[0] + p[2])], [0, p[1] / (p[1] + p[2]), p[2] / (p[1] + p[2])], p])


def _forward_haploid(eMatGrid, tm, K, initialize_only=False):
    """Rcpp_run_forward_haploid with prior = alphaMat = 1 / K as the R twin calls it (gibbs-nipt.R:207-209, 1908-1919):
    alpha_1 = prior * e_1 normalised; alpha_t = e_t * (sigma alpha_{t-1} + (1 - sigma) sum(alpha_{t-1}) / K), c_t = 1 / sum."""
    G = eMatGrid.shape[1]
    alpha = np.ones((K, G)) if initialize_only else np.zeros((K, G))
    c = np.ones(G) if initialize_only else np.zeros(G)
    alpha[:, 0] = (1 / K) * eMatGrid[:, 0]
    c[0] = 1 / _rsum(alpha[:, 0])
    alpha[:, 0] *= c[0]
    if initialize_only:
        return alpha, c
    for t in range(1, G):
        alpha[:, t] = eMatGrid[:, t] * (tm[0, t - 1] * alpha[:, t - 1] + tm[1, t - 1] * _rsum(alpha[:, t - 1]) * (1 / K))
        c[t] = 1 / _rsum(alpha[:, t])
        alpha[:, t] *= c[t]
    return alpha, c


def _backward_haploid(beta, c, eMatGrid, tm, K):
    """Rcpp_run_backward_haploid: beta_t = c_t (sigma_t e_{t+1} beta_{t+1} + (1 - sigma_t) sum(e_{t+1} beta_{t+1}) / K); the
    last column is set by the caller."""
    G = eMatGrid.shape[1]
    for t in range(G - 2, -1, -1):
        e_times_b = eMatGrid[:, t + 1] * beta[:, t + 1]
        beta[:, t] = c[t] * (tm[1, t] * _rsum(e_times_b) * (1 / K) + tm[0, t] * e_times_b)
    return beta


def forwardBackwardGibbsNIPT(panel, sample, which_haps_to_use_1based, H_start, runif_reads, first_read_0based, *, ff=0.0,
                             n_gibbs_burn_in_its=20, n_gibbs_sample_its=1, gibbs_initialize_iteratively=False,
                             maxDifferenceBetweenReads=1e10, Jmax=10000, class_sum_cutoff=0.06,
                             block_gibbs_iterations=(), runif_shard=None, runif_block=None, runif_resample=None,
                             shuffle_bin_radius=5000, block_gibbs_quantile_prob=0.95, L_grid=None, trace=None):
    """gibbs-nipt.R:4-350 with gibbs_nipt_one_iteration (:508-997): the sweeps, and -- after the (0-based) sweeps listed in
    ``block_gibbs_iterations`` (gibbs-nipt.cpp:2972-2980) -- the passes whose R twins live in gibbs-nipt-block.R:
      * ff == 0 (diploid): the shard pass (R_shard_block_gibbs_resampler, one uniform per grid boundary from
        ``runif_shard[pass * (G - 1) + boundary]``).  The block pass before it never relabels a diploid sample (SURVEY.md
        Appendix A.21) and is not run;
      * ff > 0 (NIPT): block definition (R_define_blocked_snps_using_gamma_on_the_fly) and the block pass
        (R_block_gibbs_resampler, block_approach = 6, one uniform per block from ``runif_block[pass * R + block]``), followed
        by the production tail the R twin does not carry (gibbs-nipt-block.cpp:1900-1946, read for its call sequence only):
        labels re-drawn from their classes (sample_H_using_H_class with ``runif_resample[pass * R + read]``), eMatGrid and the
        forward pass rebuilt from them, then the backward pass.
    ``trace``: a list that receives one dict per pass (blocks, chosen relabellings / flip decisions).
    Returns H, H_class, the per-label alpha / beta / c / eMatGrid and hapProbs_t."""
    K = len(which_haps_to_use_1based)
    G, R = panel.nGrids, sample.nReads
    tm = panel.transMatRate_t
    n_its = n_gibbs_burn_in_its + n_gibbs_sample_its
    prior_probs = np.array([0.5, (1 - ff) / 2, ff / 2])
    rlc = make_rlc(ff)
    eMatRead_t = make_eMatRead_t(panel, sample, which_haps_to_use_1based, maxDifferenceBetweenReads, Jmax)
    read_category, non1 = evaluate_read_variability(eMatRead_t)
    H = np.asarray(H_start, dtype=np.int64).copy()
    H_class = np.zeros(R, dtype=np.int64)
    first_read = first_read_0based + 1   # the R twin counts reads from 1
    wif1 = np.asarray(sample.wif, dtype=np.int64) + 1
    eg, al, be, cc = [], [], [], []
    if gibbs_initialize_iteratively:
        for h in range(3):
            e = np.ones((K, G))
            a, c = _forward_haploid(e, tm, K, initialize_only=True)
            eg.append(e); al.append(a); be.append(np.ones((K, G))); cc.append(c)
    else:
        for h in range(3):
            e = np.ones((K, G))
            for r in range(R):   # rcpp_make_eMatGrid_t: reads of this label multiplied into their grid, in read order
                if H[r] == h + 1:
                    e[:, wif1[r] - 1] *= eMatRead_t[:, r]
            a, c = _forward_haploid(e, tm, K)
            b = np.zeros((K, G))
            b[:, G - 1] = c[G - 1]
            b = _backward_haploid(b, c, e, tm, K)
            eg.append(e); al.append(a); be.append(b); cc.append(c)

    for iteration in range(1, n_its + 1):
        iRead = 0   # 1-based index of the last processed read
        for iGrid in range(1, G + 1):
            g = iGrid - 1
            for h in range(3):
                if iGrid > 1:   # alpha_forward_one, then the previous normalisation and the new one (:567-586)
                    al[h][:, g] = eg[h][:, g] * (tm[0, g - 1] * al[h][:, g - 1] + _rsum(al[h][:, g - 1]) * tm[1, g - 1] * (1 / K))
                    al[h][:, g] *= cc[h][g]
                    a = 1 / _rsum(al[h][:, g])
                    cc[h][g] *= a
                    al[h][:, g] *= a
                else:           # rcpp_reinitialize_in_iterations
                    al[h][:, 0] = (1 / K) * eg[h][:, 0]
                    cc[h][0] = 1 / _rsum(al[h][:, 0])
                    al[h][:, 0] *= cc[h][0]
            alphaHat_m = np.stack([al[h][:, g] for h in range(3)])
            betaHat_m = np.stack([be[h][:, g] for h in range(3)])
            pC = _rsum((alphaHat_m * betaHat_m), axis=1)
            while iRead < R and wif1[iRead] == iGrid:
                iRead += 1   # now the 1-based index of the read being processed
                r = iRead - 1
                # The R twin skips uninformative (category 1) reads always (gibbs-nipt.R:617); the C++ -- production -- only for
                # diploid samples (gibbs-nipt.cpp:813-815): with ff > 0 such a read is still drawn (from the prior, its emission
                # being 1 everywhere) and classified.  The C++ rule is followed, so that NIPT calls can be compared at all.
                if read_category[r] != 1 or ff != 0:
                    normal = ginit = through = False
                    if not gibbs_initialize_iteratively:
                        normal = True
                    elif iRead < first_read and iteration == 1:
                        through = True
                    elif first_read <= iRead and iteration == 1:
                        ginit = True
                    elif iRead < first_read and iteration == 2:
                        ginit = True
                    else:
                        normal = True
                    er = eMatRead_t[:, r]
                    if normal:
                        h_rC = int(H[r])
                        h_rA1, h_rA2 = [x for x in (1, 2, 3) if x != h_rC]
                        pA1, pA2 = pC.copy(), pC.copy()
                        ab = alphaHat_m * betaHat_m
                        if read_category[r] == 0:
                            pA1[h_rC - 1] = _rsum((ab[h_rC - 1] / er))
                            pA1[h_rA1 - 1] = _rsum((ab[h_rA1 - 1] * er))
                            pA2[h_rA2 - 1] = _rsum((ab[h_rA2 - 1] * er))
                        elif read_category[r] == 1:
                            pass   # (NIPT only) nothing to add or remove: the emission is 1 for every haplotype
                        elif read_category[r] == 2:
                            w = non1[r]
                            v = er[w[-1]]
                            pA1[h_rC - 1] += _rsum(ab[h_rC - 1, w]) * (1 / v - 1)
                            pA1[h_rA1 - 1] += _rsum(ab[h_rA1 - 1, w]) * (v - 1)
                            pA2[h_rA2 - 1] += _rsum(ab[h_rA2 - 1, w]) * (v - 1)
                        else:
                            w = non1[r]
                            pA1[h_rC - 1] += _rsum((ab[h_rC - 1, w] * (1 / er[w] - 1)))
                            pA1[h_rA1 - 1] += _rsum((ab[h_rA1 - 1, w] * (er[w] - 1)))
                            pA2[h_rA2 - 1] += _rsum((ab[h_rA2 - 1, w] * (er[w] - 1)))
                        pA2[h_rC - 1] = pA1[h_rC - 1]
                    elif ginit:
                        h_rC, h_rA1, h_rA2 = 1, 2, 3
                        pA1, pA2 = pC.copy(), pC.copy()
                        ab = alphaHat_m * betaHat_m
                        pC[0] = _rsum((ab[0] * er))
                        pA1[1] = _rsum((ab[1] * er))
                        pA2[2] = _rsum((ab[2] * er))
                    else:
                        h_rC, h_rA1, h_rA2 = 1, 2, 3
                        pA1, pA2 = pC.copy(), pC.copy()
                    prod_pC = np.prod(pC) * prior_probs[h_rC - 1]
                    prod_pA1 = np.prod(pA1) * prior_probs[h_rA1 - 1]
                    prod_pA2 = np.prod(pA2) * prior_probs[h_rA2 - 1]
                    denom = prod_pC + prod_pA1 + prod_pA2
                    norm = {h_rC: prod_pC / denom, h_rA1: prod_pA1 / denom, h_rA2: prod_pA2 / denom}
                    cum = np.cumsum([norm[1], norm[2], norm[3]])
                    chance = runif_reads[R * (iteration - 1) + iRead - 1]
                    h_rN = 0
                    for i in (3, 2, 1):
                        if chance < cum[i - 1]:
                            h_rN = i
                    if h_rN == 0:
                        raise RuntimeError("bad h_rN")
                    if (h_rN != h_rC or ginit) and not through:
                        H[r] = h_rN
                        if normal:
                            alphaHat_m[h_rC - 1] = alphaHat_m[h_rC - 1] / er
                            eg[h_rC - 1][:, g] = eg[h_rC - 1][:, g] / er
                        alphaHat_m[h_rN - 1] = alphaHat_m[h_rN - 1] * er
                        eg[h_rN - 1][:, g] = eg[h_rN - 1][:, g] * er
                        if normal:
                            pC = (pA1 if h_rN == h_rA1 else pA2).copy()
                        elif ginit:
                            if h_rN == 2:
                                pC = pA1.copy()
                            if h_rN == 3:
                                pC = pA2.copy()
                    x = np.array([norm[1], norm[2], norm[3]])
                    y = _rsum(np.abs(rlc - x[None, :]), axis=1)
                    with np.errstate(invalid="ignore"):
                        m = np.nanmin(y)
                    H_class[r] = (int(np.nanargmin(y)) + 1) if m < class_sum_cutoff else 0
            for h in range(3):   # inject back and renormalise (:903-916)
                al[h][:, g] = alphaHat_m[h]
                a = 1 / _rsum(al[h][:, g])
                cc[h][g] *= a
                al[h][:, g] *= a
        for h in range(3):
            be[h][:, G - 1] = cc[h][G - 1]
            be[h] = _backward_haploid(be[h], cc[h], eg[h], tm, K)
        passes = [i for i, b in enumerate(block_gibbs_iterations) if b == iteration - 1]
        for i_pass in passes:
            if ff == 0:
                H, info = R_shard_block_gibbs_resampler(al, be, cc, eg, H, wif1 - 1, tm, K,
                                                        np.asarray(runif_shard)[i_pass * (G - 1):(i_pass + 1) * (G - 1)])
            else:
                Lg = np.asarray(panel.L_grid if L_grid is None else L_grid)
                grid = np.arange(panel.nSNPs) // 32
                blocked = R_define_blocked_snps_using_gamma_on_the_fly(al, be, cc, eg, tm, shuffle_bin_radius, Lg, grid,
                                                                       block_gibbs_quantile_prob, ff)
                H, H_class, info = R_block_gibbs_resampler(al, be, cc, eg, H, H_class, eMatRead_t, blocked["blocked_snps"],
                                                           np.asarray(runif_block)[i_pass * R:(i_pass + 1) * R], grid,
                                                           wif1 - 1, ff, tm, K)
                info["blocked_grid"] = blocked["blocked_grid"]
                info["available_rules_agree"] = blocked["available_rules_agree"]
                # the production tail (not in the R twin): H from H_class, eMatGrid from H, forward, backward
                H = sample_H_using_H_class(H_class, ff, np.asarray(runif_resample)[i_pass * R:(i_pass + 1) * R])
                for h in range(3):
                    e = np.ones((K, G))
                    for r in range(R):
                        if H[r] == h + 1:
                            e[:, wif1[r] - 1] *= eMatRead_t[:, r]
                    eg[h] = e
                    al[h], cc[h] = _forward_haploid(e, tm, K)
                    b = np.ones((K, G))
                    b[:, G - 1] = cc[h][G - 1]
                    be[h] = _backward_haploid(b, cc[h], e, tm, K)
            if trace is not None:
                trace.append(info)

    # hapProbs from the dense haplotypes: gamma_h = alpha_h beta_h / c_h (save_various_gammas with the neutral label
    # probabilities); hapProbs[h, t] = sum_k gamma_h[k, grid(t)] * P(alt | haplotype k at t)
    which0 = np.asarray(which_haps_to_use_1based, dtype=np.int64) - 1
    hap = np.zeros((3, panel.nSNPs))
    for g in range(G):
        s, e = 32 * g, min(32 * (g + 1), panel.nSNPs)
        bits = expand_words(panel.rhb_t[which0, g], 32)[:, : e - s]
        eh = np.where(bits == 1, 1 - panel.ref_error, panel.ref_error)
        for h in range(3):
            gam = al[h][:, g] * be[h][:, g] / cc[h][g]
            hap[h, s:e] = gam @ eh
    return dict(H=H.astype(np.int32), H_class=H_class.astype(np.int32), alphaHat_t=al, betaHat_t=be, c=cc, eMatGrid_t=eg,
                hapProbs_t=hap, eMatRead_t=eMatRead_t, read_category=read_category)


# --------------------------------------------------------------------------------------------------------------------
# block definition, block pass and shard pass (gibbs-nipt-block.R)
# --------------------------------------------------------------------------------------------------------------------

def make_smoothed_rate(sigma_rate, L_grid, shuffle_bin_radius):
    """rcpp_make_smoothed_rate (copied-from-stitch.cpp:446-518; the R twin calls the C++): base-pair weighted mean of the rate
    over +- shuffle_bin_radius bp around the midpoint of every grid boundary."""
    nGrids = len(L_grid)
    L = [int(x) for x in L_grid]
    out = np.zeros(nGrids - 1)
    for iGrid in range(nGrids - 1):
        focal_point = (L[iGrid] + L[iGrid + 1]) // 2
        acc, total = 0.0, 0.0
        iL, rem, prev = iGrid, shuffle_bin_radius, focal_point
        while 0 < rem and 0 <= iL:
            add = prev - L[iL]
            if rem - add < 0:
                add, rem = rem, 0
            else:
                rem -= add
            acc = acc + add * sigma_rate[iL]
            total += add
            prev = L[iL]
            iL -= 1
        iR, rem, prev = iGrid + 1, shuffle_bin_radius, focal_point
        while 0 < rem and iR < nGrids:
            add = L[iR] - prev
            if rem - add < 0:
                add, rem = rem, 0
            else:
                rem -= add
            acc = acc + add * sigma_rate[iR - 1]
            total += add
            prev = L[iR]
            iR += 1
        out[iGrid] = acc / total
    return out


def determine_where_to_stop(smoothed_rate, available, snp_best, thresh, nGrids, is_left):
    """rcpp_determine_where_to_stop (copied-from-stitch.cpp:522-567), 0-based."""
    mult = 1 if is_left else -1
    snp_consider = snp_best
    val_prev = smoothed_rate[snp_best]
    snp_min, val_min = snp_consider, smoothed_rate[snp_consider]
    c = 1
    while True:
        snp_consider = snp_consider - mult
        val_cur = smoothed_rate[snp_consider]
        if 5 <= c:
            val_prev = smoothed_rate[snp_consider + 5 * mult]
        c += 1
        if val_cur < val_min:
            snp_min, val_min = snp_consider, val_cur
        if snp_consider <= 2 or nGrids - 3 <= snp_consider:
            break
        if not available[snp_consider - mult]:
            break
        if 3 * val_min < val_cur:
            break
        if val_cur < thresh and val_prev < val_cur:
            break
    return snp_min


def simple_quantile(x, q):
    """rcpp_simple_quantile (gibbs-nipt-block.cpp:81-85)."""
    x = np.asarray(x)
    return x[np.argsort(x, kind="stable")[int(len(x) * q)]]


def R_define_blocked_snps_using_gamma_on_the_fly(al, be, cc, eg, tm, shuffle_bin_radius, L_grid, grid,
                                                 block_gibbs_quantile_prob, ff):
    """gibbs-nipt-block.R:978-1149.  Two places where the R text and the C++ (gibbs-nipt-block.cpp:397-447) differ are taken
    from the C++ -- production runs the C++: (i) a boundary above the threshold stays available even when its rate is below
    0.01 (the R clears it), (ii) the three-wide window around a peak is clamped to the last boundary (the R indexes one
    past the end there).  ``available_rules_agree`` says whether (i) made a difference on this input."""
    nGrids = len(cc[0])
    diff2 = np.zeros((3, nGrids - 1))
    for iGrid in range(nGrids - 2):
        for h in range(3 if ff != 0 else 2):
            diff2[h, iGrid] = 1 - _rsum((tm[0, iGrid] * (al[h][:, iGrid] * be[h][:, iGrid + 1] * eg[h][:, iGrid + 1])))
    rate2 = _rsum(diff2, axis=0)
    smoothed_rate = make_smoothed_rate(rate2, L_grid, shuffle_bin_radius)
    break_thresh = 1.0
    d = simple_quantile(smoothed_rate, block_gibbs_quantile_prob)
    if d < break_thresh:
        break_thresh = d
    available_R = smoothed_rate > break_thresh
    available_R = available_R & ~(smoothed_rate < 0.01) & ~np.isnan(smoothed_rate)
    available = smoothed_rate > break_thresh          # the C++ rule (see the docstring)
    agree = bool(np.array_equal(available, available_R))
    blocked_grid = np.zeros(nGrids, dtype=np.int64)
    if _rsum(available) == 0:
        return dict(blocked_snps=np.zeros(len(grid), dtype=np.int64), blocked_grid=blocked_grid, smoothed_rate=smoothed_rate,
                    break_thresh=break_thresh, available_rules_agree=agree)
    nAvailable = int(_rsum(available))
    best = np.argsort(-smoothed_rate, kind="stable")[:nAvailable]   # R's order(decreasing = TRUE) keeps ties in place
    available = available.copy()
    to_keep = []
    for snp_best in best:
        snp_best = int(snp_best)
        if available[snp_best]:
            a = max(snp_best - 1, 0)
            b = min(snp_best + 1, nGrids - 2)
            if int(_rsum(available[a:b + 1])) == 3:
                snp_left = determine_where_to_stop(smoothed_rate, available, snp_best, break_thresh, nGrids, True)
                snp_right = determine_where_to_stop(smoothed_rate, available, snp_best, break_thresh, nGrids, False)
                available[snp_left:snp_right + 1] = False
            else:
                available[a:b + 1] = False
            to_keep.append(snp_best + 1)   # the boundary between (0-based) grids snp_best and snp_best + 1: a block starts at the latter
    if min(to_keep) != 0:
        to_keep.append(0)
    if max(to_keep) != nGrids - 1:
        to_keep.append(nGrids - 1)
    blocks_to_consider = sorted(to_keep)
    for i in range(len(blocks_to_consider) - 1):
        blocked_grid[blocks_to_consider[i]:blocks_to_consider[i + 1] + 1] = i
    blocked_snps = blocked_grid[np.asarray(grid)]
    return dict(blocked_snps=blocked_snps, blocked_grid=blocked_grid, smoothed_rate=smoothed_rate, break_thresh=break_thresh,
                available_rules_agree=agree)


def R_make_gibbs_considers(blocked_snps, grid, wif0, nGrids):
    """gibbs-nipt-block.R:2658-2862, line by line (its 1-based loop variables kept; arrays 0-based here)."""
    bs = [int(x) for x in blocked_snps]
    n_blocks = max(bs) + 1
    nSNPs, nReads = len(bs), len(wif0)
    snp_start, snp_end = [0] * n_blocks, [0] * n_blocks
    iBlock, start = 1, 0
    for iSNP in range(1, nSNPs + 1):
        record = iSNP == nSNPs or bs[iSNP - 1] < bs[iSNP]
        if record:
            snp_start[iBlock - 1] = start
            snp_end[iBlock - 1] = iSNP - 1
            start = iSNP
            iBlock += 1
    grid_start = [int(grid[snp_start[i]]) for i in range(n_blocks)]
    grid_end = [int(grid[snp_end[i]]) for i in range(n_blocks)]
    blocked_grid = [0] * nGrids
    for iBlock in range(1, n_blocks + 1):
        for i in range(grid_start[iBlock - 1], grid_end[iBlock - 1] + 1):
            blocked_grid[i] = iBlock - 1
    reads_start, reads_end = [-1] * n_blocks, [-1] * n_blocks
    previous_block_first_iRead = 1
    previous_block = blocked_grid[int(wif0[0])] + 1
    for this_iRead in range(2, nReads + 1):
        this_grid = int(wif0[this_iRead - 1]) + 1
        this_block = blocked_grid[this_grid - 1] + 1
        if this_iRead == nReads:
            reads_start[this_block - 1] = previous_block_first_iRead - 1
            reads_end[this_block - 1] = this_iRead - 1
        elif previous_block < this_block:
            reads_start[previous_block - 1] = previous_block_first_iRead - 1
            reads_end[previous_block - 1] = (this_iRead - 1) - 1
            previous_block_first_iRead = this_iRead
            previous_block = blocked_grid[int(wif0[this_iRead - 1])] + 1
    remove = [x == -1 for x in reads_start]
    w = [i + 1 for i, x in enumerate(remove) if x]   # 1-based block numbers without reads
    if len(w) > 0:
        jBefore = 1
        for jNow in range(1, len(w) + 1):
            if jNow == len(w):
                todo = True
            elif w[jNow] - w[jNow - 1] == 1:
                todo = False
                jBefore -= 1
            else:
                todo = True
            if todo:
                s1, e1 = w[jBefore - 1], w[jNow - 1]
                x = int(np.ceil(0.5 * (grid_start[s1 - 1] + grid_end[e1 - 1])))
                y = int(np.ceil(0.5 * (snp_start[s1 - 1] + snp_end[e1 - 1])))
                if s1 == 1:
                    s1, x, y = 2, 0, 0
                if e1 == n_blocks:
                    e1 = e1 - 1
                    x, y = grid_end[n_blocks - 1], snp_end[n_blocks - 1]
                grid_start[e1] = x          # [e1 + 1] in the R
                grid_end[s1 - 2] = x - 1    # [s1 - 1]
                snp_start[e1] = y
                snp_end[s1 - 2] = y - 1
                jBefore = jNow
            jBefore += 1
        keep = [i for i in range(n_blocks) if not remove[i]]
        reads_start = [reads_start[i] for i in keep]
        reads_end = [reads_end[i] for i in keep]
        grid_start = [grid_start[i] for i in keep]
        grid_end = [grid_end[i] for i in keep]
        snp_start = [snp_start[i] for i in keep]
        snp_end = [snp_end[i] for i in keep]
    n_blocks = len(snp_end)
    where = [-1] * nGrids
    for i in range(1, n_blocks + 1):
        where[grid_end[i - 1]] = i - 1
    return dict(reads_start=reads_start, reads_end=reads_end, grid_start=grid_start, grid_end=grid_end, snp_start=snp_start,
                snp_end=snp_end, grid_where=where, n_blocks=n_blocks)


def get_log_p_H_class2(n1, n2, n3, n4, n5, n6, ff):
    """gibbs-nipt-block.R:3629-3660"""
    with np.errstate(divide="ignore"):
        if ff == 0:
            return (n1 * np.log(1 / 2) + n2 * np.log(1 / 2 - ff / 2) + n3 * np.log(1e-3) + n4 * np.log(1 - ff / 2) +
                    n5 * np.log(1 / 2 + ff / 2) + n6 * np.log(1 / 2))
        if ff == 1:
            return (n1 * np.log(1 / 2) + n2 * np.log(1e-3) + n3 * np.log(ff / 2) + n4 * np.log(1 - ff / 2) +
                    n5 * np.log(1 / 2 + ff / 2) + n6 * np.log(1 / 2))
        return (n1 * np.log(1 / 2) + n2 * np.log(1 / 2 - ff / 2) + n3 * np.log(ff / 2) + n4 * np.log(1 - ff / 2) +
                n5 * np.log(1 / 2 + ff / 2) + n6 * np.log(1 / 2))


_RR = np.array([[1, 2, 3], [1, 3, 2], [2, 1, 3], [2, 3, 1], [3, 1, 2], [3, 2, 1]])
_RX = np.array([[1, 2, 3], [1, 3, 2], [2, 1, 3], [3, 1, 2], [2, 3, 1], [3, 2, 1]])


def _block_read_label_probabilities_using_H_class(read_start, read_end, H_class, ff):
    """calculate_block_read_label_probabilities_using_H_class (gibbs-nipt-block.R:1921-1949)."""
    ns = np.zeros(8)
    for iRead in range(read_start, read_end + 1):
        ns[H_class[iRead]] += 1
    n = ns[1:7]
    out = np.zeros(6)
    for ir in range(6):
        r1, r2, r3 = _RR[ir]
        out[ir] = get_log_p_H_class2(n[r1 - 1], n[r2 - 1], n[r3 - 1], n[7 - r3 - 1], n[7 - r2 - 1], n[7 - r1 - 1], ff)
    return out


def R_block_gibbs_resampler(al, be, cc, eg, H, H_class, eMatRead_t, blocked_snps, runif_block, grid, wif0, ff, tm, K):
    """gibbs-nipt-block.R:401-973 with block_approach = 6, consider_total_relabelling = FALSE (gibbs-nipt.cpp:239-240): the
    forward recursion of all six relabellings (R_gibbs_block_forward_one :2512-2543), at every block end the choice
    (R_consider_block_relabelling :1350-1451), the rebuild of the block under the chosen relabelling (:1537-1659), the reset
    (R_reset_local_variables :1973-1985), finally the backward pass (:919-945).  In place on al / be / cc / eg; returns the new H,
    H_class and the chosen relabellings."""
    nGrids, nReads = len(cc[0]), len(wif0)
    H = np.asarray(H, dtype=np.int64).copy()
    H_class = np.asarray(H_class, dtype=np.int64).copy()
    con = R_make_gibbs_considers(blocked_snps, grid, wif0, nGrids)
    n_blocks = con["n_blocks"]
    assert con["grid_start"][0] == 0 and con["grid_end"][n_blocks - 1] == nGrids - 1
    for i in range(n_blocks - 1):
        assert con["grid_start"][i + 1] - con["grid_end"][i] == 1, "bad making of consider grid"
    prior_probs = np.array([0.5, (1 - ff) / 2, ff / 2])
    logC_before = np.zeros(3)
    with np.errstate(divide="ignore"):
        logC_after = np.array([_rsum(np.log(cc[h])) for h in range(3)])
    ever_changed = 0
    alphaStore = np.zeros((K, 3, 6))
    log_cStore = np.zeros((nGrids, 3, 6))
    chosen = []
    for iGrid in range(1, nGrids + 1):
        g = iGrid - 1
        eLocal = np.stack([eg[h][:, g] for h in range(3)], axis=1)
        # R_gibbs_block_forward_one (approach 6: nothing but the forward step of the six relabellings)
        for ir in range(6):
            for i in range(3):
                h = _RR[ir, i] - 1
                if iGrid == 1:
                    alphaStore[:, h, ir] = (1 / K) * eLocal[:, i]
                else:
                    alphaStore[:, h, ir] = eLocal[:, i] * (tm[0, g - 1] * alphaStore[:, h, ir] + tm[1, g - 1] * (1 / K))
                d = 1 / _rsum(alphaStore[:, h, ir])
                log_cStore[g, h, ir] = np.log(d)
                alphaStore[:, h, ir] = d * alphaStore[:, h, ir]
        if con["grid_where"][g] > -1:
            iBlock = con["grid_where"][g]
            gs, ge = con["grid_start"][iBlock], con["grid_end"][iBlock]
            rs, re = con["reads_start"][iBlock], con["reads_end"][iBlock]
            # ---- R_consider_block_relabelling
            betaLocal = np.stack([be[h][:, g] for h in range(3)], axis=1)
            P = np.zeros(6)
            with np.errstate(divide="ignore", invalid="ignore"):
                for ir in range(6):
                    for i in range(3):
                        logC_inside = 0.0
                        for g2 in range(gs, ge + 1):
                            logC_inside = logC_inside + log_cStore[g2, i, ir]
                        P[ir] = P[ir] + (np.log(_rsum((alphaStore[:, i, ir] * betaLocal[:, i]))) + -logC_before[i] + -logC_inside +
                                         -logC_after[i])
                Hterm = _block_read_label_probabilities_using_H_class(rs, re, H_class, ff)
                clp = P + Hterm
                clp = clp - np.max(clp)
                clp[clp < -100] = -100
                probs = np.exp(clp)
                probs[np.isnan(probs)] = 0
                if ff == 0:
                    probs[[1, 3, 4, 5]] = 0
                probs = probs / _rsum(probs)
            chance = runif_block[iBlock]
            cum = np.cumsum(probs)
            ir_chosen = 0
            for i in range(6, 0, -1):
                if chance < cum[i - 1]:
                    ir_chosen = i
            rx = _RX[ir_chosen - 1]
            one_based_swap = np.array([1, 1 + rx[0], 1 + rx[1], 1 + rx[2], 8 - rx[2], 8 - rx[1], 8 - rx[0], 8])
            chosen.append(ir_chosen)
            if ever_changed == 1 or ir_chosen != 1:
                ever_changed = 1
                iRead = rs + 1
                wif_read = wif0[iRead - 1]
                for iGrid2 in range(gs + 1, ge + 2):
                    eLocal2 = np.ones((K, 3))
                    while iRead <= nReads and wif_read < iGrid2 - 1:
                        iRead += 1
                        if iRead <= nReads:
                            wif_read = wif0[iRead - 1]
                    while iRead <= nReads and wif_read == iGrid2 - 1:
                        h = one_based_swap[H[iRead - 1]] - 1
                        eLocal2[:, h - 1] = eLocal2[:, h - 1] * eMatRead_t[:, iRead - 1]
                        iRead += 1
                        if iRead <= nReads:
                            wif_read = wif0[iRead - 1]
                    for h in range(3):
                        eg[h][:, iGrid2 - 1] = eLocal2[:, h]
                        if iGrid2 == 1:
                            al[h][:, 0] = (1 / K) * eg[h][:, 0]
                        else:
                            al[h][:, iGrid2 - 1] = eg[h][:, iGrid2 - 1] * (tm[0, iGrid2 - 2] * al[h][:, iGrid2 - 2] +
                                                                             tm[1, iGrid2 - 2] * (1 / K))
                        cc[h][iGrid2 - 1] = 1 / _rsum(al[h][:, iGrid2 - 1])
                        al[h][:, iGrid2 - 1] = cc[h][iGrid2 - 1] * al[h][:, iGrid2 - 1]
                for iRead0 in range(rs, re + 1):
                    H_class[iRead0] = one_based_swap[H_class[iRead0]] - 1
                    H[iRead0] = one_based_swap[H[iRead0]] - 1
            # ---- reset for the next block (R_reset_local_variables), unless this was the last one
            if iBlock + 2 <= n_blocks:
                for ir in range(6):
                    for i in range(3):
                        alphaStore[:, i, ir] = al[i][:, g]
                        log_cStore[g, i, ir] = np.log(cc[i][g])
            for g2 in range(gs, ge + 1):
                for i in range(3):
                    logC_before[i] = logC_before[i] + np.log(cc[i][g2])
        for i in range(3):
            logC_after[i] = logC_after[i] - np.log(cc[i][g])
    for h in range(3):
        be[h][:, nGrids - 1] = cc[h][nGrids - 1]
        be[h] = _backward_haploid(be[h], cc[h], eg[h], tm, K)
    return H, H_class, dict(kind="block", ir_chosen=np.array(chosen, dtype=np.int32), grid_start=np.array(con["grid_start"]),
                            grid_end=np.array(con["grid_end"]), reads_start=np.array(con["reads_start"]),
                            reads_end=np.array(con["reads_end"]))


def sample_H_using_H_class(H_class, ff, u):
    """sample_H_using_H_class (gibbs-nipt-block.R:3715-3723): classes 1..3 are their label; 0 / 7 draw from the prior, 4 / 5 / 6
    from the prior restricted to the class's two labels.  R's sample(x, 1, prob = p) (ProbSampleNoReplace, which Rcpp's sugar
    reproduces): p / sum(p) sorted in decreasing order, the first element whose cumulative mass reaches the uniform.  One
    uniform per read (``u[read]``), used only by the reads that draw."""
    p07 = np.array([1 / 2, 1 / 2 - ff / 2, ff / 2])
    table = {0: p07, 7: p07, 4: np.array([1 / 2, 1 / 2 - ff / 2, 0]), 5: np.array([1 / 2, 0, ff / 2]),
             6: np.array([0, 1 / 2 - ff / 2, ff / 2])}
    H = np.zeros(len(H_class), dtype=np.int64)
    for r, hc in enumerate(H_class):
        hc = int(hc)
        if hc in (1, 2, 3):
            H[r] = hc
            continue
        p = table[hc] / _rsum(table[hc])
        order = np.argsort(-p, kind="stable")
        mass, pick = 0.0, order[-1]
        for j in order:
            mass += p[j]
            if u[r] <= mass:
                pick = j
                break
        H[r] = pick + 1
    return H


def R_shard_block_gibbs_resampler(al, be, cc, eg, H, wif0, tm, K, runif):
    """gibbs-nipt-block.R:2874-3581 for ff == 0 with shard_check_every_pair = TRUE (quilt.R:178): a left-to-right forward pass;
    after every grid but the last, stay versus "swap the two labels from here on" is weighed from alpha_i beta_j cross products
    (:3280-3298) and drawn with ``runif[grid]`` (the R draws runif(1) there); in flip mode the following grids' eMatGrid columns
    swap and their reads change label.  Then the backward pass.  In place on al / be / cc / eg (labels 1, 2); returns H and the
    flip decisions."""
    nGrids, nReads = len(cc[0]), len(wif0)
    H = np.asarray(H, dtype=np.int64).copy()
    minus_log_c_sum = [0.0, 0.0]
    original_c = [cc[0].copy(), cc[1].copy()]
    in_flip_mode = False
    iRead = 0
    flips, p_stay = [], []
    for iGrid in range(nGrids):
        if iGrid == 0:
            for h in range(2):
                al[h][:, 0] = (1 / K) * eg[h][:, 0]
                cc[h][0] = 1 / _rsum(al[h][:, 0])
                al[h][:, 0] = al[h][:, 0] * cc[h][0]
                minus_log_c_sum[h] = minus_log_c_sum[h] - np.log(cc[h][0])
        else:
            if in_flip_mode:
                x = eg[0][:, iGrid].copy()
                eg[0][:, iGrid] = eg[1][:, iGrid]
                eg[1][:, iGrid] = x
            for h in range(2):   # alpha_forward_one, the previous normalisation, the new one (:3143-3159)
                al[h][:, iGrid] = eg[h][:, iGrid] * (tm[0, iGrid - 1] * al[h][:, iGrid - 1] +
                                                      _rsum(al[h][:, iGrid - 1]) * tm[1, iGrid - 1] * (1 / K))
                al[h][:, iGrid] = al[h][:, iGrid] * cc[h][iGrid]
                a = 1 / _rsum(al[h][:, iGrid])
                cc[h][iGrid] = cc[h][iGrid] * a
                al[h][:, iGrid] = al[h][:, iGrid] * a
                minus_log_c_sum[h] = minus_log_c_sum[h] - np.log(cc[h][iGrid])
        while iRead <= nReads - 1 and wif0[iRead] == iGrid:
            if in_flip_mode:
                H[iRead] = 3 - H[iRead]
            iRead += 1
        if iGrid < nGrids - 1:
            w1 = slice(0, iGrid + 1)
            w2 = slice(iGrid, nGrids)
            mlo = [-_rsum(np.log(original_c[h][w2])) for h in range(2)]   # (the R carries these as running sums: the same terms)
            pA1 = minus_log_c_sum[0] + mlo[0] + np.log(_rsum((al[0][:, iGrid] * be[0][:, iGrid])))
            pA2 = -_rsum(np.log(cc[1][w1])) - _rsum(np.log(original_c[1][w2])) + np.log(_rsum((al[1][:, iGrid] * be[1][:, iGrid])))
            pB1 = -_rsum(np.log(cc[1][w1])) - _rsum(np.log(original_c[0][w2])) + np.log(_rsum((al[1][:, iGrid] * be[0][:, iGrid])))
            pB2 = -_rsum(np.log(cc[0][w1])) - _rsum(np.log(original_c[1][w2])) + np.log(_rsum((al[0][:, iGrid] * be[1][:, iGrid])))
            calculated_difference = (pB1 + pB2) - (pA1 + pA2)
            probs = np.array([1.0, np.exp(calculated_difference)])
            probs = probs / _rsum(probs)
            in_flip_mode = bool(runif[iGrid] > probs[0])
            flips.append(in_flip_mode)
            p_stay.append(probs[0])
    for h in range(2):
        be[h][:, nGrids - 1] = cc[h][nGrids - 1]
        be[h] = _backward_haploid(be[h], cc[h], eg[h], tm, K)
    return H, dict(kind="shard", flip_mode=np.array(flips, dtype=np.int32), p_stay=np.array(p_stay))

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

Two independent readings of two different reference sources agreeing -- on every output the two parameterisations share
(dosage, gamma, best-haplotype lists, read labels under the same uniforms, hapProbs, sum(log c) up to the known sigma
terms) -- is the strongest pin available in a container without R.  tests/golden/make_golden_rtwin.py
generates the committed fixtures from THIS file and cross-checks oracle/*.c against it.

Plain numpy, loops over grids and reads (small cases only).  Never imported by the product.
"""
from __future__ import annotations

import numpy as np


# --------------------------------------------------------------------------------------------------------------------
# shared: base qualities, haplotype expansion
# --------------------------------------------------------------------------------------------------------------------

def bq_to_probs(bq):
    """STITCH convertScaledBQtoProbs as QUILT uses it (reference-single.R:29; the same convention spelled out in
    make_eMatRead_t_using_binary, reference-single.R:451-459): column 0 = P(base | ref), column 1 = P(base | alt)."""
    bq = np.asarray(bq, dtype=np.float64)
    out = np.ones((len(bq), 2))
    w = bq < 0
    eps = 10.0 ** (bq[w] / 10.0)
    out[w, 0], out[w, 1] = 1 - eps, eps / 3
    w = bq > 0
    eps = 10.0 ** (-bq[w] / 10.0)
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
        for t in np.nonzero((gl < minGLValue).sum(axis=0) > 0)[0]:
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
    c[0] = 1 / alpha[:, 0].sum()
    alpha[:, 0] *= c[0]
    running = 1.0
    for g in range(1, G):
        jump_prob = tm[1, g - 1] / K
        jump_prob_plus = jump_prob if always_normalize else jump_prob * alpha[:, g - 1].sum()
        not_jump_prob = tm[0, g - 1]
        alpha[:, g] = (jump_prob_plus + not_jump_prob * alpha[:, g - 1]) * emis[g]
        if always_normalize:
            c[g] = 1 / alpha[:, g].sum()
            alpha[:, g] *= c[g]
        else:
            running *= min(1.0, emis[g].min())
            if g == G - 1 or running < min_emission_prob_normalization_threshold:
                c[g] = 1 / alpha[:, g].sum()
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
            bcol = not_jump_prob * e_times_b + jump_prob * e_times_b.sum()
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
    c[0] = 1 / alpha[:, 0].sum()
    alpha[:, 0] *= c[0]
    if initialize_only:
        return alpha, c
    for t in range(1, G):
        alpha[:, t] = eMatGrid[:, t] * (tm[0, t - 1] * alpha[:, t - 1] + tm[1, t - 1] * alpha[:, t - 1].sum() * (1 / K))
        c[t] = 1 / alpha[:, t].sum()
        alpha[:, t] *= c[t]
    return alpha, c


def _backward_haploid(beta, c, eMatGrid, tm, K):
    """Rcpp_run_backward_haploid: beta_t = c_t (sigma_t e_{t+1} beta_{t+1} + (1 - sigma_t) sum(e_{t+1} beta_{t+1}) / K); the
    last column is set by the caller."""
    G = eMatGrid.shape[1]
    for t in range(G - 2, -1, -1):
        e_times_b = eMatGrid[:, t + 1] * beta[:, t + 1]
        beta[:, t] = c[t] * (tm[1, t] * e_times_b.sum() * (1 / K) + tm[0, t] * e_times_b)
    return beta


def forwardBackwardGibbsNIPT(panel, sample, which_haps_to_use_1based, H_start, runif_reads, first_read_0based, *, ff=0.0,
                             n_gibbs_burn_in_its=20, n_gibbs_sample_its=1, gibbs_initialize_iteratively=False,
                             maxDifferenceBetweenReads=1e10, Jmax=10000, class_sum_cutoff=0.06):
    """gibbs-nipt.R:4-350 with gibbs_nipt_one_iteration (:508-997): the sweeps WITHOUT block / shard passes (the R twins of
    those live in gibbs-nipt-block.R).  Returns H, H_class, the per-label alpha / beta / c / eMatGrid and hapProbs_t."""
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
                    al[h][:, g] = eg[h][:, g] * (tm[0, g - 1] * al[h][:, g - 1] + al[h][:, g - 1].sum() * tm[1, g - 1] * (1 / K))
                    al[h][:, g] *= cc[h][g]
                    a = 1 / al[h][:, g].sum()
                    cc[h][g] *= a
                    al[h][:, g] *= a
                else:           # rcpp_reinitialize_in_iterations
                    al[h][:, 0] = (1 / K) * eg[h][:, 0]
                    cc[h][0] = 1 / al[h][:, 0].sum()
                    al[h][:, 0] *= cc[h][0]
            alphaHat_m = np.stack([al[h][:, g] for h in range(3)])
            betaHat_m = np.stack([be[h][:, g] for h in range(3)])
            pC = (alphaHat_m * betaHat_m).sum(axis=1)
            while iRead < R and wif1[iRead] == iGrid:
                iRead += 1   # now the 1-based index of the read being processed
                r = iRead - 1
                if read_category[r] != 1:
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
                            pA1[h_rC - 1] = (ab[h_rC - 1] / er).sum()
                            pA1[h_rA1 - 1] = (ab[h_rA1 - 1] * er).sum()
                            pA2[h_rA2 - 1] = (ab[h_rA2 - 1] * er).sum()
                        elif read_category[r] == 2:
                            w = non1[r]
                            v = er[w[-1]]
                            pA1[h_rC - 1] += ab[h_rC - 1, w].sum() * (1 / v - 1)
                            pA1[h_rA1 - 1] += ab[h_rA1 - 1, w].sum() * (v - 1)
                            pA2[h_rA2 - 1] += ab[h_rA2 - 1, w].sum() * (v - 1)
                        else:
                            w = non1[r]
                            pA1[h_rC - 1] += (ab[h_rC - 1, w] * (1 / er[w] - 1)).sum()
                            pA1[h_rA1 - 1] += (ab[h_rA1 - 1, w] * (er[w] - 1)).sum()
                            pA2[h_rA2 - 1] += (ab[h_rA2 - 1, w] * (er[w] - 1)).sum()
                        pA2[h_rC - 1] = pA1[h_rC - 1]
                    elif ginit:
                        h_rC, h_rA1, h_rA2 = 1, 2, 3
                        pA1, pA2 = pC.copy(), pC.copy()
                        ab = alphaHat_m * betaHat_m
                        pC[0] = (ab[0] * er).sum()
                        pA1[1] = (ab[1] * er).sum()
                        pA2[2] = (ab[2] * er).sum()
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
                    y = np.abs(rlc - x[None, :]).sum(axis=1)
                    with np.errstate(invalid="ignore"):
                        m = np.nanmin(y)
                    H_class[r] = (int(np.nanargmin(y)) + 1) if m < class_sum_cutoff else 0
            for h in range(3):   # inject back and renormalise (:903-916)
                al[h][:, g] = alphaHat_m[h]
                a = 1 / al[h][:, g].sum()
                cc[h][g] *= a
                al[h][:, g] *= a
        for h in range(3):
            be[h][:, G - 1] = cc[h][G - 1]
            be[h] = _backward_haploid(be[h], cc[h], eg[h], tm, K)

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

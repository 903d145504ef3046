// gibbs3.hip -- the read-label sampler with three labels: NIPT mode (fetal fraction ff > 0: maternal transmitted,
// maternal untransmitted, paternal transmitted) of `rcpp_forwardBackwardGibbsNIPT` (QUILT/src/gibbs-nipt.cpp:2395-3307).
//
// Same decomposition as the diploid kernel of gibbs.hip -- one workgroup of NW wavefronts per chain, the current grid's
// columns of every label in registers, DPP reductions, fp64, no FMA contraction -- written for the general label logic
// of `sample_reads_in_grid` (:733-1295): three candidate moves per read (stay, the lower of the two other labels, the
// higher), move probabilities prod_h p(h) * prior(label) with prior = (0.5, (1 - ff) / 2, ff / 2), the 7-prototype read
// class (record_read_set, :1142-1165).  It is the straightforward form (no software pipelining of the column loads yet).
//
// A call is run as segments of sweeps [it_begin, it_end) with a block-Gibbs pass between them (k_block_rate3, host block
// definition in gibbs_blocks.hpp, k_block3 -- below); the state lives in HBM across the launches.  Category 1 reads are
// not skipped in this mode (:815) but leave every probability unchanged.
#include "gibbs_dev.hpp"

#include <type_traits>

namespace {

// (two waves per chain must also mean two waves per SIMD -- 256 registers each -- or 1024 chains would not be resident
// together: amdgpu_waves_per_eu)
// LEAN: the build for TWO chains per SIMD at ten rows per lane (256 registers per wave), as the two-label kernel's (gibbs.hip):
// the grid steps of a sweep run at what HBM delivers and the read visits at a wave's serial latency, so two chains on a SIMD
// overlap the one with the other.  What makes it fit: eMatGrid's three columns wait in LDS while a grid's reads are sampled
// (15 KB per chain, 120 KB per compute unit), alpha * beta is formed in beta's registers, the next grid's columns are fetched
// at the end of the grid instead of a grid ahead, and the class prototypes live in scalar registers.
template <int NE, int NW, bool LEAN = false>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(LEAN ? 2 : NW))) void k_gibbs3(GibbsParams p) {
    __shared__ double s_red[2 * NW * 4];
    __shared__ double s_e[LEAN ? 3 * NE * 64 * NW : 1];
    const int c = blockIdx.x, t = threadIdx.x;
    using CH = Chain<NE, NW>;
    CH ch(p, c, t, s_red);
    constexpr int NT = CH::NT;
    constexpr int NH = 3;
    const int G = ch.G, Ksp = ch.Ksp, R = ch.R, Ks = ch.Ks;
    const double prior = ch.prior;
    bool (&valid)[NE] = ch.valid;
    const double *runif = p.seed_reads ? nullptr : p.runif_reads + (size_t)p.read_off[c] * p.n_its;
    const uint64_t seed_reads = p.seed_reads ? p.seed_reads[c] : 0;
    const int first_read = p.first_read[c];
    const bool init_iteratively = p.init_iteratively && first_read >= 0;
    const double one_over_K = 1 / (double)Ks;

    auto sum3 = [&](const Col<NE> (&x)[NH], double (&s)[NH]) {
#pragma unroll
        for (int h = 0; h < NH; h++) {
            s[h] = 0;
#pragma unroll
            for (int i = 0; i < NE; i++) s[h] += x[h].v[i];
        }
        ch.template bsum<NH>(s);
    };
    // wave-uniform scalars straight from memory (this kernel does not use the lane-held streams)
    auto uni_d = [](const double *q) { return rl_f64(*q, 0); };
    auto uni_i = [](int v) { return __builtin_amdgcn_readfirstlane(v); };

    // prior over labels and the read-label class prototypes (gibbs-nipt.cpp:2707-2729) of this chain's fetal fraction
    // (wave-uniform values computed on the vector unit come back through v_readlane: they then live in scalar registers for
    // the whole kernel instead of 50 vector registers)
    auto sc = [](double v) { return rl_f64(v, 0); };
    const double ffc = p.ff_chain ? uni_d(&p.ff_chain[c]) : p.ff;
    const double pp[3] = {0.5, sc((1 - ffc) * 0.5), sc(ffc * 0.5)};
    const double rlc[7][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1},
                              {sc(pp[0] / (pp[0] + pp[1])), sc(pp[1] / (pp[0] + pp[1])), 0},
                              {sc(pp[0] / (pp[0] + pp[2])), 0, sc(pp[2] / (pp[0] + pp[2]))},
                              {0, sc(pp[1] / (pp[1] + pp[2])), sc(pp[2] / (pp[1] + pp[2]))},
                              {pp[0], pp[1], pp[2]}};
    int status = 0;
    if (p.it_begin > 0) {   // a later segment of the call: the state is in HBM, a chain that underflowed stays stopped
        status = uni_i(p.status[c]);
    } else {
    // ---- c = 0 (arma::zeros, :2676-2678), H_class = 0, eMatGrid = 1
    for (int h = 0; h < 3; h++)
        for (int g = t; g < G; g += NT) ch.cv[h][g] = 0.0;
    for (int r = t; r < R; r += NT) ch.Hc[r] = 0;
    {
        Col<NE> one;
#pragma unroll
        for (int i = 0; i < NE; i++) one.v[i] = 1.0;
        for (int h = 0; h < NH; h++)
            for (int g = 0; g < G; g++) ch.st(one, ch.eg[h] + (size_t)g * Ksp);
    }
    }   // first segment
    chain_sync<NW>();

    // `masked`: only the Ks real rows of a column move (Chain::ldm / stm) -- for every call but a first segment's first one, which
    // is what gives the padding rows of beta their zeros
    auto backward_full = [&](bool faster, bool masked) {
        // per-grid scalars from lane-held streams, the next grid's eMatGrid columns fetched while this one computes
        // (as backward_both of the two-label kernel, gibbs.hip)
        Col<NE> b[NH], e[NH];
        GridStreams3<CH> gs;
        gs.load_bwd(ch, (G - 1) & ~63);
#pragma unroll
        for (int h = 0; h < NH; h++) {
            const double cl = gs.c_of(h, (G - 1) & 63);
#pragma unroll
            for (int i = 0; i < NE; i++) b[h].v[i] = valid[i] ? cl : 0.0;
            if (masked) ch.stm(b[h], ch.beta[h] + (size_t)(G - 1) * Ksp);
            else ch.st(b[h], ch.beta[h] + (size_t)(G - 1) * Ksp);
        }
        if (G >= 2) {
#pragma unroll
            for (int h = 0; h < NH; h++) ch.ldm(e[h], ch.eg[h] + (size_t)(G - 1) * Ksp);
        }
        for (int g = G - 2; g >= 0; --g) {
            if ((g & 63) == 63) gs.load_bwd(ch, g & ~63);
            const int j = g & 63;
            Col<NE> en[NH];   // the next iteration's emission columns (grid g)
#pragma unroll
            for (int h = 0; h < NH; h++) ch.ldm(en[h], ch.eg[h] + (size_t)g * Ksp);
            const double s0 = rl_f64(gs.t0, j), s1 = rl_f64(gs.t1, j);
            const bool has = !faster || rl_i32(gs.has, j) != 0;
            double x[NH];
#pragma unroll
            for (int h = 0; h < NH; h++) {
                x[h] = 0;
#pragma unroll
                for (int i = 0; i < NE; i++) {
                    if (has) b[h].v[i] = e[h].v[i] * b[h].v[i];
                    if (faster) x[h] += b[h].v[i];
                    else x[h] += valid[i] ? prior * b[h].v[i] : 0.0;
                }
            }
            ch.template bsum<NH>(x);
#pragma unroll
            for (int h = 0; h < NH; h++) {
                const double cg = gs.c_of(h, j);
                const double xx = faster ? s1 * x[h] * one_over_K : s1 * x[h];
#pragma unroll
                for (int i = 0; i < NE; i++) b[h].v[i] = valid[i] ? cg * (xx + s0 * b[h].v[i]) : 0.0;
                if (masked) ch.stm(b[h], ch.beta[h] + (size_t)g * Ksp);
                else ch.st(b[h], ch.beta[h] + (size_t)g * Ksp);
            }
#pragma unroll
            for (int h = 0; h < NH; h++) e[h] = en[h];
        }
        chain_sync<NW>();
    };

    // rcpp_make_eMatGrid_t (copied-from-stitch.cpp:262-281) from the labels in memory and Rcpp_run_forward_haploid (:340-387) in
    // ONE pass over the grids: a grid's eMatGrid columns go from the reads (sorted by grid; their scalars come from lane-held
    // streams, the next read's compact emission is fetched a read ahead) straight into the forward step -- and to memory when the
    // grid has reads: the others hold 1 since the call began -- instead of being written by one pass and read back by the next.  alphaHat_t is written only when no sweep follows in this launch (a
    // sweep carries alpha in registers from grid 0 on and the segment's last one writes it).
    auto build_forward = [&](bool store_alpha) {
        ReadStreams<CH> rs;
        rs.load(ch, 0, nullptr, 0);
        typename CH::ErPre pre{};
        if (R > 0) ch.ld_pre(pre, 0);
        int r = 0;
        Col<NE> a[NH];
        GridStreams3<CH> gs;
        for (int g = 0; g < G; g++) {
            if ((g & 63) == 0) {
                if (g) gs.store_c(ch);
                gs.load_fwd(ch, g);
            }
            const int jg = g & 63;
            Col<NE> e[NH];
#pragma unroll
            for (int h = 0; h < NH; h++)
#pragma unroll
                for (int i = 0; i < NE; i++) e[h].v[i] = 1.0;
            bool any = false;
            while (r < R) {
                if (r >= rs.base + 64) rs.load(ch, r, nullptr, 0);
                const int j = r - rs.base;
                if (rl_i32(rs.wif, j) != g) break;
                const typename CH::ErPre x = pre;
                ch.ld_pre(pre, min(r + 1, R - 1));
                Col<NE> er;
                ch.read_emission(er, x, rl_i32(rs.dn, j));
                const int hh = rl_i32(rs.H, j) - 1;
#pragma unroll
                for (int h = 0; h < NH; h++)
                    if (hh == h) {
#pragma unroll
                        for (int i = 0; i < NE; i++) e[h].v[i] *= er.v[i];
                    }
                any = true;
                r++;
            }
            if (any) {
#pragma unroll
                for (int h = 0; h < NH; h++) ch.st(e[h], ch.eg[h] + (size_t)g * Ksp);
            }
            const double s0 = rl_f64(gs.t0, jg), s1 = rl_f64(gs.t1, jg);
#pragma unroll
            for (int h = 0; h < NH; h++) {
#pragma unroll
                for (int i = 0; i < NE; i++) {
                    if (g == 0) a[h].v[i] = valid[i] ? prior * e[h].v[i] : 0.0;
                    else a[h].v[i] = valid[i] ? e[h].v[i] * (s0 * a[h].v[i] + s1 * prior) : 0.0;
                }
            }
            double sm[NH], cc[NH];
            sum3(a, sm);
#pragma unroll
            for (int h = 0; h < NH; h++) {
                cc[h] = 1 / sm[h];
#pragma unroll
                for (int i = 0; i < NE; i++) a[h].v[i] = a[h].v[i] * cc[h];
                if (store_alpha) ch.st(a[h], ch.alpha[h] + (size_t)g * Ksp);
            }
            gs.set_c(ch.lane, jg, cc[0], cc[1], cc[2]);
        }
        gs.store_c(ch);
        chain_sync<NW>();
    };
    const bool sweeps_follow = p.it_begin < p.it_end;

    // ---- rcpp_gibbs_nipt_initialize (:1629-1750)
    if (p.it_begin > 0) {
        // a later segment: nothing to initialise -- unless a block pass came before it, which leaves new labels behind and
        // eMatGrid, alpha, beta, c to be formed from them (Rcpp_block_gibbs_resampler's last steps, gibbs-nipt-block.cpp:1898-1954:
        // eMatGrid, Rcpp_run_forward_haploid, Rcpp_run_backward_haploid_QUILT_faster from beta(G - 1) = c(G - 1)); done here, one
        // wave per chain with the pipelined passes, instead of at the end of k_block3 (two waves, a barrier per sum)
        if (p.rebuild && status == 0) {
            build_forward(!sweeps_follow);
            backward_full(true, true);
        }
    } else if (!init_iteratively) {
        build_forward(true);   // (the call's first pass over alpha: it also gives the padding rows their zeros)
        backward_full(false, false);   // rcpp_initialize_gibbs_forward_backward (:453-487)
    } else {
        // alpha = beta = 1, c = 1, then only column 0 of alpha is initialised (:1725-1740)
        Col<NE> one;
#pragma unroll
        for (int i = 0; i < NE; i++) one.v[i] = valid[i] ? 1.0 : 0.0;
        for (int h = 0; h < NH; h++) {
            for (int g = 0; g < G; g++) {
                ch.st(one, ch.alpha[h] + (size_t)g * Ksp);
                ch.st(one, ch.beta[h] + (size_t)g * Ksp);
            }
            for (int g = t; g < G; g += NT) ch.cv[h][g] = 1.0;
        }
        chain_sync<NW>();
        for (int h = 0; h < NH; h++) {
            Col<NE> a;
#pragma unroll
            for (int i = 0; i < NE; i++) a.v[i] = valid[i] ? prior * 1.0 : 0.0;
            const double cc = 1 / ch.sum_col(a);
#pragma unroll
            for (int i = 0; i < NE; i++) a.v[i] = a.v[i] * cc;
            ch.st(a, ch.alpha[h]);
            if (t == 0) ch.cv[h][0] = cc;
        }
    }
    chain_sync<NW>();

    for (int it = p.it_begin; it < p.it_end && status == 0; it++) {
        // ================= rcpp_gibbs_nipt_iterate (:1756-1956) =================
        Col<NE> a[NH];
        int iRead = 0;
        // the reads' scalars (grid, label, class, category, uniform) as lane-held streams: one vector load per 64 reads
        // instead of a dependent uniform load per read and field (gibbs_dev.hpp)
        ReadStreams<CH> rs;
        bool rs_dirty = false;
        auto rs_load = [&](int base) {
            rs.load(ch, base, runif, it);
            if (!runif) rs.u = stream_uniform(seed_reads, (uint64_t)R * it + base + ch.lane);
        };
        rs_load(0);
        // the next read's compact emission (pattern bytes + table entry) is fetched one read ahead, unconditionally
        // (clamped index): its latency hides behind the current read's arithmetic
        typename CH::ErPre pre{};
        if (R > 0) ch.ld_pre(pre, 0);
        // per-grid scalars (transition, c, grid_has_read) as lane-held streams and the next grid's eMatGrid / beta columns
        // fetched a grid ahead, as in the two-label kernel: a uniform scalar loaded through the vector path would have to
        // be waited for before use, which (in-order vmcnt) would also drain the column prefetches
        GridStreams3<CH> gs;
        Col<NE> e[NH], bt[NH];
        // (state columns in the sweeps: only the Ks real rows move; the padding rows keep what the first segment's initialisation
        // stored -- 0 for alpha and beta, 1 for eMatGrid -- and read as 0 here: every use is masked by `valid` or multiplies a 0)
#pragma unroll
        for (int h = 0; h < NH; h++) {
            ch.ldm(e[h], ch.eg[h]);
            ch.ldm(bt[h], ch.beta[h]);
        }
        for (int g = 0; g < G; g++) {
            if ((g & 63) == 0) {
                if (g) gs.store_c(ch);
                gs.load_fwd(ch, g);
            }
            const int jg = g & 63;
            double cg[NH];
            Col<NE> en[LEAN ? 1 : NH];   // (LEAN: the next grid's columns are fetched at the end of this one, into e's own registers)
            const size_t gn = (size_t)min(g + 1, G - 1) * Ksp;   // clamped: the loads stay unconditional
            if constexpr (!LEAN) {
#pragma unroll
                for (int h = 0; h < NH; h++) ch.ldm(en[h], ch.eg[h] + gn);
            }
            if (g > 0) {
                // rcpp_alpha_forward_one_QUILT_faster (:671-707), normalize = true
                const double x = rl_f64(gs.t0, jg), t1 = rl_f64(gs.t1, jg);
                const bool has = rl_i32(gs.has, jg) != 0;
                double sp[NH];
                sum3(a, sp);
#pragma unroll
                for (int h = 0; h < NH; h++) {
                    const double alphaConst = t1 * sp[h];
#pragma unroll
                    for (int i = 0; i < NE; i++) {
                        const double inner = (x * a[h].v[i] + alphaConst * one_over_K);
                        a[h].v[i] = valid[i] ? (has ? e[h].v[i] * inner : inner) : 0.0;
                    }
                }
                double sn[NH];
                sum3(a, sn);
#pragma unroll
                for (int h = 0; h < NH; h++) {
                    const double c2 = gs.c_of(h, jg);
                    double aa = 1 / (c2 * sn[h]);
                    cg[h] = c2 * aa;
                    aa *= c2;
#pragma unroll
                    for (int i = 0; i < NE; i++) a[h].v[i] *= aa;
                }
            } else {
                // rcpp_reinitialize_in_iterations (:712-727)
#pragma unroll
                for (int h = 0; h < NH; h++)
#pragma unroll
                    for (int i = 0; i < NE; i++) a[h].v[i] = valid[i] ? prior * e[h].v[i] : 0.0;
                double sn[NH];
                sum3(a, sn);
#pragma unroll
                for (int h = 0; h < NH; h++) {
                    cg[h] = 1 / sn[h];
#pragma unroll
                    for (int i = 0; i < NE; i++) a[h].v[i] *= cg[h];
                }
            }
            // alpha * beta of the grid (the reference's ab_m), formed once the forward step is done; beta's registers then
            // take the next grid's columns (LEAN: the product is formed IN beta's registers, the next columns come at the end
            // of the grid; eMatGrid's columns go to LDS for the time of the read loop when the grid has reads)
            Col<NE> ab_own[LEAN ? 1 : NH];
            auto &ab = [&]() -> Col<NE> (&)[NH] { if constexpr (LEAN) return bt; else return ab_own; }();
#pragma unroll
            for (int h = 0; h < NH; h++) {
#pragma unroll
                for (int i = 0; i < NE; i++) ab[h].v[i] = a[h].v[i] * bt[h].v[i];
            }
            if constexpr (!LEAN) {
#pragma unroll
                for (int h = 0; h < NH; h++) ch.ldm(bt[h], ch.beta[h] + gn);
            }
            // ---- sample_reads_in_grid (:733-1295), three labels
            bool grid_started = false, changed = false;
            int moved = 0;   // bit h: a move changed eMatGrid's column of label h (only those go back to memory)
            double pC[3] = {1, 1, 1};
            bool normal = false, ginit = false, pass = false;
            while (iRead < R) {
                if (iRead >= rs.base + 64) {
                    if (rs_dirty) { rs.store(ch); rs_dirty = false; }
                    rs_load(iRead);
                }
                const int j = iRead - rs.base;
                if (rl_i32(rs.wif, j) != g) break;
                const int r = iRead;
                iRead++;
                // (not diploid: reads of every category are visited, :815)
                if (!init_iteratively) normal = true;
                else if (r < first_read && it == 0) pass = true;
                else if (first_read <= r && it == 0) { pass = false; ginit = true; }
                else if (r < first_read && it == 1) { pass = false; ginit = true; }
                else { ginit = false; normal = true; }
                if (!grid_started) {
                    sum3(ab, pC);
                    grid_started = true;
                    if constexpr (LEAN) {   // the grid has reads: its eMatGrid columns wait in LDS from here on
#pragma unroll
                        for (int h = 0; h < NH; h++)
#pragma unroll
                            for (int i = 0; i < NE; i++) s_e[(h * NE + i) * NT + t] = e[h].v[i];
                    }
                }
                Col<NE> er, rer;   // the read's emission column and its reciprocal (x * (1 / e) for the reference's x / e,
                {                  // as in the two-label kernel: <= 1 ulp apart)
                    const typename CH::ErPre x = pre;
                    ch.ld_pre(pre, min(r + 1, R - 1));
                    const int dn_r = rl_i32(rs.dn, j);
                    if (dn_r >= 0) {
                        ch.ld(er, ch.eMatRead + (size_t)dn_r * Ksp);
#pragma unroll
                        for (int i = 0; i < NE; i++) rer.v[i] = fast_rcp(er.v[i]);
                    } else {
                        ch.expand_with_rcp(er, rer, x);   // reciprocal once per table entry, gathered like the emission
                    }
                }
                // The three candidate labels in compile-time positions: one straight-line version of everything from the sums to
                // the draw per current label, behind a wave-uniform branch, instead of a select per element and label in the sums
                // (6 NE v_cndmask per read) and per probability after them (~40); the asm comments keep the compiler from folding
                // the versions back into one block of selects (as in the two-label kernel, gibbs.hip).
                int h_rC = 0, h_rA1 = 1, h_rN = 0;
                double pA1[3] = {pC[0], pC[1], pC[2]}, pA2[3] = {pC[0], pC[1], pC[2]};
                double x3[3];
                const double chance = rl_f64(rs.u, j);
                // products, normalisation and the draw (:998-1076) with the current label hc, the lower other label a1, the higher a2
                auto draw = [&](auto HC) {
                    constexpr int hc = decltype(HC)::value, a1 = hc == 0 ? 1 : 0, a2 = hc == 2 ? 1 : 2;
                    const double prod_pC = (pC[0] * pC[1] * pC[2]) * pp[hc];
                    const double prod_pA1 = (pA1[0] * pA1[1] * pA1[2]) * pp[a1];
                    const double prod_pA2 = (pA2[0] * pA2[1] * pA2[2]) * pp[a2];
                    const double denom = prod_pC + prod_pA1 + prod_pA2;
                    const double rden = fast_rcp(denom);   // (x * (1 / d) for the reference's x / d, as in the two-label kernel)
                    x3[hc] = prod_pC * rden; x3[a1] = prod_pA1 * rden; x3[a2] = prod_pA2 * rden;
                    const double cs0 = x3[0], cs1 = x3[1] + cs0, cs2 = x3[2] + cs1;
                    h_rN = 0;
                    if (chance < cs2) h_rN = 2;
                    if (chance < cs1) h_rN = 1;
                    if (chance < cs0) h_rN = 0;
                };
                // a read in normal progress whose current label is hc: pA1 = the read leaves hc for a1, pA2 = for a2 (:905-960);
                // dense form for categories 0, 2 and 3 (the sparse updates of 2 / 3 are the same sums), category 1 changes no sum
                auto normal_read = [&](auto HC) {
                    constexpr int hc = decltype(HC)::value, a1 = hc == 0 ? 1 : 0, a2 = hc == 2 ? 1 : 2;
                    h_rA1 = a1;
                    if (rl_i32(rs.cat1, j) == 0) {
                        double s[3] = {0, 0, 0};
#pragma unroll
                        for (int i = 0; i < NE; i++) {
                            s[0] += ab[0].v[i] * (hc == 0 ? rer.v[i] : er.v[i]);
                            s[1] += ab[1].v[i] * (hc == 1 ? rer.v[i] : er.v[i]);
                            s[2] += ab[2].v[i] * (hc == 2 ? rer.v[i] : er.v[i]);
                        }
                        ch.template bsum<3>(s);
                        pA1[hc] = s[hc]; pA1[a1] = s[a1];
                        pA2[hc] = s[hc]; pA2[a2] = s[a2];
                    }
                    draw(HC);
                };
                if (normal) {
                    h_rC = rl_i32(rs.H, j) - 1;
                    if (h_rC == 0) {
                        asm volatile("; current label 0" ::: "memory");
                        normal_read(std::integral_constant<int, 0>{});
                        asm volatile("; current label 0 done" ::: "memory");
                    } else if (h_rC == 1) {
                        asm volatile("; current label 1" ::: "memory");
                        normal_read(std::integral_constant<int, 1>{});
                        asm volatile("; current label 1 done" ::: "memory");
                    } else {
                        asm volatile("; current label 2" ::: "memory");
                        normal_read(std::integral_constant<int, 2>{});
                        asm volatile("; current label 2 done" ::: "memory");
                    }
                } else {
                    if (ginit) {
                        double s[3] = {0, 0, 0};
#pragma unroll
                        for (int h = 0; h < NH; h++)
#pragma unroll
                            for (int i = 0; i < NE; i++) s[h] += ab[h].v[i] * er.v[i];
                        ch.template bsum<3>(s);
                        pC[0] = s[0];
                        pA1[1] = s[1];
                        pA2[2] = s[2];
                    }
                    draw(std::integral_constant<int, 0>{});
                }
                if (((h_rN != h_rC) || ginit) && !pass) {
                    changed = true;
                    if (ch.lane == j) rs.H = h_rN + 1;
                    rs_dirty = true;
                    auto mul_e = [&](int h, const Col<NE> &f) {   // eMatGrid's column: in registers, or (LEAN) where it waits in LDS
                        if constexpr (LEAN) {
#pragma unroll
                            for (int i = 0; i < NE; i++) s_e[(h * NE + i) * NT + t] *= f.v[i];
                        } else {
#pragma unroll
                            for (int i = 0; i < NE; i++) e[h].v[i] *= f.v[i];
                        }
                    };
#pragma unroll
                    for (int h = 0; h < NH; h++) {
                        if (normal && h == h_rC) {
#pragma unroll
                            for (int i = 0; i < NE; i++) { a[h].v[i] *= rer.v[i]; ab[h].v[i] *= rer.v[i]; }
                            mul_e(h, rer);
                            moved |= 1 << h;
                        }
                    }
#pragma unroll
                    for (int h = 0; h < NH; h++) {
                        if (h == h_rN) {
#pragma unroll
                            for (int i = 0; i < NE; i++) { a[h].v[i] *= er.v[i]; ab[h].v[i] *= er.v[i]; }
                            mul_e(h, er);
                            moved |= 1 << h;
                        }
                    }
                    if (normal) {
                        // the A1 move goes to the lower of the two other labels, A2 to the higher (:1103-1117)
                        const bool to_a1 = h_rN == h_rA1;
                        for (int i = 0; i < 3; i++) pC[i] = to_a1 ? pA1[i] : pA2[i];
                    } else if (ginit) {
                        if (h_rN == 1) for (int i = 0; i < 3; i++) pC[i] = pA1[i];
                        if (h_rN == 2) for (int i = 0; i < 3; i++) pC[i] = pA2[i];
                    }
                }
                if (it == p.it_end - 1) {
                    // record_read_set (:1142-1165): H_class is overwritten for every read in every sweep and read only by
                    // the block pass after a segment or by the caller, so only a segment's last sweep computes it
                    double local_min = 2;
                    int which = 8;
#pragma unroll
                    for (int i = 0; i < 7; i++) {
                        const double y = fabs(rlc[i][0] - x3[0]) + fabs(rlc[i][1] - x3[1]) + fabs(rlc[i][2] - x3[2]);
                        if (y < local_min) { local_min = y; which = i; }
                    }
                    if (ch.lane == j) rs.Hc = (local_min < p.class_sum_cutoff) ? which + 1 : 0;
                    rs_dirty = true;
                }
            }
            if constexpr (LEAN) {   // the next grid's eMatGrid columns: e's registers are free (this grid's wait in LDS or are dead)
#pragma unroll
                for (int h = 0; h < NH; h++) ch.ldm(e[h], ch.eg[h] + gn);
            }
            if (changed) {
                // re-inject the moved columns and renormalise (:1262-1292)
                double sm[NH];
                sum3(a, sm);
#pragma unroll
                for (int h = 0; h < NH; h++) {
                    const double alphaConst = 1 / sm[h];
                    cg[h] *= alphaConst;
#pragma unroll
                    for (int i = 0; i < NE; i++) a[h].v[i] = a[h].v[i] * alphaConst;
                    if ((moved >> h) & 1) {
                        if constexpr (LEAN) {   // from LDS to memory through the registers of alpha * beta, dead by now
#pragma unroll
                            for (int i = 0; i < NE; i++) ab[h].v[i] = s_e[(h * NE + i) * NT + t];
                            ch.stm(ab[h], ch.eg[h] + (size_t)g * Ksp);
                        } else {
                            ch.stm(e[h], ch.eg[h] + (size_t)g * Ksp);
                        }
                    }
                }
            }
            // alphaHat_t is not read inside a segment of sweeps (every sweep carries alpha in registers from grid 0 on); what
            // follows a segment -- the block pass's rate, hapProbs -- sees the state its last sweep leaves: only that one writes it
            if (it == p.it_end - 1) {
#pragma unroll
                for (int h = 0; h < NH; h++) ch.stm(a[h], ch.alpha[h] + (size_t)g * Ksp);
            }
            if constexpr (LEAN) {
#pragma unroll
                for (int h = 0; h < NH; h++) ch.ldm(bt[h], ch.beta[h] + gn);
            } else {
#pragma unroll
                for (int h = 0; h < NH; h++) e[h] = en[h];
            }
            gs.set_c(ch.lane, jg, cg[0], cg[1], cg[2]);
        }
        gs.store_c(ch);
        if (rs_dirty) rs.store(ch);
        chain_sync<NW>();
        backward_full(true, true);
        // ---- underflow check (:2959-2969): with ff != 0 the third label's c is not looked at
        {
            double s[2] = {0, 0};
            for (int g = t; g < G; g += NT) { s[0] += ch.cv[0][g]; s[1] += ch.cv[1][g]; }
            ch.template bsum<2>(s);
            if (!isfinite(s[0]) || !isfinite(s[1])) status = 1;
        }
        if (p.per_it) {
            // add_to_per_it_likelihoods (:1583-1621): -sum(log c_h) and the number of reads per label after this sweep
            double s[6] = {0, 0, 0, 0, 0, 0};
            for (int g = t; g < G; g += NT) { s[0] -= log(ch.cv[0][g]); s[1] -= log(ch.cv[1][g]); s[2] -= log(ch.cv[2][g]); }
            for (int r = t; r < R; r += NT) { const int h = ch.H[r]; s[3] += h == 1 ? 1.0 : 0.0; s[4] += h == 2 ? 1.0 : 0.0; s[5] += h == 3 ? 1.0 : 0.0; }
            double s2[2];
            s2[0] = s[0]; s2[1] = s[1]; ch.template bsum<2>(s2); s[0] = s2[0]; s[1] = s2[1];
            s2[0] = s[2]; s2[1] = s[3]; ch.template bsum<2>(s2); s[2] = s2[0]; s[3] = s2[1];
            s2[0] = s[4]; s2[1] = s[5]; ch.template bsum<2>(s2); s[4] = s2[0]; s[5] = s2[1];
            if (t == 0) {
                double *o = p.per_it + ((size_t)c * p.n_its + it) * 8;
                for (int i = 0; i < 6; i++) o[i] = s[i];
                o[6] = o[7] = 0;
            }
        }
    }
    if (t == 0) p.status[c] = status;
}

// ---------------------------------------------------------------------------------------------
// NIPT block Gibbs (gibbs-nipt-block.cpp), between two segments of sweeps.
//   k_block_rate3   rate2 of Rcpp_define_blocked_snps_using_gamma_on_the_fly (:347-363): per grid boundary
//                   sum over labels of 1 - sigma * sum_k alpha(k, g) beta(k, g + 1) eMatGrid(k, g + 1).
//                   The smoothing / quantile / peak picking that turns it into blocks, and make_gibbs_considers, are
//                   scalar integer logic per chain and run on the host (gibbs.hip), as they run in R-facing C++.
//   k_block3        Rcpp_block_gibbs_resampler (:1636-1967), block_approach = 6: the forward recursion of all six
//                   relabellings (18 columns in registers), at every block end the choice among them
//                   (Rcpp_consider_block_relabelling, :590-949), the rebuild of the block under the chosen relabelling,
//                   then labels re-drawn from their classes (rcpp_sample_H_using_H_class, :213-246), eMatGrid, forward
//                   and backward redone.  One workgroup per chain, like the sampler.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_block_rate3(GibbsParams p) {
    const int c = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + wave;
    const int G = p.G, Ksp = p.Ksp, Ks = p.Ks;
    if (g >= G - 1) return;
    double out = 0;
    if (g < G - 2) {
        const size_t mat = (size_t)G * Ksp;
        const double d = p.sigma[g];
        for (int h = 0; h < p.nH; h++) {
            const double *a = p.alpha + ((size_t)c * p.nH + h) * mat + (size_t)g * Ksp;
            const double *b = p.beta + ((size_t)c * p.nH + h) * mat + (size_t)(g + 1) * Ksp;
            const double *e = p.eg + ((size_t)c * p.nH + h) * mat + (size_t)(g + 1) * Ksp;
            double s = 0;
            for (int k = lane; k < Ks; k += 64) s += a[k] * b[k] * e[k];
            s = wsum(s);
            out += 1 - d * s;
        }
    }
    if (lane == 0) p.blk_rate2[(size_t)c * G + g] = out;
}

__device__ __forceinline__ void block_sync() {
    __threadfence_block();
    __syncthreads();
}

// Rcpp::sample(1:3, 1, FALSE, probs) from its one uniform (Rcpp sugar: mass normalised, decreasing order, first
// cumulative mass >= u)
__device__ inline int sample3(double p0, double p1, double p2, double u);
// a read's new label from its class (gibbs-nipt-block.cpp:226-243): classes 1-3 fix it, the others draw
__device__ inline int resample_label(int hc, double ff, double u) {
    if (hc >= 1 && hc <= 3) return hc;
    if (hc == 0 || hc == 7) return sample3(0.5, 0.5 - ff * 0.5, ff * 0.5, u);
    if (hc == 4) return sample3(0.5, 0.5 - 0.5 * ff, 0, u);
    if (hc == 5) return sample3(0.5, 0, 0.5 * ff, u);
    return sample3(0, 0.5 - ff * 0.5, ff * 0.5, u);
}
__device__ inline int sample3(double p0, double p1, double p2, double u) {
    double p[3] = {p0, p1, p2};
    int perm[3] = {1, 2, 3};
    const double sum = p[0] + p[1] + p[2];
    for (int i = 0; i < 3; i++) p[i] /= sum;
    for (int i = 1; i < 3; i++) {
        const double v = p[i];
        const int q = perm[i];
        int j = i - 1;
        while (j >= 0 && p[j] < v) { p[j + 1] = p[j]; perm[j + 1] = perm[j]; j--; }
        p[j + 1] = v; perm[j + 1] = q;
    }
    p[1] += p[0];
    if (u <= p[0]) return perm[0];
    if (u <= p[1]) return perm[1];
    return perm[2];
}

__device__ inline double log_p_H_class2(const int (&n)[6], double ff) {   // rcpp_get_log_p_H_class2 (:170-207), 0 < ff < 1
    return 0 + n[0] * log(0.5) + n[1] * log(0.5 - ff * 0.5) + n[2] * log(ff * 0.5) + n[3] * log(1 - ff * 0.5) +
           n[4] * log(1 * 0.5 + ff * 0.5) + n[5] * log(1 * 0.5);
}

template <int NE, int NW>
__global__ __launch_bounds__(64 * NW) void k_block3(GibbsParams p) {
    __shared__ double s_red[2 * NW * 4];
    const int c = blockIdx.x, t = threadIdx.x;
    using CH = Chain<NE, NW>;
    CH ch(p, c, t, s_red);
    constexpr int NT = CH::NT;
    constexpr int NH = 3;
    constexpr int RR[6][3] = {{1, 2, 3}, {1, 3, 2}, {2, 1, 3}, {2, 3, 1}, {3, 1, 2}, {3, 2, 1}};   // :1755-1761
    constexpr int RX[6][3] = {{1, 2, 3}, {1, 3, 2}, {2, 1, 3}, {3, 1, 2}, {2, 3, 1}, {3, 2, 1}};   // :752-758
    const int G = ch.G, Ksp = ch.Ksp, R = ch.R, Ks = ch.Ks;
    const double prior = ch.prior, one_over_K = 1 / (double)Ks;
    const double ff = p.ff_chain ? rl_f64(p.ff_chain[c], 0) : p.ff;
    bool (&valid)[NE] = ch.valid;
    if (__builtin_amdgcn_readfirstlane(p.status[c]) != 0) return;   // the chain underflowed: the caller retries it
    auto uni_d = [](const double *q) { return rl_f64(*q, 0); };
    auto uni_i = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    const int32_t *where = p.blk_where + (size_t)c * G;
    const int32_t *tab = p.blk_tab + (size_t)c * 4 * G;
    const int n_blocks = uni_i(p.blk_n[c]);
    // the pass's uniforms: explicit (runif_shard doubles as [n_pass][2][R]: block choice, then label re-draw) or streams
    const double *ru = p.seed_shard ? nullptr : p.runif_shard + ((size_t)p.read_off[c] * p.blk_n_pass + (size_t)p.blk_pass * R) * 2;
    const uint64_t seed = p.seed_shard ? p.seed_shard[c] : 0;
    auto u_block = [&](int i) { return ru ? ru[i] : stream_uniform(seed, (uint64_t)p.blk_pass * 2 * R + i); };
    auto u_draw = [&](int r) { return ru ? ru[R + r] : stream_uniform(seed, (uint64_t)p.blk_pass * 2 * R + R + r); };
    // The block kernel runs two waves per chain at Ksp = 640 (its 18 columns need it) even when the sampler, and with it
    // the compact read emissions, were laid out for one wave of 10 rows per lane: then row l + 64 (w + 2 i) of this thread
    // (lane l, wave w) is byte w + 2 i of lane l's 12-byte pack.
    const bool foreign = NW == 2 && p.er_nt == 64;
    // grid, label and dense row of the reads being walked, 64 at a time in lanes (one vector load per field and block
    // instead of a dependent uniform load per read and field); dropped whenever labels have been rewritten
    int rb_base = -1, rb_wif = 0, rb_H = 0, rb_dn = 0;
    auto rb_get = [&](int r) {   // makes the block holding read r current; returns its lane
        if (rb_base < 0 || r < rb_base || r >= rb_base + 64) {
            rb_base = r & ~63;
            const int q = rb_base + ch.lane;
            const bool ok = q < R;
            rb_wif = ok ? ch.wif[q] : -1;
            rb_H = ok ? ch.H[q] : 1;
            rb_dn = ok ? ch.dense_of[q] : -1;
        }
        return r - rb_base;
    };
    auto wif_of = [&](int r) { const int j = rb_get(r); return rl_i32(rb_wif, j); };
    auto H_of = [&](int r) { const int j = rb_get(r); return rl_i32(rb_H, j); };
    auto emission_of = [&](Col<NE> &er, int r) {
        const int dn = rl_i32(rb_dn, rb_get(r));
        if (dn >= 0) { ch.ld(er, ch.eMatRead + (size_t)dn * Ksp); return; }
        if (!foreign) {
            typename CH::ErPre x;
            ch.ld_pre(x, r);
            ch.expand(er, x);
            return;
        }
        struct __attribute__((packed, aligned(4))) U3 { uint32_t a, b, c; };   // (padb_of(10) = 12 bytes per lane)
        const U3 q = *reinterpret_cast<const U3 *>(ch.eridx + ((size_t)r * 64 + ch.lane) * 12);
        const uint32_t w4[4] = {q.a, q.b, q.c, 0u};
        const double tv = ch.ertab[(size_t)r * 64 + ch.lane];
        const int lo = __double2loint(tv), hi = __double2hiint(tv);
#pragma unroll
        for (int i = 0; i < NE; i++) {
            uint32_t code = 0;
#pragma unroll
            for (int b = 0; b < 12; b++)   // byte ch.wave + 2 i, without dynamic register indexing
                if (b == ch.wave + 2 * i) code = (w4[b >> 2] >> ((b & 3) * 8)) & 0xffu;
            const int src = (int)code << 2;
            er.v[i] = __hiloint2double(__builtin_amdgcn_ds_bpermute(src, hi), __builtin_amdgcn_ds_bpermute(src, lo));
        }
    };
    auto sum3 = [&](const Col<NE> (&x)[NH], double (&s)[NH]) {
#pragma unroll
        for (int h = 0; h < NH; h++) {
            s[h] = 0;
#pragma unroll
            for (int i = 0; i < NE; i++) s[h] += x[h].v[i];
        }
        ch.template bsum<NH>(s);
    };

    // alphaStore / log_cStore of the six relabellings: relabelling ir runs slot h on emission column EM[ir][h], every
    // relabelling restarts a block from the same alpha, and slot h's recursion sees nothing but its own column -- so the 18
    // recursions are 9 distinct ones, each shared by two relabellings (same operations on the same inputs: the same values).
    // aS[h][i] / inside[h][i]: slot h fed with emission column i.
    constexpr int EM[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {2, 0, 1}, {1, 2, 0}, {2, 1, 0}};   // i with RR[ir][i] - 1 == h
    Col<NE> aS[3][3];
    double inside[3][3];       // sum of log_cStore over the current block, in grid order
    double logC_before[3] = {0, 0, 0}, logC_after[3];
    {
        double s[3] = {0, 0, 0};
        for (int g = t; g < G; g += NT)
#pragma unroll
            for (int h = 0; h < 3; h++) s[h] += log(ch.cv[h][g]);
        ch.template bsum<3>(s);
#pragma unroll
        for (int h = 0; h < 3; h++) logC_after[h] = s[h];
    }
#pragma unroll
    for (int h = 0; h < 3; h++)
#pragma unroll
        for (int i = 0; i < 3; i++) inside[h][i] = 0;
    bool ever_changed = false;

    // the forward recursion's inputs are fetched a grid ahead (columns) or 64 grids at a time into lanes (block index,
    // transition): a uniform value loaded through the vector path would have to be waited for, draining the column prefetch
    Col<NE> e[NH];
#pragma unroll
    for (int h = 0; h < NH; h++) ch.ld(e[h], ch.eg[h]);
    int where_l = -1;
    double t0_l = 1.0, t1_l = 0.0;
    for (int g = 0; g < G; g++) {
        if ((g & 63) == 0) {
            const int gg = g + ch.lane;
            where_l = gg < G ? where[gg] : -1;
            t0_l = (gg < G && gg > 0) ? ch.tm0(gg - 1) : 1.0;
            t1_l = (gg < G && gg > 0) ? ch.tm1(gg - 1) : 0.0;
        }
        Col<NE> en[NH];
        {
            const size_t gn = (size_t)min(g + 1, G - 1) * Ksp;
#pragma unroll
            for (int h = 0; h < NH; h++) ch.ld(en[h], ch.eg[h] + gn);
        }
        const double t0 = rl_f64(t0_l, g & 63), t1 = rl_f64(t1_l, g & 63);
        // the 9 normalisers of this grid (uniform values): lane 3 h + i keeps d(h, i), so that ONE logarithm per lane
        // replaces 9 per lane (the same function on the same inputs)
        double d_of_lane = 1.0;
        // ---- Rcpp_gibbs_block_forward_one (:1122-1253)
#pragma unroll
        for (int h = 0; h < 3; h++) {
            Col<NE> nx[NH];   // indexed by the emission column i
#pragma unroll
            for (int i = 0; i < 3; i++) {
#pragma unroll
                for (int q = 0; q < NE; q++) {
                    if (g == 0) nx[i].v[q] = valid[q] ? prior * e[i].v[q] : 0.0;
                    else nx[i].v[q] = valid[q] ? e[i].v[q] * (t0 * aS[h][i].v[q] + t1 * one_over_K) : 0.0;
                }
            }
            double sm[NH];
            sum3(nx, sm);
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const double d = 1 / sm[i];
                if (ch.lane == 3 * h + i) d_of_lane = d;
#pragma unroll
                for (int q = 0; q < NE; q++) aS[h][i].v[q] = d * nx[i].v[q];
            }
        }
        {
            const double lg = log(d_of_lane);
#pragma unroll
            for (int h = 0; h < 3; h++)
#pragma unroll
                for (int i = 0; i < 3; i++) inside[h][i] += rl_f64(lg, 3 * h + i);
        }
        const int iBlock = rl_i32(where_l, g & 63);
        if (iBlock > -1) {
            const int grid_start = uni_i(tab[iBlock]), grid_end = uni_i(tab[G + iBlock]);
            const int read_start = uni_i(tab[2 * G + iBlock]), read_end = uni_i(tab[3 * G + iBlock]);
            // ---- Rcpp_consider_block_relabelling: probability of the panel side of each relabelling
            Col<NE> bt[NH];
#pragma unroll
            for (int h = 0; h < NH; h++) ch.ld(bt[h], ch.beta[h] + (size_t)g * Ksp);
            double P[6], ldot[3][3];   // log of sum_k alphaStore(slot h fed with column i) * beta(slot h)
#pragma unroll
            for (int h = 0; h < 3; h++) {
                double dot[3] = {0, 0, 0};
#pragma unroll
                for (int i = 0; i < 3; i++)
#pragma unroll
                    for (int q = 0; q < NE; q++) dot[i] += aS[h][i].v[q] * bt[h].v[q];
                ch.template bsum<3>(dot);
#pragma unroll
                for (int i = 0; i < 3; i++) ldot[h][i] = log(dot[i]);
            }
#pragma unroll
            for (int ir = 0; ir < 6; ir++) {
                P[ir] = 0;
#pragma unroll
                for (int h = 0; h < 3; h++)
                    P[ir] += ldot[h][EM[ir][h]] + -logC_before[h] + -inside[h][EM[ir][h]] + -logC_after[h];
            }
            // ... and of the read classes under it (rcpp_calculate_block_read_label_probabilities_using_H_class)
            int ns[8];
            {
                double cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (int r = read_start + t; r <= read_end; r += NT) {
                    const int hc = ch.Hc[r];
#pragma unroll
                    for (int q = 0; q < 8; q++) cnt[q] += (hc == q) ? 1.0 : 0.0;
                }
                double a4[4] = {cnt[0], cnt[1], cnt[2], cnt[3]}, b4[4] = {cnt[4], cnt[5], cnt[6], cnt[7]};
                ch.template bsum<4>(a4);
                ch.template bsum<4>(b4);
#pragma unroll
                for (int q = 0; q < 4; q++) { ns[q] = (int)a4[q]; ns[4 + q] = (int)b4[q]; }
            }
            double clp[6], cp[6];
#pragma unroll
            for (int ir = 0; ir < 6; ir++) {
                const int n6[6] = {ns[RR[ir][0]], ns[RR[ir][1]], ns[RR[ir][2]], ns[7 - RR[ir][2]], ns[7 - RR[ir][1]], ns[7 - RR[ir][0]]};
                clp[ir] = log_p_H_class2(n6, ff) + P[ir];
            }
            double mx = clp[0];
#pragma unroll
            for (int ir = 1; ir < 6; ir++) mx = clp[ir] > mx ? clp[ir] : mx;
            double tot = 0;
#pragma unroll
            for (int ir = 0; ir < 6; ir++) {
                clp[ir] += -mx;
                if (clp[ir] < -100) clp[ir] = -100;
                cp[ir] = exp(clp[ir]);
                tot += cp[ir];
            }
            const double dn = 1 / tot;
            const double chance = rl_f64(u_block(iBlock), 0);
            int ir_chosen = 0;
            {
                double cum[6];
                cum[0] = cp[0] * dn;
#pragma unroll
                for (int ir = 1; ir < 6; ir++) cum[ir] = cp[ir] * dn + cum[ir - 1];
#pragma unroll
                for (int ir = 5; ir >= 0; ir--) if (chance < cum[ir]) ir_chosen = ir;
            }
            ir_chosen = uni_i(ir_chosen);
            int swp[8];
            swp[0] = 0; swp[7] = 7;
#pragma unroll
            for (int q = 0; q < 6; q++)
                if (q == ir_chosen) {
                    swp[1] = RX[q][0]; swp[2] = RX[q][1]; swp[3] = RX[q][2];
                    swp[4] = 7 - RX[q][2]; swp[5] = 7 - RX[q][1]; swp[6] = 7 - RX[q][0];
                }
            auto swap_of = [&](int v) {   // swp[v] without dynamic register indexing
                int o = 0;
#pragma unroll
                for (int q = 0; q < 8; q++) o = (v == q) ? swp[q] : o;
                return o;
            };
            Col<NE> al[NH];   // alpha at g after the block (for the reset)
            bool have_al = false;
            if (ever_changed || ir_chosen != 0) {
                ever_changed = true;
                // rebuild eMatGrid, alpha, c of the block's grids under the relabelling (:846-912), reads walked by grid
                int iRead = read_start;
                int wif_read = wif_of(iRead);
                if (grid_start > 0) {
#pragma unroll
                    for (int h = 0; h < NH; h++) ch.ld(al[h], ch.alpha[h] + (size_t)(grid_start - 1) * Ksp);
                }
                for (int g2 = grid_start; g2 <= grid_end; g2++) {
                    Col<NE> el[NH];
#pragma unroll
                    for (int h = 0; h < NH; h++)
#pragma unroll
                        for (int q = 0; q < NE; q++) el[h].v[q] = 1.0;
                    while ((iRead <= (R - 1)) & (wif_read < g2)) {
                        iRead += 1;
                        if (iRead < (R - 1)) wif_read = wif_of(iRead);
                    }
                    while ((iRead <= (R - 1)) & (wif_read == g2)) {
                        const int hh = swap_of(H_of(iRead)) - 1;
                        Col<NE> er;
                        emission_of(er, iRead);
#pragma unroll
                        for (int h = 0; h < NH; h++)
                            if (hh == h) {
#pragma unroll
                                for (int q = 0; q < NE; q++) el[h].v[q] *= er.v[q];
                            }
                        iRead += 1;
                        if (iRead <= (R - 1)) wif_read = wif_of(iRead);
                    }
                    const double s0 = g2 > 0 ? ch.tm0(g2 - 1) : 1.0, s1 = g2 > 0 ? ch.tm1(g2 - 1) : 0.0;
#pragma unroll
                    for (int h = 0; h < NH; h++) {
                        ch.st(el[h], ch.eg[h] + (size_t)g2 * Ksp);
#pragma unroll
                        for (int q = 0; q < NE; q++) {
                            if (g2 == 0) al[h].v[q] = valid[q] ? prior * el[h].v[q] : 0.0;
                            else al[h].v[q] = valid[q] ? el[h].v[q] * (s0 * al[h].v[q] + s1 * prior) : 0.0;
                        }
                    }
                    double sm[NH];
                    sum3(al, sm);
#pragma unroll
                    for (int h = 0; h < NH; h++) {
                        const double cc = 1 / sm[h];
#pragma unroll
                        for (int q = 0; q < NE; q++) al[h].v[q] *= cc;
                        ch.st(al[h], ch.alpha[h] + (size_t)g2 * Ksp);
                        if (t == 0) ch.cv[h][g2] = cc;
                    }
                }
                have_al = grid_end == g;
                block_sync();   // every wave is done reading the old labels
                for (int r = read_start + t; r <= read_end; r += NT) {
                    ch.H[r] = swap_of(ch.H[r]);
                    ch.Hc[r] = swap_of(ch.Hc[r]);
                }
                block_sync();
                rb_base = -1;
            }
            // Rcpp_reset_local_variables (:1257-1292)
            if ((iBlock + 1) < n_blocks) {
                if (!have_al) {
#pragma unroll
                    for (int h = 0; h < NH; h++) ch.ld(al[h], ch.alpha[h] + (size_t)g * Ksp);
                }
#pragma unroll
                for (int h = 0; h < 3; h++)
#pragma unroll
                    for (int i = 0; i < 3; i++) aS[h][i] = al[h];
            }
#pragma unroll
            for (int h = 0; h < 3; h++)
#pragma unroll
                for (int i = 0; i < 3; i++) inside[h][i] = 0;
            for (int g2 = grid_start; g2 <= grid_end; g2++)
#pragma unroll
                for (int h = 0; h < 3; h++) logC_before[h] += log(uni_d(&ch.cv[h][g2]));
        }
        {   // logC_after(h) -= log(c_h(g)): the three values in three lanes, one logarithm
            double cv_of_lane = 1.0;
#pragma unroll
            for (int h = 0; h < 3; h++) {
                const double cvh = uni_d(&ch.cv[h][g]);
                if (ch.lane == h) cv_of_lane = cvh;
            }
            const double lg = log(cv_of_lane);
#pragma unroll
            for (int h = 0; h < 3; h++) logC_after[h] -= rl_f64(lg, h);
        }
#pragma unroll
        for (int h = 0; h < NH; h++) e[h] = en[h];
    }
    block_sync();
    // ---- rcpp_sample_H_using_H_class (:213-246); with p.defer_resample the caller draws the uniforms first (in the reference's
    // order: only the reads whose class leaves a choice draw) and k_resample3 does this loop
    if (!p.defer_resample) {
        for (int r = t; r < R; r += NT) ch.H[r] = resample_label(ch.Hc[r], ff, u_draw(r));
    }
    // eMatGrid, forward and backward from the new labels (:1898-1954): the next k_gibbs3 launch does them first (p.rebuild)
}

// rcpp_sample_H_using_H_class (gibbs-nipt-block.cpp:213-246) on its own: one workgroup per chain, the pass's per-read uniforms
// from the explicit buffer (the caller has just written them: qa_gibbs_opts_t.draw_uniforms)
__global__ __launch_bounds__(256) void k_resample3(GibbsParams p) {
    const int c = blockIdx.x, R = p.read_off[c + 1] - p.read_off[c];
    const double ff = p.ff_chain ? p.ff_chain[c] : p.ff;
    const double *ru = p.runif_shard + ((size_t)p.read_off[c] * p.blk_n_pass + (size_t)p.blk_pass * R) * 2;
    int32_t *H = p.H + p.read_off[c];
    const int32_t *Hc = p.H_class + p.read_off[c];
    for (int r = threadIdx.x; r < R; r += blockDim.x) H[r] = resample_label(Hc[r], ff, ru[R + r]);
}

template <int NE, int NW>
void launch_block(const GibbsParams &prm, hipStream_t st) {
    hipLaunchKernelGGL((k_block3<NE, NW>), dim3(prm.C), dim3(64 * NW), 0, st, prm);
    QA_HIP(hipGetLastError());
}

template <int NE, int NW, bool LEAN = false>
void launch_one(const GibbsParams &prm, hipStream_t st) {
    hipLaunchKernelGGL((k_gibbs3<NE, NW, LEAN>), dim3(prm.C), dim3(64 * NW), 0, st, prm);
    QA_HIP(hipGetLastError());
}

}  // namespace

namespace qa {

// geometry at Ksp = 640: one wave per chain (10 columns per lane and label, a SIMD's whole register file) when the launch
// fills the device's 1024 SIMDs anyway, two waves (5 columns, 256 registers) for few chains; one wave otherwise
int gibbs3_waves(int Ksp, int C, int share) {
    if (Ksp != 640) return 1;
    int nw = ((long)C * 2 <= 1024 / share) ? 2 : 1;
    if (const char *forced = getenv("QA_GIBBS_NW")) {   // test hook
        const int f = atoi(forced);
        if (f == 1 || f == 2) nw = f;
    }
    return nw;
}

void launch_gibbs3(const void *params, hipStream_t st) {
    const GibbsParams &prm = *static_cast<const GibbsParams *>(params);
    switch (prm.Ksp / 64) {
        case 1: launch_one<1, 1>(prm, st); break;
        case 2: launch_one<2, 1>(prm, st); break;
        case 3: launch_one<3, 1>(prm, st); break;
        case 4: launch_one<4, 1>(prm, st); break;
        case 5: launch_one<5, 1>(prm, st); break;
        case 6: launch_one<6, 1>(prm, st); break;
        case 8: launch_one<8, 1>(prm, st); break;
        case 10:
            if (prm.er_nt == 64 && prm.lean3) launch_one<10, 1, true>(prm, st);
            else if (prm.er_nt == 64) launch_one<10, 1>(prm, st);
            else launch_one<5, 2>(prm, st);
            break;
        default: throw std::runtime_error("NIPT sampler: Ksubset geometry not built (Ksubset / 64 rounded up must be 1..6, 8 or 10)");
    }
}

void launch_block_rate3(const void *params, hipStream_t st) {
    const GibbsParams &prm = *static_cast<const GibbsParams *>(params);
    hipLaunchKernelGGL(k_block_rate3, dim3((prm.G - 1 + 3) / 4, prm.C), dim3(256), 0, st, prm);
    QA_HIP(hipGetLastError());
}

void launch_resample3(const void *params, hipStream_t st) {
    const GibbsParams &prm = *static_cast<const GibbsParams *>(params);
    hipLaunchKernelGGL(k_resample3, dim3(prm.C), dim3(256), 0, st, prm);
    QA_HIP(hipGetLastError());
}

void launch_block3(const void *params, hipStream_t st) {
    const GibbsParams &prm = *static_cast<const GibbsParams *>(params);
    switch (prm.Ksp / 64) {
        case 1: launch_block<1, 1>(prm, st); break;
        case 2: launch_block<2, 1>(prm, st); break;
        case 3: launch_block<3, 1>(prm, st); break;
        case 4: launch_block<4, 1>(prm, st); break;
        case 5: launch_block<5, 1>(prm, st); break;
        case 6: launch_block<6, 1>(prm, st); break;
        case 8: launch_block<8, 1>(prm, st); break;
        case 10: launch_block<5, 2>(prm, st); break;
        default: throw std::runtime_error("NIPT block Gibbs: Ksubset geometry not built");
    }
}

}  // namespace qa

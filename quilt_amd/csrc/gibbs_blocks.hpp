// gibbs_blocks.hpp -- host side of the NIPT block Gibbs: from the per-boundary switch rate computed on the device
// (k_block_rate3) to the block table the block kernel walks.  Scalar integer / comparison logic per chain, a few
// thousand operations: the part of Rcpp_define_blocked_snps_using_gamma_on_the_fly (QUILT/src/gibbs-nipt-block.cpp:366-523),
// rcpp_make_smoothed_rate / rcpp_determine_where_to_stop (QUILT/src/copied-from-stitch.cpp:446-567) and
// Rcpp_make_gibbs_considers (gibbs-nipt-block.cpp:1307-1553) that follows the K-wide sums.  Worked at the level of
// grids: blocked_snps[i] = blocked_grid[grid[i]] and every 32-SNP grid holds a SNP, so the per-SNP vectors of the
// reference carry no extra information.
#pragma once

#include <algorithm>
#include <cstdint>
#include <numeric>
#include <vector>

namespace qa {

struct BlockTable {
    int n_blocks = 0;
    std::vector<int32_t> grid_start, grid_end, reads_start, reads_end;   // per block
    std::vector<int32_t> grid_where;                                     // per grid: block ending there, else -1
};

// weighted mean of the rate over +- shuffle_bin_radius bp around the midpoint of each pair of neighbouring grids
inline std::vector<double> smoothed_rate(const double *rate, const int32_t *L_grid, int nGrids, int radius) {
    std::vector<double> out(std::max(nGrids - 1, 0), 0.0);
    for (int g = 0; g + 1 < nGrids; g++) {
        const int focal = (L_grid[g] + L_grid[g + 1]) / 2;
        double acc = 0, bp_total = 0;
        int left = radius, prev = focal;
        for (int i = g; left > 0 && i >= 0; i--) {       // leftwards: segment (L_grid[i], prev] carries rate[i]
            int add = prev - L_grid[i];
            if (left - add < 0) { add = left; left = 0; } else left -= add;
            acc = acc + add * rate[i];
            bp_total += add;
            prev = L_grid[i];
        }
        left = radius; prev = focal;
        for (int i = g + 1; left > 0 && i < nGrids; i++) {   // rightwards: segment (prev, L_grid[i]] carries rate[i - 1]
            int add = L_grid[i] - prev;
            if (left - add < 0) { add = left; left = 0; } else left -= add;
            acc = acc + add * rate[i - 1];
            bp_total += add;
            prev = L_grid[i];
        }
        out[g] = acc / bp_total;
    }
    return out;
}

// walk away from a peak until the rate stops falling (rcpp_determine_where_to_stop); returns the minimum seen
inline int valley_next_to_peak(const std::vector<double> &sm, const std::vector<char> &available, int peak, double thresh,
                               int nGrids, bool leftwards) {
    const int step = leftwards ? -1 : 1;
    int at = peak, best = peak, n = 1;
    double lowest = sm[peak], five_back = sm[peak];
    for (;;) {
        at += step;
        const double v = sm[at];
        if (n >= 5) five_back = sm[at - 5 * step];
        n++;
        if (v < lowest) { best = at; lowest = v; }
        if (at <= 2 || at >= nGrids - 3) break;
        if (!available[at + step]) break;
        if (3 * lowest < v) break;
        if (v < thresh && five_back < v) break;
    }
    return best;
}

inline std::vector<int32_t> define_blocked_grids(const double *rate2, const int32_t *L_grid, int nGrids, int radius,
                                                 double quantile_prob) {
    std::vector<int32_t> blocked(nGrids, 0);
    const int n = nGrids - 1;
    if (n < 1) return blocked;
    const std::vector<double> sm = smoothed_rate(rate2, L_grid, nGrids, radius);
    std::vector<int> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return sm[a] < sm[b]; });
    const double thresh = std::min(1.0, sm[order[(int)(n * quantile_prob)]]);   // rcpp_simple_quantile, capped at 1
    std::vector<char> available(n, 0);
    int n_available = 0;
    for (int i = 0; i < n; i++) {
        available[i] = thresh < sm[i];
        n_available += available[i];
    }
    if (n_available == 0) return blocked;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return sm[a] > sm[b]; });
    std::vector<int> cuts;
    for (int i = 0; i < n_available; i++) {
        const int peak = order[i];
        if (!available[peak]) continue;
        const int a = std::max(peak - 1, 0), b = std::min(peak + 1, nGrids - 2);
        int around = 0;
        for (int j = a; j <= b; j++) around += available[j];
        if (around == 3) {
            const int lo = valley_next_to_peak(sm, available, peak, thresh, nGrids, true);
            const int hi = valley_next_to_peak(sm, available, peak, thresh, nGrids, false);
            for (int j = lo; j <= hi; j++) available[j] = 0;
        } else {
            for (int j = a; j <= b; j++) available[j] = 0;
        }
        cuts.push_back(peak + 1);
    }
    if (*std::min_element(cuts.begin(), cuts.end()) != 0) cuts.push_back(0);
    if (*std::max_element(cuts.begin(), cuts.end()) != nGrids - 1) cuts.push_back(nGrids - 1);
    std::sort(cuts.begin(), cuts.end());
    for (size_t i = 0; i + 1 < cuts.size(); i++)
        for (int j = cuts[i]; j <= cuts[i + 1]; j++) blocked[j] = (int32_t)i;
    return blocked;
}

inline double ceiling_point5(double x) { return ((double)(int)x < x) ? x + 0.5 : x; }

inline BlockTable make_gibbs_considers(const std::vector<int32_t> &blocked_in, const int32_t *wif0, int nReads) {
    const int nGrids = (int)blocked_in.size();
    BlockTable T;
    int n_blocks = blocked_in[nGrids - 1] + 1;
    std::vector<int32_t> gs, ge;
    for (int g = 0, start = 0; g < nGrids; g++)
        if (g == nGrids - 1 || blocked_in[g] < blocked_in[g + 1]) { gs.push_back(start); ge.push_back(g); start = g + 1; }
    gs.resize(n_blocks); ge.resize(n_blocks);
    std::vector<int32_t> block_of(nGrids, 0);
    for (int b = 0; b < n_blocks; b++)
        for (int g = gs[b]; g <= ge[b]; g++) block_of[g] = b;
    std::vector<int32_t> rs(n_blocks, -1), re(n_blocks, -1);
    if (nReads > 0) {
        // the reference's single pass over the reads, including what it does with the last read (:1391-1407)
        int first = 0, prev_block = block_of[wif0[0]];
        for (int r = 1; r < nReads; r++) {
            const int b = block_of[wif0[r]];
            if (r == nReads - 1) { rs[b] = first; re[b] = r; }
            else if (prev_block < b) { rs[prev_block] = first; re[prev_block] = r - 1; first = r; prev_block = b; }
        }
    }
    std::vector<int> gone;
    for (int b = 0; b < n_blocks; b++) if (rs[b] == -1) gone.push_back(b);
    if (!gone.empty() && (int)gone.size() < n_blocks) {
        // a run of read-less blocks is shared out between its neighbours at its midpoint (:1432-1490)
        int run_first = 0;
        for (int j = 0; j < (int)gone.size(); j++) {
            const bool run_ends = j == (int)gone.size() - 1 || gone[j + 1] - gone[j] != 1;
            if (!run_ends) { run_first -= 1; }
            else {
                int s1 = gone[run_first], e1 = gone[j];
                double x = ceiling_point5(0.5 * (double)(gs[s1] + ge[e1]));
                if (s1 == 0) { s1 = 1; x = 0; }
                if (e1 == n_blocks - 1) { e1 -= 1; x = ge[n_blocks - 1]; }
                gs[e1 + 1] = (int32_t)x;
                ge[s1 - 1] = (int32_t)(x - 1);
                run_first = j;
            }
            run_first += 1;
        }
        int o = 0;
        for (int b = 0; b < n_blocks; b++)
            if (rs[b] != -1) { rs[o] = rs[b]; re[o] = re[b]; gs[o] = gs[b]; ge[o] = ge[b]; o++; }
        n_blocks = o;
        gs.resize(o); ge.resize(o); rs.resize(o); re.resize(o);
    }
    T.n_blocks = n_blocks;
    T.grid_start = gs; T.grid_end = ge; T.reads_start = rs; T.reads_end = re;
    T.grid_where.assign(nGrids, -1);
    for (int b = 0; b < n_blocks; b++) T.grid_where[ge[b]] = b;
    return T;
}

}  // namespace qa

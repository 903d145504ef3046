// panel.hpp -- device-resident prepared reference panel.
#pragma once

#include "common.hpp"

// HBM layout (all resident for the life of the handle):
//   hm     uint8  [nGrids][Kp]      hapMatcherR, grid-major, haplotypes contiguous, row pitch Kp =
//                                   K rounded up to 4096 (zero padded): every 16-haplotype chunk is one
//                                   aligned 16-byte load and the fp64 ranking kernels can fetch whole
//                                   256-lane chunk rows without bounds checks (K=50 000, G=2 000: 106 MB)
//   B      int32  [nGrids][nMaxDH]  distinctHapsB (same bytes as R's nMaxDH x nGrids matrix)
//   sp_*   CSR over grids of the "special" haplotypes (hapMatcher == 0): ascending k and the 32-bit
//          word the reference would decode for it
//   sigma  double [nGrids-1]        transMatRate_t row 0
struct qa_panel {
    int K = 0, G = 0, T = 0, nMaxDH = 0, nrow = 0;  // nrow = nMaxDH + 1
    int Kp = 0;
    double ref_error = 0;
    int device = 0;
    qa::DBuf<uint8_t> hm;
    qa::DBuf<int32_t> B;
    qa::DBuf<int32_t> sp_off;   // G + 1
    qa::DBuf<int32_t> sp_k;     // n_special
    qa::DBuf<uint32_t> sp_word; // n_special
    qa::DBuf<double> sigma;     // G - 1
    qa::DBuf<double> tm1;       // G - 1: transMatRate_t row 1 as the caller passed it
    // grids that hold specials: sp_gidx[g] = index among them (or -1), sp_chunk_at[index][Kq / 16] = position of the first
    // special at or after each 16-haplotype chunk in the per-pass special-emission array of the fp64 ranking kernels (each
    // such grid's list followed by 16 zero entries: k_emat), so that those kernels need no search
    qa::DBuf<int32_t> sp_gidx, sp_chunk_at;
    int n_sp_grids = 0;
    qa::DBuf<double> IE;        // only when the caller's distinctHapsIE is not the (B, eps) expansion
    bool ie_derived = true;
    int share = 1;              // host threads / panel handles sharing this device (qa_panel_set_device_share)
    bool rank_fp64 = true;      // best-haplotype lists from fp64-state passes (qa_panel_set_ranking_precision)
    bool dosage_fp64 = false;   // dosage / alpha / beta / gamma outputs from fp64-state passes (qa_panel_set_dosage_precision)
    bool sum_order_ref = false; // VALIDATION MODE: full-panel passes by the reference-order kernels (qa_panel_set_sum_order, fullpass_ref.hip)
    bool sum_order_grid0_ltr = false; // ... with grid 0's sum(alphaHat_t_col) left to right instead of Armadillo's two accumulators (mode 2)
    int n_special = 0;
    std::vector<double> h_sigma, h_tm1;  // transMatRate_t rows 0 and 1 as passed
    std::vector<int32_t> h_sp_off;
    hipStream_t stream = nullptr;
    hipStream_t pass_stream = nullptr;    // higher-priority stream of the full-panel calls (qa_panel_set_pass_priority), else null
    hipStream_t gibbs_stream = nullptr;   // CU-masked stream of the Gibbs launches (qa_panel_set_cu_partition), else null
    qa::Arena arena;            // scratch of every launch set on this panel (see common.hpp) -- unless `exclusive`
    bool exclusive = false;     // launch sets run with the device to themselves, out of the device-wide arena (qa_panel_set_exclusive)
    // the gate this handle's launch sets hold while they run (null: none; the handle's own arena then), and the arena whose
    // size plans them (GateHold::arena() is the one they carve from)
    qa::DeviceGate *gate() { return exclusive ? &qa::device_gate(device) : nullptr; }
    qa::Arena &A() { return exclusive ? qa::device_gate(device).arena : arena; }
    // at least `bytes` of scratch for the launch set in hand (an exclusive handle: the device-wide arena, grown to the
    // planning budget at once -- the caller holds the device)
    void require_scratch(size_t bytes) {
        if (!exclusive) { arena.require(bytes); return; }
        qa::DeviceGate &g = qa::device_gate(device);
        std::lock_guard<std::mutex> lk(g.mu);
        g.grow_arena(bytes);
    }
    // the bytes launch sets are planned against: the arena's budget, read under the gate's lock for an exclusive handle (another
    // host thread's admission may be re-allocating the device-wide arena at this moment)
    size_t plan_budget() {
        if (!exclusive) return arena.budget_shared(share);
        qa::DeviceGate &g = qa::device_gate(device);
        std::lock_guard<std::mutex> lk(g.mu);
        return g.arena.budget_shared(1);
    }
    int sharers() const { return exclusive ? 1 : share; }   // handles whose launch sets may be on the device at the same time
    qa::Arena aux;              // per-call index / list buffers of the driver-level entry points (grow-only: a call-local
                                // hipMalloc / hipFree pair would synchronise the device with the other host threads' launches)
    // scratch owned by the panel handle, grown on demand (see fullpass.hip)
    struct Scratch;
    Scratch *scratch = nullptr;
    ~qa_panel();
};

namespace qa {
// after sp_off / sp_k are on the device and h_sp_off on the host: sp_gidx, sp_chunk_at, tm1
void finish_panel_tables(qa_panel *p);
// fullpass.hip: delete the handle's full-pass scratch views (re-created on the next call, over the arena then in use)
void drop_pass_scratch(qa_panel *p);
}

// The all-SNP side of a QUILT2 panel (rare + common SNPs): what the final all-SNP Gibbs call needs on top of the
// common-SNP panel tables.  rare_snp holds 0-based all-SNP indices, ascending within a haplotype.
struct qa_rare_common {
    int device = 0, K = 0, T_all = 0, G_all = 0;
    qa::DBuf<int32_t> common_index;   // [T_all] 0-based index among the common SNPs, -1 for a rare SNP
    qa::DBuf<int64_t> rare_ptr;       // [K + 1]
    qa::DBuf<int32_t> rare_snp;
    std::vector<int64_t> h_rare_ptr;
    std::vector<int32_t> h_rare_snp;
    std::vector<int32_t> h_common_index;
    std::vector<double> h_sigma, h_tm1;   // transMatRate_t of the all-SNP grid, rows 0 and 1
};

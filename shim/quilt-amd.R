## quilt-amd.R -- the R side of libquilt_amd's fast path (added to QUILT/R/ by shim/QUILT-R.patch).
##
## QUILT() imputes a core's samples one at a time: `for(iSample in sampleRange[1]:sampleRange[2]) get_and_impute_one_sample(...)`
## (QUILT/R/quilt.R:832-982).  One sample's chains use a few of an MI355X's 1 024 SIMDs; the device is filled by advancing ALL
## chains of MANY samples in lock-step, which is what the library's qa_impute_samples does for a whole sample range behind ONE
## `.Call("qa_impute_sample_range", ...)` (shim/quilt_amd_shim.c; csrc/impute.cpp is get_and_impute_one_sample's loop nest,
## functions.R:330-1300, in host C++ over the batched kernels).  This file holds
##   quilt_amd_range_is_covered()   -- may this run take the range call?  Anything it does not cover falls back to the
##                                     unpatched loop (plots, HLA, truth haplotypes / genotypes, per-read outputs, ...)
##   quilt_amd_impute_sample_range() -- the range's samples through the ONE call.  Two forms:
##       (a) NATIVE I/O (the fast one; quilt_amd_native_io_is_covered): `.Call("qa_impute_bam_range", bam_files, ...)` reads the
##           BAM files on host threads (csrc/hostio.cpp), imputes, formats every sample's VCF column and sums the range's four
##           count arrays natively -- the loop body of quilt.R:832-982 with get_and_impute_one_sample's own I/O
##           (functions.R:243-298, :1380-1463) in one call.  R's serial loader and formatter handle about one sample per second;
##           the device imputes about forty.
##       (b) R I/O (the fallback: CRAM input, bx tags, QUILT_AMD_NATIVE_IO=0): per sample the reference's own loader
##           (functions.R:132-172, :251-298), then `.Call("qa_impute_sample_range", ...)`, then per sample what
##           get_and_impute_one_sample returns (functions.R:1380-1463) by the reference's own functions.
## Nothing in this FILE computes a probability.
##
## QUILT_AMD_SUM_ORDER (environment, default 0): 1 runs the full-panel passes in the library's VALIDATION MODE -- every K-wide sum
## in the order the reference's C++ adds it, bit-identical best-haplotype lists to the CPU package's on tie-rich panels, 20-50x
## slower passes (INTEGRATION.md 3a); 2: the same with grid 0's Armadillo sum() read left to right (include/quilt_amd.h).
##
## Random draws: 2 048 chains advancing together cannot consume ONE R stream in the reference's order, so the range call uses
## the library's counter streams keyed by (seed, global sample index, Gibbs sample); `seed` plays set.seed's part, and a
## sample's result does not depend on the range / core / launch set it lands in.  set.seed()-identical output to the CPU
## package is therefore NOT promised by this path (it is by the four per-call entries of QUILT-src.patch for diploid samples).

quilt_amd_range_is_covered <- function(
    method, make_plots, make_plots_block_gibbs, hla_run, have_truth_haplotypes, have_truth_genotypes,
    record_interim_dosages, output_read_label_prob, record_read_label_usage, plot_per_sample_likelihoods,
    plot_p1, make_heuristic_plot, estimate_bq_using_truth_read_labels, addOptimalHapsToVCF, use_splitreadgl,
    small_ref_panel_skip_equally_likely_reads, shard_check_every_pair, use_hapMatcherR, calculate_gamma_on_the_fly,
    RData_objects_to_save, n_gibbs_sample_its = 1
) {
    if (Sys.getenv("QUILT_AMD_RANGE", "1") == "0") return(FALSE)            ## opt out: the unpatched loop
    if (!is.loaded("qa_impute_sample_range", PACKAGE = "QUILT")) return(FALSE) ## built without QUILT-src.patch
    ok <- method %in% c("diploid", "nipt") &&
        !make_plots && !make_plots_block_gibbs &&      ## need gamma matrices (return_gamma_t) of every call
        !hla_run &&                                     ## needs fbsoL$gammaMT_t / gammaMU_t at one grid (functions.R:1264-1283)
        !have_truth_haplotypes && !have_truth_genotypes && ## phasefile / genfile: per-iteration accuracy printing, truth labels
        !record_interim_dosages && !output_read_label_prob && !record_read_label_usage &&
        !plot_per_sample_likelihoods && !plot_p1 && !make_heuristic_plot &&
        !estimate_bq_using_truth_read_labels && !addOptimalHapsToVCF &&
        !use_splitreadgl && !small_ref_panel_skip_equally_likely_reads &&
        shard_check_every_pair && use_hapMatcherR && calculate_gamma_on_the_fly &&
        is.null(RData_objects_to_save) && n_gibbs_sample_its == 1
    return(isTRUE(ok))
}


## may the range's I/O run natively?  Everything the native loader does not implement keeps the reference's R loader.
quilt_amd_native_io_is_covered <- function(bam_files, cram_files, use_bx_tag, pos, pos_all, impute_rare_common) {
    if (Sys.getenv("QUILT_AMD_NATIVE_IO", "1") == "0") return(FALSE)
    if (!is.loaded("qa_impute_bam_range", PACKAGE = "QUILT")) return(FALSE)
    one_letter <- function(p) all(nchar(as.character(p[, 3])) == 1) && all(nchar(as.character(p[, 4])) == 1)
    ok <- length(cram_files) == 0 || all(cram_files == "") || all(is.na(cram_files))   ## CRAM: not decoded natively
    ok <- ok && length(bam_files) > 0 && !any(is.na(bam_files)) && !isTRUE(use_bx_tag) ## bx tags: STITCH's loader only
    ok <- ok && one_letter(pos) && (!impute_rare_common || one_letter(pos_all))        ## biallelic SNPs, single letters
    return(isTRUE(ok))
}


## the reference's loader for one sample: functions.R:251-298 (and :132-172 with all SNPs for impute_rare_common)
quilt_amd_load_sample <- function(
    iSample, L, pos, bam_files, cram_files, reference, iSizeUpperLimit, bqFilter, useSoftClippedBases, chr, sampleNames,
    downsampleToCov, tempdir, regionName, chrStart, chrEnd, use_bx_tag, bxTagUpperLimit, grid
) {
    loadBamAndConvert(
        iBam = iSample, L = L, pos = pos, nSNPs = nrow(pos), bam_files = bam_files, cram_files = cram_files,
        reference = reference, iSizeUpperLimit = iSizeUpperLimit, bqFilter = bqFilter,
        useSoftClippedBases = useSoftClippedBases, chr = chr, N = length(sampleNames),
        downsampleToCov = downsampleToCov, sampleNames = sampleNames, inputdir = tempdir, regionName = regionName,
        tempdir = tempdir, chrStart = chrStart, chrEnd = chrEnd, chrLength = NA, save_sampleReadsInfo = TRUE,
        use_bx_tag = use_bx_tag, bxTagUpperLimit = bxTagUpperLimit, default_sample_no_read_behaviour = "return_null"
    )
    load(file_sampleReads(tempdir, iSample, regionName))
    removeTmpSamplesFile(tempdir, iSample, regionName, save_sampleReadsInfo = TRUE)
    ungridded <- sampleReads
    if (length(sampleReads) > 0) {
        sampleReads <- snap_sampleReads_to_grid(sampleReads = sampleReads, grid = grid)
    }
    return(list(sampleReads = sampleReads, ungridded = ungridded))
}


quilt_amd_impute_sample_range <- function(
    sampleRange, n_handles = 3L, device = 0L,
    ## the panel (quilt.R's objects of the same names)
    rhb_t, hapMatcherR, distinctHapsB, distinctHapsIE, eMatDH_special_matrix_helper, eMatDH_special_matrix,
    use_eMatDH_special_symbols, small_transMatRate_tc_H, ref_error, L_grid,
    ## the run's parameters
    method, nGibbsSamples, n_seek_its, n_burn_in_seek_its, Ksubset, Knew, K_top_matches, heuristic_match_thin,
    small_ref_panel_gibbs_iterations, small_ref_panel_block_gibbs_iterations, maxDifferenceBetweenReads, minGLValue,
    shuffle_bin_radius, seed, ff_values, use_mspbwt, mspbwtL, mspbwtM, mspbwt_nindices,
    impute_rare_common, special_rare_common_objects, pos_all,
    ## loading and output
    L, pos, grid, bam_files, cram_files, reference, iSizeUpperLimit, bqFilter, useSoftClippedBases, chr, sampleNames,
    downsampleToCov, tempdir, regionName, chrStart, chrEnd, use_bx_tag, bxTagUpperLimit,
    minimum_number_of_sample_reads, output_gt_phased_genotypes
) {
    w <- sampleRange[1]:sampleRange[2]
    n <- length(w)
    sum_order <- as.integer(Sys.getenv("QUILT_AMD_SUM_ORDER", "0"))
    panel_objects <- list(
        hapMatcherR = hapMatcherR, distinctHapsB = distinctHapsB, distinctHapsIE = distinctHapsIE,
        eMatDH_special_matrix_helper = eMatDH_special_matrix_helper, eMatDH_special_matrix = eMatDH_special_matrix,
        rhb_t = rhb_t, transMatRate_t = small_transMatRate_tc_H[, , 1], ref_error = ref_error,
        use_eMatDH_special_symbols = as.integer(use_eMatDH_special_symbols)
    )
    params <- list(
        nGibbsSamples = nGibbsSamples, n_seek_its = n_seek_its, Ksubset = Ksubset, Knew = Knew,
        K_top_matches = K_top_matches, heuristic_match_thin = heuristic_match_thin,
        small_ref_panel_gibbs_iterations = small_ref_panel_gibbs_iterations,
        small_ref_panel_block_gibbs_iterations = as.integer(small_ref_panel_block_gibbs_iterations - 1L),   ## 0-based, as impute_one_sample passes them on
        maxDifferenceBetweenReads = maxDifferenceBetweenReads, minGLValue = minGLValue, Jmax = 10000,   ## functions.R:688
        seed = if (is.na(seed)) 1 else as.numeric(seed),
        device = as.numeric(device),   ## 0-based, modulo the number of GPUs: mclapply's iCore - 1 (one R worker per GPU: nCores = GPUs)
        sum_order = sum_order
    )
    if (!is.na(n_burn_in_seek_its)) params[["n_burn_in_seek_its"]] <- n_burn_in_seek_its
    if (use_mspbwt) {
        params <- c(params, list(use_mspbwt = TRUE, mspbwtL = mspbwtL, mspbwtM = mspbwtM, mspbwt_nindices = mspbwt_nindices))
    }
    if (method == "nipt") {
        params <- c(params, list(method = "nipt", shuffle_bin_radius = shuffle_bin_radius))
        panel_objects[["L_grid"]] <- as.numeric(L_grid)
    }
    if (impute_rare_common) {
        params[["impute_rare_common"]] <- TRUE
        panel_objects[["rare_common"]] <- list(
            snp_is_common = special_rare_common_objects[["snp_is_common"]],
            rare_per_hap_info = special_rare_common_objects[["rare_per_hap_info"]],
            transMatRate_t = special_rare_common_objects[["small_transMatRate_tc_H"]][, , 1],
            L_grid = as.numeric(special_rare_common_objects[["L_grid"]])
        )
    }
    ## ---- (a) native I/O: BAM paths in, VCF columns and the range's counts out (one call)
    if (quilt_amd_native_io_is_covered(bam_files, cram_files, use_bx_tag, pos, pos_all, impute_rare_common)) {
        sites <- list(
            chr = chr, L = as.integer(L), ref = as.character(pos[, 3]), alt = as.character(pos[, 4]), grid = as.integer(grid),
            bqFilter = bqFilter, iSizeUpperLimit = iSizeUpperLimit, useSoftClippedBases = useSoftClippedBases,
            downsampleToCov = downsampleToCov, chrStart = chrStart, chrEnd = chrEnd,
            minimum_number_of_sample_reads = minimum_number_of_sample_reads,
            output_gt_phased_genotypes = output_gt_phased_genotypes
        )
        if (impute_rare_common) {
            sites <- c(sites, list(
                L_all = as.integer(pos_all[, 2]), ref_all = as.character(pos_all[, 3]), alt_all = as.character(pos_all[, 4]),
                grid_all = as.integer(special_rare_common_objects[["grid"]])
            ))
        }
        if (method == "nipt") params[["ff"]] <- as.numeric(ff_values[w])
        print_message(paste0("Imputing samples ", w[1], " to ", w[n], " on the GPU from their BAM files (", n, " samples in one call)"))
        out <- .Call("qa_impute_bam_range", as.character(bam_files[w]), sites, panel_objects, params, as.numeric(w - 1L),
                     as.integer(n_handles), PACKAGE = "QUILT")
        results <- as.list(1:n)
        nS <- length(out[["afCount"]])
        ## The loop of quilt.R:955-961 adds eij / fij / max_gen / per_sample_alleleCount of every imputed sample to the core's four
        ## count arrays.  The call returns those sums for the whole range, formed in sample order (the same floating-point sums):
        ## the patched loop adds them ONCE (attribute "quilt_amd_counts", QUILT-R.patch) and every sample contributes exact zeros
        ## here -- one shared zero vector, nothing per sample and SNP crosses into R but the column itself.
        zero1 <- rep(0, nS)
        zero2 <- array(0, c(nS, 2))
        no_gen <- matrix(0L, nrow = 0, ncol = 2)
        for(i in 1:n) {
            if (!out[["sample_was_imputed"]][i]) {
                print_message(paste0("Sample number ", w[i], " with sample name ", sampleNames[w[i]], " has ", out[["n_reads"]][i], " reads which is fewer than the minimum ", minimum_number_of_sample_reads, ". This sample will therefore not be imputed and all results will be set to missing"))
                results[[i]] <- list(sample_was_imputed = FALSE, per_sample_vcf_col = "./.:.,.,.:.:.,.")
                next
            }
            results[[i]] <- list(
                sample_was_imputed = TRUE, eij = zero1, fij = zero1, max_gen = no_gen, per_sample_alleleCount = zero2,
                per_sample_vcf_col = out[["per_sample_vcf_col"]][[i]],
                super_out_hap_dosages = NULL, super_out_read_labels = out[["read_labels"]][[i]],
                super_out_dosage_matrix = NULL, final_read_labels_prob = as.list(1:3)
            )
        }
        attr(results, "quilt_amd_counts") <- out[c("infoCount", "afCount", "hweCount", "alleleCount")]
        attr(results, "quilt_amd_seconds") <- out[["seconds"]]
        return(results)
    }
    ## ---- (b) R I/O around the call
    load1 <- function(iSample, L, pos, grid) {
        quilt_amd_load_sample(
            iSample = iSample, L = L, pos = pos, bam_files = bam_files, cram_files = cram_files, reference = reference,
            iSizeUpperLimit = iSizeUpperLimit, bqFilter = bqFilter, useSoftClippedBases = useSoftClippedBases, chr = chr,
            sampleNames = sampleNames, downsampleToCov = downsampleToCov, tempdir = tempdir, regionName = regionName,
            chrStart = chrStart, chrEnd = chrEnd, use_bx_tag = use_bx_tag, bxTagUpperLimit = bxTagUpperLimit, grid = grid
        )
    }
    ## ---- 1. reads of every sample of the range (the reference's own loader; functions.R:132-172, :251-298)
    loaded <- lapply(w, function(iSample) load1(iSample, L, pos, grid))
    loaded_all <- NULL
    if (impute_rare_common) {
        loaded_all <- lapply(w, function(iSample) {
            load1(iSample, pos_all[, 2], pos_all, special_rare_common_objects[["grid"]])
        })
    }
    results <- as.list(1:n)
    enough <- sapply(loaded, function(x) length(x[["sampleReads"]]) >= minimum_number_of_sample_reads)
    for(i in which(!enough)) {
        ## functions.R:280-298
        print_message(paste0("Sample number ", w[i], " with sample name ", sampleNames[w[i]], " has ", length(loaded[[i]][["sampleReads"]]), " reads which is fewer than the minimum ", minimum_number_of_sample_reads, ". This sample will therefore not be imputed and all results will be set to missing"))
        results[[i]] <- list(sample_was_imputed = FALSE, per_sample_vcf_col = "./.:.,.,.:.:.,.")
    }
    keep <- which(enough)
    if (length(keep) == 0) {
        return(results)
    }
    ## ---- 2. the ONE call: every chain of every kept sample of the range, in lock-step on the device
    if (method == "nipt") params[["ff"]] <- as.numeric(ff_values[w[keep]])
    all_reads <- NULL
    if (impute_rare_common) {
        all_reads <- lapply(loaded_all[keep], "[[", "sampleReads")
    }
    print_message(paste0("Imputing samples ", w[keep[1]], " to ", w[keep[length(keep)]], " on the GPU (", length(keep), " samples in one call)"))
    out <- .Call(
        "qa_impute_sample_range", lapply(loaded[keep], "[[", "sampleReads"), panel_objects, params,
        as.numeric(w[keep] - 1L),      ## every kept sample's own global index: its streams do not depend on which other samples
                                       ## of the range were skipped
        as.integer(n_handles), all_reads, PACKAGE = "QUILT"
    )
    ## ---- 3. per sample, what get_and_impute_one_sample returns (functions.R:1304-1463)
    nL <- if (method == "nipt") 3 else 2
    for(j in seq_along(keep)) {
        i <- keep[j]
        nS <- nrow(out[["dosage"]])                      ## all SNPs with impute_rare_common (the switch-over of functions.R:1325-1333)
        gp_t <- matrix(out[["gp_t"]][, j], nrow = 3, byrow = TRUE)
        phasing_haps <- t(matrix(out[["phasing_haps"]][, j], nrow = nL, byrow = TRUE))
        ## allele counts from the reads as loaded, before gridding changes nothing here (functions.R:1380-1398)
        sampleReads <- if (impute_rare_common) loaded_all[[i]][["sampleReads"]] else loaded[[i]][["sampleReads"]]
        a <- unlist(sapply(sampleReads, function(x) x[[3]]))
        b <- unlist(sapply(sampleReads, function(x) x[[4]]))
        bqProbs <- STITCH::convertScaledBQtoProbs(matrix(a, ncol = 1))
        c1 <- increment2N(y = as.numeric(bqProbs[, 1]), z = as.numeric(b), yT = as.integer(nrow(bqProbs)), xT = as.integer(nS - 1))
        c2 <- increment2N(y = as.numeric(bqProbs[, 2]), z = as.numeric(b), yT = as.integer(nrow(bqProbs)), xT = as.integer(nS - 1))
        per_sample_alleleCount <- cbind(c2, c1 + c2)
        eij <- round(gp_t[2, ] + 2 * gp_t[3, ], 3)
        fij <- round(gp_t[2, ] + 4 * gp_t[3, ], 3)
        max_gen <- get_max_gen_rapid(gp_t)
        if (method == "diploid") {
            per_sample_vcf_col <- STITCH::rcpp_make_column_of_vcf(
                gp_t = gp_t, use_read_proportions = FALSE, use_state_probabilities = TRUE, read_proportions = matrix(),
                q_t = t(phasing_haps), add_x_2_cols = FALSE, x_t = matrix()
            )
            if (output_gt_phased_genotypes) {
                per_sample_vcf_col <- paste0(
                    round(phasing_haps[, 1]), "|", round(phasing_haps[, 2]),
                    substring(per_sample_vcf_col, first = 4, last = 100L)
                )
            }
        } else {
            mat_dosage <- out[["dosage"]][, j]
            fet_dosage <- out[["fet_dosage"]][, j]
            fet_gp_t <- matrix(out[["fet_gp_t"]][, j], nrow = 3, byrow = TRUE)
            per_sample_vcf_col <- paste0(
                round(phasing_haps[, 1]), "|", round(phasing_haps[, 2]), "|", round(phasing_haps[, 3]), ":",
                round(gp_t[1, ], 3), ",", round(gp_t[2, ], 3), ",", round(gp_t[3, ], 3), ":", round(mat_dosage, 3), ":",
                round(fet_gp_t[1, ], 3), ",", round(fet_gp_t[2, ], 3), ",", round(fet_gp_t[3, ], 3), ":", round(fet_dosage, 3)
            )
        }
        results[[i]] <- list(
            sample_was_imputed = TRUE, eij = eij, fij = fij, max_gen = max_gen,
            per_sample_alleleCount = per_sample_alleleCount, per_sample_vcf_col = per_sample_vcf_col,
            super_out_hap_dosages = NULL, super_out_read_labels = out[["read_labels"]][[j]],
            super_out_dosage_matrix = NULL, final_read_labels_prob = as.list(1:3)
        )
    }
    return(results)
}

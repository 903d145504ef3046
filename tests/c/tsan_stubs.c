/* tsan_stubs.c -- link-time stand-ins for the library's DEVICE entry points, for tests/c/tsan_harness.cpp only: the harness links
 * the product's host sources (csrc/impute.cpp, bamrange.cpp, hostio.cpp, mspbwt.cpp) without the HIP objects, and csrc/impute.cpp's own table
 * names these functions.  None of them is ever called there (the harness runs the loop over its own table); each fails loudly if it
 * were.  No prototypes on purpose: C linkage, the arguments are not looked at. */
#include <stdio.h>
#include <stdlib.h>
#define STUB(name) int name() { fprintf(stderr, "tsan harness: device entry %s reached\n", #name); abort(); return -1; }
STUB(qa_Rcpp_make_gl_bound)
STUB(qa_fullpass_batch)
STUB(qa_fullpass_reads_select_batch)
STUB(qa_gibbs_batch)
STUB(qa_gibbs_batch_rare_common)
STUB(qa_host_alloc)
STUB(qa_host_free)
STUB(qa_panel_bind_thread)
STUB(qa_panel_get_dims)
STUB(qa_rcpp_make_eMatRead_t_hap_major)
STUB(qa_rcpp_make_eMatRead_t_nsnps)
STUB(qa_rcpp_make_eMatRead_t_rare_common)
int qa_device_count(void) { return 0; }

/*
 * mini_r.c -- TEST INFRASTRUCTURE: a small runtime behind the subset of R's C API that shim/qa_r_api.h declares, so that
 * shim/quilt_amd_shim.c can be EXECUTED (not only type-checked) on a machine without R: tests/test_shim_gpu.py builds R-shaped
 * objects through it, calls the registered `.Call` routines by name the way R's `.Call` does (look-up in the table
 * R_registerRoutines received, arity check) and reads the results back.  It is not R: no garbage collector (objects live until
 * mini_r_reset), no NA handling beyond R_NaReal, `unif_rand()` serves a sequence the test loads beforehand, Rf_error() unwinds
 * to the caller of mini_r_dotcall with the message kept.  Semantics the shim relies on and that are kept: vectors carry a type,
 * a length and `names` / `dim` attributes; matrices are column-major with Rf_nrows / Rf_ncols from `dim`; lists hold SEXPs;
 * external pointers hold an address and a finalizer (run by mini_r_reset).
 */
#include <setjmp.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../shim/qa_r_api.h"

struct SEXPREC {
    SEXPTYPE type;        /* NILSXP, LGLSXP, INTSXP, REALSXP, STRSXP, VECSXP, RAWSXP; 9 = CHARSXP; 22 = EXTPTRSXP; 1 = SYMSXP */
    R_xlen_t n;
    void *data;           /* int / double / Rbyte / SEXP[] / char[] / the external address */
    SEXP names, dim;
    R_CFinalizer_t fin;
    struct SEXPREC *next; /* allocation list */
};
#define CHARSXP 9
#define EXTPTRSXP 22
#define SYMSXP 1

static struct SEXPREC nil_rec = {NILSXP, 0, NULL, NULL, NULL, NULL, NULL};
static struct SEXPREC names_sym = {SYMSXP, 0, NULL, NULL, NULL, NULL, NULL}, dim_sym = {SYMSXP, 0, NULL, NULL, NULL, NULL, NULL},
                      dimnames_sym = {SYMSXP, 0, NULL, NULL, NULL, NULL, NULL};
SEXP R_NilValue = &nil_rec, R_NamesSymbol = &names_sym, R_DimSymbol = &dim_sym, R_DimNamesSymbol = &dimnames_sym;
double R_NaReal;

static struct SEXPREC *g_all = NULL;
static jmp_buf g_jmp;
static int g_jmp_set = 0;
static char g_error[1024];
static const R_CallMethodDef *g_calls = NULL;
static double *g_unif = NULL;
static size_t g_n_unif = 0, g_at_unif = 0;
static int g_rng_open = 0, g_rng_violations = 0;

static size_t elt_size(SEXPTYPE t) {
    switch (t) {
    case LGLSXP: case INTSXP: return sizeof(int);
    case REALSXP: return sizeof(double);
    case RAWSXP: case CHARSXP: return 1;
    case STRSXP: case VECSXP: return sizeof(SEXP);
    default: return 0;
    }
}

static SEXP new_obj(SEXPTYPE type, R_xlen_t n) {
    struct SEXPREC *s = (struct SEXPREC *)calloc(1, sizeof *s);
    s->type = type;
    s->n = n;
    s->names = s->dim = R_NilValue;
    const size_t es = elt_size(type);
    if (es) s->data = calloc((size_t)(n > 0 ? n : 1) + (type == CHARSXP ? 1 : 0), es);
    if (type == STRSXP || type == VECSXP)
        for (R_xlen_t i = 0; i < n; i++) ((SEXP *)s->data)[i] = R_NilValue;
    s->next = g_all;
    g_all = s;
    return s;
}

double *REAL(SEXP x) { return (double *)x->data; }
int *INTEGER(SEXP x) { return (int *)x->data; }
int *LOGICAL(SEXP x) { return (int *)x->data; }
Rbyte *RAW(SEXP x) { return (Rbyte *)x->data; }
SEXP VECTOR_ELT(SEXP x, R_xlen_t i) { return ((SEXP *)x->data)[i]; }
SEXP SET_VECTOR_ELT(SEXP x, R_xlen_t i, SEXP v) { ((SEXP *)x->data)[i] = v; return v; }
SEXP STRING_ELT(SEXP x, R_xlen_t i) { return ((SEXP *)x->data)[i]; }
void SET_STRING_ELT(SEXP x, R_xlen_t i, SEXP v) { ((SEXP *)x->data)[i] = v; }
const char *CHAR(SEXP x) { return (const char *)x->data; }
int TYPEOF(SEXP x) { return (int)x->type; }
R_xlen_t Rf_xlength(SEXP x) { return x->n; }
int Rf_length(SEXP x) { return (int)x->n; }
int Rf_nrows(SEXP x) { return x->dim != R_NilValue ? INTEGER(x->dim)[0] : (int)x->n; }   /* (R: a plain vector has length rows) */
int Rf_ncols(SEXP x) { return x->dim != R_NilValue && x->dim->n >= 2 ? INTEGER(x->dim)[1] : 1; }
SEXP Rf_allocVector(SEXPTYPE type, R_xlen_t n) { return new_obj(type, n); }
SEXP Rf_allocMatrix(SEXPTYPE type, int nrow, int ncol) {
    SEXP m = new_obj(type, (R_xlen_t)nrow * ncol);
    m->dim = new_obj(INTSXP, 2);
    INTEGER(m->dim)[0] = nrow;
    INTEGER(m->dim)[1] = ncol;
    return m;
}
SEXP Rf_protect(SEXP x) { return x; }
void Rf_unprotect(int n) { (void)n; }
SEXP Rf_getAttrib(SEXP x, SEXP name) {
    if (name == R_NamesSymbol) return x->names;
    if (name == R_DimSymbol) return x->dim;
    return R_NilValue;
}
SEXP Rf_setAttrib(SEXP x, SEXP name, SEXP val) {
    if (name == R_NamesSymbol) x->names = val;
    else if (name == R_DimSymbol) x->dim = val;
    return val;
}
SEXP Rf_mkChar(const char *s) {
    SEXP c = new_obj(CHARSXP, (R_xlen_t)strlen(s));
    memcpy(c->data, s, strlen(s) + 1);
    return c;
}
SEXP Rf_mkString(const char *s) {
    SEXP v = new_obj(STRSXP, 1);
    SET_STRING_ELT(v, 0, Rf_mkChar(s));
    return v;
}
SEXP Rf_ScalarLogical(int x) { SEXP v = new_obj(LGLSXP, 1); LOGICAL(v)[0] = x; return v; }
SEXP Rf_ScalarInteger(int x) { SEXP v = new_obj(INTSXP, 1); INTEGER(v)[0] = x; return v; }
int Rf_asInteger(SEXP x) {
    if (x->n < 1) return 0;
    switch (x->type) {
    case LGLSXP: case INTSXP: return INTEGER(x)[0];
    case REALSXP: return (int)REAL(x)[0];
    case RAWSXP: return RAW(x)[0];
    default: return 0;
    }
}
int Rf_asLogical(SEXP x) { return Rf_asInteger(x) != 0; }
double Rf_asReal(SEXP x) {
    if (x->n < 1) return R_NaReal;
    switch (x->type) {
    case LGLSXP: case INTSXP: return (double)INTEGER(x)[0];
    case REALSXP: return REAL(x)[0];
    default: return R_NaReal;
    }
}
void Rf_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof g_error, fmt, ap);
    va_end(ap);
    if (g_jmp_set) longjmp(g_jmp, 1);
    fprintf(stderr, "mini_r: Rf_error outside a .Call: %s\n", g_error);
    abort();
}
void Rf_warning(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fputc('\n', stderr);
}
/* the generator: a sequence loaded by the test; drawing outside GetRNGstate / PutRNGstate is counted (R requires the pair) */
void GetRNGstate(void) { g_rng_open += 1; }
void PutRNGstate(void) { g_rng_open -= 1; }
double unif_rand(void) {
    if (g_rng_open <= 0) g_rng_violations += 1;
    if (g_at_unif >= g_n_unif) Rf_error("mini_r: unif_rand() beyond the %zu loaded values", g_n_unif);
    return g_unif[g_at_unif++];
}
SEXP R_MakeExternalPtr(void *p, SEXP tag, SEXP prot) {
    (void)tag; (void)prot;
    SEXP s = new_obj(EXTPTRSXP, 0);
    s->data = p;
    return s;
}
void *R_ExternalPtrAddr(SEXP s) { return s->data; }
void R_ClearExternalPtr(SEXP s) { s->data = NULL; }
void R_RegisterCFinalizerEx(SEXP s, R_CFinalizer_t fun, Rboolean onexit) { (void)onexit; s->fin = fun; }
void R_PreserveObject(SEXP x) { (void)x; }
void R_ReleaseObject(SEXP x) { (void)x; }
int R_registerRoutines(DllInfo *info, const void *c, const R_CallMethodDef *call, const void *f, const void *e) {
    (void)info; (void)c; (void)f; (void)e;
    g_calls = call;
    return 1;
}
Rboolean R_useDynamicSymbols(DllInfo *info, Rboolean value) { (void)info; return value; }

/* ---- what the test drives ------------------------------------------------------------------------------------------------ */

void R_init_quilt_amd_shim(DllInfo *dll);

void mini_r_init(void) {
    union { unsigned long long u; double d; } na = {0x7FF00000000007A2ull};   /* R's NA_real_ payload (1954) */
    R_NaReal = na.d;
    if (!g_calls) R_init_quilt_amd_shim(NULL);
}
/* frees every object made since the last reset (external pointers: their finalizers first) */
void mini_r_reset(void) {
    for (struct SEXPREC *s = g_all; s; s = s->next)
        if (s->type == EXTPTRSXP && s->fin && s->data) s->fin(s);
    while (g_all) {
        struct SEXPREC *s = g_all;
        g_all = s->next;
        if (s->type != EXTPTRSXP) free(s->data);
        free(s);
    }
    free(g_unif);
    g_unif = NULL;
    g_n_unif = g_at_unif = 0;
}
void mini_r_load_unif(const double *u, size_t n) {
    free(g_unif);
    g_unif = (double *)malloc(sizeof(double) * (n > 0 ? n : 1));
    memcpy(g_unif, u, sizeof(double) * n);
    g_n_unif = n;
    g_at_unif = 0;
}
size_t mini_r_unif_drawn(void) { return g_at_unif; }
int mini_r_rng_violations(void) { return g_rng_violations; }
const char *mini_r_last_error(void) { return g_error; }
int mini_r_arity(const char *name) {
    for (const R_CallMethodDef *c = g_calls; c && c->name; c++)
        if (strcmp(c->name, name) == 0) return c->numArgs;
    return -1;
}
SEXP mini_r_nil(void) { return R_NilValue; }
void mini_r_set_names(SEXP x, int n, const char **names) {
    SEXP nm = new_obj(STRSXP, n);
    for (int i = 0; i < n; i++) SET_STRING_ELT(nm, i, Rf_mkChar(names[i]));
    x->names = nm;
}
void mini_r_set_dim(SEXP x, int nrow, int ncol) {
    x->dim = new_obj(INTSXP, 2);
    INTEGER(x->dim)[0] = nrow;
    INTEGER(x->dim)[1] = ncol;
}
void mini_r_set_dim3(SEXP x, int a, int b, int c) {
    x->dim = new_obj(INTSXP, 3);
    INTEGER(x->dim)[0] = a;
    INTEGER(x->dim)[1] = b;
    INTEGER(x->dim)[2] = c;
}
void *mini_r_data(SEXP x) { return x->data; }

/* `.Call(name, ...)`: the routine registered under `name`, refused unless it takes exactly n arguments (R: "Incorrect number
 * of arguments").  Returns NULL when the routine raised an R error (text: mini_r_last_error). */
typedef SEXP (*fn0)(void);
typedef SEXP (*fn3)(SEXP, SEXP, SEXP);
typedef SEXP (*fn6)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
typedef SEXP (*fn15)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
#define A8(o) a[o], a[o + 1], a[o + 2], a[o + 3], a[o + 4], a[o + 5], a[o + 6], a[o + 7]
typedef SEXP (*fn38)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP,
                     SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
typedef SEXP (*fn63)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP,
                     SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP,
                     SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP,
                     SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
SEXP mini_r_dotcall(const char *name, int n, SEXP *a) {
    const R_CallMethodDef *volatile c = g_calls;   /* (volatile: read again after the longjmp) */
    for (; c && c->name; c++)
        if (strcmp(c->name, name) == 0) break;
    if (!c || !c->name) { snprintf(g_error, sizeof g_error, "\"%s\" not available for .Call()", name); return NULL; }
    if (c->numArgs != n) {
        snprintf(g_error, sizeof g_error, "Incorrect number of arguments (%d), expecting %d for '%s'", n, c->numArgs, name);
        return NULL;
    }
    g_error[0] = 0;
    g_jmp_set = 1;
    if (setjmp(g_jmp)) {
        g_jmp_set = 0;
        g_rng_open = 0;
        return NULL;
    }
    SEXP out = NULL;
    switch (n) {
    case 0: out = ((fn0)c->fun)(); break;
    case 3: out = ((fn3)c->fun)(a[0], a[1], a[2]); break;
    case 6: out = ((fn6)c->fun)(a[0], a[1], a[2], a[3], a[4], a[5]); break;
    case 15: out = ((fn15)c->fun)(A8(0), a[8], a[9], a[10], a[11], a[12], a[13], a[14]); break;
    case 38: out = ((fn38)c->fun)(A8(0), A8(8), A8(16), A8(24), a[32], a[33], a[34], a[35], a[36], a[37]); break;
    case 63: out = ((fn63)c->fun)(A8(0), A8(8), A8(16), A8(24), A8(32), A8(40), A8(48), a[56], a[57], a[58], a[59], a[60], a[61], a[62]); break;
    default: snprintf(g_error, sizeof g_error, "mini_r: no trampoline for %d arguments", n); out = NULL;
    }
    g_jmp_set = 0;
    return out;
}

/*
 * mini_r.c -- TEST INFRASTRUCTURE: a small runtime behind the subset of R's C API that shim/qa_r_api.h declares, so that
 * shim/quilt_amd_shim.c can be EXECUTED (not only type-checked) on a machine without R: tests/test_shim_gpu.py builds R-shaped
 * objects through it, calls the registered `.Call` routines by name the way R's `.Call` does (look-up in the table
 * R_registerRoutines received, arity check) and reads the results back.  It is not R: no NA handling beyond R_NaReal, `unif_rand()` serves a sequence the test loads beforehand, Rf_error() unwinds
 * to the caller of mini_r_dotcall with the message kept.  Semantics the shim relies on and that are kept: vectors carry a type,
 * a length and `names` / `dim` attributes; matrices are column-major with Rf_nrows / Rf_ncols from `dim`; lists hold SEXPs;
 * external pointers hold an address and a finalizer (run by mini_r_reset).
 *
 * GARBAGE COLLECTION is emulated in its strictest form -- R's gctorture(TRUE): inside a `.Call`, EVERY allocation first "collects"
 * every object made during that call that is not reachable from the PROTECT stack (or a preserved object).  Collected objects are
 * not freed but poisoned: any later use of one through the API (REAL, VECTOR_ELT, SET_VECTOR_ELT as container or as element, ...)
 * is counted as a violation with the place it happened (mini_r_gc_violations / mini_r_gc_report) -- the bug class "an object left
 * unprotected across an allocation".  The PROTECT stack must be empty again when a routine returns normally
 * (mini_r_protect_imbalance: R's "stack imbalance" warning).  Arguments and objects the test made outside a call belong to the
 * caller and are never collected.
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
    int epoch;            /* the .Call during which it was made (0: by the test, outside any call) */
    int dead, mark;       /* collected by the emulated gctorture; reached in the current mark phase */
};
#define CHARSXP 9
#define EXTPTRSXP 22
#define SYMSXP 1

static struct SEXPREC nil_rec = {NILSXP, 0, NULL, NULL, NULL, NULL, NULL, 0, 0, 0};
static struct SEXPREC names_sym = {SYMSXP, 0, NULL, NULL, NULL, NULL, NULL, 0, 0, 0}, dim_sym = {SYMSXP, 0, NULL, NULL, NULL, NULL, NULL, 0, 0, 0},
                      dimnames_sym = {SYMSXP, 0, NULL, NULL, NULL, NULL, NULL, 0, 0, 0};
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
/* the emulated gctorture */
static int g_epoch = 0, g_in_call = 0, g_gc_violations = 0, g_protect_imbalance = 0;
static char g_gc_report[512];
static SEXP *g_pstack = NULL;
static int g_pdepth = 0, g_pcap = 0;
static SEXP *g_call_objs = NULL;   /* objects made during the current call */
static int g_n_call_objs = 0, g_cap_call_objs = 0;
static SEXP *g_preserved = NULL;
static int g_n_preserved = 0, g_cap_preserved = 0;
static const char *g_routine = "";
static SEXP *g_args = NULL;        /* the current call's arguments: the caller protects them, and with them whatever the routine */
static int g_n_args = 0;           /* stores INTO them (best_haps_stuff_list[[i]] <- ...) */

static void gc_mark(SEXP x) {
    if (!x || x->mark || x->type == NILSXP || x->type == SYMSXP) return;
    x->mark = 1;
    if (x->names) gc_mark(x->names);
    if (x->dim) gc_mark(x->dim);
    if (x->type == STRSXP || x->type == VECSXP)
        for (R_xlen_t i = 0; i < x->n; i++) gc_mark(((SEXP *)x->data)[i]);
}
static void gc_unmark(SEXP x) {
    if (!x || !x->mark) return;
    x->mark = 0;
    if (x->names) gc_unmark(x->names);
    if (x->dim) gc_unmark(x->dim);
    if (x->type == STRSXP || x->type == VECSXP)
        for (R_xlen_t i = 0; i < x->n; i++) gc_unmark(((SEXP *)x->data)[i]);
}
/* gctorture: everything made in this call and not reachable from the PROTECT stack or a preserved object is collected NOW */
static void gc_collect(void) {
    if (!g_in_call) return;
    for (int i = 0; i < g_pdepth; i++) gc_mark(g_pstack[i]);
    for (int i = 0; i < g_n_preserved; i++) gc_mark(g_preserved[i]);
    for (int i = 0; i < g_n_args; i++) gc_mark(g_args[i]);
    for (int i = 0; i < g_n_call_objs; i++)
        if (!g_call_objs[i]->mark) g_call_objs[i]->dead = 1;
    for (int i = 0; i < g_pdepth; i++) gc_unmark(g_pstack[i]);
    for (int i = 0; i < g_n_preserved; i++) gc_unmark(g_preserved[i]);
    for (int i = 0; i < g_n_args; i++) gc_unmark(g_args[i]);
}
/* every API entry that touches an object passes through here */
static SEXP use(SEXP x, const char *where) {
    if (x && x->dead) {
        if (g_gc_violations++ == 0)
            snprintf(g_gc_report, sizeof g_gc_report, "%s: %s on an object (type %u, length %ld) that a collection at an earlier allocation "
                     "would have freed -- it was not PROTECTed (nor reachable from a protected object) across that allocation",
                     g_routine, where, x->type, (long)x->n);
    }
    return x;
}

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
    gc_collect();
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
    if (g_in_call) {
        s->epoch = g_epoch;
        if (g_n_call_objs == g_cap_call_objs) {
            g_cap_call_objs = g_cap_call_objs ? 2 * g_cap_call_objs : 1024;
            g_call_objs = (SEXP *)realloc(g_call_objs, sizeof(SEXP) * (size_t)g_cap_call_objs);
        }
        g_call_objs[g_n_call_objs++] = s;
    }
    return s;
}

double *REAL(SEXP x) { return (double *)use(x, "REAL")->data; }
int *INTEGER(SEXP x) { return (int *)use(x, "INTEGER")->data; }
int *LOGICAL(SEXP x) { return (int *)use(x, "LOGICAL")->data; }
Rbyte *RAW(SEXP x) { return (Rbyte *)use(x, "RAW")->data; }
SEXP VECTOR_ELT(SEXP x, R_xlen_t i) { return ((SEXP *)use(x, "VECTOR_ELT")->data)[i]; }
SEXP SET_VECTOR_ELT(SEXP x, R_xlen_t i, SEXP v) { ((SEXP *)use(x, "SET_VECTOR_ELT (the list)")->data)[i] = use(v, "SET_VECTOR_ELT (the element)"); return v; }
SEXP STRING_ELT(SEXP x, R_xlen_t i) { return ((SEXP *)use(x, "STRING_ELT")->data)[i]; }
void SET_STRING_ELT(SEXP x, R_xlen_t i, SEXP v) { ((SEXP *)use(x, "SET_STRING_ELT (the vector)")->data)[i] = use(v, "SET_STRING_ELT (the string)"); }
const char *CHAR(SEXP x) { return (const char *)use(x, "CHAR")->data; }
int TYPEOF(SEXP x) { return (int)use(x, "TYPEOF")->type; }
R_xlen_t Rf_xlength(SEXP x) { return use(x, "Rf_xlength")->n; }
int Rf_length(SEXP x) { return (int)use(x, "Rf_length")->n; }
int Rf_nrows(SEXP x) { use(x, "Rf_nrows"); return x->dim != R_NilValue ? ((int *)x->dim->data)[0] : (int)x->n; }   /* (R: a plain vector has length rows) */
int Rf_ncols(SEXP x) { use(x, "Rf_ncols"); return x->dim != R_NilValue && x->dim->n >= 2 ? ((int *)x->dim->data)[1] : 1; }
SEXP Rf_allocVector(SEXPTYPE type, R_xlen_t n) { return new_obj(type, n); }
SEXP Rf_protect(SEXP x) {
    use(x, "PROTECT");
    if (g_pdepth == g_pcap) {
        g_pcap = g_pcap ? 2 * g_pcap : 256;
        g_pstack = (SEXP *)realloc(g_pstack, sizeof(SEXP) * (size_t)g_pcap);
    }
    g_pstack[g_pdepth++] = x;
    return x;
}
void Rf_unprotect(int n) {
    g_pdepth -= n;
    if (g_pdepth < 0) { g_pdepth = 0; g_protect_imbalance++; }   /* R: "unprotect(): only 0 protected items" is an error */
}
/* (the runtime's own multi-allocation helpers protect their intermediate objects, as R's do) */
SEXP Rf_allocMatrix(SEXPTYPE type, int nrow, int ncol) {
    SEXP m = Rf_protect(new_obj(type, (R_xlen_t)nrow * ncol));
    m->dim = new_obj(INTSXP, 2);
    ((int *)m->dim->data)[0] = nrow;
    ((int *)m->dim->data)[1] = ncol;
    Rf_unprotect(1);
    return m;
}
SEXP Rf_getAttrib(SEXP x, SEXP name) {
    use(x, "Rf_getAttrib");
    if (name == R_NamesSymbol) return x->names;
    if (name == R_DimSymbol) return x->dim;
    return R_NilValue;
}
SEXP Rf_setAttrib(SEXP x, SEXP name, SEXP val) {
    use(x, "Rf_setAttrib (the object)");
    use(val, "Rf_setAttrib (the value)");
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
    SEXP v = Rf_protect(new_obj(STRSXP, 1));
    SET_STRING_ELT(v, 0, Rf_mkChar(s));
    Rf_unprotect(1);
    return v;
}
SEXP Rf_ScalarLogical(int x) { SEXP v = new_obj(LGLSXP, 1); LOGICAL(v)[0] = x; return v; }
SEXP Rf_ScalarInteger(int x) { SEXP v = new_obj(INTSXP, 1); INTEGER(v)[0] = x; return v; }
int Rf_asInteger(SEXP x) {
    use(x, "Rf_asInteger");
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
    use(x, "Rf_asReal");
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
void *R_ExternalPtrAddr(SEXP s) { return use(s, "R_ExternalPtrAddr")->data; }
void R_ClearExternalPtr(SEXP s) { use(s, "R_ClearExternalPtr")->data = NULL; }
void R_RegisterCFinalizerEx(SEXP s, R_CFinalizer_t fun, Rboolean onexit) { (void)onexit; use(s, "R_RegisterCFinalizerEx")->fin = fun; }
void R_PreserveObject(SEXP x) {
    use(x, "R_PreserveObject");
    if (g_n_preserved == g_cap_preserved) {
        g_cap_preserved = g_cap_preserved ? 2 * g_cap_preserved : 16;
        g_preserved = (SEXP *)realloc(g_preserved, sizeof(SEXP) * (size_t)g_cap_preserved);
    }
    g_preserved[g_n_preserved++] = x;
}
void R_ReleaseObject(SEXP x) {
    for (int i = 0; i < g_n_preserved; i++)
        if (g_preserved[i] == x) { g_preserved[i] = g_preserved[--g_n_preserved]; return; }
}
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
    g_n_preserved = 0;
    g_pdepth = 0;
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
int mini_r_gc_violations(void) { return g_gc_violations; }          /* uses of objects the emulated gctorture had collected */
const char *mini_r_gc_report(void) { return g_gc_report; }          /* the first of them */
int mini_r_protect_imbalance(void) { return g_protect_imbalance; }  /* routines that returned with a non-empty PROTECT stack (or popped too much) */
const char *mini_r_last_error(void) { return g_error; }
int mini_r_arity(const char *name) {
    for (const R_CallMethodDef *c = g_calls; c && c->name; c++)
        if (strcmp(c->name, name) == 0) return c->numArgs;
    return -1;
}
/* Self-test of the emulated gctorture: the two classic mistakes, made on purpose inside a pretend `.Call`, must each be caught --
 * (1) an object used after an allocation it was not protected across, (2) a routine that leaves the PROTECT stack unbalanced --
 * and the correct forms must pass.  Returns a bit mask of what was detected (3 = both) and leaves the counters as it found them. */
int mini_r_gc_selftest(void) {
    const int v0 = g_gc_violations, i0 = g_protect_imbalance;
    char keep[sizeof g_gc_report];
    memcpy(keep, g_gc_report, sizeof keep);
    int found = 0;
    g_epoch += 1; g_in_call = 1; g_n_call_objs = 0; g_pdepth = 0; g_routine = "selftest";
    {   /* correct: a protected across b's allocation; an element reachable from a protected list */
        SEXP a = Rf_protect(Rf_allocVector(INTSXP, 4));
        SEXP l = Rf_protect(Rf_allocVector(VECSXP, 1));
        SET_VECTOR_ELT(l, 0, Rf_allocVector(REALSXP, 2));
        SEXP b = Rf_allocVector(REALSXP, 2);
        INTEGER(a)[0] = 1; REAL(VECTOR_ELT(l, 0))[0] = REAL(b)[0];
        Rf_unprotect(2);
        /* ... and an object stored INTO an argument (the caller's list) lives on without a PROTECT of its own */
        SEXP arg = Rf_allocVector(VECSXP, 1);
        arg->epoch = 0;
        g_n_call_objs -= 1;   /* (pretend the caller made it) */
        SEXP args1[1] = {arg};
        g_args = args1; g_n_args = 1;
        SET_VECTOR_ELT(arg, 0, Rf_allocVector(INTSXP, 3));
        (void)Rf_allocVector(REALSXP, 1);
        INTEGER(VECTOR_ELT(arg, 0))[0] = 7;
        g_n_args = 0;
        if (g_gc_violations != v0) found |= 4;   /* a false positive */
    }
    {   /* wrong: a is not protected while b is allocated */
        SEXP a = Rf_allocVector(INTSXP, 4);
        SEXP b = Rf_allocVector(REALSXP, 2);
        (void)b;
        INTEGER(a)[0] = 1;
        if (g_gc_violations > v0) found |= 1;
    }
    g_in_call = 0;
    {   /* wrong: one PROTECT too many at return */
        g_in_call = 1; g_pdepth = 0;
        (void)Rf_protect(Rf_allocVector(INTSXP, 1));
        g_in_call = 0;
        if (g_pdepth != 0) found |= 2;
        g_pdepth = 0;
    }
    g_gc_violations = v0; g_protect_imbalance = i0;
    memcpy(g_gc_report, keep, sizeof keep);
    return found;
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
    g_epoch += 1;
    g_in_call = 1;
    g_routine = c->name;
    g_n_call_objs = 0;
    g_pdepth = 0;
    g_args = a;
    g_n_args = n;
    if (setjmp(g_jmp)) {
        g_jmp_set = 0;
        g_rng_open = 0;
        g_in_call = 0;
        g_n_args = 0;
        g_pdepth = 0;   /* (R unwinds the PROTECT stack on an error) */
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
    g_in_call = 0;
    g_n_args = 0;
    if (g_pdepth != 0) {   /* R: "Warning: stack imbalance in '.Call'" */
        if (g_protect_imbalance++ == 0 && !g_gc_report[0])
            snprintf(g_gc_report, sizeof g_gc_report, "%s returned with %d object(s) left on the PROTECT stack", c->name, g_pdepth);
        g_pdepth = 0;
    }
    if (out) use(out, "the routine's return value");
    return out;
}

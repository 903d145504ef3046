/*
 * qa_r_api.h -- DECLARATION-ONLY subset of R's C API (Rinternals.h / R_ext/Rdynload.h / R_ext/Random.h), used only to
 * type-check quilt_amd_shim.c (`make -C shim check`, gcc -fsyntax-only) on machines without R.  A real build includes R's own
 * headers (`make -C shim` with R installed defines QA_HAVE_R) and nothing here is linked; the one thing that implements these
 * declarations is the TEST runtime tests/c/mini_r.c, under which the repository's tests execute the shim.  Signatures follow the
 * R 4.x headers.
 */
#ifndef QA_R_API_H
#define QA_R_API_H
#include <stddef.h>
typedef struct SEXPREC *SEXP;
typedef ptrdiff_t R_xlen_t;
typedef unsigned char Rbyte;
typedef unsigned int SEXPTYPE;
typedef void *(*DL_FUNC)(void);
typedef struct { const char *name; DL_FUNC fun; int numArgs; } R_CallMethodDef;
typedef struct _DllInfo DllInfo;
typedef void (*R_CFinalizer_t)(SEXP);
typedef enum { FALSE = 0, TRUE } Rboolean;
#define NILSXP 0
#define LGLSXP 10
#define INTSXP 13
#define REALSXP 14
#define STRSXP 16
#define VECSXP 19
#define RAWSXP 24
extern SEXP R_NilValue, R_NamesSymbol, R_DimSymbol, R_DimNamesSymbol;
extern double R_NaReal;
#define NA_REAL R_NaReal
double *REAL(SEXP x);
int *INTEGER(SEXP x);
int *LOGICAL(SEXP x);
Rbyte *RAW(SEXP x);
SEXP VECTOR_ELT(SEXP x, R_xlen_t i);
SEXP SET_VECTOR_ELT(SEXP x, R_xlen_t i, SEXP v);
SEXP STRING_ELT(SEXP x, R_xlen_t i);
void SET_STRING_ELT(SEXP x, R_xlen_t i, SEXP v);
const char *CHAR(SEXP x);
int TYPEOF(SEXP x);
R_xlen_t Rf_xlength(SEXP x);
int Rf_length(SEXP x);
int Rf_nrows(SEXP x);
int Rf_ncols(SEXP x);
SEXP Rf_allocVector(SEXPTYPE type, R_xlen_t n);
SEXP Rf_allocMatrix(SEXPTYPE type, int nrow, int ncol);
SEXP Rf_protect(SEXP x);
void Rf_unprotect(int n);
#define PROTECT(x) Rf_protect(x)
#define UNPROTECT(n) Rf_unprotect(n)
SEXP Rf_getAttrib(SEXP x, SEXP name);
SEXP Rf_setAttrib(SEXP x, SEXP name, SEXP val);
SEXP Rf_mkChar(const char *s);
SEXP Rf_mkString(const char *s);
SEXP Rf_ScalarLogical(int x);
SEXP Rf_ScalarInteger(int x);
int Rf_asInteger(SEXP x);
int Rf_asLogical(SEXP x);
double Rf_asReal(SEXP x);
void Rf_error(const char *fmt, ...) __attribute__((noreturn));
void Rf_warning(const char *fmt, ...);
void GetRNGstate(void);
void PutRNGstate(void);
double unif_rand(void);
SEXP R_MakeExternalPtr(void *p, SEXP tag, SEXP prot);
void *R_ExternalPtrAddr(SEXP s);
void R_ClearExternalPtr(SEXP s);
void R_RegisterCFinalizerEx(SEXP s, R_CFinalizer_t fun, Rboolean onexit);
void R_PreserveObject(SEXP x);
void R_ReleaseObject(SEXP x);
int R_registerRoutines(DllInfo *info, const void *c, const R_CallMethodDef *call, const void *f, const void *e);
Rboolean R_useDynamicSymbols(DllInfo *info, Rboolean value);
#endif

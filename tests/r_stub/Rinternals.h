/* Minimal stand-in for the subset of R's C API that clonealign_amd/r_shim/src/clonealign_hip_shim.c uses.
 * TEST INFRASTRUCTURE ONLY (this image has no R): it lets the suite compile the shim and drive C_clonealign_fit from a
 * C harness (tests/r_stub/harness.c).  Semantics follow "Writing R Extensions" section 5.9/6 for the calls listed; nothing here is
 * taken from R's sources. */
#ifndef RSTUB_RINTERNALS_H
#define RSTUB_RINTERNALS_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct rstub_sexp* SEXP;
typedef ptrdiff_t R_xlen_t;
typedef enum { FALSE = 0, TRUE = 1 } Rboolean;
#define NILSXP 0
#define LGLSXP 10
#define INTSXP 13
#define REALSXP 14
#define STRSXP 16
#define VECSXP 19
extern SEXP R_NilValue;
int Rf_nrows(SEXP);
int Rf_ncols(SEXP);
int Rf_asInteger(SEXP);
double Rf_asReal(SEXP);
int Rf_isNull(SEXP);
int Rf_isInteger(SEXP);
double* REAL(SEXP);
int* INTEGER(SEXP);
int* LOGICAL(SEXP);
R_xlen_t XLENGTH(SEXP);
SEXP Rf_allocVector(int type, R_xlen_t n);
SEXP Rf_allocMatrix(int type, int nrow, int ncol);
SEXP Rf_mkNamed(int type, const char** names);
SEXP SET_VECTOR_ELT(SEXP x, R_xlen_t i, SEXP v);
SEXP VECTOR_ELT(SEXP x, R_xlen_t i);
SEXP Rf_xlengthgets(SEXP x, R_xlen_t n);
SEXP Rf_protect(SEXP);
void Rf_unprotect(int);
#define PROTECT(x) Rf_protect(x)
#define UNPROTECT(n) Rf_unprotect(n)
void Rf_error(const char* fmt, ...) __attribute__((noreturn));
char* R_alloc(size_t n, int size);
typedef void* (*DL_FUNC)(void);
typedef struct { const char* name; DL_FUNC fun; int numArgs; } R_CallMethodDef;
typedef struct rstub_dllinfo DllInfo;
int R_registerRoutines(DllInfo*, const void*, const R_CallMethodDef*, const void*, const void*);
Rboolean R_useDynamicSymbols(DllInfo*, Rboolean);
Rboolean R_ToplevelExec(void (*fun)(void*), void* data);
/* harness side */
SEXP rstub_real_matrix(const double* data, int nrow, int ncol);   /* copies */
SEXP rstub_int_matrix(const int* data, int nrow, int ncol);
SEXP rstub_real_vector(const double* data, R_xlen_t n);
SEXP rstub_list(R_xlen_t n);   /* VECSXP of R_NilValue */
SEXP rstub_int_vector(const int* data, R_xlen_t n);
SEXP rstub_scalar_int(int v);
SEXP rstub_scalar_real(double v);
const char* rstub_name(SEXP list, R_xlen_t i);
void rstub_free_all(void);
void rstub_begin_call(void);                      /* arguments built: from here on every allocation collects (gctorture) */
int rstub_violations(char* msg, int cap);         /* memory-rule violations seen since the last rstub_free_all (use of a collected object, UNPROTECT underflow) */
int rstub_protect_depth(void);                    /* must be 0 when a .Call returns */
extern int rstub_interrupt_after;   /* R_CheckUserInterrupt() "sees Ctrl-C" on its n-th call (0 = never) */
#ifdef __cplusplus
}
#endif
#endif

/* Fake R runtime behind tests/r_stub/Rinternals.h (test infrastructure only). */
#define _POSIX_C_SOURCE 200809L   /* strdup */
#include <setjmp.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "Rinternals.h"
#include "R_ext/Utils.h"

/* Round 6: the stub keeps R's two memory rules, so that the shim's PROTECT discipline is TESTED and not just compiled:
 *   - a protect stack (Rf_protect / Rf_unprotect); a .Call must return with the depth it was entered with ("stack imbalance");
 *   - "gctorture": EVERY allocation collects -- an object that is neither on the protect stack, nor an argument of the call (R protects
 *     those), nor reachable from one of these through list elements, is dead: its payload is overwritten and any later access through
 *     the API is recorded as a violation (rstub_violations()).  The object allocated last is alive until the next allocation, as in R.
 * R_alloc memory lives until the call returns (R frees it then), also on the error path. */
struct rstub_sexp {
  int type; R_xlen_t n; int nrow, ncol;   /* nrow = -1: plain vector */
  void* data; const char** names; struct rstub_sexp* next;
  int is_arg, dead, mark;
};
static struct rstub_sexp nil_obj = {NILSXP, 0, -1, -1, NULL, NULL, NULL, 1, 0, 0};
#define PSTACK_MAX 256
static SEXP pstack[PSTACK_MAX]; static int pdepth = 0;
static int args_phase = 1;          /* objects made by the harness before the call are the call's arguments */
static int n_violations = 0; static char violation_msg[256];
static void violation(const char* m) { if (!n_violations) { strncpy(violation_msg, m, sizeof(violation_msg) - 1); } ++n_violations; }
static void mark_from(SEXP x) {
  if (!x || x->mark) return;
  x->mark = 1;
  if (x->type == VECSXP) for (R_xlen_t i = 0; i < x->n; ++i) mark_from(((SEXP*)x->data)[i]);
}
static struct rstub_sexp* all_objs;
static void collect(void) {          /* what R's collector would be entitled to do right now */
  for (struct rstub_sexp* o = all_objs; o; o = o->next) o->mark = 0;
  nil_obj.mark = 0;
  for (int i = 0; i < pdepth; ++i) mark_from(pstack[i]);
  for (struct rstub_sexp* o = all_objs; o; o = o->next) if (o->is_arg) mark_from(o);
  for (struct rstub_sexp* o = all_objs; o; o = o->next)
    if (!o->mark && !o->dead) {
      const size_t esz = o->type == REALSXP ? sizeof(double) : (o->type == INTSXP || o->type == LGLSXP) ? sizeof(int) : sizeof(SEXP);
      o->dead = 1;
      if (o->type != VECSXP) memset(o->data, 0xA5, (size_t)(o->n > 0 ? o->n : 1) * esz);
    }
}
static SEXP live(SEXP x, const char* what) { if (x && x->dead) violation(what); return x; }
int rstub_violations(char* msg, int cap) {
  if (msg && cap > 0) { strncpy(msg, n_violations ? violation_msg : "", (size_t)cap - 1); msg[cap - 1] = 0; }
  return n_violations;
}
int rstub_protect_depth(void) { return pdepth; }
void rstub_begin_call(void) { args_phase = 0; }   /* the harness has built the arguments; what is allocated from now on is the callee's */
SEXP R_NilValue = &nil_obj;
static void* all_ralloc[64]; static int n_ralloc = 0;
int rstub_interrupt_after = 0;
static int n_checks = 0;
jmp_buf rstub_error_jmp; int rstub_error_armed = 0; char rstub_error_msg[1024];
static jmp_buf* toplevel_jmp = NULL;

static SEXP mk(int type, R_xlen_t n, int nrow, int ncol) {
  if (!args_phase) collect();
  struct rstub_sexp* s = (struct rstub_sexp*)calloc(1, sizeof(*s));
  s->is_arg = args_phase;
  const size_t esz = type == REALSXP ? sizeof(double) : (type == INTSXP || type == LGLSXP) ? sizeof(int) : sizeof(SEXP);
  s->type = type; s->n = n; s->nrow = nrow; s->ncol = ncol;
  s->data = calloc((size_t)(n > 0 ? n : 1), esz);
  s->next = all_objs; all_objs = s;
  return s;
}
int Rf_nrows(SEXP x) { return x->nrow >= 0 ? x->nrow : (int)x->n; }
int Rf_ncols(SEXP x) { return x->nrow >= 0 ? x->ncol : 1; }
int Rf_asInteger(SEXP x) { return x->type == INTSXP ? ((int*)x->data)[0] : (int)((double*)x->data)[0]; }
double Rf_asReal(SEXP x) { return x->type == INTSXP ? (double)((int*)x->data)[0] : ((double*)x->data)[0]; }
int Rf_isNull(SEXP x) { return x == R_NilValue || x->type == NILSXP; }
int Rf_isInteger(SEXP x) { return x->type == INTSXP; }
double* REAL(SEXP x) { return (double*)live(x, "REAL() of a collected object")->data; }
int* INTEGER(SEXP x) { return (int*)live(x, "INTEGER() of a collected object")->data; }
int* LOGICAL(SEXP x) { return (int*)live(x, "LOGICAL() of a collected object")->data; }
R_xlen_t XLENGTH(SEXP x) { return live(x, "XLENGTH() of a collected object")->n; }
SEXP Rf_allocVector(int type, R_xlen_t n) {
  SEXP s = mk(type, n, -1, -1);
  if (type == VECSXP) for (R_xlen_t i = 0; i < n; ++i) ((SEXP*)s->data)[i] = R_NilValue;
  return s;
}
SEXP Rf_allocMatrix(int type, int nrow, int ncol) { return mk(type, (R_xlen_t)nrow * ncol, nrow, ncol); }
SEXP Rf_mkNamed(int type, const char** names) {
  R_xlen_t n = 0;
  while (names[n][0]) ++n;
  SEXP s = mk(type, n, -1, -1);
  /* R copies the names into the object (the caller's array is usually a local of the .Call routine) */
  const char** copy = (const char**)calloc((size_t)n + 1, sizeof(char*));
  for (R_xlen_t i = 0; i < n; ++i) copy[i] = strdup(names[i]);
  s->names = copy;
  for (R_xlen_t i = 0; i < n; ++i) ((SEXP*)s->data)[i] = R_NilValue;
  return s;
}
SEXP SET_VECTOR_ELT(SEXP x, R_xlen_t i, SEXP v) { live(x, "SET_VECTOR_ELT() into a collected list"); live(v, "SET_VECTOR_ELT() of a collected object"); ((SEXP*)x->data)[i] = v; return v; }
SEXP VECTOR_ELT(SEXP x, R_xlen_t i) { return ((SEXP*)live(x, "VECTOR_ELT() of a collected list")->data)[i]; }
SEXP Rf_xlengthgets(SEXP x, R_xlen_t n) {
  live(x, "Rf_xlengthgets() of a collected object");
  const int was_arg = x->is_arg;
  x->is_arg = 1;                       /* (R protects its argument for the duration of the call) */
  SEXP y = mk(x->type, n, -1, -1);
  x->is_arg = was_arg;
  const size_t esz = x->type == REALSXP ? sizeof(double) : (x->type == INTSXP || x->type == LGLSXP) ? sizeof(int) : sizeof(SEXP);
  memcpy(y->data, x->data, (size_t)(n < x->n ? n : x->n) * esz);
  return y;
}
SEXP Rf_protect(SEXP x) {
  live(x, "PROTECT() of an object that was already collected");
  if (pdepth < PSTACK_MAX) pstack[pdepth++] = x; else violation("protect stack overflow");
  return x;
}
void Rf_unprotect(int n) { if (n > pdepth) { violation("UNPROTECT() of more objects than are protected"); n = pdepth; } pdepth -= n; }
void Rf_error(const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(rstub_error_msg, sizeof(rstub_error_msg), fmt, ap); va_end(ap);
  pdepth = 0;                          /* R unwinds the protect stack to the context the error returns to */
  if (rstub_error_armed) longjmp(rstub_error_jmp, 1);
  fprintf(stderr, "Rf_error: %s\n", rstub_error_msg);
  abort();
}
char* R_alloc(size_t n, int size) {
  void* p = calloc(n ? n : 1, (size_t)size);
  if (n_ralloc < 64) all_ralloc[n_ralloc++] = p;
  return (char*)p;
}
int R_registerRoutines(DllInfo* d, const void* a, const R_CallMethodDef* b, const void* c, const void* e) { (void)d; (void)a; (void)b; (void)c; (void)e; return 1; }
Rboolean R_useDynamicSymbols(DllInfo* d, Rboolean v) { (void)d; return v; }
/* an interrupt inside R_ToplevelExec unwinds to it and makes it return FALSE */
void R_CheckUserInterrupt(void) {
  ++n_checks;
  if (rstub_interrupt_after > 0 && n_checks >= rstub_interrupt_after) {
    if (toplevel_jmp) longjmp(*toplevel_jmp, 1);
    Rf_error("interrupt");
  }
}
Rboolean R_ToplevelExec(void (*fun)(void*), void* data) {
  jmp_buf jb; jmp_buf* prev = toplevel_jmp;
  toplevel_jmp = &jb;
  if (setjmp(jb)) { toplevel_jmp = prev; return FALSE; }
  fun(data);
  toplevel_jmp = prev;
  return TRUE;
}
SEXP rstub_real_matrix(const double* d, int nrow, int ncol) { SEXP s = Rf_allocMatrix(REALSXP, nrow, ncol); memcpy(s->data, d, sizeof(double) * (size_t)nrow * ncol); return s; }
SEXP rstub_int_matrix(const int* d, int nrow, int ncol) { SEXP s = Rf_allocMatrix(INTSXP, nrow, ncol); memcpy(s->data, d, sizeof(int) * (size_t)nrow * ncol); return s; }
SEXP rstub_real_vector(const double* d, R_xlen_t n) { SEXP s = Rf_allocVector(REALSXP, n); memcpy(s->data, d, sizeof(double) * (size_t)n); return s; }
SEXP rstub_list(R_xlen_t n) { return Rf_allocVector(VECSXP, n); }
SEXP rstub_int_vector(const int* d, R_xlen_t n) { SEXP s = Rf_allocVector(INTSXP, n); memcpy(s->data, d, sizeof(int) * (size_t)n); return s; }
SEXP rstub_scalar_int(int v) { SEXP s = Rf_allocVector(INTSXP, 1); ((int*)s->data)[0] = v; return s; }
SEXP rstub_scalar_real(double v) { SEXP s = Rf_allocVector(REALSXP, 1); ((double*)s->data)[0] = v; return s; }
const char* rstub_name(SEXP l, R_xlen_t i) { return l->names ? l->names[i] : ""; }
void rstub_free_all(void) {
  while (all_objs) {
    struct rstub_sexp* nx = all_objs->next;
    if (all_objs->names) { for (R_xlen_t i = 0; i < all_objs->n; ++i) free((void*)all_objs->names[i]); free((void*)all_objs->names); }
    free(all_objs->data); free(all_objs); all_objs = nx;
  }
  for (int i = 0; i < n_ralloc; ++i) free(all_ralloc[i]);
  n_ralloc = 0; n_checks = 0;
  pdepth = 0; args_phase = 1; n_violations = 0;
}

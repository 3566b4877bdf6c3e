/* C harness that calls the R shim's .Call entry point exactly as R would (tests/test_gpu_boundary.py loads it with ctypes).
 * Inputs are column-major, like R matrices.  Returns 0, or 1 with the Rf_error() text in err. */
#include <setjmp.h>
#include <stdio.h>
#include <string.h>
#include "Rinternals.h"
extern jmp_buf rstub_error_jmp; extern int rstub_error_armed; extern char rstub_error_msg[1024];
SEXP C_clonealign_fit(SEXP Y, SEXP L, SEXP psi0, SEXP psi_noise, SEXP loc0, SEXP X, SEXP extra, SEXP K_, SEXP S_, SEXP max_iter_,
                      SEXP rel_tol_, SEXP lr_, SEXP eps_, SEXP devices_);

/* Round 6: every entry point is held to R's memory rules by the stub (rstub.c: protect-stack accounting + a collection at every allocation).
 * After a normal return the protect stack must be back at its entry depth and no collected object may have been touched; on the error path
 * (Rf_error's longjmp) R itself unwinds the stack, but a use of a collected object BEFORE the error still counts.  A violation turns the
 * call into a failure whose message starts with "R memory rule:". */
static int memcheck(char* err, int after_error) {
  char m[256];
  const int depth = rstub_protect_depth(), nv = rstub_violations(m, sizeof(m));
  if (nv) { snprintf(err, 1023, "R memory rule: %s (%d violation%s)", m, nv, nv == 1 ? "" : "s"); return 1; }
  if (!after_error && depth != 0) { snprintf(err, 1023, "R memory rule: protect stack imbalance (%d left protected at return)", depth); return 1; }
  return 0;
}
#define ENTER_ERROR_PATH(err) do { rstub_error_armed = 0; char keep_[1024]; strncpy(keep_, rstub_error_msg, 1023); keep_[1023] = 0; \
    if (!memcheck(err, 1)) { strncpy(err, keep_, 1023); err[1023] = 0; } rstub_free_all(); return 1; } while (0)

static void copy_out(SEXP list, const char* name, double* dst, long cap, long* len) {
  for (R_xlen_t i = 0; i < XLENGTH(list); ++i)
    if (strcmp(rstub_name(list, i), name) == 0) {
      SEXP v = VECTOR_ELT(list, i);
      const long n = (long)XLENGTH(v);
      if (len) *len = n;
      if (dst) memcpy(dst, REAL(v), sizeof(double) * (size_t)(n < cap ? n : cap));
      return;
    }
  if (len) *len = -1;
}

/* psi0 NULL with K > 0: psi is initialised on the device (prcomp + scale) plus psi_noise (may be NULL) */
int harness_fit_devices(const double* Yd, const int* Yi, int N, int G, const double* L, int C, const double* psi0, const double* psi_noise, const double* loc0,
                        const double* X, int P, const double* extra, int K, int S, int max_iter, double rel_tol, double lr,
                        const double* eps, long n_eps, int interrupt_after, double* elbo, long* n_elbo, double* finals, double* mu,
                        double* clone_probs, double* s, double* alpha, double* psi, double* W, double* chi, double* beta, char* err,
                        const int* devices, int n_dev) {
  rstub_interrupt_after = interrupt_after;
  rstub_error_armed = 1;
  if (setjmp(rstub_error_jmp)) {
    ENTER_ERROR_PATH(err);
  }
  SEXP a[14] = {Yi ? rstub_int_matrix(Yi, N, G) : rstub_real_matrix(Yd, N, G), rstub_real_matrix(L, G, C),
                K > 0 && psi0 ? rstub_real_matrix(psi0, N, K) : R_NilValue, K > 0 && psi_noise ? rstub_real_matrix(psi_noise, N, K) : R_NilValue,
                loc0 ? rstub_real_vector(loc0, G) : R_NilValue,
                P > 0 ? rstub_real_matrix(X, N, P) : R_NilValue, extra ? rstub_real_matrix(extra, N, C) : R_NilValue,
                rstub_scalar_int(K), rstub_scalar_int(S), rstub_scalar_int(max_iter), rstub_scalar_real(rel_tol),
                rstub_scalar_real(lr), eps ? rstub_real_vector(eps, n_eps) : R_NilValue,
                devices ? rstub_int_vector(devices, n_dev) : R_NilValue};
  rstub_begin_call();
  SEXP out = C_clonealign_fit(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13]);
  rstub_error_armed = 0;
  if (memcheck(err, 0)) { rstub_free_all(); return 1; }
  copy_out(out, "elbo", elbo, max_iter + 1, n_elbo);
  copy_out(out, "final_elbos", finals, 20, NULL);
  copy_out(out, "mu", mu, G, NULL);
  copy_out(out, "clone_probs", clone_probs, (long)N * C, NULL);
  copy_out(out, "s", s, N, NULL);
  copy_out(out, "alpha", alpha, C, NULL);
  copy_out(out, "psi", psi, (long)N * K, NULL);
  copy_out(out, "W", W, (long)G * K, NULL);
  copy_out(out, "chi", chi, K, NULL);
  copy_out(out, "beta", beta, (long)G * P, NULL);
  rstub_free_all();
  return 0;
}

int harness_fit(const double* Yd, const int* Yi, int N, int G, const double* L, int C, const double* psi0, const double* psi_noise, const double* loc0,
                const double* X, int P, const double* extra, int K, int S, int max_iter, double rel_tol, double lr,
                const double* eps, long n_eps, int interrupt_after, double* elbo, long* n_elbo, double* finals, double* mu,
                double* clone_probs, double* s, double* alpha, double* psi, double* W, double* chi, double* beta, char* err) {
  return harness_fit_devices(Yd, Yi, N, G, L, C, psi0, psi_noise, loc0, X, P, extra, K, S, max_iter, rel_tol, lr, eps, n_eps, interrupt_after, elbo, n_elbo,
                             finals, mu, clone_probs, s, alpha, psi, W, chi, beta, err, NULL, 0);
}


/* ---- the other .Call entry points, driven the same way ---- */
SEXP C_clonealign_multifit(SEXP Y, SEXP L, SEXP psi0, SEXP psi_noise, SEXP loc0, SEXP X, SEXP extra, SEXP K_, SEXP S_, SEXP max_iter_,
                           SEXP rel_tol_, SEXP lr_, SEXP eps_, SEXP devices_, SEXP want_sums_, SEXP call_prob_);
SEXP C_clonealign_preprocess(SEXP Y, SEXP L, SEXP min_gene_, SEXP min_cell_, SEXP outl_, SEXP nmads_, SEXP max_cn_, SEXP same_cn_, SEXP device_);
SEXP C_clonealign_allele_loglik(SEXP clone_allele, SEXP cov, SEXP ref, SEXP device_);

/* R restarts; psi is [R][N*K] (psi0 when by_noise = 0, the PCA noise otherwise), eps [R][n_eps] or NULL.  Outputs are [R][...]. */
int harness_multifit(const double* Yd, const int* Yi, int N, int G, const double* L, int C, const double* psi, int by_noise,
                     const double* loc0, int K, int S, int max_iter, double rel_tol, double lr, const double* eps, long n_eps, int R,
                     const int* devices, int n_dev, int want_sums, int interrupt_after, double* elbo, long* n_elbo, double* finals,
                     double* mu, double* clone_probs, double* alpha, double* psi_out, double* W, double* T, double* Syy, char* err) {
  rstub_interrupt_after = interrupt_after;
  rstub_error_armed = 1;
  if (setjmp(rstub_error_jmp)) {
    ENTER_ERROR_PATH(err);
  }
  SEXP plist = rstub_list(R), elist = eps ? rstub_list(R) : R_NilValue;
  for (int r = 0; r < R; ++r) {
    SET_VECTOR_ELT(plist, r, rstub_real_matrix(psi + (size_t)r * N * K, N, K));
    if (eps) SET_VECTOR_ELT(elist, r, rstub_real_vector(eps + (size_t)r * n_eps, n_eps));
  }
  SEXP a[16] = {Yi ? rstub_int_matrix(Yi, N, G) : rstub_real_matrix(Yd, N, G), rstub_real_matrix(L, G, C),
                by_noise ? R_NilValue : plist, by_noise ? plist : R_NilValue, loc0 ? rstub_real_vector(loc0, G) : R_NilValue,
                R_NilValue, R_NilValue, rstub_scalar_int(K), rstub_scalar_int(S), rstub_scalar_int(max_iter),
                rstub_scalar_real(rel_tol), rstub_scalar_real(lr), elist, rstub_int_vector(devices, n_dev),
                rstub_scalar_int(want_sums), rstub_scalar_real(0.95)};
  rstub_begin_call();
  SEXP out = C_clonealign_multifit(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13], a[14], a[15]);
  rstub_error_armed = 0;
  if (memcheck(err, 0)) { rstub_free_all(); return 1; }
  for (int r = 0; r < R; ++r) {
    SEXP f = VECTOR_ELT(out, r);
    copy_out(f, "elbo", elbo + (size_t)r * (max_iter + 1), max_iter + 1, n_elbo + r);
    copy_out(f, "final_elbos", finals + (size_t)r * 20, 20, NULL);
    copy_out(f, "mu", mu + (size_t)r * G, G, NULL);
    copy_out(f, "clone_probs", clone_probs + (size_t)r * N * C, (long)N * C, NULL);
    copy_out(f, "alpha", alpha + (size_t)r * C, C, NULL);
    copy_out(f, "psi", psi_out + (size_t)r * N * K, (long)N * K, NULL);
    copy_out(f, "W", W + (size_t)r * G * K, (long)G * K, NULL);
    if (want_sums) {
      copy_out(f, "T", T + (size_t)r * G * C, (long)G * C, NULL);
      copy_out(f, "Syy", Syy + (size_t)r * G, G, NULL);
    }
  }
  rstub_free_all();
  return 0;
}

int harness_preprocess(const double* Yd, const int* Yi, int N, int G, const double* L, int C, double min_gene, double min_cell, int outl,
                       double nmads, double max_cn, int same_cn, int* keep_gene, int* keep_cell, double* gene_sums, double* cell_sums, char* err) {
  rstub_error_armed = 1;
  if (setjmp(rstub_error_jmp)) {
    ENTER_ERROR_PATH(err);
  }
  SEXP a[9] = {Yi ? rstub_int_matrix(Yi, N, G) : rstub_real_matrix(Yd, N, G), rstub_real_matrix(L, G, C),
               rstub_scalar_real(min_gene), rstub_scalar_real(min_cell), rstub_scalar_int(outl), rstub_scalar_real(nmads),
               rstub_scalar_real(max_cn), rstub_scalar_int(same_cn), rstub_scalar_int(0)};
  rstub_begin_call();
  SEXP out = C_clonealign_preprocess(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8]);
  rstub_error_armed = 0;
  if (memcheck(err, 0)) { rstub_free_all(); return 1; }
  memcpy(keep_gene, LOGICAL(VECTOR_ELT(out, 0)), sizeof(int) * (size_t)G);
  memcpy(keep_cell, LOGICAL(VECTOR_ELT(out, 1)), sizeof(int) * (size_t)N);
  memcpy(gene_sums, REAL(VECTOR_ELT(out, 2)), sizeof(double) * (size_t)G);
  memcpy(cell_sums, REAL(VECTOR_ELT(out, 3)), sizeof(double) * (size_t)N);
  rstub_free_all();
  return 0;
}

int harness_allele(const double* clone_allele, int V, int C, const double* cov, const double* ref, int N, double* out, char* err) {
  rstub_error_armed = 1;
  if (setjmp(rstub_error_jmp)) {
    ENTER_ERROR_PATH(err);
  }
  SEXP a[4] = {rstub_real_matrix(clone_allele, V, C), rstub_real_matrix(cov, N, V), rstub_real_matrix(ref, N, V), rstub_scalar_int(0)};
  rstub_begin_call();
  SEXP o = C_clonealign_allele_loglik(a[0], a[1], a[2], a[3]);
  rstub_error_armed = 0;
  if (memcheck(err, 0)) { rstub_free_all(); return 1; }
  memcpy(out, REAL(o), sizeof(double) * (size_t)N * C);
  rstub_free_all();
  return 0;
}

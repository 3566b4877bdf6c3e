/* C harness that calls the R shim's .Call entry point exactly as R would (tests/test_gpu_shim.py loads it with ctypes).
 * Inputs are column-major, like R matrices.  Returns 0, or 1 with the Rf_error() text in err. */
#include <setjmp.h>
#include <stdio.h>
#include <string.h>
#include "Rinternals.h"
extern jmp_buf rstub_error_jmp; extern int rstub_error_armed; extern char rstub_error_msg[1024];
SEXP C_clonealign_fit(SEXP Y, SEXP L, SEXP psi0, SEXP loc0, SEXP X, SEXP extra, SEXP K_, SEXP S_, SEXP max_iter_, SEXP rel_tol_,
                      SEXP lr_, SEXP eps_);

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

int harness_fit(const double* Yd, const int* Yi, int N, int G, const double* L, int C, const double* psi0, const double* loc0,
                const double* X, int P, const double* extra, int K, int S, int max_iter, double rel_tol, double lr,
                const double* eps, long n_eps, int interrupt_after, double* elbo, long* n_elbo, double* finals, double* mu,
                double* clone_probs, double* s, double* alpha, double* psi, double* W, double* chi, double* beta, char* err) {
  rstub_interrupt_after = interrupt_after;
  rstub_error_armed = 1;
  if (setjmp(rstub_error_jmp)) {
    rstub_error_armed = 0;
    strncpy(err, rstub_error_msg, 1023); err[1023] = 0;
    rstub_free_all();
    return 1;
  }
  SEXP out = C_clonealign_fit(Yi ? rstub_int_matrix(Yi, N, G) : rstub_real_matrix(Yd, N, G), rstub_real_matrix(L, G, C),
                              K > 0 ? rstub_real_matrix(psi0, N, K) : R_NilValue, loc0 ? rstub_real_vector(loc0, G) : R_NilValue,
                              P > 0 ? rstub_real_matrix(X, N, P) : R_NilValue, extra ? rstub_real_matrix(extra, N, C) : R_NilValue,
                              rstub_scalar_int(K), rstub_scalar_int(S), rstub_scalar_int(max_iter), rstub_scalar_real(rel_tol),
                              rstub_scalar_real(lr), eps ? rstub_real_vector(eps, n_eps) : R_NilValue);
  rstub_error_armed = 0;
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

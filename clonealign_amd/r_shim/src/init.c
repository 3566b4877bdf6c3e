/* Native routine registration of the clonealign package once the MI355X engine replaces the TensorFlow path
 * (NAMESPACE: useDynLib(clonealign, .registration = TRUE); the reference has no src/ directory at all -- its hot path is
 * reticulate -> TensorFlow, R/inference-tflow.R:96-99).  The four .Call entry points are defined in clonealign_hip_shim.c;
 * their argument counts here are what R checks every .Call against. */
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>
#include <stddef.h>

extern SEXP C_clonealign_fit(SEXP Y, SEXP L, SEXP psi0, SEXP psi_noise, SEXP loc0, SEXP X, SEXP extra, SEXP K, SEXP S, SEXP max_iter,
                             SEXP rel_tol, SEXP learning_rate, SEXP eps, SEXP devices);
extern SEXP C_clonealign_multifit(SEXP Y, SEXP L, SEXP psi0, SEXP psi_noise, SEXP loc0, SEXP X, SEXP extra, SEXP K, SEXP S, SEXP max_iter,
                                  SEXP rel_tol, SEXP learning_rate, SEXP eps, SEXP devices, SEXP want_sums, SEXP call_prob);
extern SEXP C_clonealign_preprocess(SEXP Y, SEXP L, SEXP min_gene, SEXP min_cell, SEXP outlying, SEXP nmads, SEXP max_cn, SEXP same_cn,
                                    SEXP device);
extern SEXP C_clonealign_allele_loglik(SEXP clone_allele, SEXP cov, SEXP ref, SEXP device);

static const R_CallMethodDef CallEntries[] = {
  {"C_clonealign_fit", (DL_FUNC)&C_clonealign_fit, 14},
  {"C_clonealign_multifit", (DL_FUNC)&C_clonealign_multifit, 16},
  {"C_clonealign_preprocess", (DL_FUNC)&C_clonealign_preprocess, 9},
  {"C_clonealign_allele_loglik", (DL_FUNC)&C_clonealign_allele_loglik, 4},
  {NULL, NULL, 0}};

void R_init_clonealign(DllInfo* dll) {
  R_registerRoutines(dll, NULL, CallEntries, NULL, NULL);
  R_useDynamicSymbols(dll, FALSE);
}

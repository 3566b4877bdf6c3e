/*
 * R-side binding of libclonealign_hip.so: the `.Call` stub a clonealign maintainer adds under src/.
 * The build image has no R toolchain, so the suite compiles it against a minimal stand-in for the R API
 * (tests/r_stub/, tests/test_shim_compiles.py) and drives C_clonealign_fit from a C harness on the GPU box
 * (tests/test_gpu_shim.py); it is the reference-side half of the boundary documented in INTEGRATION.md and mirrors,
 * call for call, what clonealign_amd/engine.py does through ctypes with layout="col".
 *
 * Replaces the body of inference_tflow() between R/inference-tflow.R:240 (graph build) and :457
 * (sess$close): everything before (gene filter, saturate, PCA / mu init) and after (naming, return list)
 * stays R code, unchanged.
 *
 *   .Call("C_clonealign_fit", Y, L, psi0, loc0, X, extra, K, S, max_iter, rel_tol, learning_rate, eps)
 *     Y      numeric or integer matrix N x G (column-major, as R stores it)
 *     L      numeric matrix G x C;  psi0 N x K;  loc0 G or NULL (device-side mu_guess);  X N x P or NULL;  extra N x C or NULL
 *     eps    numeric vector of (2 + 2*max_iter + 20) * S * G standard normals drawn with rnorm() by the
 *            caller (so set.seed() controls the fit exactly as it does through get_next_seed(), :49-51),
 *            or NULL for the engine's built-in Philox stream
 *   returns list(mu, clone_probs, s, alpha, beta, psi, W, chi, elbo, final_elbos)
 */
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Utils.h>
#include <string.h>
#include "clonealign_hip.h"

static void fail(ca_handle h, const char* what) {
  char msg[512];
  strncpy(msg, ca_last_error(h), sizeof(msg) - 1);
  msg[sizeof(msg) - 1] = 0;
  ca_destroy(h);                       /* free device memory BEFORE the longjmp of Rf_error */
  Rf_error("%s: %s", what, msg);
}

/* The reference's loop is R-level and can be interrupted every iteration (R/inference-tflow.R:394-417).  ca_run_ex() calls
 * this between iterations; R_CheckUserInterrupt() may longjmp, so it runs inside R_ToplevelExec() and never unwinds through
 * the library: a pending interrupt makes the loop stop cleanly (CA_INTERRUPTED), the handle is freed, then R is told. */
static void check_interrupt(void* unused) { (void)unused; R_CheckUserInterrupt(); }
static int poll_interrupt(void* user, int32_t iter, double elbo) {
  (void)user; (void)iter; (void)elbo;
  return R_ToplevelExec(check_interrupt, NULL) == FALSE;
}

static SEXP fetch(ca_handle h, const char* name, R_xlen_t nrow, R_xlen_t ncol) {
  SEXP out = PROTECT(ncol < 0 ? Rf_allocVector(REALSXP, nrow) : Rf_allocMatrix(REALSXP, nrow, ncol));
  memset(REAL(out), 0, sizeof(double) * XLENGTH(out));
  if (XLENGTH(out) > 0 && ca_get_param(h, name, REAL(out)) != CA_OK) { UNPROTECT(1); fail(h, name); }
  UNPROTECT(1);
  return out;
}

SEXP C_clonealign_fit(SEXP Y, SEXP L, SEXP psi0, SEXP loc0, SEXP X, SEXP extra, SEXP K_, SEXP S_, SEXP max_iter_,
                      SEXP rel_tol_, SEXP lr_, SEXP eps_) {
  ca_problem p;
  memset(&p, 0, sizeof(p));
  p.N = Rf_nrows(Y); p.G = Rf_ncols(Y); p.C = Rf_ncols(L);
  p.K = Rf_asInteger(K_); p.S = Rf_asInteger(S_);
  p.P = Rf_isNull(X) ? 0 : Rf_ncols(X);
  p.layout = CA_COL_MAJOR;                                   /* R matrices as they are: no transpose, no copy */
  p.y_dtype = Rf_isInteger(Y) ? CA_I32 : CA_F64;
  p.Y = Rf_isInteger(Y) ? (const void*)INTEGER(Y) : (const void*)REAL(Y);
  p.L = REAL(L); p.psi0 = p.K > 0 ? REAL(psi0) : NULL; p.loc0 = Rf_isNull(loc0) ? NULL : REAL(loc0);   /* NULL: data_init_mu = TRUE guess (:220-235) taken on the device */
  p.X = p.P > 0 ? REAL(X) : NULL;
  p.extra_loglik = Rf_isNull(extra) ? NULL : REAL(extra);
  ca_options o;
  ca_default_options(&o);
  o.learning_rate = Rf_asReal(lr_);
  ca_handle h = NULL;
  if (ca_create(&p, &o, &h) != CA_OK) Rf_error("clonealign_hip: %s", ca_last_error(NULL));

  const int max_iter = Rf_asInteger(max_iter_);
  const R_xlen_t per = (R_xlen_t)p.S * p.G, ndraw = 2 + 2 * (R_xlen_t)max_iter + 20;
  float* eps = NULL;
  if (!Rf_isNull(eps_)) {                                    /* rnorm() doubles -> float32 stream */
    if (XLENGTH(eps_) < ndraw * per) { ca_destroy(h); Rf_error("eps stream too short"); }
    eps = (float*)R_alloc((size_t)(ndraw * per), sizeof(float));
    for (R_xlen_t i = 0; i < ndraw * per; ++i) eps[i] = (float)REAL(eps_)[i];
  }
  SEXP elbo = PROTECT(Rf_allocVector(REALSXP, max_iter + 1));
  int n_elbo = 0;
  /* whole loop of :368-417 in the library, interruptible between iterations like the reference's R-level loop */
  int rc = ca_run_ex(h, max_iter, Rf_asReal(rel_tol_), eps, eps ? ndraw : 0, REAL(elbo), &n_elbo, poll_interrupt, NULL);
  if (rc == CA_INTERRUPTED) { UNPROTECT(1); ca_destroy(h); Rf_error("clonealign: interrupted"); }
  if (rc == CA_ERR_NAN) { UNPROTECT(1); fail(h, "clonealign");  /* "Initial elbo is NA", :374-376 */ }
  if (rc != CA_OK) { UNPROTECT(1); fail(h, "ca_run"); }
  SEXP finals = PROTECT(Rf_allocVector(REALSXP, 20));        /* :447-449 */
  const R_xlen_t used = 2 * (R_xlen_t)n_elbo;
  if (ca_final_elbo(h, 20, eps ? eps + used * per : NULL, eps ? ndraw - used : 0, REAL(finals), NULL, NULL) != CA_OK) {
    UNPROTECT(2); fail(h, "ca_final_elbo");
  }
  const char* names[] = {"mu", "clone_probs", "s", "alpha", "beta", "psi", "W", "chi", "elbo", "final_elbos", ""};
  SEXP out = PROTECT(Rf_mkNamed(VECSXP, names));
  SET_VECTOR_ELT(out, 0, fetch(h, "mu", p.G, -1));           /* :424 */
  SET_VECTOR_ELT(out, 1, fetch(h, "clone_probs", p.N, p.C));
  SET_VECTOR_ELT(out, 2, fetch(h, "s", p.N, -1));
  SET_VECTOR_ELT(out, 3, fetch(h, "alpha", p.C, -1));
  SET_VECTOR_ELT(out, 4, fetch(h, "beta", p.G, p.P));        /* :425-427 */
  SET_VECTOR_ELT(out, 5, fetch(h, "psi", p.N, p.K));         /* :429-434 */
  SET_VECTOR_ELT(out, 6, fetch(h, "W", p.G, p.K));
  SET_VECTOR_ELT(out, 7, fetch(h, "chi", p.K, -1));
  SET_VECTOR_ELT(out, 8, Rf_xlengthgets(elbo, n_elbo));
  SET_VECTOR_ELT(out, 9, finals);
  ca_destroy(h);                                             /* :457 sess$close() */
  UNPROTECT(3);
  return out;
}

static const R_CallMethodDef CallEntries[] = {{"C_clonealign_fit", (DL_FUNC)&C_clonealign_fit, 12}, {NULL, NULL, 0}};
void R_init_clonealign(DllInfo* dll) {
  R_registerRoutines(dll, NULL, CallEntries, NULL, NULL);
  R_useDynamicSymbols(dll, FALSE);
}

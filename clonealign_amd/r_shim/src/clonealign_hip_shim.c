/*
 * R-side binding of libclonealign_hip.so: the `.Call` stub a clonealign maintainer adds under src/.
 * The build image has no R toolchain, so the suite compiles it against a minimal stand-in for the R API
 * (tests/r_stub/, tests/test_shim_compiles.py) and drives C_clonealign_fit from a C harness on the GPU box
 * (tests/test_gpu_boundary.py); it is the reference-side half of the boundary documented in INTEGRATION.md and mirrors,
 * call for call, what clonealign_amd/engine.py does through ctypes with layout="col".
 *
 * Replaces the body of inference_tflow() between R/inference-tflow.R:240 (graph build) and :457
 * (sess$close): everything before (gene filter, saturate, PCA / mu init) and after (naming, return list)
 * stays R code, unchanged.
 *
 *   .Call("C_clonealign_fit", Y, L, psi0, psi_noise, loc0, X, extra, K, S, max_iter, rel_tol, learning_rate, eps, devices)
 *     Y      numeric or integer matrix N x G (column-major, as R stores it)
 *     L      numeric matrix G x C;  loc0 G or NULL (device-side mu_guess, :220-235);  X N x P or NULL;  extra N x C or NULL
 *     psi0   N x K initial latent positions (pcs + noise, :204-208), or NULL: then psi is initialised ON THE DEVICE -- prcomp +
 *            scale of :204-207 by subspace iteration over the resident counts (ca_init_psi_pca) -- plus psi_noise (N x K, the
 *            reference's rnorm(N * K, 0, 0.05) of :208, or NULL for none); host prcomp() of a 100k x 5k matrix takes minutes
 *            and 4 GB, the fit it initialises 60 ms
 *     eps    numeric vector of (2 + 2*max_iter + 20) * S * G standard normals drawn with rnorm() by the
 *            caller (so set.seed() controls the fit exactly as it does through get_next_seed(), :49-51),
 *            or NULL for the engine's built-in Philox stream
 *     devices integer vector of HIP device ordinals (or NULL = device 0): more than one = this ONE fit cell-sharded over them
 *   returns list(mu, clone_probs, s, alpha, beta, psi, W, chi, elbo, final_elbos)
 */
#define _GNU_SOURCE   /* pthread_timedjoin_np (C_clonealign_multifit) */
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Utils.h>
#include <stdio.h>
#include <string.h>
#include "clonealign_hip.h"

static void fail(ca_group_handle g, const char* what) {
  char msg[512];
  strncpy(msg, ca_group_last_error(g), sizeof(msg) - 1);
  msg[sizeof(msg) - 1] = 0;
  ca_group_destroy(g);                 /* free device memory (every rank's) BEFORE the longjmp of Rf_error */
  Rf_error("%s: %s", what, msg);
}

/* The reference's loop is R-level and can be interrupted every iteration (R/inference-tflow.R:394-417).  ca_group_run_ex() calls
 * this between iterations ON THE CALLING THREAD (rank 0 of the device group is driven by it; the other ranks follow its decision);
 * R_CheckUserInterrupt() may longjmp, so it runs inside R_ToplevelExec() and never unwinds through the library: a pending interrupt
 * makes the loop stop cleanly (CA_INTERRUPTED), the engines are freed, then R is told. */
static void check_interrupt(void* unused) { (void)unused; R_CheckUserInterrupt(); }
static int poll_interrupt(void* user, int32_t iter, double elbo) {
  (void)user; (void)iter; (void)elbo;
  return R_ToplevelExec(check_interrupt, NULL) == FALSE;
}

static SEXP fetch(ca_group_handle g, const char* name, R_xlen_t nrow, R_xlen_t ncol) {
  SEXP out = PROTECT(ncol < 0 ? Rf_allocVector(REALSXP, nrow) : Rf_allocMatrix(REALSXP, nrow, ncol));
  memset(REAL(out), 0, sizeof(double) * XLENGTH(out));
  if (XLENGTH(out) > 0 && ca_group_get_param(g, name, REAL(out)) != CA_OK) { UNPROTECT(1); fail(g, name); }
  UNPROTECT(1);
  return out;
}

/* `devices` (integer vector of HIP ordinals; ABI 6): ONE fit cell-sharded over these devices of this R session -- rank r holds the
 * cells [N r / W, N (r + 1) / W), one engine handle and one host thread per device (ca_group_*), joined by the first transport that
 * passes its known-answer test (peer-to-peer by address -> RCCL -> host reduction).  A single ordinal is the plain one-device fit.
 * Everything R sees is as before: inputs for all cells, outputs for all cells, column-major. */
SEXP C_clonealign_fit(SEXP Y, SEXP L, SEXP psi0, SEXP psi_noise, SEXP loc0, SEXP X, SEXP extra, SEXP K_, SEXP S_, SEXP max_iter_,
                      SEXP rel_tol_, SEXP lr_, SEXP eps_, SEXP devices_) {
  ca_problem p;
  memset(&p, 0, sizeof(p));
  p.N = Rf_nrows(Y); p.G = Rf_ncols(Y); p.C = Rf_ncols(L);
  p.K = Rf_asInteger(K_); p.S = Rf_asInteger(S_);
  p.P = Rf_isNull(X) ? 0 : Rf_ncols(X);
  p.layout = CA_COL_MAJOR;                                   /* R matrices as they are: no transpose, no copy */
  p.y_dtype = Rf_isInteger(Y) ? CA_I32 : CA_F64;
  p.Y = Rf_isInteger(Y) ? (const void*)INTEGER(Y) : (const void*)REAL(Y);
  const int device_pca = p.K > 0 && Rf_isNull(psi0);
  double* psi_zero = NULL;
  if (device_pca) {                                          /* the engine wants SOME psi0 at creation; ca_init_psi_pca overwrites it */
    psi_zero = (double*)R_alloc((size_t)p.N * (size_t)p.K, sizeof(double));
    memset(psi_zero, 0, sizeof(double) * (size_t)p.N * (size_t)p.K);
  }
  p.L = REAL(L); p.psi0 = p.K > 0 ? (device_pca ? psi_zero : REAL(psi0)) : NULL;
  p.loc0 = Rf_isNull(loc0) ? NULL : REAL(loc0);   /* NULL: data_init_mu = TRUE guess (:220-235) taken on the device(s) */
  p.X = p.P > 0 ? REAL(X) : NULL;
  p.extra_loglik = Rf_isNull(extra) ? NULL : REAL(extra);
  ca_options o;
  ca_default_options(&o);
  o.learning_rate = Rf_asReal(lr_);
  int32_t one_device = 0;
  const int n_dev = Rf_isNull(devices_) ? 1 : (int)XLENGTH(devices_);
  if (n_dev < 1) Rf_error("clonealign_hip: devices must name at least one device");
  const int32_t* devs = Rf_isNull(devices_) ? &one_device : (const int32_t*)INTEGER(devices_);
  ca_group_handle h = NULL;
  if (ca_group_create(&p, &o, devs, n_dev, 0, &h) != CA_OK) Rf_error("clonealign_hip: %s", ca_group_last_error(NULL));
  if (device_pca && ca_group_init_psi_pca(h, Rf_isNull(psi_noise) ? NULL : REAL(psi_noise), 40, o.seed, NULL) != CA_OK) fail(h, "ca_init_psi_pca");

  const int max_iter = Rf_asInteger(max_iter_);
  const R_xlen_t per = (R_xlen_t)p.S * p.G, ndraw = 2 + 2 * (R_xlen_t)max_iter + 20;
  float* eps = NULL;
  if (!Rf_isNull(eps_)) {                                    /* rnorm() doubles -> float32 stream */
    if (XLENGTH(eps_) < ndraw * per) { ca_group_destroy(h); Rf_error("eps stream too short"); }
    eps = (float*)R_alloc((size_t)(ndraw * per), sizeof(float));
    for (R_xlen_t i = 0; i < ndraw * per; ++i) eps[i] = (float)REAL(eps_)[i];
  }
  SEXP elbo = PROTECT(Rf_allocVector(REALSXP, max_iter + 1));
  int n_elbo = 0;
  /* whole loop of :368-417 in the library, interruptible between iterations like the reference's R-level loop */
  int rc = ca_group_run_ex(h, max_iter, Rf_asReal(rel_tol_), eps, eps ? ndraw : 0, REAL(elbo), &n_elbo, poll_interrupt, NULL);
  if (rc == CA_INTERRUPTED) { UNPROTECT(1); ca_group_destroy(h); Rf_error("clonealign: interrupted"); }
  if (rc == CA_ERR_NAN) { UNPROTECT(1); fail(h, "clonealign");  /* "Initial elbo is NA", :374-376 */ }
  if (rc != CA_OK) { UNPROTECT(1); fail(h, "ca_run"); }
  SEXP finals = PROTECT(Rf_allocVector(REALSXP, 20));        /* :447-449 */
  const R_xlen_t used = 2 * (R_xlen_t)n_elbo;
  if (ca_group_final_elbo(h, 20, eps ? eps + used * per : NULL, eps ? ndraw - used : 0, REAL(finals), NULL, NULL) != CA_OK) {
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
  ca_group_destroy(h);                                       /* :457 sess$close() */
  UNPROTECT(3);
  return out;
}

/* ------------------------------------------------------------------------------------------------------------------------------
 * run_clonealign()'s restart loop (R/clonealign.R:50-56) on RESIDENT engines: the reference calls clonealign() -- graph build, two
 * full feeds of Y per iteration -- once per restart; here the count matrix is uploaded once per device and every further restart
 * of that device is a ca_reinit().  One worker thread per device (`devices`, HIP ordinals; restart r runs on devices[r mod D]);
 * the workers call nothing but the C ABI and write into buffers the main thread allocated beforehand -- the R API is touched by
 * the calling thread only, which meanwhile looks for a user interrupt every 50 ms and makes the workers' loops stop
 * (CA_INTERRUPTED) before it raises the R error.  which.max(final_elbo) (:65) stays R code.
 *
 *   .Call("C_clonealign_multifit", Y, L, psi0, psi_noise, loc0, X, extra, K, S, max_iter, rel_tol, learning_rate, eps, devices,
 *         want_sums, clone_call_probability)
 *     psi0       list of R numeric matrices N x K (one per restart: pcs + rnorm noise, :204-208), or NULL when psi_noise is given
 *     psi_noise  list of R numeric matrices N x K: psi is then initialised ON THE DEVICE (ca_init_psi_pca: prcomp + scale of
 *                :204-208 by subspace iteration over the resident counts) plus this noise (the reference's rnorm(.., 0, 0.05))
 *     eps        list of R numeric vectors, each (2 + 2 max_iter + 20) S G rnorm() draws, or NULL: the engines' built-in Philox stream,
 *                seeded PER DEVICE (seed + 1000003 d); a device's stream simply continues across its restarts (ca_reinit keeps it)
 *     want_sums  TRUE: each fit also carries T (G x C) and Syy (G), the sums compute_correlations() (:318-334) needs, taken on the
 *                device for the cells assigned with probability >= clone_call_probability (:22-29)
 *   returns a list of R fits, each list(mu, clone_probs, s, alpha, beta, psi, W, chi, elbo, final_elbos[, T, Syy])
 */
#include <pthread.h>
#include <stdlib.h>
#include <time.h>

typedef struct {
  double *mu, *clone_probs, *s, *alpha, *beta, *psi, *W, *chi, *elbo, *finals, *T, *Syy;
  int n_elbo;
} fit_out;
typedef struct {
  ca_problem p; ca_options o;
  int n_fits, n_dev, lane, max_iter, want_sums;
  double rel_tol, call_prob;
  const double* const* psi0; const double* const* noise; const double* const* eps; const double* loc0;
  fit_out* out;
  volatile int* cancel;
  int rc; char err[512];
} worker_arg;

static int poll_cancel(void* user, int32_t iter, double elbo) { (void)iter; (void)elbo; return *(volatile int*)user; }

static void* worker_main(void* arg_) {
  worker_arg* a = (worker_arg*)arg_;
  ca_handle h = NULL;
  a->rc = CA_OK;
  const int64_t N = a->p.N; const int G = a->p.G, C = a->p.C, K = a->p.K, P = a->p.P, S = a->p.S;
  const int64_t per = (int64_t)S * G, ndraw = 2 + 2 * (int64_t)a->max_iter + 20;
  float* epsf = NULL; int32_t* call = NULL; double* zeros = NULL;
#define WFAIL(what) do { snprintf(a->err, sizeof(a->err), "%s: %s", what, ca_last_error(h)); a->rc = rc ? rc : CA_ERR_STATE; goto done; } while (0)
  int rc = CA_OK;
  for (int r = a->lane; r < a->n_fits && !*a->cancel; r += a->n_dev) {
    const double* psi_r = a->psi0 ? a->psi0[r] : NULL;
    if (!psi_r && K > 0 && !zeros) zeros = (double*)calloc((size_t)N * K, sizeof(double));
    if (!h) {                                                  /* first restart of this device: upload + fit constants */
      ca_problem p = a->p;
      p.psi0 = K > 0 ? (psi_r ? psi_r : zeros) : NULL;
      if ((rc = ca_create(&p, &a->o, &h)) != CA_OK) { snprintf(a->err, sizeof(a->err), "ca_create: %s", ca_last_error(NULL)); a->rc = rc; goto done; }
    } else if ((rc = ca_reinit(h, K > 0 ? (psi_r ? psi_r : zeros) : NULL, a->loc0)) != CA_OK) WFAIL("ca_reinit");
    if (!psi_r && K > 0 && (rc = ca_init_psi_pca(h, a->noise ? a->noise[r] : NULL, 40, a->o.seed + (uint64_t)r, NULL)) != CA_OK) WFAIL("ca_init_psi_pca");
    const float* eps = NULL;
    if (a->eps) {
      if (!epsf) epsf = (float*)malloc(sizeof(float) * (size_t)(ndraw * per));
      for (int64_t i = 0; i < ndraw * per; ++i) epsf[i] = (float)a->eps[r][i];
      eps = epsf;
    }
    fit_out* o = &a->out[r];
    rc = ca_run_ex(h, a->max_iter, a->rel_tol, eps, eps ? ndraw : 0, o->elbo, &o->n_elbo, poll_cancel, (void*)a->cancel);
    if (rc == CA_INTERRUPTED) { a->rc = rc; goto done; }
    if (rc != CA_OK) WFAIL(rc == CA_ERR_NAN ? "clonealign" : "ca_run");
    const int64_t used = 2 * (int64_t)o->n_elbo;
    if ((rc = ca_final_elbo(h, 20, eps ? eps + used * per : NULL, eps ? ndraw - used : 0, o->finals, NULL, NULL)) != CA_OK) WFAIL("ca_final_elbo");
    if ((rc = ca_get_param(h, "mu", o->mu)) != CA_OK || (rc = ca_get_param(h, "clone_probs", o->clone_probs)) != CA_OK ||
        (rc = ca_get_param(h, "s", o->s)) != CA_OK || (rc = ca_get_param(h, "alpha", o->alpha)) != CA_OK) WFAIL("ca_get_param");
    if (P > 0 && K > 0 && (rc = ca_get_param(h, "beta", o->beta)) != CA_OK) WFAIL("ca_get_param(beta)");
    if (K > 0 && ((rc = ca_get_param(h, "psi", o->psi)) != CA_OK || (rc = ca_get_param(h, "W", o->W)) != CA_OK ||
                  (rc = ca_get_param(h, "chi", o->chi)) != CA_OK)) WFAIL("ca_get_param");
    if (a->want_sums) {                                        /* clone_assignment(), R/inference-tflow.R:22-29, then one pass over the resident Y */
      if (!call) call = (int32_t*)malloc(sizeof(int32_t) * (size_t)N);
      for (int64_t n = 0; n < N; ++n) {
        int best = 0; double mx = o->clone_probs[n];          /* column-major N x C: element (n, c) at n + c N; first maximum like which.max */
        for (int c = 1; c < C; ++c) if (o->clone_probs[n + (int64_t)c * N] > mx) { mx = o->clone_probs[n + (int64_t)c * N]; best = c; }
        call[n] = mx >= a->call_prob ? best : -1;
      }
      if ((rc = ca_clone_gene_sums(h, call, o->T, o->Syy)) != CA_OK) WFAIL("ca_clone_gene_sums");
    }
  }
done:
#undef WFAIL
  /* a failed device dooms the whole call (the R thread raises the error once everybody has joined): make the siblings stop at their
   * next poll instead of running their remaining restarts into results that will be thrown away */
  if (a->rc != CA_OK && a->rc != CA_INTERRUPTED) *a->cancel = 1;
  if (h) ca_destroy(h);
  free(epsf); free(call); free(zeros);
  return NULL;
}

static SEXP alloc_real(R_xlen_t nrow, R_xlen_t ncol) {   /* ncol < 0: plain vector; zero-filled */
  SEXP v = ncol < 0 ? Rf_allocVector(REALSXP, nrow) : Rf_allocMatrix(REALSXP, (int)nrow, (int)ncol);
  memset(REAL(v), 0, sizeof(double) * (size_t)XLENGTH(v));
  return v;
}

SEXP C_clonealign_multifit(SEXP Y, SEXP L, SEXP psi0, SEXP psi_noise, SEXP loc0, SEXP X, SEXP extra, SEXP K_, SEXP S_, SEXP max_iter_,
                           SEXP rel_tol_, SEXP lr_, SEXP eps_, SEXP devices_, SEXP want_sums_, SEXP call_prob_) {
  worker_arg base;
  memset(&base, 0, sizeof(base));
  ca_problem* p = &base.p;
  p->N = Rf_nrows(Y); p->G = Rf_ncols(Y); p->C = Rf_ncols(L);
  p->K = Rf_asInteger(K_); p->S = Rf_asInteger(S_);
  p->P = Rf_isNull(X) ? 0 : Rf_ncols(X);
  p->layout = CA_COL_MAJOR;
  p->y_dtype = Rf_isInteger(Y) ? CA_I32 : CA_F64;                /* integer count matrices go over as they are: no double copy */
  p->Y = Rf_isInteger(Y) ? (const void*)INTEGER(Y) : (const void*)REAL(Y);
  p->L = REAL(L); p->loc0 = Rf_isNull(loc0) ? NULL : REAL(loc0);
  p->X = p->P > 0 ? REAL(X) : NULL;
  p->extra_loglik = Rf_isNull(extra) ? NULL : REAL(extra);
  ca_default_options(&base.o);
  base.o.learning_rate = Rf_asReal(lr_);
  const int by_noise = Rf_isNull(psi0);
  SEXP plist = by_noise ? psi_noise : psi0;
  if (p->K > 0 && Rf_isNull(plist)) Rf_error("clonealign_multifit: psi0 or psi_noise (a list with one N x K matrix per restart) is needed when K > 0");
  const int n_fits = p->K > 0 ? (int)XLENGTH(plist) : (Rf_isNull(eps_) ? 1 : (int)XLENGTH(eps_));
  const int n_dev = (int)XLENGTH(devices_);
  if (n_fits < 1 || n_dev < 1) Rf_error("clonealign_multifit: no restarts or no devices");
  if (!Rf_isNull(eps_) && (int)XLENGTH(eps_) != n_fits) Rf_error("clonealign_multifit: one eps vector per restart");
  const int max_iter = Rf_asInteger(max_iter_), want_sums = Rf_asInteger(want_sums_) != 0;
  const R_xlen_t per = (R_xlen_t)p->S * p->G, ndraw = 2 + 2 * (R_xlen_t)max_iter + 20;
  const double** psi_ptr = (const double**)R_alloc((size_t)n_fits, sizeof(double*));
  const double** eps_ptr = (const double**)R_alloc((size_t)n_fits, sizeof(double*));
  for (int r = 0; r < n_fits; ++r) {
    psi_ptr[r] = NULL; eps_ptr[r] = NULL;
    if (p->K > 0) {
      SEXP m = VECTOR_ELT(plist, r);
      if (XLENGTH(m) != (R_xlen_t)p->N * p->K) Rf_error("clonealign_multifit: restart %d: psi matrix is not N x K", r + 1);
      psi_ptr[r] = REAL(m);
    }
    if (!Rf_isNull(eps_)) {
      SEXP e = VECTOR_ELT(eps_, r);
      if (XLENGTH(e) < ndraw * per) Rf_error("clonealign_multifit: restart %d: eps stream too short", r + 1);
      eps_ptr[r] = REAL(e);
    }
  }
  /* every output is allocated here, on the R thread, before a worker exists */
  const char* names[] = {"mu", "clone_probs", "s", "alpha", "beta", "psi", "W", "chi", "elbo", "final_elbos", "T", "Syy", ""};
  SEXP fits = PROTECT(Rf_allocVector(VECSXP, n_fits));
  fit_out* outs = (fit_out*)R_alloc((size_t)n_fits, sizeof(fit_out));
  for (int r = 0; r < n_fits; ++r) {
    SEXP f = PROTECT(Rf_mkNamed(VECSXP, names));
    SET_VECTOR_ELT(fits, r, f);
    UNPROTECT(1);                                              /* (reachable from `fits` now) */
    fit_out* o = &outs[r];
    o->n_elbo = 0;
    o->mu = REAL(SET_VECTOR_ELT(f, 0, alloc_real(p->G, -1)));
    o->clone_probs = REAL(SET_VECTOR_ELT(f, 1, alloc_real(p->N, p->C)));
    o->s = REAL(SET_VECTOR_ELT(f, 2, alloc_real(p->N, -1)));
    o->alpha = REAL(SET_VECTOR_ELT(f, 3, alloc_real(p->C, -1)));
    o->beta = REAL(SET_VECTOR_ELT(f, 4, alloc_real(p->G, p->P)));
    o->psi = REAL(SET_VECTOR_ELT(f, 5, alloc_real(p->N, p->K)));
    o->W = REAL(SET_VECTOR_ELT(f, 6, alloc_real(p->G, p->K)));
    o->chi = REAL(SET_VECTOR_ELT(f, 7, alloc_real(p->K, -1)));
    o->elbo = REAL(SET_VECTOR_ELT(f, 8, alloc_real(max_iter + 1, -1)));
    o->finals = REAL(SET_VECTOR_ELT(f, 9, alloc_real(20, -1)));
    o->T = REAL(SET_VECTOR_ELT(f, 10, alloc_real(want_sums ? p->G : 0, want_sums ? p->C : 0)));
    o->Syy = REAL(SET_VECTOR_ELT(f, 11, alloc_real(want_sums ? p->G : 0, -1)));
  }
  volatile int cancel = 0;
  const int n_workers = n_dev < n_fits ? n_dev : n_fits;
  worker_arg* wa = (worker_arg*)R_alloc((size_t)n_workers, sizeof(worker_arg));
  pthread_t* th = (pthread_t*)R_alloc((size_t)n_workers, sizeof(pthread_t));
  int started = 0;
  for (int d = 0; d < n_workers; ++d) {
    wa[d] = base;
    wa[d].o.device = INTEGER(devices_)[d];
    wa[d].o.seed = base.o.seed + 1000003ull * (uint64_t)d;
    wa[d].n_fits = n_fits; wa[d].n_dev = n_workers; wa[d].lane = d; wa[d].max_iter = max_iter; wa[d].want_sums = want_sums;
    wa[d].rel_tol = Rf_asReal(rel_tol_); wa[d].call_prob = Rf_asReal(call_prob_);
    wa[d].psi0 = by_noise ? NULL : psi_ptr; wa[d].noise = by_noise ? psi_ptr : NULL; wa[d].eps = Rf_isNull(eps_) ? NULL : eps_ptr;
    wa[d].loc0 = p->loc0; wa[d].out = outs; wa[d].cancel = &cancel; wa[d].rc = CA_ERR_STATE; wa[d].err[0] = 0;
    if (pthread_create(&th[d], NULL, worker_main, &wa[d]) != 0) { cancel = 1; break; }
    ++started;
  }
  /* the R thread: wait for the workers, looking for Ctrl-C in between (never unwinding while a worker is alive) */
  int interrupted = 0;
  for (int d = 0; d < started; ++d) {
    for (;;) {
      struct timespec ts;
      clock_gettime(CLOCK_REALTIME, &ts);
      ts.tv_nsec += 50 * 1000 * 1000;
      if (ts.tv_nsec >= 1000000000L) { ts.tv_sec += 1; ts.tv_nsec -= 1000000000L; }
      if (pthread_timedjoin_np(th[d], NULL, &ts) == 0) break;
      if (!interrupted && R_ToplevelExec(check_interrupt, NULL) == FALSE) { interrupted = 1; cancel = 1; }
    }
  }
  if (started < n_workers) { UNPROTECT(1); Rf_error("clonealign_multifit: cannot start a worker thread"); }
  if (interrupted) { UNPROTECT(1); Rf_error("clonealign: interrupted"); }
  for (int pass = 0; pass < 2; ++pass)                       /* the device that FAILED first; siblings it stopped report CA_INTERRUPTED */
    for (int d = 0; d < n_workers; ++d)
      if (wa[d].rc != CA_OK && (pass == 1 || wa[d].rc != CA_INTERRUPTED)) {
        char msg[600]; snprintf(msg, sizeof(msg), "device %d: %s", wa[d].o.device, wa[d].rc == CA_INTERRUPTED ? "stopped" : wa[d].err);
        UNPROTECT(1); Rf_error("clonealign_multifit: %s", msg);
      }
  for (int r = 0; r < n_fits; ++r) {                         /* the trace is as long as the loop ran (:414 may stop it early) */
    SEXP f = VECTOR_ELT(fits, r);
    SET_VECTOR_ELT(f, 8, Rf_xlengthgets(VECTOR_ELT(f, 8), outs[r].n_elbo));
  }
  UNPROTECT(1);
  return fits;
}

/* preprocess_for_clonealign()'s filters (R/preprocess.R:93-147) as masks: the two O(N G) statistics are taken on the device from the
 * raw matrix as R holds it (ca_preprocess); the caller subsets with the masks, or hands them to the fit as selection lists.
 *   .Call("C_clonealign_preprocess", Y, L, min_counts_per_gene, min_counts_per_cell, remove_outlying_genes, nmads, max_copy_number,
 *         remove_genes_same_copy_number, device)  ->  list(keep_gene = logical G, keep_cell = logical N, gene_sums = G, cell_sums = N) */
SEXP C_clonealign_preprocess(SEXP Y, SEXP L, SEXP min_gene_, SEXP min_cell_, SEXP outl_, SEXP nmads_, SEXP max_cn_, SEXP same_cn_, SEXP device_) {
  const int64_t N = Rf_nrows(Y); const int G = Rf_ncols(Y), C = Rf_ncols(L);
  if (Rf_nrows(L) != G) Rf_error("clonealign_preprocess: nrow(L) must equal ncol(Y)");
  ca_preprocess_params pp;
  pp.min_counts_per_gene = Rf_asReal(min_gene_); pp.min_counts_per_cell = Rf_asReal(min_cell_);
  pp.remove_outlying_genes = Rf_asInteger(outl_) != 0; pp.remove_genes_same_copy_number = Rf_asInteger(same_cn_) != 0;
  pp.nmads = Rf_asReal(nmads_); pp.max_copy_number = Rf_asReal(max_cn_);
  const char* names[] = {"keep_gene", "keep_cell", "gene_sums", "cell_sums", ""};
  SEXP out = PROTECT(Rf_mkNamed(VECSXP, names));
  SEXP kg = SET_VECTOR_ELT(out, 0, Rf_allocVector(LGLSXP, G)), kc = SET_VECTOR_ELT(out, 1, Rf_allocVector(LGLSXP, (R_xlen_t)N));
  SEXP gs = SET_VECTOR_ELT(out, 2, alloc_real(G, -1)), cs = SET_VECTOR_ELT(out, 3, alloc_real((R_xlen_t)N, -1));
  uint8_t* mg = (uint8_t*)R_alloc((size_t)G, 1); uint8_t* mc = (uint8_t*)R_alloc((size_t)N, 1);
  char err[256]; err[0] = 0;
  const int rc = ca_preprocess(N, G, C, CA_COL_MAJOR, Rf_isInteger(Y) ? CA_I32 : CA_F64, 0, Rf_isInteger(Y) ? (const void*)INTEGER(Y) : (const void*)REAL(Y),
                               REAL(L), &pp, Rf_asInteger(device_), mg, mc, REAL(gs), REAL(cs), err);
  if (rc != CA_OK) { UNPROTECT(1); Rf_error("clonealign_preprocess: %s", err); }
  for (int g = 0; g < G; ++g) LOGICAL(kg)[g] = mg[g] != 0;
  for (int64_t n = 0; n < N; ++n) LOGICAL(kc)[n] = mc[n] != 0;
  UNPROTECT(1);
  return out;
}

/* The parameter-free allele-specific addend (R/allele-specific.R:17-58 as used at R/inference-tflow.R:166-187) on the device:
 *   .Call("C_clonealign_allele_loglik", clone_allele (V x C), cov (N x V), ref (N x V), device)  ->  numeric matrix N x C
 * (the `extra` argument of the fit entry points; clone_probs_from_snv of :436-440 follows from it in R). */
SEXP C_clonealign_allele_loglik(SEXP clone_allele, SEXP cov, SEXP ref, SEXP device_) {
  const int V = Rf_nrows(clone_allele), C = Rf_ncols(clone_allele);
  const int64_t N = Rf_nrows(cov);
  if (Rf_ncols(cov) != V || Rf_nrows(ref) != N || Rf_ncols(ref) != V) Rf_error("clonealign_allele_loglik: cov and ref must be N x V, clone_allele V x C");
  SEXP out = PROTECT(alloc_real((R_xlen_t)N, C));
  char err[256]; err[0] = 0;
  if (ca_allele_loglik(N, V, C, CA_COL_MAJOR, REAL(clone_allele), REAL(cov), REAL(ref), Rf_asInteger(device_), REAL(out), err) != CA_OK) {
    UNPROTECT(1); Rf_error("clonealign_allele_loglik: %s", err);
  }
  UNPROTECT(1);
  return out;
}

/* registration table: src/init.c */

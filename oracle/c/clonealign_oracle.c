/*
 * ORACLE (test infrastructure, NOT product code): plain C + OpenMP port of the fused float64
 * restatement in oracle/fused_numpy.py, i.e. of the reference model R/inference-tflow.R:240-346
 * (ELBO :306-336, gamma init :338-342, TF1 Adam :345-346).  Used (a) as a second, independent
 * CPU checker that scales to the 10k x 2k x 4 configuration and beyond, and (b) as the CPU
 * baseline ("kind": "port") that bench.py times beside the GPU on the host cores.
 * It is never linked into, imported by, or called from the product path.
 *
 * PARITY STATUS: parity unpinned against the TensorFlow path itself (see oracle/literal_torch.py);
 * pinned against the literal autodiff oracle through the committed goldens (tests/test_oracle_c.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define LOG2PI 1.8378770664093453

typedef struct co_model {
  long N; int G, C, K, P, S, D, f32;
  double *Y, *L, *X, *A, *cn, *s, *colsum, *YtX;
  /* variables and Adam slots, in VAR order: W[G,K] v[K] psi[N,K] beta[G,P] alpha_u[C] loc[G] ls[G] glogit[N,C] */
  double *var[8], *m[8], *vv[8], *grad[8];
  long len[8];
  double lr, b1, b2, eps_adam, b1p, b2p;
  double terms[3];
  int nthreads;
} co_model;

enum { V_W = 0, V_v, V_psi, V_beta, V_alpha, V_loc, V_ls, V_gl };
static const char* VAR_NAMES[8] = {"W", "v", "psi", "beta", "alpha_unconstr", "loc", "ls", "gamma_logits"};

static double softplus(double x) { return x > 0 ? x + log1p(exp(-x)) : log1p(exp(x)); }
static double sigmoid(double x) { return 1.0 / (1.0 + exp(-x)); }
static double rnd(const co_model* m, double x) { return m->f32 ? (double)(float)x : x; }

int co_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

co_model* co_create(long N, int G, int C, int K, int P, int S, const double* Y, const double* L, const double* psi0,
                    const double* loc0, const double* X, const double* extra, double lr, int f32) {
  co_model* m = (co_model*)calloc(1, sizeof(co_model));
  m->N = N; m->G = G; m->C = C; m->K = K; m->P = P; m->S = S; m->f32 = f32;
  m->D = K > 0 ? K + P : 0;
  m->lr = lr; m->b1 = 0.9; m->b2 = 0.999; m->eps_adam = 1e-8;
  m->b1p = rnd(m, m->b1); m->b2p = rnd(m, m->b2);
  m->nthreads = co_num_threads();
  m->Y = (double*)malloc(sizeof(double) * N * G); memcpy(m->Y, Y, sizeof(double) * N * G);
  m->L = (double*)malloc(sizeof(double) * G * C); memcpy(m->L, L, sizeof(double) * G * C);
  m->X = NULL;
  if (P > 0) { m->X = (double*)malloc(sizeof(double) * N * P); memcpy(m->X, X, sizeof(double) * N * P); }
  m->A = (double*)calloc((size_t)N * C, sizeof(double));
  m->cn = (double*)calloc(N, sizeof(double));
  m->s = (double*)calloc(N, sizeof(double));
  m->colsum = (double*)calloc(G, sizeof(double));
  m->YtX = (double*)calloc((size_t)G * (P > 0 ? P : 1), sizeof(double));
  const long len[8] = {(long)G * K, K, N * K, (long)G * P, C, G, G, N * C};
  for (int i = 0; i < 8; ++i) {
    m->len[i] = len[i];
    size_t n = len[i] > 0 ? (size_t)len[i] : 1;
    m->var[i] = (double*)calloc(n, sizeof(double));
    m->m[i] = (double*)calloc(n, sizeof(double));
    m->vv[i] = (double*)calloc(n, sizeof(double));
    m->grad[i] = (double*)calloc(n, sizeof(double));
  }
  for (long i = 0; i < N * K; ++i) m->var[V_psi][i] = rnd(m, psi0[i]);
  for (int g = 0; g < G; ++g) m->var[V_loc][g] = rnd(m, loc0[g]);
#pragma omp parallel for schedule(static)
  for (long n = 0; n < N; ++n) {
    const double* y = m->Y + n * G;
    double s = 0, lg = 0;
    for (int g = 0; g < G; ++g) { s += y[g]; lg += lgamma(y[g] + 1.0); }
    m->s[n] = s;
    m->cn[n] = lgamma(s + 1.0) - lg;
    for (int c = 0; c < C; ++c) {
      double a = 0;
      for (int g = 0; g < G; ++g) if (y[g] != 0.0) a += y[g] * log(m->L[(long)g * C + c]);
      m->A[n * C + c] = a + (extra ? extra[n * C + c] : 0.0);
    }
  }
  for (long n = 0; n < N; ++n) {
    const double* y = m->Y + n * G;
    for (int g = 0; g < G; ++g) {
      m->colsum[g] += y[g];
      if (m->D > 0) for (int p = 0; p < P; ++p) m->YtX[(long)g * P + p] += y[g] * m->X[n * P + p];
    }
  }
  return m;
}

void co_destroy(co_model* m) {
  if (!m) return;
  free(m->Y); free(m->L); free(m->X); free(m->A); free(m->cn); free(m->s); free(m->colsum); free(m->YtX);
  for (int i = 0; i < 8; ++i) { free(m->var[i]); free(m->m[i]); free(m->vv[i]); free(m->grad[i]); }
  free(m);
}

static int var_index(const char* name) {
  for (int i = 0; i < 8; ++i) if (!strcmp(name, VAR_NAMES[i])) return i;
  return -1;
}
long co_get(co_model* m, const char* name, double* out, int grad) {
  if (!strcmp(name, "s")) { memcpy(out, m->s, sizeof(double) * m->N); return m->N; }
  int i = var_index(name);
  if (i < 0) return -1;
  memcpy(out, grad ? m->grad[i] : m->var[i], sizeof(double) * m->len[i]);
  return m->len[i];
}
long co_set(co_model* m, const char* name, const double* in) {
  int i = var_index(name);
  if (i < 0) return -1;
  for (long j = 0; j < m->len[i]; ++j) m->var[i][j] = rnd(m, in[j]);
  return m->len[i];
}

/* mode 0: ELBO only; 1: ELBO + gradients; 2: gamma_init (overwrites the logits) */
static double pass(co_model* m, const float* eps, int mode) {
  const long N = m->N; const int G = m->G, C = m->C, K = m->K, P = m->P, S = m->S, D = m->D;
  const double *W = m->var[V_W], *v = m->var[V_v], *psi = m->var[V_psi], *beta = m->var[V_beta], *au = m->var[V_alpha],
               *loc = m->var[V_loc], *ls = m->var[V_ls];
  double* gl = m->var[V_gl];
  double* x = (double*)malloc(sizeof(double) * S * G);
  double* mu = (double*)malloc(sizeof(double) * S * G);
  double* M = (double*)malloc(sizeof(double) * S * G * C);
  double gene0 = 0, gene1 = 0, gene2 = 0;
  for (int s = 0; s < S; ++s)
    for (int g = 0; g < G; ++g) {
      const double e = (double)eps[(long)s * G + g];
      const double xx = loc[g] + exp(ls[g]) * e;
      const double mm = softplus(xx), lm = log(mm);
      x[s * G + g] = xx; mu[s * G + g] = mm;
      for (int c = 0; c < C; ++c) M[((long)s * G + g) * C + c] = mm * m->L[(long)g * C + c];
      gene0 += m->colsum[g] * lm / S;
      gene1 += (-0.5 * lm * lm - 0.5 * LOG2PI) / S;
      gene2 += (-0.5 * e * e - ls[g] - 0.5 * LOG2PI + (mm - xx)) / S;
    }
  if (D > 0) for (int g = 0; g < G; ++g) for (int p = 0; p < P; ++p) gene0 += beta[(long)g * P + p] * m->YtX[(long)g * P + p];
  /* log alpha */
  double amx = -INFINITY, ase = 0;
  for (int c = 0; c < C; ++c) amx = fmax(amx, au[c]);
  for (int c = 0; c < C; ++c) ase += exp(au[c] - amx);
  double la[256];
  for (int c = 0; c < C; ++c) la[c] = au[c] - amx - log(ase);
  const int T = m->nthreads;
  const int WG = S + D + K;  /* per-thread per-gene accumulators: dmu[S], dV[D], YtPsi[K] */
  double* tg = (double*)calloc((size_t)T * G * (WG > 0 ? WG : 1), sizeof(double));
  double* tsum = (double*)calloc((size_t)T * (3 + C), sizeof(double));
  double* gpsi = m->grad[V_psi]; double* ggl = m->grad[V_gl];
#pragma omp parallel num_threads(T)
  {
#ifdef _OPENMP
    const int t = omp_get_thread_num();
#else
    const int t = 0;
#endif
    double* E = (double*)malloc(sizeof(double) * G);
    double* my = tg + (size_t)t * G * WG;
    double* ms = tsum + (size_t)t * (3 + C);
    double Z[64 * 8], coef[64 * 8], dF[16], lg[256], gam[256], f[256];
    /* contiguous cell ranges per thread: deterministic for a fixed thread count */
    const long n0 = N * t / T, n1 = N * (t + 1) / T;
    for (long n = n0; n < n1; ++n) {
      const double* y = m->Y + n * G;
      for (int g = 0; g < G; ++g) {
        double eta = 0;
        for (int k = 0; k < (D > 0 ? K : 0); ++k) eta += psi[n * K + k] * W[(long)g * K + k];
        for (int p = 0; p < (D > 0 ? P : 0); ++p) eta += m->X[n * P + p] * beta[(long)g * P + p];
        E[g] = D > 0 ? exp(eta) : 1.0;
      }
      for (int i = 0; i < S * C; ++i) Z[i] = 0;
      for (int s = 0; s < S; ++s)
        for (int g = 0; g < G; ++g) {
          const double e = E[g]; const double* mg = M + ((long)s * G + g) * C;
          for (int c = 0; c < C; ++c) Z[s * C + c] += e * mg[c];
        }
      double mx = -INFINITY, se = 0;
      for (int c = 0; c < C; ++c) mx = fmax(mx, gl[n * C + c]);
      for (int c = 0; c < C; ++c) se += exp(gl[n * C + c] - mx);
      const double lse = mx + log(se);
      if (mode == 2) {
        double ll[256], m2 = -INFINITY, s2 = 0;
        for (int c = 0; c < C; ++c) {
          double lz = 0; for (int s = 0; s < S; ++s) lz += log(Z[s * C + c]);
          ll[c] = S * m->A[n * C + c] - m->s[n] * lz; m2 = fmax(m2, ll[c]);
        }
        for (int c = 0; c < C; ++c) s2 += exp(ll[c] - m2);
        for (int c = 0; c < C; ++c) gl[n * C + c] = rnd(m, ll[c] - m2 - log(s2));
        continue;
      }
      double ee = m->cn[n], pr = 0, q = 0, fbar = 0;
      for (int c = 0; c < C; ++c) {
        lg[c] = gl[n * C + c] - lse; gam[c] = exp(lg[c]);
        double lz = 0; for (int s = 0; s < S; ++s) lz += log(Z[s * C + c]);
        const double llp = m->A[n * C + c] - m->s[n] * lz / S;
        f[c] = llp + la[c] - lg[c];
        if (gam[c] != 0.0) { ee += gam[c] * llp; pr += gam[c] * la[c]; q += gam[c] * lg[c]; fbar += gam[c] * f[c]; }
        ms[3 + c] += gam[c];
        for (int s = 0; s < S; ++s) coef[s * C + c] = -gam[c] * m->s[n] / (S * Z[s * C + c]);
      }
      for (int k = 0; k < K; ++k) {
        double yw = 0; for (int g = 0; g < G; ++g) yw += y[g] * W[(long)g * K + k];
        ee += psi[n * K + k] * yw; pr += -0.5 * psi[n * K + k] * psi[n * K + k] - 0.5 * LOG2PI;
        if (mode == 1) gpsi[n * K + k] = yw - psi[n * K + k];
      }
      ms[0] += ee; ms[1] += pr; ms[2] += q;
      if (mode != 1) continue;
      for (int c = 0; c < C; ++c) ggl[n * C + c] = gam[c] != 0.0 ? gam[c] * (f[c] - fbar) : 0.0;
      for (int d = 0; d < D; ++d) dF[d] = 0;
      for (int g = 0; g < G; ++g) {
        double deta = 0; double* acc = my + (size_t)g * WG;
        for (int s = 0; s < S; ++s) {
          double tt = 0; for (int c = 0; c < C; ++c) tt += coef[s * C + c] * m->L[(long)g * C + c];
          const double u = E[g] * tt;
          acc[s] += u; deta += mu[s * G + g] * u;
        }
        for (int k = 0; k < K; ++k) {
          dF[k] += deta * W[(long)g * K + k]; acc[S + k] += deta * psi[n * K + k]; acc[S + D + k] += y[g] * psi[n * K + k];
        }
        for (int p = 0; p < (D > 0 ? P : 0); ++p) acc[S + K + p] += deta * m->X[n * P + p];
      }
      for (int k = 0; k < K; ++k) gpsi[n * K + k] += dF[k];
    }
    free(E);
  }
  double cs[3] = {0, 0, 0}; double sg[256];
  for (int c = 0; c < C; ++c) sg[c] = 0;
  for (int t = 0; t < T; ++t) { for (int j = 0; j < 3; ++j) cs[j] += tsum[t * (3 + C) + j]; for (int c = 0; c < C; ++c) sg[c] += tsum[t * (3 + C) + 3 + c]; }
  double EE = cs[0] + gene0, Ep = cs[1] + gene1, Eq = cs[2] + gene2;
  const double conc = 1.0 / C;
  Ep += -(C * lgamma(conc) - lgamma(1.0));
  double dla[256], dlas = 0, al[256];
  for (int c = 0; c < C; ++c) { al[c] = exp(la[c]); Ep += (conc - 1.0) * log(al[c] + 1e-3); dla[c] = sg[c] + (conc - 1.0) * al[c] / (al[c] + 1e-3); dlas += dla[c]; }
  for (int k = 0; k < K; ++k) {
    double w2 = 0; for (int g = 0; g < G; ++g) w2 += W[(long)g * K + k] * W[(long)g * K + k];
    const double chi = exp(v[k]);
    Ep += -0.5 * chi * w2 + G * (0.5 * v[k] - 0.5 * LOG2PI) + (v[k] - chi);
    m->grad[V_v][k] = -0.5 * chi * w2 + 0.5 * G + 1.0 - chi;
  }
  m->terms[0] = EE; m->terms[1] = Ep; m->terms[2] = Eq;
  if (mode == 1) {
    for (int c = 0; c < C; ++c) m->grad[V_alpha][c] = dla[c] - al[c] * dlas;
    for (int g = 0; g < G; ++g) {
      double acc[64]; for (int j = 0; j < WG; ++j) { acc[j] = 0; for (int t = 0; t < T; ++t) acc[j] += tg[((size_t)t * G + g) * WG + j]; }
      double gl_ = 0, gs_ = 0;
      for (int s = 0; s < S; ++s) {
        const double mm = mu[s * G + g], xx = x[s * G + g], e = (double)eps[(long)s * G + g];
        const double dmu = m->colsum[g] / (S * mm) + acc[s] - log(mm) / (S * mm);
        const double sig = sigmoid(xx), dx = dmu * sig + (1.0 - sig) / S;
        gl_ += dx; gs_ += dx * e * exp(ls[g]);
      }
      m->grad[V_loc][g] = gl_; m->grad[V_ls][g] = gs_ + 1.0;
      for (int k = 0; k < K; ++k) m->grad[V_W][(long)g * K + k] = acc[S + D + k] + acc[S + k] - exp(v[k]) * W[(long)g * K + k];
      for (int p = 0; p < P; ++p) m->grad[V_beta][(long)g * P + p] = D > 0 ? m->YtX[(long)g * P + p] + acc[S + K + p] : 0.0;
    }
  }
  free(tg); free(tsum); free(x); free(mu); free(M);
  return EE + Ep - Eq;
}

double co_elbo(co_model* m, const float* eps, double* terms) {
  const double e = pass(m, eps, 0);
  if (terms) memcpy(terms, m->terms, sizeof(m->terms));
  return e;
}
void co_gamma_init(co_model* m, const float* eps) { pass(m, eps, 2); }
double co_gradients(co_model* m, const float* eps) { return pass(m, eps, 1); }
void co_step(co_model* m, const float* eps) {
  pass(m, eps, 1);
  const double lr_t = rnd(m, rnd(m, m->lr) * rnd(m, sqrt(rnd(m, 1.0 - m->b2p))) / rnd(m, 1.0 - m->b1p));
  const double b1 = rnd(m, m->b1), b2 = rnd(m, m->b2), omb1 = rnd(m, 1.0 - b1), omb2 = rnd(m, 1.0 - b2), ea = rnd(m, m->eps_adam);
  for (int i = 0; i < 8; ++i)
#pragma omp parallel for schedule(static)
    for (long j = 0; j < m->len[i]; ++j) {
      const double g = rnd(m, -m->grad[i][j]);
      m->m[i][j] = rnd(m, rnd(m, b1 * m->m[i][j]) + rnd(m, omb1 * g));
      m->vv[i][j] = rnd(m, rnd(m, b2 * m->vv[i][j]) + rnd(m, rnd(m, omb2 * g) * g));
      m->var[i][j] = rnd(m, m->var[i][j] - rnd(m, rnd(m, lr_t * m->m[i][j]) / rnd(m, rnd(m, sqrt(m->vv[i][j])) + ea)));
    }
  m->b1p = rnd(m, m->b1p * b1); m->b2p = rnd(m, m->b2p * b2);
}

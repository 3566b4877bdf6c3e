/*
 * ORACLE-SIDE CPU BASELINE (test infrastructure, NOT product code): float32, SIMD-friendly port of the fused restatement
 * in oracle/fused_numpy.py / oracle/c/clonealign_oracle.c, i.e. of the reference model R/inference-tflow.R:240-346 (ELBO
 * :306-336, gamma init :338-342, TF1 Adam :345-346) -- the arithmetic a well-built CPU implementation would do: float32 like
 * the reference's default dtype, every per-gene loop unit-stride so that gcc vectorises it (exp / log through glibc's libmvec),
 * AVX-512 picked at run time where the host has it, one contiguous cell range per OpenMP thread.  S = 1 (the reference's
 * default, the bench's configuration).  bench.py times it beside the scalar float64 port as the stronger CPU baseline;
 * tests/test_oracle_c.py holds it to the float64 port.  Never linked into, imported by, or called from the product path.
 *
 * PARITY STATUS: parity unpinned against the TensorFlow path itself (see oracle/literal_torch.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define LOG2PI 1.8378770664093453
#define CLONES __attribute__((target_clones("arch=skylake-avx512", "arch=haswell", "default")))

typedef struct cs_model {
  long N; int G, C, K, P, D, T;
  float *Y;            /* [N][G] */
  float *Lt;           /* [C][G] copy numbers, clone-major so the per-gene loops are unit stride */
  float *X;            /* [N][P] */
  double *A, *cn, *s, *colsum, *YtX;
  /* variables + Adam slots, VAR order: W[K][G] v[K] psi[N][K] beta[P][G] alpha_u[C] loc[G] ls[G] glogit[N][C]  (W and beta gene-minor) */
  float *var[8], *m[8], *vv[8], *grad[8];
  long len[8];
  float lr, b1p, b2p;
  double terms[3];
  /* per-pass scratch */
  float *mu, *xs, *Mt;          /* [G], [G], [C][G] */
  float *tacc;                  /* [T][1 + D + K][G] per-thread gene accumulators: dmu, dV[D], YtPsi[K] */
  double *tsum;                 /* [T][3 + C] */
  float *tE;                    /* [T][2][G] */
} cs_model;

enum { V_W = 0, V_v, V_psi, V_beta, V_alpha, V_loc, V_ls, V_gl };
static const char* VAR_NAMES[8] = {"W", "v", "psi", "beta", "alpha_unconstr", "loc", "ls", "gamma_logits"};

int cs_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

static void* xcalloc(size_t n, size_t sz) { void* p = NULL; if (posix_memalign(&p, 64, (n ? n : 1) * sz)) return NULL; memset(p, 0, (n ? n : 1) * sz); return p; }

cs_model* cs_create(long N, int G, int C, int K, int P, const double* Y, const double* L, const double* psi0, const double* loc0,
                    const double* X, const double* extra, double lr) {
  cs_model* m = (cs_model*)calloc(1, sizeof(cs_model));
  m->N = N; m->G = G; m->C = C; m->K = K; m->P = P; m->D = K > 0 ? K + P : 0; m->T = cs_num_threads();
  m->lr = (float)lr; m->b1p = 0.9f; m->b2p = 0.999f;
  m->Y = (float*)xcalloc((size_t)N * G, sizeof(float));
  m->Lt = (float*)xcalloc((size_t)C * G, sizeof(float));
  m->X = (float*)xcalloc((size_t)N * (P > 0 ? P : 1), sizeof(float));
  m->A = (double*)xcalloc((size_t)N * C, sizeof(double));
  m->cn = (double*)xcalloc(N, sizeof(double)); m->s = (double*)xcalloc(N, sizeof(double));
  m->colsum = (double*)xcalloc(G, sizeof(double)); m->YtX = (double*)xcalloc((size_t)G * (P > 0 ? P : 1), sizeof(double));
  for (int g = 0; g < G; ++g) for (int c = 0; c < C; ++c) m->Lt[(size_t)c * G + g] = (float)L[(size_t)g * C + c];
  for (long i = 0; i < N * (long)P; ++i) m->X[i] = (float)X[i];
  const long len[8] = {(long)G * K, K, N * K, (long)G * P, C, G, G, N * C};
  for (int i = 0; i < 8; ++i) {
    m->len[i] = len[i];
    m->var[i] = (float*)xcalloc(len[i], sizeof(float)); m->m[i] = (float*)xcalloc(len[i], sizeof(float));
    m->vv[i] = (float*)xcalloc(len[i], sizeof(float)); m->grad[i] = (float*)xcalloc(len[i], sizeof(float));
  }
  for (long i = 0; i < N * K; ++i) m->var[V_psi][i] = (float)psi0[i];
  for (int g = 0; g < G; ++g) m->var[V_loc][g] = (float)loc0[g];
  double* logL = (double*)malloc(sizeof(double) * G * C);
  for (long i = 0; i < (long)G * C; ++i) logL[i] = log(L[i]);
#pragma omp parallel for schedule(static)
  for (long n = 0; n < N; ++n) {
    const double* y = Y + n * G;
    float* yf = m->Y + n * G;
    double s = 0, lg = 0;
    for (int g = 0; g < G; ++g) { yf[g] = (float)y[g]; s += y[g]; lg += lgamma(y[g] + 1.0); }
    m->s[n] = s; m->cn[n] = lgamma(s + 1.0) - lg;
    for (int c = 0; c < C; ++c) {
      double a = 0;
      for (int g = 0; g < G; ++g) if (y[g] != 0.0) a += y[g] * logL[(long)g * C + c];
      m->A[n * C + c] = a + (extra ? extra[n * C + c] : 0.0);
    }
  }
  free(logL);
  for (long n = 0; n < N; ++n) {
    const double* y = Y + n * G;
    for (int g = 0; g < G; ++g) {
      m->colsum[g] += y[g];
      if (m->D > 0) for (int p = 0; p < P; ++p) m->YtX[(long)g * P + p] += y[g] * X[n * P + p];
    }
  }
  m->mu = (float*)xcalloc(G, sizeof(float)); m->xs = (float*)xcalloc(G, sizeof(float)); m->Mt = (float*)xcalloc((size_t)C * G, sizeof(float));
  m->tacc = (float*)xcalloc((size_t)m->T * (1 + m->D + K) * G, sizeof(float));
  m->tsum = (double*)xcalloc((size_t)m->T * (3 + C), sizeof(double));
  m->tE = (float*)xcalloc((size_t)m->T * 2 * G, sizeof(float));
  return m;
}

void cs_destroy(cs_model* m) {
  if (!m) return;
  free(m->Y); free(m->Lt); free(m->X); free(m->A); free(m->cn); free(m->s); free(m->colsum); free(m->YtX);
  for (int i = 0; i < 8; ++i) { free(m->var[i]); free(m->m[i]); free(m->vv[i]); free(m->grad[i]); }
  free(m->mu); free(m->xs); free(m->Mt); free(m->tacc); free(m->tsum); free(m->tE);
  free(m);
}

static int var_index(const char* name) {
  for (int i = 0; i < 8; ++i) if (!strcmp(name, VAR_NAMES[i])) return i;
  return -1;
}
/* W and beta are stored gene-minor; the accessors speak the oracle's [G][K] / [G][P] layout */
long cs_get(cs_model* m, const char* name, double* out, int grad) {
  if (!strcmp(name, "s")) { memcpy(out, m->s, sizeof(double) * m->N); return m->N; }
  const int i = var_index(name);
  if (i < 0) return -1;
  const float* src = grad ? m->grad[i] : m->var[i];
  if (i == V_W || i == V_beta) {
    const int R = i == V_W ? m->K : m->P;
    for (int r = 0; r < R; ++r) for (int g = 0; g < m->G; ++g) out[(long)g * R + r] = src[(long)r * m->G + g];
  } else for (long j = 0; j < m->len[i]; ++j) out[j] = src[j];
  return m->len[i];
}

/* One contiguous range of cells: forward (E, Z, per-cell ELBO terms) and, in mode 1, the backward accumulation.  mode 0: ELBO,
   1: ELBO + gradients, 2: gamma init. */
CLONES void cs_cell_range(cs_model* m, int t, long n0, long n1, int mode, const float* la) {
  const int G = m->G, C = m->C, K = m->K, P = m->P, D = m->D;
  const float *W = m->var[V_W], *psi = m->var[V_psi], *beta = m->var[V_beta], *mu = m->mu;
  float* gl = m->var[V_gl];
  float* E = m->tE + (size_t)t * 2 * G;
  float* U = E + G;
  float* acc = m->tacc + (size_t)t * (1 + D + K) * G;
  double* ms = m->tsum + (size_t)t * (3 + C);
  float Z[64], coef[64], lg[64], gam[64], f[64], llp[64];
  for (long n = n0; n < n1; ++n) {
    const float* y = m->Y + n * G;
    if (D > 0) {
      const float p0 = psi[n * K];
      const float* W0 = W;
      for (int g = 0; g < G; ++g) E[g] = p0 * W0[g];
      for (int k = 1; k < K; ++k) { const float pk = psi[n * K + k]; const float* Wk = W + (size_t)k * G; for (int g = 0; g < G; ++g) E[g] += pk * Wk[g]; }
      for (int p = 0; p < P; ++p) { const float xp = m->X[n * P + p]; const float* bp = beta + (size_t)p * G; for (int g = 0; g < G; ++g) E[g] += xp * bp[g]; }
      for (int g = 0; g < G; ++g) E[g] = expf(E[g]);
    } else for (int g = 0; g < G; ++g) E[g] = 1.f;
    for (int c = 0; c < C; ++c) {
      const float* Mc = m->Mt + (size_t)c * G;
      float z = 0.f;
      for (int g = 0; g < G; ++g) z += E[g] * Mc[g];
      Z[c] = z;
    }
    float mx = -3.0e38f, se = 0.f;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, gl[n * C + c]);
    for (int c = 0; c < C; ++c) se += expf(gl[n * C + c] - mx);
    const float lse = mx + logf(se);
    const float sn = (float)m->s[n];
    if (mode == 2) {
      float ll[64], m2 = -3.0e38f, s2 = 0.f;
      for (int c = 0; c < C; ++c) { ll[c] = (float)m->A[n * C + c] - sn * logf(Z[c]); m2 = fmaxf(m2, ll[c]); }
      for (int c = 0; c < C; ++c) s2 += expf(ll[c] - m2);
      for (int c = 0; c < C; ++c) gl[n * C + c] = ll[c] - m2 - logf(s2);
      continue;
    }
    double ee = m->cn[n], pr = 0, q = 0;
    float fbar = 0.f;
    for (int c = 0; c < C; ++c) {
      lg[c] = gl[n * C + c] - lse; gam[c] = expf(lg[c]);
      llp[c] = (float)m->A[n * C + c] - sn * logf(Z[c]);
      f[c] = llp[c] + la[c] - lg[c];
      if (gam[c] != 0.f) { ee += (double)gam[c] * llp[c]; pr += (double)gam[c] * la[c]; q += (double)gam[c] * lg[c]; fbar += gam[c] * f[c]; }
      ms[3 + c] += gam[c];
      coef[c] = -gam[c] * sn / Z[c];
    }
    for (int k = 0; k < K; ++k) {
      const float* Wk = W + (size_t)k * G;
      float yw = 0.f;
      for (int g = 0; g < G; ++g) yw += y[g] * Wk[g];
      const float pk = psi[n * K + k];
      ee += (double)pk * yw; pr += -0.5 * (double)pk * pk - 0.5 * LOG2PI;
      if (mode == 1) m->grad[V_psi][n * K + k] = yw - pk;
    }
    ms[0] += ee; ms[1] += pr; ms[2] += q;
    if (mode != 1) continue;
    for (int c = 0; c < C; ++c) m->grad[V_gl][n * C + c] = gam[c] != 0.f ? gam[c] * (f[c] - fbar) : 0.f;
    /* u_g = E_g * sum_c coef_c L_gc ; dmu_g += u_g ; deta_g = mu_g u_g */
    {
      const float c0 = coef[0]; const float* L0 = m->Lt;
      for (int g = 0; g < G; ++g) U[g] = c0 * L0[g];
      for (int c = 1; c < C; ++c) { const float cc = coef[c]; const float* Lc = m->Lt + (size_t)c * G; for (int g = 0; g < G; ++g) U[g] += cc * Lc[g]; }
      float* a0 = acc;
      for (int g = 0; g < G; ++g) { const float u = E[g] * U[g]; a0[g] += u; U[g] = mu[g] * u; }
    }
    for (int k = 0; k < K; ++k) {
      const float* Wk = W + (size_t)k * G; const float pk = psi[n * K + k];
      float* aw = acc + (size_t)(1 + k) * G; float* ay = acc + (size_t)(1 + D + k) * G;
      float dF = 0.f;
      for (int g = 0; g < G; ++g) { dF += U[g] * Wk[g]; aw[g] += U[g] * pk; ay[g] += y[g] * pk; }
      m->grad[V_psi][n * K + k] += dF;
    }
    for (int p = 0; p < (D > 0 ? P : 0); ++p) {
      const float xp = m->X[n * P + p]; float* ab = acc + (size_t)(1 + K + p) * G;
      for (int g = 0; g < G; ++g) ab[g] += U[g] * xp;
    }
  }
}

static double softplus(double x) { return x > 0 ? x + log1p(exp(-x)) : log1p(exp(x)); }
static double sigmoid(double x) { return 1.0 / (1.0 + exp(-x)); }

static double pass(cs_model* m, const float* eps, int mode) {
  const long N = m->N; const int G = m->G, C = m->C, K = m->K, P = m->P, D = m->D, T = m->T;
  const float *W = m->var[V_W], *v = m->var[V_v], *beta = m->var[V_beta], *au = m->var[V_alpha], *loc = m->var[V_loc], *ls = m->var[V_ls];
  double gene0 = 0, gene1 = 0, gene2 = 0;
  for (int g = 0; g < G; ++g) {
    const double e = eps[g], xx = (double)loc[g] + exp((double)ls[g]) * e, mm = softplus(xx), lm = log(mm);
    m->xs[g] = (float)xx; m->mu[g] = (float)mm;
    for (int c = 0; c < C; ++c) m->Mt[(size_t)c * G + g] = (float)mm * m->Lt[(size_t)c * G + g];
    gene0 += m->colsum[g] * lm; gene1 += -0.5 * lm * lm - 0.5 * LOG2PI; gene2 += -0.5 * e * e - ls[g] - 0.5 * LOG2PI + (mm - xx);
  }
  if (D > 0) for (int p = 0; p < P; ++p) for (int g = 0; g < G; ++g) gene0 += (double)beta[(size_t)p * G + g] * m->YtX[(long)g * P + p];
  double amx = -1e300, ase = 0;
  for (int c = 0; c < C; ++c) amx = fmax(amx, au[c]);
  for (int c = 0; c < C; ++c) ase += exp(au[c] - amx);
  float la[64];
  for (int c = 0; c < C; ++c) la[c] = (float)(au[c] - amx - log(ase));
  memset(m->tacc, 0, sizeof(float) * (size_t)T * (1 + D + K) * G);
  memset(m->tsum, 0, sizeof(double) * (size_t)T * (3 + C));
#pragma omp parallel num_threads(T)
  {
#ifdef _OPENMP
    const int t = omp_get_thread_num();
#else
    const int t = 0;
#endif
    cs_cell_range(m, t, N * t / T, N * (t + 1) / T, mode, la);
  }
  if (mode == 2) return 0.0;
  double cs[3] = {0, 0, 0}, sg[64];
  for (int c = 0; c < C; ++c) sg[c] = 0;
  for (int t = 0; t < T; ++t) { for (int j = 0; j < 3; ++j) cs[j] += m->tsum[t * (3 + C) + j]; for (int c = 0; c < C; ++c) sg[c] += m->tsum[t * (3 + C) + 3 + c]; }
  double EE = cs[0] + gene0, Ep = cs[1] + gene1, Eq = cs[2] + gene2;
  const double conc = 1.0 / C;
  Ep += -(C * lgamma(conc) - lgamma(1.0));
  double dla[64], dlas = 0, al[64];
  for (int c = 0; c < C; ++c) { al[c] = exp((double)la[c]); Ep += (conc - 1.0) * log(al[c] + 1e-3); dla[c] = sg[c] + (conc - 1.0) * al[c] / (al[c] + 1e-3); dlas += dla[c]; }
  for (int k = 0; k < K; ++k) {
    double w2 = 0; for (int g = 0; g < G; ++g) w2 += (double)W[(size_t)k * G + g] * W[(size_t)k * G + g];
    const double chi = exp((double)v[k]);
    Ep += -0.5 * chi * w2 + G * (0.5 * v[k] - 0.5 * LOG2PI) + (v[k] - chi);
    m->grad[V_v][k] = (float)(-0.5 * chi * w2 + 0.5 * G + 1.0 - chi);
  }
  m->terms[0] = EE; m->terms[1] = Ep; m->terms[2] = Eq;
  if (mode == 1) {
    const int WG = 1 + D + K;
    for (int c = 0; c < C; ++c) m->grad[V_alpha][c] = (float)(dla[c] - al[c] * dlas);
#pragma omp parallel for schedule(static)
    for (int g = 0; g < G; ++g) {
      double acc[40];
      for (int j = 0; j < WG; ++j) { acc[j] = 0; for (int t = 0; t < T; ++t) acc[j] += m->tacc[((size_t)t * WG + j) * G + g]; }
      const double mm = m->mu[g], xx = m->xs[g], e = eps[g];
      const double dmu = m->colsum[g] / mm + acc[0] - log(mm) / mm;
      const double sig = sigmoid(xx), dx = dmu * sig + (1.0 - sig);
      m->grad[V_loc][g] = (float)dx; m->grad[V_ls][g] = (float)(dx * e * exp((double)ls[g]) + 1.0);
      for (int k = 0; k < K; ++k)
        m->grad[V_W][(size_t)k * G + g] = (float)(acc[1 + D + k] + acc[1 + k] - exp((double)v[k]) * W[(size_t)k * G + g]);
      for (int p = 0; p < P; ++p) m->grad[V_beta][(size_t)p * G + g] = D > 0 ? (float)(m->YtX[(long)g * P + p] + acc[1 + K + p]) : 0.f;
    }
  }
  return EE + Ep - Eq;
}

double cs_elbo(cs_model* m, const float* eps, double* terms) {
  const double e = pass(m, eps, 0);
  if (terms) memcpy(terms, m->terms, sizeof(m->terms));
  return e;
}
void cs_gamma_init(cs_model* m, const float* eps) { pass(m, eps, 2); }
void cs_step(cs_model* m, const float* eps) {
  pass(m, eps, 1);
  const float lr_t = m->lr * sqrtf(1.f - m->b2p) / (1.f - m->b1p);
  for (int i = 0; i < 8; ++i) {
    float *mm = m->m[i], *vv = m->vv[i], *x = m->var[i]; const float* gr = m->grad[i];
#pragma omp parallel for simd schedule(static)
    for (long j = 0; j < m->len[i]; ++j) {
      const float g = -gr[j];
      mm[j] = 0.9f * mm[j] + 0.1f * g;
      vv[j] = 0.999f * vv[j] + (0.001f * g) * g;
      x[j] -= lr_t * mm[j] / (sqrtf(vv[j]) + 1e-8f);
    }
  }
  m->b1p *= 0.9f; m->b2p *= 0.999f;
}

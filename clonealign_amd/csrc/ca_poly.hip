// ca_poly.hip -- the cells x genes x clones reduction WITHOUT the cells x genes sweep, for a rank-one exponent (K + P = 1, one MC sample).
//
// What the reference computes (R/inference-tflow.R:278-292): E_ng = exp(psi_n W_g), Z_nc = sum_g E_ng mu_g L_gc -- N G C multiply-adds and
// N G exponentials per pass, and as many again on the way back (autodiff of :288-296).  With ONE latent dimension the exponent is a product
// x_n v_g of a per-cell and a per-gene number, and  Z_nc = sum_g M_gc exp(x_n v_g)  is, for every clone, ONE function of x evaluated at N
// points.  Cut the genes into bins by v (width delta, centres v_b) and expand each bin's share around its centre:
//     Z_nc = sum_b exp(x_n v_b) sum_k x_n^k B[b][k][c],     B[b][k][c] = sum_{g in bin b} M_gc (v_g - v_b)^k / k!
// -- a Taylor series of exp(x (v - v_b)) whose argument is bounded by |x|max delta / 2 <= 2 by the choice of delta, so twenty terms leave a
// remainder below 1e-11 relative, in float64, with no cancellation to speak of (e^2 against e^-2).  The moments B cost G R C operations, the
// evaluation N nb R C: the sweep's N G C is gone.  The way back has the same form.  With coef_nc = -gamma_nc s_n / Z_nc (the cell epilogue's,
// unchanged):
//     d ELBO / d mu_g = sum_c L_gc q_c(v_g),   d / d V_g = mu_g sum_c L_gc q_c'(v_g),   q_c(v) = sum_n coef_nc exp(x_n v)
//     q_c(v) = sum_k (v - v_b)^k / k!  Q[b][k][c]  for v in bin b,     Q[b][k][c] = sum_n coef_nc x_n^k exp(x_n v_b)        (all cells, every bin)
//     d / d F_n = sum_c coef_nc dZ_nc / dx                                                                                  (with the forward pass)
// Everything that is not this contraction -- the multinomial log-likelihood, softmax, ELBO partials, coef, d logits: the cell epilogue
// ca_cell_fused_group -- is the code the matrix-core sweep feeds, called with these Z values.  The matrix-core sweeps (ca_kernels.hip.h) remain the
// general path: two or more exponent dimensions, several MC samples, more than eight clones, or an exponent range this expansion would need more
// than CA_PL_NB bins for (the header word `bad`, surfaced as CA_ERR_STATE).  DESIGN.md section 5e.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ca_poly.h"

namespace {
#include "ca_kernels.hip.h"   // the cell epilogue and its helpers (internal linkage here: this unit instantiates only what it launches)

constexpr int R = CA_PL_R, NB = CA_PL_NB;
constexpr int TB_B = 256;       // k_poly_B block
constexpr int GPB = 128;        // genes per k_poly_B block

__device__ __forceinline__ float warp_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float warp_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- K1: bin geometry (every block makes the same one: min / max are exact in any order) and the forward moments ----------------------------------
// part[blk][b][k][col] block partials (the 1 / k! inside the powers), summed in block order by k_poly_red into tabB[b][k][col].
// col: draw A clones 0..7 | draw B clones 0..7.  (The blocks of one launch share nothing: the XCDs' L2s are not coherent with each other, and a
// device-scope fence per block costs more than the kernel boundary the reduction gets for free.)
__global__ void __launch_bounds__(TB_B) k_poly_B(const float* __restrict__ V, const float* __restrict__ F, const float* __restrict__ muA,
                                                 const float* __restrict__ muB, const float* __restrict__ Lb /*[G][8]*/, int G, int64_t N, int C,
                                                 ca_poly_hdr* __restrict__ hdr, double* __restrict__ part, unsigned int* __restrict__ bad_word) {
  __shared__ float smn[TB_B / 64], smx[TB_B / 64], sax[TB_B / 64];
  __shared__ double pw[GPB][R + 1];
  __shared__ double Mg[GPB][16];
  __shared__ int binof[GPB];
  __shared__ unsigned int present[(NB + 31) / 32];
  const int t = threadIdx.x;
  float mn = INFINITY, mx = -INFINITY, ax = 0.f;
  for (int g = t; g < G; g += TB_B) { const float v = V[g]; mn = fminf(mn, v); mx = fmaxf(mx, v); }
  for (int64_t n = t; n < N; n += TB_B) ax = fmaxf(ax, fabsf(F[n]));
  mn = warp_min(mn); mx = warp_max(mx); ax = warp_max(ax);
  if ((t & 63) == 0) { smn[t >> 6] = mn; smx[t >> 6] = mx; sax[t >> 6] = ax; }
  if (t < (NB + 31) / 32) present[t] = 0u;
  __syncthreads();
  mn = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
  mx = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
  ax = fmaxf(fmaxf(sax[0], sax[1]), fmaxf(sax[2], sax[3]));
  const double vlo = (double)mn, width = (double)mx - (double)mn, xmax = (double)ax;
  int nb = (int)ceil(xmax * width / (2.0 * CA_PL_A));
  nb = nb < 1 ? 1 : (nb > NB ? NB : nb);
  const double delta = width > 0.0 ? width / nb : 1.0;
  const int bad = !(xmax * delta * 0.5 <= CA_PL_A * 1.25) || !isfinite(xmax) || !isfinite(width);
  if (blockIdx.x == 0 && t == 0) {
    hdr->vlo = vlo; hdr->delta = delta; hdr->xmax = xmax; hdr->nb = nb;
    if (bad) { hdr->bad = 1; if (bad_word) __hip_atomic_store(bad_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }   // (the host looks at its next synchronisation)
  }
  // this block's genes: bin, powers of (v - v_b) over k!, the sixteen M columns
  const int g0 = blockIdx.x * GPB;
  if (t < GPB) {
    const int g = g0 + t;
    if (g < G) {
      const double v = (double)V[g];
      int b = (int)floor((v - vlo) / delta);
      b = b < 0 ? 0 : (b >= nb ? nb - 1 : b);
      binof[t] = b;
      atomicOr(&present[b >> 5], 1u << (b & 31));
      const double dv = v - (vlo + ((double)b + 0.5) * delta);
      double p = 1.0;
      for (int k = 0; k <= R; ++k) { pw[t][k] = p; p = p * dv / (double)(k + 1); }
      const double ma = (double)muA[g], mb = (double)muB[g];
      for (int c = 0; c < 8; ++c) {
        const double l = c < C ? (double)Lb[(int64_t)g * CA_CW + c] : 0.0;
        Mg[t][c] = ma * l; Mg[t][8 + c] = mb * l;
      }
    } else binof[t] = -1;
  }
  __syncthreads();
  const int ng = min(GPB, G - g0);
  constexpr int NO = (R + 1) * 16;
  double* mine = part + (int64_t)blockIdx.x * NB * NO;
  for (int b = 0; b < nb; ++b) {
    const bool any = (present[b >> 5] >> (b & 31)) & 1u;   // (uniform)
    for (int o = t; o < NO; o += TB_B) {
      double acc = 0.0;
      if (any) {
        const int k = o >> 4, col = o & 15;
        for (int i = 0; i < ng; ++i) acc += binof[i] == b ? pw[i][k] * Mg[i][col] : 0.0;
      }
      mine[(int64_t)b * NO + o] = acc;
    }
  }
}

// ---- K2: per cell: Z for both draws and dZ/dx for the train draw by Horner over the bins, the cell epilogue, d/dF, the backward moments --------------
template <int CP>
__global__ void __launch_bounds__(CA_TB) k_poly_cell(const ca_poly_hdr* __restrict__ hdr, const double* __restrict__ tabB, ca_cell_ptrs p,
                                                     const float* __restrict__ alpha_u, double* __restrict__ cell_part, int64_t N, int C, int K,
                                                     float* __restrict__ dF /*[N]*/, double* __restrict__ Qpart /*[grid][nb][R+2][C]*/) {
  constexpr int CPB = CA_TB / CP;             // cells per pass of the block
  constexpr int RQ = R + 2;                   // moments 0 .. R + 1 (the derivative of q needs one more)
  constexpr int NOI = (NB * RQ * 8 + CA_TB - 1) / CA_TB;
  __shared__ double sm[CA_TB];
  __shared__ double la[64];
  __shared__ double s_eb[CPB][NB];            // exp(x v_b)
  __shared__ double s_xp[CPB][RQ];            // x^k
  __shared__ double s_cf[CPB][8];             // coef
  ca_log_softmax_alpha(alpha_u, C, la);
  __syncthreads();
  const int t = threadIdx.x, c = t % CP, slot = t / CP;
  const int nb = hdr->nb;
  const double vlo = hdr->vlo, delta = hdr->delta;
  const int nout = nb * RQ * C;
  double qacc[NOI];
#pragma unroll
  for (int i = 0; i < NOI; ++i) qacc[i] = 0.0;
  ca_cell_acc acc = {0.0, 0.0, 0.0, 0.0, 0.0};
  const int64_t ngroups = (N + CPB - 1) / CPB;
  const int cc = c < C ? c : C - 1;
  for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int64_t n = grp * CPB + slot;
    const int64_t nn = n < N ? n : N - 1;
    const double x = (double)p.F[nn];
    double ZA = 0.0, ZB = 0.0, dZB = 0.0;
    for (int b = 0; b < nb; ++b) {
      const double vb = vlo + ((double)b + 0.5) * delta;
      const double e = exp(x * vb);
      const double* tb = tabB + ((int64_t)b * (R + 1)) * 16;
      double pa = tb[R * 16 + cc], pb = tb[R * 16 + 8 + cc], dpb = 0.0;
#pragma unroll 4
      for (int k = R - 1; k >= 0; --k) {
        dpb = dpb * x + pb;
        pa = pa * x + tb[k * 16 + cc];
        pb = pb * x + tb[k * 16 + 8 + cc];
      }
      ZA += e * pa; ZB += e * pb; dZB += e * (vb * pb + dpb);
      if (c == 0) s_eb[slot][b] = e;
    }
    {   // x^k, the lanes of a cell sharing the work
      double xk = 1.0;
      for (int k = 0; k < RQ; ++k) { if (k % CP == c) s_xp[slot][k] = xk; xk *= x; }
    }
    ca_cell_fused_group<CP>(p, la, n, N, C, 1, K, ZA, ZB, acc);
    // this lane's coef as the epilogue stored it (float: what the matrix-core way back reads as well); d/dF = sum_c coef dZ/dx
    const bool ok = n < N && c < C;
    const double cf = ok ? (double)p.coef[nn * CA_CW + cc] : 0.0;
    double df = cf * dZB;
#pragma unroll
    for (int o = CP / 2; o > 0; o >>= 1) df += __shfl_xor(df, o, CP);
    if (c == 0 && n < N) dF[n] = (float)df;
    if (c < 8) s_cf[slot][c] = cf;
    __syncthreads();
    // backward moments: output j = (b, k, clone) gathers over the cells of this pass, in cell order
#pragma unroll
    for (int i = 0; i < NOI; ++i) {
      const int j = t + i * CA_TB;
      if (j < nout) {
        const int cl = j % C, k = (j / C) % RQ, b = j / (C * RQ);
        double a = qacc[i];
        for (int s = 0; s < CPB; ++s) a += s_cf[s][cl] * s_eb[s][b] * s_xp[s][k];
        qacc[i] = a;
      }
    }
    __syncthreads();
  }
  ca_cell_fused_finish<CP>(acc, sm, cell_part, blockIdx.x, C);
  double* mine = Qpart + (int64_t)blockIdx.x * (NB * RQ * 8);
#pragma unroll
  for (int i = 0; i < NOI; ++i) {
    const int j = t + i * CA_TB;
    if (j < nout) mine[j] = qacc[i];
  }
}

// ---- fixed-order sums of the block partials: one wave per output, lanes stride over the blocks, then the wave's tree (same order every time) --------
// mode 0: plain sum (tabB); mode 1: sum / k! with k = (j / C) % (R + 2) (tabQ = Q_k / k!)
__global__ void __launch_bounds__(CA_TB) k_poly_red(const double* __restrict__ part, int nblk, int64_t stride, const ca_poly_hdr* __restrict__ hdr, int per_bin,
                                                    int mode, int C, double* __restrict__ out) {
  const int j = blockIdx.x * (CA_TB / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int nout = hdr->nb * per_bin;
  if (j >= nout) return;
  double a = 0.0;
  for (int blk = lane; blk < nblk; blk += 64) a += part[(int64_t)blk * stride + j];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
  if (lane == 0) {
    if (mode == 1) {
      const int k = (j / C) % (R + 2);
      double f = 1.0;
      for (int i = 2; i <= k; ++i) f *= (double)i;
      a /= f;
    }
    out[j] = a;
  }
}

// ---- K3: per gene: the two gradient sums from its bin's polynomial ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(CA_TB) k_poly_gene(const ca_poly_hdr* __restrict__ hdr, const double* __restrict__ tabQ, const float* __restrict__ V,
                                                     const float* __restrict__ mu, const float* __restrict__ Lb, int G, int C,
                                                     double* __restrict__ red_g /*[G][2]: d/dmu, d/dV*/) {
  constexpr int RQ = R + 2;
  const int g = blockIdx.x * CA_TB + threadIdx.x;
  if (g >= G) return;
  const int nb = hdr->nb;
  const double vlo = hdr->vlo, delta = hdr->delta;
  const double v = (double)V[g];
  int b = (int)floor((v - vlo) / delta);
  b = b < 0 ? 0 : (b >= nb ? nb - 1 : b);
  const double dv = v - (vlo + ((double)b + 0.5) * delta);
  const double* tq = tabQ + (int64_t)b * RQ * C;       // T_k = Q_k / k!
  double s0 = 0.0, s1 = 0.0;
  for (int c = 0; c < C; ++c) {
    // q = sum_{k <= R} dv^k T_k;  q' = sum_{k <= R} dv^k (k + 1) T_{k+1}
    double q = tq[R * C + c], dq = (double)(R + 1) * tq[(R + 1) * C + c];
    for (int k = R - 1; k >= 0; --k) {
      q = q * dv + tq[k * C + c];
      dq = dq * dv + (double)(k + 1) * tq[(k + 1) * C + c];
    }
    const double l = (double)Lb[(int64_t)g * CA_CW + c];
    s0 += l * q; s1 += l * dq;
  }
  red_g[(int64_t)g * 2 + 0] = s0;
  red_g[(int64_t)g * 2 + 1] = (double)mu[g] * s1;
}

inline int cdiv_i(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

}  // namespace

size_t ca_poly_workspace_bytes(int G, int n_cell_blocks) {
  const size_t nbg = (size_t)cdiv_i(G, GPB);
  return sizeof(ca_poly_hdr) + sizeof(double) * ((size_t)NB * (R + 1) * 16 * (nbg + 1) + (size_t)NB * (R + 2) * 8 * ((size_t)n_cell_blocks + 1));
}

void ca_poly_bind(ca_poly_ws* w, void* base, int G, int n_cell_blocks) {
  char* q = static_cast<char*>(base);
  w->hdr = reinterpret_cast<ca_poly_hdr*>(q); q += sizeof(ca_poly_hdr);
  const size_t nbg = (size_t)cdiv_i(G, GPB);
  w->tabB = reinterpret_cast<double*>(q); q += sizeof(double) * (size_t)NB * (R + 1) * 16;
  w->partB = reinterpret_cast<double*>(q); q += sizeof(double) * (size_t)NB * (R + 1) * 16 * nbg;
  w->tabQ = reinterpret_cast<double*>(q); q += sizeof(double) * (size_t)NB * (R + 2) * 8;
  w->Qpart = reinterpret_cast<double*>(q);
  w->n_cell_blocks = n_cell_blocks; w->n_gene_blocks = (int)nbg;
}

hipError_t ca_poly_forward(hipStream_t st, const ca_poly_ws* w, const float* V, const float* F, const float* muA, const float* muB, const float* Lb, int G,
                           int64_t N, int C, int K, const void* cell_ptrs, const float* alpha_u, double* cell_part, float* dF, unsigned int* bad_word) {
  const ca_cell_ptrs& p = *static_cast<const ca_cell_ptrs*>(cell_ptrs);
  hipLaunchKernelGGL(k_poly_B, dim3(w->n_gene_blocks), dim3(TB_B), 0, st, V, F, muA, muB, Lb, G, N, C, w->hdr, w->partB, bad_word);
  hipLaunchKernelGGL(k_poly_red, dim3(cdiv_i((int64_t)NB * (R + 1) * 16, CA_TB / 64)), dim3(CA_TB), 0, st, w->partB, w->n_gene_blocks, (int64_t)NB * (R + 1) * 16, w->hdr,
                     (R + 1) * 16, 0, C, w->tabB);
  int CP = 1;
  while (CP < C) CP <<= 1;
  const dim3 grid(w->n_cell_blocks);
#define CA_PCELL(CPV) hipLaunchKernelGGL((k_poly_cell<CPV>), grid, dim3(CA_TB), 0, st, w->hdr, w->tabB, p, alpha_u, cell_part, N, C, K, dF, w->Qpart)
  if (CP == 4) CA_PCELL(4); else CA_PCELL(8);   // (3 .. 8 clones: ca_poly_ok)
#undef CA_PCELL
  hipLaunchKernelGGL(k_poly_red, dim3(cdiv_i((int64_t)NB * (R + 2) * C, CA_TB / 64)), dim3(CA_TB), 0, st, w->Qpart, w->n_cell_blocks, (int64_t)NB * (R + 2) * 8, w->hdr,
                     (R + 2) * C, 1, C, w->tabQ);
  return hipGetLastError();
}

hipError_t ca_poly_backward(hipStream_t st, const ca_poly_ws* w, const float* V, const float* mu, const float* Lb, int G, int C, double* red_g) {
  hipLaunchKernelGGL(k_poly_gene, dim3(cdiv_i(G, CA_TB)), dim3(CA_TB), 0, st, w->hdr, w->tabQ, V, mu, Lb, G, C, red_g);
  return hipGetLastError();
}

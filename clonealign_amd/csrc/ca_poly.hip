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

#include <algorithm>
#include <cstring>

#include "ca_poly.h"

namespace ca_series {   // (a namespace of its own: the header's kernels get names that do not clash with the engine unit's, and profiles read "ca_series::k_poly_cell")
#include "ca_kernels.hip.h"   // the cell epilogue and its helpers (this unit instantiates only what it launches)

constexpr int R = CA_PL_R, NB = CA_PL_NB;
constexpr int TB_B = 384;       // k_poly_B block: one thread per (k, column) output (21 x 16 = 336)
constexpr int GPB = 32;         // genes per k_poly_B block

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

// the ranges of one parameter state, for the host's look ahead (ca_poly_guard in the engine): values first, the sequence number last
__device__ __forceinline__ void ca_poly_mirror_store(double* m, double seq, double xmax, double vlo, double vhi) {
  __hip_atomic_store(m + 1, xmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(m + 2, vlo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(m + 3, vhi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __threadfence_system();
  __hip_atomic_store(m, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- K0: max |x| over the cells (an order-preserving unsigned maximum: exact whatever the order); k_poly_red resets the word behind its reader ------------
__global__ void __launch_bounds__(CA_TB) k_poly_xmax(const float* __restrict__ F, int64_t N, unsigned int* __restrict__ xbits) {
  __shared__ float sx[CA_TB / 64];
  float ax = 0.f;
  const int64_t n4 = N / 4;
  const float4* F4 = reinterpret_cast<const float4*>(F);
  for (int64_t i = (int64_t)blockIdx.x * CA_TB + threadIdx.x; i < n4; i += (int64_t)gridDim.x * CA_TB) {
    const float4 v = F4[i];
    ax = fmaxf(fmaxf(ax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(N - n4 * 4)) ax = fmaxf(ax, fabsf(F[n4 * 4 + threadIdx.x]));
  ax = warp_max(ax);
  if ((threadIdx.x & 63) == 0) sx[threadIdx.x >> 6] = ax;
  __syncthreads();
  if (threadIdx.x == 0) {
    ax = fmaxf(fmaxf(sx[0], sx[1]), fmaxf(sx[2], sx[3]));
    if (!(ax == ax)) ax = INFINITY;                      // (a NaN latent position: the header's `bad` says so)
    atomicMax(xbits, __float_as_uint(ax));               // non-negative floats order like their bit patterns
  }
}

// ---- ranges only: one block scans V, takes the cells' maximum from k_poly_xmax's word (and resets it), mirrors both to the host -------------------------
// (xglob / nglob / xadd: a cell-sharded fit -- max |x| over ALL ranks' cells as the last collective left it, one slot per rank, for the state one Adam step back,
//  plus what that step can have added: the same number on every rank, so that every rank takes the same decisions; see ca_poly_xslot)
__global__ void __launch_bounds__(CA_TB) k_poly_ranges(const float* __restrict__ V, int G, unsigned int* __restrict__ xbits, double* __restrict__ mirror, double seq,
                                                       const float* __restrict__ xpart, int nx, const double* __restrict__ xglob, int nglob, double xadd) {
  __shared__ float smn[CA_TB / 64], smx[CA_TB / 64], sxm[CA_TB / 64];
  const int t = threadIdx.x;
  float mn = INFINITY, mx = -INFINITY;
  float xm = 0.f;   // (nx > 0: max |x| per piece of cells as the merged update left it)
  for (int i = t; i < nx; i += CA_TB) xm = fmaxf(xm, xpart[i]);
  xm = warp_max(xm);
  if ((t & 63) == 0) sxm[t >> 6] = xm;
  constexpr int U = 8;
  for (int g0_ = 0; g0_ < G; g0_ += CA_TB * U) {
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const int g = g0_ + u * CA_TB + t; v[u] = V[g < G ? g : G - 1]; }
#pragma unroll
    for (int u = 0; u < U; ++u) { mn = fminf(mn, v[u]); mx = fmaxf(mx, v[u]); }
  }
  mn = warp_min(mn); mx = warp_max(mx);
  if ((t & 63) == 0) { smn[t >> 6] = mn; smx[t >> 6] = mx; }
  __syncthreads();
  if (t == 0) {
    mn = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
    mx = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
    double xmax = nx > 0 ? (double)fmaxf(fmaxf(sxm[0], sxm[1]), fmaxf(sxm[2], sxm[3])) : (double)__uint_as_float(*xbits);
    if (nglob > 0) { xmax = 0.0; for (int r = 0; r < nglob; ++r) xmax = fmax(xmax, xglob[r]); xmax += xadd; }
    *xbits = 0u;
    ca_poly_mirror_store(mirror, seq, xmax, (double)mn, (double)mx);
  }
}

// ---- K1: bin geometry (every block makes the same one: min / max are exact in any order) and the forward moments ----------------------------------
// part[blk][b][k][col] block partials (the 1 / k! inside the powers), summed in block order by k_poly_red into tabB[b][k][col].
// col: draw A clones 0..7 | draw B clones 0..7.  (The blocks of one launch share nothing: the XCDs' L2s are not coherent with each other, and a
// device-scope fence per block costs more than the kernel boundary the reduction gets for free.)
__global__ void __launch_bounds__(TB_B) k_poly_B(const float* __restrict__ V, const unsigned int* __restrict__ xbits, const float* __restrict__ muA,
                                                 const float* __restrict__ muB, const float* __restrict__ Lb /*[G][8]*/, int G, int C,
                                                 ca_poly_hdr* __restrict__ hdr, double* __restrict__ part, unsigned int* __restrict__ bad_word,
                                                 double* __restrict__ mirror /* mapped host ring slot {seq, xmax, vlo, vhi} or null */, double seq,
                                                 const float* __restrict__ xpart /* nx > 0: max |x| per piece of cells (the merged update's), instead of *xbits */, int nx,
                                                 const double* __restrict__ xglob, int nglob, double xadd) {
  __shared__ float smn[TB_B / 64], smx[TB_B / 64], sxm[TB_B / 64];
  __shared__ double pw[GPB][R + 1];
  __shared__ double Mg[GPB][16];
  __shared__ int binof[GPB];
  __shared__ unsigned int present[(NB + 31) / 32];
  const int t = threadIdx.x;
  float mn = INFINITY, mx = -INFINITY;
  {   // (eight loads in flight: a load per iteration waited for the one before -- 0.5 us each)
    constexpr int U = 8;
    for (int g0_ = 0; g0_ < G; g0_ += TB_B * U) {
      float v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { const int g = g0_ + u * TB_B + t; v[u] = V[g < G ? g : G - 1]; }
#pragma unroll
      for (int u = 0; u < U; ++u) { mn = fminf(mn, v[u]); mx = fmaxf(mx, v[u]); }
    }
  }
  float xm = 0.f;
  {
    constexpr int U = 4;
    for (int i0 = 0; i0 < nx; i0 += TB_B * U) {
      float v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { const int i = i0 + u * TB_B + t; v[u] = xpart[i < nx ? i : nx - 1]; }
#pragma unroll
      for (int u = 0; u < U; ++u) xm = fmaxf(xm, v[u]);
    }
  }
  mn = warp_min(mn); mx = warp_max(mx); xm = warp_max(xm);
  if ((t & 63) == 0) { smn[t >> 6] = mn; smx[t >> 6] = mx; sxm[t >> 6] = xm; }
  if (t < (NB + 31) / 32) present[t] = 0u;
  __syncthreads();
  mn = smn[0]; mx = smx[0]; xm = sxm[0];
#pragma unroll
  for (int w_ = 1; w_ < TB_B / 64; ++w_) { mn = fminf(mn, smn[w_]); mx = fmaxf(mx, smx[w_]); xm = fmaxf(xm, sxm[w_]); }
  double xmax = nx > 0 ? (double)xm : (double)__uint_as_float(*xbits);
  if (nglob > 0) { xmax = 0.0; for (int r = 0; r < nglob; ++r) xmax = fmax(xmax, xglob[r]); xmax += xadd; }   // (uniform; a handful of ranks)
  const double vlo = (double)mn, width = (double)mx - (double)mn;
  int nb = (int)ceil(xmax * width / (2.0 * CA_PL_A));
  nb = nb < 1 ? 1 : (nb > NB ? NB : nb);
  const double delta = width > 0.0 ? width / nb : 1.0;
  // (all loadings equal -- W = 0 at the start of every fit -- is one bin of width zero: any |x| is covered)
  const int bad = !(xmax * (width > 0.0 ? delta : 0.0) * 0.5 <= CA_PL_A * 1.25) || !isfinite(xmax) || !isfinite(width);
  if (blockIdx.x == 0 && t == 0) {
    hdr->vlo = vlo; hdr->delta = delta; hdr->xmax = xmax; hdr->nb = nb;
    if (bad) { hdr->bad = 1; if (bad_word) __hip_atomic_store(bad_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }   // (the host looks at its next synchronisation)
    if (mirror) ca_poly_mirror_store(mirror, seq, xmax, vlo, (double)mx);
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
#pragma unroll
      for (int k = 0; k <= R; ++k) { pw[t][k] = p; p = p * dv * (1.0 / (double)(k + 1)); }   // (the reciprocals are compile-time constants)
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
  if (t < NO) {
    const int k = t >> 4, col = t & 15;
    // ONE pass over the block's genes for the first four bins (the usual case is one to three): four accumulators, genes in order
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (nb == 1) {   // (uniform: one bin, no selects)
#pragma unroll 8
      for (int i = 0; i < ng; ++i) a0 += pw[i][k] * Mg[i][col];
    } else {
#pragma unroll 4
      for (int i = 0; i < ng; ++i) {
        const double pr = pw[i][k] * Mg[i][col];
        const int bi = binof[i];
        a0 += bi == 0 ? pr : 0.0; a1 += bi == 1 ? pr : 0.0; a2 += bi == 2 ? pr : 0.0; a3 += bi == 3 ? pr : 0.0;
      }
    }
    mine[0 * NO + t] = a0;
    if (nb > 1) mine[1 * NO + t] = a1;
    if (nb > 2) mine[2 * NO + t] = a2;
    if (nb > 3) mine[3 * NO + t] = a3;
    for (int b = 4; b < nb; ++b) {          // (a wide exponent range)
      double acc = 0.0;
      if ((present[b >> 5] >> (b & 31)) & 1u)
        for (int i = 0; i < ng; ++i) acc += binof[i] == b ? pw[i][k] * Mg[i][col] : 0.0;
      mine[(int64_t)b * NO + t] = acc;
    }
  }
}

// ---- K2: per cell: Z for both draws and dZ/dx for the train draw by Horner over the bins, the cell epilogue, d/dF, the backward moments --------------
template <int CP>
__global__ void __launch_bounds__(CA_TB) k_poly_cell(const ca_poly_hdr* __restrict__ hdr, const double* __restrict__ tabB, ca_cell_ptrs p,
                                                     const float* __restrict__ alpha_u, double* __restrict__ cell_part, int64_t N, int C, int K,
                                                     float* __restrict__ dF /*[N]*/, double* __restrict__ Qpart /*[grid][nb][R+2][C]*/, int ncb, ca_yfin_args yfin) {
  // Blocks behind the ncb cell blocks finish the count-matrix stream's two products (column sums, row sums and psi.(YW) partials: what k_yfinish does as a
  // launch of its own, 5.4 us and a launch gap on the iteration's critical path): nothing in this launch reads them, the per-gene launch that follows does.
  // The cell blocks are two to a CU and live as long as the launch; these few hundred short blocks take the free slots beside them.
  if ((int)blockIdx.x >= ncb) {
    const int e = (int)blockIdx.x - ncb;
    const int ncolblk = (yfin.ncol + CA_TB / 64 - 1) / (CA_TB / 64);
    if (e < ncolblk) {
      const int job = e * (CA_TB / 64) + (int)(threadIdx.x >> 6);
      if (job < yfin.ncol) ca_yfin_col_wave(yfin, job);
    } else if (e - ncolblk < yfin.nrow) {
      __shared__ double ca_yfin_sm[CA_TB / 64];
      ca_yfin_row_block(yfin, e - ncolblk, ca_yfin_sm);
    }
    return;
  }
  constexpr int CPB = CA_TB / CP;             // cells per pass of the block
  constexpr int RQ = R + 2;                   // moments 0 .. R + 1 (the derivative of q needs one more)
  constexpr int NBR = 4;                      // bins whose moments a thread keeps in registers (the usual case: one to three bins); the rest go through its slab
  __shared__ double sm[CA_TB];
  __shared__ double la[64];
  __shared__ double s_eb[CPB][NB];            // exp(x v_b)
  __shared__ double s_xp[CPB][RQ];            // x^k
  __shared__ double s_cf[CPB][8];             // coef
  // the coefficient tables of the first NBL bins (the usual case has one to three) in LDS: every pass of the block reads them again, 42 loads per lane and bin,
  // and from global memory those loads -- not the arithmetic -- were what a pass took
  constexpr int NBL = CA_PL_NBL;
  __shared__ double s_tb[NBL][(R + 1) * 16];
  const int nb = hdr->nb;
  ca_log_softmax_alpha(alpha_u, C, la);
  for (int i = threadIdx.x; i < CPB * NB; i += CA_TB) (&s_eb[0][0])[i] = 0.0;   // (bins past nb: zeros, so that the gather needs no bounds)
  for (int i = threadIdx.x; i < (nb < NBL ? nb : NBL) * (R + 1) * 16; i += CA_TB) (&s_tb[0][0])[i] = tabB[i];
  __syncthreads();
  const int t = threadIdx.x, c = t % CP, slot = t / CP;
  const double vlo = hdr->vlo, delta = hdr->delta;
  // backward moments: thread t < (R + 2) C owns (k, clone) = (t / C, t % C) of EVERY bin -- nobody else adds to its outputs, cells are added in cell order
  const bool qown = t < RQ * C;
  const int qk = qown ? t / C : 0, qc = qown ? t % C : 0;
  double qa[NBR] = {0.0, 0.0, 0.0, 0.0};
  double* mine = Qpart + (int64_t)blockIdx.x * (NB * RQ * 8);
  if (qown) for (int b = NBR; b < nb; ++b) mine[(b * RQ + qk) * C + qc] = 0.0;
  ca_cell_acc acc = {0.0, 0.0, 0.0, 0.0, 0.0};
  const int64_t ngroups = (N + CPB - 1) / CPB;
  const int cc = c < C ? c : C - 1;
  // what the epilogue reads for this lane's (cell, clone) that nothing here produces: loaded a pass AHEAD, beside the arithmetic of the current one
  auto load_pass = [&](int64_t grp, double& x_, ca_cell_pre& pre_) {
    const int64_t n_ = grp * CPB + slot, nn_ = n_ < N ? n_ : N - 1;
    x_ = (double)p.F[nn_];
    pre_.gl = p.glogit[nn_ * C + cc]; pre_.sn = p.s64[nn_]; pre_.Anc = p.A[nn_ * C + cc];
  };
  double x_next = 0.0; ca_cell_pre pre_next = {0.f, 0.0, 0.0};
  if ((int64_t)blockIdx.x < ngroups) load_pass(blockIdx.x, x_next, pre_next);
  for (int64_t grp = blockIdx.x; grp < ngroups; grp += ncb) {
    const int64_t n = grp * CPB + slot;
    const double x = x_next; const ca_cell_pre pre = pre_next;
    if (grp + ncb < ngroups) load_pass(grp + ncb, x_next, pre_next);
    double ZA = 0.0, ZB = 0.0, dZB = 0.0;
    for (int b = 0; b < nb; ++b) {
      const double vb = vlo + ((double)b + 0.5) * delta;
      const double e = exp(x * vb);
      double pa, pb, dpb = 0.0;
#define CA_PL_HORNER(TB)                                                  \
      pa = (TB)[R * 16 + cc]; pb = (TB)[R * 16 + 8 + cc];                 \
      _Pragma("unroll 10")                                                \
      for (int k = R - 1; k >= 0; --k) {                                  \
        dpb = dpb * x + pb;                                               \
        pa = pa * x + (TB)[k * 16 + cc];                                  \
        pb = pb * x + (TB)[k * 16 + 8 + cc];                              \
      }
      if (b < NBL) { CA_PL_HORNER(s_tb[b]) }                              // (LDS)
      else { const double* tb = tabB + ((int64_t)b * (R + 1)) * 16; CA_PL_HORNER(tb) }
#undef CA_PL_HORNER
      ZA += e * pa; ZB += e * pb; dZB += e * (vb * pb + dpb);
      if (c == 0) s_eb[slot][b] = e;
    }
    {   // x^k, the lanes of a cell sharing the stores
      double xk = 1.0;
      for (int k = 0; k < RQ; ++k) { if (k % CP == c) s_xp[slot][k] = xk; xk *= x; }
    }
    float cff = 0.f;
    ca_cell_fused_group<CP>(p, la, n, N, C, 1, K, ZA, ZB, acc, &pre, &cff);
    // this lane's coef as the epilogue stored it (float: what the matrix-core way back reads as well); d/dF = sum_c coef dZ/dx
    const double cf = (double)cff;
    double df = cf * dZB;
#pragma unroll
    for (int o = CP / 2; o > 0; o >>= 1) df += __shfl_xor(df, o, CP);
    if (c == 0 && n < N) dF[n] = (float)df;
    if (c < 8) s_cf[slot][c] = cf;
    __syncthreads();
    if (qown) {
      // (loads batched eight cells at a time: a load per step waited for the one before, and that chain WAS this kernel; bins past nb hold zeros)
      if (nb == 1) {
#pragma unroll 8
        for (int s = 0; s < CPB; ++s) qa[0] += s_cf[s][qc] * s_xp[s][qk] * s_eb[s][0];
      } else {
#pragma unroll 8
        for (int s = 0; s < CPB; ++s) {
          const double w = s_cf[s][qc] * s_xp[s][qk];
          qa[0] += w * s_eb[s][0]; qa[1] += w * s_eb[s][1]; qa[2] += w * s_eb[s][2]; qa[3] += w * s_eb[s][3];
        }
      }
      for (int b = NBR; b < nb; ++b) {        // (a wide exponent range: the thread's own words of its block's slab, cell order all the same)
        double a = mine[(b * RQ + qk) * C + qc];
        for (int s = 0; s < CPB; ++s) a += s_cf[s][qc] * s_xp[s][qk] * s_eb[s][b];
        mine[(b * RQ + qk) * C + qc] = a;
      }
    }
    __syncthreads();
  }
  ca_cell_fused_finish<CP>(acc, sm, cell_part, blockIdx.x, C);
  if (qown) {
#pragma unroll
    for (int b = 0; b < NBR; ++b) if (b < nb) mine[(b * RQ + qk) * C + qc] = qa[b];
  }
}

// ---- fixed-order sums of the block partials: one wave per output, lanes stride over the blocks, then the wave's tree (same order every time) --------
// mode 0: plain sum (tabB); mode 1: sum / k! with k = (j / C) % (R + 2) (tabQ = Q_k / k!)
// this rank's max |x| into its slot, zeros into the other ranks' (the body of k_poly_xslot; also rides on the reduction launch of the backward moments)
struct ca_xslot_args { const float* xpart; int nx; const float* F; int64_t N; double* slots; int rank, world; };
__device__ __forceinline__ void ca_poly_xslot_body(const ca_xslot_args& a) {
  __shared__ float sx[CA_TB / 64];
  float ax = 0.f;
  if (a.nx > 0) { for (int i = threadIdx.x; i < a.nx; i += CA_TB) ax = fmaxf(ax, a.xpart[i]); }
  else { for (int64_t i = threadIdx.x; i < a.N; i += CA_TB) { const float v = fabsf(a.F[i]); ax = fmaxf(ax, v == v ? v : INFINITY); } }
  ax = warp_max(ax);
  if ((threadIdx.x & 63) == 0) sx[threadIdx.x >> 6] = ax;
  __syncthreads();
  if ((int)threadIdx.x < a.world) a.slots[threadIdx.x] = (int)threadIdx.x == a.rank ? (double)fmaxf(fmaxf(sx[0], sx[1]), fmaxf(sx[2], sx[3])) : 0.0;
}
// (nred: the blocks that reduce; a cell-sharded fit adds up to two more behind them -- the pending monitor pass's local block sums (a ca_small_args with
//  reduce_only) and this rank's max |x| slot -- what would otherwise be two small launches in front of the iteration's collective)
__global__ void __launch_bounds__(CA_TB) k_poly_red(const double* __restrict__ part, int nblk, int64_t stride, const ca_poly_hdr* __restrict__ hdr, int per_bin,
                                                    int mode, int C, double* __restrict__ out, unsigned int* __restrict__ xbits, int nred, ca_small_args tail,
                                                    ca_xslot_args xs) {
  if ((int)blockIdx.x >= nred) {
    const int e = (int)blockIdx.x - nred;
    if (e == 0 && tail.enabled) ca_final_small_body(tail);
    else if (xs.slots) ca_poly_xslot_body(xs);
    return;
  }
  if (xbits && blockIdx.x == 0 && threadIdx.x == 0) *xbits = 0u;   // (its reader, k_poly_B, is complete: ready for the next state's maximum)
  const int lane = threadIdx.x & 63, nwave = nred * (CA_TB / 64);
  const int nout = hdr->nb * per_bin;
  for (int j = blockIdx.x * (CA_TB / 64) + (threadIdx.x >> 6); j < nout; j += nwave) {
    // (up to eight loads in flight per lane: a miss to another XCD's data costs a microsecond, a chain of them is the kernel)
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int blk = lane + 64 * u; v[u] = blk < nblk ? part[(int64_t)blk * stride + j] : 0.0; }
    double a = 0.0;
#pragma unroll
    for (int u = 0; u < 8; ++u) a += v[u];
    for (int blk = lane + 512; blk < nblk; blk += 64) a += part[(int64_t)blk * stride + j];
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
}

// ---- K3: per gene: the two gradient sums from its bin's polynomial ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(CA_TB) k_poly_gene(const ca_poly_hdr* __restrict__ hdr, const double* __restrict__ tabQ, const float* __restrict__ V,
                                                     const float* __restrict__ mu, const float* __restrict__ Lb, int G, int C,
                                                     double* __restrict__ red_g /*[G][2]: d/dmu, d/dV*/, ca_small_args tail, int ngblk) {
  constexpr int RQ = R + 2;
  if ((int)blockIdx.x >= ngblk) {   // a pending monitor pass's tail (cell partials -> red, psi.(YW) sum, ELBO assembly): the extra block, as on the matrix-core way back
    if (tail.enabled) ca_final_small_body(tail);
    return;
  }
  // The moment tables of the first NBL bins (the usual case has one to three) in LDS: a gene reads 176 doubles of its bin's table, and from global memory that
  // was a chain of 320 dependent-latency loads per thread -- the 11.9 us this launch took.  One load per step: T_{k+1} of a step is T_k of the one before.
  constexpr int NBL = CA_PL_NBL;
  __shared__ double s_tq[NBL][RQ * 8];
  const int nb = hdr->nb;
  const double vlo = hdr->vlo, delta = hdr->delta;
  for (int i = threadIdx.x; i < (nb < NBL ? nb : NBL) * RQ * C; i += CA_TB) s_tq[i / (RQ * C)][i % (RQ * C)] = tabQ[i];
  __syncthreads();
  const int g = blockIdx.x * CA_TB + threadIdx.x;
  if (g >= G) return;
  const double v = (double)V[g];
  int b = (int)floor((v - vlo) / delta);
  b = b < 0 ? 0 : (b >= nb ? nb - 1 : b);
  const double dv = v - (vlo + ((double)b + 0.5) * delta);
  float lrow[8];
  {
    const float4 l0 = *reinterpret_cast<const float4*>(Lb + (int64_t)g * CA_CW), l1 = *reinterpret_cast<const float4*>(Lb + (int64_t)g * CA_CW + 4);
    lrow[0] = l0.x; lrow[1] = l0.y; lrow[2] = l0.z; lrow[3] = l0.w; lrow[4] = l1.x; lrow[5] = l1.y; lrow[6] = l1.z; lrow[7] = l1.w;
  }
  double s0 = 0.0, s1 = 0.0;
  // q = sum_{k <= R} dv^k T_k;  q' = sum_{k <= R} dv^k (k + 1) T_{k+1}   (T_k = Q_k / k!)
#define CA_PL_GENE(TQ)                                                            \
  _Pragma("unroll")                                                               \
  for (int c = 0; c < 8; ++c) {                                                   \
    if (c < C) {                                                                  \
      double tn = (TQ)[(R + 1) * C + c], tk = (TQ)[R * C + c];                    \
      double q = tk, dq = (double)(R + 1) * tn;                                   \
      _Pragma("unroll 10")                                                        \
      for (int k = R - 1; k >= 0; --k) {                                          \
        tn = tk; tk = (TQ)[k * C + c];                                            \
        q = q * dv + tk;                                                          \
        dq = dq * dv + (double)(k + 1) * tn;                                      \
      }                                                                           \
      const double l = (double)lrow[c];                                           \
      s0 += l * q; s1 += l * dq;                                                  \
    }                                                                             \
  }
  if (nb <= NBL) { CA_PL_GENE(s_tq[b]) }   // (uniform)
  else { const double* tq = tabQ + (int64_t)b * RQ * C; CA_PL_GENE(tq) }
#undef CA_PL_GENE
  red_g[(int64_t)g * 2 + 0] = s0;
  red_g[(int64_t)g * 2 + 1] = (double)mu[g] * s1;
}

inline int cdiv_i(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

}  // namespace ca_series
using namespace ca_series;

size_t ca_poly_workspace_bytes(int G, int n_cell_blocks) {
  const size_t nbg = (size_t)cdiv_i(G, GPB);
  return sizeof(ca_poly_hdr) + 64 + sizeof(double) * ((size_t)NB * (R + 1) * 16 * (nbg + 1) + (size_t)NB * (R + 2) * 8 * ((size_t)n_cell_blocks + 1));
}

void ca_poly_bind(ca_poly_ws* w, void* base, int G, int n_cell_blocks) {
  char* q = static_cast<char*>(base);
  w->hdr = reinterpret_cast<ca_poly_hdr*>(q); q += sizeof(ca_poly_hdr);
  w->xbits = reinterpret_cast<unsigned int*>(q); q += 64;
  const size_t nbg = (size_t)cdiv_i(G, GPB);
  w->tabB = reinterpret_cast<double*>(q); q += sizeof(double) * (size_t)NB * (R + 1) * 16;
  w->partB = reinterpret_cast<double*>(q); q += sizeof(double) * (size_t)NB * (R + 1) * 16 * nbg;
  w->tabQ = reinterpret_cast<double*>(q); q += sizeof(double) * (size_t)NB * (R + 2) * 8;
  w->Qpart = reinterpret_cast<double*>(q);
  w->n_cell_blocks = n_cell_blocks; w->n_gene_blocks = (int)nbg;
}

// a cell-sharded fit: this rank's max |x| of the CURRENT state into its slot of `slots[world]`, zeros into the others -- the sum over the ranks (the fit's one collective
// per iteration carries the slots) then holds every rank's maximum.  One block; from the merged update's per-piece maxima where they are current, else from F.
__global__ void __launch_bounds__(CA_TB) k_poly_xslot(const float* __restrict__ xpart, int nx, const float* __restrict__ F, int64_t N, double* __restrict__ slots, int rank,
                                                      int world) {
  const ca_xslot_args a = {xpart, nx, F, N, slots, rank, world};
  ca_poly_xslot_body(a);
}
hipError_t ca_poly_xslot(hipStream_t st, const float* xpart, int nx, const float* F, int64_t N, double* slots, int rank, int world) {
  hipLaunchKernelGGL(k_poly_xslot, dim3(1), dim3(CA_TB), 0, st, xpart, xpart ? nx : 0, F, N, slots, rank, world);
  return hipGetLastError();
}

hipError_t ca_poly_ranges(hipStream_t st, const ca_poly_ws* w, const float* V, const float* F, int G, int64_t N, double* mirror, double seq, const float* xpart, int nx,
                          const double* xglob, int nglob, double xadd) {
  if (!xpart) nx = 0;
  if (!xglob) nglob = 0;
  if (nx == 0 && nglob == 0) hipLaunchKernelGGL(k_poly_xmax, dim3((unsigned)std::min<int64_t>(128, (N + 4 * CA_TB - 1) / (4 * CA_TB))), dim3(CA_TB), 0, st, F, N, w->xbits);
  hipLaunchKernelGGL(k_poly_ranges, dim3(1), dim3(CA_TB), 0, st, V, G, w->xbits, mirror, seq, xpart, nglob ? 0 : nx, xglob, nglob, xadd);
  return hipGetLastError();
}

hipError_t ca_poly_moments(hipStream_t st, const ca_poly_ws* w, const float* V, const float* F, const float* muA, const float* muB, const float* Lb, int G,
                           int64_t N, int C, unsigned int* bad_word, double* mirror, double seq, const float* xpart, int nx, const double* xglob, int nglob, double xadd) {
  if (!xpart) nx = 0;
  if (!xglob) nglob = 0;
  if (nx == 0 && nglob == 0) hipLaunchKernelGGL(k_poly_xmax, dim3((unsigned)std::min<int64_t>(128, (N + 4 * CA_TB - 1) / (4 * CA_TB))), dim3(CA_TB), 0, st, F, N, w->xbits);
  hipLaunchKernelGGL(k_poly_B, dim3(w->n_gene_blocks), dim3(TB_B), 0, st, V, w->xbits, muA, muB, Lb, G, C, w->hdr, w->partB, bad_word, mirror, seq, xpart, nglob ? 0 : nx,
                     xglob, nglob, xadd);
  {
    ca_small_args no_tail; memset(&no_tail, 0, sizeof(no_tail));
    ca_xslot_args no_xs; memset(&no_xs, 0, sizeof(no_xs));
    hipLaunchKernelGGL(k_poly_red, dim3(256), dim3(CA_TB), 0, st, w->partB, w->n_gene_blocks, (int64_t)NB * (R + 1) * 16, w->hdr,
                       (R + 1) * 16, 0, C, w->tabB, w->xbits, 256, no_tail, no_xs);
  }
  return hipGetLastError();
}

hipError_t ca_poly_cells(hipStream_t st, const ca_poly_ws* w, int64_t N, int C, int K, const void* cell_ptrs, const float* alpha_u, double* cell_part, float* dF,
                         const void* yfin_args, const void* local_tail, const float* xs_part, int xs_n, const float* xs_F, double* xs_slots, int rank, int world) {
  const ca_cell_ptrs& p = *static_cast<const ca_cell_ptrs*>(cell_ptrs);
  ca_yfin_args yfin;
  if (yfin_args) memcpy(&yfin, yfin_args, sizeof(yfin)); else memset(&yfin, 0, sizeof(yfin));
  int CP = 1;
  while (CP < C) CP <<= 1;
  const int nextra = yfin_args ? cdiv_i(yfin.ncol, CA_TB / 64) + yfin.nrow : 0;
  const dim3 grid(w->n_cell_blocks + nextra);
#define CA_PCELL(CPV) hipLaunchKernelGGL((k_poly_cell<CPV>), grid, dim3(CA_TB), 0, st, w->hdr, w->tabB, p, alpha_u, cell_part, N, C, K, dF, w->Qpart, \
                                         w->n_cell_blocks, yfin)
  if (CP == 4) CA_PCELL(4); else CA_PCELL(8);   // (3 .. 8 clones: ca_poly_ok)
#undef CA_PCELL
  {
    ca_small_args tail;
    if (local_tail) memcpy(&tail, local_tail, sizeof(tail)); else memset(&tail, 0, sizeof(tail));
    const ca_xslot_args xs = {xs_part, xs_part ? xs_n : 0, xs_F, N, xs_slots, rank, world};
    const int nextra = (tail.enabled || xs_slots) ? 2 : 0;
    hipLaunchKernelGGL(k_poly_red, dim3(256 + nextra), dim3(CA_TB), 0, st, w->Qpart, w->n_cell_blocks, (int64_t)NB * (R + 2) * 8, w->hdr,
                       (R + 2) * C, 1, C, w->tabQ, nullptr, 256, tail, xs);
  }
  return hipGetLastError();
}

hipError_t ca_poly_backward(hipStream_t st, const ca_poly_ws* w, const float* V, const float* mu, const float* Lb, int G, int C, double* red_g, const void* small_tail) {
  ca_small_args tail;
  static_assert(sizeof(ca_small_args) <= 512, "ca_small_args");
  if (small_tail) memcpy(&tail, small_tail, sizeof(tail)); else memset(&tail, 0, sizeof(tail));
  const int ngblk = cdiv_i(G, CA_TB);
  hipLaunchKernelGGL(k_poly_gene, dim3(ngblk + (tail.enabled ? 1 : 0)), dim3(CA_TB), 0, st, w->hdr, w->tabQ, V, mu, Lb, G, C, red_g, tail, ngblk);
  return hipGetLastError();
}

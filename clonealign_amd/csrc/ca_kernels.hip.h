// Device kernels of the clonealign VI engine, written for gfx950 (CDNA4, wave64).
//
// The model is R/inference-tflow.R:240-346 of the reference; the fused evaluation order
// (what is hoisted, what each sweep computes) is DESIGN.md §3.  Notation used below:
//   F[N][D]   cell factors  (psi | X)          V[G][D]  gene loadings (W | beta)
//   E_ng = exp(F_n . V_g)                      M_gc = mu_g * L_gc
//   Z_nc = sum_g E_ng M_gc                     coef_nc = -gamma_nc s_n / (S Z_nc)
// E is never stored: both sweeps regenerate it from F and V (one v_exp_f32 per (n,g)).
// To keep v_exp_f32 in range the exponent is shifted per cell by an upper bound
// etamax2_n >= max_g log2(E_ng); the shift cancels exactly in coef * E and is added back
// to log Z in the cell epilogue.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CA_CW 8  // clones per sweep launch ("chunk"); M, L, coef, Z rows are padded to 8 floats
#define CA_LOG2E_F 1.44269504088896340736f
#define CA_LN2 0.69314718055994530942
#define CA_LOG2PI 1.83787706640934548356
#define CA_TB 256

// ------------------------------------------------------------------ wave / block helpers
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float ca_dpp_pull(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
// sum over the 64 lanes of a wave; the total is valid in lane 63 (DPP only, no LDS)
__device__ __forceinline__ float ca_wave_sum_lane63(float v) {
  v += ca_dpp_pull<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
  v += ca_dpp_pull<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
  v += ca_dpp_pull<0x141, 0xF>(v);  // row_half_mirror
  v += ca_dpp_pull<0x140, 0xF>(v);  // row_mirror
  v += ca_dpp_pull<0x142, 0xA>(v);  // row_bcast:15 into rows 1,3
  v += ca_dpp_pull<0x143, 0xC>(v);  // row_bcast:31 into rows 2,3
  return v;
}

// v + v[lane ^ 16] + v[lane ^ 32] + v[lane ^ 48] in every lane, on the VALU: gfx950's v_permlane16_swap / v_permlane32_swap exchange
// odd and even rows of 16 lanes / the two halves of 32 between two registers; with the same value in both, the pair that
// comes back is (own-or-even copy, partner-or-odd copy), so their sum is x + x[lane ^ 16] (then ^ 32).  __shfl_xor would go
// through ds_bpermute_b32 and an s_waitcnt lgkmcnt(0) each.
__device__ __forceinline__ float ca_sum_xor16_32(float v) {
  typedef unsigned v2u __attribute__((ext_vector_type(2)));
  unsigned u = __float_as_uint(v);
  v2u r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  v = __uint_as_float(r.x) + __uint_as_float(r.y);
  u = __float_as_uint(v);
  r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(r.x) + __uint_as_float(r.y);
}

// deterministic block sum of doubles (blockDim.x == CA_TB); result valid in every thread.  Xor-butterfly inside each
// wave (no LDS, no barrier), then the CA_TB / 64 wave totals through LDS in wave order: 2 barriers instead of the
// 10 of an LDS tree -- the per-gene / per-cell / O(K + C) kernels are chains of these.
__device__ __forceinline__ double ca_block_sum(double v, double* sm) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();   // sm may still be read from a previous call
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = sm[0];
#pragma unroll
  for (int w = 1; w < CA_TB / 64; ++w) r += sm[w];
  return r;
}

// NV block sums at once: the NV butterflies interleave (their shuffle latencies overlap) and share ONE pair of barriers.
// Same additions in the same order as NV calls of ca_block_sum, so the results are bitwise the same.
template <int NV>
__device__ __forceinline__ void ca_block_sum_n(double (&v)[NV], double* sm /* >= (CA_TB / 64) * NV */) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] += __shfl_xor(v[i], o, 64);
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) sm[(threadIdx.x >> 6) * NV + i] = v[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    double r = sm[i];
#pragma unroll
    for (int w = 1; w < CA_TB / 64; ++w) r += sm[w * NV + i];
    v[i] = r;
  }
}

// order-preserving image of a float in a signed int (and back: the map is its own inverse): integer atomicMin / atomicMax on the images
// give the float min / max exactly
__device__ __forceinline__ int ca_f2ord(float f) { const int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7FFFFFFF; }
__device__ __forceinline__ float ca_ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7FFFFFFF); }
__device__ __forceinline__ unsigned short ca_bf16_rn(float f) {
  unsigned u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ double ca_softplus_d(double x) { return x > 0 ? x + log1p(exp(-x)) : log1p(exp(x)); }
__device__ __forceinline__ double ca_sigmoid_d(double x) { return 1.0 / (1.0 + exp(-x)); }

// Timing-lab hooks (block stamps: tools/stamps.py, tools/stamps_small.py).  The product build compiles them to nothing; their bodies live in
// tools/lab/ca_lab_hooks.inc and come in only with -DCA_LAB (`make -C clonealign_amd/csrc lab`), a build whose ca_build_id() starts with "lab-"
// and which bench.py therefore refuses.  No hook changes a result.
#ifdef CA_LAB
#include "../../tools/lab/ca_lab_hooks.inc"
#else
#define CA_LAB_STAMP(slot, kind) do { } while (0)
#define CA_LAB_CP(blk, i) do { } while (0)
#define CA_LAB_PH(blk, i) do { } while (0)
#define CA_LAB_PH_AFTER(value, blk, i) do { } while (0)
#define CA_LAB_BLOCK_T0() do { } while (0)
#define CA_LAB_BLOCK_END(kind, idx) do { } while (0)
#define CA_LAB_LEAVE(label) return
#define CA_LAB_LABEL(label) do { } while (0)
#endif

#include "ca_k_stream.hip.h"   // ingest / fit-constant kernels, the vector count-matrix stream (k_ypass), per-gene prologues, the VALU and first matrix-core forward sweeps
#include "ca_k_bwd.hip.h"   // backward sweeps (k_bwd, k_bwd_mfma), TF1 Adam, the O(K + C) ELBO assembly body, preprocessing and allele kernels
#include "ca_k_cell.hip.h"   // per-cell epilogues (plain and fused), the count-matrix finishers, the fused forward sweep with its cell epilogue (k_fwd_cell*) and the riding vector stream
// fixed-order reduction of block partials: out[j] = sum_b part[b][j]; one block per column j
__global__ void __launch_bounds__(CA_TB) k_reduce_part(const double* __restrict__ part, double* __restrict__ out, int nblk, int W_) {
  __shared__ double sm[CA_TB];
  const int j = blockIdx.x;
  double a = 0.0;
  for (int b = threadIdx.x; b < nblk; b += CA_TB) a += part[(int64_t)b * W_ + j];
  const double r = ca_block_sum(a, sm);
  if (threadIdx.x == 0) out[j] = r;
}

// ------------------------------------------------------------------ per-gene gradients + Adam
// d ELBO / d loc, ls (through mu = softplus(loc + exp(ls) eps)), W, beta; minimises -ELBO.
__device__ __forceinline__ bool ca_psi_adam_body_at(const ca_psi_args& a, int64_t n, int apply, float lr_t, float b1, float b2, float aeps, float* psi0_new,
                                                    const ca_gate* gt = nullptr);
__device__ __forceinline__ void ca_psi_adam_body(const ca_psi_args& a, int blk, int apply, float lr_t, float b1, float b2, float aeps,
                                                 float* psi0_new = nullptr /* merged update: the cell's stepped psi_0 (0 past the last cell) */) {
  ca_psi_adam_body_at(a, (int64_t)blk * CA_TB + threadIdx.x, apply, lr_t, b1, b2, aeps, psi0_new);
}
__device__ __forceinline__ bool ca_psi_adam_body_at(const ca_psi_args& a, int64_t n_, int apply, float lr_t, float b1, float b2, float aeps, float* psi0_new,
                                                    const ca_gate* gt) {
  if (psi0_new) *psi0_new = 0.f;
  // Lanes past the last cell run the same loads on the last cell's index and ask the gate with everybody else (they store nothing): the
  // answer is then uniform over the block -- an early return here, in front of the gate, let the padding lanes of the last wave go on to
  // the psi image after a "stop" (ADVICE r4).
  const bool live = n_ < a.N;
  if (!live && !gt) return true;
  const int64_t n = live ? n_ : a.N - 1;
  for (int k = 0; k < a.K; ++k) {
    // (everything this lane reads, in one batch in front of the first use: a dependent round of loads is 1.5 us here)
    const float yw_k = a.YW[n * a.K + k], f_k = a.F[n * a.D + k], m_k = a.m_psi[n * a.K + k], v_k = a.v_psi[n * a.K + k];
    double dF = 0.0;
    for (int t0 = 0; t0 < a.ntile; t0 += 20) {   // twenty loads in flight (a plain loop waits for each: 20 round trips), added in tile order
      float v[20];
#pragma unroll
      for (int i = 0; i < 20; ++i) v[i] = a.dFpart[((int64_t)(t0 + i < a.ntile ? t0 + i : a.ntile - 1) * a.N + n) * a.D + k];
#pragma unroll
      for (int i = 0; i < 20; ++i)
        if (t0 + i < a.ntile) dF += (double)v[i];
    }
    const float gp = (float)((double)yw_k + dF - (double)f_k);
    if (gt && k == 0 && !ca_gate_spin(*gt)) return false;   // (gated update: loads and arithmetic are done, nothing is stored yet)
    if (!live) continue;
    a.g_psi[n * a.K + k] = gp;
    if (apply) {
      float th = f_k, m = m_k, v = v_k;
      ca_adam(th, m, v, -gp, lr_t, b1, b2, aeps);
      a.F[n * a.D + k] = th; a.m_psi[n * a.K + k] = m; a.v_psi[n * a.K + k] = v;
      if (k == 0 && psi0_new) *psi0_new = th;
    }
  }
  return true;
}

struct ca_gene_new { float loc, ls, V0; double cs; int stopped; };   // a gene's stepped loc / ls / first loading (and its count total), in registers
__device__ __forceinline__ float ca_final_gene_step(int g, bool ok, const double* __restrict__ red_g, const double* __restrict__ red_y,
                                                     const float* __restrict__ eps, const double* __restrict__ colsum,
                                                     const double* __restrict__ YtX, const float* __restrict__ vchi,
                                                     float* __restrict__ loc, float* __restrict__ ls, float* __restrict__ V,
                                                     float* __restrict__ m_loc, float* __restrict__ v_loc, float* __restrict__ m_ls,
                                                     float* __restrict__ v_ls, float* __restrict__ m_V, float* __restrict__ v_V,
                                                     float* __restrict__ g_loc, float* __restrict__ g_ls, float* __restrict__ g_V,
                                                     int G, int S, int D, int K, int apply, float lr_t, float b1, float b2, float aeps,
                                                     const float* __restrict__ gfold, int nfold, ca_gene_new* nw,
                                                     const double* __restrict__ aux = nullptr, int64_t aux_ld = 0, const ca_gate* gt = nullptr);
__device__ __forceinline__ void ca_final_gene_range(int g, bool ok, float Vnew0, const float* __restrict__ V, float* __restrict__ Vs,
                                                     float* __restrict__ vmm_part, int G, int D, int blk, float* smin, float* smax);
__device__ __forceinline__ void ca_final_gene_body(const double* __restrict__ red_g /*[G][S+D]*/, const double* __restrict__ red_y /*[G][K]*/,
                                                      const float* __restrict__ eps, const double* __restrict__ colsum,
                                                      const double* __restrict__ YtX, const float* __restrict__ vchi,
                                                      float* __restrict__ loc, float* __restrict__ ls, float* __restrict__ V,
                                                      float* __restrict__ m_loc, float* __restrict__ v_loc, float* __restrict__ m_ls,
                                                      float* __restrict__ v_ls, float* __restrict__ m_V, float* __restrict__ v_V,
                                                      float* __restrict__ g_loc, float* __restrict__ g_ls, float* __restrict__ g_V,
                                                      float* __restrict__ Vs, float* __restrict__ vmm_part,
                                                      int G, int S, int D, int K, int apply, float lr_t, float b1, float b2, float aeps, float* smin, float* smax,
                                                      const float* __restrict__ gfold /*[nfold][G][S+D] or null*/, int nfold,
                                                      ca_gene_new* nw = nullptr /* merged update: the stepped values stay in registers */) {
  const int g = blockIdx.x * CA_TB + threadIdx.x;
  const bool ok = g < G;
  const float Vnew0 = ca_final_gene_step(g, ok, red_g, red_y, eps, colsum, YtX, vchi, loc, ls, V, m_loc, v_loc, m_ls, v_ls, m_V, v_V, g_loc, g_ls, g_V,
                                         G, S, D, K, apply, lr_t, b1, b2, aeps, gfold, nfold, nw);
  if (!apply) return;
  ca_final_gene_range(g, ok, Vnew0, V, Vs, vmm_part, G, D, (int)blockIdx.x, smin, smax);
}
// one gene's gradients and Adam step (per lane, no block-level operation): returns the stepped first loading
__device__ __forceinline__ float ca_final_gene_step(int g, bool ok, const double* __restrict__ red_g, const double* __restrict__ red_y,
                                                     const float* __restrict__ eps, const double* __restrict__ colsum,
                                                     const double* __restrict__ YtX, const float* __restrict__ vchi,
                                                     float* __restrict__ loc, float* __restrict__ ls, float* __restrict__ V,
                                                     float* __restrict__ m_loc, float* __restrict__ v_loc, float* __restrict__ m_ls,
                                                     float* __restrict__ v_ls, float* __restrict__ m_V, float* __restrict__ v_V,
                                                     float* __restrict__ g_loc, float* __restrict__ g_ls, float* __restrict__ g_V,
                                                     int G, int S, int D, int K, int apply, float lr_t, float b1, float b2, float aeps,
                                                     const float* __restrict__ gfold, int nfold, ca_gene_new* nw,
                                                     const double* __restrict__ aux, int64_t aux_ld, const ca_gate* gt) {
  float Vnew0 = 0.f;   // the first loading after this step (kept in a register for the log2 image)
  if (ok) {
  // Every operand whose address does not depend on a result is loaded HERE, in one batch: this block is one wave per SIMD, and each
  // dependent round of loads costs it 1.5 us after the sweeps have been through the caches and the TLB (block stamps,
  // tools/stamps_small.py: the gene blocks, at four rounds, were the 8 us this kernel took).
  const float loc_g = loc[g], ls_g = ls[g];
  const double cs = colsum[g];
  const float e0f = eps[g];
  const float mloc_g = m_loc[g], vloc_g = v_loc[g], mls_g = m_ls[g], vls_g = v_ls[g];
  float V0 = 0.f, mV0 = 0.f, vV0 = 0.f, vchi0 = 0.f;
  double ry0 = 0.0;
  if (D > 0) { V0 = V[(int64_t)g * D]; mV0 = m_V[(int64_t)g * D]; vV0 = v_V[(int64_t)g * D]; }
  if (K > 0) { vchi0 = vchi[0]; ry0 = red_y[(int64_t)g * K]; }
  // (aux: the sweep-independent part of this gene's gradient as the prologue of this pass's eps left it, ca_gene_pre_draw; S = 1)
  double a_sd = 0.0, a_sig = 0.0, a_q1 = 0.0, a_q2 = 0.0, a_q3 = 0.0;
  if (aux) { a_sd = aux[g]; a_sig = aux[aux_ld + g]; a_q1 = aux[2 * aux_ld + g]; a_q2 = aux[3 * aux_ld + g]; a_q3 = aux[4 * aux_ld + g]; }
  const double l = (double)loc_g, lsd = (double)ls_g, sd = aux ? a_sd : exp(lsd);
  const int W_ = S + D;
  // small problems: the backward sweep's cell-split partials are summed here (fixed order, fp64) instead of by a k_colsum
  // launch of their own -- one launch and its gap less per iteration where launches are what an iteration costs
  // (computed where it is used, once per column: an indexed local array would live in scratch memory)
  double rg2[2] = {0.0, 0.0};
  const bool two = gfold && W_ == 2;   // the common case (one sample, one latent dimension): both columns from one 8-byte load, forty
  if (two) {                            // slices in flight -- ONE round at the 38 slices of a resident round of sweep blocks (a round of
    for (int sp0 = 0; sp0 < nfold; sp0 += 40) {   // these loads is 1.4 us: the slices come from the other XCDs' sweep blocks; checkpoints in
      float2 v[40];                                // tools/stamps_small.py), where eight at a time, one column after the other, took ten
#pragma unroll
      for (int i = 0; i < 40; ++i)
        v[i] = *reinterpret_cast<const float2*>(gfold + ((int64_t)(sp0 + i < nfold ? sp0 + i : nfold - 1) * G + g) * 2);
#pragma unroll
      for (int i = 0; i < 40; ++i)
        if (sp0 + i < nfold) { rg2[0] += (double)v[i].x; rg2[1] += (double)v[i].y; }
    }
  }
  auto rgv = [&](int w) -> double {
    if (!gfold) return red_g[(int64_t)g * W_ + w];
    if (two) return w == 0 ? rg2[0] : rg2[1];
    double a = 0.0;
    for (int sp0 = 0; sp0 < nfold; sp0 += 8) {   // eight loads in flight, added in slice order
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = gfold[((int64_t)(sp0 + i < nfold ? sp0 + i : nfold - 1) * G + g) * W_ + w];
#pragma unroll
      for (int i = 0; i < 8; ++i) a += (sp0 + i < nfold) ? (double)v[i] : 0.0;
    }
    return a;
  };
  double gl = 0.0, gs = 0.0;
  for (int s = 0; s < S; ++s) {
    const double e = s == 0 ? (double)e0f : (double)eps[(int64_t)s * G + g];
    double sig, q1, q2, q3;
    if (aux) { sig = a_sig; q1 = a_q1; q2 = a_q2; q3 = a_q3; }
    else {
      const double x = l + sd * e;
      // softplus and sigmoid from ONE exp: t = exp(-|x|); softplus = max(x, 0) + log1p(t); sigmoid = 1 / (1 + t) or t / (1 + t)
      const double tx = exp(-fabs(x));
      const double mu = (x > 0 ? x : 0.0) + log1p(tx), lm = log(mu);
      sig = (x >= 0 ? 1.0 : tx) / (1.0 + tx);
      q1 = cs / ((double)S * mu); q2 = lm / ((double)S * mu); q3 = (1.0 - sig) / (double)S;
    }
    const double dmu = q1 + rgv(s) - q2;
    const double dx = dmu * sig + q3;
    gl += dx;
    gs += dx * e * sd;
  }
  gs += 1.0;
  if (gt && !ca_gate_spin(*gt)) { if (nw) nw->stopped = 1; return 0.f; }   // (gated update: everything is loaded and summed, nothing is stored yet)
  g_loc[g] = (float)gl;
  g_ls[g] = (float)gs;
  if (apply) {
    float th = loc_g, m = mloc_g, v = vloc_g;
    ca_adam(th, m, v, -(float)gl, lr_t, b1, b2, aeps);
    loc[g] = th; m_loc[g] = m; v_loc[g] = v;
    if (nw) { nw->loc = th; nw->cs = cs; }
    th = ls_g; m = mls_g; v = vls_g;
    ca_adam(th, m, v, -(float)gs, lr_t, b1, b2, aeps);
    ls[g] = th; m_ls[g] = m; v_ls[g] = v;
    if (nw) nw->ls = th;
  }
  for (int d = 0; d < D; ++d) {
    double gv = rgv(S + d);
    const float Vd = d == 0 ? V0 : V[(int64_t)g * D + d];
    if (d < K) gv += (d == 0 ? ry0 : red_y[(int64_t)g * K + d]) - exp((double)(d == 0 ? vchi0 : vchi[d])) * (double)Vd;
    else gv += YtX[(int64_t)g * (D - K) + (d - K)];
    g_V[(int64_t)g * D + d] = (float)gv;
    if (apply) {
      float th = Vd, m = d == 0 ? mV0 : m_V[(int64_t)g * D + d], v = d == 0 ? vV0 : v_V[(int64_t)g * D + d];
      ca_adam(th, m, v, -(float)gv, lr_t, b1, b2, aeps);
      V[(int64_t)g * D + d] = th; m_V[(int64_t)g * D + d] = m; v_V[(int64_t)g * D + d] = v;
      if (d == 0) Vnew0 = th;
    }
  }
  }
  if (nw) nw->V0 = Vnew0;
  return Vnew0;
}
// the stepped loadings in log2 units and their per-block range (k_vprep fused in; same arithmetic); block-level: the four waves of a
// 256-gene block (threads 0 .. 255 by `wave4`, the wave's place among them) meet through LDS
__device__ __forceinline__ void ca_final_gene_range(int g, bool ok, float Vnew0, const float* __restrict__ V, float* __restrict__ Vs,
                                                     float* __restrict__ vmm_part, int G, int D, int blk, float* smin, float* smax) {
  for (int d = 0; d < D; ++d) {
    float v = 0.f;
    if (ok) {
      v = (d == 0 ? Vnew0 : V[(int64_t)g * D + d]) * CA_LOG2E_F;
      Vs[(int64_t)g * D + d] = v;
      if (g == G - 1)   // pad to a multiple of 32 genes with the last gene's loading (k_fwd_cell reads whole k-steps)
        for (int gp = G; gp < ((G + 31) / 32) * 32; ++gp) Vs[(int64_t)gp * D + d] = v;
    }
    float mn = ok ? v : INFINITY, mx = ok ? v : -INFINITY;   // wave butterflies, then the four wave results through LDS
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      mn = fminf(mn, __shfl_xor(mn, o, 64));
      mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = mn; smax[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
      vmm_part[((int64_t)blk * 2 + 0) * D + d] = fminf(fminf(smin[0], smin[1]), fminf(smin[2], smin[3]));
      vmm_part[((int64_t)blk * 2 + 1) * D + d] = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
    }
  }
}

__global__ void __launch_bounds__(CA_TB) k_final_gene(const double* __restrict__ red_g /*[G][S+D]*/, const double* __restrict__ red_y /*[G][K]*/,
                                                      const float* __restrict__ eps, const double* __restrict__ colsum,
                                                      const double* __restrict__ YtX, const float* __restrict__ vchi,
                                                      float* __restrict__ loc, float* __restrict__ ls, float* __restrict__ V,
                                                      float* __restrict__ m_loc, float* __restrict__ v_loc, float* __restrict__ m_ls,
                                                      float* __restrict__ v_ls, float* __restrict__ m_V, float* __restrict__ v_V,
                                                      float* __restrict__ g_loc, float* __restrict__ g_ls, float* __restrict__ g_V,
                                                      float* __restrict__ Vs, float* __restrict__ vmm_part,
                                                      int G, int S, int D, int K, int apply, float lr_t, float b1, float b2, float aeps, ca_small_args mon, int gblocks,
                                                      ca_psi_args psi, const float* __restrict__ gfold, int nfold) {
  CA_LAB_STAMP((int)blockIdx.x, (int)blockIdx.x < gblocks ? 0 : (mon.enabled && (int)blockIdx.x == gblocks) ? 1 : 2);
  if ((int)blockIdx.x >= gblocks) {
    int b = (int)blockIdx.x - gblocks;
    if (mon.enabled) {   // one extra block: the pending monitor pass's ELBO (ca_final_small_body), beside the gene blocks
      if (b == 0) { ca_final_small_body(mon); return; }
      --b;
    }
    if (b < psi.nblk) ca_psi_adam_body(psi, b, apply, lr_t, b1, b2, aeps);
    return;
  }
  __shared__ float smin[CA_TB], smax[CA_TB];
  ca_final_gene_body(red_g, red_y, eps, colsum, YtX, vchi, loc, ls, V, m_loc, v_loc, m_ls, v_ls, m_V, v_V, g_loc, g_ls, g_V, Vs, vmm_part, G, S, D, K,
                     apply, lr_t, b1, b2, aeps, smin, smax, gfold, nfold);
}

// ------------------------------------------------------------------ ELBO assembly + the O(K + C) variables (body: ca_final_small_body above)
__global__ void __launch_bounds__(CA_TB) k_final_small(ca_small_args a) { ca_final_small_body(a); }

// ------------------------------------------------------------------ per-cell variables: the q(z) logits and the exponent bound
// (psi's own step runs as extra blocks of k_final_gene, see ca_psi_args)
// ------------------------------------------------------------------ count-matrix products on the int8 matrix cores
#include "ca_ymfma.hip.h"

// q(z) logits: an elementwise Adam step over the flat [N * C] arrays, 16 bytes per lane; cell block `cblk` = cells 256 cblk ...
__device__ __forceinline__ void ca_logit_adam_body(int cblk, float* __restrict__ glogit, const float* __restrict__ dgl, float* __restrict__ m_gl,
                                                   float* __restrict__ v_gl, int64_t N, int C, float lr_t, float b1, float b2, float aeps, int tix = -1) {
  if (tix < 0) tix = (int)threadIdx.x;   // (thread's place among the 256 of the piece)
  const int64_t e0 = (int64_t)cblk * CA_TB * C, tot = N * (int64_t)C;
  const int64_t e1 = e0 + (int64_t)CA_TB * C < tot ? e0 + (int64_t)CA_TB * C : tot;   // e0 is a multiple of 4 (CA_TB = 256)
  for (int64_t i = e0 + 4 * (int64_t)tix; i < e1; i += 4 * CA_TB) {
    if (i + 4 <= e1) {
      float4 th = *reinterpret_cast<const float4*>(glogit + i), m = *reinterpret_cast<const float4*>(m_gl + i);
      float4 v = *reinterpret_cast<const float4*>(v_gl + i);
      const float4 g = *reinterpret_cast<const float4*>(dgl + i);
      ca_adam(th.x, m.x, v.x, -g.x, lr_t, b1, b2, aeps);
      ca_adam(th.y, m.y, v.y, -g.y, lr_t, b1, b2, aeps);
      ca_adam(th.z, m.z, v.z, -g.z, lr_t, b1, b2, aeps);
      ca_adam(th.w, m.w, v.w, -g.w, lr_t, b1, b2, aeps);
      *reinterpret_cast<float4*>(glogit + i) = th;
      *reinterpret_cast<float4*>(m_gl + i) = m;
      *reinterpret_cast<float4*>(v_gl + i) = v;
    } else {
      for (int64_t k = i; k < e1; ++k) {
        float th = glogit[k], m = m_gl[k], v = v_gl[k];
        ca_adam(th, m, v, -dgl[k], lr_t, b1, b2, aeps);
        glogit[k] = th; m_gl[k] = m; v_gl[k] = v;
      }
    }
  }
}

__global__ void __launch_bounds__(CA_TB) k_adam_cell(const float* __restrict__ F, float* __restrict__ glogit, const float* __restrict__ dgl,
                                                     float* __restrict__ m_gl, float* __restrict__ v_gl, int64_t N, int C, int D,
                                                     int apply, float lr_t, float b1, float b2, float aeps,
                                                     const float* __restrict__ vmm_part, int ngblk, float* __restrict__ etamax2,
                                                     ca_small_args tail, int cblocks, ca_pre_args pre, ca_ysq_args ysq) {
  // Block order = dispatch order: the two latency chains first (the next pass's per-gene prologue, then the O(K + C) update),
  // the bandwidth-bound cell blocks after them -- the chains are what the kernel's duration hangs on.  Last: the quantiser of
  // the int8 count-matrix stream (ca_ys_quant_body), when that stream is in use: W and psi are final once k_final_gene has run.
  const int nx = pre.nblk + 1;
  CA_LAB_STAMP(1024 + (int)blockIdx.x, (int)blockIdx.x < pre.nblk ? 3 : (int)blockIdx.x < nx ? 4 : (int)blockIdx.x < nx + cblocks ? 5 : 6);
  if ((int)blockIdx.x >= nx + cblocks) {
    __shared__ float smq[2 * (CA_YM_TB / 64)];
    ca_ys_quant_body((int)blockIdx.x - nx - cblocks, ysq, smq);
    return;
  }
  if ((int)blockIdx.x < nx) {
    const int b = (int)blockIdx.x;
    if (b < pre.nblk) {   // the next eps pair's per-gene prologue (ca_pre_args)
      __shared__ double smp[CA_TB];
      ca_gene_pre_fused_body(pre.loc, pre.ls, pre.epsA, pre.epsB, pre.colsum, pre.Lb, pre.V, pre.D, pre.K, pre.YtX, pre.muA, pre.muB, pre.Mb,
                             pre.gene_partA, pre.gene_partB, pre.G, pre.mrow, pre.C, pre.Mq, smp, b, pre.s2);
    } else {              // chi / alpha gradients and Adam, the range of V' (ca_final_small_body)
      if (tail.enabled) ca_final_small_body(tail);
    }
    return;
  }
  const int cblk = (int)blockIdx.x - nx;   // cell block
  (void)cblocks;
  // q(z) logits: an elementwise step over the flat [N * C] arrays, 16 bytes per lane (a lane per cell would fetch C
  // strided floats per array: 2.4 TB/s at 100k x 8)
  if (apply) ca_logit_adam_body(cblk, glogit, dgl, m_gl, v_gl, N, C, lr_t, b1, b2, aeps);
  // range of the updated V' over the gene blocks, per block (k_vmm_final folded in: same min / max as the extra block's)
  __shared__ float vmm[2 * 8];
  if (apply && D > 0 && D <= 8 && (int)threadIdx.x < 64) {   // wave 0: a lane per gene block, then butterflies (min / max: any order)
    for (int d = 0; d < D; ++d) {
      float mn = INFINITY, mx = -INFINITY;
      for (int b = threadIdx.x; b < ngblk; b += 64) {
        mn = fminf(mn, vmm_part[((int64_t)b * 2 + 0) * D + d]);
        mx = fmaxf(mx, vmm_part[((int64_t)b * 2 + 1) * D + d]);
      }
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { mn = fminf(mn, __shfl_xor(mn, o, 64)); mx = fmaxf(mx, __shfl_xor(mx, o, 64)); }
      if (threadIdx.x == 0) { vmm[d] = mn; vmm[D + d] = mx; }
    }
  }
  __syncthreads();
  const int64_t n = (int64_t)cblk * CA_TB + threadIdx.x;
  if (n >= N) return;
  if (apply && D > 0) {   // exponent bound for the updated psi and V' (k_etamax folded in)
    float e = 0.f;
    for (int d = 0; d < D; ++d) {
      const float f = F[n * D + d];
      e += fmaxf(f * vmm[d], f * vmm[D + d]);
    }
    etamax2[n] = e;
  }
}


// ------------------------------------------------------------------ the whole update half of a train pass in ONE launch (round 4)
// k_final_gene + k_adam_cell were two launches because three things waited for ALL gene blocks: the next eps pair's per-gene prologue, the
// int8 stream's quantiser (both only need THEIR gene's / cell's stepped values) and the per-cell exponent bound (needs the range of V' over
// all genes).  Here a gene block goes from its Adam step straight on to the prologue and to its four 64-steps of the W image with the
// stepped values still in registers; a psi block does the same for the psi image; the exponent bound is taken by the next forward sweep's
// blocks themselves (ca_cell_ptrs::vmm_part); the q(z) logits (they depend on the forward sweep only) and the chi / alpha step are further
// blocks of this launch -- chi and alpha go to alternate buffers, because the gene blocks and the pending monitor pass's block still read
// the values the gradients belong to.  One launch, one latency chain and one kernel boundary less per iteration; the arithmetic of every
// piece is the two-launch form's, on the same floats (tests: bitwise equal with the variant switched off).
// Block order = dispatch order: the latency chains (gene blocks, monitor block, chi / alpha block) first, then psi, then the logits.
struct ca_merge_args {
  ca_pre_args pre;         // the next (monitor, train) eps pair's prologue: pre.nblk == gene blocks
  ca_ysq_args ysq;         // nblk > 0: the int8 stream's images, made in the gene / psi blocks (pairs of maxima: gene blocks, then psi blocks)
  ca_small_args tail;      // the chi / alpha step (vchi_out / alpha_out set)
  float* glogit; const float* dgl; float* m_gl; float* v_gl; int C; int ncell;   // q(z) logits: ncell blocks of 256 cells
  const double* aux_in; double* aux_out; int64_t aux_ld;   // [5][aux_ld] doubles per gene: the sweep-independent part of the NEXT step's gradient (ca_gene_pre_draw);
                                                           // aux_in: what the prologue of THIS pass's eps left (null: the step computes it), aux_out: for the next step
  // ca_run's gate (round 4): the launch is queued BEFORE the host has seen the ELBO the stop rule needs -- that ELBO is assembled by this
  // launch's own monitor block, which is not gated -- and every other block waits for the decision.  Round 5: ONE block decides, the relay
  // (the chi / alpha block; it and the monitor block are dispatched FIRST, so they run whatever the device's occupancy -- gene blocks that
  // fill a small partition can no longer keep them out).  It polls the host's word (gate_seq << 1) | go in pinned memory for at most
  // gate_timeout ticks (100 MHz; ~1 ms by default): "go", "stop", or -- no answer in time, the host is in a slow poll hook, was descheduled or
  // is stopped in a debugger -- "gave up".  Its verdict goes to device memory for every other block (gate_local: go, or store nothing) and
  // to the host (gate_ack: (gate_seq << 2) | 1 go, 0 stop, 2 gave up).  A launch that gave up stores NOTHING and is not an error: the host
  // puts its bookkeeping of the step back and queues the update again after its decision (the lock-step loop), so a slow hook costs the
  // GPU gate_timeout of one block's polling, then the device is idle, and the fit is unchanged bit for bit.
  const unsigned long long* gate; unsigned long long gate_seq, gate_timeout; unsigned long long* gate_ack;
  unsigned long long* gate_err;     // pinned: a waiter whose safety deadline (gate_timeout + 10 s) ran out -- the relay never ran; fatal, never seen
  unsigned long long* gate_local;   // device memory: the relay's verdict, (gate_seq << 1) | go; the other blocks (and a forward sweep queued behind) read this
  int* vmm_at; int* vmm_at_next;   // range of V' over ALL genes as ordered ints [2][8]: every gene block folds its own in with one atomic min / max per
                                   // dimension (the next sweep reads 2 D words); the chi / alpha block resets the buffer of the NEXT merged update
  float* xpart;                    // series form (ca_poly.hip, K = D = 1): max |psi| of the STEPPED state per 256-cell piece, [psi.nblk] -- k_poly_B / k_poly_ranges take
                                   // the maximum of these instead of the word k_poly_xmax makes as a launch of its own; null: nobody wants it.  (One atomic
                                   // maximum per wave into that word instead made this launch 14 us longer: 1563 atomics on one address.)
};
#define CA_GATE_WAITER_EXTRA (10ull * 100000000ull)   // a waiter's deadline over the relay's: 10 s of s_memrealtime ticks
// every thread of the block calls it; true = go on (no gate, or the verdict is go).  relay: this block is the one that reads the host's word
// (pinned memory, a PCIe round trip per look) and passes the verdict on through device memory; two hundred blocks polling the host's line
// themselves made a ca_run iteration 160 us LONGER (gpurun_out/r4/run_gate1.txt)
__device__ __forceinline__ bool ca_gate_wait(const ca_merge_args& mg, bool relay) {
  if (!mg.gate) return true;
  __shared__ unsigned gate_go;
  if (threadIdx.x == 0) {
    unsigned go = 0u;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (relay) {
      unsigned outcome = 2u;   // gave up
      for (;;) {
        const unsigned long long w = __hip_atomic_load(mg.gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((w >> 1) == mg.gate_seq) { outcome = (unsigned)(w & 1ull); break; }
        if (__builtin_amdgcn_s_memrealtime() - t0 > mg.gate_timeout) break;
        __builtin_amdgcn_s_sleep(4);
      }
      go = outcome == 1u ? 1u : 0u;
      __hip_atomic_store(mg.gate_local, (mg.gate_seq << 1) | (unsigned long long)go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(mg.gate_ack, (mg.gate_seq << 2) | (unsigned long long)outcome, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } else {
      for (;;) {
        const unsigned long long w = __hip_atomic_load(mg.gate_local, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((w >> 1) == mg.gate_seq) { go = (unsigned)(w & 1ull); break; }
        if (__builtin_amdgcn_s_memrealtime() - t0 > mg.gate_timeout + CA_GATE_WAITER_EXTRA) {   // (the relay never ran: see ca_gate)
          __hip_atomic_store(mg.gate_err, mg.gate_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          break;
        }
        __builtin_amdgcn_s_sleep(16);
      }
    }
    gate_go = go;
  }
  __syncthreads();
  return gate_go != 0u;
}
#define CA_UM_TB 1024   // threads of a k_update_merged block: a gene block is 256 genes x four ROLES (below); every other kind of block uses its first 256
__global__ void __launch_bounds__(CA_UM_TB) k_update_merged(const double* __restrict__ red_g /*[G][S+D]*/, const double* __restrict__ red_y /*[G][K]*/,
                                                            const float* __restrict__ eps, const double* __restrict__ colsum,
                                                            const double* __restrict__ YtX, const float* __restrict__ vchi,
                                                            float* __restrict__ loc, float* __restrict__ ls, float* __restrict__ V,
                                                            float* __restrict__ m_loc, float* __restrict__ v_loc, float* __restrict__ m_ls,
                                                            float* __restrict__ v_ls, float* __restrict__ m_V, float* __restrict__ v_V,
                                                            float* __restrict__ g_loc, float* __restrict__ g_ls, float* __restrict__ g_V,
                                                            float* __restrict__ Vs, float* __restrict__ vmm_part,
                                                            int G, int S, int D, int K, float lr_t, float b1, float b2, float aeps, ca_small_args mon, int gblocks,
                                                            ca_psi_args psi, const float* __restrict__ gfold, int nfold, ca_merge_args mg) {
  const int nmon = mon.enabled ? 1 : 0;
  // Block order = dispatch order.  FIRST the two O(K + C) blocks: the pending monitor pass's ELBO (never gated: it makes the ELBO the host
  // decides on) and the chi / alpha step, which is also the gate's relay -- every other block of a gated launch waits for ITS verdict, so
  // it must get a slot whatever the device's occupancy (behind the gene blocks, a partition with fewer slots than gene blocks
  // never dispatched it and every ca_run stalled for the gate's time limit).  Then the latency chains (gene blocks), psi, the logits.
  if ((int)blockIdx.x < nmon + 1) {   // 256-thread blocks
    const int b = (int)blockIdx.x;
    CA_LAB_STAMP(gblocks + b, mon.enabled && b == 0 ? 1 : 4);
    if ((int)threadIdx.x >= CA_TB) return;
    if (mon.enabled && b == 0) { CA_LAB_CP(40, 0); ca_final_small_body(mon); CA_LAB_CP(40, 1); return; }
    if (!ca_gate_wait(mg, true)) return;
    if (threadIdx.x < 8) { mg.vmm_at_next[threadIdx.x] = ca_f2ord(INFINITY); mg.vmm_at_next[8 + threadIdx.x] = ca_f2ord(-INFINITY); }
    CA_LAB_CP(41, 0); if (mg.tail.enabled) ca_final_small_body(mg.tail); CA_LAB_CP(41, 1);
    return;
  }
  const int bx = (int)blockIdx.x - (nmon + 1);   // gene block index, then psi / logit pieces behind the gene blocks
  CA_LAB_STAMP(bx < gblocks ? bx : bx + nmon + 1, bx < gblocks ? 0 : bx < gblocks + (psi.nblk + 3) / 4 ? 2 : 5);
  const ca_gate gt = {mg.gate ? mg.gate_local : nullptr, mg.gate_seq, mg.gate_timeout + CA_GATE_WAITER_EXTRA, mg.gate_err};   // what the waiting blocks ask
  if (bx < gblocks) {
    // A gene's chain here is: its Adam step (one round of loads, then fp64 exp / log1p / log and three Adam steps: 5 us at one wave per
    // SIMD), THEN the two draws of the next prologue (2.7 us each: softplus, log, the operand row) and its part of the W image (2 us) --
    // 12.5 us when one thread does them in a row (block stamps, profiles/r04_update_merge.txt).  The three pieces behind the step need
    // only the stepped loc / ls / W_g0, so sixteen waves share a block of 256 genes: waves 0-3 take the step, hand the three floats over
    // through LDS and go on to V' (log2 units, range) and the sums of squares; waves 4-7 take draw A of the same genes, 8-11 draw B,
    // 12-15 the W image.  Every block-level sum keeps the order of the 256-thread form: butterflies inside a 64-gene wave, then the
    // four gene groups in order -- bitwise the same partials.
    __shared__ int stop_s;
    __shared__ float h_loc[CA_TB], h_ls[CA_TB], h_v0[CA_TB];
    __shared__ double smt[4][10];   // (three sums of either draw, then the sums of squared loadings of up to four latent dimensions)
    __shared__ float smn[8][4], smx[8][4], sma[4];
    const int tid = (int)threadIdx.x, l = tid & 63, wv = tid >> 6, role = wv >> 2, grp = wv & 3;
    const int g = bx * CA_TB + grp * 64 + l;
    const bool ok = g < G;
    ca_gene_pre_ops o = {0.f, 0.f, 0.f, 0.f, 0.f, 0.0, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    float V0n = 0.f;
    CA_LAB_CP(bx, 0);
    if (role == 0) {
      ca_gene_new nw = {0.f, 0.f, 0.f, 0.0, 0};
      V0n = ca_final_gene_step(g, ok, red_g, red_y, eps, colsum, YtX, vchi, loc, ls, V, m_loc, v_loc, m_ls, v_ls, m_V, v_V, g_loc, g_ls, g_V,
                               G, S, D, K, 1, lr_t, b1, b2, aeps, gfold, nfold, &nw, mg.aux_in, mg.aux_ld, &gt);
      h_loc[grp * 64 + l] = nw.loc; h_ls[grp * 64 + l] = nw.ls; h_v0[grp * 64 + l] = V0n;
      if (l == 0 && ok) stop_s = nw.stopped;   // (the four waves that hold genes write the same answer; lane 0 holds a gene whenever its wave does)
    } else if (role < 3 && ok) {   // the draws' operands that nothing here produces: in flight while the step runs
      o.eA = mg.pre.epsA[g]; o.eB = mg.pre.epsB[g]; o.cs = colsum[g];
      o.lr0 = *reinterpret_cast<const float4*>(mg.pre.Lb + (int64_t)g * CA_CW); o.lr1 = *reinterpret_cast<const float4*>(mg.pre.Lb + (int64_t)g * CA_CW + 4);
    }
    __syncthreads();
    CA_LAB_CP(bx, 1);
    if (mg.gate && stop_s) return;   // the host said stop: this launch stores nothing
    if (role == 0) {
      // the stepped loadings in log2 units and their range over the block (ca_final_gene_range's arithmetic, wave level)
      for (int d = 0; d < D && d < 8; ++d) {
        float v = 0.f;
        if (ok) {
          v = (d == 0 ? V0n : V[(int64_t)g * D + d]) * CA_LOG2E_F;
          Vs[(int64_t)g * D + d] = v;
          if (g == G - 1)
            for (int gp = G; gp < ((G + 31) / 32) * 32; ++gp) Vs[(int64_t)gp * D + d] = v;
        }
        float mn = ok ? v : INFINITY, mx = ok ? v : -INFINITY;
#pragma unroll
        for (int q = 1; q < 64; q <<= 1) { mn = fminf(mn, __shfl_xor(mn, q, 64)); mx = fmaxf(mx, __shfl_xor(mx, q, 64)); }
        if (l == 0) { smn[d][grp] = mn; smx[d][grp] = mx; }
      }
      // sums of squared loadings (terms 6 and 7 of the prologue's block sums)
      const float wk0 = K > 0 ? V0n : 0.f;
      double esq[4] = {ok ? (double)wk0 * (double)wk0 : 0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int k = 1; k < 4; ++k)
        if (k < K && ok) { const double wk = (double)V[(int64_t)g * D + k]; esq[k] = wk * wk; }
#pragma unroll
      for (int q = 1; q < 64; q <<= 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) esq[k] += __shfl_xor(esq[k], q, 64);
      }
      if (l == 0) { smt[grp][6] = esq[0]; smt[grp][7] = esq[1]; smt[grp][8] = esq[2]; smt[grp][9] = esq[3]; }
    } else if (role < 3) {
      double t[3] = {0.0, 0.0, 0.0};
      if (ok) {
        o.loc = h_loc[grp * 64 + l]; o.ls = h_ls[grp * 64 + l];
        ca_gene_pre_draw(role - 1, g, o, mg.pre.Lb, V, D, K, YtX, mg.pre.muA, mg.pre.muB, mg.pre.Mb, G, mg.pre.mrow, mg.pre.C, mg.pre.Mq, t,
                         role == 2 ? mg.aux_out : nullptr, mg.aux_ld);
      }
#pragma unroll
      for (int q = 1; q < 64; q <<= 1) {
#pragma unroll
        for (int i = 0; i < 3; ++i) t[i] += __shfl_xor(t[i], q, 64);
      }
      if (l == 0) { smt[grp][3 * (role - 1) + 0] = t[0]; smt[grp][3 * (role - 1) + 1] = t[1]; smt[grp][3 * (role - 1) + 2] = t[2]; }
    } else {
      float m = 0.f;
      if (mg.ysq.nblk) {
        const int64_t step = (int64_t)bx * (CA_TB / 64) + grp;
        const bool live = step < (int64_t)mg.ysq.GS;
        m = ca_ys_quant_wave(mg.ysq, live, true, live ? step : 0, h_v0[grp * 64 + l], bx == 0 && grp == 0);
      }
      if (l == 0) sma[grp] = m;
    }
    __syncthreads();
    CA_LAB_CP(bx, 2);
    if (tid == 0) {
      double e[10];
#pragma unroll
      for (int i = 0; i < 10; ++i) { double r = smt[0][i]; r += smt[1][i]; r += smt[2][i]; r += smt[3][i]; e[i] = r; }
      const int W_ = 3 + K;
      double* ga = mg.pre.gene_partA + (int64_t)bx * W_;
      double* gb = mg.pre.gene_partB + (int64_t)bx * W_;
      if (mg.pre.s2) { ga[0] = 0.5 * (e[0] + e[3]); ga[1] = 0.5 * (e[1] + e[4]); ga[2] = 0.5 * (e[2] + e[5]); }
      else { ga[0] = e[0]; ga[1] = e[1]; ga[2] = e[2]; }
      gb[0] = e[3]; gb[1] = e[4]; gb[2] = e[5];
      for (int k = 0; k < K && k < 4; ++k) { ga[3 + k] = e[6 + k]; gb[3 + k] = e[6 + k]; }
      for (int d = 0; d < D && d < 8; ++d) {
        vmm_part[((int64_t)bx * 2 + 0) * D + d] = fminf(fminf(smn[d][0], smn[d][1]), fminf(smn[d][2], smn[d][3]));
        vmm_part[((int64_t)bx * 2 + 1) * D + d] = fmaxf(fmaxf(smx[d][0], smx[d][1]), fmaxf(smx[d][2], smx[d][3]));
        atomicMin(mg.vmm_at + d, ca_f2ord(fminf(fminf(smn[d][0], smn[d][1]), fminf(smn[d][2], smn[d][3]))));
        atomicMax(mg.vmm_at + 8 + d, ca_f2ord(fmaxf(fmaxf(smx[d][0], smx[d][1]), fmaxf(smx[d][2], smx[d][3]))));
      }
      if (mg.ysq.nblk) {
        mg.ysq.amax_out[2 * bx] = fmaxf(fmaxf(sma[0], sma[1]), fmaxf(sma[2], sma[3]));
        mg.ysq.amax_out[2 * bx + 1] = 0.f;
      }
    }
    CA_LAB_CP(bx, 3);
    return;
  }
  int b = bx - gblocks;
  // psi and q(z)-logit blocks: FOUR 256-cell pieces per 1024-thread block (a quarter-filled block costs the dispatcher sixteen wave slots
  // all the same: 800 of them at cfg-3 took 15 us to get through).  Piece index = what a 256-thread block's index was.
  const int sub = (int)threadIdx.x >> 8, npsi4 = (psi.nblk + 3) / 4;
  if (b < npsi4) {
    __shared__ float smw[CA_UM_TB / 64];
    const int pb = 4 * b + sub;              // 256-cell piece
    float pn = 0.f;
    if (b == 0) CA_LAB_CP(42, 0);
    if (pb < psi.nblk) {
      if (!ca_psi_adam_body_at(psi, (int64_t)pb * CA_TB + ((int)threadIdx.x & (CA_TB - 1)), 1, lr_t, b1, b2, aeps, &pn, &gt)) return;
    }
    if (mg.xpart) {   // (behind the gate: a launch that stores nothing stores nothing here either)
      __shared__ float smx2[CA_UM_TB / 64];
      float ax = fabsf(pn);
      if (!(ax == ax)) ax = INFINITY;   // (k_poly_xmax's rule: a NaN latent position shows as an unbounded range)
#pragma unroll
      for (int q = 1; q < 64; q <<= 1) ax = fmaxf(ax, __shfl_xor(ax, q, 64));
      if ((threadIdx.x & 63) == 0) smx2[threadIdx.x >> 6] = ax;
      __syncthreads();
      if ((threadIdx.x & (CA_TB - 1)) == 0 && pb < psi.nblk)
        mg.xpart[pb] = fmaxf(fmaxf(smx2[4 * sub], smx2[4 * sub + 1]), fmaxf(smx2[4 * sub + 2], smx2[4 * sub + 3]));
    }
    if (b == 0) CA_LAB_CP(42, 1);
    if (mg.ysq.nblk) {   // the psi image: one wave per 64-step (ca_ys_quant_wave), the piece's pair of maxima from its four waves
      const int wv = (int)threadIdx.x >> 6;
      const int64_t step = (int64_t)pb * (CA_TB / 64) + (wv & 3);
      const bool live = pb < psi.nblk && step < mg.ysq.NS;
      const float m = ca_ys_quant_wave(mg.ysq, live, false, live ? step : 0, pn, false);
      if ((threadIdx.x & 63) == 0) smw[wv] = m;
      __syncthreads();
      if ((threadIdx.x & (CA_TB - 1)) == 0 && pb < psi.nblk) {
        mg.ysq.amax_out[2 * (gblocks + pb)] = 0.f;
        mg.ysq.amax_out[2 * (gblocks + pb) + 1] = fmaxf(fmaxf(smw[4 * sub], smw[4 * sub + 1]), fmaxf(smw[4 * sub + 2], smw[4 * sub + 3]));
      }
    }
    if (b == 0) CA_LAB_CP(42, 2);
    return;
  }
  b -= npsi4;
  if (!ca_gate_wait(mg, false)) return;
  if (b == 0) CA_LAB_CP(43, 0);
  if (4 * b + sub < mg.ncell) ca_logit_adam_body(4 * b + sub, mg.glogit, mg.dgl, mg.m_gl, mg.v_gl, psi.N, mg.C, lr_t, b1, b2, aeps, (int)threadIdx.x & (CA_TB - 1));
  if (b == 0) CA_LAB_CP(43, 1);
}

// Column products, engine form: the sweep of ca_yt_block plus, as extra blocks of the launch, the gene side of the overflow
// list (per-chunk sums of the counts above 255; they depend on psi only).
template <int TL, int DEPTH>
__global__ void __launch_bounds__(CA_YM_TB) k_yt_mfma(const uint4* __restrict__ Yb, const uint4* __restrict__ Pq, int GT, int64_t NS,
                                                      int64_t schunk, int* __restrict__ out, int nb_main, ca_ovf_args ovf,
                                                      const float* __restrict__ F, int Df, int K) {
  if ((int)blockIdx.x >= nb_main) {
    if (blockIdx.y == 0) ca_ovf_chunks_body(blockIdx.x - nb_main, ovf.chunk_start, ovf.row2, ovf.val2, F, Df, ovf.csum, ovf.nchunk, K, 0);
    return;
  }
  ca_yt_block<TL, DEPTH>(Yb, Pq, GT, NS, schunk, out);
}
#ifdef CA_LAB   // (CA_VARX_Y_MFMA2's finisher)
// Y^T psi from the slices' digit sums: integer sum over the slices (exact), digits combined in fp64, the fixed-point scale
// taken out, the overflow list's chunk sums of the gene added.  One thread per (gene, k); red_y is [G][K].
__global__ void __launch_bounds__(CA_TB) k_yt_finish(const int* __restrict__ out /*[csplit][GT * 16][16]*/, int csplit, int GT, int G, int K,
                                                     const unsigned* __restrict__ amax, const int* __restrict__ col_chunk_ptr,
                                                     const float* __restrict__ csum, double* __restrict__ red_y) {
  const int i = blockIdx.x * CA_TB + threadIdx.x;
  if (i >= G * K) return;
  const int g = i / K, k = i - g * K;
  double v = 0.0;
#pragma unroll
  for (int p = 3; p >= 0; --p) {
    long long a = 0;
    for (int sp = 0; sp < csplit; ++sp) a += out[(((int64_t)sp * GT * 16) + g) * 16 + 4 * k + p];
    v = v * 256.0 + (double)a;
  }
  v *= ldexp(1.0, -ca_fix_exp(__uint_as_float(amax[1])));
  if (csum)
    for (int ch = col_chunk_ptr[g]; ch < col_chunk_ptr[g + 1]; ++ch) v += (double)csum[(int64_t)ch * K + k];
  red_y[i] = v;
}

#endif   // CA_LAB
// ------------------------------------------------------------------ one-shot peer-to-peer all-reduce
// Round 4: the flag travels IN the data.  Slab of a rank (fine-grained device memory, IPC-mapped by every peer):
//   inbox[parity 2][source rank W][cap entries], one entry = 16 bytes = {low half of the double, tag} {high half, tag}, each 8-byte half
//   written with ONE store (8-byte stores are single transactions on the device and over xGMI / PCIe) and read with one load; tag = the low
//   32 bits of the call's sequence number.  A receiver polls the entry itself until both tags are the call's: an entry is complete when it
//   can be read as complete -- no fence behind the data, no flag to raise after it, no arrival counter to find the last block, no barrier.
// Rounds 2-3 had data, then a system-scope release fence per block, an arrival counter, the last block's fence and flag stores, an acquire
// spin, a third fence and the loads: five dependent round trips of uncached memory and three L2 write-backs.  Measured on ONE device, a rank
// of a sharded fit at 12.5k x 5k x 8 (tools/shard_seq_time.py; profiles/r04_p2p_allreduce.txt): that kernel took 22 us of an 84 us
// iteration (63 us unsharded); ablations put 8.6 us of it on the fences and most of the rest on the chain.
// Call `seq` (1, 2, ...) uses parity seq & 1.  A rank is at most one call ahead of any peer: it can only publish seq + 1 after it has read
// every peer's seq, and a peer overwrites parity seq & 1 with seq + 2 only after it has read this rank's seq + 1 -- which this rank publishes
// after it is done reading seq.  The slab starts zeroed and sequence numbers start at 1, so tag 0 never matches.
struct ca_p2p_args {
  double* const* peers;          // [W] slab base of every rank as mapped in THIS process (own included)
  int rank, world;
  int64_t cap;
  unsigned long long seq;
  unsigned long long* err;       // pinned host word of this rank: 0, or the sequence number of the first call that gave up
  unsigned int* err_local;       // the same fact in device memory: what a call looks at when it starts (a read of the pinned word is a PCIe round trip)
  unsigned long long timeout_ticks;   // s_memrealtime ticks (100 MHz) a thread waits for a peer's entry before it gives up
  // What used to be two launches in front of the collective rides in it.  (a) entries [fold_lo, fold_lo + fold_n) of buf are
  // the column sums of the backward sweep's nslice partial slabs gpart[slice][fold_n] -- summed here in slice order (fp64) instead of
  // by a k_colsum launch; (b) entry yw_index also gets the sum of n_yw block partials (psi.(YW) of a pending monitor pass, made by
  // blocks of the backward sweep's launch).  Every rank does the same additions, so the replicas stay bit-identical.
  const float* gpart; int nslice; int64_t fold_lo, fold_n;
  const double* yw_part; int n_yw; int64_t yw_index;
};
__device__ __forceinline__ unsigned long long* ca_p2p_entry(double* slab, int64_t cap, int world, int par, int src, int64_t i) {
  return reinterpret_cast<unsigned long long*>(slab) + 2 * ((((int64_t)par * world + src) * cap) + i);
}
// The wait for the peers is BOUNDED: a dead or desynchronised peer must end in CA_ERR_COMM on the host, not in a GPU spin that
// nothing can interrupt.  A thread that has waited timeout_ticks for an entry writes the call's sequence number to the rank's error word
// (pinned host memory, checked by the host at every synchronisation point) and returns; the call's buffer is then partly summed and the
// engine is dead.  Every later call on a failed transport returns at once, so work already queued behind it drains quickly.
#ifndef CA_P2P_LAB
#define CA_P2P_LAB 0   // (timing lab of the round-3 form; unused by this one)
#endif
__global__ void __launch_bounds__(CA_TB) k_p2p_allreduce(double* __restrict__ buf, int64_t n, ca_p2p_args a) {
  __shared__ unsigned int bad;
  if (threadIdx.x == 0) bad = __hip_atomic_load(a.err_local, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
  __syncthreads();
  if (bad) return;
  const int par = (int)(a.seq & 1ull);
  const unsigned long long tag = (a.seq & 0xFFFFFFFFull) << 32;
  const int64_t i0 = (int64_t)blockIdx.x * CA_TB + threadIdx.x, stride = (int64_t)gridDim.x * CA_TB;
  // 0. (riding) the psi.(YW) partial sum for entry yw_index: by the block that owns that entry (uniform per block)
  double yw_sum = 0.0;
  if (a.yw_part && a.yw_index >= 0 && a.yw_index < n && (a.yw_index / CA_TB) % gridDim.x == blockIdx.x) {
    __shared__ double smy[CA_TB];
    double ya = 0.0;
    for (int b = threadIdx.x; b < a.n_yw; b += CA_TB) ya += a.yw_part[b];
    yw_sum = ca_block_sum(ya, smy);
  }
  double* mine = a.peers[a.rank];
  for (int64_t i = i0; i < n; i += stride) {
    // 1. my summand ...
    double v;
    if (a.gpart && i >= a.fold_lo && i < a.fold_lo + a.fold_n) {   // column sum of the sweep's slabs, slice order, 8 loads in flight
      v = 0.0;
      const float* gp = a.gpart + (i - a.fold_lo);
      for (int sp0 = 0; sp0 < a.nslice; sp0 += 8) {
        float t[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) t[q] = gp[(int64_t)(sp0 + q < a.nslice ? sp0 + q : a.nslice - 1) * a.fold_n];
#pragma unroll
        for (int q = 0; q < 8; ++q) v += (sp0 + q < a.nslice) ? (double)t[q] : 0.0;
      }
    } else {
      v = buf[i];
    }
    if (a.yw_part && i == a.yw_index) v += yw_sum;
    // ... into my inbox on every rank: two 8-byte stores per peer, each carrying its half of the double and the call's tag
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    const unsigned long long w0 = (bits & 0xFFFFFFFFull) | tag, w1 = (bits >> 32) | tag;
    for (int p = 0; p < a.world; ++p) {
      unsigned long long* e = ca_p2p_entry(a.peers[p], a.cap, a.world, par, a.rank, i);
      __hip_atomic_store(e, w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(e + 1, w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  // 2. the same W additions in the same order on every rank, each summand taken as soon as it can be read complete
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int64_t i = i0; i < n; i += stride) {
    double s = 0.0;
    for (int r = 0; r < a.world; ++r) {
      const unsigned long long* e = ca_p2p_entry(mine, a.cap, a.world, par, r, i);
      unsigned long long w0 = __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      unsigned long long w1 = __hip_atomic_load(e + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      while (((w0 ^ tag) >> 32) != 0ull || ((w1 ^ tag) >> 32) != 0ull) {
        if (__builtin_amdgcn_s_memrealtime() - t0 > a.timeout_ticks) {
          unsigned long long expect = 0ull;
          __hip_atomic_compare_exchange_strong(a.err, &expect, a.seq, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(a.err_local, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          return;
        }
        __builtin_amdgcn_s_sleep(1);
        w0 = __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        w1 = __hip_atomic_load(e + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      s += __longlong_as_double((long long)((w0 & 0xFFFFFFFFull) | (w1 << 32)));
    }
    buf[i] = s;
  }
}

// the one-copy stream with the overflow list's per-entry work (cell side, then gene side) as extra blocks, like k_ypass
__global__ void __launch_bounds__(CA_YM_TB, CA_YS_WAVES) k_ys_mfma_ovf(const uint8_t* __restrict__ Ys, ca_ys_io io, int64_t N, int Gp, int RS,
                                                                       int nb_main, ca_ovf_args ovf, const float* __restrict__ F,
                                                                       const float* __restrict__ V, int Dstride) {
  if ((int)blockIdx.x >= nb_main) {
    const int b = (int)blockIdx.x - nb_main;
    if (b < ovf.nb_rows) ca_ovf_rows_body(b, ovf.rowptr, ovf.col, ovf.val, V, Dstride, ovf.YWextra, N, 1, 0);
    else ca_ovf_chunks_body(b - ovf.nb_rows, ovf.chunk_start, ovf.row2, ovf.val2, F, Dstride, ovf.csum, ovf.nchunk, 1, 0);
    return;
  }
  extern __shared__ __attribute__((aligned(16))) unsigned char ca_ys_dyn[];
  ca_ys_mfma_body<CA_YS_DEPTH>((int)blockIdx.x, Ys, io, N, Gp, RS, ca_ys_dyn);
}

// The one-copy int8 matrix-core stream RIDING on the forward sweep's launch (round 3).  The vector stream of k_fwd_cell_mix_y
// spends 3.5 vector instructions per count in a launch whose vector pipes are full (profiles/r02_v2_sq_counters.json: 44 % of
// the merged launch's VALU instructions are the stream's), and a matrix-core instruction occupies the same issue pipe as the
// vector ALU on this part (tools/overlap_lab.hip: MFMA + 8 v_fmac = 16 + 8 x 2.3 cycles, also across waves) -- so what counts
// is issue cycles per count: 123 per KiB for the vector stream, 8 MFMAs per 4 KiB = 32 per KiB plus the LDS transit here.
// Same block mix and dispatch order as k_fwd_cell_mix_y; stream blocks are ca_ys_mfma_body's (DEPTH pieces in flight per wave),
// the overflow list's gene-side chunk blocks follow them.
struct ca_ysride_args {
  const uint8_t* Ys; ca_ys_io io;
  const float* F; const float* V; int Df;
  int Gp, RS, nb_main, nb_y;   // nb_y = nb_main + overflow-chunk blocks
  int pat_a, pat_b;
  int pers;                    // > 0: that many long-lived stream blocks lead the grid (see ca_yride_args::pers)
  ca_ovf_args ovf;
};
#ifndef CA_YS_RIDE_WAVES
#define CA_YS_RIDE_WAVES 4   // waves per SIMD the merged launch's register budget is set for (lab: 3 = 168 VGPRs, three blocks per CU)
#endif
template <int D, int TLB, int TLS, int DEPTH, bool C16 = false, bool S2F = false>
__global__ void __launch_bounds__(CA_TB, (DEPTH == 1 && !C16 && !S2F) ? CA_YS_RIDE_WAVES : 3) k_fwd_cell_mix_ys(const float* __restrict__ F, const float* __restrict__ etamax2,
                                                           const float* __restrict__ Vs, const unsigned short* __restrict__ Mq,
                                                           ca_cell_ptrs p, const float* __restrict__ alpha_u,
                                                           double* __restrict__ cell_part, int64_t N, int C, int K, int nk, int nbig,
                                                           int nf, ca_ysride_args y) {
  constexpr size_t FW = sizeof(ca_f32x4) * 4 * TLB * 64 + sizeof(double) * (CA_TB + 64);
  constexpr size_t SM = FW > (size_t)CA_YS_LDS_BYTES ? FW : (size_t)CA_YS_LDS_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[SM];
  if (p.gate) {   // (uniform: a kernel argument) queued ahead of the host's decision -- see ca_cell_ptrs::gate
    // (a plain, uniform load -- one scalar read per wave: the word was written by the PREVIOUS launch of this stream, and that launch is complete;
    //  a device-scope atomic load here went to memory from every lane of every block and cost the small shapes more than the gap it removed)
    if (*p.gate != p.gate_go) return;
  }
  int idx;
  bool sweep;
  if (y.pers > 0) {
    const int b = (int)blockIdx.x;
    sweep = b >= y.pers && b < y.pers + nf;
    idx = sweep ? b - y.pers : (b < y.pers ? b : y.nb_main + (b - y.pers - nf));
  } else {
    sweep = ca_ride_split((int)blockIdx.x, nf, y.nb_y, y.pat_a, y.pat_b, idx);
  }
  CA_LAB_BLOCK_T0();
#define CA_YS_LEAVE() CA_LAB_LEAVE(ca_ys_out)
  if (!sweep) {
    if (idx >= y.nb_main) {   // the overflow list's blocks: cell side (an extra segment of YWpart), then gene side (chunk sums)
      const int b = idx - y.nb_main;
      if (b < y.ovf.nb_rows) ca_ovf_rows_body(b, y.ovf.rowptr, y.ovf.col, y.ovf.val, y.V, y.Df, y.ovf.YWextra, N, 1, 0);
      else ca_ovf_chunks_body(b - y.ovf.nb_rows, y.ovf.chunk_start, y.ovf.row2, y.ovf.val2, y.F, y.Df, y.ovf.csum, y.ovf.nchunk, 1, 0);
      CA_YS_LEAVE();
    }
    CA_PRIO_STREAM();
    if (y.pers > 0) {
      for (int u = idx; u < y.nb_main; u += y.pers) {
        if (u != idx) __syncthreads();   // the previous unit's combine has been read by every wave before the LDS regions are reused
        ca_ys_mfma_body<DEPTH>(u, y.Ys, y.io, N, y.Gp, y.RS, smem);
      }
    } else {
      ca_ys_mfma_body<DEPTH>(idx, y.Ys, y.io, N, y.Gp, y.RS, smem);
    }
    CA_YS_LEAVE();
  }
  {
    ca_f32x4* comb = reinterpret_cast<ca_f32x4*>(smem);
    double* sm = reinterpret_cast<double*>(smem + sizeof(ca_f32x4) * 4 * TLB * 64);
    double* la = sm + CA_TB;
    ca_log_softmax_alpha(alpha_u, C, la);
    if (nbig > 0 && idx >= nbig)
      ca_fwd_cell_body<D, TLS, C16, S2F>(F, etamax2, Vs, Mq, p, cell_part, N, C, K, nk, (int64_t)nbig * (TLB * 16) + (int64_t)(idx - nbig) * (TLS * 16), idx, comb, sm, la);
    else
      ca_fwd_cell_body<D, TLB, C16, S2F>(F, etamax2, Vs, Mq, p, cell_part, N, C, K, nk, (int64_t)idx * (TLB * 16), idx, comb, sm, la);
  }
#undef CA_YS_LEAVE
  CA_LAB_LABEL(ca_ys_out);
  CA_LAB_BLOCK_END(sweep ? (nbig > 0 && idx >= nbig ? 2 : 1) : 0, idx);   // (kind 0 = stream / overflow, 1 = big, 2 = small sweep block)
}

#include "ca_fwdbal.hip.h"   // the balanced forward sweep for small problems (round 5)

// ca_k_cell.hip.h -- part of ca_kernels.hip.h (textually included there, in this order): per-cell epilogues (plain and fused), the count-matrix finishers, the fused forward sweep with its cell epilogue (k_fwd_cell*) and the riding vector stream.
// ------------------------------------------------------------------ per-cell epilogue
// Everything of R/inference-tflow.R:294-308,322,327,332-333,338-342 that is per cell, in fp64:
// log-lik ll'_nc = A_nc - s_n mean_s log Z_snc, gamma = softmax(logits), the cell's ELBO
// summands, d ELBO / d logits, coef for the backward sweep, or (mode 2) the gamma_init logits.
//   cell_part[blk][0] = sum_n [ c_n + sum_c gamma ll' + psi_n.(YW)_n ]     (EE_p_y part)
//   cell_part[blk][1] = sum_n [ sum_c gamma log alpha + Normal(psi_n;0,1) ] (E_log_p_p part)
//   cell_part[blk][2] = sum_n sum_c gamma log gamma                         (E_log_q part)
//   cell_part[blk][3+c] = sum_n gamma_nc
#define CA_MODE_ELBO 0
#define CA_MODE_TRAIN 1
#define CA_MODE_GINIT 2
__global__ void __launch_bounds__(CA_TB) k_cell(const float* __restrict__ Zpart /*[S][nchunk][gsplit][N][8]*/, const double* __restrict__ A,
                                                const double* __restrict__ cn, const double* __restrict__ s64,
                                                const float* __restrict__ etamax2, float* __restrict__ glogit,
                                                const float* __restrict__ alpha_u, const float* __restrict__ F,
                                                const float* __restrict__ YWpart, float* __restrict__ YW,
                                                float* __restrict__ coef, float* __restrict__ dgl, double* __restrict__ scratch /*[N][C]*/,
                                                double* __restrict__ cell_part, int64_t N, int C, int S, int D, int K,
                                                int gsplit, int nchunk, int nseg, int mode) {
  __shared__ double sm[CA_TB];
  __shared__ double la[256];
  // log_alpha = log_softmax(alpha_unconstr) (:255); C is small
  ca_log_softmax_alpha(alpha_u, C, la);
  __syncthreads();
  const int64_t n = (int64_t)blockIdx.x * CA_TB + threadIdx.x;
  const bool ok = n < N;
  double ee = 0.0, pr = 0.0, q = 0.0;
  double lse = 0.0, sn = 0.0, em = 0.0;
  if (ok) {
    sn = s64[n];
    em = (D > 0) ? (double)etamax2[n] * CA_LN2 : 0.0;
    double mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmax(mx, (double)glogit[n * C + c]);
    double se = 0.0;
    for (int c = 0; c < C; ++c) se += exp((double)glogit[n * C + c] - mx);
    lse = mx + log(se);
    double fbar = 0.0;
    for (int c = 0; c < C; ++c) {
      const int ch = c / CA_CW, cc = c % CA_CW;
      const double lg = (double)glogit[n * C + c] - lse;
      const double gam = exp(lg);
      double lzsum = 0.0;
      for (int s = 0; s < S; ++s) {
        double Z = 0.0;
        for (int sp = 0; sp < gsplit; ++sp)
          Z += (double)Zpart[((((int64_t)s * nchunk + ch) * gsplit + sp) * N + n) * CA_CW + cc];
        lzsum += log(Z) + em;
        if (mode == CA_MODE_TRAIN)
          coef[(((int64_t)s * nchunk + ch) * N + n) * CA_CW + cc] = (float)(-gam * sn / ((double)S * Z));
      }
      if (mode == CA_MODE_GINIT) {
        // sum over samples, no log_alpha (:338)
        scratch[n * C + c] = (double)S * A[n * C + c] - sn * lzsum;
        continue;
      }
      const double llp = A[n * C + c] - sn * lzsum / (double)S;
      const double f = llp + la[c] - lg;
      ee += gam * llp;            // unguarded like :308 (0 * -inf = NaN for an impossible clone)
      pr += gam * la[c];
      if (gam != 0.0) {           // `tf$where(gamma == 0, 0, ...)` of :333
        q += gam * lg;
        fbar += gam * f;
      } else if (!isfinite(llp)) fbar += gam * f;
      if (mode == CA_MODE_TRAIN) scratch[n * C + c] = f;
    }
    if (mode == CA_MODE_TRAIN) {
      const double* fs = scratch;
      for (int c = 0; c < C; ++c) {
        const double gam = exp((double)glogit[n * C + c] - lse);
        dgl[n * C + c] = (gam != 0.0) ? (float)(gam * (fs[n * C + c] - fbar)) : 0.f;
      }
    }
    if (mode == CA_MODE_GINIT) {
      const double* lls = scratch;
      double m2 = -INFINITY;
      for (int c = 0; c < C; ++c) m2 = fmax(m2, lls[n * C + c]);
      double s2 = 0.0;
      for (int c = 0; c < C; ++c) s2 += exp(lls[n * C + c] - m2);
      const double l2 = m2 + log(s2);
      for (int c = 0; c < C; ++c) glogit[n * C + c] = (float)(lls[n * C + c] - l2);
    } else {
      ee += cn[n];
      for (int k = 0; k < K; ++k) {
        double yw = 0.0;
        for (int sg = 0; sg < nseg; ++sg) yw += (double)YWpart[((int64_t)sg * N + n) * K + k];
        YW[n * K + k] = (float)yw;
        const double ps = (double)F[n * D + k];
        ee += ps * yw;
        pr += -0.5 * ps * ps - 0.5 * CA_LOG2PI;
      }
    }
  }
  if (mode == CA_MODE_GINIT) return;
  const int W_ = 3 + C;
  const double r0 = ca_block_sum(ee, sm);
  const double r1 = ca_block_sum(pr, sm);
  const double r2 = ca_block_sum(q, sm);
  if (threadIdx.x == 0) {
    cell_part[(int64_t)blockIdx.x * W_ + 0] = r0;
    cell_part[(int64_t)blockIdx.x * W_ + 1] = r1;
    cell_part[(int64_t)blockIdx.x * W_ + 2] = r2;
  }
  for (int c = 0; c < C; ++c) {
    const double gam = ok ? exp((double)glogit[n * C + c] - lse) : 0.0;
    const double r = ca_block_sum(gam, sm);
    if (threadIdx.x == 0) cell_part[(int64_t)blockIdx.x * W_ + 3 + c] = r;
  }
}

// Same epilogue with CP (a power of two, C <= CP <= 64) lanes per cell: one lane per (cell, clone),
// softmax / log-sum-exp reductions by xor-shuffles inside the lane group.  CA_TB / CP cells per block.
template <int CP>
__global__ void __launch_bounds__(CA_TB) k_cell_par(const float* __restrict__ Zpart, const double* __restrict__ A,
                                                    const double* __restrict__ cn, const double* __restrict__ s64,
                                                    const float* __restrict__ etamax2, float* __restrict__ glogit,
                                                    const float* __restrict__ alpha_u, const float* __restrict__ F,
                                                    const float* __restrict__ YWpart, float* __restrict__ YW,
                                                    float* __restrict__ coef, float* __restrict__ dgl,
                                                    double* __restrict__ cell_part, int64_t N, int C, int S, int D, int K,
                                                    int gsplit, int nchunk, int nseg, int mode,
                                                    unsigned short* __restrict__ coefq, int64_t N16, int zpairs = 0, int64_t q3_off = 0) {
  // zpairs (round 6): Z comes from matrix-core sweeps over PAIRS of (sample, clone chunk) slices, [pair][gsplit][N][16] with slice j = s nchunk + ch in
  // columns 8 (j & 1) .. of pair j >> 1 (k_mq_pairs / k_fwd_mfma), instead of one vector sweep per slice, [slice][gsplit][N][8]
  __shared__ double sm[CA_TB];
  __shared__ double la[64];
  constexpr int CPB = CA_TB / CP;  // cells per block
  ca_log_softmax_alpha(alpha_u, C, la);
  __syncthreads();
  const int c = threadIdx.x % CP;
  auto gmax = [](double v) {
#pragma unroll
    for (int o = CP / 2; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, CP));
    return v;
  };
  auto gsum = [](double v) {
#pragma unroll
    for (int o = CP / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, CP);
    return v;
  };
  double ee = 0.0, pr = 0.0, q = 0.0, gsumc = 0.0;   // thread-local sums over this block's cell groups (fixed order)
  const int64_t ngroups = (N + CPB - 1) / CPB;
  for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int64_t n = grp * CPB + threadIdx.x / CP;
    const bool okn = n < N, ok = okn && c < C;
    const int64_t nn = okn ? n : N - 1;
    const int cc_ = c < C ? c : C - 1;
    const double gl = ok ? (double)glogit[nn * C + cc_] : -INFINITY;
    const double mx = gmax(gl);
    const double ex = ok ? exp(gl - mx) : 0.0;
    const double se = gsum(ex);
    const double lse = mx + log(se);
    const double lg = gl - lse;
    const double gam = ok ? ex / se : 0.0;
    const double sn = s64[nn];
    const double em = (D > 0) ? (double)etamax2[nn] * CA_LN2 : 0.0;
    const int ch = cc_ / CA_CW, cc = cc_ % CA_CW;
    double lzsum = 0.0;
    for (int s = 0; s < S; ++s) {
      double Z = 0.0;
      if (zpairs) {
        const int j = s * nchunk + ch;
        for (int sp = 0; sp < gsplit; ++sp) Z += (double)Zpart[(((int64_t)(j >> 1) * gsplit + sp) * N + nn) * 16 + 8 * (j & 1) + cc];
      } else {
        for (int sp = 0; sp < gsplit; ++sp) Z += (double)Zpart[((((int64_t)s * nchunk + ch) * gsplit + sp) * N + nn) * CA_CW + cc];
      }
      lzsum += log(Z) + em;
      if (mode == CA_MODE_TRAIN && ok) {
        const float cfv = (float)(-gam * sn / ((double)S * Z));
        coef[(((int64_t)s * nchunk + ch) * N + nn) * CA_CW + cc] = cfv;
        if (coefq) {   // bf16 parts for the matrix-core backward sweep: three for up to eight clones, two per clone chunk for 9..16
          unsigned short p1, p2, p3;
          ca_split3(cfv, p1, p2, p3);
          if (nchunk >= 2 && (nchunk & 1) && ch == nchunk - 1) {
            // an odd LAST chunk stands alone: its image takes the eight-clone layout (three parts in slot groups 0..2) and the eight-clone form of the way back --
            // 106 us at cfg-3's size where the sixteen-clone form against a chunk of zeros takes 127
            unsigned short* qp = coefq + ((((int64_t)s * ((nchunk + 1) >> 1) + (ch >> 1)) * N16 + nn) * 4) * 8 + cc;
            qp[0] = p1; qp[8] = p2; qp[16] = p3;
          } else if (nchunk >= 2) {   // (slot = 2 * part + chunk of the pair, as the sixteen-lane fused epilogue writes it; one image per sample and chunk pair)
            unsigned short* qp = coefq + ((((int64_t)s * ((nchunk + 1) >> 1) + (ch >> 1)) * N16 + nn) * 4 + (ch & 1)) * 8 + cc;
            qp[0] = p1; qp[16] = p2;
            qp[q3_off] = p3;   // (the third part: same slot of the second image, k_bwd_mfma<.., C16>)
          } else {
            unsigned short* qp = coefq + (((int64_t)s * N16 + nn) * 4) * 8 + cc;
            qp[0] = p1; qp[8] = p2; qp[16] = p3;
          }
        }
      }
    }
    const double Anc = A[nn * C + cc_];
    if (mode == CA_MODE_GINIT) {
      const double ll = ok ? (double)S * Anc - sn * lzsum : -INFINITY;   // sum over samples, no log_alpha (:338)
      const double m2 = gmax(ll);
      const double l2 = m2 + log(gsum(ok ? exp(ll - m2) : 0.0));
      if (ok) glogit[nn * C + cc_] = (float)(ll - l2);
      continue;
    }
    const double llp = Anc - sn * lzsum / (double)S;
    const double f = llp + la[cc_] - lg;
    // Only the entropy term is guarded (`tf$where(gamma == 0, 0, ...)`, :333).  gamma * ll' is not (:308): an
    // impossible clone (L = 0 where y > 0 => ll' = -inf, gamma = 0) gives 0 * -inf = NaN, the reference's
    // "Initial elbo is NA".  For finite ll' a gamma that underflowed to 0 contributes exactly 0 either way.
    const bool live = ok && gam != 0.0;
    const double gf = (live || (ok && !isfinite(llp))) ? gam * f : 0.0;
    const double fbar = gsum(gf);
    if (mode == CA_MODE_TRAIN && ok) dgl[nn * C + cc_] = (float)(live || !isfinite(llp) ? gam * (f - fbar) : 0.0);
    if (ok) { ee += gam * llp; pr += gam * la[cc_]; }
    if (live) q += gam * lg;
    gsumc += gam;
    if (okn && c == 0) {
      ee += cn[nn];
      for (int k = 0; k < K; ++k) {
        double yw = 0.0;
        for (int sg = 0; sg < nseg; ++sg) yw += (double)YWpart[((int64_t)sg * N + nn) * K + k];
        YW[nn * K + k] = (float)yw;
        const double ps = (double)F[nn * D + k];
        ee += ps * yw;
        pr += -0.5 * ps * ps - 0.5 * CA_LOG2PI;
      }
    }
  }
  if (mode == CA_MODE_GINIT) return;
  const int W_ = 3 + C;
  const double r0 = ca_block_sum(ee, sm);
  const double r1 = ca_block_sum(pr, sm);
  const double r2 = ca_block_sum(q, sm);
  if (threadIdx.x == 0) {
    cell_part[(int64_t)blockIdx.x * W_ + 0] = r0;
    cell_part[(int64_t)blockIdx.x * W_ + 1] = r1;
    cell_part[(int64_t)blockIdx.x * W_ + 2] = r2;
  }
  // per-clone sums of gamma over the block's cells, fixed order
  __syncthreads();
  sm[threadIdx.x] = gsumc;
  __syncthreads();
  if (threadIdx.x < C) {
    double a = 0.0;
    for (int i = 0; i < CPB; ++i) a += sm[i * CP + threadIdx.x];
    cell_part[(int64_t)blockIdx.x * W_ + 3 + threadIdx.x] = a;
  }
}

// Cell epilogue of the FUSED sweep: one forward sweep produced Z for two eps draws of the same parameter
// state -- group A (columns [0,C): the monitor pass, `sess$run(elbo)` :403) and group B (columns [C,2C): the
// forward half of the NEXT train pass, :401).  gamma, log gamma, log alpha are shared; A yields the ELBO partials,
// B yields coef and d ELBO / d logits for the backward sweep.  S == 1, C <= 8.
// The Y stream's products are NOT touched here (psi.(YW) of the ELBO and the YW row sums come from k_yw_dot on the
// side stream), so this epilogue depends on the forward sweep only -- and can run inside it (k_fwd_cell).
// ca_cell_fused_group: the math for CA_TB / CP cells, one lane per (cell, clone); ZA / ZB are this lane's two Z values.
struct ca_cell_acc { double ee, pr, q, gsumc, eeB; };
struct ca_cell_ptrs {
  const double* A; const double* cn; const double* s64; const float* etamax2; const float* glogit; const float* F;
  float* coef; float* dgl; unsigned short* coefq;
  double* ee_partB;   // non-null: also the second draw's expected log-likelihood per block (two ELBOs from one sweep: ca_final_elbo)
  int s2;             // 1: the two column halves are the two SAMPLES of one pass (mc_samples = 2): log-likelihood from the mean of log Z,
                      //    coef for both samples (second one N x 8 floats / N16 x 32 bf16 further on, the layout the S loops use)
  int64_t N16;
  // round 4: after a merged update (k_update_merged) nobody has made the exponent bound of the new state yet: the sweep's blocks take it
  // themselves -- the range of V' (the gene blocks of the merged update leave it with one atomic min / max each: 2 D words, one
  // uniform load here instead of a per-cell load of etamax2), then sum_d max(F_nd Vmin_d, F_nd Vmax_d) exactly as k_etamax forms
  // it -- and leave it in etamax_w (= etamax2) for their own cell epilogue and for the backward sweep.  vmm_at = null: etamax2 is
  // current, read it.
  const int* vmm_at; float* etamax_w;   // vmm_at: [2][8] order-preserving ints of min / max (ca_f2ord), see k_update_merged
  // ca_run (round 4): this sweep was queued BEHIND a gated update (k_update_merged, ca_merge_args::gate) and before the host had decided.  That launch
  // is complete when this one starts, and the word its relay block left in device memory says how it went: anything but `gate_go` (stop, or
  // the host never answered) and every block of this launch returns at once -- nothing read, nothing stored.  null: an ordinary launch.
  const unsigned long long* gate; unsigned long long gate_go;
};
// Round 5: what a lane of the epilogue reads for its (cell, clone) that nothing in the sweep produces -- the q(z) logit, the library size, the
// hoisted constant A_nc.  A small sweep block (<= 32 cells: ONE pass of the epilogue) loads them BEFORE its k-loop, so that the epilogue's fp64
// chain starts from registers instead of from a round of loads behind the combine barrier (the block's CU has nothing else to hide it with).
struct ca_cell_pre { float gl; double sn, Anc; };
template <int CP, bool WR = true>   // WR = false (mc_samples = 2, four draws in one sweep): the monitor pass's pair of samples -- sums only, no coef / d logits
__device__ __forceinline__ void ca_cell_fused_group(const ca_cell_ptrs& p, const double* la, int64_t n, int64_t N, int C, int D, int K,
                                                    double ZA, double ZB, ca_cell_acc& acc, const ca_cell_pre* pre = nullptr,
                                                    float* cf_out = nullptr /* this lane's coef as stored (0 where none), for a caller that goes on with it */) {
  const int c = threadIdx.x % CP;
  auto gmax = [](double v) {
#pragma unroll
    for (int o = CP / 2; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, CP));
    return v;
  };
  auto gsum = [](double v) {
#pragma unroll
    for (int o = CP / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, CP);
    return v;
  };
  const bool okn = n < N, ok = okn && c < C;
  const int64_t nn = okn ? n : N - 1;
  const int cc = c < C ? c : C - 1;
  const double gl = ok ? (double)(pre ? pre->gl : p.glogit[nn * C + cc]) : -INFINITY;
  const double mx = gmax(gl);
  const double ex = ok ? exp(gl - mx) : 0.0;
  const double se = gsum(ex);
  const double lse = mx + log(se);
  const double lg = gl - lse;
  const double gam = ok ? ex / se : 0.0;
  const double sn = pre ? pre->sn : p.s64[nn];
  const double em = (D > 0) ? (double)p.etamax2[nn] * CA_LN2 : 0.0;
  const double Anc = pre ? pre->Anc : p.A[nn * C + cc];
  double llpA = Anc - sn * (log(ZA) + em);
  double llpB = Anc - sn * (log(ZB) + em);
  if (CP != 16 && p.s2) {   // (uniform) two samples of one pass: ll' = A - s mean_s log Z_s (:306-308), coef_s = -gamma s / (2 Z_s)
    llpA = llpB = 0.5 * (llpA + llpB);
    if (WR && ok) {
      const float c0 = (float)(-gam * sn / (2.0 * ZA)), c1 = (float)(-gam * sn / (2.0 * ZB));
      p.coef[nn * CA_CW + cc] = c0;
      p.coef[(N + nn) * CA_CW + cc] = c1;
      if (p.coefq) {
        unsigned short p1, p2, p3;
        ca_split3(c0, p1, p2, p3);
        unsigned short* qp = p.coefq + (nn * 4) * 8 + cc;
        qp[0] = p1; qp[8] = p2; qp[16] = p3;
        ca_split3(c1, p1, p2, p3);
        qp = p.coefq + ((p.N16 + nn) * 4) * 8 + cc;
        qp[0] = p1; qp[8] = p2; qp[16] = p3;
      }
    }
  } else
  if (ok) {
    const float cfv = (float)(-gam * sn / ZB);
    if (cf_out) *cf_out = cfv;
    if constexpr (CP == 16) {   // 9..16 clones: coef in clone chunks of 8 like Lb; two bf16 parts, slot = 2 * part + chunk (k_bwd_mfma<.., C16>)
      p.coef[((int64_t)(cc >> 3) * N + nn) * CA_CW + (cc & 7)] = cfv;
      if (p.coefq) {
        unsigned short p1, p2, p3;
        ca_split3(cfv, p1, p2, p3);
        unsigned short* qp = p.coefq + (nn * 4 + (cc >> 3)) * 8 + (cc & 7);
        qp[0] = p1; qp[16] = p2;
        qp[p.N16 * 32] = p3;   // (the third part: same slot of the second image, k_bwd_mfma<.., C16>)
      }
    } else {
    p.coef[nn * CA_CW + cc] = cfv;
    if (p.coefq) {
      unsigned short p1, p2, p3;
      ca_split3(cfv, p1, p2, p3);
      unsigned short* qp = p.coefq + (nn * 4) * 8 + cc;
      qp[0] = p1; qp[8] = p2; qp[16] = p3;
    }
    }
  }
  const double fB = llpB + la[cc] - lg;
  const bool live = ok && gam != 0.0;   // see k_cell_par: only the entropy term is guarded against gamma == 0
  const double gfB = (live || (ok && !isfinite(llpB))) ? gam * fB : 0.0;
  const double fbarB = gsum(gfB);
  if (WR && ok) p.dgl[nn * C + cc] = (float)(live || !isfinite(llpB) ? gam * (fB - fbarB) : 0.0);
  if (ok) { acc.ee += gam * llpA; acc.pr += gam * la[cc]; acc.eeB += gam * llpB; }
  if (live) acc.q += gam * lg;
  acc.gsumc += gam;
  if (okn && c == 0) {
    acc.ee += p.cn[nn];
    acc.eeB += p.cn[nn];
    for (int k = 0; k < K; ++k) {
      const double ps = (double)p.F[nn * D + k];
      acc.pr += -0.5 * ps * ps - 0.5 * CA_LOG2PI;
    }
  }
}
// block partials of the epilogue: cell_part[blk][0..2] and the per-clone gamma sums
template <int CP>
__device__ __forceinline__ void ca_cell_fused_finish(const ca_cell_acc& acc, double* sm, double* __restrict__ cell_part, int blk, int C,
                                                     double* __restrict__ ee_partB = nullptr) {
  constexpr int CPB = CA_TB / CP;
  const int W_ = 3 + C;
  if (ee_partB) {   // (uniform)
    const double rb = ca_block_sum(acc.eeB, sm);
    if (threadIdx.x == 0) ee_partB[blk] = rb;
  }
  double r3[3] = {acc.ee, acc.pr, acc.q};   // one pass through the block reduction (same additions as three calls, one pair of barriers)
  ca_block_sum_n<3>(r3, sm);
  if (threadIdx.x == 0) {
    cell_part[(int64_t)blk * W_ + 0] = r3[0];
    cell_part[(int64_t)blk * W_ + 1] = r3[1];
    cell_part[(int64_t)blk * W_ + 2] = r3[2];
  }
  __syncthreads();
  sm[threadIdx.x] = acc.gsumc;
  __syncthreads();
  if ((int)threadIdx.x < C) {
    double a = 0.0;
    for (int i = 0; i < CPB; ++i) a += sm[i * CP + threadIdx.x];
    cell_part[(int64_t)blk * W_ + 3 + threadIdx.x] = a;
  }
}

template <int CP>
__global__ void __launch_bounds__(CA_TB) k_cell_fused(const float* __restrict__ Zpart /*[gsplit][N][zrow]*/, int zrow, ca_cell_ptrs p,
                                                      const float* __restrict__ alpha_u, double* __restrict__ cell_part, int64_t N, int C,
                                                      int D, int K, int gsplit) {
  __shared__ double sm[CA_TB];
  __shared__ double la[64];
  constexpr int CPB = CA_TB / CP;
  ca_log_softmax_alpha(alpha_u, C, la);
  __syncthreads();
  const int c = threadIdx.x % CP;
  const int cc = c < C ? c : C - 1;
  ca_cell_acc acc = {0.0, 0.0, 0.0, 0.0, 0.0};
  const int64_t ngroups = (N + CPB - 1) / CPB;
  for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int64_t n = grp * CPB + threadIdx.x / CP;
    const int64_t nn = n < N ? n : N - 1;
    double ZA = 0.0, ZB = 0.0;
    for (int sp = 0; sp < gsplit; ++sp) {
      const float* zp = Zpart + ((int64_t)sp * N + nn) * zrow;
      ZA += (double)zp[cc];
      ZB += (double)zp[C + cc];
    }
    ca_cell_fused_group<CP>(p, la, n, N, C, D, K, ZA, ZB, acc);
  }
  ca_cell_fused_finish<CP>(acc, sm, cell_part, blockIdx.x, C);
}

// The Y stream's row products, finished on the side stream: YW[n][k] = sum over the gene strips (+ the overflow
// list's extra strip) of YWpart, and this block's share of sum_n psi_n . (YW)_n, the one ELBO term that needs them
// (part of EE_p_y; the O(K + C) body adds the block partials).  Same strip order as the sum in k_cell_par.
__global__ void __launch_bounds__(CA_TB) k_yw_dot(const float* __restrict__ YWpart, int nseg, const float* __restrict__ F, int D, int K,
                                                  int64_t N, float* __restrict__ YW, double* __restrict__ yw_part) {
  __shared__ double sm[CA_TB];
  const int64_t n = (int64_t)blockIdx.x * CA_TB + threadIdx.x;
  double a = 0.0;
  if (n < N)
    for (int k = 0; k < K; ++k) {
      double yw = 0.0;
      for (int sg = 0; sg < nseg; ++sg) yw += (double)YWpart[((int64_t)sg * N + n) * K + k];
      YW[n * K + k] = (float)yw;
      a += (double)F[n * D + k] * yw;
    }
  const double r = ca_block_sum(a, sm);
  if (threadIdx.x == 0) yw_part[blockIdx.x] = r;
}

// Both finishing steps of the Y stream in ONE launch: the column sums of its Y^T psi slab (k_colsum's arithmetic, blocks
// [0, nb_col)) and the row sums + psi.(YW) partials (k_yw_dot's, the blocks after).  Small problems pay a launch and its gap for
// each of them otherwise.  1024 threads per block like k_colsum; the row side uses the first 256 of them.
__global__ void __launch_bounds__(1024) k_yfinish(const float* __restrict__ part, double* __restrict__ out, int rows, int64_t ld, int cols,
                                                  const int* __restrict__ col_chunk_ptr, const float* __restrict__ csum, int K, int G,
                                                  int nb_col, const float* __restrict__ YWpart, int nseg, const float* __restrict__ F, int D,
                                                  int64_t N, float* __restrict__ YW, double* __restrict__ yw_part) {
  if ((int)blockIdx.x < nb_col) {
    constexpr int RL = 16;
    __shared__ double smc[RL][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (c < cols) {
      int r = ty;
      for (; r + 3 * RL < rows; r += 4 * RL) {
        const float v0 = part[(int64_t)r * ld + c], v1 = part[(int64_t)(r + RL) * ld + c];
        const float v2 = part[(int64_t)(r + 2 * RL) * ld + c], v3 = part[(int64_t)(r + 3 * RL) * ld + c];
        a0 += (double)v0; a1 += (double)v1; a2 += (double)v2; a3 += (double)v3;
      }
      for (; r < rows; r += RL) a0 += (double)part[(int64_t)r * ld + c];
      a0 += a2; a1 += a3;
      if (csum && ty == 0) {
        const int g = c / K, k = c - g * K;
        if (g < G)
          for (int ch = col_chunk_ptr[g]; ch < col_chunk_ptr[g + 1]; ++ch) a1 += (double)csum[(int64_t)ch * K + k];
      }
    }
    smc[ty][tx] = a0 + a1;
    __syncthreads();
#pragma unroll
    for (int s_ = RL / 2; s_ > 0; s_ >>= 1) {
      if (ty < s_) smc[ty][tx] += smc[ty + s_][tx];
      __syncthreads();
    }
    if (ty == 0 && c < cols) out[c] = smc[0][tx];
    return;
  }
  // row side: one block of CA_TB cells (the same partition and order as k_yw_dot)
  __shared__ double smr[CA_TB / 64];
  const int blk = (int)blockIdx.x - nb_col;
  double a = 0.0;
  if (threadIdx.x < CA_TB) {
    const int64_t n = (int64_t)blk * CA_TB + threadIdx.x;
    if (n < N)
      for (int k = 0; k < K; ++k) {
        double yw = 0.0;
        for (int sg = 0; sg < nseg; ++sg) yw += (double)YWpart[((int64_t)sg * N + n) * K + k];
        YW[n * K + k] = (float)yw;
        a += (double)F[n * D + k] * yw;
      }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) a += __shfl_xor(a, o, 64);
    if ((threadIdx.x & 63) == 0) smr[threadIdx.x >> 6] = a;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double r = smr[0];
#pragma unroll
    for (int w = 1; w < CA_TB / 64; ++w) r += smr[w];
    yw_part[blk] = r;
  }
}

// ------------------------------------------------------------------ forward sweep + cell epilogue in one kernel
// The fused two-eps sweep with NO partial slabs: a block owns 16 * TL cells for ALL genes, its four waves take every fourth
// k-step (B operand and V' straight from L2 with one k-step of prefetch -- no LDS staging to share, each wave has its own
// gene range), the four partial accumulators meet in LDS and the block goes straight on to the cell epilogue
// (ca_cell_fused_group) for its cells: no Z partials written or re-read (39 + 26 MB per pass at 100k cells), one
// launch and one inter-kernel gap less.  Sweep alone 119 us against 108 us for k_fwd_mfma (tools/fwd_mfma_lab.hip,
// "block-split"), paid back by the 42 us cell epilogue launch it replaces.  Vs must be padded to a multiple of 32 genes
// (last gene replicated, see k_final_gene / k_vprep); Mq is zero there.
// S2F (round 4, mc_samples = 2): FOUR draws in one sweep -- the operand image at Mq carries the two samples of the monitor pass in its column
// halves (as the two-sample sweep always had them), a second image behind it (the sixteen-clone kernels' second operand set: second pair of
// B operands, second set of accumulators, six MFMAs per tile and k-step on one exp and one bf16 split) the two samples of the NEXT train
// pass.  The epilogue runs the two-sample cell group twice: sums only for the monitor pair, coef / d logits for the train pair.
template <int D, int TL, bool C16 = false, bool S2F = false>
__device__ __forceinline__ void ca_fwd_cell_body(const float* __restrict__ F, const float* __restrict__ etamax2,
                                                 const float* __restrict__ Vs /*[nk * 32][D]*/,
                                                 const unsigned short* __restrict__ Mq /*[nk][2][64][8] bf16*/, const ca_cell_ptrs& p,
                                                 double* __restrict__ cell_part, int64_t N, int C, int K, int nk, int64_t cell0,
                                                 int blk, ca_f32x4* comb /*[4][TL][64]: the four waves' partial accumulators*/,
                                                 double* sm, const double* la) {
  // C16 (round 3): 9..16 clones.  The sixteen operand columns then belong to ONE draw (clones 0..15) instead of two draws of up to
  // eight clones, the epilogue works with sixteen lanes per cell, and monitor and train passes each take a sweep of their own.
  static_assert(!(C16 && S2F), "one or the other");
  constexpr bool TWO = C16 || S2F;         // two operand sets
  constexpr int CP = C16 ? 16 : 8;         // lanes per cell in the epilogue
  const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (scalar: the k-loop's bounds, branches and operand bases then are)
  float f[TL][D], em[TL];
  ca_f32x4 acc[TL];
  CA_LAB_PH(blk, 0);
  // Round 5: a small block (<= 32 cells) is alone or nearly alone on its CU and runs as ONE latency chain -- head loads, exponent bound, first
  // operands, k-loop, combine, epilogue loads, fp64 chain -- so its independent rounds of loads are issued together at the top: the first
  // k-steps' operands (below, in front of the head's loads instead of behind the exponent bound) and the epilogue's (cell, clone) operands.
  // Same values, same arithmetic, same bits.  (96-cell blocks have four or five waves per SIMD to hide these rounds, and no registers to spare.)
  constexpr bool EARLY = TL <= 2 && !TWO;
  [[maybe_unused]] ca_cell_pre cpre = {0.f, 0.0, 0.0};
  if constexpr (EARLY) {
    constexpr int CP0 = 8;
    const int lc0 = (int)threadIdx.x / CP0, c0 = (int)threadIdx.x % CP0, cc0 = c0 < C ? c0 : C - 1;
    const int64_t n0 = cell0 + (lc0 < TL * 16 ? lc0 : 0);
    const int64_t nn0 = n0 < N ? n0 : N - 1;
    cpre.gl = p.glogit[nn0 * C + cc0]; cpre.sn = p.s64[nn0]; cpre.Anc = p.A[nn0 * C + cc0];
  }
  unsigned m0, m1;   // (-1, 0) and (0, -1) as bf16 pairs, see k_fwd_mfma
  asm volatile("s_mov_b32 %0, 0x0000bf80" : "=s"(m0));
  asm volatile("s_mov_b32 %0, 0xbf800000" : "=s"(m1));
  const ca_bf16x2 neg_lo = __builtin_bit_cast(ca_bf16x2, m0), neg_hi = __builtin_bit_cast(ca_bf16x2, m1);
  const uint4* Bq = reinterpret_cast<const uint4*>(Mq);
  constexpr int NV4 = 2 * D;   // float4 per lane and k-step: V'[8 genes][D]
  // One k-step of operands in flight, in TWO register sets used alternately (the loop runs two k-steps per trip): the step at hand
  // reads its set in place while the next one's loads land in the other.  With one set the operands had to be copied out before the
  // refill was issued -- 12 moves per k-step on the issue port the sweep is bound by.
  constexpr int NS = (TL <= 2 && !TWO) ? 4 : 2;   // operand register sets (32- and 16-cell blocks: three k-steps in flight, see below)
  uint4 b1r[NS], b2r[NS];
  float4 vr[NS][NV4];
  // C16: the second draw's sixteen columns are a second pair of B operands (its image follows the first draw's) and a second set of
  // accumulators -- six MFMAs per tile and k-step on ONE exp and one bf16 split, instead of a sweep per draw
  [[maybe_unused]] uint4 b1s[2], b2s[2];
  [[maybe_unused]] ca_f32x4 accB[TL];
  if constexpr (TWO) {
#pragma unroll
    for (int t = 0; t < TL; ++t) accB[t] = (ca_f32x4){0.f, 0.f, 0.f, 0.f};
  }
  auto fetch = [&](int set, int ks) {
    const uint4* bp = Bq + (int64_t)ks * 128;
    b1r[set] = bp[lane];
    b2r[set] = bp[64 + lane];
    if constexpr (TWO) {
      const uint4* bs = Bq + ((int64_t)nk + ks) * 128;
      b1s[set] = bs[lane];
      b2s[set] = bs[64 + lane];
    }
    const float4* vp = reinterpret_cast<const float4*>(Vs + ((int64_t)ks * 32 + 8 * q) * D);
#pragma unroll
    for (int i = 0; i < NV4; ++i) vr[set][i] = vp[i];
  };
  const int nkw = nk > wv ? (nk - wv + 3) / 4 : 0;          // this wave's k-steps: wv, wv + 4, ...
  [[maybe_unused]] auto kc = [&](int i) { return wv + 4 * (i < nkw ? i : nkw - 1); };
  if constexpr (NS == 4) {   // (small blocks: the first three k-steps' operands go out NOW, beside the head's loads, not behind the exponent bound)
    if (nkw > 0) { fetch(0, kc(0)); fetch(1, kc(1)); fetch(2, kc(2)); }
  }
  float vmn[D], vmx[D];   // (merged update: range of V' over all genes)
  if (p.vmm_at) {
#pragma unroll
    for (int d = 0; d < D; ++d) { vmn[d] = ca_ord2f(p.vmm_at[d]); vmx[d] = ca_ord2f(p.vmm_at[8 + d]); }
  }
  // (all loads of the head in ONE batch, whichever way the bound comes: a branch inside the tile loop would put a round trip per tile here)
#pragma unroll
  for (int t = 0; t < TL; ++t) {
    const int64_t n = cell0 + 16 * t + j;
    const int64_t nn = n < N ? n : N - 1;
#pragma unroll
    for (int d = 0; d < D; ++d) f[t][d] = F[nn * D + d];
    em[t] = p.vmm_at ? 0.f : etamax2[nn];
    acc[t] = (ca_f32x4){0.f, 0.f, 0.f, 0.f};
  }
  if (p.vmm_at) {
#pragma unroll
    for (int t = 0; t < TL; ++t) {
      const int64_t n = cell0 + 16 * t + j;
      float e = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d) e += fmaxf(f[t][d] * vmn[d], f[t][d] * vmx[d]);   // (k_etamax's arithmetic)
      em[t] = e;
      if (wv == 0 && q == 0 && n < N) p.etamax_w[n] = e;   // for this block's epilogue (behind the barriers below) and the backward sweep
    }
  }
  CA_LAB_PH_AFTER(em[0], blk, 1);
  auto step = [&](int set) {
    const ca_bf16x8 B1 = __builtin_bit_cast(ca_bf16x8, b1r[set]), B2 = __builtin_bit_cast(ca_bf16x8, b2r[set]);
    auto vf = [&](int i) -> float { const float4& w = vr[set][i >> 2]; return (i & 3) == 0 ? w.x : (i & 3) == 1 ? w.y : (i & 3) == 2 ? w.z : w.w; };
#pragma unroll
    for (int t = 0; t < TL; ++t) {
      unsigned hi[4], lo[4];
#pragma unroll
      for (int pp = 0; pp < 4; ++pp) {
        ca_f32x2 eta = (ca_f32x2){vf((2 * pp) * D), vf((2 * pp + 1) * D)} * f[t][0] - em[t];
#pragma unroll
        for (int d = 1; d < D; ++d) eta = (ca_f32x2){vf((2 * pp) * D + d), vf((2 * pp + 1) * D + d)} * f[t][d] + eta;
        const float e0 = __builtin_amdgcn_exp2f(eta.x), e1 = __builtin_amdgcn_exp2f(eta.y);
        hi[pp] = ca_pk_bf16(e0, e1);
        const ca_bf16x2 hb = __builtin_bit_cast(ca_bf16x2, hi[pp]);
        const float r0 = __builtin_amdgcn_fdot2_f32_bf16(hb, neg_lo, e0, false);
        const float r1 = __builtin_amdgcn_fdot2_f32_bf16(hb, neg_hi, e1, false);
        lo[pp] = ca_pk_bf16(r0, r1);
      }
      const ca_bf16x8 A1 = __builtin_bit_cast(ca_bf16x8, ((uint4){hi[0], hi[1], hi[2], hi[3]}));
      const ca_bf16x8 A2 = __builtin_bit_cast(ca_bf16x8, ((uint4){lo[0], lo[1], lo[2], lo[3]}));
      ca_f32x4 a = acc[t];
      a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A2, B1, a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1, B2, a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1, B1, a, 0, 0, 0);
      acc[t] = a;
      if constexpr (TWO) {
        const ca_bf16x8 S1 = __builtin_bit_cast(ca_bf16x8, b1s[set]), S2 = __builtin_bit_cast(ca_bf16x8, b2s[set]);
        ca_f32x4 b = accB[t];
        b = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A2, S1, b, 0, 0, 0);
        b = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1, S2, b, 0, 0, 0);
        b = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1, S1, b, 0, 0, 0);
        accB[t] = b;
      }
    }
  };
  // The loop body is ONE basic block: a pair of k-steps, no branch inside (the last pair and the odd last k-step are peeled, so
  // every load is consumed; the priority changes between four loops instead of inside one).  With branches in the body the
  // compiler's wait-count pass met the loop header with loads outstanding from several paths and waited for ALL of them at the top of
  // every k-step (s_waitcnt vmcnt(0) ... vmcnt(2) where vmcnt(4) would do) -- the operands fetched one step earlier were then waited for
  // right away.  With four or five waves per SIMD (cfg-3) others fill that; a small shard's one or two waves ran every k-step at the L2's
  // latency: 1370 cycles against 420 of issue (profiles/r03_ab_ystream.txt section 16).
  if constexpr (NS == 4) {
    // Small blocks compute 0.2 us per k-step, a third of an L2 round trip: THREE k-steps of operands in flight, four register sets
    // in rotation, four k-steps per trip of a branch-free loop.  Refills past the end re-read the last k-step (never used); the
    // explicit wait behind the loop makes sure they have landed before their registers mean anything else.
    const int ntrip = nkw >> 2;   // (the first three k-steps' operands were requested in front of the head)
    int ti = 0;
#if CA_PROG_PRIO
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const int tend = qd == 3 ? ntrip : (ntrip * (qd + 1)) / 4;
      if (qd == 0) __builtin_amdgcn_s_setprio(3);
      else if (qd == 1) __builtin_amdgcn_s_setprio(2);
      else if (qd == 2) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
#else
    {
      const int tend = ntrip;
#endif
      for (; ti < tend; ++ti) {
        const int i0 = 4 * ti;
        fetch(3, kc(i0 + 3)); __builtin_amdgcn_sched_barrier(0); step(0); __builtin_amdgcn_sched_barrier(0);
        fetch(0, kc(i0 + 4)); __builtin_amdgcn_sched_barrier(0); step(1); __builtin_amdgcn_sched_barrier(0);
        fetch(1, kc(i0 + 5)); __builtin_amdgcn_sched_barrier(0); step(2); __builtin_amdgcn_sched_barrier(0);
        fetch(2, kc(i0 + 6)); __builtin_amdgcn_sched_barrier(0); step(3); __builtin_amdgcn_sched_barrier(0);
      }
    }
    const int rem = nkw - 4 * ntrip;   // 0 .. 3 k-steps left, their operands in sets 0, 1, 2
    if (rem > 0) step(0);
    if (rem > 1) step(1);
    if (rem > 2) step(2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
  const int npair = nkw >> 1, nloop = npair > 0 ? npair - 1 : 0;   // pairs in the loops; the last pair follows them
  auto kof = [&](int i) { return wv + 4 * i; };
  if (nkw > 0) fetch(0, kof(0));
  int pi = 0;
#if CA_PROG_PRIO
#pragma unroll
  for (int qd = 0; qd < 4; ++qd) {
    const int pend = qd == 3 ? nloop : (nloop * (qd + 1)) / 4;
    if (qd == 0) __builtin_amdgcn_s_setprio(3);
    else if (qd == 1) __builtin_amdgcn_s_setprio(2);
    else if (qd == 2) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
#else
  {
    const int pend = nloop;
#endif
    for (; pi < pend; ++pi) {
      fetch(1, kof(2 * pi + 1));
      __builtin_amdgcn_sched_barrier(0);   // (the loads stay IN FRONT of the k-step they run beside: the scheduler otherwise sinks them
      step(0);                             //  to their first use, which is the end of a prefetch)
      __builtin_amdgcn_sched_barrier(0);
      fetch(0, kof(2 * pi + 2));
      __builtin_amdgcn_sched_barrier(0);
      step(1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (npair > 0) {   // the last pair: its refill only if an odd k-step follows
    fetch(1, kof(2 * nloop + 1));
    __builtin_amdgcn_sched_barrier(0);
    step(0);
    __builtin_amdgcn_sched_barrier(0);
    if (nkw & 1) fetch(0, kof(nkw - 1));
    __builtin_amdgcn_sched_barrier(0);
    step(1);
  }
  if (nkw & 1) step(0);
  }
  CA_PRIO_DONE();
  CA_LAB_PH_AFTER(acc[0][0], blk, 2);
#pragma unroll
  for (int t = 0; t < TL; ++t) comb[(wv * TL + t) * 64 + lane] = acc[t];
  __syncthreads();
  CA_LAB_PH(blk, 3);
  // ---- cell epilogue for the block's cells; Z[cell][column] = sum over the four waves of comb[w][tile][16 q + column][r]
  //      with cell = 16 tile + 4 q + r (accumulator layout of the 16x16 MFMA)
  constexpr int CPB = CA_TB / CP;
  const int c = threadIdx.x % CP;
  const int cc = c < C ? c : C - 1;
  ca_cell_acc cacc = {0.0, 0.0, 0.0, 0.0, 0.0};
  // C16: the combine buffer holds one draw's accumulators at a time -- the first draw's Z go to registers (TL values per thread:
  // sixteen cells per pass of the block), then the second draw's accumulators take the buffer
  [[maybe_unused]] double ZAr[TL];
  if constexpr (C16) {
    static_assert(CPB == 16, "one 16-cell tile per pass of the block");
#pragma unroll
    for (int t = 0; t < TL; ++t) {
      const int row = (int)threadIdx.x / CP, qq = row >> 2, r = row & 3, la_ = 16 * qq + cc;
      auto cz = [&](int w) { return (double)comb[(w * TL + t) * 64 + la_][r]; };
      ZAr[t] = (cz(0) + cz(1)) + (cz(2) + cz(3));
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < TL; ++t) comb[(wv * TL + t) * 64 + lane] = accB[t];
    __syncthreads();
  }
  auto cells = [&](int g0, double ZA16) {
    const int lc = g0 + (int)threadIdx.x / CP;       // local cell
    const bool inb = lc < TL * 16;
    const int lcc = inb ? lc : 0;
    const int t = lcc >> 4, row = lcc & 15, qq = row >> 2, r = row & 3;
    const int la_ = 16 * qq + cc, lb_ = C16 ? la_ : 16 * qq + C + cc;
    auto cz = [&](int w, int col) { return (double)comb[(w * TL + t) * 64 + col][r]; };
    const double ZA = C16 ? ZA16 : (cz(0, la_) + cz(1, la_)) + (cz(2, la_) + cz(3, la_));
    const double ZB = (cz(0, lb_) + cz(1, lb_)) + (cz(2, lb_) + cz(3, lb_));
    if constexpr (EARLY) ca_cell_fused_group<CP>(p, la, inb ? cell0 + lc : N, N, C, D, K, ZA, ZB, cacc, &cpre);   // (one pass: g0 == 0)
    else ca_cell_fused_group<CP>(p, la, inb ? cell0 + lc : N, N, C, D, K, ZA, ZB, cacc);
  };
  if constexpr (C16) {
#pragma unroll
    for (int t = 0; t < TL; ++t) cells(16 * t, ZAr[t]);   // (compile-time index into the registers)
  } else if constexpr (S2F) {
    // the monitor pair's Z (first operand set) out of the combine buffer into registers, then the train pair's accumulators take the buffer
    constexpr int NP = (TL * 16 + CPB - 1) / CPB;
    double Z1a[NP], Z1b[NP];
    auto zof = [&](int g0, double& za, double& zb) {
      const int lc = g0 + (int)threadIdx.x / CP;
      const int lcc = lc < TL * 16 ? lc : 0;
      const int t = lcc >> 4, row = lcc & 15, qq = row >> 2, r = row & 3;
      const int la_ = 16 * qq + cc, lb_ = 16 * qq + C + cc;
      auto cz = [&](int w, int col) { return (double)comb[(w * TL + t) * 64 + col][r]; };
      za = (cz(0, la_) + cz(1, la_)) + (cz(2, la_) + cz(3, la_));
      zb = (cz(0, lb_) + cz(1, lb_)) + (cz(2, lb_) + cz(3, lb_));
    };
#pragma unroll
    for (int i = 0; i < NP; ++i) zof(i * CPB, Z1a[i], Z1b[i]);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < TL; ++t) comb[(wv * TL + t) * 64 + lane] = accB[t];
    __syncthreads();
    ca_cell_acc scratch = {0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int lc = i * CPB + (int)threadIdx.x / CP;
      const int64_t n = lc < TL * 16 ? cell0 + lc : N;
      ca_cell_fused_group<CP, false>(p, la, n, N, C, D, K, Z1a[i], Z1b[i], cacc);     // monitor pass: the sums of its ELBO
      double za, zb;
      zof(i * CPB, za, zb);
      ca_cell_fused_group<CP, true>(p, la, n, N, C, D, K, za, zb, scratch);           // next train pass: coef for both samples, d logits
    }
  } else {
    for (int g0 = 0; g0 < TL * 16; g0 += CPB) cells(g0, 0.0);
  }
  ca_cell_fused_finish<CP>(cacc, sm, cell_part, blk, C, p.ee_partB);
}

template <int D, int TL, bool C16 = false, bool S2F = false>
__global__ void __launch_bounds__(CA_TB) k_fwd_cell(const float* __restrict__ F, const float* __restrict__ etamax2,
                                                    const float* __restrict__ Vs, const unsigned short* __restrict__ Mq, ca_cell_ptrs p,
                                                    const float* __restrict__ alpha_u, double* __restrict__ cell_part, int64_t N,
                                                    int C, int K, int nk) {
  __shared__ ca_f32x4 comb[4 * TL * 64];
  __shared__ double sm[CA_TB];
  __shared__ double la[64];
  ca_log_softmax_alpha(alpha_u, C, la);    // wave 0; visible to all after the body's barrier
  ca_fwd_cell_body<D, TL, C16, S2F>(F, etamax2, Vs, Mq, p, cell_part, N, C, K, nk, (int64_t)blockIdx.x * (TL * 16), blockIdx.x, comb, sm, la);
}

// Two block sizes in one launch: the first `nbig` blocks (one resident round: CUs x blocks per CU) own 16 * TLB cells each, the
// rest of the cells go out in blocks of 16 * TLS.  Blocks are dispatched in index order, so the small ones fill the slots the
// big ones free: the ragged end of the kernel -- CUs left with one wave per SIMD, or none, while the last big blocks finish --
// shrinks from one big block's duration to one small block's.  (Small blocks everywhere would re-read the B operand from L2
// three times as often: 64-cell blocks lose 5 % to 96-cell blocks at 100k cells.)
template <int D, int TLB, int TLS, bool C16 = false, bool S2F = false>
__global__ void __launch_bounds__(CA_TB) k_fwd_cell_mix(const float* __restrict__ F, const float* __restrict__ etamax2,
                                                        const float* __restrict__ Vs, const unsigned short* __restrict__ Mq,
                                                        ca_cell_ptrs p, const float* __restrict__ alpha_u,
                                                        double* __restrict__ cell_part, int64_t N, int C, int K, int nk, int nbig) {
  __shared__ ca_f32x4 comb[4 * TLB * 64];
  __shared__ double sm[CA_TB];
  __shared__ double la[64];
  ca_log_softmax_alpha(alpha_u, C, la);
  if ((int)blockIdx.x < nbig)
    ca_fwd_cell_body<D, TLB, C16, S2F>(F, etamax2, Vs, Mq, p, cell_part, N, C, K, nk, (int64_t)blockIdx.x * (TLB * 16), blockIdx.x, comb, sm, la);
  else
    ca_fwd_cell_body<D, TLS, C16, S2F>(F, etamax2, Vs, Mq, p, cell_part, N, C, K, nk,
                                       (int64_t)nbig * (TLB * 16) + (int64_t)((int)blockIdx.x - nbig) * (TLS * 16), blockIdx.x, comb, sm, la);
}


// The Y stream RIDING on the forward sweep's launch (u8 storage, K = 1): the blocks of k_ypass and the blocks of the sweep are
// interleaved in one grid, so both are resident side by side from the first microsecond and no second queue, no cross-stream
// event and none of the ~6 us dispatch gaps that each of those costs is involved (profiles/r02_v1_gaps.txt: 33 us of gaps per
// iteration with the side stream).  Block b: even -> sweep block b / 2, odd -> stream block b / 2, until one kind runs out.
struct ca_yride_args {
  const uint8_t* Y; const float* F; int Dstride; const float* V; float* YWpart; float* YTpart;
  int G, Gp, nseg, nrb, TR, nb_main, nb_y;   // nb_y = nb_main + overflow-list blocks
  int pat_a, pat_b;                          // interleave: pat_a sweep blocks, then pat_b stream blocks, ...
  int pers;                                  // > 0: that many LONG-LIVED stream blocks lead the grid, block s takes units s, s + pers, ...
  ca_ovf_args ovf;
};
// true: sweep block idx, false: stream block idx.  Periods of pa sweep blocks followed by pb stream blocks while both kinds last,
// then the sweep's remainder, then the stream's.
__device__ __forceinline__ bool ca_ride_split(int b, int nf, int ny, int pa, int pb, int& idx) {
  const int per = pa + pb;
  const int m = (nf / pa) < (ny / pb) ? (nf / pa) : (ny / pb);
  if (b < m * per) {
    const int p = b / per, r = b - p * per;
    if (r < pa) { idx = p * pa + r; return true; }
    idx = p * pb + (r - pa);
    return false;
  }
  const int t = b - m * per, restf = nf - m * pa;
  if (t < restf) { idx = m * pa + t; return true; }
  idx = m * pb + (t - restf);
  return false;
}
#ifndef CA_RIDE_WAVES
#define CA_RIDE_WAVES 1   // (lab: minimum waves per SIMD the merged launch's register budget is set for)
#endif
template <int D, int TLB, int TLS>
__global__ void __launch_bounds__(CA_TB, CA_RIDE_WAVES) k_fwd_cell_mix_y(const float* __restrict__ F, const float* __restrict__ etamax2,
                                                          const float* __restrict__ Vs, const unsigned short* __restrict__ Mq,
                                                          ca_cell_ptrs p, const float* __restrict__ alpha_u,
                                                          double* __restrict__ cell_part, int64_t N, int C, int K, int nk, int nbig,
                                                          int nf, ca_yride_args y) {
  constexpr size_t FW = sizeof(ca_f32x4) * 4 * TLB * 64 + sizeof(double) * (CA_TB + 64);
  constexpr size_t YW_ = sizeof(float) * (CA_TB / 64) * 64 * 17;
  __shared__ __attribute__((aligned(16))) unsigned char smem[FW > YW_ ? FW : YW_];
  int idx;
  CA_LAB_BLOCK_T0();
  bool sweep;
  if (y.pers > 0) {
    // Long-lived stream blocks first: y.pers of them (two per CU) take the leading slots and walk through ALL units of the count
    // matrix, so the stream holds the same share of every CU's slots for as long as it lasts -- with stream blocks of one unit
    // the slots they free go to whatever comes next in the grid, mostly sweep blocks, and the CUs end up with unequal numbers of
    // those (tools/stamps.py).  The sweep's blocks follow, then the overflow list's.
    const int b = (int)blockIdx.x;
    sweep = b >= y.pers && b < y.pers + nf;
    idx = sweep ? b - y.pers : (b < y.pers ? b : y.nb_main + (b - y.pers - nf));
  } else {
    sweep = ca_ride_split((int)blockIdx.x, nf, y.nb_y, y.pat_a, y.pat_b, idx);
  }
  if (!sweep) {
    CA_PRIO_STREAM();
    if (y.pers > 0 && idx < y.pers) {
      for (int u = idx; u < y.nb_main; u += y.pers)
        ca_ypass_body<uint8_t, 1, 0>(u, y.Y, y.F, y.Dstride, y.V, 0, y.YWpart, y.YTpart, N, y.G, y.Gp, y.nseg, y.nrb, y.TR, 1, y.ovf, y.nb_main,
                                     reinterpret_cast<float (*)[64][17]>(smem));
    } else
    ca_ypass_body<uint8_t, 1, 0>(idx, y.Y, y.F, y.Dstride, y.V, 0, y.YWpart, y.YTpart, N, y.G, y.Gp, y.nseg, y.nrb, y.TR, 1, y.ovf, y.nb_main,
                                 reinterpret_cast<float (*)[64][17]>(smem));
  } else {
    ca_f32x4* comb = reinterpret_cast<ca_f32x4*>(smem);
    double* sm = reinterpret_cast<double*>(smem + sizeof(ca_f32x4) * 4 * TLB * 64);
    double* la = sm + CA_TB;
    ca_log_softmax_alpha(alpha_u, C, la);
    if (nbig > 0 && idx >= nbig)
      ca_fwd_cell_body<D, TLS>(F, etamax2, Vs, Mq, p, cell_part, N, C, K, nk, (int64_t)nbig * (TLB * 16) + (int64_t)(idx - nbig) * (TLS * 16), idx, comb, sm, la);
    else
      ca_fwd_cell_body<D, TLB>(F, etamax2, Vs, Mq, p, cell_part, N, C, K, nk, (int64_t)idx * (TLB * 16), idx, comb, sm, la);
  }
  CA_LAB_BLOCK_END(sweep ? (nbig > 0 && idx >= nbig ? 2 : 1) : 0, idx);
}

// The Y stream FUSED IN SEQUENCE with the sweep (round 3): every sweep block also streams one unit of the count matrix (one gene
// segment x four row blocks, what a k_ypass block does), either before or after its sweep.  Block timelines of the interleaved
// form (tools/stamps.py, profiles/r03_ab_ystream.txt) show why: everything resident on a CU -- sweep and stream blocks alike --
// ends when that CU's vector work is done, CUs that drew two, three or four sweep blocks at the start end at 42, 62 and 83 us,
// a stream block needs 83 us instead of the 40 it takes alone, and the launch ends when the last stragglers have gone through.
// Here every block carries the same work, so every CU carries the same work, and at any time about half the blocks of a CU are
// in their (latency-bound) stream phase while the other half has the vector pipes: which half goes first alternates along the
// XCD's own block sequence, whichever way the dispatcher deals that sequence over the CUs (i = b / 8: i ^ (i >> 5)).
// Blocks past the sweep's own: leftover stream units (small shards have more units than sweep blocks), then the overflow list's.
template <int D, int TLB, int TLS>
__global__ void __launch_bounds__(CA_TB, CA_RIDE_WAVES) k_fwd_cell_seq_y(const float* __restrict__ F, const float* __restrict__ etamax2,
                                                                         const float* __restrict__ Vs, const unsigned short* __restrict__ Mq,
                                                                         ca_cell_ptrs p, const float* __restrict__ alpha_u,
                                                                         double* __restrict__ cell_part, int64_t N, int C, int K, int nk, int nbig,
                                                                         int nf, ca_yride_args y) {
  constexpr size_t FW = sizeof(ca_f32x4) * 4 * TLB * 64 + sizeof(double) * (CA_TB + 64);
  constexpr size_t YW_ = sizeof(float) * (CA_TB / 64) * 64 * 17;
  __shared__ __attribute__((aligned(16))) unsigned char smem[FW > YW_ ? FW : YW_];
  const int b = (int)blockIdx.x;
  CA_LAB_BLOCK_T0();
  const bool sweep_blk = b < nf;
  int unit;
  bool first = false;
  if (!sweep_blk) {                                // stream-only blocks
    const int e = b - nf, rest = y.nb_main > nf ? y.nb_main - nf : 0;
    unit = e < rest ? nf + e : y.nb_main + (e - rest);
  } else {
    const int i = b >> 3;
    unit = b < y.nb_main ? b : -1;
    first = ((i ^ (i >> 5)) & 1) != 0;
  }
  if (unit >= 0 && (!sweep_blk || first)) {
    CA_PRIO_STREAM();
    ca_ypass_body<uint8_t, 1, 0>(unit, y.Y, y.F, y.Dstride, y.V, 0, y.YWpart, y.YTpart, N, y.G, y.Gp, y.nseg, y.nrb, y.TR, 1, y.ovf, y.nb_main,
                                 reinterpret_cast<float (*)[64][17]>(smem));
    if (sweep_blk) __syncthreads();
  }
  if (sweep_blk) {
    ca_f32x4* comb = reinterpret_cast<ca_f32x4*>(smem);
    double* sm = reinterpret_cast<double*>(smem + sizeof(ca_f32x4) * 4 * TLB * 64);
    double* la = sm + CA_TB;
    ca_log_softmax_alpha(alpha_u, C, la);
    if (nbig > 0 && b >= nbig)
      ca_fwd_cell_body<D, TLS>(F, etamax2, Vs, Mq, p, cell_part, N, C, K, nk, (int64_t)nbig * (TLB * 16) + (int64_t)(b - nbig) * (TLS * 16), b, comb, sm, la);
    else
      ca_fwd_cell_body<D, TLB>(F, etamax2, Vs, Mq, p, cell_part, N, C, K, nk, (int64_t)b * (TLB * 16), b, comb, sm, la);
    if (unit >= 0 && !first) {
      __syncthreads();
      CA_PRIO_STREAM();
      ca_ypass_body<uint8_t, 1, 0>(unit, y.Y, y.F, y.Dstride, y.V, 0, y.YWpart, y.YTpart, N, y.G, y.Gp, y.nseg, y.nrb, y.TR, 1, y.ovf, y.nb_main,
                                   reinterpret_cast<float (*)[64][17]>(smem));
    }
  }
  CA_LAB_BLOCK_END(b >= nf ? 0 : (nbig > 0 && b >= nbig ? 2 : 1), b);
}


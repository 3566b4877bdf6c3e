// ca_k_bwd.hip.h -- part of ca_kernels.hip.h (textually included there, in this order): backward sweeps (k_bwd, k_bwd_mfma), TF1 Adam, the O(K + C) ELBO assembly body, preprocessing and allele kernels.
// ------------------------------------------------------------------ backward sweep
// Reverse mode of Z = E.M given coef = dELBO/dZ.  lane = gene (RG genes per lane), loop over a
// slice of cells whose coef/F/etamax are wave-uniform (scalar loads).  Ablations (tools/bwd_lab2.hip, 100k x 5k x 8,
// RG = 4: 280 us): the t contraction 82 us, the per-cell wave reduction 60 us, loop skeleton + operand fetch 99 us,
// v_exp_f32 ~5 us; vector-fetch + v_readlane, LDS staging and software prefetch of the operands all land within
// +-10 %, RG = 8 gains 14 %:
//   t_ng  = sum_c coef_nc L_gc          u_ng = E_ng t_ng
//   gpart[split][g][s]     += sum_n u_ng                         (-> d/d mu_sg)
//   gpart[split][g][S + d] += mu_g sum_n u_ng F_nd               (-> d/d V_gd)
//   dFpart[tile][n][d]     += sum_{g in tile} mu_g u_ng V_gd     (-> d/d F_nd), DPP wave reduction
template <int NC, int D, int RG>
__global__ void __launch_bounds__(CA_TB) k_bwd(const float* __restrict__ coef /*[N][8]*/, const float* __restrict__ F,
                                               const float* __restrict__ etamax2, const float* __restrict__ Lb /*[G][8]*/,
                                               const float* __restrict__ mu /*[G]*/, const float* __restrict__ Vs,
                                               const float* __restrict__ V, float* __restrict__ gpart /*[csplit][G][S+Dn]*/,
                                               float* __restrict__ dFpart /*[ntile][N][Dn]*/, int64_t N, int G,
                                               int64_t cchunk, int Drt, int S, int sidx, int first_s, int first) {
  constexpr int DM = (D < 0) ? 8 : (D > 0 ? D : 1);
  const int Dn = (D < 0) ? Drt : D;
  const int lane = threadIdx.x & 63;
  const int tile = blockIdx.x * (CA_TB / 64) + (threadIdx.x >> 6);
  const int gbase = tile * 64 * RG;
  if (gbase >= G) return;
  float l[RG][NC], m_[RG], vs[RG][DM], v[RG][DM], accU[RG], accUF[RG][DM];
#pragma unroll
  for (int r = 0; r < RG; ++r) {
    const int g = gbase + r * 64 + lane;
    const bool ok = g < G;
    const int gg = ok ? g : G - 1;
#pragma unroll
    for (int c = 0; c < NC; ++c) l[r][c] = ok ? Lb[(int64_t)gg * CA_CW + c] : 0.f;
    m_[r] = ok ? mu[gg] : 0.f;
#pragma unroll
    for (int d = 0; d < DM; ++d) {
      vs[r][d] = (ok && d < Dn) ? Vs[(int64_t)gg * Dn + d] : 0.f;
      v[r][d] = (ok && d < Dn) ? V[(int64_t)gg * Dn + d] : 0.f;
      accUF[r][d] = 0.f;
    }
    accU[r] = 0.f;
  }
  const int64_t n0 = (int64_t)blockIdx.y * cchunk;
  const int64_t n1 = (n0 + cchunk < N) ? n0 + cchunk : N;
  float keepF[DM];
#pragma unroll
  for (int d = 0; d < DM; ++d) keepF[d] = 0.f;
  for (int64_t n = n0; n < n1; ++n) {
    float cf[NC], f[DM];
#pragma unroll
    for (int c = 0; c < NC; ++c) cf[c] = coef[n * CA_CW + c];
#pragma unroll
    for (int d = 0; d < DM; ++d) f[d] = (d < Dn) ? F[n * Dn + d] : 0.f;
    const float em = (Dn > 0) ? etamax2[n] : 0.f;
    float dsum[DM];
#pragma unroll
    for (int d = 0; d < DM; ++d) dsum[d] = 0.f;
#pragma unroll
    for (int r = 0; r < RG; ++r) {
      float e = 1.f;
      if (Dn > 0) {
        float eta = -em;
#pragma unroll
        for (int d = 0; d < DM; ++d) eta = fmaf(f[d], vs[r][d], eta);
        e = __builtin_amdgcn_exp2f(eta);
      }
      float t = 0.f;
#pragma unroll
      for (int c = 0; c < NC; ++c) t = fmaf(cf[c], l[r][c], t);
      const float u = e * t;
      accU[r] += u;
      const float deta = m_[r] * u;
#pragma unroll
      for (int d = 0; d < DM; ++d) {
        accUF[r][d] = fmaf(u, f[d], accUF[r][d]);
        dsum[d] = fmaf(deta, v[r][d], dsum[d]);
      }
    }
    const int slot = (int)(n - n0) & 63;
#pragma unroll
    for (int d = 0; d < DM; ++d) {
      if (d < Dn) {
        const float tot = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ca_wave_sum_lane63(dsum[d])), 63));
        keepF[d] = (lane == slot) ? tot : keepF[d];
      }
    }
    if (slot == 63 || n + 1 == n1) {   // wave-uniform: write the last (up to 64) cells' totals, one cell per lane
      const int64_t fb = n - slot;
      if (fb + lane <= n) {
#pragma unroll
        for (int d = 0; d < DM; ++d)
          if (d < Dn) {
            float* p = dFpart + ((int64_t)tile * N + fb + lane) * Dn + d;
            *p = first ? keepF[d] : (*p + keepF[d]);
          }
      }
    }
  }
  const int W_ = S + Dn;
#pragma unroll
  for (int r = 0; r < RG; ++r) {
    const int g = gbase + r * 64 + lane;
    if (g < G) {
      float* gp = gpart + ((int64_t)blockIdx.y * G + g) * W_;
      gp[sidx] = first_s ? accU[r] : gp[sidx] + accU[r];
#pragma unroll
      for (int d = 0; d < DM; ++d)
        if (d < Dn) {
          const float val = m_[r] * accUF[r][d];
          gp[S + d] = first ? val : gp[S + d] + val;
        }
    }
  }
}

// TF1 Adam (tf.train.AdamOptimizer, R/inference-tflow.R:345): epsilon outside the bias correction
__device__ __forceinline__ void ca_adam(float& th, float& m, float& v, float g, float lr_t, float b1, float b2, float eps) {
  m = b1 * m + (1.f - b1) * g;
  v = b2 * v + (1.f - b2) * g * g;
  th = th - lr_t * m / (sqrtf(v) + eps);
}

// ------------------------------------------------------------------ ELBO assembly + the O(K + C) variables
// red[0..2] cell sums (all-reduced when sharded), red[3..3+C) sum_n gamma_nc; gene_part block partials.
// Uses W^2 sums taken BEFORE this step's Adam update of W (k_gene_pre), as autodiff does.
// One 256-thread block runs the body: as its own launch (k_final_small), or as an extra block of a kernel it does not
// depend on -- the monitor pass's ELBO assembly rides on the backward sweep, the train pass's chi / alpha update on
// the per-cell Adam kernel -- so the fp64 exp/log chains of this single wave are off the critical path.
struct ca_small_args {
  int enabled;
  double* red; const double* gene_part; int ngblk;
  float *vchi, *alpha_u, *m_v, *v_v, *m_a, *v_a, *g_v, *g_a;
  double *elbo_out, *terms_out;
  int G, C, K, apply;
  float lr_t, b1, b2, aeps;
  const float* vmm_part; float* vmm; int D;
  double dir_const;
  const double* cell_part; int ncblk;   // when set: first reduce the cell epilogue's block partials into red[0 .. 3 + C)
  double* host_out; unsigned long long* host_flag; unsigned long long host_seq;   // ELBO mirrored into pinned host memory (ca_run)
  int reduce_only;          // stop after the cell-partial reduction (sharded: the sums are all-reduced before the ELBO assembly)
  const double* yw_part; int n_yw;      // with cell_part: block partials of sum_n psi_n.(YW)_n (k_yw_dot), added to red[0]
  const double* ee_part; int n_ee;      // without cell_part: block partials of the OTHER draw's EE_p_y cell sum (pair sweep), replace red[0]
  float *vchi_out, *alpha_out;          // round 4 (k_update_merged): the stepped chi / alpha go HERE (null: in place) -- the gene blocks and the monitor
                                        // block of the same launch still read the values the gradients were taken at; the host swaps the buffers
};

// wave 0 of the O(K + C) body: one lane per clone / latent dimension
// operands of wave 0 that nothing in the body produces: loaded at the body's entry, so that their latency is behind the block
// reductions in front of wave 0's own fp64 chains (4 of this block's 7 us, tools/stamps_small.py)
struct ca_small_pre { float au, vch, m_v, v_v, m_a, v_a; };
__device__ __forceinline__ ca_small_pre ca_final_small_preload(const ca_small_args& sa) {
  ca_small_pre q = {-INFINITY, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int c = threadIdx.x;
  if (c < 64) {
    if (c < sa.C && sa.C <= 64) { q.au = sa.alpha_u[c]; if (sa.apply) { q.m_a = sa.m_a[c]; q.v_a = sa.v_a[c]; } }
    if (c < sa.K) { q.vch = sa.vchi[c]; if (sa.apply) { q.m_v = sa.m_v[c]; q.v_v = sa.v_v[c]; } }
  }
  return q;
}
__device__ __forceinline__ void ca_final_small_wave0(const ca_small_args& sa, const double* gs, const ca_small_pre& pq) {
  // One lane per clone (and per latent dimension): the fp64 exp/log chains of this kernel are long, so they
  // run side by side in wave 0 and meet through xor-shuffles.  (C <= 64 here; larger C takes the loop form.)
  const int c = threadIdx.x;
  const double conc = 1.0 / (double)sa.C;
  auto wsum = [](double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
  };
  auto wmax = [](double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
  };
  double dir_sum, dla_c = 0.0, al_c = 0.0, dla_sum;
  if (sa.C <= 64) {
    const double au = c < sa.C ? (double)pq.au : -INFINITY;
    const double mx = wmax(au);
    const double se = wsum(c < sa.C ? exp(au - mx) : 0.0);
    const double lse = mx + log(se);
    al_c = c < sa.C ? exp(au - lse) : 0.0;
    dla_c = c < sa.C ? sa.red[3 + c] + (conc - 1.0) * al_c / (al_c + 1e-3) : 0.0;
    dir_sum = wsum(c < sa.C ? (conc - 1.0) * log(al_c + 1e-3) : 0.0);   // Dirichlet(1/C) log-pdf at alpha + 1e-3 (:324)
    dla_sum = wsum(dla_c);
  } else {
    double mx = -INFINITY, se = 0.0;
    for (int j = 0; j < sa.C; ++j) mx = fmax(mx, (double)sa.alpha_u[j]);
    for (int j = 0; j < sa.C; ++j) se += exp((double)sa.alpha_u[j] - mx);
    const double lse = mx + log(se);
    dir_sum = 0.0; dla_sum = 0.0;
    for (int j = 0; j < sa.C; ++j) {
      const double al = exp((double)sa.alpha_u[j] - lse);
      dir_sum += (conc - 1.0) * log(al + 1e-3);
      dla_sum += sa.red[3 + j] + (conc - 1.0) * al / (al + 1e-3);
    }
  }
  // chi terms: lane k < K
  double ep_k = 0.0;
  if (c < sa.K) {
    const double v = (double)pq.vch, chi = exp(v);
    ep_k = -0.5 * chi * gs[3 + c] + (double)sa.G * (0.5 * v - 0.5 * CA_LOG2PI) + (v - chi);
    const double gv = -0.5 * chi * gs[3 + c] + 0.5 * (double)sa.G + 1.0 - chi;
    sa.g_v[c] = (float)gv;
    if (sa.apply) {
      float th = pq.vch, m = pq.m_v, vv = pq.v_v;
      ca_adam(th, m, vv, -(float)gv, sa.lr_t, sa.b1, sa.b2, sa.aeps);
      (sa.vchi_out ? sa.vchi_out : sa.vchi)[c] = th; sa.m_v[c] = m; sa.v_v[c] = vv;
    }
  }
  const double ep_chi = wsum(ep_k);
  if (c == 0) {
    const double EE = sa.red[0] + gs[0];
    const double Ep = sa.red[1] + gs[1] + sa.dir_const + dir_sum + ep_chi;
    const double Eq = sa.red[2] + gs[2];
    if (sa.elbo_out) *sa.elbo_out = EE + Ep - Eq;
    if (sa.terms_out) { sa.terms_out[0] = EE; sa.terms_out[1] = Ep; sa.terms_out[2] = Eq; }
    if (sa.host_out) {   // the host loop of ca_run polls the flag instead of draining the stream
      *sa.host_out = EE + Ep - Eq;
      __threadfence_system();
      __hip_atomic_store(sa.host_flag, sa.host_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  if (sa.C <= 64) {
    if (c < sa.C) {
      const float ga = (float)(dla_c - al_c * dla_sum);
      sa.g_a[c] = ga;
      if (sa.apply) {
        float th = pq.au, m = pq.m_a, vv = pq.v_a;
        ca_adam(th, m, vv, -ga, sa.lr_t, sa.b1, sa.b2, sa.aeps);
        (sa.alpha_out ? sa.alpha_out : sa.alpha_u)[c] = th; sa.m_a[c] = m; sa.v_a[c] = vv;
      }
    }
  } else if (c == 0) {
    double mx = -INFINITY, se = 0.0;
    for (int j = 0; j < sa.C; ++j) mx = fmax(mx, (double)sa.alpha_u[j]);
    for (int j = 0; j < sa.C; ++j) se += exp((double)sa.alpha_u[j] - mx);
    const double lse = mx + log(se);
    for (int j = 0; j < sa.C; ++j) {
      const double al = exp((double)sa.alpha_u[j] - lse);
      const double dla = sa.red[3 + j] + (conc - 1.0) * al / (al + 1e-3);
      sa.g_a[j] = (float)(dla - al * dla_sum);
    }
    if (sa.apply)
      for (int j = 0; j < sa.C; ++j) {
        float th = sa.alpha_u[j], m = sa.m_a[j], vv = sa.v_a[j];
        ca_adam(th, m, vv, -sa.g_a[j], sa.lr_t, sa.b1, sa.b2, sa.aeps);
        (sa.alpha_out ? sa.alpha_out : sa.alpha_u)[j] = th; sa.m_a[j] = m; sa.v_a[j] = vv;
      }
  }
}

__device__ __forceinline__ void ca_final_small_body(const ca_small_args& sa) {
  __shared__ double sm[CA_TB];
  __shared__ double gs[3 + 16];
  const ca_small_pre pq = sa.reduce_only ? ca_small_pre{-INFINITY, 0.f, 0.f, 0.f, 0.f, 0.f} : ca_final_small_preload(sa);
  if (sa.cell_part) {   // k_reduce_part folded in (same fixed order: strided partial sums, then the block tree)
    const int Wc = 3 + sa.C;
    for (int j0 = 0; j0 < Wc; j0 += 4) {   // four columns per pass (one pair of barriers, interleaved butterflies)
      double a4[4] = {0.0, 0.0, 0.0, 0.0};
      for (int b = threadIdx.x; b < sa.ncblk; b += CA_TB) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (j0 + i < Wc) a4[i] += sa.cell_part[(int64_t)b * Wc + j0 + i];
      }
      ca_block_sum_n<4>(a4, sm);
      if (j0 == 0 && sa.yw_part) {   // the psi.(YW) term of EE_p_y, from the side stream's k_yw_dot
        double ya = 0.0;
        for (int b = threadIdx.x; b < sa.n_yw; b += CA_TB) ya += sa.yw_part[b];
        a4[0] += ca_block_sum(ya, sm);
      }
      if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (j0 + i < Wc) sa.red[j0 + i] = a4[i];
      }
    }
    __threadfence_block();
    __syncthreads();
  } else if (sa.ee_part) {   // second ELBO of a pair sweep: red[1 .. 3 + C) stand (same parameters), only EE_p_y is the other draw's
    double ea = 0.0;
    for (int b = threadIdx.x; b < sa.n_ee; b += CA_TB) ea += sa.ee_part[b];
    double r = ca_block_sum(ea, sm);
    if (sa.yw_part) {
      double ya = 0.0;
      for (int b = threadIdx.x; b < sa.n_yw; b += CA_TB) ya += sa.yw_part[b];
      r += ca_block_sum(ya, sm);
    }
    if (threadIdx.x == 0) sa.red[0] = r;
    __threadfence_block();
    __syncthreads();
  } else if (sa.yw_part) {   // second stage of a split tail: the cell partials are in red already (reduce_only stage on the backward
    double ya = 0.0;         // sweep), the psi.(YW) partials were not there yet (they are made by extra blocks of that same launch)
    for (int b = threadIdx.x; b < sa.n_yw; b += CA_TB) ya += sa.yw_part[b];
    const double r = ca_block_sum(ya, sm);
    if (threadIdx.x == 0) sa.red[0] += r;
    __threadfence_block();
    __syncthreads();
  }
  if (sa.reduce_only) return;   // (uniform) sharded runs: the ELBO is assembled after the all-reduce
  const int W_ = 3 + sa.K;
  {
    double a3[3] = {0.0, 0.0, 0.0};
    for (int b = threadIdx.x; b < sa.ngblk; b += CA_TB) {
      a3[0] += sa.gene_part[(int64_t)b * W_ + 0];
      a3[1] += sa.gene_part[(int64_t)b * W_ + 1];
      a3[2] += sa.gene_part[(int64_t)b * W_ + 2];
    }
    ca_block_sum_n<3>(a3, sm);
    if (threadIdx.x == 0) { gs[0] = a3[0]; gs[1] = a3[1]; gs[2] = a3[2]; }
  }
  for (int j = 3; j < W_; ++j) {
    double acc = 0.0;
    for (int b = threadIdx.x; b < sa.ngblk; b += CA_TB) acc += sa.gene_part[(int64_t)b * W_ + j];
    const double r = ca_block_sum(acc, sm);
    if (threadIdx.x == 0) gs[j] = r;
  }
  // range of the updated V' over the gene blocks (k_vmm_final folded in)
  if (sa.apply && sa.vmm_part && threadIdx.x >= CA_TB - 64) {   // the last wave: a lane per gene block, then butterflies
    const int ln = threadIdx.x & 63;
    for (int d = 0; d < sa.D; ++d) {
      float mn = INFINITY, mx2 = -INFINITY;
      for (int b = ln; b < sa.ngblk; b += 64) {
        mn = fminf(mn, sa.vmm_part[((int64_t)b * 2 + 0) * sa.D + d]);
        mx2 = fmaxf(mx2, sa.vmm_part[((int64_t)b * 2 + 1) * sa.D + d]);
      }
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { mn = fminf(mn, __shfl_xor(mn, o, 64)); mx2 = fmaxf(mx2, __shfl_xor(mx2, o, 64)); }
      if (ln == 0) { sa.vmm[d] = mn; sa.vmm[sa.D + d] = mx2; }
    }
  }
  __syncthreads();
  if (threadIdx.x < 64) ca_final_small_wave0(sa, gs, pq);
}

// ------------------------------------------------------------------ backward sweep on the matrix cores
// t_ng = sum_c coef_nc L_gc as ONE v_mfma_f32_16x16x32_bf16 per 16 genes x 16 cells: coef is split into three bf16
// parts by the cell epilogue (K = 3 parts x 8 clones = 24 of 32), copy numbers that are bf16-exact (integers up to
// 256: the normal case) make every product exact, accumulation is fp32 -- same result as the fp32 VALU chain up to
// summation order (tools/bwd_lab3.hip: 4e-8 relative).  Rows = genes, columns = cells, so a lane owns ONE cell per
// batch: d/dF needs a 4-lane-group sum per batch, the per-gene sums stay in-lane over the whole cell slice.
// The per-element math is written 2-wide (v_pk_*): measured equal to the scalar form here (tools/bwd_lab3.hip v6: 151 vs 148 us).
// Used when D is 1 or 2 (template DD), C <= 8 and L is bf16-exact; k_bwd is the general fallback.

// three bf16 parts of a float: x = p1 + p2 + p3 up to 2^-24 relative
__device__ __forceinline__ void ca_split3(float x, unsigned short& p1, unsigned short& p2, unsigned short& p3) {
  p1 = ca_bf16_rn(x); x -= __uint_as_float((unsigned)p1 << 16);
  p2 = ca_bf16_rn(x); x -= __uint_as_float((unsigned)p2 << 16);
  p3 = ca_bf16_rn(x);
}

// Progress priority (round 3).  The SIMD's arbiter serves the OLDEST ready wave first, so of the co-resident sweep blocks of a CU the
// first retires at a third of the round and the last runs alone at the end, one wave per SIMD on an issue port that wants three
// (tools/stamps.py: 4 x 96-cell blocks per CU end at 62 / 90 / 120 / 150 us).  A wave that lowers its own priority as it advances
// (s_setprio 3 in its first quarter ... 0 in its last) hands the issue slots to the waves behind it, and the blocks of a round end
// together.  The riding count-matrix stream's waves (HBM-bound, few instructions) and the finisher's extra blocks stay at 3.
// CA_PROG_PRIO: 0 off, 1 on, 2 on and the phase after the loop (accumulator combine, cell epilogue / partial writes) back at 3.
// Measured: neutral while the backward sweep still waited on its matrix-core products and 64-bit index arithmetic (3150 vs 3141 it/s),
// +3.6 % after those were gone (98 304 cells: 3683 -> 3816 it/s; profiles/r03_ab_ystream.txt section 15).
#ifndef CA_PROG_PRIO
#define CA_PROG_PRIO 2
#endif
#if CA_PROG_PRIO
#define CA_PRIO_STEP(i, qstep)                                                   \
  do {                                                                           \
    if ((i) == 0) __builtin_amdgcn_s_setprio(3);                                  \
    else if ((i) == (qstep)) __builtin_amdgcn_s_setprio(2);                       \
    else if ((i) == 2 * (qstep)) __builtin_amdgcn_s_setprio(1);                   \
    else if ((i) == 3 * (qstep)) __builtin_amdgcn_s_setprio(0);                   \
  } while (0)
#define CA_PRIO_DONE() __builtin_amdgcn_s_setprio(CA_PROG_PRIO == 2 ? 3 : 0)
#define CA_PRIO_STREAM() __builtin_amdgcn_s_setprio(3)
#else
#define CA_PRIO_STEP(i, qstep) do { } while (0)
#define CA_PRIO_DONE() do { } while (0)
#define CA_PRIO_STREAM() do { } while (0)
#endif
#ifndef CA_BWD_TL
#define CA_BWD_TL 4   // gene tiles of 16 per wave in the backward sweep
#endif
#ifndef CA_BWD_TL_D34
#define CA_BWD_TL_D34 3   // ... with three or four exponent dimensions (the per-gene accumulators grow with D: 207 / 250 registers at three tiles)
#endif
#ifndef CA_BWD_PD
#define CA_BWD_PD 2   // batches of operands in flight per wave (3 and more cost the third wave per SIMD: 140 -> 200 us)
#endif
// FRAC (round 3): copy numbers that are not bf16-exact (clonealign() accepts any non-negative matrix; saturate() only caps it at 6,
// R/clonealign.R:394-397).  L is then split in two bf16 parts like M in the forward sweep, coef in two, and the 24 operand slots
// carry [c1 L_hi | c2 L_hi | c1 L_lo]: what is dropped (c2 L_lo, and the third part of coef) is below 2^-17 of the product -- the
// forward sweep's own accuracy.  Integer copy numbers keep the exact three-part form.
// C16 (round 3): 9..16 clones with integer copy numbers.  The 32 operand slots carry two bf16 parts of coef for sixteen clones,
// slot group q = 2 * part + chunk (what the sixteen-lane cell epilogue writes), against L of clone chunk q & 1 in both parts.
// Round 6: the THIRD part rides in a second operand image (cq1: slot groups 0, 1 = part 3 of the two chunks, groups 2, 3 zero and never written) through a
// second matrix-core product into the same accumulator.  With two parts a product was good to 2^-17, and the per-gene gradients -- differences of sums over
// the cells that nearly cancel at a fit's start -- came out 40x less accurate than the vector sweep's (9000 x 5000 x 18 clones, four iterations: W off by
// 5e-5 at the median and by 0.2 at one gene, 1.3e-6 / 2e-5 on the vector unit; tools/lab/diag_c18.py).  The matrix cores are idle four fifths of this sweep.
// S2 (round 4, mc_samples = 2): BOTH samples of a train pass in one sweep.  exp(eta) does not depend on the sample (same psi, same W): one
// exponential per (cell, gene) serves two products -- the second sample brings its own coef operand (cq1), its own mu (mu1), its own matrix-core
// products and its own accumulators, and everything is summed in the order the sweep-per-sample form sums it (sample 0's partial first, then
// sample 1's added to it): bit for bit the two sweeps.  d/dF needs a second set of per-wave LDS slices (the host halves the cell slice).
template <int TL, int DD, bool FRAC = false, bool C16 = false, bool S2 = false>
__global__ void __launch_bounds__(CA_TB) k_bwd_mfma(const unsigned short* __restrict__ cq /*[N16][4][8] bf16 parts of coef*/,
                                                    const float* __restrict__ F /*[N16][DD]*/, const float* __restrict__ etamax2 /*[N16]*/,
                                                    const float* __restrict__ Lb /*[G][8]*/, const float* __restrict__ mu,
                                                    const float* __restrict__ Vs, const float* __restrict__ V,
                                                    float* __restrict__ gpart /*[csplit][G][S+DD]*/, float* __restrict__ dFpart /*[gridDim.x][N][DD]*/,
                                                    int64_t N, int G, int64_t cchunk, int S, int sidx, int first_s, int first,
                                                    ca_small_args tail, int yblocks, ca_yfin_args yfin,
                                                    const unsigned short* __restrict__ cq1 = nullptr, const float* __restrict__ mu1 = nullptr) {
  static_assert(!(S2 && C16), "two samples: up to eight clones");
  constexpr int NSM = S2 ? 2 : 1;     // samples per sweep
  extern __shared__ float ca_lds[];   // [NSM][4 waves][cchunk][DD]: per-wave d/dF of the block's cell slice, summed at the end
  // Extra block ROWS behind the sweep's own (blockIdx.y >= yblocks), so that they are dispatched LAST: the sweep's grid is exactly one
  // resident round, and extra blocks anywhere earlier in the dispatch order -- even ones that return at once -- take the first slots
  // of sweep blocks that then start late and finish 30 us after the rest (cfg-3: 145 -> 177 us).  Behind the sweep they get the
  // slots of the first blocks to retire, a third of the way through.
  if ((int)blockIdx.y >= yblocks) {   // the first of them assembles the pending monitor pass's ELBO (its fp64 chains hide under the sweep)
    const int e = ((int)blockIdx.y - yblocks) * (int)gridDim.x + (int)blockIdx.x;
    if (e == 0) { if (tail.enabled) ca_final_small_body(tail); return; }
    // ... the others finish the riding count-matrix stream's two products (ca_yfin_args).  At raised priority: the SIMD's arbiter
    // serves the oldest wave first, and beside sweep waves that are older and never short of instructions these few loads and adds
    // took 30 us to get through -- holding the slots of sweep blocks that then started that much later (cfg-3: sweep 145 -> 175 us)
    __builtin_amdgcn_s_setprio(3);
    const int ncolblk = (yfin.ncol + CA_TB / 64 - 1) / (CA_TB / 64);
    if (e - 1 < ncolblk) {
      const int job = (e - 1) * (CA_TB / 64) + (int)(threadIdx.x >> 6);
      if (job < yfin.ncol) ca_yfin_col_wave(yfin, job);
    } else if (e - 1 - ncolblk < yfin.nrow) {
      __shared__ double ca_yfin_sm[CA_TB / 64];
      ca_yfin_row_block(yfin, e - 1 - ncolblk, ca_yfin_sm);
    }
    return;
  }
  const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (scalar: the slice's bounds and the loop's branches then are)
  const int wtile = blockIdx.x * (CA_TB / 64) + wv;
  const int gbase = wtile * TL * 16;
  const bool active = gbase < G;
  ca_bf16x8 Lf[TL];
  ca_f32x2 vs[TL][2][DD], mv[NSM][TL][2][DD], accU[NSM][TL][2], accUF[NSM][TL][2][DD];
#pragma unroll
  for (int m = 0; m < TL; ++m) {
    {  // MFMA A operand: lane (row j, k-group q) holds L[gene gbase+16m+j][0..8), once per coef part (q < 3).
       // All prologue loads are unconditional on a clamped index and masked afterwards: guarded loads compile to one
       // branch + wait each and ran back to back (13 us per block, tools/bwd_lab3.hip)
      const int g = gbase + 16 * m + j;
      const bool ok = g < G && (C16 || q < 3);
      const int gg = g < G ? g : G - 1;
      const int64_t lrow = C16 ? (int64_t)(q & 1) * G + gg : (int64_t)gg;      // (C16: the clone chunk of this slot group)
      const float4 r0 = *reinterpret_cast<const float4*>(Lb + lrow * CA_CW);
      const float4 r1 = *reinterpret_cast<const float4*>(Lb + lrow * CA_CW + 4);
      const float lr[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
      unsigned short b[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        b[c] = ca_bf16_rn(lr[c]);
        if (FRAC && q == 2) b[c] = ca_bf16_rn(lr[c] - __uint_as_float((unsigned)b[c] << 16));   // the third slot group multiplies L_lo
      }
      const unsigned msk = ok ? 0xFFFFFFFFu : 0u;
      const uint4 raw = {((unsigned)b[0] | ((unsigned)b[1] << 16)) & msk, ((unsigned)b[2] | ((unsigned)b[3] << 16)) & msk,
                         ((unsigned)b[4] | ((unsigned)b[5] << 16)) & msk, ((unsigned)b[6] | ((unsigned)b[7] << 16)) & msk};
      Lf[m] = __builtin_bit_cast(ca_bf16x8, raw);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {  // this lane's output rows: genes gbase + 16m + 4q + {2h, 2h+1}
      float a[DD][2], b[NSM][DD][2];
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        const int g = gbase + 16 * m + 4 * q + 2 * h + x;
        const bool ok = g < G;
        const int gg = ok ? g : G - 1;
        float muv[NSM];
        muv[0] = mu[gg];
        if constexpr (S2) muv[1] = mu1[gg];
#pragma unroll
        for (int d = 0; d < DD; ++d) {
          const float vsv = Vs[(int64_t)gg * DD + d], vv = V[(int64_t)gg * DD + d];
          a[d][x] = vsv;             // rows past G keep a real gene's loading (exponent <= 0, never inf); their t is 0
#pragma unroll
          for (int sm_ = 0; sm_ < NSM; ++sm_) b[sm_][d][x] = ok ? muv[sm_] * vv : 0.f;
        }
      }
#pragma unroll
      for (int sm_ = 0; sm_ < NSM; ++sm_) accU[sm_][m][h] = (ca_f32x2){0.f, 0.f};
#pragma unroll
      for (int d = 0; d < DD; ++d) {
        vs[m][h][d] = (ca_f32x2){a[d][0], a[d][1]};
#pragma unroll
        for (int sm_ = 0; sm_ < NSM; ++sm_) {
          mv[sm_][m][h][d] = (ca_f32x2){b[sm_][d][0], b[sm_][d][1]};
          accUF[sm_][m][h][d] = (ca_f32x2){0.f, 0.f};
        }
      }
    }
  }
  const int64_t n0 = (int64_t)blockIdx.y * cchunk;
  const int64_t n1 = (n0 + cchunk < N) ? n0 + cchunk : N;
  float* myd = ca_lds + (int64_t)wv * cchunk * DD;
  constexpr int NWV = CA_TB / 64;
  if (!active)
    for (int64_t i = lane; i < (n1 - n0) * DD; i += 64) {
      myd[i] = 0.f;
      if constexpr (S2) myd[(int64_t)NWV * cchunk * DD + i] = 0.f;
    }
  // MFMA B operand: lane (column j, k-group q) holds part q of coef[cell b0+j][0..8): 16 bytes, 1 KiB per wave.
  // The operands of the next PD batches are in flight while the current one is in the pipes (cell arrays padded to 16): a
  // batch is 380 issue cycles = 0.6 us of wall time at three waves per SIMD, one batch of look-ahead left the wave parked on
  // s_waitcnt for 31 % of its cycles (SQ_WAIT_ANY, profiles/r01_v11_sq_counters.json) and far more beside an HBM stream.
  // Indices are 32-bit and relative to the slice, bases are the slice's (uniform) and lane offsets 32-bit: the loads take the
  // scalar-base form and the loop's compares are scalar -- with 64-bit cell indices every batch paid four 64-bit adds, three
  // 64-bit compares and their moves on the VALU, 14 of its 84 issue slots (round 3, from the ISA).
  constexpr int PD = CA_BWD_PD;
  const int qc = (FRAC && q == 2) ? 0 : q;   // which part of coef this lane group carries (FRAC: c1, c2, c1 again)
  const int len = active ? (int)(n1 - n0) : 0;
  const unsigned short* cqb = cq + n0 * 32;
  [[maybe_unused]] const unsigned short* cqb1 = (S2 || C16) ? cq1 + n0 * 32 : nullptr;
  const float* Fb = F + n0 * DD;
  const float* eb = etamax2 + n0;
  const unsigned lo_c = (unsigned)((j * 4 + qc) * 8), lo_f = (unsigned)(j * DD), lo_e = (unsigned)j;
  const int jl = len - j;                    // cell r + j is inside the slice iff r < jl
  float* myd_lane = myd + j * DD;
  uint4 craw_r[NSM][PD];
  [[maybe_unused]] uint4 c3_r[PD];
  float fc_r[PD][DD], ec_r[PD];
  auto fetch = [&](int slot, int r) {        // r: uniform, a multiple of 16, inside the padded arrays
    const unsigned short* pc = cqb + (int64_t)r * 32;
    const float* pf = Fb + (int64_t)r * DD;
    const float* pe = eb + r;
    craw_r[0][slot] = *reinterpret_cast<const uint4*>(pc + lo_c);
    if constexpr (S2) craw_r[1][slot] = *reinterpret_cast<const uint4*>(cqb1 + (int64_t)r * 32 + lo_c);
    if constexpr (C16) c3_r[slot] = *reinterpret_cast<const uint4*>(cqb1 + (int64_t)r * 32 + lo_c);
#pragma unroll
    for (int d = 0; d < DD; ++d) fc_r[slot][d] = pf[lo_f + d];
    ec_r[slot] = pe[lo_e];
  };
#pragma unroll
  for (int d_ = 0; d_ < PD; ++d_) fetch(d_, 16 * d_ < len ? 16 * d_ : 0);   // past the slice: re-read its first batch (never used)
  [[maybe_unused]] const int prio_q = ((len + 16 * PD - 1) / (16 * PD) + 3) / 4;
  [[maybe_unused]] int prio_i = 0;
  for (int r00 = 0; r00 < len; r00 += 16 * PD) {
  CA_PRIO_STEP(prio_i, prio_q);
  ++prio_i;
  [[maybe_unused]] float ddv[NSM][PD][DD];
#pragma unroll
  for (int d_ = 0; d_ < PD; ++d_) {
    const int r0 = r00 + 16 * d_;
#pragma unroll
    for (int sm_ = 0; sm_ < NSM; ++sm_)
#pragma unroll
      for (int d = 0; d < DD; ++d) ddv[sm_][d_][d] = 0.f;
    if (r0 < len) {   // wave-uniform
    uint4 craw[NSM];
#pragma unroll
    for (int sm_ = 0; sm_ < NSM; ++sm_) craw[sm_] = craw_r[sm_][d_];
    [[maybe_unused]] ca_bf16x8 Cf3;
    if constexpr (C16) Cf3 = __builtin_bit_cast(ca_bf16x8, c3_r[d_]);
    float fc[DD];
#pragma unroll
    for (int d = 0; d < DD; ++d) fc[d] = fc_r[d_][d];
    const float ec = ec_r[d_];
    if (r0 + 16 * PD < len) fetch(d_, r0 + 16 * PD);
    ca_bf16x8 Cf[NSM];
    ca_f32x2 dF[NSM][DD];
#pragma unroll
    for (int sm_ = 0; sm_ < NSM; ++sm_) {
      Cf[sm_] = __builtin_bit_cast(ca_bf16x8, craw[sm_]);
#pragma unroll
      for (int d = 0; d < DD; ++d) dF[sm_][d] = (ca_f32x2){0.f, 0.f};
    }
    // The batch's matrix-core products are issued AHEAD of their consumers: in the compiler's order each sat right in front of its
    // consumer and was waited for with s_nop (22 idle issue cycles per batch, and only three waves per SIMD to fill them; cfg-3
    // 3570 -> 3665 it/s).  CA_BWD_AHEAD products in flight: all four (default), or two with the next one issued as one is consumed.
#ifndef CA_BWD_AHEAD
#define CA_BWD_AHEAD TL
#endif
    constexpr int AH = CA_BWD_AHEAD < TL ? CA_BWD_AHEAD : TL;
    ca_f32x4 tt[NSM][TL];
#pragma unroll
    for (int m = 0; m < AH; ++m) {
#pragma unroll
      for (int sm_ = 0; sm_ < NSM; ++sm_) {
        tt[sm_][m] = (ca_f32x4){0.f, 0.f, 0.f, 0.f};
        // (the smallest part first, and for all tiles before the main products: a product into the accumulator of the one just issued waits for it)
        if constexpr (C16) tt[sm_][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Lf[m], Cf3, tt[sm_][m], 0, 0, 0);
      }
    }
#pragma unroll
    for (int m = 0; m < AH; ++m) {
#pragma unroll
      for (int sm_ = 0; sm_ < NSM; ++sm_)
        tt[sm_][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Lf[m], Cf[sm_], tt[sm_][m], 0, 0, 0);   // tt[.][m][r]: gene gbase+16m+4q+r, cell n0+r0+j
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < TL; ++m) {
      if constexpr (AH < TL) {
        if (m + AH < TL) {
#pragma unroll
          for (int sm_ = 0; sm_ < NSM; ++sm_) {
            tt[sm_][m + AH] = (ca_f32x4){0.f, 0.f, 0.f, 0.f};
            if constexpr (C16) tt[sm_][m + AH] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Lf[m + AH], Cf3, tt[sm_][m + AH], 0, 0, 0);
            tt[sm_][m + AH] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Lf[m + AH], Cf[sm_], tt[sm_][m + AH], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        ca_f32x2 eta = vs[m][h][0] * fc[0] - ec;
#pragma unroll
        for (int d = 1; d < DD; ++d) eta = vs[m][h][d] * fc[d] + eta;
        const ca_f32x2 ex = {__builtin_amdgcn_exp2f(eta.x), __builtin_amdgcn_exp2f(eta.y)};
#pragma unroll
        for (int sm_ = 0; sm_ < NSM; ++sm_) {
          const ca_f32x4 t = tt[sm_][m];
          const ca_f32x2 t2 = h == 0 ? (ca_f32x2){t[0], t[1]} : (ca_f32x2){t[2], t[3]};
          const ca_f32x2 u = ex * t2;
          accU[sm_][m][h] += u;
#pragma unroll
          for (int d = 0; d < DD; ++d) {
            accUF[sm_][m][h][d] = u * fc[d] + accUF[sm_][m][h][d];
            dF[sm_][d] = u * mv[sm_][m][h][d] + dF[sm_][d];
          }
        }
      }
      if constexpr (AH < TL) __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int sm_ = 0; sm_ < NSM; ++sm_)
#pragma unroll
    for (int d = 0; d < DD; ++d) {
      float dd = dF[sm_][d].x + dF[sm_][d].y;
      if constexpr (PD == 2) {
        ddv[sm_][d_][d] = dd;
      } else {
        dd = ca_sum_xor16_32(dd);   // over the four lane groups q (v_permlane16/32_swap: no LDS round trip, no lgkmcnt wait per batch)
        if (q == 0 && r0 < jl) myd_lane[(int64_t)sm_ * NWV * cchunk * DD + r0 * DD + d] = dd;
      }
    }
    }   // r0 < len
  }     // ring slot
  if constexpr (PD == 2) {
    // d/dF of the ring's two batches, summed over the four lane groups q TOGETHER: one v_permlane16_swap exchanges the odd rows of
    // batch 0 with the even rows of batch 1, so one add gives (q0 + q1), (q2 + q3) of both; the 32-lane swap then finishes both.
    // Rows 0 / 1 end up with batch 0 / 1: lane l < 32 holds cell r00 + l.  Same additions in the same order as one batch at a
    // time (ca_sum_xor16_32), half the swaps and adds, one LDS write instead of two.
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int sm_ = 0; sm_ < NSM; ++sm_)
#pragma unroll
    for (int d = 0; d < DD; ++d) {
      v2u r = __builtin_amdgcn_permlane16_swap(__float_as_uint(ddv[sm_][0][d]), __float_as_uint(ddv[sm_][1][d]), false, false);
      const float c = __uint_as_float(r.x) + __uint_as_float(r.y);
      r = __builtin_amdgcn_permlane32_swap(__float_as_uint(c), __float_as_uint(c), false, false);
      const float tot = __uint_as_float(r.x) + __uint_as_float(r.y);
      if (lane < 32 && r00 + lane < len) myd[(int64_t)sm_ * NWV * cchunk * DD + (r00 + lane) * DD + d] = tot;
    }
  }
  }
  CA_PRIO_DONE();
  __syncthreads();
  const int64_t wstride = cchunk * DD;
  for (int64_t i = threadIdx.x; i < (n1 - n0) * DD; i += CA_TB) {
    float d = (ca_lds[i] + ca_lds[wstride + i]) + (ca_lds[2 * wstride + i] + ca_lds[3 * wstride + i]);
    if constexpr (S2) {   // (sample 0's sum, then sample 1's added to it: what the second sweep did through memory)
      const float* l1 = ca_lds + 4 * wstride;
      d = d + ((l1[i] + l1[wstride + i]) + (l1[2 * wstride + i] + l1[3 * wstride + i]));
    }
    float* p = dFpart + ((int64_t)blockIdx.x * N + n0) * DD + i;
    *p = first ? d : (*p + d);
  }
  if (!active) return;
  const int W_ = S + DD;
#pragma unroll
  for (int m = 0; m < TL; ++m)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      // sum over the 16 cell lanes of the row; lanes j = 0,1 write genes 2h, 2h+1 of the lane group
      auto row16 = [](float v) {
        v += ca_dpp_pull<0xB1, 0xF>(v); v += ca_dpp_pull<0x4E, 0xF>(v); v += ca_dpp_pull<0x141, 0xF>(v); v += ca_dpp_pull<0x140, 0xF>(v);
        return v;
      };
      float a0[NSM], a1[NSM], bx[NSM][DD], by[NSM][DD];
#pragma unroll
      for (int sm_ = 0; sm_ < NSM; ++sm_) {
        a0[sm_] = row16(accU[sm_][m][h].x); a1[sm_] = row16(accU[sm_][m][h].y);
#pragma unroll
        for (int d = 0; d < DD; ++d) { bx[sm_][d] = row16(accUF[sm_][m][h][d].x); by[sm_][d] = row16(accUF[sm_][m][h][d].y); }
      }
      if (j < 2) {
        const int g = gbase + 16 * m + 4 * q + 2 * h + j;
        if (g < G) {
          float* gp = gpart + ((int64_t)blockIdx.y * G + g) * W_;
          const float su = j ? a1[0] : a0[0];
          gp[sidx] = first_s ? su : gp[sidx] + su;
          if constexpr (S2) gp[sidx + 1] = j ? a1[1] : a0[1];
#pragma unroll
          for (int d = 0; d < DD; ++d) {
            // (products rounded on their own, then added: the sum over the samples is the same float whether the second sample's term comes
            //  from this sweep or from a second one through memory -- no fused multiply-add across that boundary)
            //  (the empty asm makes the product an opaque value: __fmul_rn is a plain multiplication to this compiler and would be contracted)
            float suf = mu[g] * (j ? by[0][d] : bx[0][d]);
            asm volatile("" : "+v"(suf));
            if constexpr (S2) {
              float suf1 = mu1[g] * (j ? by[1][d] : bx[1][d]);
              asm volatile("" : "+v"(suf1));
              suf = suf + suf1;
            }
            gp[S + d] = first ? suf : gp[S + d] + suf;
          }
        }
      }
    }
}

// ------------------------------------------------------------------ preprocessing statistics
// R/preprocess.R:93-147 needs two statistics of the RAW count matrix: colSums(Y) (per gene, all cells) and, after the
// gene filters, rowSums(Y[, kept]) (per cell).  Both are single passes over the caller's matrix in its own dtype and
// layout (element (n, g) at src[n * sn + g * sg]); sums are fp64 and taken in a fixed order.
template <typename ST>
__global__ void __launch_bounds__(CA_TB) k_pre_colsum(const ST* __restrict__ src, int64_t N, int G, int64_t sn, int64_t sg,
                                                      int rows_per_block, double* __restrict__ part /*[gridDim.y][G]*/) {
  const int g = blockIdx.x * CA_TB + threadIdx.x;
  if (g >= G) return;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = (r0 + rows_per_block < N) ? r0 + rows_per_block : N;
  double a = 0.0;
  for (int64_t n = r0; n < r1; ++n) a += (double)src[n * sn + (int64_t)g * sg];
  part[(int64_t)blockIdx.y * G + g] = a;
}
__global__ void __launch_bounds__(CA_TB) k_pre_colsum_final(const double* __restrict__ part, int nrb, int G, double* __restrict__ out) {
  const int g = blockIdx.x * CA_TB + threadIdx.x;
  if (g >= G) return;
  double a = 0.0;
  for (int r = 0; r < nrb; ++r) a += part[(int64_t)r * G + g];
  out[g] = a;
}
// one wave per cell, lanes over genes (row-major input: coalesced); fixed-order DPP tree
template <typename ST>
__global__ void __launch_bounds__(CA_TB) k_pre_rowsum(const ST* __restrict__ src, const unsigned char* __restrict__ keep_gene, int64_t N,
                                                      int G, int64_t sn, int64_t sg, double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t n = (int64_t)blockIdx.x * (CA_TB / 64) + (threadIdx.x >> 6);
  if (n >= N) return;
  double a = 0.0;
  for (int g = lane; g < G; g += 64)
    if (keep_gene[g]) a += (double)src[n * sn + (int64_t)g * sg];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
  if (lane == 0) out[n] = a;
}
// one thread per cell, loop over genes (column-major input: coalesced across cells)
template <typename ST>
__global__ void __launch_bounds__(CA_TB) k_pre_rowsum_cm(const ST* __restrict__ src, const unsigned char* __restrict__ keep_gene, int64_t N,
                                                         int G, int64_t sn, int64_t sg, double* __restrict__ out) {
  const int64_t n = (int64_t)blockIdx.x * CA_TB + threadIdx.x;
  if (n >= N) return;
  double a = 0.0;
  for (int g = 0; g < G; ++g)
    if (keep_gene[g]) a += (double)src[n * sn + (int64_t)g * sg];
  out[n] = a;
}

// ------------------------------------------------------------------ allele-specific term
// R/allele-specific.R:17-58: the parameter-free [N, C] addend of the log-likelihood.  Per (variant, cell) two
// log-probabilities of alt reads out of cov: p1 = logsumexp(log .5 + BB(.1, 1.9), log .5 + BB(1.9, .1)) for a clone
// with allelic imbalance at the variant, p2 = BB(2, 2) for copy number 2; out[n, c] = sum_v (cn[v, c] == 2 ? p2 : p1)
//                                                                                   = sum_v p1 + sum_v is2[v, c] (p2 - p1).
// One block per cell, threads over variants (12 lgamma per pair: the binomial coefficient is shared by the three BB
// terms, the beta-function constants are kernel arguments), p2 - p1 staged in LDS for the per-clone sums.
__device__ __forceinline__ double ca_bb_tail(double k, double n, double a, double b, double cab) {
  return lgamma(k + a) + lgamma(n - k + b) - lgamma(a + b + n) + cab;   // cab = lgamma(a + b) - lgamma(a) - lgamma(b)
}
__global__ void __launch_bounds__(CA_TB) k_allele_loglik(const double* __restrict__ cov, const double* __restrict__ ref, int64_t sn, int64_t sv,
                                                         const unsigned char* __restrict__ is2 /*[V][C]*/, double* __restrict__ out,
                                                         int64_t on, int64_t oc, int64_t N, int V, int C, int vtile, double c_low,
                                                         double c_high, double c_two) {
  extern __shared__ double ca_ldsd[];   // [vtile] p2 - p1 of the current variant tile
  __shared__ double sm[CA_TB];
  const int64_t n = blockIdx.x;
  const double LOG_HALF = -0.69314718055994530942;
  double s1 = 0.0;   // this thread's share of sum_v p1
  for (int v0 = 0; v0 < V; v0 += vtile) {
    const int nv = min(vtile, V - v0);
    for (int i = threadIdx.x; i < nv; i += CA_TB) {
      const double cv = cov[n * sn + (int64_t)(v0 + i) * sv], rf = ref[n * sn + (int64_t)(v0 + i) * sv];
      const double k = cv - rf;   // alt = cov - ref (R/inference-tflow.R:173)
      const double binom = lgamma(cv + 1.0) - lgamma(k + 1.0) - lgamma(cv - k + 1.0);
      const double lo = LOG_HALF + binom + ca_bb_tail(k, cv, 0.1, 1.9, c_low);
      const double hi = LOG_HALF + binom + ca_bb_tail(k, cv, 1.9, 0.1, c_high);
      const double mx = fmax(lo, hi);
      const double p1 = (mx == -INFINITY) ? -INFINITY : mx + log(exp(lo - mx) + exp(hi - mx));
      const double p2 = binom + ca_bb_tail(k, cv, 2.0, 2.0, c_two);
      s1 += p1;
      ca_ldsd[i] = p2 - p1;
    }
    __syncthreads();
    for (int c = 0; c < C; ++c) {
      double a = 0.0;
      for (int i = threadIdx.x; i < nv; i += CA_TB)
        if (is2[(int64_t)(v0 + i) * C + c]) a += ca_ldsd[i];
      const double r = ca_block_sum(a, sm);
      if (threadIdx.x == 0) {
        double* o = out + n * on + (int64_t)c * oc;
        *o = (v0 == 0 ? 0.0 : *o) + r;
      }
    }
    __syncthreads();
  }
  const double t1 = ca_block_sum(s1, sm);
  if (threadIdx.x == 0)
    for (int c = 0; c < C; ++c) out[n * on + (int64_t)c * oc] += t1;
}

// log_alpha = log_softmax(alpha_unconstr) (R/inference-tflow.R:255) into LDS, by wave 0: one lane per clone, the C
// exponentials side by side (they were a serial chain on thread 0: ~2 us at the head of every cell-epilogue block)
__device__ __forceinline__ void ca_log_softmax_alpha(const float* __restrict__ alpha_u, int C, double* la) {
  // (the block's LAST wave: in the fused sweep wave 0 has one k-step more than the others whenever the k-step count is 4 n + 1,
  //  and this fp64 chain stood in front of its loop)
  if (threadIdx.x < CA_TB - 64) return;
  if (C <= 64) {
    const int c = threadIdx.x - (CA_TB - 64);
    const double au = c < C ? (double)alpha_u[c] : -INFINITY;
    double mx = au;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
    double se = c < C ? exp(au - mx) : 0.0;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) se += __shfl_xor(se, o, 64);
    if (c < C) la[c] = au - (mx + log(se));
  } else if (threadIdx.x == CA_TB - 64) {
    double mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmax(mx, (double)alpha_u[c]);
    double se = 0.0;
    for (int c = 0; c < C; ++c) se += exp((double)alpha_u[c] - mx);
    const double lse = mx + log(se);
    for (int c = 0; c < C; ++c) la[c] = (double)alpha_u[c] - lse;
  }
}


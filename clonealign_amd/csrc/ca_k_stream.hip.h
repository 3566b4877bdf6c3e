// ca_k_stream.hip.h -- part of ca_kernels.hip.h (textually included there, in this order): ingest / fit-constant kernels, the vector count-matrix stream (k_ypass), per-gene prologues, the VALU and first matrix-core forward sweeps.

// ------------------------------------------------------------------ count-matrix element decode
template <typename YT> struct YVec;
template <> struct YVec<float> {
  static constexpr int VEC = 4;
  __device__ static void decode(const uint4 v, float (&y)[4]) {
    y[0] = __uint_as_float(v.x); y[1] = __uint_as_float(v.y); y[2] = __uint_as_float(v.z); y[3] = __uint_as_float(v.w);
  }
};
template <> struct YVec<uint16_t> {
  static constexpr int VEC = 8;
  __device__ static void decode(const uint4 v, float (&y)[8]) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      y[2 * i] = (float)(w[i] & 0xFFFFu);
      y[2 * i + 1] = (float)(w[i] >> 16);
    }
  }
};
template <> struct YVec<uint8_t> {
  static constexpr int VEC = 16;
  __device__ static void decode(const uint4 v, float (&y)[16]) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      y[4 * i] = (float)(w[i] & 0xFFu);            // v_cvt_f32_ubyte0..3
      y[4 * i + 1] = (float)((w[i] >> 8) & 0xFFu);
      y[4 * i + 2] = (float)((w[i] >> 16) & 0xFFu);
      y[4 * i + 3] = (float)(w[i] >> 24);
    }
  }
};

// value transforms of the count stream (template parameter TF of k_ypass / runtime tf of the overflow kernels):
//   0 identity (the VI loop), 1 log2(y + 1), 2 log2(y + 1)^2   (PCA initialisation, R/inference-tflow.R:204),
//   3 y^2 (post-hoc gene/copy-number correlations, R/clonealign.R:318-334)
template <int TF>
__device__ __forceinline__ float ca_ytf(float y) {
  if (TF == 0) return y;
  if (TF == 3) return y * y;
  const float x = __builtin_amdgcn_logf(y + 1.f);   // v_log_f32 = log2
  return TF == 1 ? x : x * x;
}
__device__ __forceinline__ float ca_ytf_rt(float y, int tf) {
  return tf == 0 ? ca_ytf<0>(y) : tf == 1 ? ca_ytf<1>(y) : tf == 2 ? ca_ytf<2>(y) : ca_ytf<3>(y);
}

// ------------------------------------------------------------------ upload / conversion
// One thread per element of the N x Gp matrix: a launch's x extent is a 32-bit count of work-items (the dispatch packet's grid size), so 2^32 elements --
// 838 860 cells at 5120 padded genes -- is where a one-dimensional grid silently wraps (round 5: a 1M-cell matrix came up with its first 161 140 cells
// converted and the rest zero).  These kernels take a two-dimensional grid (ca_grid_flat on the host) and flatten it here.
__device__ __forceinline__ int64_t ca_flat_index() {
  return ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
}
// src is N x G in either layout and any ca_dtype; dst is row-major [N][Gp] of YT, zero padded.
template <typename ST, typename YT>
__global__ void k_convert_y(const ST* __restrict__ src, YT* __restrict__ dst, int64_t N, int G, int Gp,
                            int64_t sn, int64_t sg, int* __restrict__ flags) {
  const int64_t i = ca_flat_index();
  if (i >= N * (int64_t)Gp) return;
  const int64_t n = i / Gp;
  const int g = (int)(i - n * Gp);
  YT out = 0;
  if (g < G) {
    const double v = (double)src[n * sn + (int64_t)g * sg];
    out = (YT)v;
    if ((double)out != v) atomicOr(flags, 1);  // not representable in the storage type
    if (!(v >= 0.0)) atomicOr(flags, 2);       // negative or NaN count
  }
  dst[i] = out;
}

// Row / column selection at upload (ca_problem.cell_index / gene_index): dst [N][G] row-major in the source's own type,
// element (n, g) = src[cell_index[n] * sn + gene_index[g] * sg].  The raw matrix is uploaded once and cut here instead of
// on the host (the reference copies Y[cells, genes] in R: R/preprocess.R:141-147, R/inference-tflow.R:117-124).
template <typename ST>
__global__ void k_gather_y(const ST* __restrict__ src, ST* __restrict__ dst, int64_t N, int G, int64_t sn, int64_t sg,
                           const int64_t* __restrict__ cell_index, const int32_t* __restrict__ gene_index) {
  const int64_t i = ca_flat_index();
  if (i >= N * (int64_t)G) return;
  const int64_t n = i / G;
  const int g = (int)(i - n * G);
  const int64_t rn = cell_index ? cell_index[n] : n;
  const int64_t rg = gene_index ? (int64_t)gene_index[g] : (int64_t)g;
  dst[i] = src[rn * sn + rg * sg];
}

// u8 storage with an overflow list: the dense byte holds min(y, 255); the (rare) excess y - 255 goes to a
// COO list (appended in arbitrary order here, sorted on the host afterwards so that every later sum over it
// has a fixed order).
template <typename ST>
__global__ void k_convert_y_u8ovf(const ST* __restrict__ src, uint8_t* __restrict__ dst, int64_t N, int G, int Gp, int64_t sn,
                                  int64_t sg, unsigned long long* __restrict__ counter, int* __restrict__ orow,
                                  int* __restrict__ ocol, float* __restrict__ oval) {
  const int64_t i = ca_flat_index();
  if (i >= N * (int64_t)Gp) return;
  const int64_t n = i / Gp;
  const int g = (int)(i - n * Gp);
  uint8_t out = 0;
  if (g < G) {
    const double v = (double)src[n * sn + (int64_t)g * sg];
    if (v > 255.0) {
      out = 255;
      const unsigned long long k = atomicAdd(counter, 1ull);
      orow[k] = (int)n; ocol[k] = g; oval[k] = (float)(v - 255.0);
    } else {
      out = (uint8_t)v;
    }
  }
  dst[i] = out;
}

// overflow-list contributions to the Y stream products, one thread per cell (CSR order) / per gene (CSC order)
__device__ __forceinline__ void ca_ovf_rows_body(int blk, const int64_t* __restrict__ rowptr, const int* __restrict__ col,
                                                 const float* __restrict__ val, const float* __restrict__ V, int Dstride,
                                                 float* __restrict__ YWextra /*[N][K]*/, int64_t N, int K, int tf, int bdim = 0 /* rows per block; 0 = blockDim.x */) {
  const int64_t n = (int64_t)blk * (bdim ? bdim : (int)blockDim.x) + threadIdx.x;
  if (n >= N) return;
  for (int k = 0; k < K; ++k) {
    float a = 0.f;
    for (int64_t e = rowptr[n]; e < rowptr[n + 1]; ++e) {
      const float dv = tf == 0 ? val[e] : ca_ytf_rt(255.f + val[e], tf) - ca_ytf_rt(255.f, tf);   // T(y) - T(255)
      a = fmaf(dv, V[(int64_t)col[e] * Dstride + k], a);
    }
    YWextra[n * K + k] = a;
  }
}
__global__ void k_ovf_rows(const int64_t* __restrict__ rowptr, const int* __restrict__ col, const float* __restrict__ val,
                           const float* __restrict__ V, int Dstride, float* __restrict__ YWextra /*[N][K]*/, int64_t N, int K,
                           int tf) {
  ca_ovf_rows_body(blockIdx.x, rowptr, col, val, V, Dstride, YWextra, N, K, tf);
}
// Gene side of the overflow list.  The excess entries concentrate in a few highly expressed genes (one entry
// per cell there), so each gene's CSC range is cut into chunks of <= 256 entries: one wave per chunk
// (k_ovf_chunks), then one thread per gene adds its chunk sums in order (k_ovf_cols).
__device__ __forceinline__ void ca_ovf_chunks_body(int blk, const int64_t* __restrict__ chunk_start, const int* __restrict__ row,
                                                   const float* __restrict__ val, const float* __restrict__ F, int Dstride,
                                                   float* __restrict__ csum /*[nchunk][K]*/, int nchunk, int K, int tf) {
  const int lane = threadIdx.x & 63;
  const int ch = blk * (CA_TB / 64) + (threadIdx.x >> 6);
  if (ch >= nchunk) return;
  const int64_t e0 = chunk_start[ch], e1 = chunk_start[ch + 1];
  for (int k = 0; k < K; ++k) {
    float a = 0.f;
    for (int64_t e = e0 + lane; e < e1; e += 64) {
      const float dv = tf == 0 ? val[e] : ca_ytf_rt(255.f + val[e], tf) - ca_ytf_rt(255.f, tf);
      a = fmaf(dv, F[(int64_t)row[e] * Dstride + k], a);
    }
    const float tot = ca_wave_sum_lane63(a);
    if (lane == 63) csum[(int64_t)ch * K + k] = tot;
  }
}
__global__ void __launch_bounds__(CA_TB) k_ovf_chunks(const int64_t* __restrict__ chunk_start, const int* __restrict__ row,
                                                      const float* __restrict__ val, const float* __restrict__ F, int Dstride,
                                                      float* __restrict__ csum /*[nchunk][K]*/, int nchunk, int K, int tf) {
  ca_ovf_chunks_body(blockIdx.x, chunk_start, row, val, F, Dstride, csum, nchunk, K, tf);
}
// the overflow list's two per-entry kernels as extra blocks of the Y stream launch (k_ypass): they depend on nothing
// the stream computes, and as launches of their own they were 2 x 5 us of pure latency on the side stream
struct ca_ovf_args {
  int nb_rows, nb_chunks;   // extra blocks after the stream's own (0 = none)
  const int64_t* rowptr; const int* col; const float* val; float* YWextra;
  const int64_t* chunk_start; const int* row2; const float* val2; float* csum; int nchunk;
};
__global__ void k_ovf_cols(const int* __restrict__ col_chunk_ptr, const float* __restrict__ csum,
                           float* __restrict__ YTextra /*[Gp][K]*/, int Gp, int G, int K) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= Gp) return;
  for (int k = 0; k < K; ++k) {
    float a = 0.f;
    if (g < G)
      for (int c = col_chunk_ptr[g]; c < col_chunk_ptr[g + 1]; ++c) a += csum[(int64_t)c * K + k];
    YTextra[(int64_t)g * K + k] = a;
  }
}

// max / integrality scan used to choose the storage width (flags bit0: non-integer, bit1: negative/NaN)
template <typename ST>
__global__ void k_scan_y(const ST* __restrict__ src, int64_t total, double* __restrict__ maxv, int* __restrict__ flags,
                         unsigned long long* __restrict__ n_over255) {
  __shared__ double sm[CA_TB];
  double m = 0.0;
  int f = 0;
  unsigned long long over = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const double v = (double)src[i];
    if (!(v >= 0.0)) f |= 2;
    if (v != floor(v)) f |= 1;
    if (v > 255.0) ++over;
    m = v > m ? v : m;
  }
  if (f) atomicOr(flags, f);
  if (over) atomicAdd(n_over255, over);
  sm[threadIdx.x] = m;
  __syncthreads();
  for (int s = CA_TB / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) sm[threadIdx.x] = sm[threadIdx.x] > sm[threadIdx.x + s] ? sm[threadIdx.x] : sm[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    // doubles >= 0 order like their bit patterns
    atomicMax(reinterpret_cast<unsigned long long*>(maxv), (unsigned long long)__double_as_longlong(sm[0]));
  }
}

// Column sums of x = log2(y + 1) and of x^2 in FLOAT64, for prcomp(center = TRUE, scale. = TRUE) of R/inference-tflow.R:204-205 on the
// device (ca_init_psi_pca): thread = gene (a block row reads 256 consecutive columns of the row-major resident matrix), block column = a
// slice of cells; out [slices][2][G].  (Round 5: these two statistics came out of the float32 streaming pass before; the standard
// deviation is a difference of two nearly equal sums for a well-expressed gene, and float32 partials cost the device PCA a factor ten
// in agreement with prcomp.)  Counts stored as 255 + overflow-list excess are corrected by the caller.
template <typename YT>
__global__ void __launch_bounds__(CA_TB) k_col_logstats(const YT* __restrict__ Y, int64_t N, int G, int Gp, int64_t rows_per,
                                                        double* __restrict__ out) {
  const int g = (int)blockIdx.x * CA_TB + (int)threadIdx.x;
  if (g >= G) return;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per, r1 = r0 + rows_per < N ? r0 + rows_per : N;
  double s = 0.0, ss = 0.0;
  for (int64_t r = r0; r < r1; ++r) {
    const double y = (double)Y[r * (int64_t)Gp + g];
    if (y != 0.0) { const double x = log2(y + 1.0); s += x; ss += x * x; }
  }
  out[((int64_t)blockIdx.y * 2 + 0) * G + g] = s;
  out[((int64_t)blockIdx.y * 2 + 1) * G + g] = ss;
}

// ------------------------------------------------------------------ fit constants (once per fit)
// One block per cell: s_n, c_n = lgamma(s_n+1) - sum_g lgamma(y+1), A_nc = sum_g xlogy(y, L_gc).
// (the terms TF recomputes in every run of tfd$Multinomial$log_prob, R/inference-tflow.R:294-296)
template <typename YT>
__global__ void __launch_bounds__(CA_TB) k_prep_cells(const YT* __restrict__ Y, const double* __restrict__ logL /*[G][C]*/,
                                                      const double* __restrict__ extra /*[N][C] or null*/,
                                                      double* __restrict__ A, double* __restrict__ cn,
                                                      double* __restrict__ s64, float* __restrict__ s32, int64_t N, int G,
                                                      int Gp, int C, const int64_t* __restrict__ orowptr,
                                                      const int* __restrict__ ocol, const float* __restrict__ oval) {
  __shared__ double sm[CA_TB];
  const int64_t n = blockIdx.x;
  const YT* row = Y + n * (int64_t)Gp;
  double ssum = 0.0, lg = 0.0;
  for (int g = threadIdx.x; g < G; g += CA_TB) {
    const double y = (double)row[g];
    ssum += y;
    if (y > 1.0) lg += lgamma(y + 1.0);
    else if (y > 0.0 && y < 1.0) lg += lgamma(y + 1.0);
  }
  // entries stored as 255 + overflow: add the excess and swap lgamma(256) for lgamma(256 + excess)
  const int64_t oe0 = orowptr ? orowptr[n] : 0, oe1 = orowptr ? orowptr[n + 1] : 0;
  for (int64_t e = oe0 + threadIdx.x; e < oe1; e += CA_TB) {
    const double x = (double)oval[e];
    ssum += x;
    lg += lgamma(256.0 + x) - lgamma(256.0);
  }
  const double st = ca_block_sum(ssum, sm);
  const double lt = ca_block_sum(lg, sm);
  if (threadIdx.x == 0) {
    s64[n] = st;
    s32[n] = (float)st;
    cn[n] = lgamma(st + 1.0) - lt;
  }
  for (int c = 0; c < C; ++c) {
    double a = 0.0;
    for (int g = threadIdx.x; g < G; g += CA_TB) {
      const double y = (double)row[g];
      if (y != 0.0) a += y * logL[(int64_t)g * C + c];  // xlogy: 0*log(0) := 0, y>0 & L=0 -> -inf
    }
    for (int64_t e = oe0 + threadIdx.x; e < oe1; e += CA_TB) a += (double)oval[e] * logL[(int64_t)ocol[e] * C + c];
    const double at = ca_block_sum(a, sm);
    if (threadIdx.x == 0) A[n * C + c] = at + (extra ? extra[n * C + c] : 0.0);
  }
}

// The same constants for 1-byte storage, the usual case: one WAVE per cell (16-byte loads, wave sums by shuffles, no
// barriers in the cell loop), lgamma(y + 1) from a 256-entry table built once per block, log L gathered from L2 only for
// the non-zero counts, grid-stride over cells.  13.0 -> 3.3 ms at 100k x 5k x 8 (the old form was a tenth of a 200-iteration fit).
__global__ void __launch_bounds__(CA_TB) k_prep_cells_u8(const uint8_t* __restrict__ Y, const double* __restrict__ logL /*[G][C]*/,
                                                         const double* __restrict__ extra /*[N][C] or null*/, double* __restrict__ A,
                                                         double* __restrict__ cn, double* __restrict__ s64, float* __restrict__ s32,
                                                         int64_t N, int G, int Gp, int C, const int64_t* __restrict__ orowptr,
                                                         const int* __restrict__ ocol, const float* __restrict__ oval) {
  __shared__ double lgt[CA_TB];   // CA_TB == 256: lgt[y] = lgamma(y + 1)
  lgt[threadIdx.x] = lgamma((double)threadIdx.x + 1.0);
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  auto wsum = [](double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
  };
  for (int64_t n = (int64_t)blockIdx.x * (CA_TB / 64) + wv; n < N; n += (int64_t)gridDim.x * (CA_TB / 64)) {
    const uint8_t* row = Y + n * (int64_t)Gp;
    const int64_t oe0 = orowptr ? orowptr[n] : 0, oe1 = orowptr ? orowptr[n + 1] : 0;
    for (int c0 = 0; c0 < C; c0 += 8) {   // eight clone columns per sweep of the row (one sweep when C <= 8)
      const int nc = C - c0 < 8 ? C - c0 : 8;
      double a[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
      double ssum = 0.0, lg = 0.0;
      for (int g0 = lane * 16; g0 < G; g0 += 64 * 16) {
        const uint4 raw = *reinterpret_cast<const uint4*>(row + g0);   // rows are padded to whole 1 KiB strips (zeros)
        const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const unsigned y = (w[j >> 2] >> (8 * (j & 3))) & 255u;
          const int g = g0 + j;
          if (y != 0u && g < G) {   // xlogy: 0 * log(0) := 0; y > 0 with L = 0 gives -inf like the reference
            const double yd = (double)y;
            ssum += yd;
            lg += lgt[y];
            const double* lp = logL + (int64_t)g * C + c0;
#pragma unroll
            for (int c = 0; c < 8; ++c)
              if (c < nc) a[c] += yd * lp[c];
          }
        }
      }
      // entries stored as 255 + overflow: add the excess and swap lgamma(256) for lgamma(256 + excess)
      for (int64_t e = oe0 + lane; e < oe1; e += 64) {
        const double x = (double)oval[e];
        ssum += x;
        lg += lgamma(256.0 + x) - lgt[255];
        const double* lp = logL + (int64_t)ocol[e] * C + c0;
#pragma unroll
        for (int c = 0; c < 8; ++c)
          if (c < nc) a[c] += x * lp[c];
      }
      if (c0 == 0) {
        const double st = wsum(ssum), lt = wsum(lg);
        if (lane == 0) {
          s64[n] = st;
          s32[n] = (float)st;
          cn[n] = lgamma(st + 1.0) - lt;
        }
      }
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        if (c < nc) {
          const double at = wsum(a[c]);
          if (lane == 0) A[n * C + c0 + c] = at + (extra ? extra[n * C + c0 + c] : 0.0);
        }
      }
    }
  }
}

// ------------------------------------------------------------------ Y stream: YW = Y.W and YtPsi = Y^T.psi
// The only kernel that reads the count matrix inside the iteration loop (HBM-bound).  One wave
// owns a strip of TR cells x (64*VEC) genes: 16-byte coalesced loads, per-lane column partials
// in registers, per-row partial reduced across the wave with DPP.
//   YWpart[seg][n][k]  = sum over the strip's genes of y_ng W_gk          (summed over seg later)
//   YTpart[rb][g][k]   = sum over the strip's cells of y_ng psi_nk        (summed over rb later)
// Everything that is the same for the whole wave is kept in SGPRs on purpose (v_readfirstlane of the wave index): the
// strip bounds, the row loop, the row base address (loads are `global_load_dwordx4 v, v_off, s[base]`), and the row's
// psi, which is fetched once per strip into one VGPR per 64 rows and read back with v_readlane.  Before, the
// compiler carried the row index in 64-bit vector registers (10 VALU per load address) and fetched psi with a
// vector load per row whose s_waitcnt vmcnt(0) also drained the prefetched Y rows.  Rows are processed in two
// alternating groups of U so the prefetch needs no register copies; row totals are parked one per lane with
// v_writelane and stored 64 at a time (a per-row store would sit in the same in-order vmcnt queue as the loads).
template <typename YT, int KK, int TF = 0>
__device__ __forceinline__ void ca_ypass_body(int blk, const YT* __restrict__ Y, const float* __restrict__ F, int Dstride,
                                              const float* __restrict__ V, int koff, float* __restrict__ YWpart,
                                              float* __restrict__ YTpart, int64_t N, int G, int Gp, int nseg,
                                              int nrb, int TR, int K, const ca_ovf_args& ovf, int nb_main,
                                              float (*ycomb)[64][YVec<YT>::VEC + 1] /* [CA_TB / 64] rows of shared memory */) {
  constexpr int VEC = YVec<YT>::VEC;
  if (blk >= nb_main) {   // overflow-list blocks (identity transform only: the VI loop)
    const int b = blk - nb_main;
    if (b < ovf.nb_rows) ca_ovf_rows_body(b, ovf.rowptr, ovf.col, ovf.val, V, Dstride, ovf.YWextra, N, K, 0);
    else ca_ovf_chunks_body(b - ovf.nb_rows, ovf.chunk_start, ovf.row2, ovf.val2, F, Dstride, ovf.csum, ovf.nchunk, K, 0);
    return;
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // block = one gene segment x 4 consecutive row blocks (one per wave): the four column partials are combined in LDS at
  // the end, so YTpart has one row per BLOCK (a quarter of the slab the column-sum kernel has to read back)
  const int rg = blk / nseg;                    // wave-uniform from here on
  const int sg = blk - rg * nseg;
  const int rb = rg * (CA_TB / 64) + wave;
  const bool live = rb < nrb;
  const int col0 = sg * 64 * VEC + lane * VEC;
  float w[VEC][KK], acc[VEC][KK];
#pragma unroll
  for (int j = 0; j < VEC; ++j)
#pragma unroll
    for (int k = 0; k < KK; ++k) {
      const int g = col0 + j;
      const float wv = V[(int64_t)(g < G ? g : G - 1) * Dstride + koff + k];   // unconditional load, masked after
      w[j][k] = g < G ? wv : 0.f;
      acc[j][k] = 0.f;
    }
  const int64_t r0 = live ? (int64_t)rb * TR : 0;
  const int nrows = live ? (int)(((r0 + TR < N) ? r0 + TR : N) - r0) : 0;
  // psi of the strip's rows: row i lives in lane i & 63 of psv[.][i >> 6]   (TR <= 128)
  float psv[KK][2];
#pragma unroll
  for (int k = 0; k < KK; ++k)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int i = lane + 64 * hh;
      const int64_t r = r0 + (i < nrows ? i : (nrows > 0 ? nrows - 1 : 0));
      psv[k][hh] = F[r * Dstride + koff + k];
    }
  const char* base = reinterpret_cast<const char*>(Y) + r0 * (int64_t)Gp * (int64_t)sizeof(YT);   // scalar
  const int voff = col0 * (int)sizeof(YT);                                                          // per lane
  const int64_t pitch = (int64_t)Gp * (int64_t)sizeof(YT);
#ifndef CA_YP_U
#define CA_YP_U 4
#endif
  constexpr int U = CA_YP_U;   // rows per group; two groups alternate (2 x U 16-byte loads in flight per lane)
  auto fetch = [&](uint4 (&buf)[U], int i0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = (i0 + u < nrows) ? i0 + u : nrows - 1;   // tail rows re-read the last row (never consumed)
      {   // Streamed once per pass: NON-TEMPORAL, so the matrix does not push what the sweeps share -- the B operand every sweep
          // block re-reads, coef, the partial slabs -- out of the XCDs' L2.  With default-policy loads the 512 MB of a pass went
          // through 8 x 4 MB of L2: the merged forward launch AND the kernels after it were slower (backward sweep 146 -> 140 us,
          // the small kernels 44 -> 40 us; 3035 -> 3090 it/s at cfg-3, profiles/r03_ab_ystream.txt).
        typedef unsigned v4u_ __attribute__((ext_vector_type(4)));
        const v4u_ t_ = __builtin_nontemporal_load(reinterpret_cast<const v4u_*>(base + (int64_t)i * pitch + voff));
        buf[u] = (uint4){t_.x, t_.y, t_.z, t_.w};
      }
    }
  };
  float keep[KK];
#pragma unroll
  for (int k = 0; k < KK; ++k) keep[k] = 0.f;
  auto consume = [&](const uint4 (&buf)[U], int i0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u;
      if (i < nrows) {   // wave-uniform (scalar branch)
        float y[VEC];
        YVec<YT>::decode(buf[u], y);
        if (TF != 0) {
#pragma unroll
          for (int j = 0; j < VEC; ++j) y[j] = ca_ytf<TF>(y[j]);
        }
#pragma unroll
        for (int k = 0; k < KK; ++k) {
          const float ps = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, i < 64 ? psv[k][0] : psv[k][1]), i & 63));
          float p0 = 0.f, p1 = 0.f;
#pragma unroll
          for (int j = 0; j < VEC; j += 2) {
            p0 = fmaf(y[j], w[j][k], p0);
            p1 = fmaf(y[j + 1], w[j + 1][k], p1);
            acc[j][k] = fmaf(y[j], ps, acc[j][k]);
            acc[j + 1][k] = fmaf(y[j + 1], ps, acc[j + 1][k]);
          }
          const int tot = __builtin_amdgcn_readlane(__builtin_bit_cast(int, ca_wave_sum_lane63(p0 + p1)), 63);
          {   // keep[k] lane (i & 63) <- tot  (v_writelane_b32: value and lane select are both scalars, the select goes through m0)
            const int slot = i & 63;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"   // m0 is "reserved" to the allocator; this kernel has no other m0 user (no LDS-direct loads, no sendmsg)
            asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(keep[k]) : "s"(tot), "s"(slot) : "m0");
#pragma clang diagnostic pop
          }
        }
        if ((i & 63) == 63 || i == nrows - 1) {   // wave-uniform flush of the last (up to 64) row totals
          const int fb = i & ~63;
          if (fb + lane <= i) {
#pragma unroll
            for (int k = 0; k < KK; ++k) YWpart[((int64_t)sg * N + r0 + fb + lane) * K + koff + k] = keep[k];
          }
        }
      }
    }
  };
  uint4 bufA[U], bufB[U];
  if (nrows > 0) fetch(bufA, 0);
  for (int i0 = 0; i0 < nrows; i0 += 2 * U) {
    if (i0 + U < nrows) fetch(bufB, i0 + U);
    consume(bufA, i0);
    if (i0 + 2 * U < nrows) fetch(bufA, i0 + 2 * U);
    if (i0 + U < nrows) consume(bufB, i0 + U);
  }
  // combine the four waves' column partials (fixed order) and write the block's row of YTpart
#pragma unroll
  for (int k = 0; k < KK; ++k) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < VEC; ++j) ycomb[wave][lane][j] = acc[j][k];
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * VEC; i += CA_TB) {
      const int l = i / VEC, j = i - l * VEC;
      const float v = (ycomb[0][l][j] + ycomb[1][l][j]) + (ycomb[2][l][j] + ycomb[3][l][j]);
      YTpart[((int64_t)rg * Gp + sg * 64 * VEC + i) * K + koff + k] = v;
    }
  }
}
template <typename YT, int KK, int TF = 0>
__global__ void __launch_bounds__(CA_TB) k_ypass(const YT* __restrict__ Y, const float* __restrict__ F, int Dstride,
                                                 const float* __restrict__ V, int koff, float* __restrict__ YWpart,
                                                 float* __restrict__ YTpart, int64_t N, int G, int Gp, int nseg,
                                                 int nrb, int TR, int K, ca_ovf_args ovf, int nb_main) {
  __shared__ float ycomb[CA_TB / 64][64][YVec<YT>::VEC + 1];
  ca_ypass_body<YT, KK, TF>((int)blockIdx.x, Y, F, Dstride, V, koff, YWpart, YTpart, N, G, Gp, nseg, nrb, TR, K, ovf, nb_main, ycomb);
}

// Column sums of a [rows][ld] float slab in fp64 and in a fixed order: out[c] = sum_r part[r*ld + c].
// Used for every cross-block reduction of per-gene partials (Y^T.psi strips, backward-sweep cell
// splits).  Block = 64 columns x 16 row lanes (256-byte coalesced row reads), LDS tree combine.
__global__ void __launch_bounds__(1024) k_colsum(const float* __restrict__ part, double* __restrict__ out, int rows,
                                                 int64_t ld, int cols, const int* __restrict__ col_chunk_ptr = nullptr,
                                                 const float* __restrict__ csum = nullptr, int K = 1, int G = 0) {
  constexpr int RL = 16;   // row lanes per column: block = 64 columns x 16 row lanes
  __shared__ double sm[RL][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;   // four independent chains: the loads of a row lane overlap
  if (c < cols) {
    int r = ty;
    for (; r + 3 * RL < rows; r += 4 * RL) {
      const float v0 = part[(int64_t)r * ld + c], v1 = part[(int64_t)(r + RL) * ld + c];
      const float v2 = part[(int64_t)(r + 2 * RL) * ld + c], v3 = part[(int64_t)(r + 3 * RL) * ld + c];
      a0 += (double)v0; a1 += (double)v1; a2 += (double)v2; a3 += (double)v3;
    }
    for (; r < rows; r += RL) a0 += (double)part[(int64_t)r * ld + c];
    a0 += a2; a1 += a3;
    if (csum && ty == 0) {   // gene side of the overflow list (k_ovf_cols folded in): column c = gene * K + k
      const int g = c / K, k = c - g * K;
      if (g < G)
        for (int ch = col_chunk_ptr[g]; ch < col_chunk_ptr[g + 1]; ++ch) a1 += (double)csum[(int64_t)ch * K + k];
    }
  }
  sm[ty][tx] = a0 + a1;
  __syncthreads();
#pragma unroll
  for (int s = RL / 2; s > 0; s >>= 1) {
    if (ty < s) sm[ty][tx] += sm[ty + s][tx];
    __syncthreads();
  }
  if (ty == 0 && c < cols) out[c] = sm[0][tx];
}

// The count-matrix stream's finisher (k_yfinish, K = 1) as EXTRA BLOCKS of the backward sweep's launch instead of a launch of its own
// between the two sweeps (round 3).  Nothing the sweep reads depends on it, what follows the sweep does.
//   column jobs: k_colsum's sums of the Y^T psi slab, ONE WAVE per 64 columns; each lane walks its column's rows in k_colsum's own
//     order -- sixteen row lanes of four chains each, the overflow list's chunk sums on row lane 0, the same pairwise tree -- so the
//     result is bitwise k_colsum's.  No LDS, no barrier.
//   row jobs: k_yw_dot's block of CA_TB cells (YW from the segment shares, the block's share of sum_n psi_n (YW)_n).
struct ca_yfin_args {
  int ncol, nrow;              // 64-column waves, CA_TB-cell blocks (0, 0: none)
  const float* part; double* out; int rows; int64_t ld; int cols;
  const int* col_chunk_ptr; const float* csum; int G;
  const float* YWpart; int nseg; const float* F; int D; int64_t N; float* YW; double* yw_part;
};
__device__ __forceinline__ void ca_yfin_col_wave_few(const ca_yfin_args& a, int job) {
  constexpr int RL = 16;
  const int c = job * 64 + (int)(threadIdx.x & 63);
  if (c >= a.cols) return;
  // (the form for FEW slab rows -- under 64: most row lanes then have no whole trip of four rows, and one lane after the other is the faster order)
  // row lanes in bit-reversed order (0, 8, 4, 12, 2, 10, 6, 14, then the odd ones), eight at a time -- the sweep's register budget --
  // so that each half folds into one subtree of k_colsum's LDS tree: ((x0 + x1) + (x2 + x3)) + ((x4 + x5) + (x6 + x7))
  double half[2];
#pragma unroll 1
  for (int i8 = 0; i8 < 2; ++i8) {
    double x[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int ty = ((k & 1) << 3) | ((k & 2) << 1) | ((k & 4) >> 1) | i8;
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
      int r = ty;
      for (; r + 3 * RL < a.rows; r += 4 * RL) {
        const float v0 = a.part[(int64_t)r * a.ld + c], v1 = a.part[(int64_t)(r + RL) * a.ld + c];
        const float v2 = a.part[(int64_t)(r + 2 * RL) * a.ld + c], v3 = a.part[(int64_t)(r + 3 * RL) * a.ld + c];
        a0 += (double)v0; a1 += (double)v1; a2 += (double)v2; a3 += (double)v3;
      }
      // the (at most three) rows left go to the first chain in order: loaded together, rows past the end skipped
      float t[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int rr = r + i * RL;
        t[i] = a.part[(int64_t)(rr < a.rows ? rr : 0) * a.ld + c];
      }
#pragma unroll
      for (int i = 0; i < 3; ++i)
        if (r + i * RL < a.rows) a0 += (double)t[i];
      a0 += a2; a1 += a3;
      if (ty == 0 && a.csum && c < a.G)
        for (int ch = a.col_chunk_ptr[c]; ch < a.col_chunk_ptr[c + 1]; ++ch) a1 += (double)a.csum[ch];
      x[k] = a0 + a1;
    }
    half[i8] = ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
  }
  a.out[c] = half[0] + half[1];
}
__device__ __forceinline__ void ca_yfin_col_wave(const ca_yfin_args& a, int job) {
  constexpr int RL = 16;
  if (a.rows < 4 * RL) { ca_yfin_col_wave_few(a, job); return; }   // (uniform; same additions in the same order either way -- measured: 12 500 cells, 50 rows: 64.2 vs 65.2 us per iteration)
  const int c = job * 64 + (int)(threadIdx.x & 63);
  if (c >= a.cols) return;
  // row lanes in bit-reversed order (0, 8, 4, 12, 2, 10, 6, 14, then the odd ones), eight at a time, so that each half folds into one subtree
  // of k_colsum's LDS tree: ((x0 + x1) + (x2 + x3)) + ((x4 + x5) + (x6 + x7)).  Round 5: the eight row lanes of a half advance TOGETHER --
  // 32 loads in flight per trip instead of eight chains of 4 one after the other (2 x 8 x 2 dependent rounds of loads at 98 slab rows: on a
  // small shard, where every sweep block of the one resident round ends at the same moment, these trailing blocks run behind the sweep, and
  // at 25 000 cells they were 4.9 us of the backward launch).  Every chain receives the same addends in the same order: the same bits.
  const float* col = a.part + c;
  double half[2];
#pragma unroll 1
  for (int i8 = 0; i8 < 2; ++i8) {
    double acc[8][4];
    int ty[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      ty[k] = ((k & 1) << 3) | ((k & 2) << 1) | ((k & 4) >> 1) | i8;
      acc[k][0] = acc[k][1] = acc[k][2] = acc[k][3] = 0.0;
    }
    // whole trips of four rows per lane: lane ty covers rows ty + 64 t + {0, 16, 32, 48} while the last of them exists
    const int tmax = a.rows > 3 * RL ? (a.rows - 3 * RL - 1) / (4 * RL) + 1 : 0;   // trips of row lane 0 (the longest)
    for (int t = 0; t < tmax; ++t) {
      float v[8][4];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int r = ty[k] + 4 * RL * t;
        const bool ok = r + 3 * RL < a.rows;
        const int rr = ok ? r : 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) v[k][u] = col[(int64_t)(rr + u * RL) * a.ld];
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (ty[k] + 4 * RL * t + 3 * RL < a.rows) {
#pragma unroll
          for (int u = 0; u < 4; ++u) acc[k][u] += (double)v[k][u];
        }
      }
    }
    // the (at most three) rows left of every lane go to its first chain in order: loaded together, rows past the end skipped
    float tl[8][3];
    int rl[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      int r = ty[k];
      while (r + 3 * RL < a.rows) r += 4 * RL;   // (uniform per k: where this lane's whole trips ended)
      rl[k] = r;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int rr = r + i * RL;
        tl[k][i] = col[(int64_t)(rr < a.rows ? rr : 0) * a.ld];
      }
    }
    double x[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
#pragma unroll
      for (int i = 0; i < 3; ++i)
        if (rl[k] + i * RL < a.rows) acc[k][0] += (double)tl[k][i];
      double a0 = acc[k][0] + acc[k][2], a1 = acc[k][1] + acc[k][3];
      if (ty[k] == 0 && a.csum && c < a.G)
        for (int ch = a.col_chunk_ptr[c]; ch < a.col_chunk_ptr[c + 1]; ++ch) a1 += (double)a.csum[ch];
      x[k] = a0 + a1;
    }
    half[i8] = ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
  }
  a.out[c] = half[0] + half[1];
}
__device__ __forceinline__ void ca_yfin_row_block(const ca_yfin_args& a, int blk, double* sm) {
  const int64_t n = (int64_t)blk * CA_TB + threadIdx.x;
  double acc = 0.0;
  if (n < a.N) {
    double yw = 0.0;
    for (int sg = 0; sg < a.nseg; ++sg) yw += (double)a.YWpart[(int64_t)sg * a.N + n];
    a.YW[n] = (float)yw;
    acc += (double)a.F[n * a.D] * yw;
  }
  const double r = ca_block_sum(acc, sm);
  if (threadIdx.x == 0) a.yw_part[blk] = r;
}

// PCA init: scores of one pass, A[n][k] = sum_seg YWpart[seg][n][k] - c[k]
__global__ void k_pca_rows(const float* __restrict__ YWpart, const double* __restrict__ c, float* __restrict__ A, int64_t N, int q,
                           int nseg) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * q) return;
  const int k = (int)(i % q);
  double a = 0.0;
  for (int sg = 0; sg < nseg; ++sg) a += (double)YWpart[(int64_t)sg * N * q + i];
  A[i] = (float)(a - c[k]);
}

// ------------------------------------------------------------------ per-gene preparation of one pass
// x = loc + exp(ls) eps, mu = softplus(x) (R/inference-tflow.R:260-269), M = mu * L (:288), and the
// per-gene ELBO terms of :322-323,332 reduced per block (W_ = 3 + K slots per block):
//   [0] sum_g (1/S) sum_s colsum_g log mu_sg  + sum_p beta_gp (Y^T X)_gp      (part of EE_p_y)
//   [1] sum_g (1/S) sum_s Normal(log mu_sg; 0, 1)                             (part of E_log_p_p)
//   [2] sum_g (1/S) sum_s log q(mu_sg)                                        (part of E_log_q)
//   [3+k] sum_g W_gk^2
__global__ void __launch_bounds__(CA_TB) k_gene_pre(const float* __restrict__ loc, const float* __restrict__ ls,
                                                    const float* __restrict__ eps /*[S][G]*/,
                                                    const double* __restrict__ colsum, const float* __restrict__ Lb /*[nchunk][G][8]*/,
                                                    const float* __restrict__ V, int D, int K, const double* __restrict__ YtX,
                                                    float* __restrict__ mu32 /*[S][G]*/, float* __restrict__ Mb /*[S][nchunk][G][mrow]*/,
                                                    double* __restrict__ gene_part, int G, int S, int nchunk, int mrow, int mcol, int ncol) {
  __shared__ double sm[CA_TB];
  const int g = blockIdx.x * CA_TB + threadIdx.x;
  const bool ok = g < G;
  double t0 = 0.0, t1 = 0.0, t2 = 0.0;
  if (ok) {
    const double l = (double)loc[g], sd = exp((double)ls[g]), lsd = (double)ls[g];
    const double cs = colsum[g];
    for (int s = 0; s < S; ++s) {
      const double e = (double)eps[(int64_t)s * G + g];
      const double x = l + sd * e;
      const double mu = ca_softplus_d(x);
      const double lm = log(mu);
      const float muf = (float)mu;
      mu32[(int64_t)s * G + g] = muf;
      for (int ch = 0; ch < nchunk; ++ch) {
        // row stride mrow / column offset mcol / ncol columns: 8/0/8 normally; the fused two-eps sweep packs
        // [mu_A L | mu_B L] into one row
        const float* lp = Lb + ((int64_t)ch * G + g) * CA_CW;
        float* mp = Mb + (((int64_t)s * nchunk + ch) * G + g) * mrow + mcol;
        for (int c = 0; c < ncol; ++c) mp[c] = lp[c] * muf;
      }
      t0 += cs * lm;
      t1 += -0.5 * lm * lm - 0.5 * CA_LOG2PI;
      // log q(mu) = Normal(x; loc, sd) + softplus(-x),  softplus(-x) = softplus(x) - x
      t2 += -0.5 * e * e - lsd - 0.5 * CA_LOG2PI + (mu - x);
    }
    t0 /= (double)S; t1 /= (double)S; t2 /= (double)S;
    for (int p = K; p < D; ++p) t0 += (double)V[(int64_t)g * D + p] * YtX[(int64_t)g * (D - K) + (p - K)];
  }
  const int W_ = 3 + K;
  const double s0 = ca_block_sum(t0, sm);
  const double s1 = ca_block_sum(t1, sm);
  const double s2 = ca_block_sum(t2, sm);
  if (threadIdx.x == 0) {
    gene_part[(int64_t)blockIdx.x * W_ + 0] = s0;
    gene_part[(int64_t)blockIdx.x * W_ + 1] = s1;
    gene_part[(int64_t)blockIdx.x * W_ + 2] = s2;
  }
  for (int k = 0; k < K; ++k) {
    const double w = ok ? (double)V[(int64_t)g * D + k] : 0.0;
    const double wsum = ca_block_sum(w * w, sm);
    if (threadIdx.x == 0) gene_part[(int64_t)blockIdx.x * W_ + 3 + k] = wsum;
  }
}

// Plain passes on the matrix cores (round 6): the float rows of a pass's (sample, clone chunk) slices, Mb [nslice][G][8], as two bf16 parts in the B-operand
// layout of k_fwd_mfma, two slices to a sixteen-column image: Mq [pair][g / 32][part][16 (g % 32) / 8 + 8 (slice & 1) + column][g % 8].  Padding genes and a
// missing last slice are never written (the image is zero-filled once).
__global__ void __launch_bounds__(CA_TB) k_mq_pairs(const float* __restrict__ Mb, unsigned short* __restrict__ Mq, int G, int nk) {
  const int g = blockIdx.x * CA_TB + threadIdx.x, j = blockIdx.y;
  if (g >= G) return;
  const float4 r0 = *reinterpret_cast<const float4*>(Mb + ((int64_t)j * G + g) * CA_CW), r1 = *reinterpret_cast<const float4*>(Mb + ((int64_t)j * G + g) * CA_CW + 4);
  const float x[CA_CW] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
  unsigned short* mq = Mq + (int64_t)(j >> 1) * nk * 1024 + ((int64_t)(g >> 5) * 128 + 16 * ((g & 31) >> 3) + 8 * (j & 1)) * 8 + (g & 7);
#pragma unroll
  for (int c = 0; c < CA_CW; ++c) {
    const unsigned short p1 = ca_bf16_rn(x[c]);
    mq[c * 8] = p1;
    mq[(64 + c) * 8] = ca_bf16_rn(x[c] - __uint_as_float((unsigned)p1 << 16));
  }
}

// Fused two-eps variant (S == 1, one clone chunk): both draws A (monitor pass) and B (next train pass) in one
// launch; M row = [mu_A L (C cols) | mu_B L (C cols)], per-draw mu and gene partials kept apart.  With Mq the row
// goes out as two bf16 parts in the operand layout of the matrix-core sweep instead (k_fwd_mfma).
struct ca_gene_pre_ops { float loc, ls, eA, eB, wk0; double cs; float4 lr0, lr1; };   // one gene's operands of the prologue
__device__ __forceinline__ void ca_gene_pre_fused_core(const ca_gene_pre_ops& o, const float* __restrict__ Lb, const float* __restrict__ V, int D, int K,
                                                          const double* __restrict__ YtX, float* __restrict__ muA, float* __restrict__ muB,
                                                          float* __restrict__ Mb, double* __restrict__ gene_partA, double* __restrict__ gene_partB, int G,
                                                          int mrow, int C, unsigned short* __restrict__ Mq, double* sm, int blk, int s2);
__device__ __forceinline__ void ca_gene_pre_fused_body(const float* __restrict__ loc, const float* __restrict__ ls,
                                                          const float* __restrict__ epsA, const float* __restrict__ epsB,
                                                          const double* __restrict__ colsum, const float* __restrict__ Lb,
                                                          const float* __restrict__ V, int D, int K, const double* __restrict__ YtX,
                                                          float* __restrict__ muA, float* __restrict__ muB, float* __restrict__ Mb,
                                                          double* __restrict__ gene_partA, double* __restrict__ gene_partB, int G,
                                                          int mrow, int C, unsigned short* __restrict__ Mq, double* sm, int blk, int s2 = 0) {
  // s2 (round 3): the two "draws" are the two SAMPLES of one pass with mc_samples = 2 (R/inference-tflow.R:268-269, :306-308): the
  // per-gene terms of the ELBO are then their mean (as k_gene_pre leaves them), in gene_partA
  const int g = blk * CA_TB + threadIdx.x;
  const bool ok = g < G;
  ca_gene_pre_ops o = {0.f, 0.f, 0.f, 0.f, 0.f, 0.0, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  if (ok) {
    // all operands in one batch in front of the first use (one wave per SIMD here: a dependent round of loads is 1.5 us)
    o.loc = loc[g]; o.ls = ls[g]; o.eA = epsA[g]; o.eB = epsB[g];
    o.cs = colsum[g];
    o.lr0 = *reinterpret_cast<const float4*>(Lb + (int64_t)g * CA_CW); o.lr1 = *reinterpret_cast<const float4*>(Lb + (int64_t)g * CA_CW + 4);
    if (K > 0) o.wk0 = V[(int64_t)g * D];
  }
  ca_gene_pre_fused_core(o, Lb, V, D, K, YtX, muA, muB, Mb, gene_partA, gene_partB, G, mrow, C, Mq, sm, blk, s2);
}
// ... the same on operands that are in registers already: the merged update (k_update_merged) goes from a gene's Adam step straight on to
// the next eps pair's prologue for that gene -- no second kernel, no reload of loc / ls / W (same arithmetic on the same floats: bitwise the same)
// one draw of the prologue for one gene (w = 0: the monitor pass's eps, 1: the next train pass's): mu, its row of the sweep's B operand,
// the gene's three ELBO terms.  Per lane, no block-level operation (the merged update gives the two draws to two waves).
__device__ __forceinline__ void ca_gene_pre_draw(int w, int g, const ca_gene_pre_ops& o, const float* __restrict__ Lb, const float* __restrict__ V, int D, int K,
                                                    const double* __restrict__ YtX, float* __restrict__ muA, float* __restrict__ muB,
                                                    float* __restrict__ Mb, int G, int mrow, int C, unsigned short* __restrict__ Mq, double (&t)[3],
                                                    double* __restrict__ aux = nullptr, int64_t aux_ld = 0) {
    const float loc_g = o.loc, ls_g = o.ls, eA = o.eA, eB = o.eB;
    const double cs = o.cs;
    const float4 lr0 = o.lr0, lr1 = o.lr1;
    const float lrow[CA_CW] = {lr0.x, lr0.y, lr0.z, lr0.w, lr1.x, lr1.y, lr1.z, lr1.w};
    const double l = (double)loc_g, lsd = (double)ls_g, sd = exp(lsd);
    double bx = 0.0;
    for (int p = K; p < D; ++p) bx += (double)V[(int64_t)g * D + p] * YtX[(int64_t)g * (D - K) + (p - K)];
    const float* lp = lrow;
    const bool c16 = C > CA_CW;   // 9..16 clones: ONE draw per sweep, its clones 8.. in the second column half (copy numbers: second chunk of Lb)
    {
      const double e = (double)(w ? eB : eA);
      const double x = l + sd * e;
      // softplus with its exp kept: t = exp(-|x|), softplus = max(x, 0) + log1p(t) -- the very doubles ca_softplus_d(x) gives
      // (x > 0: x + log1p(exp(-x)); else 0 + log1p(exp(x))), and t is what the sigmoid below wants
      const double tx = exp(-fabs(x));
      const double mu = (x > 0 ? x : 0.0) + log1p(tx), lm = log(mu);
      const float muf = (float)mu;
      (w ? muB : muA)[g] = muf;
      if (aux) {
        // Round 4: this draw is the eps of the NEXT train pass, and everything in that pass's per-gene gradient that does not depend on
        // the backward sweep is known here: exp(ls), the sigmoid, cs / mu, log(mu) / mu, (1 - sigmoid) -- ca_final_gene_step's own
        // expressions (S = 1), kept as doubles so that the step after the sweep is a load, three additions and the Adam arithmetic
        // instead of an fp64 exp / log1p / log / four divisions chain on the iteration's critical path
        const double sig = (x >= 0 ? 1.0 : tx) / (1.0 + tx);
        aux[g] = sd; aux[aux_ld + g] = sig; aux[2 * aux_ld + g] = cs / (1.0 * mu); aux[3 * aux_ld + g] = lm / (1.0 * mu);
        aux[4 * aux_ld + g] = (1.0 - sig) / 1.0;
      }
      if (Mq && c16) {   // sixteen columns per draw: the second draw's image follows the first one's ([2][nk][2][64][8])
        unsigned short* mq = Mq + (int64_t)w * ((G + 31) / 32) * 1024 + ((int64_t)(g >> 5) * 128 + 16 * ((g & 31) >> 3)) * 8 + (g & 7);
#pragma unroll
        for (int c = 0; c < 2 * CA_CW; ++c) {   // (compile-time indices: the copy-number row is in registers)
          if (c < C) {
            const float x = (c < CA_CW ? lp[c < CA_CW ? c : 0] : Lb[((int64_t)G + g) * CA_CW + (c - CA_CW)]) * muf;
            const unsigned short p1 = ca_bf16_rn(x);
            mq[c * 8] = p1;
            mq[(64 + c) * 8] = ca_bf16_rn(x - __uint_as_float((unsigned)p1 << 16));
          }
        }
      } else if (Mq) {   // two bf16 parts in the B-operand layout of k_fwd_mfma: [g / 32][part][16 (g % 32) / 8 + column][g % 8]
        unsigned short* mq = Mq + ((int64_t)(g >> 5) * 128 + 16 * ((g & 31) >> 3) + w * C) * 8 + (g & 7);
#pragma unroll
        for (int c = 0; c < CA_CW; ++c) {
          if (c < C) {
            const float x = lp[c] * muf;
            const unsigned short p1 = ca_bf16_rn(x);
            mq[c * 8] = p1;
            mq[(64 + c) * 8] = ca_bf16_rn(x - __uint_as_float((unsigned)p1 << 16));
          }
        }
      } else {
        float* mp = Mb + (int64_t)g * mrow + w * C;
#pragma unroll
        for (int c = 0; c < CA_CW; ++c)
          if (c < C) mp[c] = lp[c] * muf;
      }
      t[0] = cs * lm + bx;
      t[1] = -0.5 * lm * lm - 0.5 * CA_LOG2PI;
      t[2] = -0.5 * e * e - lsd - 0.5 * CA_LOG2PI + (mu - x);
    }
}
__device__ __forceinline__ void ca_gene_pre_fused_core(const ca_gene_pre_ops& o, const float* __restrict__ Lb, const float* __restrict__ V, int D, int K,
                                                          const double* __restrict__ YtX, float* __restrict__ muA, float* __restrict__ muB,
                                                          float* __restrict__ Mb, double* __restrict__ gene_partA, double* __restrict__ gene_partB, int G,
                                                          int mrow, int C, unsigned short* __restrict__ Mq, double* sm, int blk, int s2) {
  const int g = blk * CA_TB + threadIdx.x;
  const bool ok = g < G;
  double t[2][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
  const float wk0 = o.wk0;   // W_g0, for the sum of squares below
  if (ok) {
    ca_gene_pre_draw(0, g, o, Lb, V, D, K, YtX, muA, muB, Mb, G, mrow, C, Mq, t[0]);
    ca_gene_pre_draw(1, g, o, Lb, V, D, K, YtX, muA, muB, Mb, G, mrow, C, Mq, t[1]);
  }
  const int W_ = 3 + K;
  // the six term sums and (up to two) sums of squared loadings in ONE pass through the block reduction: one pair of barriers
  double w1sq = 0.0;
  if (K > 1 && ok) { const double w1 = (double)V[(int64_t)g * D + 1]; w1sq = w1 * w1; }
  {
    double eight[8] = {t[0][0], t[0][1], t[0][2], t[1][0], t[1][1], t[1][2], ok ? (double)wk0 * (double)wk0 : 0.0, w1sq};
    ca_block_sum_n<8>(eight, sm);
    if (threadIdx.x == 0) {
      double* ga = gene_partA + (int64_t)blk * W_;
      double* gb = gene_partB + (int64_t)blk * W_;
      if (s2) { ga[0] = 0.5 * (eight[0] + eight[3]); ga[1] = 0.5 * (eight[1] + eight[4]); ga[2] = 0.5 * (eight[2] + eight[5]); }
      else { ga[0] = eight[0]; ga[1] = eight[1]; ga[2] = eight[2]; }
      gb[0] = eight[3]; gb[1] = eight[4]; gb[2] = eight[5];
      for (int k = 0; k < K && k < 2; ++k) { ga[3 + k] = eight[6 + k]; gb[3 + k] = eight[6 + k]; }
    }
  }
  for (int k = 2; k < K; ++k) {
    const double wv = ok ? (double)V[(int64_t)g * D + k] : 0.0;
    const double wsum = ca_block_sum(wv * wv, sm);
    if (threadIdx.x == 0) {
      gene_partA[(int64_t)blk * W_ + 3 + k] = wsum;
      gene_partB[(int64_t)blk * W_ + 3 + k] = wsum;
    }
  }
}

__global__ void __launch_bounds__(CA_TB) k_gene_pre_fused(const float* __restrict__ loc, const float* __restrict__ ls,
                                                          const float* __restrict__ epsA, const float* __restrict__ epsB,
                                                          const double* __restrict__ colsum, const float* __restrict__ Lb,
                                                          const float* __restrict__ V, int D, int K, const double* __restrict__ YtX,
                                                          float* __restrict__ muA, float* __restrict__ muB, float* __restrict__ Mb,
                                                          double* __restrict__ gene_partA, double* __restrict__ gene_partB, int G,
                                                          int mrow, int C, unsigned short* __restrict__ Mq, int s2) {
  __shared__ double sm[CA_TB];
  ca_gene_pre_fused_body(loc, ls, epsA, epsB, colsum, Lb, V, D, K, YtX, muA, muB, Mb, gene_partA, gene_partB, G, mrow, C, Mq, sm, blockIdx.x, s2);
}
// the same per-gene prologue for the NEXT (monitor, train) eps pair, as extra blocks of the per-cell kernel of a train pass
// (k_adam_cell): the per-gene variables are final once k_final_gene has run, so the following fused pass starts at its sweep
struct ca_pre_args {
  int nblk;   // 0: none
  const float* loc; const float* ls; const float* epsA; const float* epsB; const double* colsum; const float* Lb; const float* V;
  const double* YtX; float* muA; float* muB; float* Mb; double* gene_partA; double* gene_partB; unsigned short* Mq;
  int G, D, K, mrow, C, s2;
};
// ca_run's gate, per lane (round 4): a block of the gated update does its loads and its arithmetic first and asks HERE, right before its first
// store, whether the launch goes on.  Every lane of a wave reads the same word with the same instruction, so the lanes agree without talking.
// Round 5: the word is the RELAY block's verdict in device memory (ca_gate_wait) -- go, or "store nothing" (the host said stop, or the host did
// not answer within the relay's short deadline) -- and ONLY the relay decides: a waiter's own deadline (`timeout`, the relay's plus ten
// seconds) can run out only if the relay block never ran, which the block order rules out (it is dispatched first); it then reports a
// fatal error (`err`), the one case the host cannot recover from.
struct ca_gate { const unsigned long long* word; unsigned long long seq, timeout; unsigned long long* err; };   // word = null: no gate
__device__ __forceinline__ bool ca_gate_spin(const ca_gate& gt) {
  if (!gt.word) return true;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (;;) {
    const unsigned long long w = __hip_atomic_load(gt.word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((w >> 1) == gt.seq) return (w & 1ull) != 0ull;
    if (__builtin_amdgcn_s_memrealtime() - t0 > gt.timeout) {
      __hip_atomic_store(gt.err, gt.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return false;
    }
    __builtin_amdgcn_s_sleep(8);
  }
}
// psi's gradient and Adam step, as extra blocks of the per-gene kernel (k_final_gene): psi is all the Y stream needs, so the
// side stream can start on the next pass's Y kernel while the main stream is still updating the q(z) logits
struct ca_psi_args {
  int nblk;   // 0: none
  float* F; const float* YW; const float* dFpart; float* m_psi; float* v_psi; float* g_psi;
  int64_t N; int D, K, ntile;
};

// Vs = V * log2(e) and per-block min/max of each column (for the per-cell exponent bound)
__global__ void __launch_bounds__(CA_TB) k_vprep(const float* __restrict__ V, float* __restrict__ Vs,
                                                 float* __restrict__ vmm_part /*[nblk][2][D]*/, int G, int D) {
  __shared__ float smin[CA_TB], smax[CA_TB];
  const int g = blockIdx.x * CA_TB + threadIdx.x;
  for (int d = 0; d < D; ++d) {
    float v = 0.f;
    if (g < G) {
      v = V[(int64_t)g * D + d] * CA_LOG2E_F;
      Vs[(int64_t)g * D + d] = v;
      if (g == G - 1)   // pad to a multiple of 32 genes with the last gene's loading (k_fwd_cell reads whole k-steps)
        for (int gp = G; gp < ((G + 31) / 32) * 32; ++gp) Vs[(int64_t)gp * D + d] = v;
    }
    __syncthreads();
    smin[threadIdx.x] = (g < G) ? v : INFINITY;
    smax[threadIdx.x] = (g < G) ? v : -INFINITY;
    __syncthreads();
    for (int s = CA_TB / 2; s > 0; s >>= 1) {
      if (threadIdx.x < s) {
        smin[threadIdx.x] = fminf(smin[threadIdx.x], smin[threadIdx.x + s]);
        smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + s]);
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      vmm_part[((int64_t)blockIdx.x * 2 + 0) * D + d] = smin[0];
      vmm_part[((int64_t)blockIdx.x * 2 + 1) * D + d] = smax[0];
    }
  }
}

__global__ void k_vmm_final(const float* __restrict__ vmm_part, float* __restrict__ vmm /*[2][D]*/, int nblk, int D) {
  const int d = threadIdx.x;
  if (d >= D) return;
  float mn = INFINITY, mx = -INFINITY;
  for (int b = 0; b < nblk; ++b) {
    mn = fminf(mn, vmm_part[((int64_t)b * 2 + 0) * D + d]);
    mx = fmaxf(mx, vmm_part[((int64_t)b * 2 + 1) * D + d]);
  }
  vmm[d] = mn;
  vmm[D + d] = mx;
}

// etamax2_n = sum_d max(F_nd * Vs_min_d, F_nd * Vs_max_d)  >=  max_g log2 E_ng
__global__ void k_etamax(const float* __restrict__ F, const float* __restrict__ vmm, float* __restrict__ etamax2,
                         int64_t N, int D) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float e = 0.f;
  for (int d = 0; d < D; ++d) {
    const float f = F[n * D + d];
    e += fmaxf(f * vmm[d], f * vmm[D + d]);
  }
  etamax2[n] = e;
}

// ------------------------------------------------------------------ forward sweep  Z = E . M
// (R/inference-tflow.R:278-292 without materialising [S,G,C,N]).  VALU form: lane = R cells, loop over a slice of genes,
// 1 fma (exponent) + v_exp_f32 + NC fma per (n,g).  A first version fetched M_g and V'_g through the scalar cache
// (158 us); this one copies the block's gene slice of M (and V') into LDS once and every lane sweeps it -- broadcast
// ds_read_b128, R independent exp chains per lane: 132-137 us at 100k x 5k x 8 (tools/fwd_lab.hip).
template <int NC, int D, int R, int CWS = CA_CW>
__global__ void __launch_bounds__(CA_TB) k_fwd_lds(const float* __restrict__ F, const float* __restrict__ etamax2,
                                                   const float* __restrict__ Vs, const float* __restrict__ M /*[G][CWS]*/,
                                                   float* __restrict__ Zpart /*[gsplit][N][CWS]*/, int64_t N, int G,
                                                   int gchunk, int Drt) {
  constexpr int DM = (D < 0) ? 8 : (D > 0 ? D : 1);
  const int Dn = (D < 0) ? Drt : D;
  extern __shared__ float ca_lds[];  // [gchunk][CWS] M slice, then [gchunk][Dn] V' slice
  const int g0 = blockIdx.y * gchunk;
  const int ng = ((g0 + gchunk < G) ? g0 + gchunk : G) - g0;
  float4* l4 = reinterpret_cast<float4*>(ca_lds);
  const float4* m4 = reinterpret_cast<const float4*>(M + (int64_t)g0 * CWS);
  for (int i = threadIdx.x; i < ng * (CWS / 4); i += CA_TB) l4[i] = m4[i];
  float* lv = ca_lds + (int64_t)gchunk * CWS;
  for (int i = threadIdx.x; i < ng * Dn; i += CA_TB) lv[i] = Vs[(int64_t)g0 * Dn + i];
  __syncthreads();
  const int64_t nb = (int64_t)blockIdx.x * CA_TB * R + threadIdx.x;
  float f[R][DM], em[R], z[R][NC];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int64_t n = nb + r * CA_TB;
    const int64_t nn = n < N ? n : N - 1;
#pragma unroll
    for (int d = 0; d < DM; ++d) f[r][d] = (d < Dn) ? F[nn * Dn + d] : 0.f;
    em[r] = (Dn > 0) ? etamax2[nn] : 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) z[r][c] = 0.f;
  }
#pragma unroll 4
  for (int g = 0; g < ng; ++g) {
    float m[CWS];
#pragma unroll
    for (int j = 0; j < CWS / 4; ++j) {
      const float4 a = l4[(CWS / 4) * g + j];
      m[4 * j] = a.x; m[4 * j + 1] = a.y; m[4 * j + 2] = a.z; m[4 * j + 3] = a.w;
    }
    float v[DM];
#pragma unroll
    for (int d = 0; d < DM; ++d) v[d] = (d < Dn) ? lv[g * Dn + d] : 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float e = 1.f;
      if (Dn > 0) {
        float eta = -em[r];
#pragma unroll
        for (int d = 0; d < DM; ++d) eta = fmaf(f[r][d], v[d], eta);
        e = __builtin_amdgcn_exp2f(eta);
      }
#pragma unroll
      for (int c = 0; c < NC; ++c) z[r][c] = fmaf(e, m[c], z[r][c]);
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int64_t n = nb + r * CA_TB;
    if (n < N) {
      float* zp = Zpart + ((int64_t)blockIdx.y * N + n) * CWS;
#pragma unroll
      for (int c = 0; c < NC; ++c) zp[c] = z[r][c];
    }
  }
}

// ------------------------------------------------------------------ forward sweep on the matrix cores
// Z = E.M for the 16 columns of the fused two-eps pass as bf16 MFMAs with fp32 accumulation:
//   rows = 16 cells, k = 32 genes, columns = [mu_A L | mu_B L | 0]   (v_mfma_f32_16x16x32_bf16)
// A lane owns ONE cell and 8 consecutive genes of the k-step (the A-operand layout), so it generates its 8 E values,
// rounds them to bf16 (hi, v_cvt_pk_bf16_f32), takes the exact remainder e - hi with v_dot2c_f32_bf16 and rounds
// that too (lo): E = hi + lo up to 2^-18.  M arrives pre-split the same way (Mq, written by k_gene_pre_fused in the
// B-operand layout) and Z += lo.M1 + hi.M2 + hi.M1 -- the dropped terms are <= 3 x 2^-18 relative per product with
// random sign; measured against float64 the result is as accurate as the fp32 VALU chain (tools/fwd_mfma_lab.hip:
// rms 1.4e-7 vs 1.3e-7).  The gene slice of a block streams through LDS in double-buffered chunks of KC k-steps, so
// few slices suffice (fewer Z partials for the cell epilogue to re-read).  Issue-bound: 20 VALU + 8 v_exp_f32 +
// 3 MFMA per 8 genes x 16 columns per lane (tools/inst_lab.hip: 220 cycles per wave and k-step-tile) against
// 8 + 8 v_exp + 64 for the VALU kernel.  D in {1, 2}; k_fwd_lds is the general fallback.
typedef __bf16 ca_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 ca_bf16x2 __attribute__((ext_vector_type(2)));
typedef float ca_f32x4 __attribute__((ext_vector_type(4)));
typedef float ca_f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned ca_pk_bf16(float a, float b) {   // v_cvt_pk_bf16_f32, round to nearest even
  const ca_f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, ca_bf16x2));
}

constexpr int CA_FM_TL = 4;   // 16-cell tiles per wave  -> 256 cells per block
constexpr int CA_FM_KC = 4;   // k-steps (of 32 genes) per LDS chunk

template <int D>
__global__ void __launch_bounds__(CA_TB) k_fwd_mfma(const float* __restrict__ F, const float* __restrict__ etamax2,
                                                    const float* __restrict__ Vs /*[G][D]*/,
                                                    const unsigned short* __restrict__ Mq /*[nk][2][64][8] bf16*/,
                                                    float* __restrict__ Zpart /*[fsplit][N][16]*/, int64_t N, int G, int kchunk,
                                                    int nk) {
  constexpr int TL = CA_FM_TL, KC = CA_FM_KC;
  constexpr int NB = KC * 128;            // uint4 of B per chunk (2 parts x 64 lanes per k-step)
  constexpr int NV = KC * 32 * D;         // floats of V' per chunk, [ks][d][32]
  constexpr int BUF = NB + NV / 4;        // uint4 per buffer
  constexpr int NLD = NB / CA_TB;
  static_assert(NB % CA_TB == 0 && NV <= CA_TB, "chunk shape");
  __shared__ uint4 lds[2 * BUF];
  const int k0 = blockIdx.y * kchunk;
  const int nks = min(nk, k0 + kchunk) - k0;
  const int nch = (nks + KC - 1) / KC;
  const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4, wv = threadIdx.x >> 6;
  const int64_t cell0 = ((int64_t)blockIdx.x * (CA_TB / 64) + wv) * (TL * 16);
  uint4 st[NLD];
  float sv = 0.f;
  auto gload = [&](int c) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = threadIdx.x + CA_TB * i;   // [ks][part][lane]
      const int kk = k0 + c * KC + (idx >> 7);
      st[i] = (kk < k0 + nks) ? reinterpret_cast<const uint4*>(Mq)[(int64_t)kk * 128 + (idx & 127)] : (uint4){0u, 0u, 0u, 0u};
    }
    if (threadIdx.x < NV) {
      const int ks = threadIdx.x / (32 * D), rem = threadIdx.x % (32 * D), d = rem / 32, gi = rem % 32;
      // padding genes (M = 0 there) borrow the last real gene's loadings: their exponent then stays <= 0 like every
      // real one (eta - etamax), where V' = 0 would give 2^(-etamax) -- inf x 0 for a cell with etamax < -128
      const int kk = k0 + c * KC + ks, g = min(kk * 32 + gi, G - 1);
      sv = Vs[(int64_t)g * D + d];
    }
  };
  auto lstore = [&](int b) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) lds[b * BUF + threadIdx.x + CA_TB * i] = st[i];
    if (threadIdx.x < NV) reinterpret_cast<float*>(lds + b * BUF + NB)[threadIdx.x] = sv;
  };
  float f[TL][D], em[TL];
  ca_f32x4 acc[TL];
#pragma unroll
  for (int t = 0; t < TL; ++t) {
    const int64_t n = cell0 + 16 * t + j;
    const int64_t nn = n < N ? n : N - 1;
#pragma unroll
    for (int d = 0; d < D; ++d) f[t][d] = F[nn * D + d];
    em[t] = etamax2[nn];
    acc[t] = (ca_f32x4){0.f, 0.f, 0.f, 0.f};
  }
  // (-1, 0) and (0, -1) as bf16 pairs for v_dot2c_f32_bf16; kept out of the compiler's sight, which would turn them
  // into the fp32 inline constant -1.0 (wrong half of the pair)
  unsigned m0, m1;
  asm volatile("s_mov_b32 %0, 0x0000bf80" : "=s"(m0));
  asm volatile("s_mov_b32 %0, 0xbf800000" : "=s"(m1));
  const ca_bf16x2 neg_lo = __builtin_bit_cast(ca_bf16x2, m0), neg_hi = __builtin_bit_cast(ca_bf16x2, m1);
  gload(0);
  lstore(0);
  __syncthreads();
  for (int c = 0; c < nch; ++c) {
    const int b = c & 1;
    if (c + 1 < nch) gload(c + 1);
    const uint4* lb = lds + b * BUF;
    const float4* lv4 = reinterpret_cast<const float4*>(lds + b * BUF + NB);
#pragma unroll 2
    for (int ks = 0; ks < KC; ++ks) {
      const uint4 b1r = lb[ks * 128 + lane], b2r = lb[ks * 128 + 64 + lane];
      const ca_bf16x8 B1 = __builtin_bit_cast(ca_bf16x8, b1r), B2 = __builtin_bit_cast(ca_bf16x8, b2r);
      ca_f32x2 v2[D][4];   // V'_d of this lane's 8 genes
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const float4 va = lv4[(ks * D + d) * 8 + 2 * q], vb = lv4[(ks * D + d) * 8 + 2 * q + 1];
        v2[d][0] = (ca_f32x2){va.x, va.y}; v2[d][1] = (ca_f32x2){va.z, va.w};
        v2[d][2] = (ca_f32x2){vb.x, vb.y}; v2[d][3] = (ca_f32x2){vb.z, vb.w};
      }
#pragma unroll
      for (int t = 0; t < TL; ++t) {
        unsigned hi[4], lo[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          ca_f32x2 eta = v2[0][p] * f[t][0] - em[t];
#pragma unroll
          for (int d = 1; d < D; ++d) eta = v2[d][p] * f[t][d] + eta;
          const float e0 = __builtin_amdgcn_exp2f(eta.x), e1 = __builtin_amdgcn_exp2f(eta.y);
          hi[p] = ca_pk_bf16(e0, e1);
          const ca_bf16x2 hb = __builtin_bit_cast(ca_bf16x2, hi[p]);
          const float r0 = __builtin_amdgcn_fdot2_f32_bf16(hb, neg_lo, e0, false);   // e0 - hi.lo, exact
          const float r1 = __builtin_amdgcn_fdot2_f32_bf16(hb, neg_hi, e1, false);
          lo[p] = ca_pk_bf16(r0, r1);
        }
        const ca_bf16x8 A1 = __builtin_bit_cast(ca_bf16x8, ((uint4){hi[0], hi[1], hi[2], hi[3]}));
        const ca_bf16x8 A2 = __builtin_bit_cast(ca_bf16x8, ((uint4){lo[0], lo[1], lo[2], lo[3]}));
        ca_f32x4 a = acc[t];
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A2, B1, a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1, B2, a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1, B1, a, 0, 0, 0);
        acc[t] = a;
      }
    }
    if (c + 1 < nch) lstore(b ^ 1);
    __syncthreads();
  }
  // accumulator layout: lane (column j, rows 4q .. 4q+3 of the tile)
#pragma unroll
  for (int t = 0; t < TL; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t n = cell0 + 16 * t + 4 * q + r;
      if (n < N) Zpart[((int64_t)blockIdx.y * N + n) * 16 + j] = acc[t][r];
    }
}


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
#ifndef CA_BWD_PD
#define CA_BWD_PD 2   // batches of operands in flight per wave (3 and more cost the third wave per SIMD: 140 -> 200 us)
#endif
// FRAC (round 3): copy numbers that are not bf16-exact (clonealign() accepts any non-negative matrix; saturate() only caps it at 6,
// R/clonealign.R:394-397).  L is then split in two bf16 parts like M in the forward sweep, coef in two, and the 24 operand slots
// carry [c1 L_hi | c2 L_hi | c1 L_lo]: what is dropped (c2 L_lo, and the third part of coef) is below 2^-17 of the product -- the
// forward sweep's own accuracy.  Integer copy numbers keep the exact three-part form.
// C16 (round 3): 9..16 clones with integer copy numbers.  The 32 operand slots carry two bf16 parts of coef for sixteen clones,
// slot group q = 2 * part + chunk (what the sixteen-lane cell epilogue writes), against L of clone chunk q & 1 in both parts.
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
  [[maybe_unused]] const unsigned short* cqb1 = S2 ? cq1 + n0 * 32 : nullptr;
  const float* Fb = F + n0 * DD;
  const float* eb = etamax2 + n0;
  const unsigned lo_c = (unsigned)((j * 4 + qc) * 8), lo_f = (unsigned)(j * DD), lo_e = (unsigned)j;
  const int jl = len - j;                    // cell r + j is inside the slice iff r < jl
  float* myd_lane = myd + j * DD;
  uint4 craw_r[NSM][PD];
  float fc_r[PD][DD], ec_r[PD];
  auto fetch = [&](int slot, int r) {        // r: uniform, a multiple of 16, inside the padded arrays
    const unsigned short* pc = cqb + (int64_t)r * 32;
    const float* pf = Fb + (int64_t)r * DD;
    const float* pe = eb + r;
    craw_r[0][slot] = *reinterpret_cast<const uint4*>(pc + lo_c);
    if constexpr (S2) craw_r[1][slot] = *reinterpret_cast<const uint4*>(cqb1 + (int64_t)r * 32 + lo_c);
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
        tt[sm_][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Lf[m], Cf[sm_], tt[sm_][m], 0, 0, 0);   // tt[.][m][r]: gene gbase+16m+4q+r, cell n0+r0+j
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < TL; ++m) {
      if constexpr (AH < TL) {
        if (m + AH < TL) {
#pragma unroll
          for (int sm_ = 0; sm_ < NSM; ++sm_) {
            tt[sm_][m + AH] = (ca_f32x4){0.f, 0.f, 0.f, 0.f};
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
                                                    unsigned short* __restrict__ coefq, int64_t N16) {
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
      for (int sp = 0; sp < gsplit; ++sp) Z += (double)Zpart[((((int64_t)s * nchunk + ch) * gsplit + sp) * N + nn) * CA_CW + cc];
      lzsum += log(Z) + em;
      if (mode == CA_MODE_TRAIN && ok) {
        const float cfv = (float)(-gam * sn / ((double)S * Z));
        coef[(((int64_t)s * nchunk + ch) * N + nn) * CA_CW + cc] = cfv;
        if (coefq) {   // bf16 parts for the matrix-core backward sweep: three for up to eight clones, two per clone chunk for 9..16
          unsigned short p1, p2, p3;
          ca_split3(cfv, p1, p2, p3);
          if (nchunk == 2) {   // (slot = 2 * part + chunk, as the sixteen-lane fused epilogue writes it)
            unsigned short* qp = coefq + (((int64_t)s * N16 + nn) * 4 + ch) * 8 + cc;
            qp[0] = p1; qp[16] = p2;
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
    __shared__ double smt[4][8];
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
      double e6 = ok ? (double)wk0 * (double)wk0 : 0.0, e7 = 0.0;
      if (K > 1 && ok) { const double w1 = (double)V[(int64_t)g * D + 1]; e7 = w1 * w1; }
#pragma unroll
      for (int q = 1; q < 64; q <<= 1) { e6 += __shfl_xor(e6, q, 64); e7 += __shfl_xor(e7, q, 64); }
      if (l == 0) { smt[grp][6] = e6; smt[grp][7] = e7; }
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
      double e[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) { double r = smt[0][i]; r += smt[1][i]; r += smt[2][i]; r += smt[3][i]; e[i] = r; }
      const int W_ = 3 + K;
      double* ga = mg.pre.gene_partA + (int64_t)bx * W_;
      double* gb = mg.pre.gene_partB + (int64_t)bx * W_;
      if (mg.pre.s2) { ga[0] = 0.5 * (e[0] + e[3]); ga[1] = 0.5 * (e[1] + e[4]); ga[2] = 0.5 * (e[2] + e[5]); }
      else { ga[0] = e[0]; ga[1] = e[1]; ga[2] = e[2]; }
      gb[0] = e[3]; gb[1] = e[4]; gb[2] = e[5];
      for (int k = 0; k < K && k < 2; ++k) { ga[3 + k] = e[6 + k]; gb[3 + k] = e[6 + k]; }
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

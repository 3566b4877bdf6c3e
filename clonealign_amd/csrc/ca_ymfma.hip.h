// The count matrix's two products of the VI loop on the int8 matrix cores (gfx950):
//   YW[n][k]    = sum_g y_ng W_gk        (row products:    psi's gradient and the psi.(YW) term of EE_p_y)
//   YtPsi[g][k] = sum_n y_ng psi_nk      (column products: W's gradient)
// They are the `y * log p` part of tfd$Multinomial$log_prob (R/inference-tflow.R:294-296) that depends on Y and the
// parameters only.  k_ypass computes both on the VALU from one row-major copy: 3.5 vector
// instructions per count, which it takes from the forward sweep it runs beside.  Here each product streams its OWN tiled
// copy of the matrix -- stored byte for byte in the operand layout of v_mfma_i32_16x16x64_i8, so a wave's load is 1 KiB
// contiguous and goes to the matrix core as it is -- against the parameter quantised to 32-bit fixed point in four
// signed base-256 digits (one operand column per digit).  No vector arithmetic per count at all; integer accumulation is
// exact, so the result is sum_g y_ng * round(W_gk 2^e) 2^-e to the last bit, in any summation order.
//
// Stored byte = y ^ 0x80, i.e. y - 128 as a signed byte (counts above 255 keep 255 here and their excess in the
// overflow list, like the row-major copy); the bias is undone with one more MFMA per step whose A operand is all ones:
// it yields the digit sums of the parameter, so out = D + 128 * D1 in the accumulator layout, no second pass.
//
//   Yf  [NT16][GS64][64 lanes][16 B]  byte (l, b) = Y[16 T + (l & 15)][64 s + 16 (l >> 4) + b] ^ 0x80   ("cell-tiled")
//   Yb  [GT16][NS64][64 lanes][16 B]  byte (l, b) = Y[64 s + 16 (l >> 4) + b][16 T + (l & 15)] ^ 0x80   ("gene-tiled")
//   Wq  [GS64][64 lanes][16 B]        byte (l, b) = digit (l & 3) of fix(W[64 s + 16 (l >> 4) + b][(l & 15) >> 2])
//   Pq  [NS64][64 lanes][16 B]        byte (l, b) = digit (l & 3) of fix(psi[64 s + 16 (l >> 4) + b][(l & 15) >> 2])
// (padding rows / columns: 0x80 in the matrix images = count 0, 0 in the parameter images; K <= 4.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef int ca_i32x4 __attribute__((ext_vector_type(4)));

#define CA_YM_TB 256

// fixed point of a parameter block: x = rint(v * 2^e) with |x| <= 2^30, e from the block's largest magnitude
__device__ __forceinline__ int ca_fix_exp(float amax) { return amax > 0.f ? 29 - ilogbf(amax) : 0; }

// ---------------------------------------------------------------- tiling of the count matrix (once per fit)
// row-major u8 [N][Gp] (Gp a multiple of 64) -> Yf.  One thread per 16-byte chunk of the image.
__global__ void __launch_bounds__(CA_YM_TB) k_tile_yf(const uint8_t* __restrict__ Y, uint4* __restrict__ Yf, int64_t N, int Gp,
                                                      int64_t NT, int GS) {
  const int64_t i = (int64_t)blockIdx.x * CA_YM_TB + threadIdx.x;
  if (i >= NT * GS * 64) return;
  const int l = (int)(i & 63);
  const int64_t ts = i >> 6;
  const int s = (int)(ts % GS);
  const int64_t T = ts / GS;
  const int64_t n = T * 16 + (l & 15);
  const int g0 = s * 64 + 16 * (l >> 4);
  uint4 v = {0u, 0u, 0u, 0u};
  if (n < N && g0 < Gp) v = *reinterpret_cast<const uint4*>(Y + n * (int64_t)Gp + g0);
  v.x ^= 0x80808080u; v.y ^= 0x80808080u; v.z ^= 0x80808080u; v.w ^= 0x80808080u;
  Yf[i] = v;
}
// row-major u8 [N][Gp] -> Yb.  Block = 64 cells x 64 genes through LDS (byte transpose).
__global__ void __launch_bounds__(CA_YM_TB) k_tile_yb(const uint8_t* __restrict__ Y, uint4* __restrict__ Yb, int64_t N, int Gp,
                                                      int GT, int64_t NS) {
  __shared__ uint8_t tile[64][80];   // [cell][gene], row pitch 80: 16-byte aligned rows
  const int64_t s = blockIdx.x;      // cell step
  const int gq = blockIdx.y;         // group of 4 gene tiles = 64 genes
  {
    const int r = threadIdx.x >> 2, c16 = threadIdx.x & 3;   // 64 rows x 4 chunks of 16 genes
    const int64_t n = s * 64 + r;
    const int g0 = gq * 64 + 16 * c16;
    uint4 v = {0u, 0u, 0u, 0u};
    if (n < N && g0 < Gp) v = *reinterpret_cast<const uint4*>(Y + n * (int64_t)Gp + g0);
    *reinterpret_cast<uint4*>(&tile[r][16 * c16]) = v;
  }
  __syncthreads();
  const int tg = threadIdx.x >> 6, l = threadIdx.x & 63, i = l & 15, q = l >> 4;
  const int T = gq * 4 + tg;
  if (T >= GT) return;
  unsigned w[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    unsigned x = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) x |= (unsigned)tile[16 * q + 4 * d + b][16 * tg + i] << (8 * b);
    w[d] = x ^ 0x80808080u;
  }
  Yb[((int64_t)T * NS + s) * 64 + l] = (uint4){w[0], w[1], w[2], w[3]};
}

// ---------------------------------------------------------------- parameter images
// amax[0] = max |W_gk|, amax[1] = max |psi_nk| as float bit patterns (non-negative floats order like unsigned ints, so
// atomicMax gives the same value in any order); zeroed before each use.
__global__ void __launch_bounds__(CA_YM_TB) k_ym_absmax(const float* __restrict__ V, int Dv, int64_t G, const float* __restrict__ F, int Df,
                                                        int64_t N, int K, unsigned* __restrict__ amax) {
  const int64_t i = (int64_t)blockIdx.x * CA_YM_TB + threadIdx.x;
  float mw = 0.f, mp = 0.f;
  if (i < G) for (int k = 0; k < K; ++k) mw = fmaxf(mw, fabsf(V[i * Dv + k]));
  if (i < N) for (int k = 0; k < K; ++k) mp = fmaxf(mp, fabsf(F[i * Df + k]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { mw = fmaxf(mw, __shfl_xor(mw, o, 64)); mp = fmaxf(mp, __shfl_xor(mp, o, 64)); }
  if ((threadIdx.x & 63) == 0) {
    if (i < G + 64) atomicMax(amax, __float_as_uint(mw));      // NaN parameters: the bit pattern is larger than any finite one,
    if (i < N + 64) atomicMax(amax + 1, __float_as_uint(mp));  // ca_fix_exp() of NaN is what it is -- the ELBO is NaN by then anyway
  }
}
// digit p (0..3) of the signed base-256 expansion of x: x = sum_p d_p 256^p, d_p in [-128, 127]
__device__ __forceinline__ int ca_digit(int x, int p) {
  int d = 0;
#pragma unroll
  for (int i = 0; i <= 3; ++i) {
    d = (int)(signed char)(x & 0xFF);
    if (i == p) break;
    x = (x - d) >> 8;
  }
  return d;
}
// one thread per (step, lane) of an image; src [rows][ld], columns 0..K-1
__device__ __forceinline__ uint4 ca_quant16(const float* __restrict__ src, int ld, int64_t rows, int K, int64_t step, int l, float sc) {
  const int col = l & 15, k = col >> 2, p = col & 3;
  unsigned w[4] = {0u, 0u, 0u, 0u};
  if (k < K) {
    const int64_t r0 = step * 64 + 16 * (l >> 4);
#pragma unroll
    for (int b = 0; b < 16; ++b) {
      const int64_t r = r0 + b;
      const float v = r < rows ? src[r * ld + k] : 0.f;
      const int x = (int)rintf(v * sc);
      w[b >> 2] |= ((unsigned)ca_digit(x, p) & 0xFFu) << (8 * (b & 3));
    }
  }
  return (uint4){w[0], w[1], w[2], w[3]};
}
__global__ void __launch_bounds__(CA_YM_TB) k_ym_quant(const float* __restrict__ V, int Dv, int64_t G, int GS, const float* __restrict__ F,
                                                       int Df, int64_t N, int64_t NS, int K, const unsigned* __restrict__ amax,
                                                       uint4* __restrict__ Wq, uint4* __restrict__ Pq) {
  const int64_t i = (int64_t)blockIdx.x * CA_YM_TB + threadIdx.x;
  const int l = (int)(i & 63);
  const int64_t st = i >> 6;
  if (st < GS) {
    const float sc = ldexpf(1.f, ca_fix_exp(__uint_as_float(amax[0])));
    Wq[st * 64 + l] = ca_quant16(V, Dv, G, K, st, l, sc);
  } else if (st < GS + NS) {
    const float sc = ldexpf(1.f, ca_fix_exp(__uint_as_float(amax[1])));
    Pq[(st - GS) * 64 + l] = ca_quant16(F, Df, N, K, st - GS, l, sc);
  }
}

// ---------------------------------------------------------------- the two streams
// streamed once: non-temporal, so the tiles do not push the sweeps' shared operands out of the XCD's L2
__device__ __forceinline__ uint4 ca_ld_stream(const uint4* p) {
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  const v4u v = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(p));
  return (uint4){v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ ca_i32x4 ca_mfma_i8(uint4 a, uint4 b, ca_i32x4 c) {
  return __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(ca_i32x4, a), __builtin_bit_cast(ca_i32x4, b), c, 0, 0, 0);
}
// value of the four digit sums held by the four lanes of a quad (lane & 3 = digit): every lane of the quad gets the total
__device__ __forceinline__ double ca_digits_to_double(int o) {
  double v = (double)o * (double)(1 << (8 * (int)(threadIdx.x & 3)));
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  return v;
}

// One wave's stream: TL tiles of the matrix image (tile t at mat[t], consecutive steps 64 uint4 apart) against the
// parameter image par, steps [s0, s1), DEPTH steps in flight (DEPTH * TL KiB per wave; the ring is fully unrolled so
// every buffer is a fixed register).  acc[t] = D of tile t, acc1 = the all-ones tile's D (digit sums of the parameter).
template <int TL, int DEPTH>
__device__ __forceinline__ void ca_ym_sweep(const uint4* const (&mat)[TL], const uint4* __restrict__ par, int64_t s0, int64_t s1,
                                            ca_i32x4 (&acc)[TL], ca_i32x4& acc1) {
  const uint4 ones = {0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u};
#pragma unroll
  for (int t = 0; t < TL; ++t) acc[t] = (ca_i32x4){0, 0, 0, 0};
  acc1 = (ca_i32x4){0, 0, 0, 0};
  uint4 a[DEPTH][TL], b[DEPTH];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) {
    const int64_t s = (s0 + d < s1) ? s0 + d : (s1 > s0 ? s1 - 1 : s0);   // past the end: re-read the last step (never used)
#pragma unroll
    for (int t = 0; t < TL; ++t) a[d][t] = ca_ld_stream(mat[t] + s * 64);
    b[d] = par[s * 64];
  }
  for (int64_t s = s0; s < s1; s += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      if (s + d < s1) {   // wave-uniform
        acc1 = ca_mfma_i8(ones, b[d], acc1);
#pragma unroll
        for (int t = 0; t < TL; ++t) acc[t] = ca_mfma_i8(a[d][t], b[d], acc[t]);
        if (s + d + DEPTH < s1) {
#pragma unroll
          for (int t = 0; t < TL; ++t) a[d][t] = ca_ld_stream(mat[t] + (s + d + DEPTH) * 64);
          b[d] = par[(s + d + DEPTH) * 64];
        }
      }
    }
  }
}

// Row products.  A wave owns TL cell tiles (16 cells each) for ALL gene steps; accumulators D[cell][4 k + digit].
template <int TL, int DEPTH>
__device__ __forceinline__ void ca_yw_wave(const uint4* __restrict__ Yf, const uint4* __restrict__ Wq, int64_t T0, int64_t NT, int GS,
                                           ca_i32x4 (&acc)[TL], ca_i32x4& acc1) {
  const int lane = threadIdx.x & 63;
  const uint4* mat[TL];
#pragma unroll
  for (int t = 0; t < TL; ++t) mat[t] = Yf + ((T0 + t < NT) ? T0 + t : NT - 1) * GS * 64 + lane;   // tiles past the end re-read the last one
  ca_ym_sweep<TL, DEPTH>(mat, Wq + lane, 0, GS, acc, acc1);
}
// lab / test form: raw digit sums out[n][16] (int32, bias undone)
template <int TL, int DEPTH>
__global__ void __launch_bounds__(CA_YM_TB) k_yw_mfma_raw(const uint4* __restrict__ Yf, const uint4* __restrict__ Wq, int64_t NT, int GS,
                                                          int* __restrict__ out /*[NT * 16][16]*/) {
  const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
  const int64_t T0 = ((int64_t)blockIdx.x * (CA_YM_TB / 64) + (threadIdx.x >> 6)) * TL;
  if (T0 >= NT) return;
  ca_i32x4 acc[TL], acc1;
  ca_yw_wave<TL, DEPTH>(Yf, Wq, T0, NT, GS, acc, acc1);
#pragma unroll
  for (int t = 0; t < TL; ++t)
    if (T0 + t < NT)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[((T0 + t) * 16 + 4 * q + r) * 16 + j] = acc[t][r] + 128 * acc1[r];
}
// engine form: finishes in the kernel.  YW[n][k] (float, what psi's gradient reads) and the block's share of
// sum_n psi_n . (YW)_n in fp64 (the one ELBO term that needs the row products); the overflow list's entries of each cell
// (counts above 255, CSR) are added by the four lanes of the cell's quad.  Block = 4 waves = 64 TL cells.
template <int TL, int DEPTH>
__global__ void __launch_bounds__(CA_YM_TB) k_yw_mfma(const uint4* __restrict__ Yf, const uint4* __restrict__ Wq, int64_t NT, int GS,
                                                      int64_t N, int K, const float* __restrict__ F, int Df, const float* __restrict__ V,
                                                      int Dv, const unsigned* __restrict__ amax, const int64_t* __restrict__ ovf_rowptr,
                                                      const int* __restrict__ ovf_col, const float* __restrict__ ovf_val,
                                                      float* __restrict__ YW, double* __restrict__ yw_part) {
  __shared__ double sm[CA_YM_TB / 64];
  const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4, k = j >> 2, p = j & 3;
  const int64_t T0 = ((int64_t)blockIdx.x * (CA_YM_TB / 64) + (threadIdx.x >> 6)) * TL;
  double part = 0.0;
  if (T0 < NT) {   // wave-uniform
    ca_i32x4 acc[TL], acc1;
    ca_yw_wave<TL, DEPTH>(Yf, Wq, T0, NT, GS, acc, acc1);
    const double inv = ldexp(1.0, -ca_fix_exp(__uint_as_float(amax[0])));
#pragma unroll
    for (int t = 0; t < TL; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t n = (T0 + t) * 16 + 4 * q + r;
        const bool live = n < N && k < K;
        double v = ca_digits_to_double(live ? acc[t][r] + 128 * acc1[r] : 0) * inv;
        if (ovf_rowptr) {   // (uniform)
          double a = 0.0;
          if (live)
            for (int64_t e = ovf_rowptr[n] + p; e < ovf_rowptr[n + 1]; e += 4) a += (double)ovf_val[e] * (double)V[(int64_t)ovf_col[e] * Dv + k];
          a += __shfl_xor(a, 1, 64);
          a += __shfl_xor(a, 2, 64);
          v += a;
        }
        if (live && p == 0) {
          const float vf = (float)v;         // YW is a float32 array; the ELBO term is taken from the same values
          YW[n * K + k] = vf;
          part += (double)F[n * Df + k] * (double)vf;
        }
      }
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) part += __shfl_xor(part, o, 64);
  if (lane == 0) sm[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0) yw_part[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

// Column products.  A wave owns TL gene tiles for the cell steps [s0, s1) of its slice; accumulators D[gene][4 k + digit].
// grid (ceil(GT / (4 TL)) [+ extra blocks], csplit): digit sums of the slice, out[slice][GT * 16][16] int32 (bias undone).
template <int TL, int DEPTH>
__device__ __forceinline__ void ca_yt_block(const uint4* __restrict__ Yb, const uint4* __restrict__ Pq, int GT, int64_t NS, int64_t schunk,
                                            int* __restrict__ out) {
  const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
  const int T0 = (blockIdx.x * (CA_YM_TB / 64) + (int)(threadIdx.x >> 6)) * TL;
  if (T0 >= GT) return;
  const int64_t s0 = (int64_t)blockIdx.y * schunk, s1 = (s0 + schunk < NS) ? s0 + schunk : NS;
  const uint4* mat[TL];
#pragma unroll
  for (int t = 0; t < TL; ++t) mat[t] = Yb + (int64_t)((T0 + t < GT) ? T0 + t : GT - 1) * NS * 64 + lane;
  ca_i32x4 acc[TL], acc1;
  ca_ym_sweep<TL, DEPTH>(mat, Pq + lane, s0, s1, acc, acc1);
  int* o = out + (int64_t)blockIdx.y * GT * 256;
#pragma unroll
  for (int t = 0; t < TL; ++t)
    if (T0 + t < GT)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[((int64_t)(T0 + t) * 16 + 4 * q + r) * 16 + j] = acc[t][r] + 128 * acc1[r];
}
template <int TL, int DEPTH>
__global__ void __launch_bounds__(CA_YM_TB) k_yt_mfma_raw(const uint4* __restrict__ Yb, const uint4* __restrict__ Pq, int GT, int64_t NS,
                                                          int64_t schunk, int* __restrict__ out) {
  ca_yt_block<TL, DEPTH>(Yb, Pq, GT, NS, schunk, out);
}

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

#ifdef CA_LAB   // (the two-copy form, CA_VARX_Y_MFMA2: measured slower than the one-copy stream, lab library only)
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

#endif   // CA_LAB
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
__device__ __forceinline__ uint4 ca_quant16(const float* __restrict__ src, int ld, int64_t rows, int K, int64_t step, int l, float sc, int rep = 0) {
  const int col = l & 15, k = rep ? 0 : col >> 2, p = col & 3;   // rep: K = 1, the digits repeated in all four column groups
  unsigned w[4] = {0u, 0u, 0u, 0u};
  if (k < K) {
    const int64_t r0 = step * 64 + 16 * (l >> 4);
#pragma unroll
    for (int b = 0; b < 16; ++b) {
      const int64_t r = r0 + b;
      const float v = r < rows ? src[r * ld + k] : 0.f;
      const int x = (int)rintf(fminf(fmaxf(v * sc, -2147483000.f), 2147483000.f));
      w[b >> 2] |= ((unsigned)ca_digit(x, p) & 0xFFu) << (8 * (b & 3));
    }
  }
  return (uint4){w[0], w[1], w[2], w[3]};
}
__global__ void __launch_bounds__(CA_YM_TB) k_ym_quant(const float* __restrict__ V, int Dv, int64_t G, int GS, const float* __restrict__ F,
                                                       int Df, int64_t N, int64_t NS, int K, const unsigned* __restrict__ amax,
                                                       uint4* __restrict__ Wq, uint4* __restrict__ Pq, int rep = 0) {
  const int64_t i = (int64_t)blockIdx.x * CA_YM_TB + threadIdx.x;
  const int l = (int)(i & 63);
  const int64_t st = i >> 6;
  if (st < GS) {
    const float sc = ldexpf(1.f, ca_fix_exp(__uint_as_float(amax[0])));
    Wq[st * 64 + l] = ca_quant16(V, Dv, G, K, st, l, sc, rep);
  } else if (st < GS + NS) {
    const float sc = ldexpf(1.f, ca_fix_exp(__uint_as_float(amax[1])));
    Pq[(st - GS) * 64 + l] = ca_quant16(F, Df, N, K, st - GS, l, sc, rep);
  }
}
// row-major u8 [N][Gp] -> biased copy in 4-KiB pieces of 64 cells x 64 genes, [N64/64][Gp/64][64 rows][64 B] (rows past N:
// count 0): a wave's load instruction then covers 1 KiB of consecutive addresses, like the row-major stream's, instead of
// sixteen 64-byte pieces in sixteen DRAM pages (3.4 TB/s measured with the row-major copy).  One thread per 16 bytes.
__global__ void __launch_bounds__(CA_YM_TB) k_bias_y(const uint8_t* __restrict__ Y, uint4* __restrict__ Ys, int64_t N, int64_t N64, int Gp) {
  const int64_t i = (int64_t)blockIdx.x * CA_YM_TB + threadIdx.x;     // index of the 16-byte chunk in the DESTINATION
  const int64_t per_piece = 256, nb = Gp / 64;
  if (i >= (N64 / 64) * nb * per_piece) return;
  const int c = (int)(i & 3), r = (int)((i >> 2) & 63);
  const int64_t piece = i >> 8;
  const int64_t cs = piece / nb, gb = piece - cs * nb;
  const int64_t n = cs * 64 + r;
  uint4 v = {0u, 0u, 0u, 0u};
  if (n < N) v = *reinterpret_cast<const uint4*>(Y + n * (int64_t)Gp + gb * 64 + 16 * c);
  v.x ^= 0x80808080u; v.y ^= 0x80808080u; v.z ^= 0x80808080u; v.w ^= 0x80808080u;
  Ys[i] = v;
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

// =====================================================================================================================
// ONE copy, both products: the count matrix stays row-major (biased bytes, y ^ 0x80) and every 64-cell x 128-gene piece
// goes through the wave's own LDS region once.  Read back row-wise (ds_read_b128) it is the A operand of the row
// products; read back through gfx950's transposing LDS read (ds_read_b64_tr_b8: per 16-lane group a block of 8 rows x 16
// bytes comes back column-major, lane 2q+p supplies the address of row q / bytes 8p..8p+7 and lane i receives column i --
// measured, tools/trb8_lab.hip) it is the B operand of the column products.  Half the bytes of the two-copy form.  K = 1.
//
//   Ys   [N64/64][Gp/64][64][64 B]  biased bytes in 4-KiB pieces (64 cells x 64 genes, row-major inside), Gp a multiple of 512,
//        rows past N and genes past G hold 0x80 (count 0)
//   Wr   [Gp/64][64 lanes][16 B]  byte (l, b) = digit (l & 3) of fix(W[64 s + 16 (l >> 4) + b])   (every column group the same)
//   Pr   [N64/64][64 lanes][16 B] byte (l, b) = digit (l & 3) of fix(psi[64 s + 16 (l >> 4) + b])
// The digits are replicated over the four groups of four operand columns (rows) so that FOUR tiles share one accumulator:
// tile t multiplies against the image masked down to group t, and its sums land in columns (rows) 4t .. 4t+3.
//   YWi  [nseg][N][4]   int32 digit sums of a gene segment (512 genes), bias undone
//   YTi  [nrg][Gp][4]   int32 digit sums of a row group (4 strips of RS cells), bias undone
constexpr int CA_YS_GW = 512;      // genes per segment (block = one segment x four strips); the switch in k_ys_mfma assumes 8 x 64
static_assert(CA_YS_GW == 512, "k_ys_mfma's accumulator switch has eight cases");
constexpr int CA_YS_PITCH = 80;    // LDS row pitch of the 64-byte rows: 16 lanes of a ds_read_b128 group fall on 16 different bank quads

// The eight transposed reads of a 64-gene block (four gene tiles x two halves of the 16 cells) and their wait in ONE asm
// statement: the compiler treats an asm output as valid as soon as the statement ends, and would otherwise be free to copy a
// result register (to line up the four VGPRs of an MFMA operand) before the data has arrived.
__device__ __forceinline__ void ca_ds_read_tr_b8_x8(unsigned addr, uint2 (&lo)[4], uint2 (&hi)[4]) {
  asm volatile(
      "ds_read_b64_tr_b8 %0, %8\n\tds_read_b64_tr_b8 %4, %8 offset:640\n\t"
      "ds_read_b64_tr_b8 %1, %8 offset:16\n\tds_read_b64_tr_b8 %5, %8 offset:656\n\t"
      "ds_read_b64_tr_b8 %2, %8 offset:32\n\tds_read_b64_tr_b8 %6, %8 offset:672\n\t"
      "ds_read_b64_tr_b8 %3, %8 offset:48\n\tds_read_b64_tr_b8 %7, %8 offset:688\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(lo[0]), "=&v"(lo[1]), "=&v"(lo[2]), "=&v"(lo[3]), "=&v"(hi[0]), "=&v"(hi[1]), "=&v"(hi[2]), "=&v"(hi[3])
      : "v"(addr)
      : "memory");
}
// two gene tiles at a time (the riding form: eight result registers live instead of sixteen; addr + 32 for tiles 2 and 3)
__device__ __forceinline__ void ca_ds_read_tr_b8_x4(unsigned addr, uint2 (&lo)[2], uint2 (&hi)[2]) {
  asm volatile(
      "ds_read_b64_tr_b8 %0, %4\n\tds_read_b64_tr_b8 %2, %4 offset:640\n\t"
      "ds_read_b64_tr_b8 %1, %4 offset:16\n\tds_read_b64_tr_b8 %3, %4 offset:656\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(lo[0]), "=&v"(lo[1]), "=&v"(hi[0]), "=&v"(hi[1])
      : "v"(addr)
      : "memory");
}
static_assert(8 * CA_YS_PITCH == 640, "offsets in ca_ds_read_tr_b8_x8 / _x4 assume the 80-byte pitch");
__device__ __forceinline__ uint4 ca_and4(uint4 a, unsigned m) { return (uint4){a.x & m, a.y & m, a.z & m, a.w & m}; }

// The bias (stored byte = y - 128) is undone by the finisher: 128 x the digit sums of the parameter images, which the
// quantiser leaves per 64-step (Wsum[Gp/64][4], Psum[N64/64][4]) -- no all-ones MFMA and no accumulator for it here.
// DEPTH pieces (4 KiB each: 64 cells x 64 genes) and their W digits are in flight per wave; the piece loop is unrolled by
// DEPTH only (fully unrolled, the scheduler hoists every piece's loads and spills).
#ifndef CA_YS_DEPTH
#define CA_YS_DEPTH 2   // measured at 100k x 5k: depth 2 / 3 waves 92 us, depth 4 / 2 waves 95, depth 1 / 4 waves 104 (5.5 TB/s stored)
#endif
#ifndef CA_YS_WAVES
#define CA_YS_WAVES 3   // waves per SIMD the register budget is set for
#endif
// What the stream needs besides the matrix: the parameter images, their per-step digit sums (the bias of the stored bytes), the two
// fixed-point exponents the quantiser used, and the float partial slabs it leaves for the finisher -- the SAME slabs, with the same
// meaning, as the vector stream's (k_ypass): YWpart [nseg][N] = the segment's share of (Y.W)_n, YTpart [nrg][Gp] = the row group's
// share of (Y^T psi)_g, so one finisher (k_yfinish) serves both.  Digits are combined in fp64 from exact integer sums.
struct ca_ys_io {
  const uint4* Wr; const uint4* Pr; const int* Wsum; const int* Psum; const int* exps;   // exps[0] for W, exps[1] for psi
  float* YWpart; float* YTpart;
};
template <int DEPTH = CA_YS_DEPTH>
__device__ __forceinline__ void ca_ys_mfma_body(int blk, const uint8_t* __restrict__ Ys, const ca_ys_io& io, int64_t N, int Gp,
                                                int RS /* cells per strip, multiple of 64 */,
                                                unsigned char* ca_ys_lds /* 16-byte aligned, CA_YS_LDS_BYTES: [4 waves][64][CA_YS_PITCH], reused for the combine */) {
  const uint4* __restrict__ Wr = io.Wr;
  const uint4* __restrict__ Pr = io.Pr;
  constexpr int NP = CA_YS_GW / 64;
  static_assert(NP % DEPTH == 0, "pieces per cell step must be a multiple of the pipeline depth");
  const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
  // the wave index as a SCALAR: strip bounds, piece addresses and the piece loop are then wave-uniform (scalar registers, scalar
  // branches, loads of the form global_load v, v_off, s[base]) instead of 64-bit vector arithmetic per lane -- the kernel that rides
  // on the forward sweep has 128 vector registers for everything
  // (& 3, & 255 below: a 512-thread launch runs TWO units per block, waves 0-3 and 4-7 each with their own LDS region -- k_fwd_bal_ys;
  //  in a 256-thread launch they change nothing)
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6) & 3);
  const int nseg = Gp / CA_YS_GW;
  const int rg = blk / nseg, seg = blk - rg * nseg;
  unsigned char* my = ca_ys_lds + (size_t)wv * 64 * CA_YS_PITCH;
  const int64_t c0 = ((int64_t)rg * 4 + wv) * RS;             // first cell of this wave's strip
  const int64_t c1 = (c0 + RS < N) ? c0 + RS : N;             // (rows up to the next multiple of 64 exist and hold zeros)
  const int g0 = seg * CA_YS_GW;
  const unsigned grp = (unsigned)(j >> 2);
  unsigned msk[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) msk[t] = grp == (unsigned)t ? 0xFFFFFFFFu : 0u;
  ca_i32x4 acc_yt[NP];
#pragma unroll
  for (int a = 0; a < NP; ++a) acc_yt[a] = (ca_i32x4){0, 0, 0, 0};
  // staging: load instruction i of a piece covers rows 16 i .. 16 i + 15, 64 bytes each
  const int lrow = lane >> 2, lch = lane & 3;
  unsigned char* wr_dst = my + lrow * CA_YS_PITCH + 16 * lch;
  const unsigned char* rd_row = my + j * CA_YS_PITCH + 16 * q;
  const unsigned rd_tr = (unsigned)(size_t)my + (unsigned)((16 * q + (j >> 1)) * CA_YS_PITCH + 8 * (j & 1));
  const uint8_t* src = Ys + ((c0 >> 6) * (int64_t)(Gp / 64) + (g0 >> 6)) * 4096;   // (scalar) piece (cell step, gene block), 1 KiB per load
  const uint4* wsrc = Wr + (int64_t)(g0 >> 6) * 64;                                 // (scalar)
  const unsigned voff = 16u * (unsigned)lane;                                       // the lane's 16 bytes of a 1-KiB load
  uint4 R[DEPTH][4], W[DEPTH];
  // pieces of the strip: cell step st (64 cells), gene block a (0 .. NP-1); DEPTH pieces in flight
  const int nsteps = c0 < c1 ? (int)((c1 - c0 + 63) / 64) : 0;
  // Buffer loads: a scalar base (the strip's first piece; the segment's W image; psi's image) in a resource descriptor, the piece's byte offset in a
  // scalar register, the lane's 16 bytes in ONE vector register -- no 64-bit vector address per stream, which the riding form's register budget
  // (128, the sweep's) has no room for.  Offsets are 32-bit: the host uses this stream only where RS * Gp and 16 N stay far below 2^31.
  typedef unsigned ca_v4u __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(src), 0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(wsrc), 0, NP * 1024, 0x00020000);
  const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(Pr + (c0 >> 6) * 64), 0, 0x7FFFFFFF, 0x00020000);
  auto issue = [&](int slot, int st, int a) {
    const int so = (st * (Gp / 64) + a) * 4096;   // (scalar)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const ca_v4u v = __builtin_amdgcn_raw_buffer_load_b128(ry, (int)(voff + 1024u * (unsigned)i), so, 2 /* nt: streamed once */);
      R[slot][i] = (uint4){v.x, v.y, v.z, v.w};
    }
    const ca_v4u w = __builtin_amdgcn_raw_buffer_load_b128(rw, (int)voff, a * 1024, 0);
    W[slot] = (uint4){w.x, w.y, w.z, w.w};
  };
  if (nsteps > 0) {
#pragma unroll
    for (int d_ = 0; d_ < DEPTH; ++d_) issue(d_, 0, d_);
  }
  // Round 5: the gene blocks of a cell step are UNROLLED, so that every accumulator is a fixed register (the loop over pieces with a switch on the
  // block index moved the 32 accumulator registers through chains of copies: ~40 v_mov per piece), the column products chain straight onto their
  // accumulator, and the images are never masked per piece -- row products: one accumulator per cell tile against the UNMASKED W image (K = 1: its
  // four column groups repeat the digits, so every group holds the tile's sums and the flush picks its own); column products: psi's image masked to
  // one row group per gene tile once per cell step.  Vector instructions per 4-KiB piece: ~76 -> 2, and with the buffer loads above the merged forward kernel
  // no longer spills (128 VGPRs, 6 spilled -> 0).  Measured: the launch's time at cfg-3 does NOT change (within +-1 us; 25k cells -1 us) -- what the
  // riding stream costs the sweep is its wave slot, not its instructions (profiles/r05_stream_slot.txt).
  ca_i32x4 acc_yw[4];
  uint4 prm[4];
  // bias of the row products: 128 x (digit sums of the segment's W image) per digit -- wave-uniform addresses, so the sums live
  // in scalar registers for the whole strip and cost the piece loop no vector register; a lane picks digit p = j & 3 at the flush
  int wtot[4] = {0, 0, 0, 0};
  {
    const int* ws = io.Wsum + (g0 >> 6) * 4;
#pragma unroll
    for (int a = 0; a < NP; ++a)
#pragma unroll
      for (int p_ = 0; p_ < 4; ++p_) wtot[p_] += ws[a * 4 + p_];
  }
  const int e_w = io.exps[0];
  for (int st = 0; st < nsteps; ++st) {
    const int64_t cs = c0 + (int64_t)st * 64;
    const bool more = st + 1 < nsteps;   // (wave-uniform)
    {
      // (the wait for this load is also the wait for the step's first piece, which is needed next anyway; a stream wave has ~2000 cycles per piece)
      const ca_v4u pv = __builtin_amdgcn_raw_buffer_load_b128(rp, (int)voff, st * 1024, 0);
      const uint4 pr = {pv.x, pv.y, pv.z, pv.w};
#pragma unroll
      for (int t = 0; t < 4; ++t) { acc_yw[t] = (ca_i32x4){0, 0, 0, 0}; prm[t] = ca_and4(pr, msk[t]); }
    }
#pragma unroll
    for (int a = 0; a < NP; ++a) {
      const int slot = a % DEPTH;
      __builtin_amdgcn_sched_barrier(0);   // (pieces one after the other: hoisting the next piece's LDS reads over this one's costs registers the sweep's budget has not)
      // the piece is in R[slot]: park it in LDS, start the loads of the piece DEPTH further on, then feed the matrix core
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(wr_dst + 16 * i * CA_YS_PITCH) = R[slot][i];
      const uint4 wr = W[slot];
      if (a + DEPTH < NP) issue(slot, st, a + DEPTH);
      else if (more) issue(slot, st + 1, a + DEPTH - NP);
      // row products: the four cell tiles against this 64-gene block
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const uint4 av = *reinterpret_cast<const uint4*>(rd_row + 16 * t * CA_YS_PITCH);
        acc_yw[t] = ca_mfma_i8(av, wr, acc_yw[t]);
      }
      // column products: the four gene tiles of the block against the 64 cells, one accumulator (tile t -> rows 4t .. 4t+3)
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        uint2 lo[2], hi[2];
        ca_ds_read_tr_b8_x4(rd_tr + 32u * (unsigned)h2, lo, hi);
#pragma unroll
        for (int t = 0; t < 2; ++t) acc_yt[a] = ca_mfma_i8(prm[2 * h2 + t], (uint4){lo[t].x, lo[t].y, hi[t].x, hi[t].y}, acc_yt[a]);
      }
    }
    {   // the cell step is complete: lane (column 4t + p, q) holds cells 16 t + 4 q + r, digit p
      const int t = j >> 2, p = j & 3;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t n = cs + 16 * t + 4 * q + r;
        const int wb = p == 0 ? wtot[0] : p == 1 ? wtot[1] : p == 2 ? wtot[2] : wtot[3];
        const int ar = t == 0 ? acc_yw[0][r] : t == 1 ? acc_yw[1][r] : t == 2 ? acc_yw[2][r] : acc_yw[3][r];
        long long v = ((long long)ar + 128ll * (long long)wb) << (8 * p);   // digit p of the quad's four (lanes j & 3): exact in 64 bits
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        if (p == 0 && n < N) io.YWpart[(int64_t)seg * N + n] = (float)ldexp((double)v, -e_w);
      }
    }
  }
  // column products of the strip: lane (gene n = j, q), accumulator a: gene tile 4 a + q, digits r = 0..3; combine the four
  // strips of the block in LDS (integer sums: any order), one row of YTi per block
  __syncthreads();
  int* comb = reinterpret_cast<int*>(ca_ys_lds);   // [4 waves][8 acc][64 lanes][4]: 32 KB > the 20 KB of tiles: the launch asks for 32 KB
#pragma unroll
  for (int a = 0; a < NP; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) comb[((wv * NP + a) * 64 + lane) * 4 + r] = acc_yt[a][r];
  __syncthreads();
  // bias of the column products: 128 x (digit sums of psi's image over the block's cell steps); scale 2^-e_psi
  long long pb[4] = {0, 0, 0, 0};
  {
    const int64_t s0_ = ((int64_t)rg * 4 * RS) >> 6;
    const int64_t s1_ = (((int64_t)rg * 4 + 4) * RS < ((N + 63) / 64) * 64 ? ((int64_t)rg * 4 + 4) * RS : ((N + 63) / 64) * 64) >> 6;
    for (int64_t st = s0_; st < s1_; ++st) {
      const int4 d4 = *reinterpret_cast<const int4*>(io.Psum + st * 4);
      pb[0] += d4.x; pb[1] += d4.y; pb[2] += d4.z; pb[3] += d4.w;
    }
  }
  const double inv_p = ldexp(1.0, -io.exps[1]);
  for (int i = (int)threadIdx.x & (CA_YM_TB - 1); i < NP * 64; i += CA_YM_TB) {
    const int l = i & 63, a = i >> 6;
    double v = 0.0;
#pragma unroll
    for (int r = 3; r >= 0; --r) {
      const int o = i * 4 + r;
      const int sr = (comb[o] + comb[o + NP * 256]) + (comb[o + 2 * NP * 256] + comb[o + 3 * NP * 256]);
      v = v * 256.0 + (double)((long long)sr + 128 * pb[r]);
    }
    const int gene = g0 + 16 * (4 * a + (l >> 4)) + (l & 15);
    io.YTpart[(int64_t)rg * Gp + gene] = (float)(v * inv_p);
  }
}
__global__ void __launch_bounds__(CA_YM_TB, CA_YS_WAVES) k_ys_mfma(const uint8_t* __restrict__ Ys, ca_ys_io io, int64_t N, int Gp, int RS) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ca_ys_dyn[];
  ca_ys_mfma_body<CA_YS_DEPTH>((int)blockIdx.x, Ys, io, N, Gp, RS, ca_ys_dyn);
}
constexpr int CA_YS_LDS_BYTES = 4 * (CA_YS_GW / 64) * 64 * 4 * 4;   // the combine buffer (32 KB) >= 4 x 64 x CA_YS_PITCH

// per-step digit sums of a replicated image (for the bias): sums[step][p] = sum over the step's 64 entries of digit p
__global__ void __launch_bounds__(CA_YM_TB) k_ym_digit_sums(const uint4* __restrict__ img, int64_t steps, int* __restrict__ sums) {
  const int64_t i = (int64_t)blockIdx.x * CA_YM_TB + threadIdx.x;
  const int l = (int)(i & 63);
  const int64_t st = i >> 6;
  int s = 0;
  if (st < steps) {
    const uint4 v = img[st * 64 + l];
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int b = 0; b < 4; ++b) s += (int)(signed char)((w[d] >> (8 * b)) & 0xFFu);
  }
  // lanes (col j = p, q = 0..3) hold digit p: sum over q
  s += __shfl_xor(s, 16, 64);
  s += __shfl_xor(s, 32, 64);
  if (st < steps && l < 4) sums[st * 4 + l] = s;
}

// Parameter images of the one-copy stream for one parameter state.  The fixed-point exponents need the largest magnitudes of W
// and psi: lag = 0 takes them from amax_in as exact maxima (k_ym_absmax ran before: ONE pair); in the loop amax_in holds the PREVIOUS
// state's maxima as per-block pairs (what this body left last time) and `slack` bounds what the Adam steps since can add to any
// magnitude (TF1 Adam: |step| <= lr_t (1 - b1) / sqrt((1 - b2)(1 - b1^2 / b2)), Cauchy-Schwarz on the two moving averages), so
// 2^e (max + slack) < 2^30 holds without a second pass.  Every block reduces the n_in pairs itself (a few KB from L2) and leaves
// its own pair in amax_out -- no atomics: the first form of this kernel raised one atomicMax per wave on two addresses and spent
// 20 of its 25 us queueing there (rocprofv3, profiles/r03_ab_ystream.txt).  Block 0 writes the exponents used to exps[2] for
// the stream.  Digit sums per 64-step go to Wsum / Psum (the bias of the stored bytes).  K = 1.
// In the loop the body runs as EXTRA BLOCKS of the per-cell Adam kernel (k_adam_cell), which is where W and psi become final.
struct ca_ysq_args {
  int nblk;                       // blocks of the quantiser (0 = none riding)
  const float* V; int Dv; int64_t G; int GS; const float* F; int Df; int64_t N; int64_t NS;
  const float* amax_in; int n_in; float slack_w, slack_p;
  float* amax_out;                // [nblk][2]
  int* exps; uint4* Wr; uint4* Pr; int* Wsum; int* Psum;
};
// One wave's 64-step of an image from entries that are in registers: vals[b] = entry 16 (l >> 4) + b of the step, own = entry l.
// Block-level: every thread of the block calls it (two barriers); `out` = this block's pair in amax_out.
__device__ __forceinline__ void ca_ys_quant_core(const ca_ysq_args& a, bool live, bool isw, int64_t step, const float (&vals)[16], float own,
                                                 int out, bool write_exps, float* sm /* >= 2 * (CA_YM_TB / 64) floats */) {
  const int tid = threadIdx.x, l = tid & 63, wv = tid >> 6;
  // largest magnitudes of the state amax_in describes
  float mw = 0.f, mp = 0.f;
  for (int i = tid; i < a.n_in; i += CA_YM_TB) { mw = fmaxf(mw, a.amax_in[2 * i]); mp = fmaxf(mp, a.amax_in[2 * i + 1]); }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { mw = fmaxf(mw, __shfl_xor(mw, o, 64)); mp = fmaxf(mp, __shfl_xor(mp, o, 64)); }
  if (l == 0) { sm[2 * wv] = mw; sm[2 * wv + 1] = mp; }
  __syncthreads();
  mw = fmaxf(fmaxf(sm[0], sm[2]), fmaxf(sm[4], sm[6]));
  mp = fmaxf(fmaxf(sm[1], sm[3]), fmaxf(sm[5], sm[7]));
  __syncthreads();
  const int ew = ca_fix_exp(mw + a.slack_w), ep = ca_fix_exp(mp + a.slack_p);
  if (write_exps && tid == 0) { a.exps[0] = ew; a.exps[1] = ep; }
  float m = 0.f;
  if (live) {
    const float sc = ldexpf(1.f, isw ? ew : ep);
    // ca_quant16(src, ld, rows, 1, step, l, sc, 1) on the entries already in registers: digit p = l & 3 of each
    const int p = l & 3;
    unsigned w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int b = 0; b < 16; ++b) {
      const int x = (int)rintf(fminf(fmaxf(vals[b] * sc, -2147483000.f), 2147483000.f));
      w[b >> 2] |= ((unsigned)ca_digit(x, p) & 0xFFu) << (8 * (b & 3));
    }
    const uint4 v = {w[0], w[1], w[2], w[3]};
    (isw ? a.Wr : a.Pr)[step * 64 + l] = v;
    int sd = 0;   // digit sums of the step (lanes with column p = l & 3 in the first group hold digit p for the 16 entries of group q)
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int b = 0; b < 4; ++b) sd += (int)(signed char)((w[d] >> (8 * b)) & 0xFFu);
    sd += __shfl_xor(sd, 16, 64);
    sd += __shfl_xor(sd, 32, 64);
    if (l < 4) (isw ? a.Wsum : a.Psum)[step * 4 + l] = sd;
    // exact maximum of this step's 64 entries, one per lane
    m = fabsf(own);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  }
  if (l == 0) { sm[2 * wv] = isw ? m : 0.f; sm[2 * wv + 1] = isw ? 0.f : m; }
  __syncthreads();
  if (tid == 0) {
    a.amax_out[2 * out] = fmaxf(fmaxf(sm[0], sm[2]), fmaxf(sm[4], sm[6]));
    a.amax_out[2 * out + 1] = fmaxf(fmaxf(sm[1], sm[3]), fmaxf(sm[5], sm[7]));
  }
}
__device__ __forceinline__ void ca_ys_quant_body(int blk, const ca_ysq_args& a, float* sm /* >= 2 * (CA_YM_TB / 64) floats */) {
  const int tid = threadIdx.x, l = tid & 63, wv = tid >> 6;
  // this wave's 64-step of an image: its sixteen entries per lane (and the lane's own entry, for the step's exact maximum) are
  // loaded FIRST -- their addresses do not depend on the exponents, so the round trip runs beside the reduction in the core
  const int64_t st = (int64_t)blk * (CA_YM_TB / 64) + wv;      // one wave per 64-step of an image
  const bool live = st < a.GS + a.NS;
  const bool isw = !live || st < a.GS;
  const float* src = isw ? a.V : a.F;
  const int ld = isw ? a.Dv : a.Df;
  const int64_t rows = isw ? a.G : a.N, step = live ? (isw ? st : st - a.GS) : 0;
  float vals[16], own = 0.f;
  {
    const int64_t r0 = step * 64 + 16 * (l >> 4);
#pragma unroll
    for (int b = 0; b < 16; ++b) {
      const int64_t r = r0 + b;
      vals[b] = (live && r < rows) ? src[r * ld] : 0.f;
    }
    const int64_t r = step * 64 + l;
    if (live && r < rows) own = src[r * ld];
  }
  ca_ys_quant_core(a, live, isw, step, vals, own, blk, blk == 0, sm);
}
// Wave-level form of ca_ys_quant_core for ONE 64-step whose entries are in registers (round 4, the gene blocks of k_update_merged): no
// block-level operation; returns the step's exact maximum (wave-uniform).  The exponents come from the same maxima (a maximum does not
// depend on the order it is taken in), everything else is the core's arithmetic.
__device__ __forceinline__ float ca_ys_quant_wave(const ca_ysq_args& a, bool live, bool isw, int64_t step, float own, bool write_exps) {
  const int l = threadIdx.x & 63;
  float vals[16];
#pragma unroll
  for (int b = 0; b < 16; ++b) vals[b] = __shfl(own, 16 * (l >> 4) + b, 64);
  float mw = 0.f, mp = 0.f;
  for (int i = l; i < a.n_in; i += 64) { mw = fmaxf(mw, a.amax_in[2 * i]); mp = fmaxf(mp, a.amax_in[2 * i + 1]); }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { mw = fmaxf(mw, __shfl_xor(mw, o, 64)); mp = fmaxf(mp, __shfl_xor(mp, o, 64)); }
  const int ew = ca_fix_exp(mw + a.slack_w), ep = ca_fix_exp(mp + a.slack_p);
  if (write_exps && l == 0) { a.exps[0] = ew; a.exps[1] = ep; }
  float m = 0.f;
  if (live) {
    const float sc = ldexpf(1.f, isw ? ew : ep);
    const int p = l & 3;
    unsigned w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int b = 0; b < 16; ++b) {
      const int x = (int)rintf(fminf(fmaxf(vals[b] * sc, -2147483000.f), 2147483000.f));
      w[b >> 2] |= ((unsigned)ca_digit(x, p) & 0xFFu) << (8 * (b & 3));
    }
    const uint4 v = {w[0], w[1], w[2], w[3]};
    (isw ? a.Wr : a.Pr)[step * 64 + l] = v;
    int sd = 0;
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int b = 0; b < 4; ++b) sd += (int)(signed char)((w[d] >> (8 * b)) & 0xFFu);
    sd += __shfl_xor(sd, 16, 64);
    sd += __shfl_xor(sd, 32, 64);
    if (l < 4) (isw ? a.Wsum : a.Psum)[step * 4 + l] = sd;
    m = fabsf(own);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  }
  return m;
}
// The same images made where the entries are BORN (round 4, k_update_merged): a block of 256 genes (cells) has just stepped W_g0 (psi_n0),
// one entry per lane = four 64-steps of the W (psi) image; the sixteen entries a lane packs come from its wave-mates by shuffle.
// `own` = this lane's entry (0 past the last row).  Same arithmetic on the same floats as ca_ys_quant_body reading them back.
__device__ __forceinline__ void ca_ys_quant_inreg(const ca_ysq_args& a, bool isw, int blk, float own, int out, bool write_exps, float* sm) {
  const int l = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t step = (int64_t)blk * (CA_YM_TB / 64) + wv;
  const bool live = step < (isw ? (int64_t)a.GS : a.NS);
  float vals[16];
#pragma unroll
  for (int b = 0; b < 16; ++b) vals[b] = __shfl(own, 16 * (l >> 4) + b, 64);
  ca_ys_quant_core(a, live, isw, live ? step : 0, vals, own, out, write_exps, sm);
}
__global__ void __launch_bounds__(CA_YM_TB) k_ys_quant(ca_ysq_args a) {
  __shared__ float sm[2 * (CA_YM_TB / 64)];
  ca_ys_quant_body((int)blockIdx.x, a, sm);
}

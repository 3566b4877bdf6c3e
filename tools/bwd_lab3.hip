// Backward sweep on the matrix cores: t = coef . L^T as ONE v_mfma_f32_16x16x32_bf16 per 16 cells x 16 genes, exact
// because coef is split into three bf16 parts (K = 3 x 8 clones) and integer copy numbers are bf16-exact.
// Lab: verifies against the VALU kernel and times both.  Not product code.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
constexpr int NC = 8;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_pull(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum63(float v) {
  v += dpp_pull<0xB1, 0xF>(v); v += dpp_pull<0x4E, 0xF>(v); v += dpp_pull<0x141, 0xF>(v); v += dpp_pull<0x140, 0xF>(v);
  v += dpp_pull<0x142, 0xA>(v); v += dpp_pull<0x143, 0xC>(v);
  return v;
}
__device__ __forceinline__ float row16_sum(float v) {   // sum over the 16 lanes of a DPP row, result in every lane of the row
  v += dpp_pull<0xB1, 0xF>(v); v += dpp_pull<0x4E, 0xF>(v); v += dpp_pull<0x141, 0xF>(v); v += dpp_pull<0x140, 0xF>(v);
  return v;
}
// ---- reference: VALU kernel (as in the library)
template <int RG>
__global__ void __launch_bounds__(256) bwd_valu(const float* __restrict__ coef, const float* __restrict__ F, const float* __restrict__ em2,
                                                const float* __restrict__ Lb, const float* __restrict__ mu, const float* __restrict__ Vs,
                                                const float* __restrict__ V, float* __restrict__ gpart, float* __restrict__ dFpart,
                                                long N, int G, long cchunk) {
  const int lane = threadIdx.x & 63, tile = blockIdx.x * 4 + (threadIdx.x >> 6), gbase = tile * 64 * RG;
  if (gbase >= G) return;
  float l[RG][NC], m_[RG], vs[RG], v[RG], accU[RG], accUF[RG];
#pragma unroll
  for (int r = 0; r < RG; ++r) {
    const int g = gbase + r * 64 + lane; const bool ok = g < G; const int gg = ok ? g : G - 1;
#pragma unroll
    for (int c = 0; c < NC; ++c) l[r][c] = ok ? Lb[(long)gg * 8 + c] : 0.f;
    m_[r] = ok ? mu[gg] : 0.f; vs[r] = ok ? Vs[gg] : 0.f; v[r] = ok ? V[gg] : 0.f; accU[r] = accUF[r] = 0.f;
  }
  const long n0 = (long)blockIdx.y * cchunk, n1 = std::min(n0 + cchunk, N);
  float keep = 0.f;
  for (long n = n0; n < n1; ++n) {
    float cf[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) cf[c] = coef[n * 8 + c];
    const float f = F[n], em = em2[n];
    float dsum = 0.f;
#pragma unroll
    for (int r = 0; r < RG; ++r) {
      const float e = __builtin_amdgcn_exp2f(fmaf(f, vs[r], -em));
      float t = 0.f;
#pragma unroll
      for (int c = 0; c < NC; ++c) t = fmaf(cf[c], l[r][c], t);
      const float u = e * t;
      accU[r] += u; accUF[r] = fmaf(u, f, accUF[r]);
      dsum = fmaf(m_[r] * u, v[r], dsum);
    }
    const int slot = (int)(n - n0) & 63;
    const float tot = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wave_sum63(dsum)), 63));
    keep = (lane == slot) ? tot : keep;
    if (slot == 63 || n + 1 == n1) { const long fb = n - slot; if (fb + lane <= n) dFpart[(long)tile * N + fb + lane] = keep; }
  }
#pragma unroll
  for (int r = 0; r < RG; ++r) {
    const int g = gbase + r * 64 + lane;
    if (g < G) { float* gp = gpart + ((long)blockIdx.y * G + g) * 2; gp[0] = accU[r]; gp[1] = m_[r] * accUF[r]; }
  }
}
// ---- coef -> three bf16 parts, layout [N16][4][8] (part 3 = 0)
__device__ __forceinline__ unsigned short bf16_rn(float f) {
  unsigned u = __float_as_uint(f); u += 0x7FFFu + ((u >> 16) & 1u); return (unsigned short)(u >> 16);
}
__global__ void split_coef(const float* __restrict__ coef, unsigned short* __restrict__ cq, long N, long N16) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x; if (i >= N16 * 8) return;
  const long n = i >> 3; const int c = i & 7;
  float x = n < N ? coef[n * 8 + c] : 0.f;
  const unsigned short p1 = bf16_rn(x); x -= __uint_as_float((unsigned)p1 << 16);
  const unsigned short p2 = bf16_rn(x); x -= __uint_as_float((unsigned)p2 << 16);
  const unsigned short p3 = bf16_rn(x);
  cq[(n * 4 + 0) * 8 + c] = p1; cq[(n * 4 + 1) * 8 + c] = p2; cq[(n * 4 + 2) * 8 + c] = p3; cq[(n * 4 + 3) * 8 + c] = 0;
}
// ---- MFMA kernel: wave = TL tiles of 16 genes; block = 4 waves; batches of 16 cells
template <int TL>
__global__ void __launch_bounds__(256) bwd_mfma(const unsigned short* __restrict__ cq, const float* __restrict__ F,
                                                const float* __restrict__ em2, const float* __restrict__ Lb,
                                                const float* __restrict__ mu, const float* __restrict__ Vs, const float* __restrict__ V,
                                                float* __restrict__ gpart, float* __restrict__ dFpart, long N, int G, long cchunk) {
  const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
  const int wtile = blockIdx.x * 4 + (threadIdx.x >> 6);      // wave's gene super-tile (TL x 16 genes)
  const int gbase = wtile * TL * 16;
  if (gbase >= G) return;
  bf16x8 Bf[TL];
  float vs[TL], mv[TL], mug[TL], accU[TL], accUF[TL];
#pragma unroll
  for (int m = 0; m < TL; ++m) {
    const int g = gbase + 16 * m + j; const bool ok = g < G; const int gg = ok ? g : G - 1;
    unsigned short b[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) b[c] = (ok && q < 3) ? bf16_rn(Lb[(long)gg * 8 + c]) : 0;   // exact for integer copy numbers
    uint4 raw = {(unsigned)b[0] | ((unsigned)b[1] << 16), (unsigned)b[2] | ((unsigned)b[3] << 16),
                 (unsigned)b[4] | ((unsigned)b[5] << 16), (unsigned)b[6] | ((unsigned)b[7] << 16)};
    Bf[m] = __builtin_bit_cast(bf16x8, raw);
    vs[m] = ok ? Vs[gg] : 0.f; mug[m] = ok ? mu[gg] : 0.f; mv[m] = ok ? mu[gg] * V[gg] : 0.f;
    accU[m] = 0.f; accUF[m] = 0.f;
  }
  const long n0 = (long)blockIdx.y * cchunk, n1 = std::min(n0 + cchunk, N);
  for (long b0 = n0; b0 < n1; b0 += 16) {
    // A fragment: lane (i = j, q) holds part q of cell b0 + j: 8 bf16 = 16 bytes, 1 KiB contiguous per wave
    const uint4 araw = *reinterpret_cast<const uint4*>(cq + ((b0 + j) * 4 + q) * 8);
    const bf16x8 Af = __builtin_bit_cast(bf16x8, araw);
    float f4[4], e4[4], dF[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long n = std::min(b0 + 4 * q + r, N - 1);
      f4[r] = F[n]; e4[r] = em2[n]; dF[r] = 0.f;
    }
#pragma unroll
    for (int m = 0; m < TL; ++m) {
      f32x4 t = {0.f, 0.f, 0.f, 0.f};
      t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Af, Bf[m], t, 0, 0, 0);   // t[r]: cell b0 + 4q + r, gene gbase + 16m + j
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __builtin_amdgcn_exp2f(fmaf(f4[r], vs[m], -e4[r]));
        const float u = e * t[r];
        accU[m] += u; accUF[m] = fmaf(u, f4[r], accUF[m]);
        dF[r] = fmaf(mv[m], u, dF[r]);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) dF[r] = row16_sum(dF[r]);
    const float mine = j == 0 ? dF[0] : j == 1 ? dF[1] : j == 2 ? dF[2] : dF[3];
    const long n = b0 + 4 * q + j;
    if (j < 4 && n < n1) dFpart[(long)wtile * N + n] = mine;
  }
#pragma unroll
  for (int m = 0; m < TL; ++m) {
    float a = accU[m], b = accUF[m];
    a += __shfl_xor(a, 16); a += __shfl_xor(a, 32);
    b += __shfl_xor(b, 16); b += __shfl_xor(b, 32);
    const int g = gbase + 16 * m + j;
    if (q == 0 && g < G) { float* gp = gpart + ((long)blockIdx.y * G + g) * 2; gp[0] = a; gp[1] = mug[m] * b; }
  }
}

template <int TL>
__global__ void __launch_bounds__(256) bwd_mfma2(const unsigned short* __restrict__ cq, const float* __restrict__ F,
                                                 const float* __restrict__ em2, const float* __restrict__ Lb,
                                                 const float* __restrict__ mu, const float* __restrict__ Vs, const float* __restrict__ V,
                                                 float* __restrict__ gpart, float* __restrict__ dFpart, long N, int G, long cchunk) {
  const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
  const int wtile = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int gbase = wtile * TL * 16;
  if (gbase >= G) return;
  bf16x8 Bf[TL];
  float vs[TL], mv[TL], mug[TL], accU[TL], accUF[TL];
#pragma unroll
  for (int m = 0; m < TL; ++m) {
    const int g = gbase + 16 * m + j; const bool ok = g < G; const int gg = ok ? g : G - 1;
    unsigned short b[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) b[c] = (ok && q < 3) ? bf16_rn(Lb[(long)gg * 8 + c]) : 0;
    uint4 raw = {(unsigned)b[0] | ((unsigned)b[1] << 16), (unsigned)b[2] | ((unsigned)b[3] << 16),
                 (unsigned)b[4] | ((unsigned)b[5] << 16), (unsigned)b[6] | ((unsigned)b[7] << 16)};
    Bf[m] = __builtin_bit_cast(bf16x8, raw);
    vs[m] = ok ? Vs[gg] : 0.f; mug[m] = ok ? mu[gg] : 0.f; mv[m] = ok ? mu[gg] * V[gg] : 0.f;
    accU[m] = 0.f; accUF[m] = 0.f;
  }
  const long n0 = (long)blockIdx.y * cchunk, n1 = std::min(n0 + cchunk, N);
  // F / em2 / cq are padded to a multiple of 16 cells, n0 is a multiple of 16: aligned float4 operand loads
  uint4 araw = *reinterpret_cast<const uint4*>(cq + ((n0 + j) * 4 + q) * 8);
  float4 fv = *reinterpret_cast<const float4*>(F + n0 + 4 * q), ev = *reinterpret_cast<const float4*>(em2 + n0 + 4 * q);
  for (long b0 = n0; b0 < n1; b0 += 16) {
    const bf16x8 Af = __builtin_bit_cast(bf16x8, araw);
    const float f4[4] = {fv.x, fv.y, fv.z, fv.w}, e4[4] = {ev.x, ev.y, ev.z, ev.w};
    if (b0 + 16 < n1) {   // prefetch the next batch
      araw = *reinterpret_cast<const uint4*>(cq + ((b0 + 16 + j) * 4 + q) * 8);
      fv = *reinterpret_cast<const float4*>(F + b0 + 16 + 4 * q); ev = *reinterpret_cast<const float4*>(em2 + b0 + 16 + 4 * q);
    }
    float dF[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < TL; ++m) {
      f32x4 t = {0.f, 0.f, 0.f, 0.f};
      t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Af, Bf[m], t, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __builtin_amdgcn_exp2f(fmaf(f4[r], vs[m], -e4[r]));
        const float u = e * t[r];
        accU[m] += u; accUF[m] = fmaf(u, f4[r], accUF[m]);
        dF[r] = fmaf(mv[m], u, dF[r]);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) dF[r] = row16_sum(dF[r]);
    const float mine = j == 0 ? dF[0] : j == 1 ? dF[1] : j == 2 ? dF[2] : dF[3];
    const long n = b0 + 4 * q + j;
    if (j < 4 && n < n1) dFpart[(long)wtile * N + n] = mine;
  }
#pragma unroll
  for (int m = 0; m < TL; ++m) {
    float a = accU[m], b = accUF[m];
    a += __shfl_xor(a, 16); a += __shfl_xor(a, 32);
    b += __shfl_xor(b, 16); b += __shfl_xor(b, 32);
    const int g = gbase + 16 * m + j;
    if (q == 0 && g < G) { float* gp = gpart + ((long)blockIdx.y * G + g) * 2; gp[0] = a; gp[1] = mug[m] * b; }
  }
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
// v3: explicit 2-wide packed math (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32): on gfx950 every VALU instruction costs
// 4 cycles per wave64 whether packed or not (tools/valu_lab.hip), so halving the instruction count halves the time.
template <int TL>
__global__ void __launch_bounds__(256) bwd_mfma3(const unsigned short* __restrict__ cq, const float* __restrict__ F,
                                                 const float* __restrict__ em2, const float* __restrict__ Lb,
                                                 const float* __restrict__ mu, const float* __restrict__ Vs, const float* __restrict__ V,
                                                 float* __restrict__ gpart, float* __restrict__ dFpart, long N, int G, long cchunk) {
  const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
  const int wtile = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int gbase = wtile * TL * 16;
  if (gbase >= G) return;
  bf16x8 Bf[TL];
  float vs[TL], mv[TL], mug[TL];
  f32x2 accU[TL], accUF[TL];
#pragma unroll
  for (int m = 0; m < TL; ++m) {
    const int g = gbase + 16 * m + j; const bool ok = g < G; const int gg = ok ? g : G - 1;
    unsigned short b[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) b[c] = (ok && q < 3) ? bf16_rn(Lb[(long)gg * 8 + c]) : 0;
    uint4 raw = {(unsigned)b[0] | ((unsigned)b[1] << 16), (unsigned)b[2] | ((unsigned)b[3] << 16),
                 (unsigned)b[4] | ((unsigned)b[5] << 16), (unsigned)b[6] | ((unsigned)b[7] << 16)};
    Bf[m] = __builtin_bit_cast(bf16x8, raw);
    vs[m] = ok ? Vs[gg] : 0.f; mug[m] = ok ? mu[gg] : 0.f; mv[m] = ok ? mu[gg] * V[gg] : 0.f;
    accU[m] = (f32x2){0.f, 0.f}; accUF[m] = (f32x2){0.f, 0.f};
  }
  const long n0 = (long)blockIdx.y * cchunk, n1 = std::min(n0 + cchunk, N);
  for (long b0 = n0; b0 < n1; b0 += 16) {
    const uint4 araw = *reinterpret_cast<const uint4*>(cq + ((b0 + j) * 4 + q) * 8);
    const bf16x8 Af = __builtin_bit_cast(bf16x8, araw);
    const float4 fv = *reinterpret_cast<const float4*>(F + b0 + 4 * q), ev = *reinterpret_cast<const float4*>(em2 + b0 + 4 * q);
    const f32x2 f2[2] = {{fv.x, fv.y}, {fv.z, fv.w}}, e2[2] = {{ev.x, ev.y}, {ev.z, ev.w}};
    f32x2 dF[2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
    for (int m = 0; m < TL; ++m) {
      f32x4 t = {0.f, 0.f, 0.f, 0.f};
      t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Af, Bf[m], t, 0, 0, 0);
      const f32x2 t2[2] = {{t[0], t[1]}, {t[2], t[3]}};
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const f32x2 eta = f2[h] * vs[m] - e2[h];
        const f32x2 ex = {__builtin_amdgcn_exp2f(eta.x), __builtin_amdgcn_exp2f(eta.y)};
        const f32x2 u = ex * t2[h];
        accU[m] += u;
        accUF[m] = u * f2[h] + accUF[m];
        dF[h] = u * mv[m] + dF[h];
      }
    }
    float d4[4] = {dF[0].x, dF[0].y, dF[1].x, dF[1].y};
#pragma unroll
    for (int r = 0; r < 4; ++r) d4[r] = row16_sum(d4[r]);
    const float mine = j == 0 ? d4[0] : j == 1 ? d4[1] : j == 2 ? d4[2] : d4[3];
    const long n = b0 + 4 * q + j;
    if (j < 4 && n < n1) dFpart[(long)wtile * N + n] = mine;
  }
#pragma unroll
  for (int m = 0; m < TL; ++m) {
    float a = accU[m].x + accU[m].y, b = accUF[m].x + accUF[m].y;
    a += __shfl_xor(a, 16); a += __shfl_xor(a, 32);
    b += __shfl_xor(b, 16); b += __shfl_xor(b, 32);
    const int g = gbase + 16 * m + j;
    if (q == 0 && g < G) { float* gp = gpart + ((long)blockIdx.y * G + g) * 2; gp[0] = a; gp[1] = mug[m] * b; }
  }
}

// v4: operands swapped (rows = genes, columns = cells): each lane owns ONE cell per batch, so d/dF needs only a
// 4-lane-group sum per batch; the per-gene sums stay in-lane for the whole cell slice and are row-reduced once.
template <int TL, bool PF, int ABL = 0>
__global__ void __launch_bounds__(256) bwd_mfma4(const unsigned short* __restrict__ cq, const float* __restrict__ F,
                                                 const float* __restrict__ em2, const float* __restrict__ Lb,
                                                 const float* __restrict__ mu, const float* __restrict__ Vs, const float* __restrict__ V,
                                                 float* __restrict__ gpart, float* __restrict__ dFpart, long N, int G, long cchunk) {
  const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
  const int wtile = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int gbase = wtile * TL * 16;
  if (gbase >= G) return;
  bf16x8 Lf[TL];
  f32x2 vs[TL][2], mv[TL][2], accU[TL][2], accUF[TL][2];
#pragma unroll
  for (int m = 0; m < TL; ++m) {
    {   // MFMA A operand: lane (row i = j, k-group q) holds L[gene gbase+16m+j][0..8) for the three coef parts
      const int g = gbase + 16 * m + j; const bool ok = g < G; const int gg = ok ? g : G - 1;
      unsigned short b[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) b[c] = (ok && q < 3) ? bf16_rn(Lb[(long)gg * 8 + c]) : 0;
      uint4 raw = {(unsigned)b[0] | ((unsigned)b[1] << 16), (unsigned)b[2] | ((unsigned)b[3] << 16),
                   (unsigned)b[4] | ((unsigned)b[5] << 16), (unsigned)b[6] | ((unsigned)b[7] << 16)};
      Lf[m] = __builtin_bit_cast(bf16x8, raw);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {   // output rows of this lane: genes gbase + 16m + 4q + {2h, 2h+1}
      float a[2], b[2];
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        const int g = gbase + 16 * m + 4 * q + 2 * h + x; const bool ok = g < G; const int gg = ok ? g : G - 1;
        a[x] = ok ? Vs[gg] : 0.f; b[x] = ok ? mu[gg] * V[gg] : 0.f;
      }
      vs[m][h] = (f32x2){a[0], a[1]}; mv[m][h] = (f32x2){b[0], b[1]};
      accU[m][h] = (f32x2){0.f, 0.f}; accUF[m][h] = (f32x2){0.f, 0.f};
    }
  }
  const long n0 = (long)blockIdx.y * cchunk, n1 = std::min(n0 + cchunk, N);
  uint4 craw = *reinterpret_cast<const uint4*>(cq + ((n0 + j) * 4 + q) * 8);
  float f = F[n0 + j], em = em2[n0 + j];
  for (long b0 = n0; b0 < n1; b0 += 16) {
    const bf16x8 Cf = __builtin_bit_cast(bf16x8, craw);
    const float fc = f, ec = em;
    if (PF && b0 + 16 < n1) {
      craw = *reinterpret_cast<const uint4*>(cq + ((b0 + 16 + j) * 4 + q) * 8);
      f = F[b0 + 16 + j]; em = em2[b0 + 16 + j];
    }
    f32x2 dF = {0.f, 0.f};
#pragma unroll
    for (int m = 0; m < TL; ++m) {
      f32x4 t = {0.f, 0.f, 0.f, 0.f};
      if (ABL == 2) { t[0] = __uint_as_float(craw.x); t[1] = __uint_as_float(craw.y); t[2] = __uint_as_float(craw.z); t[3] = __uint_as_float(craw.w); }
      else t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Lf[m], Cf, t, 0, 0, 0);   // t[r]: gene gbase+16m+4q+r, cell b0+j
      const f32x2 t2[2] = {{t[0], t[1]}, {t[2], t[3]}};
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const f32x2 eta = vs[m][h] * fc - ec;
        const f32x2 ex = ABL == 1 ? eta : (f32x2){__builtin_amdgcn_exp2f(eta.x), __builtin_amdgcn_exp2f(eta.y)};
        const f32x2 u = ex * t2[h];
        if (ABL != 4) {
        accU[m][h] += u;
        accUF[m][h] = u * fc + accUF[m][h];
        } else accU[m][h] = u;
        if (ABL != 3) dF = u * mv[m][h] + dF; else dF = u;
      }
    }
    if (ABL == 3) { if (dF.x == 123.f) dFpart[b0] = dF.y; continue; }
    if (!PF && b0 + 16 < n1) {
      craw = *reinterpret_cast<const uint4*>(cq + ((b0 + 16 + j) * 4 + q) * 8);
      f = F[b0 + 16 + j]; em = em2[b0 + 16 + j];
    }
    float d = dF.x + dF.y;
    d += __shfl_xor(d, 16); d += __shfl_xor(d, 32);
    const long n = b0 + j;
    if (q == 0 && n < n1) dFpart[(long)wtile * N + n] = d;
  }
#pragma unroll
  for (int m = 0; m < TL; ++m)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float a0 = row16_sum(accU[m][h].x), a1 = row16_sum(accU[m][h].y);
      const float b0_ = row16_sum(accUF[m][h].x), b1 = row16_sum(accUF[m][h].y);
      if (j < 2) {
        const int g = gbase + 16 * m + 4 * q + 2 * h + j;
        if (g < G) { float* gp = gpart + ((long)blockIdx.y * G + g) * 2; gp[0] = j ? a1 : a0; gp[1] = mu[g] * (j ? b1 : b0_); }
      }
    }
}
// v5: MFMAs of batch b+1 issued before the VALU work of batch b (results parked in 16 VGPRs), operands two batches ahead
template <int TL, int TAIL>
__global__ void __launch_bounds__(256) bwd_mfma5(const unsigned short* __restrict__ cq, const float* __restrict__ F,
                                                 const float* __restrict__ em2, const float* __restrict__ Lb,
                                                 const float* __restrict__ mu, const float* __restrict__ Vs, const float* __restrict__ V,
                                                 float* __restrict__ gpart, float* __restrict__ dFpart, long N, int G, long cchunk) {
  const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
  const int wtile = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int gbase = wtile * TL * 16;
  if (gbase >= G) return;
  bf16x8 Lf[TL];
  f32x2 vs[TL][2], mv[TL][2], accU[TL][2], accUF[TL][2];
#pragma unroll
  for (int m = 0; m < TL; ++m) {
    {
      const int g = gbase + 16 * m + j; const bool ok = g < G; const int gg = ok ? g : G - 1;
      unsigned short b[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) b[c] = (ok && q < 3) ? bf16_rn(Lb[(long)gg * 8 + c]) : 0;
      uint4 raw = {(unsigned)b[0] | ((unsigned)b[1] << 16), (unsigned)b[2] | ((unsigned)b[3] << 16),
                   (unsigned)b[4] | ((unsigned)b[5] << 16), (unsigned)b[6] | ((unsigned)b[7] << 16)};
      Lf[m] = __builtin_bit_cast(bf16x8, raw);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float a[2], b[2];
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        const int g = gbase + 16 * m + 4 * q + 2 * h + x; const bool ok = g < G; const int gg = ok ? g : G - 1;
        a[x] = ok ? Vs[gg] : 0.f; b[x] = ok ? mu[gg] * V[gg] : 0.f;
      }
      vs[m][h] = (f32x2){a[0], a[1]}; mv[m][h] = (f32x2){b[0], b[1]};
      accU[m][h] = (f32x2){0.f, 0.f}; accUF[m][h] = (f32x2){0.f, 0.f};
    }
  }
  const long n0 = (long)blockIdx.y * cchunk, n1 = std::min(n0 + cchunk, N);
  auto ldc = [&](long b0) { return *reinterpret_cast<const uint4*>(cq + ((b0 + j) * 4 + q) * 8); };
  f32x4 t[TL];
  {
    const bf16x8 Cf = __builtin_bit_cast(bf16x8, ldc(n0));
#pragma unroll
    for (int m = 0; m < TL; ++m) t[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Lf[m], Cf, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
  }
  float f = F[n0 + j], em = em2[n0 + j];
  uint4 craw = (n0 + 16 < n1) ? ldc(n0 + 16) : (uint4){0, 0, 0, 0};
  for (long b0 = n0; b0 < n1; b0 += 16) {
    const float fc = f, ec = em;
    f32x4 tc[TL];
#pragma unroll
    for (int m = 0; m < TL; ++m) tc[m] = t[m];
    const bf16x8 Cf = __builtin_bit_cast(bf16x8, craw);
    if (b0 + 16 < n1) {
#pragma unroll
      for (int m = 0; m < TL; ++m) t[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Lf[m], Cf, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      f = F[b0 + 16 + j]; em = em2[b0 + 16 + j];
    }
    if (b0 + 32 < n1) craw = ldc(b0 + 32);
    f32x2 dF = {0.f, 0.f};
#pragma unroll
    for (int m = 0; m < TL; ++m) {
      const f32x2 t2[2] = {{tc[m][0], tc[m][1]}, {tc[m][2], tc[m][3]}};
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const f32x2 eta = vs[m][h] * fc - ec;
        const f32x2 ex = {__builtin_amdgcn_exp2f(eta.x), __builtin_amdgcn_exp2f(eta.y)};
        const f32x2 u = ex * t2[h];
        accU[m][h] += u;
        accUF[m][h] = u * fc + accUF[m][h];
        dF = u * mv[m][h] + dF;
      }
    }
    float d = dF.x + dF.y;
    if (TAIL == 0) {
      d += __shfl_xor(d, 16); d += __shfl_xor(d, 32);
      const long n = b0 + j;
      if (q == 0 && n < n1) dFpart[(long)wtile * N + n] = d;
    } else if (TAIL == 1) {   // DPP row_bcast-free: two v_permlane-like swaps via ds_swizzle are not available across 16; use readlane-free adds
      d += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, d), 0));   // placeholder cost probe
      const long n = b0 + j;
      if (q == 0 && n < n1) dFpart[(long)wtile * N + n] = d;
    } else {                  // no cross-lane sum: every k-group writes its own partial (4x the d/dF partial traffic)
      const long n = b0 + j;
      if (n < n1) dFpart[((long)wtile * 4 + q) * N + n] = d;
    }
  }
#pragma unroll
  for (int m = 0; m < TL; ++m)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float a0 = row16_sum(accU[m][h].x), a1 = row16_sum(accU[m][h].y);
      const float b0_ = row16_sum(accUF[m][h].x), b1 = row16_sum(accUF[m][h].y);
      if (j < 2) {
        const int g = gbase + 16 * m + 4 * q + 2 * h + j;
        if (g < G) { float* gp = gpart + ((long)blockIdx.y * G + g) * 2; gp[0] = j ? a1 : a0; gp[1] = mu[g] * (j ? b1 : b0_); }
      }
    }
}
__device__ unsigned long long g_stamp[4096 * 2];
__device__ unsigned long long g_abs[4096 * 2];
__device__ unsigned long long g_ent[4096 * 2];
// v6: v4 with scalar (non-packed) f32 math: VOP2 v_mul/v_add/v_fmac issue at 2 cycles per wave64 with >= 2 waves per SIMD
// and do not pay the packed-op penalty beside MFMAs
template <int TL>
__global__ void __launch_bounds__(256) bwd_mfma6(const unsigned short* __restrict__ cq, const float* __restrict__ F,
                                                 const float* __restrict__ em2, const float* __restrict__ Lb,
                                                 const float* __restrict__ mu, const float* __restrict__ Vs, const float* __restrict__ V,
                                                 float* __restrict__ gpart, float* __restrict__ dFpart, long N, int G, long cchunk) {
  const unsigned long long t_entry = __builtin_amdgcn_s_memrealtime();
  const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
  const int wtile = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int gbase = wtile * TL * 16;
  if (gbase >= G) return;
  bf16x8 Lf[TL];
  float vs[TL][4], mv[TL][4], accU[TL][4], accUF[TL][4];
#pragma unroll
  for (int m = 0; m < TL; ++m) {
    {
      const int g = gbase + 16 * m + j; const bool ok = g < G && q < 3; const int gg = g < G ? g : G - 1;
      const float4 r0 = *reinterpret_cast<const float4*>(Lb + (long)gg * 8), r1 = *reinterpret_cast<const float4*>(Lb + (long)gg * 8 + 4);
      const float lr[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
      unsigned short b[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) b[c] = bf16_rn(lr[c]);
      const unsigned msk = ok ? 0xFFFFFFFFu : 0u;
      uint4 raw = {((unsigned)b[0] | ((unsigned)b[1] << 16)) & msk, ((unsigned)b[2] | ((unsigned)b[3] << 16)) & msk,
                   ((unsigned)b[4] | ((unsigned)b[5] << 16)) & msk, ((unsigned)b[6] | ((unsigned)b[7] << 16)) & msk};
      Lf[m] = __builtin_bit_cast(bf16x8, raw);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int g = gbase + 16 * m + 4 * q + r; const bool ok = g < G; const int gg = ok ? g : G - 1;
      const float a_ = Vs[gg], b_ = mu[gg], c_ = V[gg];
      vs[m][r] = ok ? a_ : 0.f; mv[m][r] = ok ? b_ * c_ : 0.f;
      accU[m][r] = 0.f; accUF[m][r] = 0.f;
    }
  }
  const long n0 = (long)blockIdx.y * cchunk, n1 = std::min(n0 + cchunk, N);
  uint4 craw = *reinterpret_cast<const uint4*>(cq + ((n0 + j) * 4 + q) * 8);
  float f = F[n0 + j], em = em2[n0 + j];
  const unsigned long long st0 = __builtin_amdgcn_s_memtime(), sr0 = __builtin_amdgcn_s_memrealtime();
  for (long b0 = n0; b0 < n1; b0 += 16) {
    const bf16x8 Cf = __builtin_bit_cast(bf16x8, craw);
    const float fc = f, ec = em;
    if (b0 + 16 < n1) {
      craw = *reinterpret_cast<const uint4*>(cq + ((b0 + 16 + j) * 4 + q) * 8);
      f = F[b0 + 16 + j]; em = em2[b0 + 16 + j];
    }
    float dF = 0.f;
#pragma unroll
    for (int m = 0; m < TL; ++m) {
      f32x4 t = {0.f, 0.f, 0.f, 0.f};
      t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Lf[m], Cf, t, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float ex = __builtin_amdgcn_exp2f(fmaf(vs[m][r], fc, -ec));
        const float u = ex * t[r];
        accU[m][r] += u;
        accUF[m][r] = fmaf(u, fc, accUF[m][r]);
        dF = fmaf(u, mv[m][r], dF);
      }
    }
    float d = dF;
    d += __shfl_xor(d, 16); d += __shfl_xor(d, 32);
    const long n = b0 + j;
    if (q == 0 && n < n1) dFpart[(long)wtile * N + n] = d;
  }
  {
    const unsigned long long st1 = __builtin_amdgcn_s_memtime(), sr1 = __builtin_amdgcn_s_memrealtime();
    const int bid = blockIdx.y * gridDim.x + blockIdx.x;
    if (threadIdx.x == 0 && bid < 4096) { g_stamp[2 * bid] = st1 - st0; g_stamp[2 * bid + 1] = sr1 - sr0; g_abs[2 * bid] = sr0; g_abs[2 * bid + 1] = sr1; g_ent[2 * bid] = t_entry; }
  }
#pragma unroll
  for (int m = 0; m < TL; ++m)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float a0 = row16_sum(accU[m][2 * h]), a1 = row16_sum(accU[m][2 * h + 1]);
      const float b0_ = row16_sum(accUF[m][2 * h]), b1 = row16_sum(accUF[m][2 * h + 1]);
      if (j < 2) {
        const int g = gbase + 16 * m + 4 * q + 2 * h + j;
        if (g < G) { float* gp = gpart + ((long)blockIdx.y * G + g) * 2; gp[0] = j ? a1 : a0; gp[1] = mu[g] * (j ? b1 : b0_); }
      }
    }
  { const int bid = blockIdx.y * gridDim.x + blockIdx.x; if (threadIdx.x == 0 && bid < 4096) g_ent[2 * bid + 1] = __builtin_amdgcn_s_memrealtime(); }
}
int main() {
  long N = 100000; int G = 5000; const long N16 = (N + 15) / 16 * 16;
  std::vector<float> F(N), em(N), Vs(G), V(G), mu(G), Lb((size_t)G * 8), coef((size_t)N * 8);
  srand(1); auto rnd = []() { return (float)rand() / RAND_MAX; };
  float vmin = 1e9, vmax = -1e9;
  for (int g = 0; g < G; ++g) { V[g] = (rnd() - 0.5f) * 0.8f; Vs[g] = V[g] * 1.442695f; vmin = std::min(vmin, Vs[g]); vmax = std::max(vmax, Vs[g]); mu[g] = rnd() + 0.1f; }
  for (long i = 0; i < N; ++i) { F[i] = (rnd() - 0.5f) * 4.f; em[i] = std::max(F[i] * vmin, F[i] * vmax); }
  for (auto& x : Lb) x = 1.f + (int)(rnd() * 3.99f);
  for (auto& x : coef) x = -rnd() * 1e-3f;
  float *dc, *dF, *dem, *dL, *dmu, *dVs, *dV, *dg, *ddF; unsigned short* dq;
  CK(hipMalloc(&dc, N * 32)); CK(hipMalloc(&dF, N16 * 4)); CK(hipMalloc(&dem, N16 * 4)); CK(hipMemset(dF, 0, N16 * 4)); CK(hipMemset(dem, 0, N16 * 4)); CK(hipMalloc(&dL, (size_t)G * 32)); CK(hipMalloc(&dmu, G * 4));
  CK(hipMalloc(&dVs, G * 4)); CK(hipMalloc(&dV, G * 4)); CK(hipMalloc(&dg, (size_t)2048 * G * 8)); CK(hipMalloc(&ddF, (size_t)512 * N * 4));
  CK(hipMalloc(&dq, (size_t)N16 * 64));
  CK(hipMemcpy(dc, coef.data(), N * 32, hipMemcpyHostToDevice)); CK(hipMemcpy(dF, F.data(), N * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dem, em.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dL, Lb.data(), (size_t)G * 32, hipMemcpyHostToDevice));
  CK(hipMemcpy(dmu, mu.data(), G * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dVs, Vs.data(), G * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dV, V.data(), G * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(split_coef, dim3((N16 * 8 + 255) / 256), dim3(256), 0, 0, dc, dq, N, N16); CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<double> refg, refF;
  auto check = [&](int ntile, int csplit, const char* name, float ms) {
    std::vector<float> g((size_t)csplit * G * 2), d((size_t)ntile * N);
    CK(hipMemcpy(g.data(), dg, g.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(d.data(), ddF, d.size() * 4, hipMemcpyDeviceToHost));
    std::vector<double> sg((size_t)G * 2, 0.0), sF(N, 0.0);
    for (int s = 0; s < csplit; ++s) for (size_t i = 0; i < sg.size(); ++i) sg[i] += g[(size_t)s * G * 2 + i];
    for (int t = 0; t < ntile; ++t) for (long n = 0; n < N; ++n) sF[n] += d[(size_t)t * N + n];
    if (refg.empty()) { refg = sg; refF = sF; }
    double eg = 0, eF = 0, mg = 0, mF = 0;
    for (size_t i = 0; i < sg.size(); ++i) { eg = std::max(eg, std::fabs(sg[i] - refg[i])); mg = std::max(mg, std::fabs(refg[i])); }
    for (long n = 0; n < N; ++n) { eF = std::max(eF, std::fabs(sF[n] - refF[n])); mF = std::max(mF, std::fabs(refF[n])); }
    printf("%-26s csplit %4d %8.1f us   err gene %.1e cell %.1e\n", name, csplit, ms * 1e3, eg / mg, eF / mF);
  };
  {
    long cchunk = (N + 409) / 410; int cs = (int)((N + cchunk - 1) / cchunk); const int ntile = (G + 255) / 256; dim3 grid((ntile + 3) / 4, cs); float best = 1e9;
    for (int it = 0; it < 4; ++it) { CK(hipMemset(ddF, 0, (size_t)ntile * N * 4)); CK(hipEventRecord(e0));
      hipLaunchKernelGGL(bwd_valu<4>, grid, dim3(256), 0, 0, dc, dF, dem, dL, dmu, dVs, dV, dg, ddF, N, G, cchunk);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError()); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (it) best = std::min(best, ms); }
    check(ntile, cs, "VALU RG=4", best);
  }
#define RUNM(TL, csplit_req)                                                                                          \
  { long cchunk = (N + (csplit_req) - 1) / (csplit_req); cchunk = (cchunk + 15) / 16 * 16; int cs = (int)((N + cchunk - 1) / cchunk); \
    const int nwt = (G + TL * 16 - 1) / (TL * 16); dim3 grid((nwt + 3) / 4, cs); float best = 1e9;                       \
    for (int it = 0; it < 4; ++it) { CK(hipMemset(ddF, 0, (size_t)nwt * N * 4)); CK(hipEventRecord(e0));                  \
      hipLaunchKernelGGL(bwd_mfma<TL>, grid, dim3(256), 0, 0, dq, dF, dem, dL, dmu, dVs, dV, dg, ddF, N, G, cchunk);    \
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError()); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (it) best = std::min(best, ms); } \
    char nm[64]; snprintf(nm, 64, "MFMA bf16x3 TL=%d", TL); check(nwt, cs, nm, best); }
  RUNM(8, 200);
#define RUNM2(TL, csplit_req)                                                                                         \
  { long cchunk = (N + (csplit_req) - 1) / (csplit_req); cchunk = (cchunk + 15) / 16 * 16; int cs = (int)((N + cchunk - 1) / cchunk); \
    const int nwt = (G + TL * 16 - 1) / (TL * 16); dim3 grid((nwt + 3) / 4, cs); float best = 1e9;                       \
    for (int it = 0; it < 4; ++it) { CK(hipMemset(ddF, 0, (size_t)nwt * N * 4)); CK(hipEventRecord(e0));                  \
      hipLaunchKernelGGL(bwd_mfma2<TL>, grid, dim3(256), 0, 0, dq, dF, dem, dL, dmu, dVs, dV, dg, ddF, N, G, cchunk);   \
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError()); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (it) best = std::min(best, ms); } \
    char nm[64]; snprintf(nm, 64, "MFMA v2 prefetch TL=%d", TL); check(nwt, cs, nm, best); }
  RUNM2(8, 100);
#define RUNM3(TL, csplit_req)                                                                                         \
  { long cchunk = (N + (csplit_req) - 1) / (csplit_req); cchunk = (cchunk + 15) / 16 * 16; int cs = (int)((N + cchunk - 1) / cchunk); \
    const int nwt = (G + TL * 16 - 1) / (TL * 16); dim3 grid((nwt + 3) / 4, cs); float best = 1e9;                       \
    for (int it = 0; it < 4; ++it) { CK(hipMemset(ddF, 0, (size_t)nwt * N * 4)); CK(hipEventRecord(e0));                  \
      hipLaunchKernelGGL(bwd_mfma3<TL>, grid, dim3(256), 0, 0, dq, dF, dem, dL, dmu, dVs, dV, dg, ddF, N, G, cchunk);   \
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError()); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (it) best = std::min(best, ms); } \
    char nm[64]; snprintf(nm, 64, "MFMA v3 packed TL=%d", TL); check(nwt, cs, nm, best); }
  RUNM3(4, 100);
#define RUNM4(TL, PF, csplit_req)                                                                                     \
  { long cchunk = (N + (csplit_req) - 1) / (csplit_req); cchunk = (cchunk + 15) / 16 * 16; int cs = (int)((N + cchunk - 1) / cchunk); \
    const int nwt = (G + TL * 16 - 1) / (TL * 16); dim3 grid((nwt + 3) / 4, cs); float best = 1e9;                       \
    for (int it = 0; it < 4; ++it) { CK(hipMemset(ddF, 0, (size_t)nwt * N * 4)); CK(hipEventRecord(e0));                  \
      hipLaunchKernelGGL((bwd_mfma4<TL, PF>), grid, dim3(256), 0, 0, dq, dF, dem, dL, dmu, dVs, dV, dg, ddF, N, G, cchunk);   \
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError()); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (it) best = std::min(best, ms); } \
    char nm[64]; snprintf(nm, 64, "MFMA v4 transposed TL=%d pf=%d", TL, (int)PF); check(nwt, cs, nm, best); }
  RUNM4(4, true, 51); RUNM4(4, true, 100);
#define RUNM4A(TL, PF, AB, csplit_req)                                                                                     \
  { long cchunk = (N + (csplit_req) - 1) / (csplit_req); cchunk = (cchunk + 15) / 16 * 16; int cs = (int)((N + cchunk - 1) / cchunk); \
    const int nwt = (G + TL * 16 - 1) / (TL * 16); dim3 grid((nwt + 3) / 4, cs); float best = 1e9;                       \
    for (int it = 0; it < 4; ++it) { CK(hipMemset(ddF, 0, (size_t)nwt * N * 4)); CK(hipEventRecord(e0));                  \
      hipLaunchKernelGGL((bwd_mfma4<TL, PF, AB>), grid, dim3(256), 0, 0, dq, dF, dem, dL, dmu, dVs, dV, dg, ddF, N, G, cchunk);   \
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError()); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (it) best = std::min(best, ms); } \
    char nm[64]; snprintf(nm, 64, "v4 TL=%d abl=%d", TL, AB); check(nwt, cs, nm, best); }
  RUNM4(2, true, 51); RUNM4(2, true, 102); RUNM4(8, true, 51); RUNM4(8, true, 26); RUNM4(3, true, 51); RUNM4(6, true, 51);
#define RUNM5(TL, TAILV, csplit_req)                                                                                     \
  { long cchunk = (N + (csplit_req) - 1) / (csplit_req); cchunk = (cchunk + 15) / 16 * 16; int cs = (int)((N + cchunk - 1) / cchunk); \
    const int nwt = (G + TL * 16 - 1) / (TL * 16); dim3 grid((nwt + 3) / 4, cs); float best = 1e9;                       \
    for (int it = 0; it < 4; ++it) { CK(hipMemset(ddF, 0, (size_t)nwt * N * 4)); CK(hipEventRecord(e0));                  \
      hipLaunchKernelGGL((bwd_mfma5<TL, TAILV>), grid, dim3(256), 0, 0, dq, dF, dem, dL, dmu, dVs, dV, dg, ddF, N, G, cchunk);   \
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError()); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (it) best = std::min(best, ms); } \
    char nm[64]; snprintf(nm, 64, "v5 pipelined TL=%d tail=%d", TL, TAILV); check(nwt, cs, nm, best); }
#define RUNM6(TL, csplit_req)                                                                                     \
  { long cchunk = (N + (csplit_req) - 1) / (csplit_req); cchunk = (cchunk + 15) / 16 * 16; int cs = (int)((N + cchunk - 1) / cchunk); \
    const int nwt = (G + TL * 16 - 1) / (TL * 16); dim3 grid((nwt + 3) / 4, cs); float best = 1e9;                       \
    for (int it = 0; it < 4; ++it) { CK(hipMemset(ddF, 0, (size_t)nwt * N * 4)); CK(hipEventRecord(e0));                  \
      hipLaunchKernelGGL((bwd_mfma6<TL>), grid, dim3(256), 0, 0, dq, dF, dem, dL, dmu, dVs, dV, dg, ddF, N, G, cchunk);   \
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError()); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (it) best = std::min(best, ms); } \
    char nm[64]; snprintf(nm, 64, "v6 scalar TL=%d", TL); check(nwt, cs, nm, best); \
    { std::vector<unsigned long long> st(4096 * 2); CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamp), st.size() * 8)); \
      const int nb = std::min(4096, (int)(grid.x * grid.y)); std::vector<double> ck, cyc; for (int i = 0; i < nb; ++i) if (st[2*i+1] > 0) { ck.push_back((double)st[2*i] / (double)st[2*i+1] * 100.0); cyc.push_back((double)st[2*i]); } \
      std::sort(ck.begin(), ck.end()); std::sort(cyc.begin(), cyc.end()); \
      { std::vector<unsigned long long> ab(4096 * 2); CK(hipMemcpyFromSymbol(ab.data(), HIP_SYMBOL(g_abs), ab.size() * 8)); \
        unsigned long long t0 = ~0ull, t1 = 0; for (int i = 0; i < nb; ++i) { t0 = std::min(t0, ab[2*i]); t1 = std::max(t1, ab[2*i+1]); } \
        int late = 0; double mid = 0; for (int i = 0; i < nb; ++i) { if (ab[2*i] - t0 > 2000) ++late; } \
        int maxc = 0; for (int i = 0; i < nb; i += 7) { int c = 0; for (int k = 0; k < nb; ++k) if (ab[2*k] <= ab[2*i] && ab[2*k+1] > ab[2*i]) ++c; maxc = std::max(maxc, c); } \
        { std::vector<double> du; for (int i = 0; i < nb; ++i) du.push_back((ab[2*i+1] - ab[2*i]) / 100.0); std::vector<double> sd = du; std::sort(sd.begin(), sd.end()); \
          printf("   loop us: p5 %.0f p25 %.0f p50 %.0f p75 %.0f p95 %.0f max %.0f;", sd[nb/20], sd[nb/4], sd[nb/2], sd[3*nb/4], sd[19*nb/20], sd[nb-1]); \
          double xs[8] = {0}; int xc[8] = {0}; for (int i = 0; i < nb; ++i) { xs[i % 8] += du[i]; xc[i % 8]++; } printf(" mean by bid%%8:"); for (int x = 0; x < 8; ++x) printf(" %.0f", xs[x] / xc[x]); \
          const int gx = (int)grid.x; double bs[64] = {0}; int bc[64] = {0}; for (int i = 0; i < nb; ++i) { bs[(i % gx) % 64] += du[i]; bc[(i % gx) % 64]++; } printf("\n   mean by blockIdx.x:"); for (int x = 0; x < gx && x < 64; ++x) printf(" %.0f", bs[x] / bc[x]); printf("\n"); } \
        { std::vector<unsigned long long> en(4096 * 2); CK(hipMemcpyFromSymbol(en.data(), HIP_SYMBOL(g_ent), en.size() * 8)); \
          std::vector<double> pro, epi; for (int i = 0; i < nb; ++i) { pro.push_back((ab[2*i] - en[2*i]) / 100.0); epi.push_back((en[2*i+1] - ab[2*i+1]) / 100.0); } \
          std::sort(pro.begin(), pro.end()); std::sort(epi.begin(), epi.end()); \
          unsigned long long e0 = ~0ull, e1 = 0; for (int i = 0; i < nb; ++i) { e0 = std::min(e0, en[2*i]); e1 = std::max(e1, en[2*i+1]); } \
          printf("   prologue us p50 %.1f p95 %.1f; epilogue us p50 %.1f p95 %.1f; first entry -> last exit %.1f us\n", pro[nb/2], pro[19*nb/20], epi[nb/2], epi[19*nb/20], (e1 - e0) / 100.0); } \
        printf("   blocks %d: span %.1f us, %d blocks started > 20 us after the first, max concurrent blocks %d\n", nb, (t1 - t0) / 100.0, late, maxc); (void)mid; } \
      printf("   in-kernel clock median %.0f MHz (min %.0f max %.0f), loop cycles median %.0f, batches per wave %ld -> %.0f cycles per batch per wave\n", ck[ck.size()/2], ck.front(), ck.back(), cyc[cyc.size()/2], (long)(cchunk / 16), cyc[cyc.size()/2] / (cchunk / 16)); } }
  RUNM6(4, 51); RUNM6(4, 100); RUNM6(4, 204); RUNM6(4, 391);
  RUNM4A(4, true, 1, 51); RUNM4A(4, true, 2, 51); RUNM4A(4, true, 3, 51); RUNM4A(4, true, 4, 51);
  return 0;
}

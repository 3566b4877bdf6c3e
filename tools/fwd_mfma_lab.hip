// Kernel lab: forward sweep Z = E.M with 16 columns (fused two-eps pass, C = 8) on the bf16 matrix cores.
// E_ng = exp2(psi'_n W'_g - shift_n) is generated in the MFMA A-operand layout (lane = 1 cell x 8 consecutive genes),
// split into NE bf16 parts (NE = 2: 2^-18 relative, NE = 3: fp32-exact), M pre-split into 3 bf16 parts in the B-operand
// layout.  Compared against the packed-VALU LDS kernel of the library and a float64 host reference on a cell sample.
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -o tools/fwd_mfma_lab.bin tools/fwd_mfma_lab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
constexpr int NC = 16;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---- reference VALU kernel (library k_fwd_lds<16,1,2,16>)
template <int R>
__global__ void __launch_bounds__(256) fwd_valu(const float* __restrict__ F, const float* __restrict__ em2, const float* __restrict__ Vs,
                                                const float* __restrict__ M, float* __restrict__ Zp, long N, int G, int gchunk) {
  extern __shared__ float lds[];
  const int g0 = blockIdx.y * gchunk;
  const int ng = min(G, g0 + gchunk) - g0;
  float4* l4 = reinterpret_cast<float4*>(lds);
  const float4* m4 = reinterpret_cast<const float4*>(M + (long)g0 * NC);
  for (int i = threadIdx.x; i < ng * 4; i += 256) l4[i] = m4[i];
  float* lv = lds + (long)gchunk * NC;
  for (int i = threadIdx.x; i < ng; i += 256) lv[i] = Vs[g0 + i];
  __syncthreads();
  const long nb = (long)blockIdx.x * 256 * R + threadIdx.x;
  float f[R], em[R], z[R][NC];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const long n = nb + r * 256; const long nn = n < N ? n : N - 1;
    f[r] = F[nn]; em[r] = em2[nn];
#pragma unroll
    for (int c = 0; c < NC; ++c) z[r][c] = 0.f;
  }
#pragma unroll 4
  for (int g = 0; g < ng; ++g) {
    float m[NC];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float4 a = l4[4 * g + j]; m[4*j] = a.x; m[4*j+1] = a.y; m[4*j+2] = a.z; m[4*j+3] = a.w; }
    const float v = lv[g];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float e = __builtin_amdgcn_exp2f(fmaf(f[r], v, -em[r]));
#pragma unroll
      for (int c = 0; c < NC; ++c) z[r][c] = fmaf(e, m[c], z[r][c]);
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const long n = nb + r * 256;
    if (n < N) { float* zp = Zp + ((long)blockIdx.y * N + n) * NC;
#pragma unroll
      for (int c = 0; c < NC; ++c) zp[c] = z[r][c]; }
  }
}

// ---- M -> 3 bf16 parts in B-operand layout: Mq[kstep][part][lane = 16 q + j][8]  (gene = 32 kstep + 8 q + i, column j)
__device__ __forceinline__ unsigned short bf16_rn(float f) {
  unsigned u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__global__ void split_m(const float* __restrict__ M, unsigned short* __restrict__ Mq, int G, int nk) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per (kstep, lane)
  if (t >= (long)nk * 64) return;
  const int ks = (int)(t >> 6), lane = (int)(t & 63), j = lane & 15, q = lane >> 4;
  unsigned short p[3][8];
  for (int i = 0; i < 8; ++i) {
    const int g = 32 * ks + 8 * q + i;
    float x = g < G ? M[(long)g * NC + j] : 0.f;
    for (int s = 0; s < 3; ++s) { p[s][i] = bf16_rn(x); x -= __uint_as_float((unsigned)p[s][i] << 16); }
  }
  for (int s = 0; s < 3; ++s) {
    uint4 raw = {(unsigned)p[s][0] | ((unsigned)p[s][1] << 16), (unsigned)p[s][2] | ((unsigned)p[s][3] << 16),
                 (unsigned)p[s][4] | ((unsigned)p[s][5] << 16), (unsigned)p[s][6] | ((unsigned)p[s][7] << 16)};
    *reinterpret_cast<uint4*>(Mq + (((long)ks * 3 + s) * 64 + lane) * 8) = raw;
  }
}

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {   // v_cvt_pk_bf16_f32 (RN-even)
  const f32x2 v = {a, b};
  const bf16x2 r = __builtin_convertvector(v, bf16x2);
  return __builtin_bit_cast(unsigned, r);
}

// ---- MFMA forward: block = 4 waves, each wave TL tiles of 16 cells, the block's gene slice staged in LDS
template <int TL, int NE, int ABL = 0>
__global__ void __launch_bounds__(256) fwd_mfma(const float* __restrict__ F, const float* __restrict__ em2, const float* __restrict__ Vs,
                                                const unsigned short* __restrict__ Mq, float* __restrict__ Zp, long N, int G,
                                                int kchunk /*k-steps per slice*/, int nk) {
  extern __shared__ uint4 ldsq[];   // [kchunk][3][64] uint4 (B parts), then [kchunk][32] float V'
  const int k0 = blockIdx.y * kchunk;
  const int nks = min(nk, k0 + kchunk) - k0;
  {
    const uint4* src = reinterpret_cast<const uint4*>(Mq) + (long)k0 * 192;
    for (int i = threadIdx.x; i < nks * 192; i += 256) ldsq[i] = src[i];
    float* lv = reinterpret_cast<float*>(ldsq + (long)kchunk * 192);
    for (int i = threadIdx.x; i < nks * 32; i += 256) { const int g = k0 * 32 + i; lv[i] = g < G ? Vs[g] : 0.f; }
  }
  __syncthreads();
  const float4* lv4 = reinterpret_cast<const float4*>(ldsq + (long)kchunk * 192);
  const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4, wv = threadIdx.x >> 6;
  const long cell0 = ((long)blockIdx.x * 4 + wv) * (TL * 16);
  float f[TL], em[TL];
  f32x4 acc[TL];
#pragma unroll
  for (int t = 0; t < TL; ++t) {
    const long n = cell0 + 16 * t + j; const long nn = n < N ? n : N - 1;
    f[t] = F[nn]; em[t] = em2[nn];
    acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  for (int ks = 0; ks < nks; ++ks) {
    const uint4 b1r = ldsq[(ks * 3 + 0) * 64 + lane], b2r = ldsq[(ks * 3 + 1) * 64 + lane], b3r = ldsq[(ks * 3 + 2) * 64 + lane];
    const bf16x8 B1 = __builtin_bit_cast(bf16x8, b1r), B2 = __builtin_bit_cast(bf16x8, b2r), B3 = __builtin_bit_cast(bf16x8, b3r);
    const float4 va = lv4[ks * 8 + 2 * q], vb = lv4[ks * 8 + 2 * q + 1];
    const f32x2 v2[4] = {{va.x, va.y}, {va.z, va.w}, {vb.x, vb.y}, {vb.z, vb.w}};
#pragma unroll
    for (int t = 0; t < TL; ++t) {
      unsigned hi[4], mid[4], lo[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const f32x2 eta = v2[p] * f[t] - em[t];
        const float e0 = ABL == 1 ? eta.x : __builtin_amdgcn_exp2f(eta.x), e1 = ABL == 1 ? eta.y : __builtin_amdgcn_exp2f(eta.y);
        hi[p] = pk_bf16(e0, e1);
        f32x2 r;
        if (ABL == 4) {   // v_dot2_f32_bf16: e - hi straight from the packed pair
          const bf16x2 hb = __builtin_bit_cast(bf16x2, hi[p]);
          const bf16x2 m0 = __builtin_bit_cast(bf16x2, 0x0000BF80u), m1 = __builtin_bit_cast(bf16x2, 0xBF800000u);
          r.x = __builtin_amdgcn_fdot2_f32_bf16(hb, m0, e0, false);
          r.y = __builtin_amdgcn_fdot2_f32_bf16(hb, m1, e1, false);
        } else {
          r = (f32x2){e0, e1} - (f32x2){__uint_as_float(hi[p] << 16), __uint_as_float(hi[p] & 0xffff0000u)};
        }
        mid[p] = ABL == 3 ? hi[p] : pk_bf16(r.x, r.y);
        if (NE == 3) {
          r = r - (f32x2){__uint_as_float(mid[p] << 16), __uint_as_float(mid[p] & 0xffff0000u)};
          lo[p] = pk_bf16(r.x, r.y);
        }
      }
      const bf16x8 A1 = __builtin_bit_cast(bf16x8, ((uint4){hi[0], hi[1], hi[2], hi[3]}));
      const bf16x8 A2 = __builtin_bit_cast(bf16x8, ((uint4){mid[0], mid[1], mid[2], mid[3]}));
      f32x4 a = acc[t];
      if (NE == 3) {
        const bf16x8 A3 = __builtin_bit_cast(bf16x8, ((uint4){lo[0], lo[1], lo[2], lo[3]}));
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A3, B1, a, 0, 0, 0);
      }
      if (ABL != 2) {
      a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A2, B2, a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1, B3, a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1, B2, a, 0, 0, 0);
      }
      a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A2, B1, a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1, B1, a, 0, 0, 0);
      acc[t] = a;
    }
  }
  // D layout: lane (column j, rows 4q..4q+3)
#pragma unroll
  for (int t = 0; t < TL; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long n = cell0 + 16 * t + 4 * q + r;
      if (n < N) Zp[((long)blockIdx.y * N + n) * NC + j] = acc[t][r];
    }
}

template <int NM, int IL, int LEFT>
__device__ __forceinline__ void sgb_tiles() {
  if constexpr (LEFT > 0) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x402, (LEFT % NM == 1 || NM == 1) ? 32 - (NM - 1) * IL : IL, 0);
    sgb_tiles<NM, IL, LEFT - 1>();
  }
}
// ---- pipelined variant: the MFMAs of tile t-1 are interleaved with the VALU work of tile t
template <int TL, int NM /*MFMAs per tile: 5 or 3*/, int IL /*VALU per MFMA in the interleave*/>
__global__ void __launch_bounds__(256) fwd_mfma_p(const float* __restrict__ F, const float* __restrict__ em2, const float* __restrict__ Vs,
                                                  const unsigned short* __restrict__ Mq, float* __restrict__ Zp, long N, int G,
                                                  int kchunk, int nk) {
  extern __shared__ uint4 ldsq[];
  const int k0 = blockIdx.y * kchunk;
  const int nks = min(nk, k0 + kchunk) - k0;
  {
    const uint4* src = reinterpret_cast<const uint4*>(Mq) + (long)k0 * 192;
    for (int i = threadIdx.x; i < nks * 192; i += 256) ldsq[i] = src[i];
    float* lv = reinterpret_cast<float*>(ldsq + (long)kchunk * 192);
    for (int i = threadIdx.x; i < nks * 32; i += 256) { const int g = k0 * 32 + i; lv[i] = g < G ? Vs[g] : 0.f; }
  }
  __syncthreads();
  const float4* lv4 = reinterpret_cast<const float4*>(ldsq + (long)kchunk * 192);
  const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4, wv = threadIdx.x >> 6;
  const long cell0 = ((long)blockIdx.x * 4 + wv) * (TL * 16);
  float f[TL], em[TL];
  f32x4 acc[TL];
#pragma unroll
  for (int t = 0; t < TL; ++t) {
    const long n = cell0 + 16 * t + j; const long nn = n < N ? n : N - 1;
    f[t] = F[nn]; em[t] = em2[nn];
    acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  for (int ks = 0; ks < nks; ++ks) {
    const uint4 b1r = ldsq[(ks * 3 + 0) * 64 + lane], b2r = ldsq[(ks * 3 + 1) * 64 + lane], b3r = ldsq[(ks * 3 + 2) * 64 + lane];
    const bf16x8 B1 = __builtin_bit_cast(bf16x8, b1r), B2 = __builtin_bit_cast(bf16x8, b2r), B3 = __builtin_bit_cast(bf16x8, b3r);
    const float4 va = lv4[ks * 8 + 2 * q], vb = lv4[ks * 8 + 2 * q + 1];
    const f32x2 v2[4] = {{va.x, va.y}, {va.z, va.w}, {vb.x, vb.y}, {vb.z, vb.w}};
    bf16x8 A1[TL], A2[TL];
#pragma unroll
    for (int t = 0; t <= TL; ++t) {
      if (t < TL) {
        unsigned hi[4], lo[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const f32x2 eta = v2[p] * f[t] - em[t];
          const float e0 = __builtin_amdgcn_exp2f(eta.x), e1 = __builtin_amdgcn_exp2f(eta.y);
          hi[p] = pk_bf16(e0, e1);
          const f32x2 r = (f32x2){e0, e1} - (f32x2){__uint_as_float(hi[p] << 16), __uint_as_float(hi[p] & 0xffff0000u)};
          lo[p] = pk_bf16(r.x, r.y);
        }
        A1[t] = __builtin_bit_cast(bf16x8, ((uint4){hi[0], hi[1], hi[2], hi[3]}));
        A2[t] = __builtin_bit_cast(bf16x8, ((uint4){lo[0], lo[1], lo[2], lo[3]}));
      }
      if (t > 0) {
        f32x4 a = acc[t - 1];
        if (NM == 5) {
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A2[t - 1], B2, a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1[t - 1], B3, a, 0, 0, 0);
        }
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A2[t - 1], B1, a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1[t - 1], B2, a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1[t - 1], B1, a, 0, 0, 0);
        acc[t - 1] = a;
      }
    }
    if (IL > 0) {
      __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);     // the k-step's LDS reads
      __builtin_amdgcn_sched_group_barrier(0x402, 32, 0);    // tile 0
      sgb_tiles<NM, IL, (TL - 1) * NM>();
      __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
    }
  }
#pragma unroll
  for (int t = 0; t < TL; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long n = cell0 + 16 * t + 4 * q + r;
      if (n < N) Zp[((long)blockIdx.y * N + n) * NC + j] = acc[t][r];
    }
}

// ---- streaming variant: B (2 bf16 parts) and V' double-buffered through LDS in chunks of KC k-steps, 3 MFMAs per tile,
//      lo = e - hi by v_dot2c_f32_bf16 (DOT = 1) or unpack + subtract (DOT = 0)
template <int TL, int KC, int DOT, int ABL = 0, int STG = 0>
__global__ void __launch_bounds__(256) fwd_mfma_s(const float* __restrict__ F, const float* __restrict__ em2, const float* __restrict__ Vs,
                                                  const unsigned short* __restrict__ Mq /*[nk][3][64][8], parts 0,1 used*/,
                                                  float* __restrict__ Zp, long N, int G, int kchunk, int nk) {
  constexpr int BUF = KC * (128 + 8);            // uint4 per buffer: B 2 x 64 per k-step, V' 8 per k-step
  __shared__ uint4 lds[2 * BUF];
  const int k0 = blockIdx.y * kchunk;
  const int nks = min(nk, k0 + kchunk) - k0;
  const int nch = (nks + KC - 1) / KC;
  const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4, wv = threadIdx.x >> 6;
  const long cell0 = ((long)blockIdx.x * 4 + wv) * (TL * 16);
  constexpr int NLD = (KC * 128 + 255) / 256;    // uint4 loads per thread per chunk for B
  uint4 st[NLD]; float sv = 0.f;
  auto gload = [&](int c) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = threadIdx.x + 256 * i;     // [ks][part][lane]
      const int ks = idx >> 7, rem = idx & 127;
      const int kk = k0 + c * KC + ks;
      st[i] = (idx < KC * 128 && kk < k0 + nks) ? reinterpret_cast<const uint4*>(Mq)[((long)kk * 3) * 64 + rem] : (uint4){0, 0, 0, 0};
    }
    if (threadIdx.x < KC * 32) { const int g = (k0 + c * KC) * 32 + threadIdx.x; sv = (g < G && g < (k0 + nks) * 32) ? Vs[g] : 0.f; }
  };
  auto lstore = [&](int b) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) { const int idx = threadIdx.x + 256 * i; if (idx < KC * 128) lds[b * BUF + idx] = st[i]; }
    if (threadIdx.x < KC * 32) reinterpret_cast<float*>(lds + b * BUF + KC * 128)[threadIdx.x] = sv;
  };
  float f[TL], em[TL];
  f32x4 acc[TL];
#pragma unroll
  for (int t = 0; t < TL; ++t) {
    const long n = cell0 + 16 * t + j; const long nn = n < N ? n : N - 1;
    f[t] = F[nn]; em[t] = em2[nn];
    acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  unsigned m0, m1;
  asm volatile("s_mov_b32 %0, 0x0000bf80" : "=s"(m0));
  asm volatile("s_mov_b32 %0, 0xbf800000" : "=s"(m1));
  gload(0); lstore(0);
  __syncthreads();
  for (int c = 0; c < nch; ++c) {
    const int b = c & 1;
    if (c + 1 < nch) gload(c + 1);
    const uint4* lb = lds + b * BUF;
    const float4* lv4 = reinterpret_cast<const float4*>(lds + b * BUF + KC * 128);
#pragma unroll 2
    for (int ks = 0; ks < KC; ++ks) {
      const uint4 b1r = lb[ks * 128 + lane], b2r = lb[ks * 128 + 64 + lane];
      const bf16x8 B1 = __builtin_bit_cast(bf16x8, b1r), B2 = __builtin_bit_cast(bf16x8, b2r);
      const float4 va = lv4[ks * 8 + 2 * q], vb = lv4[ks * 8 + 2 * q + 1];
      const f32x2 v2[4] = {{va.x, va.y}, {va.z, va.w}, {vb.x, vb.y}, {vb.z, vb.w}};
#pragma unroll
      for (int t = 0; t < TL; ++t) {
        unsigned hi[4], lo[4];
        if (STG) {   // stage-wise over the 4 gene pairs: independent instructions back to back
          f32x2 eta[4], e[4], r[4];
#pragma unroll
          for (int p = 0; p < 4; ++p) eta[p] = v2[p] * f[t] - em[t];
#pragma unroll
          for (int p = 0; p < 4; ++p) e[p] = (f32x2){__builtin_amdgcn_exp2f(eta[p].x), __builtin_amdgcn_exp2f(eta[p].y)};
#pragma unroll
          for (int p = 0; p < 4; ++p) hi[p] = pk_bf16(e[p].x, e[p].y);
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            const bf16x2 hb = __builtin_bit_cast(bf16x2, hi[p]);
            r[p].x = __builtin_amdgcn_fdot2_f32_bf16(hb, __builtin_bit_cast(bf16x2, m0), e[p].x, false);
            r[p].y = __builtin_amdgcn_fdot2_f32_bf16(hb, __builtin_bit_cast(bf16x2, m1), e[p].y, false);
          }
#pragma unroll
          for (int p = 0; p < 4; ++p) lo[p] = pk_bf16(r[p].x, r[p].y);
          if (STG == 2) {
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x400, 8, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
          }
        } else {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          if (ABL == 2) { hi[p] = __float_as_uint(f[t]) + p; lo[p] = __float_as_uint(em[t]) + p; continue; }   // MFMA work only
          if (DOT >= 2) {   // scalar (VOP2/VOP3 single) math only
            const float x0 = fmaf(v2[p].x, f[t], -em[t]), x1 = fmaf(v2[p].y, f[t], -em[t]);
            const float e0 = __builtin_amdgcn_exp2f(x0), e1 = __builtin_amdgcn_exp2f(x1);
            hi[p] = pk_bf16(e0, e1);
            float r0, r1;
            if (DOT == 3) {
              const bf16x2 hb = __builtin_bit_cast(bf16x2, hi[p]);
              r0 = __builtin_amdgcn_fdot2_f32_bf16(hb, __builtin_bit_cast(bf16x2, m0), e0, false);
              r1 = __builtin_amdgcn_fdot2_f32_bf16(hb, __builtin_bit_cast(bf16x2, m1), e1, false);
            } else {
              r0 = e0 - __uint_as_float(hi[p] << 16);
              r1 = e1 - __uint_as_float(hi[p] & 0xffff0000u);
            }
            lo[p] = pk_bf16(r0, r1);
            continue;
          }
          const f32x2 eta = v2[p] * f[t] - em[t];
          const float e0 = __builtin_amdgcn_exp2f(eta.x), e1 = __builtin_amdgcn_exp2f(eta.y);
          hi[p] = pk_bf16(e0, e1);
          f32x2 r;
          if (DOT) {
            const bf16x2 hb = __builtin_bit_cast(bf16x2, hi[p]);
            r.x = __builtin_amdgcn_fdot2_f32_bf16(hb, __builtin_bit_cast(bf16x2, m0), e0, false);
            r.y = __builtin_amdgcn_fdot2_f32_bf16(hb, __builtin_bit_cast(bf16x2, m1), e1, false);
          } else {
            r = (f32x2){e0, e1} - (f32x2){__uint_as_float(hi[p] << 16), __uint_as_float(hi[p] & 0xffff0000u)};
          }
          lo[p] = pk_bf16(r.x, r.y);
        }
        }
        const bf16x8 A1 = __builtin_bit_cast(bf16x8, ((uint4){hi[0], hi[1], hi[2], hi[3]}));
        const bf16x8 A2 = __builtin_bit_cast(bf16x8, ((uint4){lo[0], lo[1], lo[2], lo[3]}));
        f32x4 a = acc[t];
        if (ABL == 1) {   // VALU work only: fold the operands into the accumulator with 2 cheap ops
          a[0] = __uint_as_float(__float_as_uint(a[0]) ^ hi[0] ^ hi[1] ^ hi[2] ^ hi[3]);
          a[1] = __uint_as_float(__float_as_uint(a[1]) ^ lo[0] ^ lo[1] ^ lo[2] ^ lo[3]);
        } else {
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A2, B1, a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1, B2, a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1, B1, a, 0, 0, 0);
        }
        acc[t] = a;
      }
    }
    if (c + 1 < nch) lstore(b ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int t = 0; t < TL; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long n = cell0 + 16 * t + 4 * q + r;
      if (n < N) Zp[((long)blockIdx.y * N + n) * NC + j] = acc[t][r];
    }
}

// ---- in-block gene split: block = 16*TL cells, its 4 waves take every 4th k-step, B and V' straight from global (L2) with
//      one k-step of prefetch, partial accumulators combined through LDS: complete Z per block, no partial slabs, no LDS staging
template <int TL>
__global__ void __launch_bounds__(256) fwd_mfma_g(const float* __restrict__ F, const float* __restrict__ em2, const float* __restrict__ Vs,
                                                  const unsigned short* __restrict__ Mq /*[nk][3][64][8]*/, float* __restrict__ Zp, long N, int G,
                                                  int nk) {
  __shared__ f32x4 comb[4][TL][64];
  const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4, wv = threadIdx.x >> 6;
  const long cell0 = (long)blockIdx.x * (TL * 16);
  float f[TL], em[TL];
  f32x4 acc[TL];
#pragma unroll
  for (int t = 0; t < TL; ++t) {
    const long n = cell0 + 16 * t + j; const long nn = n < N ? n : N - 1;
    f[t] = F[nn]; em[t] = em2[nn];
    acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  unsigned m0, m1;
  asm volatile("s_mov_b32 %0, 0x0000bf80" : "=s"(m0));
  asm volatile("s_mov_b32 %0, 0xbf800000" : "=s"(m1));
  const uint4* Bq = reinterpret_cast<const uint4*>(Mq);
  auto ldv = [&](int ks, float4& a, float4& b) {
    const int g = ks * 32 + 8 * q;
    const int g0 = min(g, G - 8 > 0 ? G - 8 : 0);   // lab: G multiple of 8
    a = *reinterpret_cast<const float4*>(Vs + g0); b = *reinterpret_cast<const float4*>(Vs + g0 + 4);
  };
  int ks = wv;
  uint4 b1n = {0,0,0,0}, b2n = {0,0,0,0}; float4 van = {0,0,0,0}, vbn = {0,0,0,0};
  if (ks < nk) { b1n = Bq[((long)ks * 3 + 0) * 64 + lane]; b2n = Bq[((long)ks * 3 + 1) * 64 + lane]; ldv(ks, van, vbn); }
  for (; ks < nk; ks += 4) {
    const uint4 b1r = b1n, b2r = b2n; const float4 va = van, vb = vbn;
    if (ks + 4 < nk) { b1n = Bq[((long)(ks + 4) * 3 + 0) * 64 + lane]; b2n = Bq[((long)(ks + 4) * 3 + 1) * 64 + lane]; ldv(ks + 4, van, vbn); }
    const bf16x8 B1 = __builtin_bit_cast(bf16x8, b1r), B2 = __builtin_bit_cast(bf16x8, b2r);
    const f32x2 v2[4] = {{va.x, va.y}, {va.z, va.w}, {vb.x, vb.y}, {vb.z, vb.w}};
#pragma unroll
    for (int t = 0; t < TL; ++t) {
      unsigned hi[4], lo[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const f32x2 eta = v2[p] * f[t] - em[t];
        const float e0 = __builtin_amdgcn_exp2f(eta.x), e1 = __builtin_amdgcn_exp2f(eta.y);
        hi[p] = pk_bf16(e0, e1);
        const bf16x2 hb = __builtin_bit_cast(bf16x2, hi[p]);
        const float r0 = __builtin_amdgcn_fdot2_f32_bf16(hb, __builtin_bit_cast(bf16x2, m0), e0, false);
        const float r1 = __builtin_amdgcn_fdot2_f32_bf16(hb, __builtin_bit_cast(bf16x2, m1), e1, false);
        lo[p] = pk_bf16(r0, r1);
      }
      const bf16x8 A1 = __builtin_bit_cast(bf16x8, ((uint4){hi[0], hi[1], hi[2], hi[3]}));
      const bf16x8 A2 = __builtin_bit_cast(bf16x8, ((uint4){lo[0], lo[1], lo[2], lo[3]}));
      f32x4 a = acc[t];
      a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A2, B1, a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1, B2, a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1, B1, a, 0, 0, 0);
      acc[t] = a;
    }
  }
#pragma unroll
  for (int t = 0; t < TL; ++t) comb[wv][t][lane] = acc[t];
  __syncthreads();
  // thread -> (tile t, lane l): sum the 4 waves, write Z rows (cells 4q+r of the tile, column j)
  for (int i = threadIdx.x; i < TL * 64; i += 256) {
    const int t = i >> 6, l = i & 63, jj = l & 15, qq = l >> 4;
    const f32x4 z = (comb[0][t][l] + comb[1][t][l]) + (comb[2][t][l] + comb[3][t][l]);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long n = cell0 + 16 * t + 4 * qq + r;
      if (n < N) Zp[n * NC + jj] = z[r];
    }
  }
}

int main(int argc, char** argv) {
  long N = argc > 1 ? atol(argv[1]) : 100000; int G = argc > 2 ? atoi(argv[2]) : 5000;
  std::vector<float> F(N), em(N), Vs(G), M((size_t)G * NC);
  srand(1);
  auto rnd = []() { return (float)rand() / RAND_MAX; };
  float vmin = 1e9, vmax = -1e9;
  for (auto& v : Vs) { v = (rnd() - 0.5f) * 1.2f; vmin = std::min(vmin, v); vmax = std::max(vmax, v); }
  for (long i = 0; i < N; ++i) { F[i] = (rnd() - 0.5f) * 4.f; em[i] = std::max(F[i] * vmin, F[i] * vmax); }
  for (auto& v : M) v = (rnd() * 3.f + 0.01f) * expf((rnd() - 0.5f) * 6.f);
  const int nk = (G + 31) / 32;
  float *dF, *dem, *dVs, *dM, *dZ; unsigned short* dMq;
  const int maxsplit = 64;
  CK(hipMalloc(&dF, N * 4)); CK(hipMalloc(&dem, N * 4)); CK(hipMalloc(&dVs, G * 4)); CK(hipMalloc(&dM, (size_t)G * NC * 4));
  CK(hipMalloc(&dMq, (size_t)nk * 192 * 16));
  CK(hipMalloc(&dZ, (size_t)maxsplit * N * NC * 4));
  CK(hipMemcpy(dF, F.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dem, em.data(), N * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dVs, Vs.data(), G * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dM, M.data(), (size_t)G * NC * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(split_m, dim3((nk * 64 + 255) / 256), dim3(256), 0, 0, dM, dMq, G, nk);
  CK(hipDeviceSynchronize());
  // float64 host reference on a cell sample
  const int NS = 512;
  std::vector<double> ref((size_t)NS * NC, 0.0);
  std::vector<long> cells(NS);
  for (int i = 0; i < NS; ++i) cells[i] = (long)((double)i / NS * N);
  for (int i = 0; i < NS; ++i) {
    const long n = cells[i];
    for (int g = 0; g < G; ++g) {
      const double e = exp2((double)F[n] * (double)Vs[g] - (double)em[n]);
      for (int c = 0; c < NC; ++c) ref[(size_t)i * NC + c] += e * (double)M[(size_t)g * NC + c];
    }
  }
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  std::vector<float> cur;
  auto check = [&](int gsplit, const char* name, float ms) {
    cur.assign((size_t)gsplit * N * NC, 0.f);
    CK(hipMemcpy(cur.data(), dZ, cur.size() * 4, hipMemcpyDeviceToHost));
    double err = 0, rms = 0;
    for (int i = 0; i < NS; ++i) for (int c = 0; c < NC; ++c) {
      double z = 0; for (int s = 0; s < gsplit; ++s) z += cur[((size_t)s * N + cells[i]) * NC + c];
      const double r = std::fabs(z - ref[(size_t)i * NC + c]) / std::fabs(ref[(size_t)i * NC + c]);
      err = std::max(err, r); rms += r * r;
    }
    const double flops = (double)N * G * (2.0 * NC + 2 + 1);
    printf("%-22s gsplit %3d  %8.1f us  %6.1f TFLOP/s  maxrel %.2e rms %.2e\n", name, gsplit, ms * 1e3, flops / ms / 1e9, err,
           std::sqrt(rms / (NS * NC)));
  };
#define TIME(name, gs, LAUNCH)                                                                  \
  { float best = 1e9;                                                                           \
    for (int it = 0; it < 6; ++it) {                                                            \
      CK(hipEventRecord(a)); LAUNCH; CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipGetLastError()); \
      float ms; CK(hipEventElapsedTime(&ms, a, b)); if (it) best = std::min(best, ms); }        \
    check(gs, name, best); }
  for (int rep = 0; rep < 2; ++rep) {
    for (int gsplit : {8, 16}) {
      const int gchunk = (G + gsplit - 1) / gsplit, gs = (G + gchunk - 1) / gchunk;
      TIME("valu lds R=2", gs, hipLaunchKernelGGL((fwd_valu<2>), dim3((unsigned)((N + 511) / 512), gs), dim3(256), (size_t)gchunk * 68, 0,
                                                   dF, dem, dVs, dM, dZ, N, G, gchunk));
    }
    {
#define MFG(TLV) { char nm[64]; snprintf(nm, 64, "block-split TL=%d", TLV);                              \
      TIME(nm, 1, hipLaunchKernelGGL((fwd_mfma_g<TLV>), dim3((unsigned)((N + 16 * TLV - 1) / (16 * TLV))), dim3(256), 0, 0, \
                                      dF, dem, dVs, dMq, dZ, N, G, nk)); }
      MFG(1); MFG(2); MFG(4); MFG(8);
    }
    for (int gsplit : {5, 8}) {
      const int kchunk = (nk + gsplit - 1) / gsplit, gs = (nk + kchunk - 1) / kchunk;
#define MFS(TLV, KCV, DV) { char nm[64]; snprintf(nm, 64, "stream TL=%d KC=%d dot%d", TLV, KCV, DV);                              \
      TIME(nm, gs, hipLaunchKernelGGL((fwd_mfma_s<TLV, KCV, DV>), dim3((unsigned)((N + 64 * TLV - 1) / (64 * TLV)), gs), dim3(256), 0, 0, \
                                      dF, dem, dVs, dMq, dZ, N, G, kchunk, nk)); }
      MFS(4, 4, 1); MFS(4, 4, 2); MFS(4, 4, 3);
#define MFSA(TLV, KCV, DV, AB) { char nm[64]; snprintf(nm, 64, "stream TL=%d KC=%d abl%d", TLV, KCV, AB);                              \
      TIME(nm, gs, hipLaunchKernelGGL((fwd_mfma_s<TLV, KCV, DV, AB>), dim3((unsigned)((N + 64 * TLV - 1) / (64 * TLV)), gs), dim3(256), 0, 0, \
                                      dF, dem, dVs, dMq, dZ, N, G, kchunk, nk)); }
#define MFSG(TLV, KCV, SG) { char nm[64]; snprintf(nm, 64, "stream TL=%d KC=%d stg%d", TLV, KCV, SG);                              \
      TIME(nm, gs, hipLaunchKernelGGL((fwd_mfma_s<TLV, KCV, 1, 0, SG>), dim3((unsigned)((N + 64 * TLV - 1) / (64 * TLV)), gs), dim3(256), 0, 0, \
                                      dF, dem, dVs, dMq, dZ, N, G, kchunk, nk)); }

    }
    for (int gsplit : {16}) {
      const int kchunk = (nk + gsplit - 1) / gsplit, gs = (nk + kchunk - 1) / kchunk;
      const size_t lds = (size_t)kchunk * (192 * 16 + 128);
#define MF(TLV, NEV) { char nm[64]; snprintf(nm, 64, "mfma TL=%d NE=%d", TLV, NEV);                                             \
      TIME(nm, gs, hipLaunchKernelGGL((fwd_mfma<TLV, NEV>), dim3((unsigned)((N + 64 * TLV - 1) / (64 * TLV)), gs), dim3(256), lds, 0, \
                                      dF, dem, dVs, dMq, dZ, N, G, kchunk, nk)); }
      if (lds > 64 * 1024) continue;
      MF(4, 2);
#define MFA(TLV, NEV, AB) { char nm[64]; snprintf(nm, 64, "mfma TL=%d NE=%d abl%d", TLV, NEV, AB);                              \
      TIME(nm, gs, hipLaunchKernelGGL((fwd_mfma<TLV, NEV, AB>), dim3((unsigned)((N + 64 * TLV - 1) / (64 * TLV)), gs), dim3(256), lds, 0, \
                                      dF, dem, dVs, dMq, dZ, N, G, kchunk, nk)); }
#define MFP(TLV, NMV, ILV) { char nm[64]; snprintf(nm, 64, "pipe TL=%d NM=%d IL=%d", TLV, NMV, ILV);                              \
      TIME(nm, gs, hipLaunchKernelGGL((fwd_mfma_p<TLV, NMV, ILV>), dim3((unsigned)((N + 64 * TLV - 1) / (64 * TLV)), gs), dim3(256), lds, 0, \
                                      dF, dem, dVs, dMq, dZ, N, G, kchunk, nk)); }
      MFP(4, 3, 0);
    }
  }
  return 0;
}

// Ablation lab for the backward sweep (NC=8, D=1): where do the cycles go? Not product code.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
constexpr int NC = 8;
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_pull(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum63(float v) {
  v += dpp_pull<0xB1, 0xF>(v); v += dpp_pull<0x4E, 0xF>(v); v += dpp_pull<0x141, 0xF>(v); v += dpp_pull<0x140, 0xF>(v);
  v += dpp_pull<0x142, 0xA>(v); v += dpp_pull<0x143, 0xC>(v);
  return v;
}
// MODE bits: 1 = skip exp, 2 = skip dF reduction/store, 4 = skip t contraction, 8 = skip gene accumulators
template <int RG, int MODE>
__global__ void __launch_bounds__(256) bwd(const float* __restrict__ coef, const float* __restrict__ F, const float* __restrict__ em2,
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
      const float eta = fmaf(f, vs[r], -em);
      const float e = (MODE & 1) ? eta : __builtin_amdgcn_exp2f(eta);
      float t = 0.f;
      if (MODE & 4) t = cf[0] * l[r][0];
      else {
#pragma unroll
        for (int c = 0; c < NC; ++c) t = fmaf(cf[c], l[r][c], t);
      }
      const float u = e * t;
      if (!(MODE & 8)) { accU[r] += u; accUF[r] = fmaf(u, f, accUF[r]); }
      dsum = fmaf(m_[r] * u, v[r], dsum);
    }
    if (MODE & 2) keep += dsum;
    else {
      const int slot = (int)(n - n0) & 63;
      const float tot = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wave_sum63(dsum)), 63));
      keep = (lane == slot) ? tot : keep;
      if (slot == 63 || n + 1 == n1) { const long fb = n - slot; if (fb + lane <= n) dFpart[(long)tile * N + fb + lane] = keep; }
    }
  }
  if (MODE & 2) dFpart[(long)tile * N + n0 + lane] = keep;
#pragma unroll
  for (int r = 0; r < RG; ++r) {
    const int g = gbase + r * 64 + lane;
    if (g < G) { float* gp = gpart + ((long)blockIdx.y * G + g) * 2; gp[0] = accU[r]; gp[1] = m_[r] * accUF[r]; }
  }
}

// scalar loads with one-cell-ahead software prefetch
template <int RG>
__global__ void __launch_bounds__(256) bwd_pf(const float* __restrict__ coef, const float* __restrict__ F, const float* __restrict__ em2,
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
  float cfn[NC], fn, emn;
#pragma unroll
  for (int c = 0; c < NC; ++c) cfn[c] = coef[n0 * 8 + c];
  fn = F[n0]; emn = em2[n0];
  for (long n = n0; n < n1; ++n) {
    float cf[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) cf[c] = cfn[c];
    const float f = fn, em = emn;
    const long nx = (n + 1 < n1) ? n + 1 : n;
#pragma unroll
    for (int c = 0; c < NC; ++c) cfn[c] = coef[nx * 8 + c];
    fn = F[nx]; emn = em2[nx];
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
// LDS-staged cell slice, CU cells per iteration with all LDS reads issued first
template <int RG, int CU>
__global__ void __launch_bounds__(256) bwd_lds(const float* __restrict__ coef, const float* __restrict__ F, const float* __restrict__ em2,
                                               const float* __restrict__ Lb, const float* __restrict__ mu, const float* __restrict__ Vs,
                                               const float* __restrict__ V, float* __restrict__ gpart, float* __restrict__ dFpart,
                                               long N, int G, long cchunk) {
  extern __shared__ float lds[];
  const long n0 = (long)blockIdx.y * cchunk, n1 = std::min(n0 + cchunk, N);
  const int nc = (int)(n1 - n0);
  for (int i = threadIdx.x; i < nc * 2; i += 256) reinterpret_cast<float4*>(lds)[(i >> 1) * 3 + (i & 1)] = reinterpret_cast<const float4*>(coef + n0 * 8)[i];
  for (int i = threadIdx.x; i < nc; i += 256) { lds[i * 12 + 8] = F[n0 + i]; lds[i * 12 + 9] = em2[n0 + i]; }
  __syncthreads();
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
  float keep = 0.f;
  const int ncu = nc / CU * CU;
  for (int i = 0; i < ncu; i += CU) {
    float4 c0[CU], c1[CU], c2[CU];
#pragma unroll
    for (int j = 0; j < CU; ++j) { const float4* p = reinterpret_cast<const float4*>(lds) + (i + j) * 3; c0[j] = p[0]; c1[j] = p[1]; c2[j] = p[2]; }
#pragma unroll
    for (int j = 0; j < CU; ++j) {
      const float cf[NC] = {c0[j].x, c0[j].y, c0[j].z, c0[j].w, c1[j].x, c1[j].y, c1[j].z, c1[j].w};
      const float f = c2[j].x, em = c2[j].y;
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
      const int slot = (i + j) & 63;
      const float tot = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wave_sum63(dsum)), 63));
      keep = (lane == slot) ? tot : keep;
    }
    const int last = i + CU - 1;
    if ((last & 63) == 63 || last + 1 == ncu) { const int fb = last & ~63; if (fb + lane <= last) dFpart[(long)tile * N + n0 + fb + lane] = keep; }
  }
#pragma unroll
  for (int r = 0; r < RG; ++r) {
    const int g = gbase + r * 64 + lane;
    if (g < G) { float* gp = gpart + ((long)blockIdx.y * G + g) * 2; gp[0] = accU[r]; gp[1] = m_[r] * accUF[r]; }
  }
}
int main() {
  long N = 100000; int G = 5000;
  std::vector<float> F(N), em(N), Vs(G), V(G), mu(G), Lb((size_t)G * 8), coef((size_t)N * 8);
  srand(1); auto rnd = []() { return (float)rand() / RAND_MAX; };
  float vmin = 1e9, vmax = -1e9;
  for (int g = 0; g < G; ++g) { V[g] = (rnd() - 0.5f) * 0.8f; Vs[g] = V[g] * 1.442695f; vmin = std::min(vmin, Vs[g]); vmax = std::max(vmax, Vs[g]); mu[g] = rnd() + 0.1f; }
  for (long i = 0; i < N; ++i) { F[i] = (rnd() - 0.5f) * 4.f; em[i] = std::max(F[i] * vmin, F[i] * vmax); }
  for (auto& x : Lb) x = 1.f + (int)(rnd() * 3.99f);
  for (auto& x : coef) x = -rnd() * 1e-3f;
  float *dc, *dF, *dem, *dL, *dmu, *dVs, *dV, *dg, *ddF;
  CK(hipMalloc(&dc, N * 32)); CK(hipMalloc(&dF, N * 4)); CK(hipMalloc(&dem, N * 4)); CK(hipMalloc(&dL, (size_t)G * 32)); CK(hipMalloc(&dmu, G * 4));
  CK(hipMalloc(&dVs, G * 4)); CK(hipMalloc(&dV, G * 4)); CK(hipMalloc(&dg, (size_t)2048 * G * 8)); CK(hipMalloc(&ddF, (size_t)128 * N * 4));
  CK(hipMemcpy(dc, coef.data(), N * 32, hipMemcpyHostToDevice)); CK(hipMemcpy(dF, F.data(), N * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dem, em.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dL, Lb.data(), (size_t)G * 32, hipMemcpyHostToDevice));
  CK(hipMemcpy(dmu, mu.data(), G * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dVs, Vs.data(), G * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dV, V.data(), G * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
#define RUN(name, RG, csplit_req, KERNEL)                                                                        \
  { long cchunk = (N + (csplit_req) - 1) / (csplit_req); int cs = (int)((N + cchunk - 1) / cchunk);              \
    const int ntile = (G + 64 * RG - 1) / (64 * RG); dim3 grid((ntile + 3) / 4, cs); float best = 1e9;           \
    for (int it = 0; it < 5; ++it) { CK(hipEventRecord(e0));                                                     \
      hipLaunchKernelGGL(KERNEL, grid, dim3(256), 0, 0, dc, dF, dem, dL, dmu, dVs, dV, dg, ddF, N, G, cchunk);   \
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());                                \
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (it) best = std::min(best, ms); }                       \
    printf("%-34s RG %d csplit %4d %8.1f us\n", name, RG, cs, best * 1e3); }
  RUN("full", 4, 410, (bwd<4, 0>));
  RUN("full", 4, 358, (bwd<4, 0>));
  RUN("no exp", 4, 410, (bwd<4, 1>));
  RUN("no dF reduce", 4, 410, (bwd<4, 2>));
  RUN("no t contraction", 4, 410, (bwd<4, 4>));
  RUN("no gene accumulators", 4, 410, (bwd<4, 8>));
  RUN("no exp, no reduce", 4, 410, (bwd<4, 3>));
  RUN("only exp (no t, no red, no acc)", 4, 410, (bwd<4, 14>));
  RUN("nothing (no exp,t,red,acc)", 4, 410, (bwd<4, 15>));
  RUN("scalar + prefetch", 4, 410, (bwd_pf<4>));
  RUN("scalar + prefetch RG=8", 8, 820, (bwd_pf<8>));
#define RUNL(name, RG, csplit_req, KERNEL)                                                                       \
  { long cchunk = (N + (csplit_req) - 1) / (csplit_req); cchunk = (cchunk + 3) / 4 * 4; int cs = (int)((N + cchunk - 1) / cchunk);  \
    const int ntile = (G + 64 * RG - 1) / (64 * RG); dim3 grid((ntile + 3) / 4, cs); float best = 1e9;           \
    for (int it = 0; it < 5; ++it) { CK(hipEventRecord(e0));                                                     \
      hipLaunchKernelGGL(KERNEL, grid, dim3(256), (size_t)cchunk * 48, 0, dc, dF, dem, dL, dmu, dVs, dV, dg, ddF, N, G, cchunk);   \
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());                                \
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (it) best = std::min(best, ms); }                       \
    printf("%-34s RG %d csplit %4d %8.1f us\n", name, RG, cs, best * 1e3); }
  RUNL("lds CU=1", 4, 410, (bwd_lds<4, 1>));
  RUNL("lds CU=2", 4, 410, (bwd_lds<4, 2>));
  RUNL("lds CU=4", 4, 410, (bwd_lds<4, 4>));
  RUNL("lds CU=2 RG=8", 8, 820, (bwd_lds<8, 2>));
  RUN("full RG=8", 8, 820, (bwd<8, 0>));
  RUN("full RG=2", 2, 205, (bwd<2, 0>));
  RUN("full RG=1", 1, 103, (bwd<1, 0>));
  return 0;
}

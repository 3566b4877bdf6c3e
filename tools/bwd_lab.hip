// Kernel lab for the backward sweep (NC = 8, D = 1, RG genes per lane). Not part of the product.
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
struct Args { const float *coef, *F, *em, *Lb, *mu, *Vs, *V; float *gpart, *dFpart; long N; int G; long cchunk; };

// ---- v0: library kernel
template <int RG>
__global__ void __launch_bounds__(256) bwd_v0(Args a) {
  const int lane = threadIdx.x & 63, tile = blockIdx.x * 4 + (threadIdx.x >> 6), gbase = tile * 64 * RG;
  if (gbase >= a.G) return;
  float l[RG][NC], m_[RG], vs[RG], v[RG], accU[RG], accUF[RG];
#pragma unroll
  for (int r = 0; r < RG; ++r) {
    const int g = gbase + r * 64 + lane; const bool ok = g < a.G; const int gg = ok ? g : a.G - 1;
#pragma unroll
    for (int c = 0; c < NC; ++c) l[r][c] = ok ? a.Lb[(long)gg * 8 + c] : 0.f;
    m_[r] = ok ? a.mu[gg] : 0.f; vs[r] = ok ? a.Vs[gg] : 0.f; v[r] = ok ? a.V[gg] : 0.f; accU[r] = accUF[r] = 0.f;
  }
  const long n0 = (long)blockIdx.y * a.cchunk, n1 = std::min(n0 + a.cchunk, a.N);
  for (long n = n0; n < n1; ++n) {
    float cf[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) cf[c] = a.coef[n * 8 + c];
    const float f = a.F[n], em = a.em[n];
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
    const float tot = wave_sum63(dsum);
    if (lane == 63) a.dFpart[(long)tile * a.N + n] = tot;
  }
#pragma unroll
  for (int r = 0; r < RG; ++r) {
    const int g = gbase + r * 64 + lane;
    if (g < a.G) { float* gp = a.gpart + ((long)blockIdx.y * a.G + g) * 2; gp[0] = accU[r]; gp[1] = m_[r] * accUF[r]; }
  }
}

// ---- v1: cell data staged in LDS ([cchunk][12]: coef 8, f, em), CU cells per iteration
template <int RG, int CU>
__global__ void __launch_bounds__(256) bwd_v1(Args a) {
  extern __shared__ float lds[];
  const long n0 = (long)blockIdx.y * a.cchunk, n1 = std::min(n0 + a.cchunk, a.N);
  const int nc = (int)(n1 - n0);
  for (int i = threadIdx.x; i < nc * 2; i += 256) reinterpret_cast<float4*>(lds)[(i >> 1) * 3 + (i & 1)] = reinterpret_cast<const float4*>(a.coef + n0 * 8)[i];
  for (int i = threadIdx.x; i < nc; i += 256) { lds[i * 12 + 8] = a.F[n0 + i]; lds[i * 12 + 9] = a.em[n0 + i]; }
  __syncthreads();
  const int lane = threadIdx.x & 63, tile = blockIdx.x * 4 + (threadIdx.x >> 6), gbase = tile * 64 * RG;
  if (gbase >= a.G) return;
  float l[RG][NC], m_[RG], vs[RG], v[RG], accU[RG], accUF[RG];
#pragma unroll
  for (int r = 0; r < RG; ++r) {
    const int g = gbase + r * 64 + lane; const bool ok = g < a.G; const int gg = ok ? g : a.G - 1;
#pragma unroll
    for (int c = 0; c < NC; ++c) l[r][c] = ok ? a.Lb[(long)gg * 8 + c] : 0.f;
    m_[r] = ok ? a.mu[gg] : 0.f; vs[r] = ok ? a.Vs[gg] : 0.f; v[r] = ok ? a.V[gg] : 0.f; accU[r] = accUF[r] = 0.f;
  }
  float* out = a.dFpart + (long)tile * a.N + n0;
  int i = 0;
  for (; i + CU <= nc; i += CU) {
    float tot[CU];
#pragma unroll
    for (int j = 0; j < CU; ++j) {
      const float4 c0 = reinterpret_cast<const float4*>(lds)[(i + j) * 3], c1 = reinterpret_cast<const float4*>(lds)[(i + j) * 3 + 1];
      const float4 c2 = reinterpret_cast<const float4*>(lds)[(i + j) * 3 + 2];
      const float cf[NC] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
      const float f = c2.x, em = c2.y;
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
      tot[j] = dsum;
    }
#pragma unroll
    for (int j = 0; j < CU; ++j) tot[j] = wave_sum63(tot[j]);
    if (lane == 63) {
#pragma unroll
      for (int j = 0; j < CU; ++j) out[i + j] = tot[j];
    }
  }
  for (; i < nc; ++i) {
    const float* c = lds + i * 12; const float f = c[8], em = c[9];
    float dsum = 0.f;
#pragma unroll
    for (int r = 0; r < RG; ++r) {
      const float e = __builtin_amdgcn_exp2f(fmaf(f, vs[r], -em));
      float t = 0.f;
#pragma unroll
      for (int cc = 0; cc < NC; ++cc) t = fmaf(c[cc], l[r][cc], t);
      const float u = e * t;
      accU[r] += u; accUF[r] = fmaf(u, f, accUF[r]);
      dsum = fmaf(m_[r] * u, v[r], dsum);
    }
    const float tt = wave_sum63(dsum);
    if (lane == 63) out[i] = tt;
  }
#pragma unroll
  for (int r = 0; r < RG; ++r) {
    const int g = gbase + r * 64 + lane;
    if (g < a.G) { float* gp = a.gpart + ((long)blockIdx.y * a.G + g) * 2; gp[0] = accU[r]; gp[1] = m_[r] * accUF[r]; }
  }
}

int main(int argc, char** argv) {
  long N = argc > 1 ? atol(argv[1]) : 100000; int G = argc > 2 ? atoi(argv[2]) : 5000;
  std::vector<float> F(N), em(N), Vs(G), V(G), mu(G), Lb((size_t)G * 8), coef((size_t)N * 8);
  srand(1); auto rnd = []() { return (float)rand() / RAND_MAX; };
  float vmin = 1e9, vmax = -1e9;
  for (int g = 0; g < G; ++g) { V[g] = (rnd() - 0.5f) * 0.8f; Vs[g] = V[g] * 1.442695f; vmin = std::min(vmin, Vs[g]); vmax = std::max(vmax, Vs[g]); mu[g] = rnd() + 0.1f; }
  for (long i = 0; i < N; ++i) { F[i] = (rnd() - 0.5f) * 4.f; em[i] = std::max(F[i] * vmin, F[i] * vmax); }
  for (auto& v : Lb) v = 1.f + (int)(rnd() * 3.99f);
  for (auto& v : coef) v = -rnd() * 1e-3f;
  Args a; float *dc, *dF, *dem, *dL, *dmu, *dVs, *dV, *dg, *ddF;
  const int maxsplit = 1024, maxtile = 128;
  CK(hipMalloc(&dc, N * 32)); CK(hipMalloc(&dF, N * 4)); CK(hipMalloc(&dem, N * 4)); CK(hipMalloc(&dL, (size_t)G * 32)); CK(hipMalloc(&dmu, G * 4));
  CK(hipMalloc(&dVs, G * 4)); CK(hipMalloc(&dV, G * 4)); CK(hipMalloc(&dg, (size_t)maxsplit * G * 8)); CK(hipMalloc(&ddF, (size_t)maxtile * N * 4));
  CK(hipMemcpy(dc, coef.data(), N * 32, hipMemcpyHostToDevice)); CK(hipMemcpy(dF, F.data(), N * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dem, em.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dL, Lb.data(), (size_t)G * 32, hipMemcpyHostToDevice));
  CK(hipMemcpy(dmu, mu.data(), G * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dVs, Vs.data(), G * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dV, V.data(), G * 4, hipMemcpyHostToDevice));
  a = {dc, dF, dem, dL, dmu, dVs, dV, dg, ddF, N, G, 0};
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<double> refg, refF;
  auto check = [&](int RG, int csplit, const char* name, float ms) {
    const int ntile = (G + 64 * RG - 1) / (64 * RG);
    std::vector<float> g((size_t)csplit * G * 2), d((size_t)ntile * N);
    CK(hipMemcpy(g.data(), dg, g.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(d.data(), ddF, d.size() * 4, hipMemcpyDeviceToHost));
    std::vector<double> sg((size_t)G * 2, 0.0), sF(N, 0.0);
    for (int s = 0; s < csplit; ++s) for (size_t i = 0; i < sg.size(); ++i) sg[i] += g[(size_t)s * G * 2 + i];
    for (int t = 0; t < ntile; ++t) for (long n = 0; n < N; ++n) sF[n] += d[(size_t)t * N + n];
    if (refg.empty()) { refg = sg; refF = sF; }
    double eg = 0, eF = 0, mg = 0, mF = 0;
    for (size_t i = 0; i < sg.size(); ++i) { eg = std::max(eg, std::fabs(sg[i] - refg[i])); mg = std::max(mg, std::fabs(refg[i])); }
    for (long n = 0; n < N; ++n) { eF = std::max(eF, std::fabs(sF[n] - refF[n])); mF = std::max(mF, std::fabs(refF[n])); }
    const double flops = (double)N * G * (4.0 * 8 + 6 + 1);
    printf("%-24s RG %d csplit %4d %8.1f us %6.1f TFLOP/s  err gene %.1e cell %.1e\n", name, RG, csplit, ms * 1e3, flops / ms / 1e9, eg / mg, eF / mF);
  };
#define RUN(name, RG, csplit_req, lds, KERNEL)                                                                  \
  { long cchunk = (N + (csplit_req) - 1) / (csplit_req); int cs = (int)((N + cchunk - 1) / cchunk); a.cchunk = cchunk; \
    const int ntile = (G + 64 * RG - 1) / (64 * RG); dim3 grid((ntile + 3) / 4, cs); float best = 1e9;           \
    for (int it = 0; it < 5; ++it) { CK(hipEventRecord(e0));                                                     \
      hipLaunchKernelGGL(KERNEL, grid, dim3(256), (lds) ? (size_t)cchunk * 48 : 0, 0, a);                        \
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());                                \
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (it) best = std::min(best, ms); }                       \
    check(RG, cs, name, best); }
  for (int rep = 0; rep < 2; ++rep) {
    RUN("v0 scalar", 4, 410, 0, (bwd_v0<4>));
    RUN("v0 scalar", 4, 820, 0, (bwd_v0<4>));
    RUN("v0 scalar", 2, 205, 0, (bwd_v0<2>));
    RUN("v0 scalar", 8, 820, 0, (bwd_v0<8>));
    RUN("v1 lds CU=1", 4, 410, 1, (bwd_v1<4, 1>));
    RUN("v1 lds CU=2", 4, 410, 1, (bwd_v1<4, 2>));
    RUN("v1 lds CU=4", 4, 410, 1, (bwd_v1<4, 4>));
    RUN("v1 lds CU=4", 4, 820, 1, (bwd_v1<4, 4>));
    RUN("v1 lds CU=4", 2, 205, 1, (bwd_v1<2, 4>));
    RUN("v1 lds CU=2", 8, 820, 1, (bwd_v1<8, 2>));
    RUN("v1 lds CU=4", 8, 820, 1, (bwd_v1<8, 4>));
  }
  return 0;
}

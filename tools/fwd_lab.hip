// Kernel lab for the forward sweep Z = E.M (NC = 8, D = 1): variants timed against each other in ONE
// process on the same random data (cdna_hip_programming.md §5.4 rule 24).  Not part of the product.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/fwd_lab tools/fwd_lab.hip && /tmp/fwd_lab [N G]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
constexpr int NC = 8;

// ---- v0: the library kernel (lane = cell, scalar-loaded M)
__global__ void __launch_bounds__(256) fwd_v0(const float* __restrict__ F, const float* __restrict__ em2, const float* __restrict__ Vs,
                                              const float* __restrict__ M, float* __restrict__ Zp, long N, int G, int gchunk) {
  const long n = (long)blockIdx.x * 256 + threadIdx.x;
  const long nn = n < N ? n : N - 1;
  const int g0 = blockIdx.y * gchunk, g1 = min(G, g0 + gchunk);
  const float f = F[nn], em = em2[nn];
  float z[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) z[c] = 0.f;
#pragma unroll 4
  for (int g = g0; g < g1; ++g) {
    const float e = __builtin_amdgcn_exp2f(fmaf(f, Vs[g], -em));
    const float* mg = M + (long)g * 8;
#pragma unroll
    for (int c = 0; c < NC; ++c) z[c] = fmaf(e, mg[c], z[c]);
  }
  if (n < N) { float* zp = Zp + ((long)blockIdx.y * N + n) * 8;
#pragma unroll
    for (int c = 0; c < NC; ++c) zp[c] = z[c]; }
}

// ---- v1: R cells per lane (scalar loads amortised, R independent exp chains)
template <int R>
__global__ void __launch_bounds__(256) fwd_v1(const float* __restrict__ F, const float* __restrict__ em2, const float* __restrict__ Vs,
                                              const float* __restrict__ M, float* __restrict__ Zp, long N, int G, int gchunk) {
  const long nb = (long)blockIdx.x * 256 * R + threadIdx.x;
  const int g0 = blockIdx.y * gchunk, g1 = min(G, g0 + gchunk);
  float f[R], em[R], z[R][NC];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const long n = nb + r * 256; const long nn = n < N ? n : N - 1;
    f[r] = F[nn]; em[r] = em2[nn];
#pragma unroll
    for (int c = 0; c < NC; ++c) z[r][c] = 0.f;
  }
#pragma unroll 2
  for (int g = g0; g < g1; ++g) {
    const float v = Vs[g];
    const float* mg = M + (long)g * 8;
    float m[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) m[c] = mg[c];
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
    if (n < N) { float* zp = Zp + ((long)blockIdx.y * N + n) * 8;
#pragma unroll
      for (int c = 0; c < NC; ++c) zp[c] = z[r][c]; }
  }
}

// ---- v2: M slice staged in LDS (broadcast ds_read_b128), R cells per lane
template <int R>
__global__ void __launch_bounds__(256) fwd_v2(const float* __restrict__ F, const float* __restrict__ em2, const float* __restrict__ Vs,
                                              const float* __restrict__ M, float* __restrict__ Zp, long N, int G, int gchunk) {
  extern __shared__ float lds[];   // [gchunk][8] M then [gchunk] Vs
  const int g0 = blockIdx.y * gchunk, g1 = min(G, g0 + gchunk);
  const int ng = g1 - g0;
  float4* l4 = reinterpret_cast<float4*>(lds);
  const float4* m4 = reinterpret_cast<const float4*>(M + (long)g0 * 8);
  for (int i = threadIdx.x; i < ng * 2; i += 256) l4[i] = m4[i];
  float* lv = lds + (long)gchunk * 8;
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
    const float v = lv[g];
    const float4 a = l4[2 * g], b = l4[2 * g + 1];
    const float m[NC] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
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
    if (n < N) { float* zp = Zp + ((long)blockIdx.y * N + n) * 8;
#pragma unroll
      for (int c = 0; c < NC; ++c) zp[c] = z[r][c]; }
  }
}

// ---- v3: MFMA 4x4x1 (16 blocks): lane = cell, B operand = M[g][(lane&3) + 4h] read from LDS
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) fwd_v3(const float* __restrict__ F, const float* __restrict__ em2, const float* __restrict__ Vs,
                                              const float* __restrict__ M, float* __restrict__ Zp, long N, int G, int gchunk) {
  extern __shared__ float lds[];   // [gchunk][4][2] : M'[g][j][h] = M[g][4h + j], then Vs
  const int g0 = blockIdx.y * gchunk, g1 = min(G, g0 + gchunk);
  const int ng = g1 - g0;
  for (int i = threadIdx.x; i < ng * 8; i += 256) {
    const int g = i >> 3, c = i & 7;
    lds[g * 8 + (c & 3) * 2 + (c >> 2)] = M[(long)(g0 + g) * 8 + c];
  }
  float* lv = lds + (long)gchunk * 8;
  for (int i = threadIdx.x; i < ng; i += 256) lv[i] = Vs[g0 + i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const long n = (long)blockIdx.x * 256 + threadIdx.x;
  const long nn = n < N ? n : N - 1;
  const float f = F[nn], em = em2[nn];
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  const float2* lb = reinterpret_cast<const float2*>(lds) + (lane & 3);
#pragma unroll 4
  for (int g = 0; g < ng; ++g) {
    const float e = __builtin_amdgcn_exp2f(fmaf(f, lv[g], -em));
    const float2 b = lb[g * 4];
    acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(e, b.x, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(e, b.y, acc1, 0, 0, 0);
  }
  // D_b[i][j]: lane (b, j) register i  ->  cell 4b + i (of this wave), clone j (+4 for acc1)
  const long wbase = (long)blockIdx.x * 256 + (threadIdx.x & ~63);
  const int b = lane >> 2, j = lane & 3;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long nc = wbase + 4 * b + i;
    if (nc < N) {
      float* zp = Zp + ((long)blockIdx.y * N + nc) * 8;
      zp[j] = acc0[i];
      zp[4 + j] = acc1[i];
    }
  }
}

int main(int argc, char** argv) {
  long N = argc > 1 ? atol(argv[1]) : 100000; int G = argc > 2 ? atoi(argv[2]) : 5000;
  std::vector<float> F(N), em(N), Vs(G), M((size_t)G * 8);
  srand(1);
  auto rnd = []() { return (float)rand() / RAND_MAX; };
  float vmin = 1e9, vmax = -1e9;
  for (auto& v : Vs) { v = (rnd() - 0.5f) * 1.2f; vmin = std::min(vmin, v); vmax = std::max(vmax, v); }
  for (long i = 0; i < N; ++i) { F[i] = (rnd() - 0.5f) * 4.f; em[i] = std::max(F[i] * vmin, F[i] * vmax); }
  for (auto& v : M) v = rnd() * 3.f + 0.01f;
  float *dF, *dem, *dVs, *dM, *dZ, *dZref;
  const int maxsplit = 64;
  CK(hipMalloc(&dF, N * 4)); CK(hipMalloc(&dem, N * 4)); CK(hipMalloc(&dVs, G * 4)); CK(hipMalloc(&dM, (size_t)G * 32));
  CK(hipMalloc(&dZ, (size_t)maxsplit * N * 32)); CK(hipMalloc(&dZref, (size_t)N * 32));
  CK(hipMemcpy(dF, F.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dem, em.data(), N * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dVs, Vs.data(), G * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dM, M.data(), (size_t)G * 32, hipMemcpyHostToDevice));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  std::vector<float> ref, cur;
  auto check = [&](int gsplit, const char* name, float ms) {
    cur.assign((size_t)gsplit * N * 8, 0.f);
    CK(hipMemcpy(cur.data(), dZ, cur.size() * 4, hipMemcpyDeviceToHost));
    std::vector<double> z((size_t)N * 8, 0.0);
    for (int s = 0; s < gsplit; ++s) for (size_t i = 0; i < z.size(); ++i) z[i] += cur[(size_t)s * N * 8 + i];
    double err = 0;
    if (ref.empty()) { ref.resize(z.size()); for (size_t i = 0; i < z.size(); ++i) ref[i] = (float)z[i]; }
    for (size_t i = 0; i < z.size(); ++i) err = std::max(err, std::fabs(z[i] - ref[i]) / std::fabs(ref[i]));
    const double flops = (double)N * G * (2.0 * 8 + 2 + 1);
    printf("%-28s gsplit %3d  %8.1f us  %6.1f TFLOP/s  maxrel %.2e\n", name, gsplit, ms * 1e3, flops / ms / 1e9, err);
  };
#define RUN(name, gsplit, cellsPerBlock, lds, KERNEL)                                                              \
  { int gchunk = (G + (gsplit) - 1) / (gsplit); int gs = (G + gchunk - 1) / gchunk;                               \
    dim3 grid((unsigned)((N + (cellsPerBlock) - 1) / (cellsPerBlock)), gs);                                       \
    float best = 1e9;                                                                                             \
    for (int it = 0; it < 6; ++it) {                                                                              \
      CK(hipEventRecord(a));                                                                                      \
      hipLaunchKernelGGL(KERNEL, grid, dim3(256), (lds) ? (size_t)gchunk * 36 : 0, 0, dF, dem, dVs, dM, dZ, N, G, gchunk); \
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipGetLastError());                                   \
      float ms; CK(hipEventElapsedTime(&ms, a, b)); if (it) best = std::min(best, ms); }                          \
    check(gs, name, best); }
  for (int rep = 0; rep < 2; ++rep) {
    RUN("v0 scalar R=1", 5, 256, 0, fwd_v0);
    RUN("v0 scalar R=1", 6, 256, 0, fwd_v0);
    RUN("v1 scalar R=2", 10, 512, 0, (fwd_v1<2>));
    RUN("v1 scalar R=2", 5, 512, 0, (fwd_v1<2>));
    RUN("v1 scalar R=4", 20, 1024, 0, (fwd_v1<4>));
    RUN("v1 scalar R=4", 10, 1024, 0, (fwd_v1<4>));
    RUN("v2 lds R=1", 5, 256, 1, (fwd_v2<1>));
    RUN("v2 lds R=2", 10, 512, 1, (fwd_v2<2>));
    RUN("v2 lds R=2", 5, 512, 1, (fwd_v2<2>));
    RUN("v2 lds R=4", 20, 1024, 1, (fwd_v2<4>));
    RUN("v2 lds R=4", 10, 1024, 1, (fwd_v2<4>));
    RUN("v3 mfma4x4", 5, 256, 1, fwd_v3);
    RUN("v3 mfma4x4", 10, 256, 1, fwd_v3);
    RUN("v3 mfma4x4", 20, 256, 1, fwd_v3);
  }
  return 0;
}

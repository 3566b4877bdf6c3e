// Lab for clonealign_amd/csrc/ca_ymfma.hip.h: the count matrix's two products on the int8 matrix cores.
// Checks the tiled images, the fixed-point parameter images and both streams against an int64 reference (must be
// EXACT), then times them.  Not product code.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/ymfma_lab.bin tools/ymfma_lab.hip && tools/ymfma_lab.bin [N G]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include "../clonealign_amd/csrc/ca_ymfma.hip.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ void fill_y(uint8_t* Y, int64_t N, int G, int Gp) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= N * Gp) return;
  const int g = (int)(i % Gp);
  unsigned h = (unsigned)(i * 2654435761u) ^ (unsigned)(i >> 13);
  h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12;
  uint8_t v = 0;
  if (g < G) { const unsigned r = h & 0xFF; v = r < 180 ? 0 : r < 230 ? 1 + ((h >> 8) & 3) : r < 254 ? (h >> 8) & 0x3F : 255 - ((h >> 8) & 1) * 100; }
  Y[i] = v;
}
// int64 references, per digit
__global__ void ref_yw(const uint8_t* Y, const float* V, int Dv, int K, const unsigned* amax, int64_t N, int G, int Gp, long long* out /*[N][16]*/) {
  const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (n >= N) return;
  const float sc = ldexpf(1.f, ca_fix_exp(__uint_as_float(amax[0])));
  for (int col = 0; col < 4 * K; ++col) {
    long long a = 0;
    for (int g = lane; g < G; g += 64) a += (long long)Y[n * Gp + g] * ca_digit((int)rintf(V[(int64_t)g * Dv + (col >> 2)] * sc), col & 3);
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if (lane == 0) out[n * 16 + col] = a;
  }
}
__global__ void ref_yt(const uint8_t* Y, const float* F, int Df, int K, const unsigned* amax, int64_t N, int G, int Gp, long long* out /*[G][16]*/) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= G) return;
  const float sc = ldexpf(1.f, ca_fix_exp(__uint_as_float(amax[1])));
  for (int col = 0; col < 4 * K; ++col) {
    long long a = 0;
    for (int64_t n = 0; n < N; ++n) a += (long long)Y[n * Gp + g] * ca_digit((int)rintf(F[n * Df + (col >> 2)] * sc), col & 3);
    out[(int64_t)g * 16 + col] = a;
  }
}

template <int TL, int DEPTH> float time_yw(const uint4* Yf, const uint4* Wq, int64_t NT, int GS, int* out, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int grid = (int)((NT + 4 * TL - 1) / (4 * TL));
  hipLaunchKernelGGL((k_yw_mfma_raw<TL, DEPTH>), dim3(grid), dim3(256), 0, 0, Yf, Wq, NT, GS, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_yw_mfma_raw<TL, DEPTH>), dim3(grid), dim3(256), 0, 0, Yf, Wq, NT, GS, out);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps * 1e3f;
}
template <int TL, int DEPTH> float time_yt(const uint4* Yb, const uint4* Pq, int GT, int64_t NS, int csplit, int* out, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int64_t schunk = (NS + csplit - 1) / csplit;
  const dim3 grid((GT + 4 * TL - 1) / (4 * TL), csplit);
  hipLaunchKernelGGL((k_yt_mfma_raw<TL, DEPTH>), grid, dim3(256), 0, 0, Yb, Pq, GT, NS, schunk, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_yt_mfma_raw<TL, DEPTH>), grid, dim3(256), 0, 0, Yb, Pq, GT, NS, schunk, out);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps * 1e3f;
}

int main(int argc, char** argv) {
  const int64_t N = argc > 1 ? atoll(argv[1]) : 100000;
  const int G = argc > 2 ? atoi(argv[2]) : 5000;
  const int K = argc > 3 ? atoi(argv[3]) : 1;
  const int Gp = (G + 1023) / 1024 * 1024;
  const int64_t NT = (N + 15) / 16, NS = (N + 63) / 64;
  const int GS = (G + 63) / 64, GT = (G + 15) / 16;
  printf("N=%lld G=%d K=%d Gp=%d NT=%lld GS=%d GT=%d NS=%lld\n", (long long)N, G, K, Gp, (long long)NT, GS, GT, (long long)NS);
  uint8_t* Y; uint4 *Yf, *Yb, *Wq, *Pq; float *V, *F; unsigned* amax; int *oyw, *oyt; long long *ryw, *ryt;
  const int maxsplit = 16;
  CK(hipMalloc(&Y, N * Gp)); CK(hipMalloc(&Yf, NT * GS * 1024)); CK(hipMalloc(&Yb, (int64_t)GT * NS * 1024));
  CK(hipMalloc(&Wq, (int64_t)GS * 1024)); CK(hipMalloc(&Pq, NS * 1024));
  CK(hipMalloc(&V, (int64_t)G * K * 4)); CK(hipMalloc(&F, N * K * 4)); CK(hipMalloc(&amax, 8));
  CK(hipMalloc(&oyw, NT * 256 * 4)); CK(hipMalloc(&oyt, (int64_t)maxsplit * GT * 256 * 4));
  CK(hipMalloc(&ryw, N * 16 * 8)); CK(hipMalloc(&ryt, (int64_t)G * 16 * 8));
  hipLaunchKernelGGL(fill_y, dim3((unsigned)((N * Gp + 255) / 256)), dim3(256), 0, 0, Y, N, G, Gp);
  std::vector<float> hV((size_t)G * K), hF((size_t)N * K);
  srand(7);
  for (auto& v : hV) v = (float)((rand() / (double)RAND_MAX - 0.5) * 0.7);
  for (auto& v : hF) v = (float)((rand() / (double)RAND_MAX - 0.5) * 6.0);
  hV[3] = 0.f; hF[5] = -3.0f;
  CK(hipMemcpy(V, hV.data(), hV.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(F, hF.data(), hF.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(amax, 0, 8));
  hipLaunchKernelGGL(k_ym_absmax, dim3((unsigned)((std::max<int64_t>(N, G) + 255) / 256)), dim3(256), 0, 0, V, K, (int64_t)G, F, K, N, K, amax);
  hipLaunchKernelGGL(k_ym_quant, dim3((unsigned)(((GS + NS) * 64 + 255) / 256)), dim3(256), 0, 0, V, K, (int64_t)G, GS, F, K, N, NS, K, amax, Wq, Pq);
  hipLaunchKernelGGL(k_tile_yf, dim3((unsigned)((NT * GS * 64 + 255) / 256)), dim3(256), 0, 0, Y, Yf, N, Gp, NT, GS);
  hipLaunchKernelGGL(k_tile_yb, dim3((unsigned)NS, (GT + 3) / 4), dim3(256), 0, 0, Y, Yb, N, Gp, GT, NS);
  CK(hipDeviceSynchronize());
  unsigned ham[2]; CK(hipMemcpy(ham, amax, 8, hipMemcpyDeviceToHost));
  float fa[2]; memcpy(fa, ham, 8);
  printf("amax W %.6f psi %.6f\n", fa[0], fa[1]);
  // ---- exactness
  hipLaunchKernelGGL(ref_yw, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, 0, Y, V, K, K, amax, N, G, Gp, ryw);
  hipLaunchKernelGGL(ref_yt, dim3((G + 255) / 256), dim3(256), 0, 0, Y, F, K, K, amax, N, G, Gp, ryt);
  hipLaunchKernelGGL((k_yw_mfma_raw<4, 2>), dim3((unsigned)((NT + 15) / 16)), dim3(256), 0, 0, Yf, Wq, NT, GS, oyw);
  const int csplit = 6;
  const int64_t schunk = (NS + csplit - 1) / csplit;
  hipLaunchKernelGGL((k_yt_mfma_raw<1, 4>), dim3((GT + 3) / 4, csplit), dim3(256), 0, 0, Yb, Pq, GT, NS, schunk, oyt);
  CK(hipDeviceSynchronize());
  {
    std::vector<int> o((size_t)NT * 256); std::vector<long long> r((size_t)N * 16);
    CK(hipMemcpy(o.data(), oyw, o.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(r.data(), ryw, r.size() * 8, hipMemcpyDeviceToHost));
    long long bad = 0;
    for (int64_t n = 0; n < N; ++n) for (int c = 0; c < 4 * K; ++c) if ((long long)o[n * 16 + c] != r[n * 16 + c]) { if (bad < 5) printf("  yw mismatch n=%lld col=%d got %d want %lld\n", (long long)n, c, o[n * 16 + c], r[n * 16 + c]); ++bad; }
    printf("YW digits: %lld mismatches of %lld\n", bad, (long long)N * 4 * K);
  }
  {
    std::vector<int> o((size_t)csplit * GT * 256); std::vector<long long> r((size_t)G * 16);
    CK(hipMemcpy(o.data(), oyt, o.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(r.data(), ryt, r.size() * 8, hipMemcpyDeviceToHost));
    long long bad = 0;
    for (int g = 0; g < G; ++g) for (int c = 0; c < 4 * K; ++c) {
      long long a = 0;
      for (int sp = 0; sp < csplit; ++sp) a += o[((size_t)sp * GT * 16 + g) * 16 + c];
      if (a != r[(size_t)g * 16 + c]) { if (bad < 5) printf("  yt mismatch g=%d col=%d got %lld want %lld\n", g, c, a, r[(size_t)g * 16 + c]); ++bad; }
    }
    printf("YtPsi digits: %lld mismatches of %lld\n", bad, (long long)G * 4 * K);
  }
  // ---- one copy, both products (k_ys_mfma), K = 1
  // (Round 2's form of the kernel, which left raw int32 digit sums: that is what this section checks digit for digit -- result in
  //  profiles/r02_lab_trb8_corun.txt.  Since round 3 the engine's body combines the digits itself and leaves float partial slabs for
  //  the vector stream's finisher (ca_ys_io); tests/test_gpu_parity.py holds that form to the oracle.  Build with
  //  -DYMFMA_LAB_ROUND2_ONE_COPY against the round-2 header to re-run this section.)
#ifdef YMFMA_LAB_ROUND2_ONE_COPY
  if (K == 1) {
    const int64_t N64 = (N + 63) / 64 * 64;
    const int Gp5 = (G + CA_YS_GW - 1) / CA_YS_GW * CA_YS_GW;      // here Gp (multiple of 1024) serves
    (void)Gp5;
    uint4 *Ys, *Wr, *Pr; int *YWi, *YTi;
    const int nseg = Gp / CA_YS_GW;
    CK(hipMalloc(&Ys, N64 * Gp)); CK(hipMalloc(&Wr, (int64_t)(Gp / 64) * 1024)); CK(hipMalloc(&Pr, (N64 / 64) * 1024));
    CK(hipMalloc(&YWi, (int64_t)nseg * N * 16));
    hipLaunchKernelGGL(k_bias_y, dim3((unsigned)((N64 * (Gp / 16) + 255) / 256)), dim3(256), 0, 0, Y, Ys, N, N64, Gp);
    hipLaunchKernelGGL(k_ym_quant, dim3((unsigned)(((Gp / 64 + N64 / 64) * 64 + 255) / 256)), dim3(256), 0, 0, V, K, (int64_t)G, Gp / 64, F, K, N, N64 / 64, K, amax, Wr, Pr, 1);
    int *Wsum, *Psum;
    CK(hipMalloc(&Wsum, (Gp / 64) * 16)); CK(hipMalloc(&Psum, (N64 / 64) * 16));
    hipLaunchKernelGGL(k_ym_digit_sums, dim3((unsigned)((Gp / 64 * 64 + 255) / 256)), dim3(256), 0, 0, Wr, (int64_t)(Gp / 64), Wsum);
    hipLaunchKernelGGL(k_ym_digit_sums, dim3((unsigned)((N64 / 64 * 64 + 255) / 256)), dim3(256), 0, 0, Pr, N64 / 64, Psum);
    CK(hipDeviceSynchronize());
    std::vector<int> hW((size_t)(Gp / 64) * 4), hP((size_t)(N64 / 64) * 4);
    CK(hipMemcpy(hW.data(), Wsum, hW.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hP.data(), Psum, hP.size() * 4, hipMemcpyDeviceToHost));
    long long Wtot[4] = {0, 0, 0, 0}, Ptot[4] = {0, 0, 0, 0};
    for (size_t i = 0; i < hW.size(); ++i) Wtot[i & 3] += hW[i];
    for (size_t i = 0; i < hP.size(); ++i) Ptot[i & 3] += hP[i];
    for (int RS : {128, 256, 512}) {
      const int nrg = (int)((N + 4 * RS - 1) / (4 * RS));
      CK(hipMalloc(&YTi, (int64_t)nrg * Gp * 16));
      CK(hipMemset(YWi, 0xFF, (int64_t)nseg * N * 16));
      const size_t lds = CA_YS_LDS_BYTES;
      hipLaunchKernelGGL(k_ys_mfma, dim3(nrg * nseg), dim3(256), lds, 0, reinterpret_cast<const uint8_t*>(Ys), Wr, Pr, N, Gp, RS, YWi, YTi);
      CK(hipDeviceSynchronize());
      std::vector<int> o((size_t)nseg * N * 4), ot((size_t)nrg * Gp * 4); std::vector<long long> r((size_t)N * 16), rt((size_t)G * 16);
      CK(hipMemcpy(o.data(), YWi, o.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(ot.data(), YTi, ot.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(r.data(), ryw, r.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(rt.data(), ryt, rt.size() * 8, hipMemcpyDeviceToHost));
      long long bad = 0, badt = 0;
      for (int64_t n = 0; n < N; ++n) for (int pd = 0; pd < 4; ++pd) {
        long long a = 128 * Wtot[pd]; for (int sg = 0; sg < nseg; ++sg) a += o[((size_t)sg * N + n) * 4 + pd];
        if (a != r[n * 16 + pd]) { if (bad < 4) printf("  ys/yw mismatch n=%lld p=%d got %lld want %lld\n", (long long)n, pd, a, r[n * 16 + pd]); ++bad; }
      }
      for (int g = 0; g < G; ++g) for (int pd = 0; pd < 4; ++pd) {
        long long a = 128 * Ptot[pd]; for (int q = 0; q < nrg; ++q) a += ot[((size_t)q * Gp + g) * 4 + pd];
        if (a != rt[(size_t)g * 16 + pd]) { if (badt < 4 || (pd == 0 && badt < 200 && RS == 128)) printf("  ys/yt mismatch g=%d (mod512 %d) p=%d got %lld want %lld\n", g, g % 512, pd, a, rt[(size_t)g * 16 + pd]); ++badt; }
      }
      hipEvent_t ea, eb; CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
      CK(hipEventRecord(ea));
      for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_ys_mfma, dim3(nrg * nseg), dim3(256), lds, 0, reinterpret_cast<const uint8_t*>(Ys), Wr, Pr, N, Gp, RS, YWi, YTi);
      CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb));
      float ms; CK(hipEventElapsedTime(&ms, ea, eb));
      printf("ONE COPY k_ys_mfma RS=%d: YW %lld / YtPsi %lld mismatches;  %.1f us  %.2f TB/s stored (%d blocks)\n", RS, bad, badt, ms / 20 * 1e3,
             (double)N64 * Gp / (ms / 20 * 1e-3) * 1e-12, nrg * nseg);
      CK(hipFree(YTi));
    }
  }
#endif
  // ---- timing
  const double bytes = (double)N * G;
  const int reps = 30;
#define TYW(TL, DP) { const float us = time_yw<TL, DP>(Yf, Wq, NT, GS, oyw, reps); printf("yw  TL=%d depth=%d           %8.1f us  %.2f TB/s (stored %.0f MB)\n", TL, DP, us, bytes / us * 1e-6, (double)NT * GS * 1024 / 1e6); }
  TYW(1, 2) TYW(1, 4) TYW(1, 8) TYW(2, 2) TYW(2, 4) TYW(4, 1) TYW(4, 2) TYW(4, 3)
#define TYT(TL, DP, CS) { const float us = time_yt<TL, DP>(Yb, Pq, GT, NS, CS, oyt, reps); printf("yt  TL=%d depth=%d csplit=%2d %8.1f us  %.2f TB/s\n", TL, DP, CS, us, bytes / us * 1e-6); }
  TYT(1, 2, 8) TYT(1, 4, 8) TYT(1, 8, 8) TYT(1, 4, 12) TYT(1, 4, 16) TYT(2, 2, 12) TYT(2, 4, 12) TYT(2, 4, 8) TYT(2, 4, 6) TYT(4, 2, 12) TYT(4, 2, 16)
  return 0;
}

// VALU calibration on gfx950: cycles per wave-instruction for v_fma / v_pk_fma / v_exp mixes at full occupancy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed, unsigned long long* clk) {
  float z[8]; float x = seed + threadIdx.x * 1e-6f;
#pragma unroll
  for (int c = 0; c < 8; ++c) z[c] = c;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float e = x;
      if (MODE == 0 || MODE == 2) e = __builtin_amdgcn_exp2f(x * 0.999f - 0.5f);   // fma + exp
      if (MODE == 1 || MODE == 2) {
#pragma unroll
        for (int c = 0; c < 8; ++c) z[c] = fmaf(e, 1.0001f + c, z[c]);
      } else z[u] += e;
      x = x + 1e-7f;
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0; for (int c = 0; c < 8; ++c) s += z[c];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
int main() {
  float* out; unsigned long long* clk; CK(hipMalloc(&out, 8192 * 256 * 4)); CK(hipMalloc(&clk, 16));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int iters = 20000;
  for (int blocks : {256, 1024, 2048, 4096}) for (int mode = 0; mode < 3; ++mode) {
    float best = 1e9; unsigned long long h[2];
    for (int it = 0; it < 3; ++it) {
      CK(hipEventRecord(a));
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.1f, clk);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.1f, clk);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.1f, clk);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    double waves_per_simd = blocks * 4.0 / 1024.0;
    double ghz = (double)h[0] / ((double)h[1] / 100e6) / 1e9;
    // per "gene step": total SIMD time / (wave-steps per SIMD)
    double steps = (double)iters * 4 * waves_per_simd;
    double cyc = best * 1e-3 * ghz * 1e9 / steps;
    printf("blocks %5d mode %d (%s): %8.3f ms  clock %.2f GHz  -> %.1f cycles per wave-step per SIMD\n", blocks, mode,
           mode == 0 ? "fma+exp+add" : mode == 1 ? "8 fma" : "fma+exp+8fma", best, ghz, cyc);
  }
}

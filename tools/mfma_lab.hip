// MFMA issue-rate calibration (f32 shapes) on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed, unsigned long long* clk) {
  float a = seed + threadIdx.x * 1e-3f, b = 1.f + threadIdx.x * 1e-4f;
  f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  f16v d0 = {0}, d1 = {0};
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {  // 4x4x1 x4 independent accumulators
      c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c3, 0, 0, 0);
    } else if (MODE == 1) {  // 16x16x4 x4
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
    } else if (MODE == 2) {  // 32x32x2 x2
      d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d1, 0, 0, 0);
    } else if (MODE == 3) {  // 4x4x1 x2 + exp (the fwd_v3 mix)
      float e = __builtin_amdgcn_exp2f(a * 0.999f - 0.5f);
      c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(e, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(e, b, c1, 0, 0, 0);
      a += 1e-7f;
    } else if (MODE == 4) {  // 16x16x4 + 4 exp (cells x genes x 16 clone columns)
      float e = __builtin_amdgcn_exp2f(a * 0.999f - 0.5f);
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(e, b, c0, 0, 0, 0);
      a += 1e-7f;
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int i = 0; i < 4; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  for (int i = 0; i < 16; ++i) s += d0[i] + d1[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
int main() {
  float* out; unsigned long long* clk; CK(hipMalloc(&out, 8192 * 256 * 4)); CK(hipMalloc(&clk, 16));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int iters = 20000;
  const char* names[] = {"4x4x1 x4", "16x16x4 x4", "32x32x2 x2", "exp + 2x 4x4x1", "exp + 16x16x4"};
  const int per[] = {4, 4, 2, 1, 1};
  for (int blocks : {256, 1024, 2048}) for (int mode = 0; mode < 5; ++mode) {
    float best = 1e9; unsigned long long h[2];
    for (int it = 0; it < 3; ++it) {
      CK(hipEventRecord(a));
      switch (mode) {
        case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.1f, clk); break;
        case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.1f, clk); break;
        case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.1f, clk); break;
        case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.1f, clk); break;
        default: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.1f, clk); break;
      }
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    double wps = blocks * 4.0 / 1024.0, ghz = (double)h[0] / ((double)h[1] / 100e6) / 1e9;
    double cyc = best * 1e-3 * ghz * 1e9 / ((double)iters * per[mode] * wps);
    printf("blocks %5d %-16s: %8.3f ms  clock %.2f GHz -> %.1f SIMD-cycles per %s\n", blocks, names[mode], best, ghz, cyc,
           mode < 3 ? "MFMA" : "step");
  }
}

// Co-run lab: what does a VALU-bound kernel lose when an HBM stream runs beside it on another queue, and why?
// A spin kernel with NO memory traffic (4 waves per SIMD of dependent-free v_fma_f32 + v_exp_f32, the forward sweep's mix)
// is timed alone and beside the int8-MFMA row stream of ca_ymfma.hip.h; each spin block stamps s_memtime / s_memrealtime
// so the shader clock it ran at is known.  If the spin kernel slows down with no memory access of its own, the loss is
// issue arbitration or clock (power), not memory latency.  Not product code.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/corun_lab.bin tools/corun_lab.hip && tools/corun_lab.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "../clonealign_amd/csrc/ca_ymfma.hip.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int EXP>
__global__ void __launch_bounds__(256) spin(float* out, int iters, unsigned long long* stamps) {
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = 1.0f + 0.001f * (threadIdx.x + i);
  const float m = 0.9999f, c = 0.0001f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = fmaf(a[i], m, c);
    }
    if (EXP) {
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = __builtin_amdgcn_exp2f(a[i] * 0.001f) ;
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i];
  if (s == 123.456f) out[0] = s;
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - c0;
    stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
}

__global__ void fill(uint4* p, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = (uint4){(unsigned)i * 2654435761u, (unsigned)(i >> 3), 0x80808080u, (unsigned)i};
}

static double ghz(const std::vector<unsigned long long>& st) {
  std::vector<double> v;
  for (size_t b = 0; b < st.size() / 2; ++b) if (st[2 * b + 1]) v.push_back((double)st[2 * b] / (double)st[2 * b + 1] * 0.1);
  std::sort(v.begin(), v.end());
  return v.empty() ? 0.0 : v[v.size() / 2];
}

int main() {
  const int64_t N = 100000; const int G = 5000;
  const int64_t NT = (N + 15) / 16; const int GS = (G + 63) / 64;
  uint4 *Yf, *Yf2, *Wq; int* o; float* so; unsigned long long* stamps;
  const int nblk = 1024;   // 4 blocks per CU = 4 waves per SIMD
  CK(hipMalloc(&Yf, NT * GS * 1024)); CK(hipMalloc(&Yf2, NT * GS * 1024)); CK(hipMalloc(&Wq, (int64_t)GS * 1024)); CK(hipMalloc(&o, NT * 256 * 4));
  CK(hipMalloc(&so, 64)); CK(hipMalloc(&stamps, nblk * 16));
  hipLaunchKernelGGL(fill, dim3((unsigned)((NT * GS * 64 + 255) / 256)), dim3(256), 0, 0, Yf, NT * GS * 64);
  hipLaunchKernelGGL(fill, dim3((unsigned)((NT * GS * 64 + 255) / 256)), dim3(256), 0, 0, Yf2, NT * GS * 64);
  hipLaunchKernelGGL(fill, dim3((unsigned)((GS * 64 + 255) / 256)), dim3(256), 0, 0, Wq, (int64_t)GS * 64);
  CK(hipDeviceSynchronize());
  hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t a1, b1, a2, b2; CK(hipEventCreate(&a1)); CK(hipEventCreate(&b1)); CK(hipEventCreate(&a2)); CK(hipEventCreate(&b2));
  const int grid_yw = (int)((NT + 15) / 16);
  std::vector<unsigned long long> st(2 * nblk);
  for (int expv = 0; expv < 2; ++expv) {
    const int iters = expv ? 1300 : 2600;
    auto launch_spin = [&](hipStream_t s) {
      if (expv) hipLaunchKernelGGL(spin<1>, dim3(nblk), dim3(256), 0, s, so, iters, stamps);
      else hipLaunchKernelGGL(spin<0>, dim3(nblk), dim3(256), 0, s, so, iters, stamps);
    };
    auto launch_yw = [&](hipStream_t s, int rep) {   // alternate two images so the Infinity Cache cannot serve re-reads
      hipLaunchKernelGGL((k_yw_mfma_raw<4, 2>), dim3(grid_yw), dim3(256), 0, s, (rep & 1) ? Yf2 : Yf, Wq, NT, GS, o);
    };
    float t_spin = 0, t_yw = 0, t_spin_co = 0, t_yw_co = 0, t_spin_co2 = 0, t_yw_co2 = 0; double g0 = 0, g1 = 0, g2 = 0;
    for (int w = 0; w < 3; ++w) { launch_spin(s1); launch_yw(s2, w); }
    CK(hipDeviceSynchronize());
    const int reps = 10;
    // alone
    CK(hipEventRecord(a1, s1)); for (int r = 0; r < reps; ++r) launch_spin(s1); CK(hipEventRecord(b1, s1)); CK(hipEventSynchronize(b1));
    CK(hipEventElapsedTime(&t_spin, a1, b1)); CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost)); g0 = ghz(st);
    CK(hipEventRecord(a2, s2)); for (int r = 0; r < reps; ++r) launch_yw(s2, r); CK(hipEventRecord(b2, s2)); CK(hipEventSynchronize(b2));
    CK(hipEventElapsedTime(&t_yw, a2, b2));
    // together, stream launched first (its waves are the older ones)
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a2, s2)); CK(hipEventRecord(a1, s1));
    for (int r = 0; r < reps; ++r) { launch_yw(s2, r); launch_spin(s1); }
    CK(hipEventRecord(b1, s1)); CK(hipEventRecord(b2, s2)); CK(hipEventSynchronize(b1)); CK(hipEventSynchronize(b2));
    CK(hipEventElapsedTime(&t_spin_co, a1, b1)); CK(hipEventElapsedTime(&t_yw_co, a2, b2));
    CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost)); g1 = ghz(st);
    // together, spin launched first
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a1, s1)); CK(hipEventRecord(a2, s2));
    for (int r = 0; r < reps; ++r) { launch_spin(s1); launch_yw(s2, r); }
    CK(hipEventRecord(b1, s1)); CK(hipEventRecord(b2, s2)); CK(hipEventSynchronize(b1)); CK(hipEventSynchronize(b2));
    CK(hipEventElapsedTime(&t_spin_co2, a1, b1)); CK(hipEventElapsedTime(&t_yw_co2, a2, b2));
    CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost)); g2 = ghz(st);
    printf("spin(%s): alone %.1f us @ %.2f GHz | stream alone %.1f us (%.2f TB/s, no cache re-use)\n", expv ? "fma+exp" : "fma", t_spin / reps * 1e3,
           g0, t_yw / reps * 1e3, (double)N * G / (t_yw / reps * 1e-3) * 1e-12);
    printf("   together (stream first): spin %.1f us @ %.2f GHz, stream %.1f us;  sum alone %.1f, wall %.1f\n", t_spin_co / reps * 1e3, g1,
           t_yw_co / reps * 1e3, (t_spin + t_yw) / reps * 1e3, std::max(t_spin_co, t_yw_co) / reps * 1e3);
    printf("   together (spin first)  : spin %.1f us @ %.2f GHz, stream %.1f us;  wall %.1f\n", t_spin_co2 / reps * 1e3, g2, t_yw_co2 / reps * 1e3,
           std::max(t_spin_co2, t_yw_co2) / reps * 1e3);
  }
  return 0;
}

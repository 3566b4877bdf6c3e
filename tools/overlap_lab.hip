// Does a matrix-core instruction keep the vector ALU of its SIMD busy?  inst_lab.hip says so for the VGPR form
// (MFMA + 8 v_fmac = 16 + 8 x 2.3 cycles).  This lab asks the same question for accumulators in AccVGPRs, for the int8
// MFMA the count-matrix streams use, and for SPECIALISED waves (half the waves of a SIMD issue only MFMAs, the other
// half only vector instructions): if the two pipes were independent the specialised mix would take max(), not the sum.
//   hipcc -O3 --offload-arch=gfx950 -o tools/overlap_lab.bin tools/overlap_lab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
constexpr int ITER = 2000;
#define R8(X) X X X X X X X X
#define FM8 "v_fmac_f32 v10, %0, %1\nv_fmac_f32 v11, %0, %1\nv_fmac_f32 v30, %0, %1\nv_fmac_f32 v31, %0, %1\nv_fmac_f32 v32, %0, %1\nv_fmac_f32 v33, %0, %1\nv_fmac_f32 v34, %0, %1\nv_fmac_f32 v35, %0, %1\n"
#define CLOB "v10","v11","v20","v21","v22","v23","v24","v25","v26","v27","v30","v31","v32","v33","v34","v35","a0","a1","a2","a3","a4","a5","a6","a7"
#define KERNEL(name, body)                                                                         \
  __global__ void __launch_bounds__(256) name(float* out, int mode) {                              \
    float a = threadIdx.x * 1e-3f, b = 1.0001f;                                                    \
    for (int i = 0; i < ITER; ++i) { asm volatile(R8(R8(body)) : "+v"(a), "+v"(b) : : CLOB); }     \
    out[blockIdx.x * 256 + threadIdx.x] = a + b;                                                   \
  }
#define MF_V "v_mfma_f32_16x16x32_bf16 v[20:23], v[12:15], v[16:19], v[20:23]\n"
#define MF_A "v_mfma_f32_16x16x32_bf16 a[0:3], v[12:15], v[16:19], a[0:3]\n"
#define MF_AI "v_mfma_f32_16x16x32_bf16 a[0:3], v[12:15], v[16:19], a[4:7]\n"
#define MI_V "v_mfma_i32_16x16x64_i8 v[20:23], v[12:15], v[16:19], v[20:23]\n"
#define MI_A "v_mfma_i32_16x16x64_i8 a[0:3], v[12:15], v[16:19], a[0:3]\n"
#define MI_Z "v_mfma_i32_16x16x64_i8 v[20:23], v[12:15], v[16:19], 0\n"
KERNEL(k_fm8, FM8)
KERNEL(k_mf_v, MF_V)
KERNEL(k_mf_a, MF_A)
KERNEL(k_mi_v, MI_V)
KERNEL(k_mi_a, MI_A)
KERNEL(k_mix_v, MF_V FM8)
KERNEL(k_mix_a, MF_A FM8)
KERNEL(k_mix_ai, MF_AI FM8)
KERNEL(k_mixi_v, MI_V FM8)
KERNEL(k_mixi_a, MI_A FM8)
KERNEL(k_mixi_z, MI_Z FM8)
// specialised: even blocks issue only MFMAs, odd blocks only vector instructions (same instruction counts per pair of blocks
// as the mixes above: one MFMA per 8 v_fmac)
#define SPEC(name, mf)                                                                             \
  __global__ void __launch_bounds__(256) name(float* out, int mode) {                              \
    float a = threadIdx.x * 1e-3f, b = 1.0001f;                                                    \
    const bool domf = mode == 0 ? (blockIdx.x & 1) : ((blockIdx.x >> 3) & 1);                      \
    if (domf) { for (int i = 0; i < ITER; ++i) { asm volatile(R8(R8(mf mf)) : "+v"(a), "+v"(b) : : CLOB); } }                       \
    else { for (int i = 0; i < ITER; ++i) { asm volatile(R8(R8(FM8 FM8)) : "+v"(a), "+v"(b) : : CLOB); } }                       \
    out[blockIdx.x * 256 + threadIdx.x] = a + b;                                                   \
  }
SPEC(k_spec_v, MF_V)
SPEC(k_spec_a, MF_A)
SPEC(k_speci_v, MI_V)

int main() {
  float* out; CK(hipMalloc(&out, 256 * 8192 * 4));
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount; const double clk = p.clockRate * 1e3;
  printf("CUs %d clock %.0f MHz (cycles below are at that nominal clock)\n", cus, clk / 1e6);
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
#define RUN(k, mode, label)                                                                       \
  for (int w : {2, 4}) {                                                                          \
    float best = 1e9;                                                                             \
    for (int it = 0; it < 4; ++it) {                                                              \
      CK(hipEventRecord(a)); hipLaunchKernelGGL(k, dim3(cus * w), dim3(256), 0, 0, out, mode);   \
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (it) best = std::min(best, ms); } \
    printf("%-10s %-28s waves/SIMD %d  %7.2f cycles per body per wave-slot\n", #k, label, w, best * 1e-3 * clk / ((double)ITER * 64 * w)); }
  RUN(k_fm8, 0, "8 v_fmac")
  RUN(k_mf_v, 0, "bf16 MFMA, VGPR acc")
  RUN(k_mf_a, 0, "bf16 MFMA, AGPR acc")
  RUN(k_mi_v, 0, "i8 MFMA, VGPR acc")
  RUN(k_mi_a, 0, "i8 MFMA, AGPR acc")
  RUN(k_mix_v, 0, "bf16 MFMA(V) + 8 fmac")
  RUN(k_mix_a, 0, "bf16 MFMA(A) + 8 fmac")
  RUN(k_mix_ai, 0, "bf16 MFMA(A, indep) + 8 fmac")
  RUN(k_mixi_v, 0, "i8 MFMA(V) + 8 fmac")
  RUN(k_mixi_a, 0, "i8 MFMA(A) + 8 fmac")
  RUN(k_mixi_z, 0, "i8 MFMA(V, C=0) + 8 fmac")
  RUN(k_spec_v, 0, "spec even/odd blk, bf16 V")
  RUN(k_spec_v, 1, "spec by blk>>3, bf16 V")
  RUN(k_spec_a, 0, "spec even/odd blk, bf16 A")
  RUN(k_spec_a, 1, "spec by blk>>3, bf16 A")
  RUN(k_speci_v, 1, "spec by blk>>3, i8 V")
  return 0;
}

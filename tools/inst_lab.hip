// Instruction-issue lab: cycles per wave64 instruction on gfx950 for the opcodes the MFMA forward sweep leans on.
// Each kernel runs ITER x 32 independent instructions of one kind per wave; W waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 -o tools/inst_lab.bin tools/inst_lab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
constexpr int ITER = 2000;

#define R8(X) X X X X X X X X
#define KERNEL(name, body, ...)                                                  \
  __global__ void __launch_bounds__(256) name(float* out) {                        \
    float a = threadIdx.x * 1e-3f, b = 1.0001f;                                    \
    for (int i = 0; i < ITER; ++i) { asm volatile(R8(R8(body)) : "+v"(a), "+v"(b) : : __VA_ARGS__); } \
    out[blockIdx.x * 256 + threadIdx.x] = a + b;                                   \
  }
// each body is ONE instruction writing a scratch register (independent of the others)
KERNEL(k_fma, "v_fma_f32 v10, %0, %1, %0\n", "v10")
KERNEL(k_pkfma, "v_pk_fma_f32 v[10:11], v[12:13], v[14:15], v[16:17]\n", "v10", "v11")
KERNEL(k_pkadd, "v_pk_add_f32 v[10:11], v[12:13], v[14:15]\n", "v10", "v11")
KERNEL(k_exp, "v_exp_f32 v10, %0\n", "v10")
KERNEL(k_cvtpk, "v_cvt_pk_bf16_f32 v10, %0, %1\n", "v10")
KERNEL(k_dot2c, "v_dot2c_f32_bf16 v10, %0, %1\n", "v10")
KERNEL(k_and, "v_and_b32 v10, %0, %1\n", "v10")
KERNEL(k_lshl, "v_lshlrev_b32 v10, 16, %0\n", "v10")
KERNEL(k_perm, "v_perm_b32 v10, %0, %1, %0\n", "v10")
KERNEL(k_mfma, "v_mfma_f32_16x16x32_bf16 v[20:23], v[12:15], v[16:19], v[20:23]\n", "v20", "v21", "v22", "v23")
KERNEL(k_mfma_ind, "v_mfma_f32_16x16x32_bf16 v[20:23], v[12:15], v[16:19], v[24:27]\n", "v20", "v21", "v22", "v23")
KERNEL(k_mfma32, "v_mfma_f32_32x32x16_bf16 v[20:35], v[12:15], v[16:19], v[20:35]\n", "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35")
KERNEL(k_mix, "v_mfma_f32_16x16x32_bf16 v[20:23], v[12:15], v[16:19], v[20:23]\nv_fma_f32 v10, %0, %1, %0\nv_fma_f32 v11, %0, %1, %0\nv_fma_f32 v10, %0, %1, %0\nv_fma_f32 v11, %0, %1, %0\n", "v10","v11","v20","v21","v22","v23")
KERNEL(k_fmac, "v_fmac_f32 v10, %0, %1\n", "v10")
KERNEL(k_mul, "v_mul_f32 v10, %0, %1\n", "v10")
KERNEL(k_add, "v_add_f32 v10, %0, %1\n", "v10")
KERNEL(k_fma3, "v_fma_f32 v10, %0, %1, v11\n", "v10", "v11")
KERNEL(k_mix_pk, "v_mfma_f32_16x16x32_bf16 v[20:23], v[12:15], v[16:19], v[20:23]\nv_pk_fma_f32 v[10:11], v[12:13], v[14:15], v[16:17]\nv_pk_fma_f32 v[30:31], v[12:13], v[14:15], v[16:17]\nv_pk_fma_f32 v[32:33], v[12:13], v[14:15], v[16:17]\nv_pk_fma_f32 v[34:35], v[12:13], v[14:15], v[16:17]\n", "v10","v11","v20","v21","v22","v23","v30","v31","v32","v33","v34","v35")
KERNEL(k_mix_fmac8, "v_mfma_f32_16x16x32_bf16 v[20:23], v[12:15], v[16:19], v[20:23]\nv_fmac_f32 v10, %0, %1\nv_fmac_f32 v11, %0, %1\nv_fmac_f32 v30, %0, %1\nv_fmac_f32 v31, %0, %1\nv_fmac_f32 v32, %0, %1\nv_fmac_f32 v33, %0, %1\nv_fmac_f32 v34, %0, %1\nv_fmac_f32 v35, %0, %1\n", "v10","v11","v20","v21","v22","v23","v30","v31","v32","v33","v34","v35")
KERNEL(k_mix_exp4, "v_mfma_f32_16x16x32_bf16 v[20:23], v[12:15], v[16:19], v[20:23]\nv_exp_f32 v10, %0\nv_exp_f32 v11, %0\nv_exp_f32 v30, %0\nv_exp_f32 v31, %0\n", "v10","v11","v20","v21","v22","v23","v30","v31")
KERNEL(k_expfma, "v_exp_f32 v10, %0\nv_fma_f32 v11, %0, %1, %0\n", "v10", "v11")


__global__ void __launch_bounds__(256) k_tile(float* out) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  for (int i = 0; i < ITER; ++i) { asm volatile(
"v_pk_fma_f32 v[40:41], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[42:43], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[44:45], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[46:47], v[12:13], v[14:15], v[16:17]\n"
"v_exp_f32 v50, %0\n"
"v_exp_f32 v51, %0\n"
"v_exp_f32 v52, %0\n"
"v_exp_f32 v53, %0\n"
"v_exp_f32 v54, %0\n"
"v_exp_f32 v55, %0\n"
"v_exp_f32 v56, %0\n"
"v_exp_f32 v57, %0\n"
"v_cvt_pk_bf16_f32 v60, %0, %1\n"
"v_cvt_pk_bf16_f32 v61, %0, %1\n"
"v_cvt_pk_bf16_f32 v62, %0, %1\n"
"v_cvt_pk_bf16_f32 v63, %0, %1\n"
"v_dot2c_f32_bf16 v64, %0, %1\n"
"v_dot2c_f32_bf16 v65, %0, %1\n"
"v_dot2c_f32_bf16 v66, %0, %1\n"
"v_dot2c_f32_bf16 v67, %0, %1\n"
"v_dot2c_f32_bf16 v68, %0, %1\n"
"v_dot2c_f32_bf16 v69, %0, %1\n"
"v_dot2c_f32_bf16 v70, %0, %1\n"
"v_dot2c_f32_bf16 v71, %0, %1\n"
"v_cvt_pk_bf16_f32 v72, %0, %1\n"
"v_cvt_pk_bf16_f32 v73, %0, %1\n"
"v_cvt_pk_bf16_f32 v74, %0, %1\n"
"v_cvt_pk_bf16_f32 v75, %0, %1\n"
"v_mfma_f32_16x16x32_bf16 v[20:23], v[12:15], v[16:19], v[20:23]\n"
"v_mfma_f32_16x16x32_bf16 v[20:23], v[12:15], v[16:19], v[20:23]\n"
"v_mfma_f32_16x16x32_bf16 v[20:23], v[12:15], v[16:19], v[20:23]\n"
 : "+v"(a), "+v"(b) : : "v20","v21","v22","v23","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79"); }
  out[blockIdx.x * 256 + threadIdx.x] = a + b;
}
__global__ void __launch_bounds__(256) k_tile_nomfma(float* out) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  for (int i = 0; i < ITER; ++i) { asm volatile(
"v_pk_fma_f32 v[40:41], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[42:43], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[44:45], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[46:47], v[12:13], v[14:15], v[16:17]\n"
"v_exp_f32 v50, %0\n"
"v_exp_f32 v51, %0\n"
"v_exp_f32 v52, %0\n"
"v_exp_f32 v53, %0\n"
"v_exp_f32 v54, %0\n"
"v_exp_f32 v55, %0\n"
"v_exp_f32 v56, %0\n"
"v_exp_f32 v57, %0\n"
"v_cvt_pk_bf16_f32 v60, %0, %1\n"
"v_cvt_pk_bf16_f32 v61, %0, %1\n"
"v_cvt_pk_bf16_f32 v62, %0, %1\n"
"v_cvt_pk_bf16_f32 v63, %0, %1\n"
"v_dot2c_f32_bf16 v64, %0, %1\n"
"v_dot2c_f32_bf16 v65, %0, %1\n"
"v_dot2c_f32_bf16 v66, %0, %1\n"
"v_dot2c_f32_bf16 v67, %0, %1\n"
"v_dot2c_f32_bf16 v68, %0, %1\n"
"v_dot2c_f32_bf16 v69, %0, %1\n"
"v_dot2c_f32_bf16 v70, %0, %1\n"
"v_dot2c_f32_bf16 v71, %0, %1\n"
"v_cvt_pk_bf16_f32 v72, %0, %1\n"
"v_cvt_pk_bf16_f32 v73, %0, %1\n"
"v_cvt_pk_bf16_f32 v74, %0, %1\n"
"v_cvt_pk_bf16_f32 v75, %0, %1\n"
 : "+v"(a), "+v"(b) : : "v20","v21","v22","v23","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79"); }
  out[blockIdx.x * 256 + threadIdx.x] = a + b;
}

__global__ void __launch_bounds__(256) k_btile(float* out) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  for (int i = 0; i < ITER; ++i) { asm volatile(
"v_mfma_f32_16x16x32_bf16 v[20:23], v[12:15], v[16:19], 0\n"
"v_mfma_f32_16x16x32_bf16 v[24:27], v[12:15], v[16:19], 0\n"
"v_mfma_f32_16x16x32_bf16 v[28:31], v[12:15], v[16:19], 0\n"
"v_mfma_f32_16x16x32_bf16 v[32:35], v[12:15], v[16:19], 0\n"
"v_pk_fma_f32 v[40:41], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[42:43], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[44:45], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[46:47], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[48:49], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[50:51], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[52:53], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[54:55], v[12:13], v[14:15], v[16:17]\n"
"v_exp_f32 v56, %0\n"
"v_exp_f32 v57, %0\n"
"v_exp_f32 v58, %0\n"
"v_exp_f32 v59, %0\n"
"v_exp_f32 v60, %0\n"
"v_exp_f32 v61, %0\n"
"v_exp_f32 v62, %0\n"
"v_exp_f32 v63, %0\n"
"v_exp_f32 v64, %0\n"
"v_exp_f32 v65, %0\n"
"v_exp_f32 v66, %0\n"
"v_exp_f32 v67, %0\n"
"v_exp_f32 v68, %0\n"
"v_exp_f32 v69, %0\n"
"v_exp_f32 v70, %0\n"
"v_exp_f32 v71, %0\n"
"v_pk_mul_f32 v[72:73], v[12:13], v[14:15]\n"
"v_pk_mul_f32 v[74:75], v[12:13], v[14:15]\n"
"v_pk_mul_f32 v[76:77], v[12:13], v[14:15]\n"
"v_pk_mul_f32 v[78:79], v[12:13], v[14:15]\n"
"v_pk_mul_f32 v[80:81], v[12:13], v[14:15]\n"
"v_pk_mul_f32 v[82:83], v[12:13], v[14:15]\n"
"v_pk_mul_f32 v[84:85], v[12:13], v[14:15]\n"
"v_pk_mul_f32 v[86:87], v[12:13], v[14:15]\n"
"v_pk_add_f32 v[88:89], v[12:13], v[14:15]\n"
"v_pk_add_f32 v[90:91], v[12:13], v[14:15]\n"
"v_pk_add_f32 v[92:93], v[12:13], v[14:15]\n"
"v_pk_add_f32 v[94:95], v[12:13], v[14:15]\n"
"v_pk_add_f32 v[96:97], v[12:13], v[14:15]\n"
"v_pk_add_f32 v[98:99], v[12:13], v[14:15]\n"
"v_pk_add_f32 v[100:101], v[12:13], v[14:15]\n"
"v_pk_add_f32 v[102:103], v[12:13], v[14:15]\n"
"v_pk_fma_f32 v[104:105], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[106:107], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[108:109], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[110:111], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[112:113], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[114:115], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[116:117], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[118:119], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[104:105], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[106:107], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[108:109], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[110:111], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[112:113], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[114:115], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[116:117], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[118:119], v[12:13], v[14:15], v[16:17]\n"
 : "+v"(a), "+v"(b) : : "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119"); }
  out[blockIdx.x * 256 + threadIdx.x] = a + b;
}

__global__ void __launch_bounds__(256) k_btile_nomfma(float* out) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  for (int i = 0; i < ITER; ++i) { asm volatile(
"v_pk_fma_f32 v[40:41], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[42:43], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[44:45], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[46:47], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[48:49], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[50:51], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[52:53], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[54:55], v[12:13], v[14:15], v[16:17]\n"
"v_exp_f32 v56, %0\n"
"v_exp_f32 v57, %0\n"
"v_exp_f32 v58, %0\n"
"v_exp_f32 v59, %0\n"
"v_exp_f32 v60, %0\n"
"v_exp_f32 v61, %0\n"
"v_exp_f32 v62, %0\n"
"v_exp_f32 v63, %0\n"
"v_exp_f32 v64, %0\n"
"v_exp_f32 v65, %0\n"
"v_exp_f32 v66, %0\n"
"v_exp_f32 v67, %0\n"
"v_exp_f32 v68, %0\n"
"v_exp_f32 v69, %0\n"
"v_exp_f32 v70, %0\n"
"v_exp_f32 v71, %0\n"
"v_pk_mul_f32 v[72:73], v[12:13], v[14:15]\n"
"v_pk_mul_f32 v[74:75], v[12:13], v[14:15]\n"
"v_pk_mul_f32 v[76:77], v[12:13], v[14:15]\n"
"v_pk_mul_f32 v[78:79], v[12:13], v[14:15]\n"
"v_pk_mul_f32 v[80:81], v[12:13], v[14:15]\n"
"v_pk_mul_f32 v[82:83], v[12:13], v[14:15]\n"
"v_pk_mul_f32 v[84:85], v[12:13], v[14:15]\n"
"v_pk_mul_f32 v[86:87], v[12:13], v[14:15]\n"
"v_pk_add_f32 v[88:89], v[12:13], v[14:15]\n"
"v_pk_add_f32 v[90:91], v[12:13], v[14:15]\n"
"v_pk_add_f32 v[92:93], v[12:13], v[14:15]\n"
"v_pk_add_f32 v[94:95], v[12:13], v[14:15]\n"
"v_pk_add_f32 v[96:97], v[12:13], v[14:15]\n"
"v_pk_add_f32 v[98:99], v[12:13], v[14:15]\n"
"v_pk_add_f32 v[100:101], v[12:13], v[14:15]\n"
"v_pk_add_f32 v[102:103], v[12:13], v[14:15]\n"
"v_pk_fma_f32 v[104:105], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[106:107], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[108:109], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[110:111], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[112:113], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[114:115], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[116:117], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[118:119], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[104:105], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[106:107], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[108:109], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[110:111], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[112:113], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[114:115], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[116:117], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[118:119], v[12:13], v[14:15], v[16:17]\n"
 : "+v"(a), "+v"(b) : : "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119"); }
  out[blockIdx.x * 256 + threadIdx.x] = a + b;
}

__global__ void __launch_bounds__(256) k_btile_noexp(float* out) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  for (int i = 0; i < ITER; ++i) { asm volatile(
"v_mfma_f32_16x16x32_bf16 v[20:23], v[12:15], v[16:19], 0\n"
"v_mfma_f32_16x16x32_bf16 v[24:27], v[12:15], v[16:19], 0\n"
"v_mfma_f32_16x16x32_bf16 v[28:31], v[12:15], v[16:19], 0\n"
"v_mfma_f32_16x16x32_bf16 v[32:35], v[12:15], v[16:19], 0\n"
"v_pk_fma_f32 v[40:41], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[42:43], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[44:45], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[46:47], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[48:49], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[50:51], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[52:53], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[54:55], v[12:13], v[14:15], v[16:17]\n"
"v_pk_mul_f32 v[72:73], v[12:13], v[14:15]\n"
"v_pk_mul_f32 v[74:75], v[12:13], v[14:15]\n"
"v_pk_mul_f32 v[76:77], v[12:13], v[14:15]\n"
"v_pk_mul_f32 v[78:79], v[12:13], v[14:15]\n"
"v_pk_mul_f32 v[80:81], v[12:13], v[14:15]\n"
"v_pk_mul_f32 v[82:83], v[12:13], v[14:15]\n"
"v_pk_mul_f32 v[84:85], v[12:13], v[14:15]\n"
"v_pk_mul_f32 v[86:87], v[12:13], v[14:15]\n"
"v_pk_add_f32 v[88:89], v[12:13], v[14:15]\n"
"v_pk_add_f32 v[90:91], v[12:13], v[14:15]\n"
"v_pk_add_f32 v[92:93], v[12:13], v[14:15]\n"
"v_pk_add_f32 v[94:95], v[12:13], v[14:15]\n"
"v_pk_add_f32 v[96:97], v[12:13], v[14:15]\n"
"v_pk_add_f32 v[98:99], v[12:13], v[14:15]\n"
"v_pk_add_f32 v[100:101], v[12:13], v[14:15]\n"
"v_pk_add_f32 v[102:103], v[12:13], v[14:15]\n"
"v_pk_fma_f32 v[104:105], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[106:107], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[108:109], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[110:111], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[112:113], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[114:115], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[116:117], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[118:119], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[104:105], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[106:107], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[108:109], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[110:111], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[112:113], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[114:115], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[116:117], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[118:119], v[12:13], v[14:15], v[16:17]\n"
 : "+v"(a), "+v"(b) : : "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119"); }
  out[blockIdx.x * 256 + threadIdx.x] = a + b;
}

__global__ void __launch_bounds__(256) k_btile_pkonly(float* out) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  for (int i = 0; i < ITER; ++i) { asm volatile(
"v_pk_fma_f32 v[40:41], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[42:43], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[44:45], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[46:47], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[48:49], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[50:51], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[52:53], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[54:55], v[12:13], v[14:15], v[16:17]\n"
"v_pk_mul_f32 v[72:73], v[12:13], v[14:15]\n"
"v_pk_mul_f32 v[74:75], v[12:13], v[14:15]\n"
"v_pk_mul_f32 v[76:77], v[12:13], v[14:15]\n"
"v_pk_mul_f32 v[78:79], v[12:13], v[14:15]\n"
"v_pk_mul_f32 v[80:81], v[12:13], v[14:15]\n"
"v_pk_mul_f32 v[82:83], v[12:13], v[14:15]\n"
"v_pk_mul_f32 v[84:85], v[12:13], v[14:15]\n"
"v_pk_mul_f32 v[86:87], v[12:13], v[14:15]\n"
"v_pk_add_f32 v[88:89], v[12:13], v[14:15]\n"
"v_pk_add_f32 v[90:91], v[12:13], v[14:15]\n"
"v_pk_add_f32 v[92:93], v[12:13], v[14:15]\n"
"v_pk_add_f32 v[94:95], v[12:13], v[14:15]\n"
"v_pk_add_f32 v[96:97], v[12:13], v[14:15]\n"
"v_pk_add_f32 v[98:99], v[12:13], v[14:15]\n"
"v_pk_add_f32 v[100:101], v[12:13], v[14:15]\n"
"v_pk_add_f32 v[102:103], v[12:13], v[14:15]\n"
"v_pk_fma_f32 v[104:105], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[106:107], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[108:109], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[110:111], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[112:113], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[114:115], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[116:117], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[118:119], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[104:105], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[106:107], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[108:109], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[110:111], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[112:113], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[114:115], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[116:117], v[12:13], v[14:15], v[16:17]\n"
"v_pk_fma_f32 v[118:119], v[12:13], v[14:15], v[16:17]\n"
 : "+v"(a), "+v"(b) : : "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119"); }
  out[blockIdx.x * 256 + threadIdx.x] = a + b;
}

int main() {
  float* out; CK(hipMalloc(&out, 256 * 4096 * 4));
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount; const double clk = p.clockRate * 1e3;
  printf("CUs %d clock %.0f MHz\n", cus, clk / 1e6);
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
#define RUNR(k, ninst, reps)                                                                   \
  for (int w : {1, 2, 4}) {                                                             \
    float best = 1e9;                                                                   \
    for (int it = 0; it < 4; ++it) {                                                    \
      CK(hipEventRecord(a)); hipLaunchKernelGGL(k, dim3(cus * w), dim3(256), 0, 0, out); \
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (it) best = std::min(best, ms); } \
    printf("%-12s waves/SIMD %d  %7.2f cycles per instruction-slot per SIMD\n", #k, w, best * 1e-3 * clk / ((double)ITER * (reps) * (ninst) * w)); }
#define RUN(k, n) RUNR(k, n, 64)
  RUNR(k_tile, 1, 1) RUNR(k_tile_nomfma, 1, 1) RUNR(k_btile, 1, 1) RUNR(k_btile_nomfma, 1, 1) RUNR(k_btile_noexp, 1, 1) RUNR(k_btile_pkonly, 1, 1)
  RUN(k_fma, 1) RUN(k_pkfma, 1) RUN(k_pkadd, 1) RUN(k_exp, 1) RUN(k_cvtpk, 1) RUN(k_dot2c, 1) RUN(k_and, 1) RUN(k_lshl, 1) RUN(k_perm, 1)
  RUN(k_mfma, 1) RUN(k_mfma_ind, 1) RUN(k_mfma32, 1) RUN(k_mix, 5) RUN(k_fmac, 1) RUN(k_mul, 1) RUN(k_add, 1) RUN(k_fma3, 1) RUN(k_mix_pk, 1) RUN(k_mix_fmac8, 1) RUN(k_mix_exp4, 1) RUN(k_expfma, 2)
  return 0;
}

// What would ONE persistent launch per iteration cost at its kernel boundaries?  A grid-wide barrier (all resident blocks: arrive
// on a device counter behind a device-scope release, spin until the generation flips, acquire) against the floor of an empty
// launch on this runtime.   hipcc --offload-arch=gfx950 -O3 -o gridsync_lab tools/gridsync_lab.hip && ./gridsync_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ void grid_barrier(unsigned* count, volatile unsigned* gen, unsigned nblk) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned g = *gen;
    __threadfence();                                   // release: this block's writes visible device-wide (L2 write-back per XCD)
    if (atomicAdd(count, 1u) == nblk - 1) { *count = 0; __threadfence(); atomicAdd((unsigned*)gen, 1u); }
    else while (*gen == g) __builtin_amdgcn_s_sleep(1);
    __threadfence();                                   // acquire
  }
  __syncthreads();
}

// `work` floats written per thread between barriers (0: the bare barrier; > 0: dirty lines the release has to write back)
__global__ void __launch_bounds__(256) k_barriers(unsigned* count, unsigned* gen, int nbar, float* buf, int work) {
  for (int b = 0; b < nbar; ++b) {
    for (int w = 0; w < work; ++w) buf[((size_t)blockIdx.x * 256 + threadIdx.x) * work + w] = (float)(b + w);
    grid_barrier(count, gen, gridDim.x);
  }
}
__global__ void k_empty(float* p) { if (p == nullptr) __builtin_trap(); }

int main() {
  unsigned *count, *gen; float* buf;
  CK(hipMalloc(&count, 4)); CK(hipMalloc(&gen, 4)); CK(hipMalloc(&buf, (size_t)1024 * 256 * 64 * 4));
  CK(hipMemset(count, 0, 4)); CK(hipMemset(gen, 0, 4));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int nbar = 200;
  for (int work : {0, 4, 64})
    for (int nblk : {256, 512, 1024}) {
      hipLaunchKernelGGL(k_barriers, dim3(nblk), dim3(256), 0, 0, count, gen, 10, buf, work);   // warm-up
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(a));
      hipLaunchKernelGGL(k_barriers, dim3(nblk), dim3(256), 0, 0, count, gen, nbar, buf, work);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
      printf("grid barrier, %4d blocks x 256 threads, %2d floats written per thread between barriers: %6.2f us per barrier\n", nblk, work, ms * 1e3 / nbar);
    }
  // the alternative: kernel boundaries.  Back-to-back empty launches on one stream (what a dependent small kernel costs at least)
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(a));
    for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, 0, buf);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    if (rep) printf("empty kernel, 256 blocks, back to back on one stream: %6.2f us per launch (GPU-side, host enqueues ahead)\n", ms * 1e3 / 2000);
  }
  return 0;
}

// Where do the four waves of a 256-thread block land?  One record per wave: HW_ID (wave slot, SIMD, CU, SE) and XCC_ID.
//   hipcc --offload-arch=gfx950 -O3 -o simd_lab tools/simd_lab.hip && ./simd_lab [blocks] [lds_bytes]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void __launch_bounds__(256) k_where(unsigned* out, int spin) {
  extern __shared__ float lds[];
  const int wv = threadIdx.x >> 6;
  float a = threadIdx.x;
  for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;   // stay resident long enough for the whole grid to be placed
  if ((threadIdx.x & 63) == 0) {
    out[(blockIdx.x * 4 + wv) * 2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    out[(blockIdx.x * 4 + wv) * 2 + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 20) | (a == 12345.f ? 1u << 31 : 0u);
  }
  if (a == 3.f) lds[0] = a;
}
int main(int argc, char** argv) {
  const int nblk = argc > 1 ? atoi(argv[1]) : 391, ldsb = argc > 2 ? atoi(argv[2]) : 32768;
  unsigned* d; CK(hipMalloc(&d, (size_t)nblk * 8 * 4));
  hipLaunchKernelGGL(k_where, dim3(nblk), dim3(256), ldsb, 0, d, 20000);
  CK(hipDeviceSynchronize());
  std::vector<unsigned> h((size_t)nblk * 8);
  CK(hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost));
  std::map<int, int> hist;   // distinct SIMDs per block
  std::map<long, int> per_simd;
  for (int b = 0; b < nblk; ++b) {
    int mask = 0;
    for (int w = 0; w < 4; ++w) {
      const unsigned hw = h[(b * 4 + w) * 2], xcc = h[(b * 4 + w) * 2 + 1] & 0xF;
      const int simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
      mask |= 1 << simd;
      per_simd[(((long)xcc * 8 + se) * 2 + sh) * 64 + cu * 4 + simd]++;
    }
    hist[__builtin_popcount(mask)]++;
  }
  for (auto& kv : hist) printf("blocks whose 4 waves sit on %d distinct SIMDs: %d\n", kv.first, kv.second);
  std::map<int, int> load;
  for (auto& kv : per_simd) load[kv.second]++;
  for (auto& kv : load) printf("SIMDs holding %d waves: %d\n", kv.first, kv.second);
  printf("SIMDs in use: %zu of 1024\n", per_simd.size());
  return 0;
}

// What does ds_read_b64_tr_b8 return?  (gfx950; the guide documents the 16-bit form only.)  Every lane supplies an 8-byte-aligned
// LDS address; the LDS is filled so that each byte identifies its own address; the output shows which address every byte of every
// lane's result came from.  Not product code.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/trb8_lab.bin tools/trb8_lab.hip && tools/trb8_lab.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__global__ void probe(const int* addr, unsigned long long* out, int hi) {
  __shared__ unsigned char lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = hi ? (unsigned char)(i >> 8) : (unsigned char)(i & 0xFF);
  __syncthreads();
  const unsigned a = (unsigned)(size_t)lds + addr[threadIdx.x];
  unsigned long long v;
  asm volatile("ds_read_b64_tr_b8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
  out[threadIdx.x] = v;
}
int main() {
  int h_addr[64]; int* d_addr; unsigned long long *d_out, lo[64], hi[64];
  CK(hipMalloc(&d_addr, 256)); CK(hipMalloc(&d_out, 512));
  for (int scheme = 0; scheme < 3; ++scheme) {
    for (int l = 0; l < 64; ++l) {
      const int g = l >> 4, i = l & 15;
      if (scheme == 0) h_addr[l] = 1024 * g + 64 * (i >> 1) + 8 * (i & 1);        // lane 2q+p: row q (64-B rows), bytes 8p..8p+7
      else if (scheme == 1) h_addr[l] = 1024 * g + 64 * i;                        // lane i: row i, bytes 0..7
      else h_addr[l] = 1024 * g + 64 * (i & 7) + 8 * (i >> 3);                    // lane q + 8p: row q, bytes 8p..
    }
    CK(hipMemcpy(d_addr, h_addr, 256, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_addr, d_out, 0); CK(hipMemcpy(lo, d_out, 512, hipMemcpyDeviceToHost));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_addr, d_out, 1); CK(hipMemcpy(hi, d_out, 512, hipMemcpyDeviceToHost));
    printf("scheme %d (addresses relative to the group's 1024-byte region; row = addr / 64, col = addr %% 64)\n", scheme);
    for (int l = 0; l < 64; ++l) {
      if ((l & 15) == 0) printf(" group %d\n", l >> 4);
      printf("  lane %2d (gave r%2d c%2d):", l, (h_addr[l] % 1024) / 64, h_addr[l] % 64);
      for (int b = 0; b < 8; ++b) {
        const int a = (int)((lo[l] >> (8 * b)) & 0xFF) | ((int)((hi[l] >> (8 * b)) & 0xFF) << 8);
        printf(" [g%d r%2d c%2d]", a / 1024, (a % 1024) / 64, a % 64);
      }
      printf("\n");
      if (l == 15 && scheme > 0) break;
    }
  }
  return 0;
}

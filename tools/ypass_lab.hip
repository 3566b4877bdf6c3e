// Y-stream lab: how fast can the u8 count matrix be read under different work mappings? Not product code.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
// v0: strip mapping of k_ypass (wave = TR rows x 1 KB), loads only, U rows in flight
template <int U>
__global__ void __launch_bounds__(256) rd_strip(const uint8_t* Y, unsigned* out, long N, int Gp, int nseg, int nrb, int TR) {
  const int lane = threadIdx.x & 63; const long task = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long rb = task / nseg; const int sg = (int)(task - rb * nseg); if (rb >= nrb) return;
  const uint8_t* base = Y + sg * 1024 + lane * 16; const long r0 = rb * TR, r1 = std::min(r0 + TR, N);
  unsigned acc = 0;
  for (long rr = r0; rr < r1; rr += U) {
    uint4 raw[U];
#pragma unroll
    for (int u = 0; u < U; ++u) raw[u] = *reinterpret_cast<const uint4*>(base + std::min(rr + u, r1 - 1) * Gp);
#pragma unroll
    for (int u = 0; u < U; ++u) acc += raw[u].x ^ raw[u].y ^ raw[u].z ^ raw[u].w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}
// v1: flat streaming read (each block walks a contiguous span)
__global__ void __launch_bounds__(256) rd_flat(const uint8_t* Y, unsigned* out, long bytes) {
  const uint4* p = reinterpret_cast<const uint4*>(Y); const long n16 = bytes / 16;
  unsigned acc = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) { uint4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) out[0] = acc;
}
// v2: wave owns TR full rows (all segments): contiguous 5 KB per row
template <int U>
__global__ void __launch_bounds__(256) rd_rows(const uint8_t* Y, unsigned* out, long N, int Gp, int nseg, int TR) {
  const int lane = threadIdx.x & 63; const long rb = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long r0 = rb * TR, r1 = std::min(r0 + TR, N); if (r0 >= N) return;
  unsigned acc = 0;
  for (long r = r0; r < r1; ++r) {
    const uint8_t* row = Y + r * Gp + lane * 16;
    for (int s = 0; s < nseg; s += U) {
      uint4 raw[U];
#pragma unroll
      for (int u = 0; u < U; ++u) raw[u] = *reinterpret_cast<const uint4*>(row + std::min(s + u, nseg - 1) * 1024);
#pragma unroll
      for (int u = 0; u < U; ++u) acc += raw[u].x ^ raw[u].y ^ raw[u].z ^ raw[u].w;
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}
int main() {
  const long N = 100000; const int Gp = 5120, nseg = 5; const long bytes = N * Gp;
  uint8_t* Y; unsigned* out; CK(hipMalloc(&Y, bytes)); CK(hipMalloc(&out, 64)); CK(hipMemset(Y, 1, bytes));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto time = [&](const char* name, auto launch) {
    float best = 1e9; for (int it = 0; it < 6; ++it) { CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipGetLastError());
      float ms; CK(hipEventElapsedTime(&ms, a, b)); if (it) best = std::min(best, ms); }
    printf("%-34s %7.1f us  %6.2f TB/s\n", name, best * 1e3, bytes / best / 1e9);
  };
  for (int TR : {32, 64, 128, 256}) {
    const int nrb = (N + TR - 1) / TR; const long tasks = (long)nrb * nseg; char nm[64];
    snprintf(nm, 64, "strip TR=%d U=4", TR); time(nm, [&] { hipLaunchKernelGGL(rd_strip<4>, dim3((tasks + 3) / 4), dim3(256), 0, 0, Y, out, N, Gp, nseg, nrb, TR); });
    snprintf(nm, 64, "strip TR=%d U=8", TR); time(nm, [&] { hipLaunchKernelGGL(rd_strip<8>, dim3((tasks + 3) / 4), dim3(256), 0, 0, Y, out, N, Gp, nseg, nrb, TR); });
  }
  for (int blocks : {1024, 2048, 4096, 8192}) { char nm[64]; snprintf(nm, 64, "flat blocks=%d", blocks); time(nm, [&] { hipLaunchKernelGGL(rd_flat, dim3(blocks), dim3(256), 0, 0, Y, out, bytes); }); }
  for (int TR : {8, 16, 32}) { const long nw = (N + TR - 1) / TR; char nm[64];
    snprintf(nm, 64, "rows TR=%d U=5", TR); time(nm, [&] { hipLaunchKernelGGL(rd_rows<5>, dim3((nw + 3) / 4), dim3(256), 0, 0, Y, out, N, Gp, nseg, TR); }); }
  return 0;
}

// tools/lab/read_bw.hip -- what a kernel that ONLY reads a 500 MB buffer reaches on this device, in the access shapes the count-matrix stream could take:
//   hipcc -O3 --offload-arch=gfx950 -o tools/lab/read_bw.bin tools/lab/read_bw.hip && tools/lab/read_bw.bin
// (lab: the ceiling the stream kernel's 5.5-5.8 TB/s is to be read against; MI355X_MICROARCH.md quotes 6.3 TB/s for a float4 COPY)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld_nt(const uint4* p) { const v4u v = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(p)); return (uint4){v.x, v.y, v.z, v.w}; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// each wave reads `chunk` bytes contiguous (16 B per lane per load, U loads in flight), waves of a block adjacent, blocks adjacent: the plain streaming shape
template <int U, bool NT>
__global__ void __launch_bounds__(256) k_read(const uint4* __restrict__ p, size_t n16, unsigned* __restrict__ out) {
  const size_t stride = (size_t)gridDim.x * 256;
  uint4 acc = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride * U) {
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t j = i + u * stride;
      v[u] = j < n16 ? (NT ? ld_nt(p + j) : p[j]) : (uint4){0, 0, 0, 0};
    }
#pragma unroll
    for (int u = 0; u < U; ++u) { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}
// the stream's shape: a wave owns a strip and walks it in 4-KiB pieces (4 loads of 1 KiB), DEPTH pieces in flight; pieces of a wave are `jump` bytes apart
template <int DEPTH>
__global__ void __launch_bounds__(256) k_read_pieces(const uint4* __restrict__ p, size_t npieces, int pieces_per_wave, unsigned* __restrict__ out) {
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const size_t p0 = wave * pieces_per_wave;
  uint4 acc = {0, 0, 0, 0};
  for (int k = 0; k < pieces_per_wave; k += DEPTH) {
    uint4 v[DEPTH][4];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const size_t pc = p0 + k + d;
        v[d][i] = pc < npieces ? ld_nt(p + pc * 256 + i * 64 + lane) : (uint4){0, 0, 0, 0};
      }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
      for (int i = 0; i < 4; ++i) { acc.x ^= v[d][i].x; acc.y ^= v[d][i].y; acc.z ^= v[d][i].z; acc.w ^= v[d][i].w; }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}
// the engine's layout and walk: [cell step][gene block of 64] pieces; a block = one segment of 8 gene blocks x four strips of `steps` cell steps (one per wave);
// a wave reads the segment's 8 pieces of a cell step (32 KiB contiguous), then jumps a whole row of gene blocks (gb pieces) to the next cell step
template <int DEPTH>
__global__ void __launch_bounds__(256) k_read_engine(const uint4* __restrict__ p, int nsteps_total, int gb, int steps, unsigned* __restrict__ out) {
  const int nseg = gb / 8, rg = blockIdx.x / nseg, seg = blockIdx.x % nseg, wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int st0 = (rg * 4 + wv) * steps;
  uint4 acc = {0, 0, 0, 0};
  for (int k = 0; k < steps * 8; k += DEPTH) {
    uint4 v[DEPTH][4];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int st = st0 + (k + d) / 8, a = (k + d) % 8;
      const size_t pc = (size_t)st * gb + seg * 8 + a;
#pragma unroll
      for (int i = 0; i < 4; ++i) v[d][i] = st < nsteps_total ? ld_nt(p + pc * 256 + i * 64 + lane) : (uint4){0, 0, 0, 0};
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
      for (int i = 0; i < 4; ++i) { acc.x ^= v[d][i].x; acc.y ^= v[d][i].y; acc.z ^= v[d][i].z; acc.w ^= v[d][i].w; }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}
int main() {
  const size_t bytes = 500ull << 20;
  uint4* buf; unsigned* out;
  CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 4));
  CK(hipMemset(buf, 1, bytes));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto timeit = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    (void)hipDeviceSynchronize();
    float best = 1e9f, tot = 0.f;
    for (int r = 0; r < 10; ++r) { (void)hipEventRecord(a); launch(); (void)hipEventRecord(b); (void)hipEventSynchronize(b); float ms; (void)hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best; tot += ms; }
    printf("%-58s best %.1f us = %.2f TB/s, mean %.1f us\n", name, best * 1e3, bytes / (best * 1e-3) / 1e12, tot / 10 * 1e3);
  };
  const size_t n16 = bytes / 16;
  for (int blocks : {1024, 2048, 4096, 8192}) {
    char nm[128];
    snprintf(nm, sizeof nm, "grid-stride, 4 x 16 B in flight, %d blocks, nt", blocks);
    timeit(nm, [&] { hipLaunchKernelGGL((k_read<4, true>), dim3(blocks), dim3(256), 0, 0, buf, n16, out); });
    snprintf(nm, sizeof nm, "grid-stride, 8 x 16 B in flight, %d blocks, nt", blocks);
    timeit(nm, [&] { hipLaunchKernelGGL((k_read<8, true>), dim3(blocks), dim3(256), 0, 0, buf, n16, out); });
  }
  timeit("grid-stride, 8 x 16 B in flight, 4096 blocks, default policy", [&] { hipLaunchKernelGGL((k_read<8, false>), dim3(4096), dim3(256), 0, 0, buf, n16, out); });
  const size_t npieces = bytes / 4096;
  for (int ppw : {8, 16, 32, 64}) {
    const int waves = (int)((npieces + ppw - 1) / ppw), blocks = (waves + 3) / 4;
    char nm[128];
    snprintf(nm, sizeof nm, "wave strips of %d 4-KiB pieces, 2 in flight, %d blocks", ppw, blocks);
    timeit(nm, [&] { hipLaunchKernelGGL((k_read_pieces<2>), dim3(blocks), dim3(256), 0, 0, buf, npieces, ppw, out); });
    snprintf(nm, sizeof nm, "wave strips of %d 4-KiB pieces, 4 in flight, %d blocks", ppw, blocks);
    timeit(nm, [&] { hipLaunchKernelGGL((k_read_pieces<4>), dim3(blocks), dim3(256), 0, 0, buf, npieces, ppw, out); });
  }
  {   // 100k cells x 5120 genes at one byte: 1563 cell steps x 80 gene blocks = 125 040 pieces = 512 MB
    const int gb = 80, nst = 1563;
    uint4* big; CK(hipMalloc(&big, (size_t)nst * gb * 4096)); CK(hipMemset(big, 1, (size_t)nst * gb * 4096));
    const size_t eb = (size_t)nst * gb * 4096;
    for (int steps : {1, 2, 4, 8}) {
      const int nrg = (nst + 4 * steps - 1) / (4 * steps), blocks = nrg * (gb / 8);
      for (int dep = 2; dep <= 4; dep += 2) {
        for (int i = 0; i < 3; ++i) { if (dep == 2) hipLaunchKernelGGL((k_read_engine<2>), dim3(blocks), dim3(256), 0, 0, big, nst, gb, steps, out); else hipLaunchKernelGGL((k_read_engine<4>), dim3(blocks), dim3(256), 0, 0, big, nst, gb, steps, out); }
        (void)hipDeviceSynchronize();
        float best = 1e9f;
        for (int r = 0; r < 10; ++r) {
          (void)hipEventRecord(a);
          if (dep == 2) hipLaunchKernelGGL((k_read_engine<2>), dim3(blocks), dim3(256), 0, 0, big, nst, gb, steps, out); else hipLaunchKernelGGL((k_read_engine<4>), dim3(blocks), dim3(256), 0, 0, big, nst, gb, steps, out);
          (void)hipEventRecord(b); (void)hipEventSynchronize(b); float ms; (void)hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best;
        }
        printf("engine layout, %d cell steps per wave (%d blocks), %d pieces in flight: best %.1f us = %.2f TB/s\n", steps, blocks, dep, best * 1e3, eb / (best * 1e-3) / 1e12);
      }
    }
  }
  {   // the same reader held to the stream's occupancy: dynamic LDS sized so that only `bpc` four-wave blocks fit a CU (160 KB)
    const int gb = 80, nst = 1563, steps = 4;
    uint4* big; CK(hipMalloc(&big, (size_t)nst * gb * 4096)); CK(hipMemset(big, 1, (size_t)nst * gb * 4096));
    const size_t eb = (size_t)nst * gb * 4096;
    const int nrg = (nst + 4 * steps - 1) / (4 * steps), blocks = nrg * (gb / 8);
    for (int bpc : {1, 2, 3, 4, 6}) {
      const size_t lds = (size_t)(160 * 1024 / bpc) - 1024;
      CK(hipFuncSetAttribute((const void*)k_read_engine<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      CK(hipFuncSetAttribute((const void*)k_read_engine<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      for (int dep = 2; dep <= 4; dep += 2) {
        auto go = [&] { if (dep == 2) hipLaunchKernelGGL((k_read_engine<2>), dim3(blocks), dim3(256), lds, 0, big, nst, gb, steps, out); else hipLaunchKernelGGL((k_read_engine<4>), dim3(blocks), dim3(256), lds, 0, big, nst, gb, steps, out); };
        for (int i = 0; i < 3; ++i) go();
        (void)hipDeviceSynchronize();
        float best = 1e9f;
        for (int r = 0; r < 10; ++r) { (void)hipEventRecord(a); go(); (void)hipEventRecord(b); (void)hipEventSynchronize(b); float ms; (void)hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best; }
        printf("engine layout, %d blocks per CU (%d waves per SIMD), %d pieces in flight per wave: best %.1f us = %.2f TB/s\n", bpc, bpc, dep, best * 1e3, eb / (best * 1e-3) / 1e12);
      }
    }
  }
  return 0;
}

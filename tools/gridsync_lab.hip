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

// XCD-hierarchical form (MI355X_MICROARCH.md row `barrier-xcd`; VERDICT r2 #4 asked for THIS one to be measured, not the flat counter):
// blocks arrive on their own XCC's counter (relaxed, agent scope); the last arriver of an XCC -- its leader for this round -- does the
// one release fence of the XCC, arrives on the top counter, and the last leader flips the generation word every block polls.
// Every waiter does an agent-scope acquire fence after the flip.  State: cnt[8] (one 64-byte line each), top, gen.
struct xbar { unsigned cnt[8 * 16]; unsigned top; unsigned pad0[15]; unsigned gen; unsigned pad1[15]; unsigned per_xcc[8]; };
__device__ __forceinline__ void grid_barrier_xcd(xbar* b, unsigned n_xcc_present) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned x = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 7u;     // HW_REG_XCC_ID
    const unsigned g = __hip_atomic_load(&b->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned mine = b->per_xcc[x];
    if (__hip_atomic_fetch_add(&b->cnt[16 * x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == mine - 1) {
      __hip_atomic_store(&b->cnt[16 * x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");                   // the XCC's one L2 write-back
      if (__hip_atomic_fetch_add(&b->top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n_xcc_present - 1) {
        __hip_atomic_store(&b->top, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&b->gen, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    while (__hip_atomic_load(&b->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == g) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}
// (a block's non-leader writes reach L2 through the write-through / the leader's fence only if they were made with agent scope or the
//  XCC's L2 is written back by the leader: plain stores of other CUs of the same XCC sit in that same L2, which the leader's release
//  writes back -- that is the point of the hierarchy)
__global__ void __launch_bounds__(256) k_count_xcc(xbar* b) {
  if (threadIdx.x == 0) atomicAdd(&b->per_xcc[__builtin_amdgcn_s_getreg((31 << 11) | 20) & 7u], 1u);
}
__global__ void __launch_bounds__(256) k_barriers_xcd(xbar* b, unsigned n_xcc_present, int nbar, float* buf, int work) {
  for (int i = 0; i < nbar; ++i) {
    for (int w = 0; w < work; ++w) buf[((size_t)blockIdx.x * 256 + threadIdx.x) * work + w] = (float)(i + w);
    grid_barrier_xcd(b, n_xcc_present);
  }
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
  // XCD-hierarchical barrier: which XCC a block lands on is the dispatcher's choice (round-robin), counted once per grid size by a
  // probe launch of the same geometry (persistent blocks stay where they start)
  xbar* xb; CK(hipMalloc(&xb, sizeof(xbar)));
  for (int work : {0, 4, 64})
    for (int nblk : {256, 512, 1024}) {
      CK(hipMemset(xb, 0, sizeof(xbar)));
      hipLaunchKernelGGL(k_count_xcc, dim3(nblk), dim3(256), 0, 0, xb);
      CK(hipDeviceSynchronize());
      xbar hx; CK(hipMemcpy(&hx, xb, sizeof(xbar), hipMemcpyDeviceToHost));
      unsigned present = 0; bool even = true;
      for (int x = 0; x < 8; ++x) { present += hx.per_xcc[x] > 0; even = even && hx.per_xcc[x] == (unsigned)nblk / 8; }
      if (!even) { printf("xcd barrier, %4d blocks: the probe launch was not dealt evenly over the XCCs (%u %u %u %u %u %u %u %u): skipped\n", nblk,
                          hx.per_xcc[0], hx.per_xcc[1], hx.per_xcc[2], hx.per_xcc[3], hx.per_xcc[4], hx.per_xcc[5], hx.per_xcc[6], hx.per_xcc[7]); continue; }
      hipLaunchKernelGGL(k_barriers_xcd, dim3(nblk), dim3(256), 0, 0, xb, present, 10, buf, work);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(a));
      hipLaunchKernelGGL(k_barriers_xcd, dim3(nblk), dim3(256), 0, 0, xb, present, nbar, buf, work);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
      printf("XCD-hierarchical barrier, %4d blocks x 256 threads, %2d floats written per thread between barriers: %6.2f us per barrier\n", nblk, work, ms * 1e3 / nbar);
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

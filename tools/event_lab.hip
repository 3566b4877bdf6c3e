// How long does a cross-stream hipStreamWaitEvent hold up the waiting stream on MI355X when the event (a) completed long
// ago, (b) completes right before the wait is reached?  Gaps measured from device timestamps (s_memrealtime, 100 MHz).
//   hipcc -O3 --offload-arch=gfx950 -o tools/event_lab.bin tools/event_lab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void spin(unsigned long long* t, int slot, long ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while ((long)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) {}
  if (threadIdx.x == 0 && blockIdx.x == 0) { t[2 * slot] = t0; t[2 * slot + 1] = __builtin_amdgcn_s_memrealtime(); }
}
int main() {
  hipStream_t a, b; CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  hipEvent_t ev, ev2; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ev2, hipEventDisableTiming));
  unsigned long long* t; CK(hipMalloc(&t, 64 * 8));
  std::vector<unsigned long long> h(64);
  auto run = [&](int mode, const char* name) -> int {
    std::vector<double> gaps;
    for (int rep = 0; rep < 20; ++rep) {
      // stream a: k0 (20 us), [wait], k1 (5 us).  stream b: kb whose end the wait depends on.
      const long kb_ticks = mode == 1 ? 200 : 1950;   // early: ends 18 us before k0; late: ends with k0
      CK(hipEventRecord(ev2, a)); CK(hipStreamWaitEvent(b, ev2, 0));       // start both together
      hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, b, t, 2, kb_ticks);
      CK(hipEventRecord(ev, b));
      hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, a, t, 0, 2000);
      if (mode != 0) CK(hipStreamWaitEvent(a, ev, 0));
      hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, a, t, 1, 500);
      CK(hipStreamSynchronize(a)); CK(hipStreamSynchronize(b));
      CK(hipMemcpy(h.data(), t, 64 * 8, hipMemcpyDeviceToHost));
      if (rep >= 4) gaps.push_back((double)(h[2] - h[1]) / 100.0);
    }
    std::sort(gaps.begin(), gaps.end());
    printf("%-44s gap k0 end -> k1 start: median %.1f us (min %.1f, max %.1f)\n", name, gaps[gaps.size() / 2], gaps.front(), gaps.back());
    return 0;
  };
  run(0, "no wait (same stream back to back)");
  run(1, "wait on an event completed 18 us earlier");
  run(2, "wait on an event completing with k0");
  // The same dependency through stream memory operations (hipStreamWriteValue32 on the producer, hipStreamWaitValue32 on the
  // consumer) instead of an event, and what the producer side pays: the kernel that follows the record / the write.
  int can = 0;
  CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  unsigned* flag = nullptr;
  CK(hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory));
  CK(hipMemset(flag, 0, 8));
  unsigned seq = 0;
  auto run2 = [&](int mode, const char* name) -> int {   // modes: 0/1 = value written early / late; 2/3 = producer side, event / value
    std::vector<double> gaps;
    for (int rep = 0; rep < 20; ++rep) {
      ++seq;
      CK(hipEventRecord(ev2, a)); CK(hipStreamWaitEvent(b, ev2, 0));
      if (mode <= 1) {
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, b, t, 2, mode == 0 ? 200 : 1950);
        CK(hipStreamWriteValue32(b, flag, seq, 0));
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, a, t, 0, 2000);
        CK(hipStreamWaitValue32(a, flag, seq, hipStreamWaitValueGte, 0xFFFFFFFFu));
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, a, t, 1, 500);
      } else {
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, b, t, 0, 1000);
        if (mode == 2) { CK(hipEventRecord(ev, b)); CK(hipStreamWaitEvent(a, ev, 0)); }
        else { CK(hipStreamWriteValue32(b, flag, seq, 0)); CK(hipStreamWaitValue32(a, flag, seq, hipStreamWaitValueGte, 0xFFFFFFFFu)); }
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, b, t, 1, 500);     // the producer's next kernel
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, a, t, 3, 500);     // the consumer
      }
      CK(hipStreamSynchronize(a)); CK(hipStreamSynchronize(b));
      CK(hipMemcpy(h.data(), t, 64 * 8, hipMemcpyDeviceToHost));
      if (rep >= 4) gaps.push_back((double)(h[2] - h[1]) / 100.0);
    }
    std::sort(gaps.begin(), gaps.end());
    printf("%-60s median %.1f us (min %.1f, max %.1f)\n", name, gaps[gaps.size() / 2], gaps.front(), gaps.back());
    return 0;
  };
  if (can) {
    run2(0, "wait on a VALUE written 18 us earlier: k0 end -> k1 start");
    run2(1, "wait on a VALUE written as k0 ends: k0 end -> k1 start");
  }
  run2(2, "producer side, event record + cross-stream wait: kb end -> next");
  if (can) run2(3, "producer side, value write + cross-stream value wait: kb end -> next");
  return 0;
}

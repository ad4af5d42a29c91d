// What does a fork / join between two HIP streams cost inside a dependent chain of short kernels?
//   hipcc --offload-arch=gfx950 -O2 -o event_dep_probe event_dep_probe.hip && ./event_dep_probe
// Chain per iteration on the main stream: A (spin 20 us, 1024 workgroups) -> B (spin 100 us) -> next A.  A side kernel (spin 10 us, 2
// workgroups) depends on A and must be finished before the next A.  Variants:
//   serial : side kernel on the main stream between A and B
//   fork   : record(main) -> wait(side stream) -> side kernel -> record(side) ... B on main ... wait(main) before the next A
// for event flags DisableTiming, | DisableSystemFence, | ReleaseToDevice.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void spin_kernel(long long ticks, int* sink) {   // wall_clock64: 100 MHz
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (sink && threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(sink, 1);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
  hipStream_t m, s;
  CK(hipStreamCreateWithFlags(&m, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  int* sink; CK(hipMalloc(&sink, 4));
  const int N = 200;
  const unsigned flagsets[3] = {hipEventDisableTiming, hipEventDisableTiming | hipEventDisableSystemFence, hipEventDisableTiming | hipEventReleaseToDevice};
  const char* names[3] = {"DisableTiming", "DisableTiming|DisableSystemFence", "DisableTiming|ReleaseToDevice"};
  auto run = [&](int mode, unsigned flags, double* us) -> int {
    std::vector<hipEvent_t> e1(N), e2(N);
    if (mode) for (int i = 0; i < N; ++i) { CK(hipEventCreateWithFlags(&e1[i], flags)); CK(hipEventCreateWithFlags(&e2[i], flags)); }
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipStreamSynchronize(m)); CK(hipStreamSynchronize(s));
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < N; ++i) {
        spin_kernel<<<1024, 64, 0, m>>>(2000, sink);
        if (mode == 0) {
          spin_kernel<<<2, 64, 0, m>>>(1000, sink);
          spin_kernel<<<1020, 64, 0, m>>>(10000, sink);
        } else if (mode == 1) {
          CK(hipEventRecord(e1[i], m)); CK(hipStreamWaitEvent(s, e1[i], 0));
          spin_kernel<<<2, 64, 0, s>>>(1000, sink);
          CK(hipEventRecord(e2[i], s));
          spin_kernel<<<1020, 64, 0, m>>>(10000, sink);
          CK(hipStreamWaitEvent(m, e2[i], 0));
        } else {
          spin_kernel<<<1020, 64, 0, m>>>(10000, sink);   // no side kernel at all: the floor
        }
      }
      CK(hipStreamSynchronize(m)); CK(hipStreamSynchronize(s));
      *us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
    }
    if (mode) for (int i = 0; i < N; ++i) { (void)hipEventDestroy(e1[i]); (void)hipEventDestroy(e2[i]); }
    return 0;
  };
  double us;
  if (run(2, 0, &us)) return 1;
  printf("A(20us) + B(100us), no side kernel                      %7.1f us per iteration\n", us);
  if (run(0, 0, &us)) return 1;
  printf("side kernel (10 us) in series on the main stream          %7.1f us per iteration\n", us);
  for (int f = 0; f < 3; ++f) {
    if (run(1, flagsets[f], &us)) return 1;
    printf("side kernel on a second stream, events %-34s %7.1f us per iteration\n", names[f], us);
  }
  return 0;
}

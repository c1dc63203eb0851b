// Host-side cost of the HIP calls the step is made of (enqueue only, queue kept shallow): tools/launchcost.hip -> tools/bin/launchcost
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
struct Big { long a[24]; };
__global__ void k_small(int* p) { if (p && threadIdx.x == 999) *p = 1; }
__global__ void k_big(Big b, int* p) { if (p && threadIdx.x == 999) *p = (int)b.a[3]; }
template <class F> static double per_call(F f, int n, hipStream_t s0, hipStream_t s1) {
  hipDeviceSynchronize();
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; ++i) { f(); if ((i & 63) == 63) { hipStreamSynchronize(s0); hipStreamSynchronize(s1); } }
  auto t1 = std::chrono::steady_clock::now();
  hipDeviceSynchronize();
  return std::chrono::duration<double, std::micro>(t1 - t0).count() / n;
}
int main() {
  hipStream_t s0, s1;
  hipStreamCreateWithPriority(&s0, hipStreamNonBlocking, -1);
  hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, 0);
  hipEvent_t ev, ev2;
  hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  hipEventCreateWithFlags(&ev2, hipEventDisableTiming);
  int* d; hipMalloc(&d, 1 << 24);
  Big b{};
  const int n = 2048;
  // the sync every 64 calls is inside every measurement alike: measure it first
  double base = per_call([&] {}, n, s0, s1);
  printf("loop + sync/64                  %6.2f us\n", base);
  printf("hipLaunchKernelGGL small        %6.2f us\n", per_call([&] { hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s0, d); }, n, s0, s1) - base);
  printf("hipLaunchKernelGGL 192 B args   %6.2f us\n", per_call([&] { hipLaunchKernelGGL(k_big, dim3(1), dim3(64), 0, s0, b, d); }, n, s0, s1) - base);
  printf("hipExtLaunch + stop event       %6.2f us\n", per_call([&] { hipExtLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s0, nullptr, ev, 0, d); }, n, s0, s1) - base);
  printf("launch + hipEventRecord         %6.2f us\n", per_call([&] { hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s0, d); hipEventRecord(ev, s0); }, n, s0, s1) - base);
  printf("hipEventRecord                  %6.2f us\n", per_call([&] { hipEventRecord(ev, s0); }, n, s0, s1) - base);
  printf("record + wait on other stream   %6.2f us\n", per_call([&] { hipEventRecord(ev, s0); hipStreamWaitEvent(s1, ev, 0); }, n, s0, s1) - base);
  printf("launch s0, record, wait s1, launch s1 %6.2f us\n", per_call([&] { hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s0, d); hipEventRecord(ev, s0); hipStreamWaitEvent(s1, ev, 0); hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s1, d); }, n, s0, s1) - base);
  printf("hipMemsetAsync 4 KB             %6.2f us\n", per_call([&] { hipMemsetAsync(d, 0, 4096, s0); }, n, s0, s1) - base);
  printf("hipMemsetAsync 4 MB             %6.2f us\n", per_call([&] { hipMemsetAsync(d, 0, 4 << 20, s0); }, n, s0, s1) - base);
  printf("hipMemsetAsync 4 MB + 1 B tail  %6.2f us\n", per_call([&] { hipMemsetAsync(d, 0, (4 << 20) + 1, s0); }, n, s0, s1) - base);
  printf("hipMemcpyAsync D2D 4 KB         %6.2f us\n", per_call([&] { hipMemcpyAsync(d, d + 4096, 4096, hipMemcpyDeviceToDevice, s0); }, n, s0, s1) - base);
  printf("hipStreamQuery                  %6.2f us\n", per_call([&] { hipStreamQuery(s0); }, n, s0, s1) - base);
  printf("hipGetLastError                 %6.2f us\n", per_call([&] { (void)hipGetLastError(); }, n, s0, s1) - base);
  // alternating streams (the backward sweep alternates main / side)
  printf("launch alternating s0 / s1      %6.2f us\n", per_call([&] { hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s0, d); hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s1, d); }, n, s0, s1) / 2 - base / 2);
  return 0;
}

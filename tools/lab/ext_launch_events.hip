// Lab: what timing a kernel through hipExtLaunchKernelGGL's start / stop events costs against hipEventRecord between launches.
// Five small kernels per "call", 2000 calls; wall clock per call and the elapsed times the two ways give.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
__global__ void spin(unsigned *p, int n) {
  unsigned a = threadIdx.x;
  for (int i = 0; i < n; i++) a = a * 1664525u + 1013904223u;
  if (a == 12345u) p[0] = a;
}
int main() {
  unsigned *d; hipMalloc(&d, 4);
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t ev[6];
  for (auto &e : ev) hipEventCreateWithFlags(&e, hipEventDisableSystemFence);
  const int N = 2000, work[5] = {1200, 600, 150, 250, 150};
  auto wall = [&](int mode) {
    auto t0 = std::chrono::steady_clock::now();
    for (int c = 0; c < N; c++) {
      if (mode == 1) hipEventRecord(ev[0], s);
      for (int k = 0; k < 5; k++) {
        if (mode == 2)
          hipExtLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s, k == 0 ? ev[0] : nullptr, ev[k + 1], 0, d, work[k]);
        else
          hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s, d, work[k]);
        if (mode == 1) hipEventRecord(ev[k + 1], s);
      }
      hipStreamSynchronize(s);
    }
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
  };
  for (int mode = 0; mode < 3; mode++) {
    wall(mode);
    const double us = wall(mode);
    float tot = 0, first = 0;
    if (mode) { hipEventElapsedTime(&tot, ev[0], ev[5]); hipEventElapsedTime(&first, ev[0], ev[1]); }
    printf("%-44s %7.1f us per call   first kernel %.1f us, all five %.1f us\n",
           mode == 0 ? "no events" : mode == 1 ? "hipEventRecord between the launches" : "hipExtLaunchKernelGGL start / stop events", us, first * 1e3, tot * 1e3);
  }
  return 0;
}

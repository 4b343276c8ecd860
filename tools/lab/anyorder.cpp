// tools/lab/anyorder.cpp: does hipExtAnyOrderLaunch (AQL barrier bit cleared) let two kernels of ONE stream
// overlap on this part?  Two 64-workgroup spin kernels (each far from filling the chip) back to back.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
__global__ void spin(unsigned long long ticks, int *sink) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {}
  if (sink && threadIdx.x == 12345) *sink = 1;
}
int main() {
  hipStream_t s;
  hipStreamCreate(&s);
  int *sink;
  hipMalloc(&sink, 4);
  for (int flags = 0; flags < 2; ++flags) {
    for (int rep = 0; rep < 3; ++rep) {
      hipStreamSynchronize(s);
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < 20; ++i)
        hipExtLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s, nullptr, nullptr, flags ? hipExtAnyOrderLaunch : 0,
                              10000ull /* 100 us at 100 MHz */, sink);
      hipStreamSynchronize(s);
      double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      printf("flags=%d: 20 x 100us spin kernels on one stream: %.0f us (%s)\n", flags, us, hipGetErrorString(hipGetLastError()));
    }
  }
  return 0;
}

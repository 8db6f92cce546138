// tests/gpuhog/hog.hip -- TEST INFRASTRUCTURE: a kernel that does nothing but occupy compute units (one workgroup per
// unit through its LDS footprint) for a given time, so that a test can run the library next to another process that holds
// part of the device (tests/test_gpu_parity.py::test_one_launch_front_starved_by_another_process).  Bounded by the
// 100 MHz wall clock: it always ends.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k_hog(long long ticks, int *sink) {
  extern __shared__ char lds[];
  lds[threadIdx.x] = (char)threadIdx.x;
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(127);
  if (lds[threadIdx.x] == 77 && sink) sink[0] = 1;                  // (keeps the LDS allocation alive)
}

// occupy `nwg` compute units (lds_bytes of LDS each) of device `dev` for `ms` milliseconds; prints "started" once the
// launch is in the queue and returns when it has ended
extern "C" int hog_run(int dev, int nwg, int lds_bytes, int ms) {
  if (hipSetDevice(dev) != hipSuccess) return 1;
  if (hipFuncSetAttribute((const void *)k_hog, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) return 2;
  hipStream_t st;
  if (hipStreamCreate(&st) != hipSuccess) return 3;
  hipLaunchKernelGGL(k_hog, dim3(nwg), dim3(64), lds_bytes, st, (long long)ms * 100000LL, (int *)nullptr);
  if (hipGetLastError() != hipSuccess) return 4;
  printf("started\n"); fflush(stdout);
  if (hipStreamSynchronize(st) != hipSuccess) return 5;
  printf("ended\n"); fflush(stdout);
  return 0;
}

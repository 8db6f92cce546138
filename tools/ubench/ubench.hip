// micro-benchmarks that calibrate design choices of the solve / factor kernels (not product code)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void k_empty(int *p) { if (p && threadIdx.x == 1000000) *p = 1; }

// one workgroup streams n doubles (L2/HBM -> registers), 16 B per lane per load, U loads in flight
template <int U>
__global__ void __launch_bounds__(1024) k_stream1(const double2 *src, double *out, long n2) {
  double acc = 0;
  long i = threadIdx.x + (long)blockIdx.x * n2;
  const long e = (long)(blockIdx.x + 1) * n2;
  for (; i + (U - 1) * (long)blockDim.x < e; i += U * (long)blockDim.x) {
    double2 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = src[i + u * (long)blockDim.x];
#pragma unroll
    for (int u = 0; u < U; u++) acc += v[u].x + v[u].y;
  }
  if (acc == 12345.678) out[0] = acc;
}

__device__ __forceinline__ double bcast(double v, int lane) {
  union { double d; int i[2]; } u; u.d = v;
  u.i[0] = __builtin_amdgcn_readlane(u.i[0], lane);
  u.i[1] = __builtin_amdgcn_readlane(u.i[1], lane);
  return u.d;
}
// readlane TRSV chain: 64 steps, repeated R times
__global__ void k_trsv(double *y, const double *L, int R) {
  const int tid = threadIdx.x;
  double lr[64];
#pragma unroll
  for (int c = 0; c < 64; c++) lr[c] = c < tid ? L[c * 64 + tid] : 0.0;
  double w = y[tid];
  for (int r = 0; r < R; r++) {
#pragma unroll
    for (int k = 0; k < 64; k++) { const double wk = bcast(w, k); if (tid > k) w -= lr[k] * wk; }
  }
  y[tid] = w;
}
// LDL of a 64x64 block held one row per lane
__global__ void k_ldl64(double *A, int R) {
  const int i = threadIdx.x;
  double x[64];
  for (int r = 0; r < R; r++) {
#pragma unroll
    for (int j = 0; j < 64; j++) x[j] = j <= i ? A[j * 64 + i] : 0.0;
#pragma unroll
    for (int k = 0; k < 64; k++) {
      const double xkk = bcast(x[k], k);
      const double l = x[k] / xkk;
#pragma unroll
      for (int j = k + 1; j < 64; j++) { const double ljk = bcast(l, j); if (i >= j) x[j] -= ljk * x[k]; }
      if (i > k) x[k] = l;
    }
#pragma unroll
    for (int j = 0; j < 64; j++) if (j <= i) A[64 * 64 + j * 64 + i] = x[j];
  }
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float ms;
  // 1. launch boundary
  for (int wg : {1, 64, 1024}) {
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k_empty, dim3(wg), dim3(256), 0, st, (int *)nullptr);
    CK(hipEventRecord(a, st));
    for (int i = 0; i < 1000; i++) hipLaunchKernelGGL(k_empty, dim3(wg), dim3(256), 0, st, (int *)nullptr);
    CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
    printf("empty kernel x1000, %d WGs: %.2f us per launch (eager)\n", wg, ms);
  }
  // graph of 100 empty launches
  {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 100; i++) hipLaunchKernelGGL(k_empty, dim3(64), dim3(256), 0, st, (int *)nullptr);
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 3; i++) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(a, st));
    for (int i = 0; i < 20; i++) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
    printf("graph of 100 empty launches: %.2f us per launch, %.1f us per replay\n", ms * 1000 / 2000, ms * 1000 / 20);
  }
  // 2. single-WG streaming
  const long N = 8 << 20;  // doubles = 64 MB
  double *d, *o; CK(hipMalloc(&d, N * 8)); CK(hipMalloc(&o, 64)); CK(hipMemset(d, 0, N * 8));
  for (long bytes : {1800000L, 16000000L}) {
    long n2 = bytes / 16 / 8192 * 8192;
    for (int thr : {256, 512, 1024}) {
      for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(a, st));
        hipLaunchKernelGGL(k_stream1<8>, dim3(1), dim3(thr), 0, st, (const double2 *)d, o, n2);
        CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
      }
      printf("1 WG x %4d thr U=8 streams %.1f MB: %.1f us = %.1f GB/s\n", thr, n2 * 16 / 1e6, ms * 1e3, n2 * 16 / ms / 1e6);
    }
    for (int rep = 0; rep < 2; rep++) {
      CK(hipEventRecord(a, st));
      hipLaunchKernelGGL(k_stream1<4>, dim3(1), dim3(1024), 0, st, (const double2 *)d, o, n2);
      CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
    }
    printf("1 WG x 1024 thr U=4 streams %.1f MB: %.1f us = %.1f GB/s\n", n2 * 16 / 1e6, ms * 1e3, n2 * 16 / ms / 1e6);
    for (int nwg : {2, 4, 8, 16, 64, 256}) {
      long per = n2 / nwg / 1024 * 1024;
      for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(a, st));
        hipLaunchKernelGGL(k_stream1<4>, dim3(nwg), dim3(1024), 0, st, (const double2 *)d, o, per);
        CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
      }
      printf("%3d WG x 1024 thr U=4 stream %.1f MB total: %.1f us = %.1f GB/s\n", nwg, per * nwg * 16 / 1e6, ms * 1e3, per * nwg * 16 / ms / 1e6);
    }
  }
  // 3. TRSV chain / LDL64
  {
    std::vector<double> h(64 * 64 * 2, 0.0);
    for (int i = 0; i < 64; i++) for (int j = 0; j <= i; j++) h[j * 64 + i] = (i == j) ? 64.0 + i : 0.01 * ((i * 7 + j * 3) % 11);
    double *L, *y; CK(hipMalloc(&L, h.size() * 8)); CK(hipMalloc(&y, 64 * 8));
    CK(hipMemcpy(L, h.data(), h.size() * 8, hipMemcpyHostToDevice)); CK(hipMemset(y, 0, 64 * 8));
    for (int R : {1, 101}) {
      for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(a, st)); hipLaunchKernelGGL(k_trsv, dim3(1), dim3(64), 0, st, y, L, R);
        CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
      }
      printf("trsv64 readlane chain R=%d: %.2f us\n", R, ms * 1e3);
    }
    for (int R : {1, 11}) {
      for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(a, st)); hipLaunchKernelGGL(k_ldl64, dim3(1), dim3(64), 0, st, L, R);
        CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
      }
      printf("ldl64 lane-per-row R=%d: %.2f us\n", R, ms * 1e3);
    }
  }
  return 0;
}

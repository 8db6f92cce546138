// phase timing of the single-front forward sweep (calibration only, not product code)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int SNB = 64;
__device__ __forceinline__ double bcast(double v, int lane) {
  union { double d; int i[2]; } u; u.d = v;
  u.i[0] = __builtin_amdgcn_readlane(u.i[0], lane);
  u.i[1] = __builtin_amdgcn_readlane(u.i[1], lane);
  return u.d;
}
template <int VAR>
__global__ void __launch_bounds__(512) k_fw(const double *Fs, int ns, int ms, double *y, long long *tim) {
  extern __shared__ double w[];
  __shared__ double wb[SNB];
  const int tid = threadIdx.x, bs = blockDim.x;
  for (int i = tid; i < ms; i += bs) w[i] = y[i];
  __syncthreads();
  double lr[SNB];
  long long t0 = wall_clock64(), tt = 0, tg = 0;
  if (tid < 64) {
    const int kb = min(SNB, ns);
#pragma unroll
    for (int c = 0; c < SNB; c++) lr[c] = (c < tid && tid < kb) ? Fs[(int64_t)c * ms + tid] : 0.0;
  }
  for (int k0 = 0; k0 < ns; k0 += SNB) {
    const int kb = min(SNB, ns - k0);
    long long a = wall_clock64();
    if (tid < 64) {
      double wi = tid < kb ? w[k0 + tid] : 0.0;
      if (VAR != 1) {
#pragma unroll
      for (int k = 0; k < SNB; k++) {
        if (k < kb) { const double wk = bcast(wi, k); if (tid > k) wi -= lr[k] * wk; }
      }
      }
      if (tid < kb) { w[k0 + tid] = wi; wb[tid] = wi; }
      const int k1 = k0 + SNB;
      if (k1 < ns && VAR != 2) {
        const int kbn = min(SNB, ns - k1);
#pragma unroll
        for (int c = 0; c < SNB; c++) lr[c] = (c < tid && tid < kbn) ? Fs[(int64_t)(k1 + c) * ms + k1 + tid] : 0.0;
      }
    }
    __syncthreads();
    long long b = wall_clock64();
    if (VAR != 3)
    for (int r = k0 + kb + tid; r < ms; r += bs) {
      const double *col = Fs + (int64_t)k0 * ms + r;
      double acc = 0.0;
      if (kb == SNB) {
#pragma unroll
        for (int c0 = 0; c0 < SNB; c0 += 16) {
          double v[16];
#pragma unroll
          for (int c = 0; c < 16; c++) v[c] = col[(int64_t)(c0 + c) * ms];
#pragma unroll
          for (int c = 0; c < 16; c++) acc += v[c] * wb[c0 + c];
        }
      } else {
        for (int c = 0; c < kb; c++) acc += col[(int64_t)c * ms] * wb[c];
      }
      w[r] -= acc;
    }
    __syncthreads();
    long long c = wall_clock64();
    tt += b - a; tg += c - b;
  }
  for (int i = tid; i < ms; i += bs) y[i] = w[i];
  if (tid == 0) { tim[0] = wall_clock64() - t0; tim[1] = tt; tim[2] = tg; }
}
template <int VAR> void run(const char *name, double *F, int m, double *y, long long *tim, hipStream_t st, int thr) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); float ms;
  for (int rep = 0; rep < 3; rep++) {
    CK(hipEventRecord(a, st));
    hipLaunchKernelGGL(k_fw<VAR>, dim3(1), dim3(thr), m * 8, st, F, m, m, y, tim);
    CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
  }
  long long h[3]; CK(hipMemcpy(h, tim, 24, hipMemcpyDeviceToHost));
  printf("%-28s thr=%d: %.1f us  (wall_clock ticks: total %lld, trsv %lld, gemv %lld; 100 MHz => %.1f / %.1f / %.1f us)\n", name, thr, ms * 1e3, h[0], h[1], h[2], h[0] / 100.0, h[1] / 100.0, h[2] / 100.0);
}
int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  for (int m : {666, 2000}) {
    std::vector<double> h((size_t)m * m, 0.0);
    for (int j = 0; j < m; j++) for (int i = j; i < m; i++) h[(size_t)j * m + i] = (i == j) ? 1.0 : 1e-3 * ((i * 7 + j * 3) % 11);
    double *F, *y; long long *tim; CK(hipMalloc(&F, h.size() * 8)); CK(hipMalloc(&y, m * 8)); CK(hipMalloc(&tim, 64));
    CK(hipMemcpy(F, h.data(), h.size() * 8, hipMemcpyHostToDevice)); CK(hipMemset(y, 0, m * 8));
    printf("m = %d\n", m);
    for (int thr : {256, 512}) {
      run<0>("full", F, m, y, tim, st, thr);
      run<1>("no trsv chain", F, m, y, tim, st, thr);
      run<2>("no lr prefetch", F, m, y, tim, st, thr);
      run<3>("no gemv", F, m, y, tim, st, thr);
    }
  }
  return 0;
}

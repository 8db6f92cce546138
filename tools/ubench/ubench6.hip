// latency of dependent FP64 chains on one CU (calibration only, not product code):
// division, mul+sub, v_readlane broadcast, and the 8-column register sweep of k_ldl_panel without its trailing update
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ double bcast(double v, int lane) {
  union { double d; int i[2]; } u; u.d = v;
  u.i[0] = __builtin_amdgcn_readlane(u.i[0], lane); u.i[1] = __builtin_amdgcn_readlane(u.i[1], lane);
  return u.d;
}
constexpr int N = 512;
template <int MODE>
__global__ void __launch_bounds__(1024) k_chain(const double *in, double *out, long long *tim) {
#pragma clang fp contract(off)
  double x = in[threadIdx.x], y = in[1024 + threadIdx.x], c = in[2048 + threadIdx.x];
  __syncthreads();
  long long w0 = wall_clock64(), c0 = clock64();
  if (MODE == 0) { for (int i = 0; i < N; i++) x = x / y + c; }                       // div + add
  if (MODE == 1) { for (int i = 0; i < N; i++) x = x * y - c; }                       // mul + sub
  if (MODE == 2) { for (int i = 0; i < N; i++) x = bcast(x, i & 63) + c; }            // readlane + add
  if (MODE == 3) { for (int i = 0; i < N; i++) { double r = __builtin_amdgcn_rcp(y); x = x * r + c; y = y + x; } }   // raw rcp
  if (MODE == 4) { for (int i = 0; i < N; i++) { x = c / x; } }                       // div only, denominator chain
  if (MODE == 5) { for (int i = 0; i < N; i++) { double xk = bcast(x, i & 63); x = (x / xk) * y - c; } }   // the column step: bcast, div, mul, sub
  long long c1 = clock64(), w1 = wall_clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x + y;
  if (threadIdx.x == 0) { tim[0] = c1 - c0; tim[1] = w1 - w0; }
}
constexpr int NB = 64;
template <int SW>
__global__ void __launch_bounds__(512) k_sweep(const double *A, double *out, long long *tim) {
#pragma clang fp contract(off)
  __shared__ double S[NB][NB + 1];
  const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6, ny = blockDim.x >> 6;
  for (int j = ty; j < NB; j += ny) S[tx][j] = A[j * 64 + tx];
  __syncthreads();
  long long w0 = wall_clock64(), c0c = clock64();
  double accd = 0.0;
  for (int c0 = 0; c0 < NB; c0 += SW) {
    double x[SW];
#pragma unroll
    for (int cc = 0; cc < SW; cc++) x[cc] = S[tx][c0 + cc];
#pragma unroll
    for (int k = 0; k < SW; k++) {
      const int gc = c0 + k;
      const double xkk = bcast(x[k], gc);
      const bool accept = xkk > 1e-30;
      const double l = accept ? x[k] / xkk : 0.0;
#pragma unroll
      for (int j = k + 1; j < SW; j++) x[j] -= bcast(l, c0 + j) * x[k];
      accd += l;
    }
    __syncthreads();
  }
  long long c1 = clock64(), w1 = wall_clock64();
  out[tid] = accd;
  if (tid == 0) { tim[0] = c1 - c0c; tim[1] = w1 - w0; }
}
int main() {
  std::vector<double> h(3072 + 4096);
  for (int i = 0; i < 1024; i++) { h[i] = 1.0 + 1e-3 * i; h[1024 + i] = 1.0000001 + 1e-9 * i; h[2048 + i] = 1e-7; }
  for (int j = 0; j < 64; j++) for (int i = 0; i < 64; i++) h[3072 + j * 64 + i] = (i == j) ? 70.0 : 1.0 / (1 + abs(i - j));
  double *d, *o; long long *t;
  CK(hipMalloc(&d, h.size() * 8)); CK(hipMalloc(&o, 1 << 16)); CK(hipMalloc(&t, 64));
  CK(hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  long long ht[2];
#define RUN(name, steps, ...) do { for (int rep = 0; rep < 3; rep++) { __VA_ARGS__; CK(hipDeviceSynchronize()); } \
    CK(hipMemcpy(ht, t, 16, hipMemcpyDeviceToHost)); \
    printf("%-44s %8.1f clk/step  %8.2f ns/step\n", name, (double)ht[0] / (steps), 10.0 * ht[1] / (steps)); } while (0)
  for (int thr : {64, 256, 512, 1024}) {
    char nm[96];
    snprintf(nm, 96, "div+add chain, %d thr", thr); RUN(nm, N, hipLaunchKernelGGL(k_chain<0>, dim3(1), dim3(thr), 0, 0, d, o, t));
    snprintf(nm, 96, "mul+sub chain, %d thr", thr); RUN(nm, N, hipLaunchKernelGGL(k_chain<1>, dim3(1), dim3(thr), 0, 0, d, o, t));
    snprintf(nm, 96, "readlane+add chain, %d thr", thr); RUN(nm, N, hipLaunchKernelGGL(k_chain<2>, dim3(1), dim3(thr), 0, 0, d, o, t));
    snprintf(nm, 96, "rcp+mul+add+add chain, %d thr", thr); RUN(nm, N, hipLaunchKernelGGL(k_chain<3>, dim3(1), dim3(thr), 0, 0, d, o, t));
    snprintf(nm, 96, "div only chain, %d thr", thr); RUN(nm, N, hipLaunchKernelGGL(k_chain<4>, dim3(1), dim3(thr), 0, 0, d, o, t));
    snprintf(nm, 96, "column step (bcast,div,mul,sub), %d thr", thr); RUN(nm, N, hipLaunchKernelGGL(k_chain<5>, dim3(1), dim3(thr), 0, 0, d, o, t));
  }
  for (int thr : {64, 256, 512}) {
    char nm[96];
    snprintf(nm, 96, "sweep SW=8 (per column), %d thr", thr); RUN(nm, 64, hipLaunchKernelGGL(k_sweep<8>, dim3(1), dim3(thr), 0, 0, d + 3072, o, t));
    snprintf(nm, 96, "sweep SW=16 (per column), %d thr", thr); RUN(nm, 64, hipLaunchKernelGGL(k_sweep<16>, dim3(1), dim3(thr), 0, 0, d + 3072, o, t));
    snprintf(nm, 96, "sweep SW=4 (per column), %d thr", thr); RUN(nm, 64, hipLaunchKernelGGL(k_sweep<4>, dim3(1), dim3(thr), 0, 0, d + 3072, o, t));
  }
  return 0;
}

// variants of the 64x64 diagonal-block LDL' loop and of the row solve (calibration only, not product code)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int NB = 64;
// VAR 1: per-thread division, 1 barrier/step.  VAR 2: lcol by tid<kb, 2 barriers/step.
template <int VAR>
__global__ void __launch_bounds__(1024) k_ldl(const double *A, double *out, long long *tim) {
#pragma clang fp contract(off)
  __shared__ double S[NB][NB + 1];
  __shared__ double lcol[NB], ds[NB];
  const int tid = threadIdx.x, bs = blockDim.x;
  const int tx = tid & 63, ty = tid >> 6, ny = bs >> 6;
  const int kb = 64;
  for (int j = ty; j < NB; j += ny) S[tx][j] = (j <= tx) ? A[j * 64 + tx] : 0.0;
  __syncthreads();
  long long t0 = wall_clock64();
  for (int k = 0; k < kb; k++) {
    double xkk = S[k][k];
    if (xkk > 1e-30) {
      if (VAR == 1) {
        const double sik = S[tx][k];
        if (tid == 0) ds[k] = xkk;
        for (int i = k + 1 + ty; i < kb; i += ny)
          if (tx >= i) S[tx][i] -= (S[i][k] / xkk) * sik;
      } else {
        if (tid > k && tid < kb) lcol[tid] = S[tid][k] / xkk;
        if (tid == 0) ds[k] = xkk;
        __syncthreads();
        const double sik = S[tx][k];
        for (int i = k + 1 + ty; i < kb; i += ny)
          if (tx >= i) S[tx][i] -= lcol[i] * sik;
      }
    }
    __syncthreads();
  }
  long long t1 = wall_clock64();
  for (int j = ty; j < NB; j += ny) out[j * 64 + tx] = S[tx][j];
  if (tid == 0) tim[0] = t1 - t0;
}
// VAR 3: one wave, lane per row, registers, 16-column sub-blocks; trailing columns updated by a rolled loop via LDS
__global__ void __launch_bounds__(64) k_ldl_w(const double *A, double *out, long long *tim) {
#pragma clang fp contract(off)
  __shared__ double S[NB][NB + 1];     // S[row][col]
  __shared__ double Lc[NB][NB];        // Lc[k][row] scaled column k
  const int r = threadIdx.x;
  for (int j = 0; j < NB; j++) S[r][j] = (j <= r) ? A[j * 64 + r] : 0.0;
  long long t0 = wall_clock64();
  for (int c0 = 0; c0 < NB; c0 += 16) {
    double x[16];
#pragma unroll
    for (int c = 0; c < 16; c++) x[c] = S[r][c0 + c];
#pragma unroll
    for (int k = 0; k < 16; k++) {
      union { double d; int i[2]; } u; u.d = x[k];
      u.i[0] = __builtin_amdgcn_readlane(u.i[0], c0 + k); u.i[1] = __builtin_amdgcn_readlane(u.i[1], c0 + k);
      const double xkk = u.d;
      const double l = x[k] / xkk;
#pragma unroll
      for (int j = k + 1; j < 16; j++) {
        union { double d; int i[2]; } v; v.d = l;
        v.i[0] = __builtin_amdgcn_readlane(v.i[0], c0 + j); v.i[1] = __builtin_amdgcn_readlane(v.i[1], c0 + j);
        x[j] -= v.d * x[k];      // rows r < c0+j carry garbage in the upper triangle: never read
      }
      Lc[c0 + k][r] = l;
    }
    // trailing columns of the block: x(r,j) -= sum_k l_jk * x_rk  (k ascending, as the right-looking sweep would)
    for (int j = c0 + 16; j < NB; j++) {
      double v = S[r][j];
#pragma unroll
      for (int k = 0; k < 16; k++) v -= Lc[c0 + k][j] * x[k];
      S[r][j] = v;
    }
  }
  long long t1 = wall_clock64();
  for (int j = 0; j < NB; j++) out[j * 64 + r] = Lc[j][r];
  if (r == 0) tim[0] = t1 - t0;
}
// row solve variants: thread per row, L11 (scaled) in LDS
template <int VAR>
__global__ void __launch_bounds__(256) k_trsm(const double *L11, const double *A, double *X, int ms, long long *tim) {
#pragma clang fp contract(off)
  __shared__ double S[NB][NB + 1];
  __shared__ double Xs[48 * 256];
  __shared__ double ds[NB];
  const int tid = threadIdx.x;
  for (int idx = tid; idx < 64 * 64; idx += 256) { int i = idx & 63, j = idx >> 6; S[i][j] = L11[j * 64 + i]; }
  if (tid < 64) ds[tid] = 1.0 + tid;
  __syncthreads();
  long long t0 = wall_clock64();
  const int r = blockIdx.x * 256 + tid;
  if (r < ms) {
    for (int c0 = 0; c0 < 64; c0 += 16) {
      double acc[16], x[16];
#pragma unroll
      for (int cc = 0; cc < 16; cc++) acc[cc] = A[(int64_t)(c0 + cc) * ms + r];
      for (int j = 0; j < c0; j++) {
        const double xj = Xs[j * 256 + tid];
#pragma unroll
        for (int cc = 0; cc < 16; cc++) acc[cc] -= xj * S[c0 + cc][j];
      }
#pragma unroll
      for (int cc = 0; cc < 16; cc++) {
        double v = acc[cc];
#pragma unroll
        for (int jj = 0; jj < 16; jj++) if (jj < cc) v -= x[jj] * S[c0 + cc][c0 + jj];
        const double dc = ds[c0 + cc];
        x[cc] = dc > 0.0 ? v : 0.0;
        X[(int64_t)(c0 + cc) * ms + r] = dc > 0.0 ? v / dc : 0.0;
        if (c0 < 48) Xs[(c0 + cc) * 256 + tid] = x[cc];
      }
    }
  }
  long long t1 = wall_clock64();
  if (tid == 0 && blockIdx.x == 0) tim[0] = t1 - t0;
}
int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); float ms;
  std::vector<double> h(64 * 64, 0.0);
  for (int j = 0; j < 64; j++) for (int i = j; i < 64; i++) h[j * 64 + i] = (i == j) ? 64.0 + i : 0.01 * ((i * 7 + j * 3) % 11);
  double *A, *o1, *o2; long long *tim; CK(hipMalloc(&A, 64 * 64 * 8)); CK(hipMalloc(&o1, 64 * 64 * 8)); CK(hipMalloc(&o2, 64 * 64 * 8)); CK(hipMalloc(&tim, 64));
  CK(hipMemcpy(A, h.data(), 64 * 64 * 8, hipMemcpyHostToDevice));
  long long t;
#define RUN(NAME, LAUNCH) for (int rep = 0; rep < 3; rep++) { CK(hipEventRecord(a, st)); LAUNCH; CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b)); } \
  CK(hipMemcpy(&t, tim, 8, hipMemcpyDeviceToHost)); printf("%-40s %.1f us (in-kernel loop %.1f us)\n", NAME, ms * 1e3, t / 100.0);
  for (int thr : {256, 512, 1024}) {
    char nm[64];
    snprintf(nm, 64, "ldl V1 per-thread div, %d thr", thr); RUN(nm, hipLaunchKernelGGL(k_ldl<1>, dim3(1), dim3(thr), 0, st, A, o1, tim));
    snprintf(nm, 64, "ldl V2 lcol + 2 barriers, %d thr", thr); RUN(nm, hipLaunchKernelGGL(k_ldl<2>, dim3(1), dim3(thr), 0, st, A, o2, tim));
  }
  RUN("ldl V3 one wave, 16-col register sweeps", hipLaunchKernelGGL(k_ldl_w, dim3(1), dim3(64), 0, st, A, o2, tim));
  { std::vector<double> r1(4096), r2(4096); CK(hipMemcpy(r1.data(), o1, 32768, hipMemcpyDeviceToHost)); CK(hipMemcpy(r2.data(), o2, 32768, hipMemcpyDeviceToHost));
    double md = 0; for (int j = 0; j < 64; j++) for (int i = j + 1; i < 64; i++) md = fmax(md, fabs(r1[j * 64 + i] - r2[j * 64 + i])); printf("max |V1 - V3| on L: %g\n", md); }
  const int msz = 4096; double *P, *X; CK(hipMalloc(&P, (size_t)64 * msz * 8)); CK(hipMalloc(&X, (size_t)64 * msz * 8)); CK(hipMemset(P, 0, (size_t)64 * msz * 8));
  for (int nwg : {1, 3, 16}) { char nm[64]; snprintf(nm, 64, "trsm thread-per-row, %d WGs", nwg); RUN(nm, hipLaunchKernelGGL(k_trsm<0>, dim3(nwg), dim3(256), 0, st, o1, P, X, nwg * 256, tim)); }
  return 0;
}

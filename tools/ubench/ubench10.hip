// ubench10: what a row-dot sweep (one wavefront per matrix row, the shape of k_sfw_rows / k_sfw_diag / k_sbw_*) streams on a whole MI355X,
// as a function of (a) the row pitch -- power-of-two pitches put the concurrent streams of all wavefronts on the same memory channels --,
// (b) loads in flight per lane, (c) software pipelining across the chunks of a row, (d) the vector read beside the matrix, (e) nontemporal loads,
// (f) launch size (burst against steady state).   hipcc --offload-arch=gfx950 -O3 ubench10.hip -o ubench10
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef double d2 __attribute__((ext_vector_type(2)));

template <int NL, bool VEC, bool NT, bool PIPE>
__global__ void __launch_bounds__(256) k_rows(const double *__restrict__ M, long long pitch, int n, const double *__restrict__ x, double *__restrict__ y, int nrows) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = 4 * blockIdx.x + wave;
  if (r >= nrows) return;
  const d2 *M2 = (const d2 *)(M + (long long)r * pitch);
  const d2 *x2 = (const d2 *)x;
  const int npair = n >> 1;
  double a0 = 0, a1 = 0;
  if (!PIPE) {
    for (int p0 = 0; p0 < npair; p0 += 64 * NL) {
      d2 v[NL], xv[NL];
#pragma unroll
      for (int k = 0; k < NL; k++) {
        const int pc = min(p0 + lane + 64 * k, npair - 1);
        v[k] = NT ? __builtin_nontemporal_load(M2 + pc) : M2[pc];
        if (VEC) xv[k] = x2[pc];
      }
#pragma unroll
      for (int k = 0; k < NL; k++) {
        const bool in = p0 + lane + 64 * k < npair;
        a0 += in ? v[k].x * (VEC ? xv[k].x : 1.0) : 0.0;
        a1 += in ? v[k].y * (VEC ? xv[k].y : 1.0) : 0.0;
      }
    }
  } else {
    d2 v[NL], xv[NL], w[NL], xw[NL];
#pragma unroll
    for (int k = 0; k < NL; k++) {
      const int pc = min(lane + 64 * k, npair - 1);
      v[k] = NT ? __builtin_nontemporal_load(M2 + pc) : M2[pc];
      if (VEC) xv[k] = x2[pc];
    }
    for (int p0 = 0; p0 < npair; p0 += 64 * NL) {
      const int p1 = p0 + 64 * NL;
      if (p1 < npair) {
#pragma unroll
        for (int k = 0; k < NL; k++) {
          const int pc = min(p1 + lane + 64 * k, npair - 1);
          w[k] = NT ? __builtin_nontemporal_load(M2 + pc) : M2[pc];
          if (VEC) xw[k] = x2[pc];
        }
      }
#pragma unroll
      for (int k = 0; k < NL; k++) {
        const bool in = p0 + lane + 64 * k < npair;
        a0 += in ? v[k].x * (VEC ? xv[k].x : 1.0) : 0.0;
        a1 += in ? v[k].y * (VEC ? xv[k].y : 1.0) : 0.0;
      }
#pragma unroll
      for (int k = 0; k < NL; k++) { v[k] = w[k]; if (VEC) xv[k] = xw[k]; }
    }
  }
  double a = a0 + a1;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
  if (lane == 0) y[r] = a;
}

// persistent form: G workgroups, each takes rows r = wg, wg + G, ... (grid-stride), the next row's first chunk issued before the reduction of this one
template <int NL, bool VEC>
__global__ void __launch_bounds__(256) k_rows_persist(const double *__restrict__ M, long long pitch, int n, const double *__restrict__ x, double *__restrict__ y, int nrows) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int npair = n >> 1;
  const d2 *x2 = (const d2 *)x;
  for (int r = 4 * blockIdx.x + wave; r < nrows; r += 4 * gridDim.x) {
    const d2 *M2 = (const d2 *)(M + (long long)r * pitch);
    double a0 = 0, a1 = 0;
    for (int p0 = 0; p0 < npair; p0 += 64 * NL) {
      d2 v[NL], xv[NL];
#pragma unroll
      for (int k = 0; k < NL; k++) {
        const int pc = min(p0 + lane + 64 * k, npair - 1);
        v[k] = M2[pc];
        if (VEC) xv[k] = x2[pc];
      }
#pragma unroll
      for (int k = 0; k < NL; k++) {
        const bool in = p0 + lane + 64 * k < npair;
        a0 += in ? v[k].x * (VEC ? xv[k].x : 1.0) : 0.0;
        a1 += in ? v[k].y * (VEC ? xv[k].y : 1.0) : 0.0;
      }
    }
    double a = a0 + a1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if (lane == 0) y[r] = a;
  }
}

static double *M, *x, *y;
static hipEvent_t e0, e1;
template <class F> double timeit(F f, int reps = 20) {
  for (int i = 0; i < 3; i++) f();
  hipDeviceSynchronize();
  std::vector<float> t;
  for (int i = 0; i < reps; i++) {
    hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); t.push_back(ms);
  }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2] * 1e3;      // median, us
}
#define RUN(name, launch, rows, n)                                                                                         \
  { const double us = timeit([&] { launch; });                                                                              \
    printf("%-58s rows %6d x %5d  pitch %6lld : %8.1f us  %6.2f TB/s\n", name, rows, n, pitch, us, 8.0 * (double)(rows) * (n) / us / 1e6); }

int main(int argc, char **argv) {
  const long long cap = 1LL << 30;                       // 1 GB arena
  hipMalloc(&M, cap + (1 << 20)); hipMalloc(&x, 1 << 20); hipMalloc(&y, 1 << 22);
  hipMemset(M, 0, cap); hipMemset(x, 0, 1 << 20);
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int n : {2048, 4000}) {
    for (int rows : {2048, 8192, 32768}) {
      for (long long pad : {0LL, 16LL, 32LL, 96LL, 272LL}) {
        const long long pitch = n + pad;
        if (pitch * rows * 8 > cap) continue;
        const int wgs = (rows + 3) / 4;
        RUN("wave/row, 8 loads/lane, vector", (k_rows<8, true, false, false><<<wgs, 256>>>(M, pitch, n, x, y, rows)), rows, n);
        if (pad == 0 || pad == 32) {
          RUN("wave/row, 8 loads/lane, no vector", (k_rows<8, false, false, false><<<wgs, 256>>>(M, pitch, n, x, y, rows)), rows, n);
          RUN("wave/row, 16 loads/lane, vector", (k_rows<16, true, false, false><<<wgs, 256>>>(M, pitch, n, x, y, rows)), rows, n);
          RUN("wave/row, 4 loads/lane, vector", (k_rows<4, true, false, false><<<wgs, 256>>>(M, pitch, n, x, y, rows)), rows, n);
          RUN("wave/row, 8 loads/lane, vector, nontemporal", (k_rows<8, true, true, false><<<wgs, 256>>>(M, pitch, n, x, y, rows)), rows, n);
          RUN("wave/row, 4 loads/lane, vector, pipelined", (k_rows<4, true, false, true><<<wgs, 256>>>(M, pitch, n, x, y, rows)), rows, n);
          RUN("wave/row, 8 loads/lane, vector, pipelined", (k_rows<8, true, false, true><<<wgs, 256>>>(M, pitch, n, x, y, rows)), rows, n);
          for (int g : {256, 512, 1024, 2048})
            if (4 * g <= rows) {
              char nm[96]; snprintf(nm, 96, "persistent %4d workgroups, 8 loads/lane, vector", g);
              RUN(nm, (k_rows_persist<8, true><<<g, 256>>>(M, pitch, n, x, y, rows)), rows, n);
            }
        }
      }
      printf("\n");
    }
  }
  // dependent launches back to back: what a boundary costs (empty-ish kernels) -- 6 launches of 2048 x 2048 in a row
  {
    const long long pitch = 2048 + 32; const int rows = 2048, n = 2048, wgs = 512;
    RUN("6 dependent launches (2048 x 2048 each)", ({ for (int i = 0; i < 6; i++) k_rows<8, true, false, false><<<wgs, 256>>>(M + (long long)i * rows * pitch, pitch, n, x, y, rows); }), 6 * rows, n);
    RUN("1 launch of the same bytes (12288 x 2048)", (k_rows<8, true, false, false><<<6 * wgs, 256>>>(M, pitch, n, x, y, 6 * rows)), 6 * rows, n);
  }
  return 0;
}

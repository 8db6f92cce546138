// ubench8: what v_mfma_f64_16x16x4_f64 sustains on a whole MI355X -- every compute unit busy, registers only -- and the shader clock it
// runs at meanwhile (clock64 ticks per microsecond of wall_clock64, 100 MHz).  hipcc --offload-arch=gfx950 -O3 ubench8.hip -o ubench8
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void __launch_bounds__(512) k(double *out, int iters, long long *clk) {
  d4 acc[NACC];
  for (int a = 0; a < NACC; a++) acc[a] = d4{0, 0, 0, 0};
  double x = threadIdx.x * 1e-3, y = blockIdx.x * 1e-6;
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < iters; i++)
#pragma unroll
    for (int a = 0; a < NACC; a++) acc[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[a], 0, 0, 0);
  const long long c1 = clock64(), w1 = wall_clock64();
  double s = 0;
  for (int a = 0; a < NACC; a++) s += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
  if (s == 12345.678) out[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}
template <int NACC> void run(int wgs, int iters) {
  double *out; long long *clk, h[2];
  hipMalloc(&out, 8); hipMalloc(&clk, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NACC><<<wgs, 512>>>(out, iters, clk);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<NACC><<<wgs, 512>>>(out, iters, clk);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  const double flops = (double)wgs * 8 * iters * NACC * 2048.0;
  printf("wgs %4d x 8 waves, %d accumulators/wave, %d iters: %.3f ms  %.1f TF/s   shader clock %.0f MHz (clock64 / wall_clock64)  cycles per MFMA per SIMD %.1f\n", wgs, NACC, iters, ms,
         flops / ms / 1e9, (double)h[0] / ((double)h[1] / 100.0), (double)h[0] / (2.0 * iters * NACC));
}
int main() {
  run<2>(256, 4000); run<4>(256, 4000); run<2>(512, 4000); run<2>(64, 4000); run<2>(8, 4000);
  return 0;
}

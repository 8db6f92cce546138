// ubench9: the MFMA loop of panel_role_tiles_stream in isolation -- a 128 x 128 x 64 product per step out of LDS (operand blocks
// [64][72] doubles, 8 wavefronts of 64 x 32, 8 accumulators, 6 operand reads per 8 MFMAs), every compute unit busy, nothing else.
// hipcc --offload-arch=gfx950 -O3 ubench9.hip -o ubench9
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int NB = 64, UTP = 72;
template <int VARIANT>
__global__ void __launch_bounds__(512) k(double *out, int steps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double (*Ab)[UTP] = (double (*)[UTP])smem;
  const int tid = threadIdx.x;
  for (int e = tid; e < 4 * NB * UTP; e += 512) ((double *)smem)[e] = 1e-3 * (e % 97);
  __syncthreads();
  const int wv = tid >> 6, l = tid & 63;
  const int qa = wv >> 2, qb = (wv & 3) >> 1, cj = (wv & 1) * 32, lk = l >> 4, ll = l & 15;
  double (*As)[UTP] = Ab + qa * NB, (*Bs)[UTP] = Ab + (2 + qb) * NB;
  double c[4][2][4];
  for (int ra = 0; ra < 4; ra++) for (int cb = 0; cb < 2; cb++) for (int r = 0; r < 4; r++) c[ra][cb][r] = 0.0;
  for (int s = 0; s < steps; s++) {
    d4 acc[4][2];
    for (int ra = 0; ra < 4; ra++) for (int cb = 0; cb < 2; cb++) acc[ra][cb] = d4{0, 0, 0, 0};
    double ao[4], bo[2];
#pragma unroll
    for (int ra = 0; ra < 4; ra++) ao[ra] = As[lk][ra * 16 + ll];
#pragma unroll
    for (int cb = 0; cb < 2; cb++) bo[cb] = Bs[lk][cj + cb * 16 + ll];
#pragma unroll
    for (int kk = 0; kk < NB; kk += 4) {
      const int kn = kk + 4 < NB ? kk + 4 : NB - 4;
      double an[4], bn[2];
#pragma unroll
      for (int ra = 0; ra < 4; ra++) an[ra] = As[kn + lk][ra * 16 + ll];
#pragma unroll
      for (int cb = 0; cb < 2; cb++) bn[cb] = Bs[kn + lk][cj + cb * 16 + ll];
#pragma unroll
      for (int ra = 0; ra < 4; ra++)
#pragma unroll
        for (int cb = 0; cb < 2; cb++) acc[ra][cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(bo[cb], ao[ra], acc[ra][cb], 0, 0, 0);
#pragma unroll
      for (int ra = 0; ra < 4; ra++) ao[ra] = an[ra];
#pragma unroll
      for (int cb = 0; cb < 2; cb++) bo[cb] = bn[cb];
    }
    for (int ra = 0; ra < 4; ra++) for (int cb = 0; cb < 2; cb++) for (int r = 0; r < 4; r++) c[ra][cb][r] -= acc[ra][cb][r];
    if (VARIANT == 1) __syncthreads();
  }
  double sum = 0;
  for (int ra = 0; ra < 4; ra++) for (int cb = 0; cb < 2; cb++) for (int r = 0; r < 4; r++) sum += c[ra][cb][r];
  if (sum == 12345.678) out[0] = sum;
}
template <int V> void run(int wgs, int steps) {
  double *out; (void)hipMalloc(&out, 8);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipFuncSetAttribute((const void *)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * NB * UTP * 8);
  k<V><<<wgs, 512, 4 * NB * UTP * 8>>>(out, steps);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<V><<<wgs, 512, 4 * NB * UTP * 8>>>(out, steps);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)wgs * steps * 128.0 * 128.0 * 64.0 * 2.0;
  printf("variant %d (%s): %d workgroups x %d steps: %.3f ms, %.2f us per step, %.1f TF/s\n", V, V ? "barrier per step" : "no barrier", wgs, steps, ms, 1e3 * ms / steps, flops / ms / 1e9);
}
int main() { run<0>(256, 200); run<1>(256, 200); run<0>(128, 200); run<0>(32, 200); return 0; }

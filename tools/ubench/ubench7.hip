// the in-block solve chains of the sweep kernels in isolation (calibration only): clocks per 64-step chain
#include "../../sedumi_amd/csrc/sdm_chol.hip"
using namespace sdm;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
template <int MODE>
__global__ void __launch_bounds__(1024) k_chain(const double *in, double *out, long long *tim, int reps) {
  SDM_DYN_SMEM(smem);
  double *Sd = (double *)smem, *Sb = Sd + 2 * SNB * SNB;
  for (int i = threadIdx.x; i < SOLVE_STAGE_DOUBLES; i += blockDim.x) Sd[i] = in[i % 4096] * 1e-3;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  double wi = in[lane], cacc = 0.0;
  long long c0 = clock64(), w0 = wall_clock64();
  if (threadIdx.x < 64) {
    for (int r = 0; r < reps; r++) {
      if (MODE == 0) wi = trsv_fw_fused(Sd + (r & 1) * SNB * SNB, Sb + (r & 1) * SNB * SNB, wi, cacc, lane);
      if (MODE == 1) wi = trsv_bw_fused(Sd + (r & 1) * SNB * SNB, Sb + (r & 1) * SNB * SBP, wi, cacc, lane);
      if (MODE == 2) wi = trsv_fw_block(Sd + (r & 1) * SNB * SNB, wi, lane);
      if (MODE == 3) wi = trsv_bw_block(Sd + (r & 1) * SNB * SNB, wi, lane);
    }
  }
  long long c1 = clock64(), w1 = wall_clock64();
  out[threadIdx.x] = wi + cacc;
  if (threadIdx.x == 0) { tim[0] = c1 - c0; tim[1] = w1 - w0; }
}
int main() {
  std::vector<double> h(4096);
  for (int i = 0; i < 4096; i++) h[i] = 0.001 * (i % 97);
  double *d, *o; long long *t;
  CK(hipMalloc(&d, 4096 * 8)); CK(hipMalloc(&o, 1 << 16)); CK(hipMalloc(&t, 64));
  CK(hipMemcpy(d, h.data(), 4096 * 8, hipMemcpyHostToDevice));
  const size_t lds = SOLVE_STAGE_DOUBLES * 8;
  CK(hipFuncSetAttribute((const void *)k_chain<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute((const void *)k_chain<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute((const void *)k_chain<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute((const void *)k_chain<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  long long ht[2];
  const int reps = 20;
#define RUN(name, K) do { for (int rep = 0; rep < 3; rep++) { hipLaunchKernelGGL(K, dim3(1), dim3(thr), lds, 0, d, o, t, reps); CK(hipDeviceSynchronize()); } \
    CK(hipMemcpy(ht, t, 16, hipMemcpyDeviceToHost)); printf("%-28s %4d thr %8.0f clk/chain %8.1f ns/chain\n", name, thr, (double)ht[0] / reps, 10.0 * ht[1] / reps); } while (0)
  for (int thr : {64, 1024}) {
    RUN("fw fused (TCF ping-pong)", k_chain<0>);
    RUN("bw fused (TCF ping-pong)", k_chain<1>);
    RUN("fw plain (TCH=32)", k_chain<2>);
    RUN("bw plain (TCH=32)", k_chain<3>);
  }
  return 0;
}

// phase clocks of the product solve kernel (k_ldl_single) on a synthetic dense front (calibration only)
#define SDM_PHASES 1
#include "../../sedumi_amd/csrc/sdm_chol.hip"
namespace sdm { void set_error(const std::string &) {} }
using namespace sdm;
int main(int argc, char **argv) {
  const int m = argc > 1 ? atoi(argv[1]) : 666, thr = argc > 2 ? atoi(argv[2]) : 1024;
  const int ld = m + (m & 1), npan = (m + 63) / 64;
  std::vector<double> h((size_t)ld * m, 0.0), hT((size_t)npan * 4096, 0.0), hd(m, 2.0), rhs(m, 1.0);
  for (int j = 0; j < m; j++) for (int i = j; i < m; i++) {
    const double v = (i == j) ? 1.0 : 1e-3 * ((i * 7 + j * 3) % 11 - 5);
    h[(size_t)j * ld + i] = v;
    if (i / 64 == j / 64) hT[(size_t)(j / 64) * 4096 + (i % 64) * 64 + (j % 64)] = v;
  }
  std::vector<int> perm(m); for (int i = 0; i < m; i++) perm[i] = i;
  double *F, *DT, *d, *r, *y, *wg; int *p;
  SDM_HIP_CHECK(hipMalloc(&F, h.size() * 8)); SDM_HIP_CHECK(hipMalloc(&DT, hT.size() * 8)); SDM_HIP_CHECK(hipMalloc(&d, m * 8));
  SDM_HIP_CHECK(hipMalloc(&r, m * 8)); SDM_HIP_CHECK(hipMalloc(&y, m * 8)); SDM_HIP_CHECK(hipMalloc(&wg, m * 8)); SDM_HIP_CHECK(hipMalloc(&p, m * 4));
  SDM_HIP_CHECK(hipMemcpy(F, h.data(), h.size() * 8, hipMemcpyHostToDevice)); SDM_HIP_CHECK(hipMemcpy(DT, hT.data(), hT.size() * 8, hipMemcpyHostToDevice));
  SDM_HIP_CHECK(hipMemcpy(d, hd.data(), m * 8, hipMemcpyHostToDevice)); SDM_HIP_CHECK(hipMemcpy(r, rhs.data(), m * 8, hipMemcpyHostToDevice));
  SDM_HIP_CHECK(hipMemcpy(p, perm.data(), m * 4, hipMemcpyHostToDevice));
  hipStream_t st; SDM_HIP_CHECK(hipStreamCreate(&st));
  hipEvent_t a, b; SDM_HIP_CHECK(hipEventCreate(&a)); SDM_HIP_CHECK(hipEventCreate(&b));
  const size_t lds = (size_t)m * 8;
  SDM_HIP_CHECK(hipFuncSetAttribute((const void *)k_ldl_single, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  for (int mode : {1, 4, 7}) {
    float ms = 0; unsigned long long z[32] = {0};
    for (int rep = 0; rep < 3; rep++) {
      SDM_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(sdm_phase_acc), z, sizeof(z)));
      SDM_HIP_CHECK(hipEventRecord(a, st));
      hipLaunchKernelGGL(k_ldl_single, dim3(1), dim3(thr), lds, st, F, DT, m, p, d, r, y, wg, 1, mode);
      SDM_HIP_CHECK(hipEventRecord(b, st)); SDM_HIP_CHECK(hipEventSynchronize(b)); SDM_HIP_CHECK(hipEventElapsedTime(&ms, a, b));
    }
    SDM_HIP_CHECK(hipMemcpyFromSymbol(z, HIP_SYMBOL(sdm_phase_acc), sizeof(z)));
    printf("m=%d thr=%d mode=%d: %.1f us | fw: stage0 %.1f trsv %.1f bar %.1f stage %.1f gemv %.1f bar %.1f | bw: stage+dots %.1f bar %.1f trsv %.1f bar %.1f\n", m, thr, mode, ms * 1e3,
           z[0] / 100., z[1] / 100., z[2] / 100., z[3] / 100., z[4] / 100., z[5] / 100., z[8] / 100., z[9] / 100., z[10] / 100., z[11] / 100.);
  }
  return 0;
}

// phase clocks of the product factor kernels on a synthetic dense front (calibration only)
#define SDM_PHASES 1
#include "../../sedumi_amd/csrc/sdm_chol.hip"
namespace sdm { void set_error(const std::string &) {} }
using namespace sdm;
int main(int argc, char **argv) {
  const int m = argc > 1 ? atoi(argv[1]) : 666;
  std::vector<sdm_int> Ljc(m + 1), Lir, perm(m), xs = {0, m}, Ajc(m + 1), Air;
  for (int j = 0; j < m; j++) { Ljc[j] = (sdm_int)Lir.size(); for (int i = j; i < m; i++) Lir.push_back(i); perm[j] = j; }
  Ljc[m] = (sdm_int)Lir.size();
  for (int j = 0; j < m; j++) { Ajc[j] = (sdm_int)Air.size(); for (int i = 0; i < m; i++) Air.push_back(i); }
  Ajc[m] = (sdm_int)Air.size();
  std::vector<double> A((size_t)m * m);
  for (int j = 0; j < m; j++) for (int i = 0; i < m; i++) A[(size_t)j * m + i] = (i == j) ? m : 0.5 * cos(0.37 * i * j + i + j);
  for (int j = 0; j < m; j++) for (int i = 0; i < j; i++) A[(size_t)j * m + i] = A[(size_t)i * m + j];
  sdm_plan P; SDM_HIP_CHECK(hipStreamCreate(&P.stream));
  chol_build(&P, m, Ljc.data(), Lir.data(), perm.data(), 1, xs.data(), Ajc.data(), Air.data());
  SDM_HIP_CHECK(hipMemcpy(P.ada_val.p, A.data(), A.size() * 8, hipMemcpyHostToDevice));
  hipEvent_t a, b; SDM_HIP_CHECK(hipEventCreate(&a)); SDM_HIP_CHECK(hipEventCreate(&b));
  float ms = 0; unsigned long long z[32] = {0};
  for (int rep = 0; rep < 3; rep++) {
    SDM_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(sdm_phase_acc), z, sizeof(z)));
    SDM_HIP_CHECK(hipEventRecord(a, P.stream));
    chol_factor(&P, 1e-12, 5e5, 1e-20, 0);
    SDM_HIP_CHECK(hipEventRecord(b, P.stream)); SDM_HIP_CHECK(hipEventSynchronize(b)); SDM_HIP_CHECK(hipEventElapsedTime(&ms, a, b));
  }
  unsigned long long y[32]; SDM_HIP_CHECK(hipMemcpyFromSymbol(y, HIP_SYMBOL(sdm_phase_acc), sizeof(y)));
  const int np = (m + 63) / 64;
  printf("m=%d factor %.1f us (%d panels) | panel kernel, work-item 0 of every WG summed over WGs: load+bar %.1f sweep %.1f lc+trail %.1f bar %.1f | copy %.1f bar %.1f writeback %.1f rows %.1f\n",
         m, ms * 1e3, np, y[16] / 100., y[17] / 100., y[18] / 100., y[19] / 100., y[20] / 100., y[21] / 100., y[22] / 100., y[23] / 100.);
  return 0;
}

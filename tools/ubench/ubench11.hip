// ubench11: the TRIANGULAR row-dot launch of the solves (k_sfw_diag: row r of the inverse block has r + 1 entries; nb = 2048: 16.8 MB) and the
// full-row launch (k_sfw_rows: 1952 rows of 2048) in the decompositions considered for round 6: one wavefront per row looping over 8 KB chunks
// (the round-3 kernel), the same with the long rows first, and "every wavefront exactly one round trip" (rows longer than CH entries split over
// the wavefronts of their workgroup, partial sums combined in LDS in a fixed order; rows of at most 256 entries four to a wavefront).
// hipcc --offload-arch=gfx950 -O3 ubench11.hip -o ubench11
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double d2 __attribute__((ext_vector_type(2)));

template <int NL>
__device__ __forceinline__ double seg_dot(const d2 *M2, const d2 *x2, int p_lo, int p_hi, int lane) {   // pairs [p_lo, p_hi), at most 64 * NL of them
  d2 v[NL], xv[NL];
#pragma unroll
  for (int k = 0; k < NL; k++) { const int pc = min(p_lo + lane + 64 * k, p_hi - 1); v[k] = M2[pc]; xv[k] = x2[pc]; }
  double a0 = 0, a1 = 0;
#pragma unroll
  for (int k = 0; k < NL; k++) { const bool in = p_lo + lane + 64 * k < p_hi; a0 += in ? v[k].x * xv[k].x : 0.0; a1 += in ? v[k].y * xv[k].y : 0.0; }
  return a0 + a1;
}
__device__ __forceinline__ double wsum(double a) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
  return a;
}
// A / B: one wavefront per row, loop over chunks; REV: long rows first
template <bool REV>
__global__ void __launch_bounds__(256) k_tri_loop(const double *__restrict__ M, long long pitch, int nb, const double *__restrict__ x, double *__restrict__ y) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int r = 4 * blockIdx.x + wave;
  if (r >= nb) return;
  if (REV) r = nb - 1 - r;
  const d2 *M2 = (const d2 *)(M + (long long)r * pitch); const d2 *x2 = (const d2 *)x;
  const int npair = (r + 2) >> 1;
  double a = 0;
  for (int p0 = 0; p0 < npair; p0 += 512) a += seg_dot<8>(M2, x2, p0, min(npair, p0 + 512), lane);
  a = wsum(a);
  if (lane == 0) y[r] = a;
}
// C: every wavefront one round trip of at most NL loads per lane.  Rows are dealt to workgroups of 4 wavefronts: a row of more than CH = 128 * NL
// entries takes ceil(len / CH) wavefronts of ONE workgroup (LDS combine, fixed order); the host passes the first row of every class.
template <int NL>
__global__ void __launch_bounds__(256) k_tri_flat(const double *__restrict__ M, long long pitch, int nb, const double *__restrict__ x, double *__restrict__ y, int wg1, int wg2) {
  // workgroups [0, wg1): rows of <= CH entries, one wavefront each (4 rows per workgroup); [wg1, wg2): rows of <= 2 CH entries, 2 wavefronts each
  // (2 rows per workgroup); [wg2, ..): rows of <= 4 CH entries, 4 wavefronts each
  __shared__ double part[4];
  constexpr int CH = 128 * NL, CP = CH / 2;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, b = blockIdx.x;
  int r, seg, nseg;
  if (b < wg1) { r = 4 * b + wave; seg = 0; nseg = 1; }
  else if (b < wg2) { r = 4 * wg1 + 2 * (b - wg1) + (wave >> 1); seg = wave & 1; nseg = 2; }
  else { r = 4 * wg1 + 2 * (wg2 - wg1) + (b - wg2); seg = wave; nseg = 4; }
  const bool live = r < nb;
  const int rr = live ? r : nb - 1;
  const d2 *M2 = (const d2 *)(M + (long long)rr * pitch); const d2 *x2 = (const d2 *)x;
  const int npair = (rr + 2) >> 1;
  const int lo = seg * CP, hi = min(npair, lo + CP);
  double a = lo < hi ? seg_dot<NL>(M2, x2, lo, hi, lane) : 0.0;
  a = wsum(a);
  if (nseg == 1) { if (lane == 0 && live) y[r] = a; return; }
  if (lane == 0) part[wave] = a;
  __syncthreads();
  if (lane == 0 && live) {
    if (nseg == 2 && seg == 0) y[r] = part[wave] + part[wave + 1];
    if (nseg == 4 && seg == 0) y[r] = (part[0] + part[1]) + (part[2] + part[3]);
  }
}
// full rows of n entries (k_sfw_rows): SPLIT wavefronts per row
template <int NL, int SPLIT>
__global__ void __launch_bounds__(256) k_full(const double *__restrict__ M, long long pitch, int n, int nrows, const double *__restrict__ x, double *__restrict__ y) {
  __shared__ double part[4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = (4 / SPLIT) * blockIdx.x + wave / SPLIT, seg = wave % SPLIT;
  const bool live = r < nrows;
  const d2 *M2 = (const d2 *)(M + (long long)(live ? r : 0) * pitch); const d2 *x2 = (const d2 *)x;
  const int npair = n >> 1, per = (npair + SPLIT - 1) / SPLIT;
  const int lo = seg * per, hi = min(npair, lo + per);
  double a = 0;
  for (int p0 = lo; p0 < hi; p0 += 64 * NL) a += seg_dot<NL>(M2, x2, p0, min(hi, p0 + 64 * NL), lane);
  a = wsum(a);
  if (SPLIT == 1) { if (lane == 0 && live) y[r] = a; return; }
  if (lane == 0) part[wave] = a;
  __syncthreads();
  if (lane == 0 && live && seg == 0) { double s = part[wave]; for (int q = 1; q < SPLIT; q++) s += part[wave + q]; y[r] = s; }
}

static hipEvent_t e0, e1;
static hipStream_t st;
// `f` enqueues ONE launch on `st`.  100 of them are captured into a hipGraph (serialised by the stream order: every launch depends on the one
// before it, as in a sweep) and the graph is replayed between two events: the host's own launch rate (6 us per launch on this box) is out of the
// picture, what is left is the device's boundary + the launch's execution.
template <class F> double timeit(F f, int reps = 15) {
  constexpr int NG = 100;
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  for (int q = 0; q < NG; q++) f(q);
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  for (int i = 0; i < 2; i++) hipGraphLaunch(ge, st);
  hipStreamSynchronize(st);
  std::vector<float> t;
  for (int i = 0; i < reps; i++) {
    hipEventRecord(e0, st); hipGraphLaunch(ge, st); hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); t.push_back(ms / NG);
  }
  hipGraphExecDestroy(ge); hipGraphDestroy(g);
  std::sort(t.begin(), t.end());
  return t[t.size() / 2] * 1e3;
}
static double *M, *x, *y;
static long long wrap_bytes = 1LL << 30;          // launches walk through an arena of this size: the reuse distance of the data
static double *Mq(int q, double mb) {             // base of launch q: consecutive launches read consecutive, distinct regions
  const long long step = ((long long)(mb * 1e6) + (1 << 20)) & ~((1LL << 20) - 1);
  const long long nfit = std::max(1LL, wrap_bytes / step);
  return M + (q % nfit) * (step / 8);
}
int main(int argc, char **argv) {
  const long long cap = 1LL << 30;
  hipMalloc(&M, cap + (64 << 20)); hipMalloc(&x, 1 << 20); hipMalloc(&y, 1 << 20);
  hipMemset(M, 0, cap + (64 << 20)); hipMemset(x, 0, 1 << 20);
  hipEventCreate(&e0); hipEventCreate(&e1); hipStreamCreate(&st);
  for (long long wrap : {1LL << 30, 128LL << 20}) {
  wrap_bytes = wrap;
  printf("---- arena the launches walk through: %lld MB (1024: every byte from HBM; 128: the reuse distance of a MAXCUT-4000 solve)\n", wrap >> 20);

  for (int nb : {2048, 1952, 1024, 666}) {
    const long long pitch = 2048;
    const double mb = 8.0 * nb * (nb + 1) / 2 / 1e6;
    double us;
    us = timeit([&](int q) { k_tri_loop<false><<<(nb + 3) / 4, 256, 0, st>>>(Mq(q, mb), pitch, nb, x, y); });
    printf("triangle nb %4d (%5.1f MB)  wave/row loop, short rows first     : %6.2f us  %5.2f TB/s\n", nb, mb, us, mb / us);
    us = timeit([&](int q) { k_tri_loop<true><<<(nb + 3) / 4, 256, 0, st>>>(Mq(q, mb), pitch, nb, x, y); });
    printf("triangle nb %4d (%5.1f MB)  wave/row loop, long rows first      : %6.2f us  %5.2f TB/s\n", nb, mb, us, mb / us);
    {
      constexpr int NL = 8, CH = 128 * NL;
      const int n1 = std::min(nb, CH), n2 = std::min(nb, 2 * CH) - n1, n4 = nb - n1 - n2;
      const int wg1 = (n1 + 3) / 4, wg2 = wg1 + (n2 + 1) / 2, wgs = wg2 + n4;
      us = timeit([&](int q) { k_tri_flat<NL><<<wgs, 256, 0, st>>>(Mq(q, mb), pitch, nb, x, y, wg1, wg2); });
      printf("triangle nb %4d (%5.1f MB)  one round trip, 8 loads/lane, %4d wgs: %6.2f us  %5.2f TB/s\n", nb, mb, wgs, us, mb / us);
    }
    {
      constexpr int NL = 4, CH = 128 * NL;
      const int n1 = std::min(nb, CH), n2 = std::min(nb, 2 * CH) - n1, n4 = nb - n1 - n2;
      const int wg1 = (n1 + 3) / 4, wg2 = wg1 + (n2 + 1) / 2, wgs = wg2 + n4;
      us = timeit([&](int q) { k_tri_flat<NL><<<wgs, 256, 0, st>>>(Mq(q, mb), pitch, nb, x, y, wg1, wg2); });
      printf("triangle nb %4d (%5.1f MB)  one round trip, 4 loads/lane, %4d wgs: %6.2f us  %5.2f TB/s\n", nb, mb, wgs, us, mb / us);
    }
  }
  for (int nrows : {1952, 2048, 6000}) {
    const int n = 2048; const long long pitch = 2048;
    const double mb = 8.0 * nrows * n / 1e6;
    double us;
    us = timeit([&](int q) { k_full<8, 1><<<(nrows + 3) / 4, 256, 0, st>>>(Mq(q, mb), pitch, n, nrows, x, y); });
    printf("full rows %4d x %d (%5.1f MB)  wave/row, 8 loads/lane x 2 trips    : %6.2f us  %5.2f TB/s\n", nrows, n, mb, us, mb / us);
    us = timeit([&](int q) { k_full<8, 2><<<(nrows + 1) / 2, 256, 0, st>>>(Mq(q, mb), pitch, n, nrows, x, y); });
    printf("full rows %4d x %d (%5.1f MB)  2 waves/row, 8 loads/lane, 1 trip   : %6.2f us  %5.2f TB/s\n", nrows, n, mb, us, mb / us);
    us = timeit([&](int q) { k_full<4, 4><<<nrows, 256, 0, st>>>(Mq(q, mb), pitch, n, nrows, x, y); });
    printf("full rows %4d x %d (%5.1f MB)  4 waves/row, 4 loads/lane, 1 trip   : %6.2f us  %5.2f TB/s\n", nrows, n, mb, us, mb / us);
    us = timeit([&](int q) { k_full<4, 2><<<(nrows + 1) / 2, 256, 0, st>>>(Mq(q, mb), pitch, n, nrows, x, y); });
    printf("full rows %4d x %d (%5.1f MB)  2 waves/row, 4 loads/lane x 2 trips : %6.2f us  %5.2f TB/s\n", nrows, n, mb, us, mb / us);
  }
  }
  // an empty launch, for the floor
  { const double us = timeit([&](int q) { k_full<8, 1><<<1, 256, 0, st>>>(M, 2048, 2, 1, x, y); }); printf("one workgroup, nothing to read: %6.2f us\n", us); }
  return 0;
}

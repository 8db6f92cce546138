// sdm_dpr1.hip -- the dense-column unit of an IPM iteration, resident on the device (deninfac.m:58-94):
//   LAD  = L \ Ad(perm,:)                       all dense columns side by side (sparfwslv.m -> fwblkslv.c:150-183)
//   Lden = dpr1fact(LAD, L.d, symLden, smult)   product-form factors of diag(d) + sum_k smult_k p_k p_k'  (dpr1fact.c)
// and, inside every solve (wrapPcg.m:56-59),  fwdpr1 / bwdpr1 (fwdpr1.c:70-90, bwdpr1.c:65-160, auxfwdpr1.c:44-122).
//
// Every one of these is a FIRST-ORDER LINEAR RECURRENCE in one running scalar t:
//   factor  (dodpr1fact, dpr1fact.c:97-135):  t_{j+1} = (p_j^2 + t_j d_j) / d_j = t_j + p_j^2 / d_j            (a prefix sum)
//   forward (fwipr1, auxfwdpr1.c:44-74):      y_i -= t_i p_i ;  t_{i+1} = t_i + y_i beta_i = (1 - p_i beta_i) t_i + y_i^old beta_i
//   backward (bwipr1, bwdpr1.c:65-86):        y_i -= t_i beta_i ;  t_{i-1} = t_i + p_i y_i = (1 - p_i beta_i) t_i + p_i y_i^old
// i.e. t' = a t + b with data-independent composition -- an associative scan.  One workgroup runs the scan over a
// column (chunks per work-item, a log-depth scan of the chunk maps in LDS, fixed shape: deterministic) instead of a
// sequential walk of length m.  The factor kernel handles the case every IPM iteration is in -- all d > 0,
// smult > 0, every pivot stable (dpr1fact.c:97-135 accepts all rows in the first round) -- and raises a flag
// otherwise; the host then runs the general algorithm (sdm_dense.hip: postponed pivots, sorting, dependent rows,
// Lorentz trace columns) on the downloaded LAD and uploads its factors, so the solves stay resident either way.
#include "sdm_plan.h"
#include <algorithm>
#include <cstring>

namespace sdm {

constexpr int DT = 256;                       // work-items of the kernels in this file

// exclusive scan of the affine maps t -> a_i t + b_i over i = 0 .. n-1 (in that order), started from t0:
// calls use(i, t_i) with the state BEFORE element i, for every i, and returns the state after the last element.
// get(i, a, b) must not depend on what use() writes for the same or later elements.  sa / sb: DT doubles of LDS each.
template <class Get, class Use>
__device__ __forceinline__ double affine_scan(int n, double t0, Get get, Use use, double *sa, double *sb) {
  const int tid = threadIdx.x;
  const int C = (n + DT - 1) / DT, lo = min(n, tid * C), hi = min(n, lo + C);
  double A = 1.0, B = 0.0;
  for (int i = lo; i < hi; i++) { double a, b; get(i, a, b); A = a * A; B = a * B + b; }
  sa[tid] = A; sb[tid] = B;
  __syncthreads();
  // inclusive Hillis-Steele scan of the chunk maps: (A2,B2) after (A1,B1) = (A2 A1, A2 B1 + B2)
  for (int off = 1; off < DT; off <<= 1) {
    double A1 = 1.0, B1 = 0.0;
    if (tid >= off) { A1 = sa[tid - off]; B1 = sb[tid - off]; }
    __syncthreads();
    if (tid >= off) { const double A2 = sa[tid], B2 = sb[tid]; sa[tid] = A2 * A1; sb[tid] = A2 * B1 + B2; }
    __syncthreads();
  }
  const double tend = sa[DT - 1] * t0 + sb[DT - 1];
  double t = tid == 0 ? t0 : sa[tid - 1] * t0 + sb[tid - 1];
  __syncthreads();                                                   // everybody has read sa / sb
  for (int i = lo; i < hi; i++) { double a, b; get(i, a, b); use(i, t); t = a * t + b; }
  __syncthreads();
  return tend;
}

// ---- p_k = LAD(dz rows, colperm[k]) for all k; dgat = d(dz rows)
__global__ void __launch_bounds__(DT)
k_dpr1_gather(double *p, double *dgat, const double *lad, const double *d, const int *dzir, const int64_t *dzjc, const int64_t *poff,
              const int *colperm, int m, int nden) {
  const int k = blockIdx.y;
  const int gid = blockIdx.x * DT + threadIdx.x, gs = gridDim.x * DT;
  const int mk = (int)dzjc[k + 1];
  const double *col = lad + (int64_t)colperm[k] * m;
  for (int i = gid; i < mk; i += gs) p[poff[k] + i] = col[dzir[i]];
  if (k == nden - 1) for (int i = gid; i < mk; i += gs) dgat[i] = d[dzir[i]];
}

// ---- rank-1 step k, every pivot accepted in natural order (dpr1fact.c:97-135 with nph2 = 0):
//   t_j = 1/smult + sum_{i<j} p_i^2 / d_i ;  fi_j = p_j^2 + t_j d_j ;  d_j <- fi_j / t_j ;  beta_j = p_j / fi_j
// accepted iff  p_j^2 * max_{i>j} p_i^2 <= (maxu fi_j)^2  for all j, all d_j > 0 and smult > 0; else need_host is raised
// and nothing is written.
__global__ void __launch_bounds__(DT)
k_dpr1_factor(const double *p, double *beta, double *dgat, int mk, double tmul, double maxu, int *need_host) {
  __shared__ double sa[DT], sb[DT], smx[DT];
  __shared__ int bad;
  const int tid = threadIdx.x;
  if (tid == 0) bad = 0;
  const int C = (mk + DT - 1) / DT, lo = min(mk, tid * C), hi = min(mk, lo + C);
  double cm = 0.0;
  int anybad = (tmul > 0.0) ? 0 : 1;
  for (int i = lo; i < hi; i++) { cm = fmax(cm, p[i] * p[i]); if (!(dgat[i] > 0.0)) anybad = 1; }
  smx[tid] = cm;
  __syncthreads();
  if (anybad) bad = 1;
  __syncthreads();
  if (bad) { if (tid == 0) sdm_raise_flag(need_host); return; }
  // mu_j = max_{i>j} p_i^2: the chunks behind this one, then a backward walk through the chunk; kept in beta (scratch
  // until the commit pass overwrites it with the real beta)
  double mu = 0.0;
  for (int q = tid + 1; q < DT; q++) mu = fmax(mu, smx[q]);
  for (int i = hi - 1; i >= lo; i--) { beta[i] = mu; mu = fmax(mu, p[i] * p[i]); }
  __syncthreads();
  const double t0 = 1.0 / tmul;
  auto get = [&](int i, double &a, double &b) { a = 1.0; b = p[i] * p[i] / dgat[i]; };
  // pass 1: stability of every pivot with the scanned t (dpr1fact.c:118-121), no writes
  int unstable = 0;
  auto check = [&](int i, double t) {
    const double pj2 = p[i] * p[i], fij = pj2 + t * dgat[i], sfi = maxu * fij;
    if (!(pj2 * beta[i] <= sfi * sfi)) unstable = 1;
  };
  affine_scan(mk, t0, get, check, sa, sb);
  if (unstable) bad = 1;
  __syncthreads();
  if (bad) { if (tid == 0) sdm_raise_flag(need_host); return; }
  // pass 2: commit (d in place: element i is read by get() right before use() replaces it, within its owner's chunk)
  auto commit = [&](int i, double t) {
    const double pj2 = p[i] * p[i], dj = dgat[i], fij = pj2 + t * dj;
    beta[i] = p[i] / fij;
    dgat[i] = fij / t;
  };
  affine_scan(mk, t0, get, commit, sa, sb);
}

// ---- L(p_k, beta_k)^{-1} applied to a later column y (fwipr1, auxfwdpr1.c:44-74; all nk = mk rows in natural order)
__global__ void __launch_bounds__(DT)
k_dpr1_apply(double *pall, const double *beta, const int64_t *poff, const int *later, int k, int mk) {
  __shared__ double sa[DT], sb[DT];
  const int j = later[blockIdx.x];
  const double *pk = pall + poff[k];
  double *y = pall + poff[j];
  auto get = [&](int i, double &a, double &b) { a = 1.0 - pk[i] * beta[i]; b = y[i] * beta[i]; };
  auto use = [&](int i, double t) { y[i] -= t * pk[i]; };
  affine_scan(mk, 0.0, get, use, sa, sb);
}

// ---- the product-form part of a solve on the resident factor:  y <- bwdpr1(Lden, fwdpr1(Lden, y) ./ Ld), y in the
// factor's (permuted) order; ONE workgroup walks the factors forward, divides, walks them backward.
struct Pr1R {
  const int64_t *dzjc, *betajc, *poff, *permoff;
  const int *dopiv, *pivperm, *dzir;
  const double *beta, *p, *dden;
};
__global__ void __launch_bounds__(DT)
k_pr1_resident(double *y, Pr1R T, int nden, int dznnz, double *fwglob, int use_lds, int with_divide, const double *dsolve, int m) {
  SDM_DYN_SMEM(smem);
  __shared__ double sa[DT], sb[DT], red[DT];
  double *fw = use_lds ? (double *)smem : fwglob;
  const int tid = threadIdx.x;
  for (int i = tid; i < dznnz; i += DT) fw[i] = y[T.dzir[i]];
  __syncthreads();
  for (int k = 0; k < nden; k++) {                                   // fwprodform (fwdpr1.c:70-90)
    const int mk = (int)T.dzjc[k + 1], nk = (int)(T.betajc[k + 1] - T.betajc[k]);
    if (nk < 1) continue;
    const double *pk = T.p + T.poff[k], *bk = T.beta + T.betajc[k];
    const int *perm = T.dopiv[k] ? T.pivperm + T.permoff[k] : nullptr;
    auto get = [&](int i, double &a, double &b) { const int r = perm ? perm[i] : i; a = 1.0 - pk[r] * bk[i]; b = fw[r] * bk[i]; };
    auto use = [&](int i, double t) { const int r = perm ? perm[i] : i; fw[r] -= t * pk[r]; };
    const double t = affine_scan(nk, 0.0, get, use, sa, sb);
    for (int i = nk + tid; i < mk; i += DT) { const int r = perm ? perm[i] : i; fw[r] -= t * pk[r]; }
    __syncthreads();
  }
  if (with_divide) {
    // ./ Ld (wrapPcg.m:57): the rows touched by dense columns here, the rest below
    for (int i = tid; i < dznnz; i += DT) fw[i] /= dsolve[T.dzir[i]];
    __syncthreads();
  }
  for (int k = nden - 1; k >= 0; k--) {                              // bwprodform (bwdpr1.c:140-160)
    const int mk = (int)T.dzjc[k + 1], nk = (int)(T.betajc[k + 1] - T.betajc[k]);
    if (nk < 1) continue;
    const double *pk = T.p + T.poff[k], *bk = T.beta + T.betajc[k];
    const int *perm = T.dopiv[k] ? T.pivperm + T.permoff[k] : nullptr;
    double a0 = 0.0;                                                 // t = p(nk:mk-1)' y(nk:mk-1), fixed-shape reduction
    for (int i = nk + tid; i < mk; i += DT) { const int r = perm ? perm[i] : i; a0 += pk[r] * fw[r]; }
    red[tid] = a0;
    __syncthreads();
    for (int s = DT >> 1; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
    const double t0 = red[0];
    __syncthreads();
    // i = nk-1 .. 0:  y_i -= t beta_i ;  t += p_i y_i   (scan order = decreasing i)
    auto get = [&](int q, double &a, double &b) { const int i = nk - 1 - q, r = perm ? perm[i] : i; a = 1.0 - pk[r] * bk[i]; b = pk[r] * fw[r]; };
    auto use = [&](int q, double t) { const int i = nk - 1 - q, r = perm ? perm[i] : i; fw[r] -= t * bk[i]; };
    affine_scan(nk, t0, get, use, sa, sb);
  }
  // rows not touched by dense columns only see the division
  if (with_divide) {
    for (int i = tid; i < m; i += DT) y[i] /= dsolve[i];
    __syncthreads();
  }
  for (int i = tid; i < dznnz; i += DT) y[T.dzir[i]] = fw[i];
}

// d for the solves with dense columns: Ld of dpr1fact scattered over L.d; deninfac.m:89-94: skipped pivots whose Ld
// is still <= dtol = max(canceltol absd, abstol) (= lb) act as 1
__global__ void k_dden(double *dden, double *dsolve, const double *d, const double *dgat, const int *dzir, int dznnz, const double *lb,
                       const int *pivstat, int m, int phase) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (phase == 0) { if (i < m) dden[i] = d[i]; return; }
  if (phase == 1) { if (i < dznnz) dden[dzir[i]] = dgat[i]; return; }
  if (i < m) { const double v = dden[i]; dsolve[i] = (pivstat[i] == 1 && v <= lb[i]) ? 1.0 : (v > 0.0 ? v : 1.0); }
}

// =========================================================================== host
void dense_set(sdm_plan *P, sdm_int nden, const sdm_int *LADjc, const sdm_int *LADir, const sdm_int *dzjc, const sdm_int *dzir,
               const sdm_int *colperm, const sdm_int *first) {
  DensePlan &D = P->dense;
  CholPlan &C = P->chol;
  const sdm_int m = C.m;
  D = DensePlan();
  if (nden <= 0) return;
  D.nden = nden; D.dznnz = dzjc[nden];
  D.LADjc.assign(LADjc, LADjc + nden + 1); D.LADir.assign(LADir, LADir + LADjc[nden]);
  D.dzjc.assign(dzjc, dzjc + nden + 1); D.dzir.assign(dzir, dzir + D.dznnz);
  D.colperm.assign(colperm, colperm + nden); D.first.assign(first, first + nden);
  D.poff.assign(nden + 1, 0);
  for (sdm_int k = 0; k < nden; k++) D.poff[k + 1] = D.poff[k] + dzjc[k + 1];
  D.pnnz = D.poff[nden];
  D.later.assign(nden, std::vector<int>());
  std::vector<int> flat;
  for (sdm_int k = 0; k < nden; k++)
    for (sdm_int j = k + 1; j < nden; j++)
      if (first[colperm[j]] <= k) D.later[k].push_back((int)j);
  { std::vector<int> v(D.dzir.begin(), D.dzir.end()); D.d_dzir.upload(v); }
  { std::vector<int> v(D.colperm.begin(), D.colperm.end()); D.d_colperm.upload(v); }
  { std::vector<int64_t> v(D.dzjc.begin(), D.dzjc.end()); D.d_dzjc.upload(v); }
  D.d_poff.upload(D.poff);
  D.d_betajc.alloc(nden + 1); D.d_permoff.alloc(nden + 1); D.d_dopiv.alloc(nden); D.d_pivperm.alloc((size_t)std::max<sdm_int>(D.pnnz, 1));
  D.d_later.alloc((size_t)std::max<sdm_int>(nden, 1));
  D.ad.alloc((size_t)(m * nden)); D.lad.alloc((size_t)(m * nden)); D.wvb.alloc((size_t)(C.wsize * nden));
  D.p.alloc((size_t)std::max<sdm_int>(D.pnnz, 1)); D.beta.alloc((size_t)std::max<sdm_int>(D.pnnz, 1));
  D.dgat.alloc((size_t)std::max<sdm_int>(D.dznnz, 1)); D.smult.alloc(nden); D.dden.alloc(m);
  D.need_host.ensure();
  D.active = true;
}

static void upload_factor_tables(sdm_plan *P) {
  DensePlan &D = P->dense;
  SDM_HIP_CHECK(hipMemcpyAsync(D.d_betajc.p, D.betajc.data(), (D.nden + 1) * sizeof(int64_t), hipMemcpyHostToDevice, P->stream));
  SDM_HIP_CHECK(hipMemcpyAsync(D.d_permoff.p, D.permoff.data(), (D.nden + 1) * sizeof(int64_t), hipMemcpyHostToDevice, P->stream));
  SDM_HIP_CHECK(hipMemcpyAsync(D.d_dopiv.p, D.dopiv.data(), D.nden * sizeof(int), hipMemcpyHostToDevice, P->stream));
  SDM_HIP_CHECK(hipStreamSynchronize(P->stream));                    // the tables are host vectors of the plan: stable until the next factor
}

// smult: host, nden values in dense.cols order (deninfac.m:60-62).  The dense columns Ad are in plan buffer "ad".
void dense_factor(sdm_plan *P, const double *smult, double maxuden, int *host_fallback) {
  DensePlan &D = P->dense;
  CholPlan &C = P->chol;
  if (!D.active) throw std::runtime_error("deninfac: no dense columns set (sdm_plan_set_dense)");
  if (!P->factored) throw std::runtime_error("deninfac: no factor resident (sdm_plan_blkchol first)");
  const int m = (int)C.m, nden = (int)D.nden;
  hipStream_t st = P->stream;
  // LAD = L \ Ad(perm,:): all dense columns in one set of launches
  solve_fw_batch(P, D.ad.p, m, D.lad.p, m, D.wvb.p, nden);
  SDM_KLAUNCH(P, k_dpr1_gather, dim3(std::max(1, std::min(64, (m + DT - 1) / DT)), nden), dim3(DT), 0, D.p.p, D.dgat.p, D.lad.p, C.d.p,
              D.d_dzir.p, D.d_dzjc.p, D.d_poff.p, D.d_colperm.p, m, nden);
  *D.need_host.host = 0;
  // all-accepted tables: nk = mk for every factor with smult != 0, no row reordering
  D.betajc.assign(nden + 1, 0); D.permoff.assign(nden + 1, 0); D.dopiv.assign(nden, 0);
  for (int k = 0; k < nden; k++) D.betajc[k + 1] = D.betajc[k] + (smult[D.colperm[k]] != 0.0 ? D.dzjc[k + 1] : 0);
  bool simple = true;
  for (int k = 0; k < nden; k++) if (!(smult[k] > 0.0)) simple = false;          // Lorentz trace columns (smult < 0), empty columns: host
  if (simple) {
    for (int k = 0; k < nden; k++) {
      const int mk = (int)D.dzjc[k + 1];
      SDM_KLAUNCH(P, k_dpr1_factor, dim3(1), dim3(DT), 0, D.p.p + D.poff[k], D.beta.p + D.betajc[k], D.dgat.p, mk, smult[D.colperm[k]], maxuden,
                  D.need_host.dev());
      if (!D.later[k].empty()) {
        SDM_HIP_CHECK(hipMemcpyAsync(D.d_later.p, D.later[k].data(), D.later[k].size() * sizeof(int), hipMemcpyHostToDevice, st));
        SDM_KLAUNCH(P, k_dpr1_apply, dim3((unsigned)D.later[k].size()), dim3(DT), 0, D.p.p, D.beta.p + D.betajc[k], D.d_poff.p, D.d_later.p, k, mk);
      }
    }
  }
  SDM_HIP_CHECK(hipStreamSynchronize(st));
  const bool need_host = !simple || *(volatile int *)D.need_host.host != 0;
  if (host_fallback) *host_fallback = need_host ? 1 : 0;
  if (need_host) {
    // general algorithm on the host (postponed pivots, dependent rows, Lorentz trace columns): LAD and L.d down,
    // the factors up
    std::vector<double> lad((size_t)m * nden), d(m), Xpr;
    SDM_HIP_CHECK(hipMemcpy(lad.data(), D.lad.p, lad.size() * sizeof(double), hipMemcpyDeviceToHost));
    SDM_HIP_CHECK(hipMemcpy(d.data(), C.d.p, (size_t)m * sizeof(double), hipMemcpyDeviceToHost));
    Xpr.resize(D.LADir.size());
    for (int j = 0; j < nden; j++)
      for (sdm_int t = D.LADjc[j]; t < D.LADjc[j + 1]; t++) Xpr[t] = lad[(size_t)j * m + D.LADir[t]];
    std::vector<sdm_int> bj, pp; std::vector<double> be, pv; std::vector<int> ord;
    dpr1fact_host(m, nden, D.LADjc.data(), D.LADir.data(), Xpr.data(), d.data(), D.dzjc.data(), D.dzir.data(), D.colperm.data(),
                  D.first.data(), smult, maxuden, bj, be, pv, pp, ord);
    D.betajc.assign(bj.begin(), bj.end());
    D.dopiv.assign(ord.begin(), ord.end());
    for (int k = 0; k < nden; k++) D.permoff[k + 1] = D.permoff[k] + (ord[k] ? D.dzjc[k + 1] : 0);
    std::vector<int> pp32(std::max<size_t>(pp.size(), 1), 0);
    for (size_t i = 0; i < pp.size(); i++) pp32[i] = (int)pp[i];
    if (!pv.empty()) SDM_HIP_CHECK(hipMemcpy(D.p.p, pv.data(), pv.size() * sizeof(double), hipMemcpyHostToDevice));
    if (!be.empty()) SDM_HIP_CHECK(hipMemcpy(D.beta.p, be.data(), be.size() * sizeof(double), hipMemcpyHostToDevice));
    if (!pp.empty()) SDM_HIP_CHECK(hipMemcpy(D.d_pivperm.p, pp32.data(), pp.size() * sizeof(int), hipMemcpyHostToDevice));
    SDM_HIP_CHECK(hipMemcpy(D.dden.p, d.data(), (size_t)m * sizeof(double), hipMemcpyHostToDevice));
    SDM_KLAUNCH(P, k_dden, dim3((m + 255) / 256), dim3(256), 0, D.dden.p, C.dsolve.p, C.d.p, D.dgat.p, D.d_dzir.p, 0, C.lb.p, C.pivstat.p, m, 2);
  } else {
    SDM_KLAUNCH(P, k_dden, dim3((m + 255) / 256), dim3(256), 0, D.dden.p, C.dsolve.p, C.d.p, D.dgat.p, D.d_dzir.p, (int)D.dznnz, C.lb.p, C.pivstat.p, m, 0);
    if (D.dznnz > 0)                                                    // (dense columns without structural rows: nothing to scatter, and an empty grid is an invalid launch)
      SDM_KLAUNCH(P, k_dden, dim3(((int)D.dznnz + 255) / 256), dim3(256), 0, D.dden.p, C.dsolve.p, C.d.p, D.dgat.p, D.d_dzir.p, (int)D.dznnz, C.lb.p, C.pivstat.p, m, 1);
    SDM_KLAUNCH(P, k_dden, dim3((m + 255) / 256), dim3(256), 0, D.dden.p, C.dsolve.p, C.d.p, D.dgat.p, D.d_dzir.p, (int)D.dznnz, C.lb.p, C.pivstat.p, m, 2);
  }
  upload_factor_tables(P);
  D.factored = true;
}

void dense_prodform(sdm_plan *P, double *y, bool with_divide) {
  DensePlan &D = P->dense;
  CholPlan &C = P->chol;
  Pr1R T;
  T.dzjc = D.d_dzjc.p; T.betajc = D.d_betajc.p; T.poff = D.d_poff.p; T.permoff = D.d_permoff.p; T.dopiv = D.d_dopiv.p;
  T.pivperm = D.d_pivperm.p; T.dzir = D.d_dzir.p; T.beta = D.beta.p; T.p = D.p.p; T.dden = D.dden.p;
  const int use_lds = D.dznnz <= 8192 ? 1 : 0;
  const size_t lds = use_lds ? (size_t)D.dznnz * sizeof(double) : 0;
#ifndef SDM_EMU
  if (lds > 40 * 1024) SDM_HIP_CHECK(hipFuncSetAttribute((const void *)k_pr1_resident, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
#endif
  // (beyond 8192 touched rows the gathered vector lives in dgat, free between two factorisations)
  SDM_KLAUNCH(P, k_pr1_resident, dim3(1), dim3(DT), lds, y, T, (int)D.nden, (int)D.dznnz, D.dgat.p, use_lds, with_divide ? 1 : 0,
              (const double *)C.dsolve.p, (int)C.m);
}

}  // namespace sdm

// sdm_dense.hip -- dense-column handling of the normal equations.
//
// SeDuMi removes dense columns of A from ADA' and re-introduces them as a product of rank-1 factors
//   diag(d) + sum_k smult_k p_k p_k'  =  L_1 ... L_n diag(d_new) L_n' ... L_1',   L_k = I + tril(p_k beta_k', -1)
// (deninfac.m:58-94).  What is here:
//   * sdm_symbfwblk   symbolic pattern of L \ b(perm,:)           (symbfwblk.c:87-263)   host, integers
//   * sdm_finsymbden  column order / first affecting pivot         (finsymbden.c:71-214)  host, integers
//   * sdm_dpr1fact    the product-form factorisation itself        (dpr1fact.c:97-621)    on the DEVICE since round 4 (sdm_dpr1.hip:
//                     k_dpr1_deps, k_dpr1_general -- the whole of dodpr1fact incl. postponed pivots, findnewdep and the sort -- , k_dpr1_apply);
//                     this file keeps the entry point, which uploads the gathered columns, runs those kernels and downloads
//   * sdm_fwdpr1 / sdm_bwdpr1  apply prod_k L_k^{-1} / its transpose  (fwdpr1.c:70-90, bwdpr1.c:65-160,
//                     auxfwdpr1.c:44-122): device kernels -- these run inside every normal-equation solve
//                     (4+ times per iteration, wrapPcg.m:56-59).  Each factor is a first-order recurrence in a
//                     running scalar t; one workgroup walks the factors in order, the data-parallel tails
//                     (rows beyond the last beta) are spread over the workgroup, the recurrence itself runs on
//                     LDS-staged chunks.
#include "../../include/sedumi_hip.h"
#include "sdm_plan.h"
#include <algorithm>
#include <cmath>
#include <cstring>

namespace sdm {

// ===================================================================== symbolic (host)
// first index k in [from, n) with x[k] >= key, x ascending  (intbsearch, sdmauxCmp.c:83-112)
static sdm_int lower_from(const sdm_int *x, sdm_int from, sdm_int n, sdm_int key) {
  return (sdm_int)(std::lower_bound(x + std::min(from, n), x + n, key) - x);
}

void symbfwblk(sdm_int m, const sdm_int *Ljc, const sdm_int *Lir, const sdm_int *perm, sdm_int nsuper,
               const sdm_int *xsuper, sdm_int n, const sdm_int *Bjc, const sdm_int *Bir, std::vector<sdm_int> &Xjc,
               std::vector<sdm_int> &Xir) {
  std::vector<sdm_int> invperm(m), snode(m), xlindx(nsuper + 1);
  for (sdm_int i = 0; i < m; i++) invperm[perm[i]] = i;
  for (sdm_int s = 0; s < nsuper; s++)
    for (sdm_int j = xsuper[s]; j < xsuper[s + 1]; j++) snode[j] = s;
  // compressed subscripts = row list of the first column of every supernode (symbfwblk.c:58-84)
  std::vector<sdm_int> lindx;
  for (sdm_int s = 0; s < nsuper; s++) {
    xlindx[s] = (sdm_int)lindx.size();
    const sdm_int c = xsuper[s];
    lindx.insert(lindx.end(), Lir + Ljc[c], Lir + Ljc[c + 1]);
  }
  xlindx[nsuper] = (sdm_int)lindx.size();
  std::vector<char> seen(nsuper, 0);
  std::vector<sdm_int> from(nsuper, 0);
  Xjc.assign(n + 1, 0);
  Xir.clear();
  for (sdm_int j = 0; j < n; j++) {
    Xjc[j] = (sdm_int)Xir.size();
    for (sdm_int t = Bjc[j]; t < Bjc[j + 1]; t++) {
      sdm_int i = invperm[Bir[t]];                       // b(perm): position of the nonzero in the factor's order
      sdm_int s = snode[i];
      if (seen[s]) { from[s] = std::min(from[s], i); continue; }
      seen[s] = 1; from[s] = i;
      // walk up the supernodal elimination tree (symbfwblk.c:106-130): first row below the diagonal block
      sdm_int w = xsuper[s + 1] - xsuper[s];
      while (xlindx[s] + w < xlindx[s + 1]) {
        i = lindx[xlindx[s] + w];
        s = snode[i];
        if (seen[s]) { from[s] = std::min(from[s], i); break; }
        seen[s] = 1; from[s] = i;
        w = xsuper[s + 1] - xsuper[s];
      }
    }
    for (sdm_int s = 0; s < nsuper; s++)
      if (seen[s]) {
        seen[s] = 0;
        for (sdm_int i = from[s]; i < xsuper[s + 1]; i++) Xir.push_back(i);
      }
  }
  Xjc[n] = (sdm_int)Xir.size();
}

// ===================================================================== fwdpr1 / bwdpr1 (device)
struct Pr1Tab {
  const int64_t *dzjc;      // cumulative row counts: factor k acts on the first dzjc[k+1] entries of the gathered vector
  const int64_t *betajc;    // start of beta_k (0-based), length nden + 1
  const int64_t *poff;      // start of p_k in p, length nden + 1
  const int64_t *permoff;   // start of the row order of factor k in pivperm (valid where dopiv[k])
  const int *dopiv;
  const double *beta, *p;
  const int *pivperm;
};
constexpr int PR1_CH = 512;   // entries of a factor staged in LDS per pass of the sequential recurrence

// y = prod_k L_k^{-1} b   (fwprodform, fwdpr1.c:70-90) on the gathered vector fw (LDS or HBM scratch)
__device__ void pr1_forward(double *fw, const Pr1Tab &T, int nden, double *cp, double *cb, int *ci) {
  const int tid = threadIdx.x, bs = blockDim.x;
  __shared__ double tcarry[2];
  for (int k = 0; k < nden; k++) {
    const int mk = (int)T.dzjc[k + 1], nk = (int)(T.betajc[k + 1] - T.betajc[k]);
    if (nk < 1) continue;                                        // L = I
    const double *pk = T.p + T.poff[k], *bk = T.beta + T.betajc[k];
    const int *perm = T.dopiv[k] ? T.pivperm + T.permoff[k] : nullptr;
    // sequential part: t_i = t_{i-1} + y_{i-1} beta_{i-1};  y_i -= t_i p_i   for i < nk   (fwipr1, auxfwdpr1.c:44-76)
    if (tid == 0) { tcarry[0] = 0.0; }
    for (int c0 = 0; c0 < nk; c0 += PR1_CH) {
      const int cn = min(PR1_CH, nk - c0);
      for (int i = tid; i < cn; i += bs) {
        const int r = perm ? perm[c0 + i] : c0 + i;
        ci[i] = r; cp[i] = pk[r]; cb[i] = bk[c0 + i];
      }
      __syncthreads();
      if (tid == 0) {
        double t = tcarry[0];
        for (int i = 0; i < cn; i++) {
          const int r = ci[i];
          double yi;
          if (c0 + i == 0) yi = fw[r];
          else { yi = fw[r] - t * cp[i]; fw[r] = yi; }
          t += yi * cb[i];                                        // after the last i < nk this is the t of the tail
        }
        tcarry[0] = t;
      }
      __syncthreads();
    }
    // tail rows nk..mk-1: y_r -= t p_r, data parallel
    const double t = tcarry[0];
    for (int i = nk + tid; i < mk; i += bs) { const int r = perm ? perm[i] : i; fw[r] -= t * pk[r]; }
    __syncthreads();
  }
}

// y = (prod_k L_k)^{-T} b   (bwprodform, bwdpr1.c:137-160)
__device__ void pr1_backward(double *fw, const Pr1Tab &T, int nden, double *cp, double *cb, int *ci) {
  const int tid = threadIdx.x, bs = blockDim.x;
  __shared__ double red[1024];
  __shared__ double tcarry[2];
  for (int k = nden - 1; k >= 0; k--) {
    const int mk = (int)T.dzjc[k + 1], nk = (int)(T.betajc[k + 1] - T.betajc[k]);
    if (nk < 1) continue;
    const double *pk = T.p + T.poff[k], *bk = T.beta + T.betajc[k];
    const int *perm = T.dopiv[k] ? T.pivperm + T.permoff[k] : nullptr;
    // t = p(nk:mk-1)' y(nk:mk-1): fixed-shape tree reduction (deterministic)
    double a = 0.0;
    for (int i = nk + tid; i < mk; i += bs) { const int r = perm ? perm[i] : i; a += pk[r] * fw[r]; }
    red[tid] = a;
    __syncthreads();
    for (int s = bs >> 1; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
    if (tid == 0) tcarry[0] = red[0];
    __syncthreads();
    // i = nk-1 .. 0:  y_i -= t beta_i;  t += p_i y_i        (bwipr1, bwdpr1.c:65-90)
    for (int c1 = nk; c1 > 0; c1 -= PR1_CH) {
      const int c0 = max(0, c1 - PR1_CH), cn = c1 - c0;
      for (int i = tid; i < cn; i += bs) {
        const int r = perm ? perm[c0 + i] : c0 + i;
        ci[i] = r; cp[i] = pk[r]; cb[i] = bk[c0 + i];
      }
      __syncthreads();
      if (tid == 0) {
        double t = tcarry[0];
        for (int i = cn - 1; i >= 0; i--) {
          const int r = ci[i];
          const double yi = fw[r] - t * cb[i];
          fw[r] = yi;
          t += cp[i] * yi;
        }
        tcarry[0] = t;
      }
      __syncthreads();
    }
  }
}

__global__ void __launch_bounds__(256)
k_pr1_solve(double *y, const double *b, int m, int dznnz, const int *dzir, Pr1Tab T, int nden, int backward,
            double *scratch, int use_lds) {
  SDM_DYN_SMEM(smem);
  __shared__ double cp[PR1_CH], cb[PR1_CH];
  __shared__ int ci[PR1_CH];
  const int col = blockIdx.x;                                    // one right-hand side per workgroup
  const double *bc = b + (int64_t)col * m;
  double *yc = y + (int64_t)col * m;
  double *fw = use_lds ? (double *)smem : scratch + (int64_t)col * dznnz;
  const int tid = threadIdx.x, bs = blockDim.x;
  for (int i = tid; i < m; i += bs) yc[i] = bc[i];               // y = b outside the rows touched by dense columns
  for (int i = tid; i < dznnz; i += bs) fw[i] = bc[dzir[i]];     // fwork = y(dz.ir)   (fwdpr1.c:187-188)
  __syncthreads();
  if (backward) pr1_backward(fw, T, nden, cp, cb, ci); else pr1_forward(fw, T, nden, cp, cb, ci);
  __syncthreads();
  for (int i = tid; i < dznnz; i += bs) yc[dzir[i]] = fw[i];     // y(dz.ir) = fwork
}

void pr1_solve(bool backward, sdm_int m, sdm_int nrhs, sdm_int nden, const sdm_int *dzjc, const sdm_int *dzir,
               const sdm_int *betajc, const double *beta, const double *p, const sdm_int *pivperm, sdm_int npivperm,
               const int *dopiv, const double *b, double *y) {
  if (nden == 0 || m == 0) { if (y != b) memcpy(y, b, (size_t)(m * nrhs) * sizeof(double)); return; }   // fwdpr1.c:132-135
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) throw std::runtime_error("no HIP device available (libsedumi_hip has no CPU fallback)");
  const sdm_int dznnz = dzjc[nden];
  std::vector<int64_t> poff(nden + 1, 0), permoff(nden + 1, 0), dz64(dzjc, dzjc + nden + 1), bj64(betajc, betajc + nden + 1);
  for (sdm_int k = 0; k < nden; k++) { poff[k + 1] = poff[k] + dzjc[k + 1]; permoff[k + 1] = permoff[k] + (dopiv[k] ? dzjc[k + 1] : 0); }
  if (permoff[nden] > npivperm) throw std::runtime_error("Lden.pivperm is shorter than the reordered columns need");
  DevBuf<int64_t> d_dz, d_bj, d_po, d_pm;
  DevBuf<int> d_dopiv, d_perm, d_dzir;
  DevBuf<double> d_beta, d_p, d_b, d_y, d_scr;
  d_dz.upload(dz64); d_bj.upload(bj64); d_po.upload(poff); d_pm.upload(permoff);
  { std::vector<int> v(dopiv, dopiv + nden); d_dopiv.upload(v); }
  { std::vector<int> v((size_t)std::max<sdm_int>(npivperm, 1), 0); for (sdm_int i = 0; i < npivperm; i++) v[i] = (int)pivperm[i]; d_perm.upload(v); }
  { std::vector<int> v((size_t)std::max<sdm_int>(dznnz, 1), 0); for (sdm_int i = 0; i < dznnz; i++) v[i] = (int)dzir[i]; d_dzir.upload(v); }
  d_beta.upload(beta, (size_t)betajc[nden]); d_p.upload(p, (size_t)poff[nden]);
  d_b.upload(b, (size_t)(m * nrhs)); d_y.alloc((size_t)(m * nrhs));
  const int use_lds = dznnz <= SOLVE_LDS_MAX ? 1 : 0;
  if (!use_lds) d_scr.alloc((size_t)(dznnz * nrhs));
  const size_t lds = use_lds ? (size_t)dznnz * sizeof(double) : 0;
#ifndef SDM_EMU
  if (lds > 48 * 1024) SDM_HIP_CHECK(hipFuncSetAttribute((const void *)k_pr1_solve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
#endif
  Pr1Tab T;
  T.dzjc = d_dz.p; T.betajc = d_bj.p; T.poff = d_po.p; T.permoff = d_pm.p; T.dopiv = d_dopiv.p;
  T.beta = d_beta.p; T.p = d_p.p; T.pivperm = d_perm.p;
  SDM_LAUNCH(k_pr1_solve, dim3((unsigned)nrhs), dim3(256), lds, (hipStream_t)0, d_y.p, d_b.p, (int)m, (int)dznnz, d_dzir.p, T,
             (int)nden, backward ? 1 : 0, d_scr.p, use_lds);
  SDM_HIP_CHECK(hipGetLastError());
  SDM_HIP_CHECK(hipMemcpy(y, d_y.p, (size_t)(m * nrhs) * sizeof(double), hipMemcpyDeviceToHost));
}

}  // namespace sdm

using namespace sdm;
#define SDM_TRY try {
#define SDM_CATCH                                                    \
  }                                                                  \
  catch (const std::exception &e) { set_error(e.what()); return 1; } \
  catch (...) { set_error("unknown error"); return 1; }              \
  return 0;

extern "C" {

int sdm_symbfwblk(sdm_int m, const sdm_int *Ljc, const sdm_int *Lir, const sdm_int *perm, sdm_int nsuper,
                  const sdm_int *xsuper, sdm_int n, const sdm_int *Bjc, const sdm_int *Bir, sdm_int *Xjc, sdm_int *Xir) {
  SDM_TRY
  std::vector<sdm_int> jc, ir;
  symbfwblk(m, Ljc, Lir, perm, nsuper, xsuper, n, Bjc, Bir, jc, ir);
  std::copy(jc.begin(), jc.end(), Xjc);
  if (Xir) std::copy(ir.begin(), ir.end(), Xir);
  SDM_CATCH
}

int sdm_finsymbden(sdm_int m, sdm_int n, const sdm_int *LADjc, const sdm_int *LADir, sdm_int nperm, const sdm_int *perm,
                   const sdm_int *dzjc, const sdm_int *dzir, sdm_int firstq, sdm_int *perm_out, sdm_int *dzjc_out,
                   sdm_int *first_out) {
  SDM_TRY
  const sdm_int lastq = firstq + n - nperm;
  std::vector<sdm_int> invdz(std::max<sdm_int>(m, 1), 0);
  for (sdm_int i = dzjc[0]; i < dzjc[nperm]; i++) invdz[dzir[i]] = i;
  sdm_int inz = 0;
  for (sdm_int i = 0; i < nperm; i++) {                    // attach the Lorentz trace columns (finsymbden.c:166-181)
    const sdm_int j = perm[i];
    perm_out[inz] = j; dzjc_out[inz++] = dzjc[i];
    if (j >= firstq && j < lastq) { perm_out[inz] = nperm + j - firstq; dzjc_out[inz++] = dzjc[i + 1]; }
  }
  if (inz != n) throw std::runtime_error("finsymbden: perm / firstq inconsistent with the number of dense columns");
  dzjc_out[n] = dzjc[nperm];
  for (sdm_int j = 0; j < n; j++) {                        // getfirstpiv (finsymbden.c:71-95)
    if (LADjc[j] < LADjc[j + 1]) {
      sdm_int firstj = invdz[LADir[LADjc[j]]];
      for (sdm_int t = LADjc[j] + 1; t < LADjc[j + 1]; t++) firstj = std::min(firstj, invdz[LADir[t]]);
      first_out[j] = lower_from(dzjc_out + 1, 0, n - 1, firstj + 1);
    } else first_out[j] = n;
  }
  SDM_CATCH
}

int sdm_dpr1fact(sdm_int m, sdm_int n, const sdm_int *Xjc, const sdm_int *Xir, const double *Xpr, double *d,
                 const sdm_int *dzjc, const sdm_int *dzir, const sdm_int *colperm, const sdm_int *first,
                 const double *smult, double maxu, sdm_int *betajc, double *beta, double *p, sdm_int *pivperm,
                 sdm_int *npivperm, sdm_int *dopiv) {
  SDM_TRY
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) throw std::runtime_error("no HIP device available (libsedumi_hip has no CPU fallback)");
  for (sdm_int k = 0; k <= n; k++) betajc[k] = 0;
  *npivperm = 0;
  if (n <= 0) return 0;
  DensePlan D;
  dense_tables(D, m, n, dzjc, dzir, colperm, first);
  const sdm_int dznnz = dzjc[n], pnnz = D.pnnz;
  // the gateway's marshalling (dpr1fact.c:728-752): d(1:dznnz) = lab(dz.ir), p(invrowperm,:) = x(:,colperm)
  std::vector<double> dg((size_t)std::max<sdm_int>(dznnz, 1)), pv((size_t)std::max<sdm_int>(pnnz, 1), 0.0);
  std::vector<sdm_int> invrow((size_t)std::max<sdm_int>(m, 1), -1);
  for (sdm_int i = 0; i < dznnz; i++) { dg[i] = d[dzir[i]]; invrow[dzir[i]] = i; }
  for (sdm_int j = 0; j < n; j++)
    for (sdm_int t = Xjc[colperm[j]]; t < Xjc[colperm[j] + 1]; t++) {
      const sdm_int r = invrow[Xir[t]];
      if (r < 0 || r >= dzjc[j + 1]) throw std::runtime_error("dpr1fact: x has a nonzero outside Lsymb.dz");
      pv[D.poff[j] + r] = Xpr[t];
    }
  hipStream_t st = nullptr;
  SDM_HIP_CHECK(hipMemcpyAsync(D.p.p, pv.data(), (size_t)pnnz * sizeof(double), hipMemcpyHostToDevice, st));
  SDM_HIP_CHECK(hipMemcpyAsync(D.dgat.p, dg.data(), (size_t)dznnz * sizeof(double), hipMemcpyHostToDevice, st));
  dense_prodformfact(nullptr, st, D, smult, maxu);
  dense_fetch_tables(st, D);
  const sdm_int nb = D.betajc[n], np = D.permoff[n];
  for (sdm_int k = 0; k <= n; k++) betajc[k] = D.betajc[k];
  for (sdm_int k = 0; k < n; k++) dopiv[k] = D.dopiv[k];
  if (nb) SDM_HIP_CHECK(hipMemcpy(beta, D.beta.p, (size_t)nb * sizeof(double), hipMemcpyDeviceToHost));
  if (pnnz) SDM_HIP_CHECK(hipMemcpy(p, D.p.p, (size_t)pnnz * sizeof(double), hipMemcpyDeviceToHost));
  std::vector<int> pp((size_t)std::max<sdm_int>(np, 1));
  if (np) SDM_HIP_CHECK(hipMemcpy(pp.data(), D.d_pivperm.p, (size_t)np * sizeof(int), hipMemcpyDeviceToHost));
  for (sdm_int i = 0; i < np; i++) pivperm[i] = pp[i];
  *npivperm = np;
  if (dznnz) SDM_HIP_CHECK(hipMemcpy(dg.data(), D.dgat.p, (size_t)dznnz * sizeof(double), hipMemcpyDeviceToHost));
  for (sdm_int i = 0; i < dznnz; i++) d[dzir[i]] = dg[i];                // lab(dz.ir) = d (dpr1fact.c:779-780)
  SDM_CATCH
}

// Ad = adendotd(dense, d, sparAd, Ablk, blkstart)      adendotd.c:74-127 (gateway :135-232), called by getDAtm.m:45
// For the nq dense Lorentz blocks q(k): column k of the result = sparAd(:,k) (the sparse part's a_i[k]'d[k])
//   + d.q1(q(k)) * (Lorentz-trace column k of dense.A) + sum over the dense norm-bound columns j of block q(k) of
//   d.q2(dencols(j)) * (their column of dense.A), written on the pattern of Ablk (adjc / adir).
// aden = dense.A(:, dense.l+1:end): m x (nq + nden) CSC given by adenjc[nq+nden+1] (may start beyond 0), adenir, adenpr.
// q[nq], dencols[nden] 0-based; blkend[nq] = 0-based one-past-last subscript of block q(k); d2 indexed by the global
// subscript minus firstQ (the gateway passes d2 - firstQ).
int sdm_adendotd(sdm_int m, sdm_int nq, sdm_int nden, const sdm_int *adjc, const sdm_int *adir, double *adpr,
                 const sdm_int *sjc, const sdm_int *sir, const double *spr, const sdm_int *adenjc, const sdm_int *adenir,
                 const double *adenpr, const double *d1, const double *d2, sdm_int firstQ, const sdm_int *q,
                 const sdm_int *dencols, const sdm_int *blkend) {
  SDM_TRY
  std::vector<double> fwork((size_t)std::max<sdm_int>(m, 1), 0.0);
  const sdm_int *aden2jc = adenjc + nq;                     // the norm-bound columns follow the nq trace columns
  sdm_int j = 0, inz = nden > 0 || nq > 0 ? aden2jc[0] : 0;
  for (sdm_int k = 0; k < nq; k++) {
    for (sdm_int i = adjc[k]; i < adjc[k + 1]; i++) fwork[adir[i]] = 0.0;
    for (sdm_int i = sjc[k]; i < sjc[k + 1]; i++) fwork[sir[i]] = spr[i];
    double dj = d1[q[k]];
    for (sdm_int i = adenjc[k]; i < adenjc[k + 1]; i++) fwork[adenir[i]] += dj * adenpr[i];
    for (; j < nden; j++) {
      const sdm_int c = dencols[j];
      if (c >= blkend[k]) break;
      dj = d2[c - firstQ];
      for (; inz < aden2jc[j + 1]; inz++) fwork[adenir[inz]] += dj * adenpr[inz];
    }
    for (sdm_int i = adjc[k]; i < adjc[k + 1]; i++) adpr[i] = fwork[adir[i]];
  }
  SDM_CATCH
}

// smult(norm-bound part) = adenscale(dense, d, blkstart)      adenscale.c:62-80: det(d_k) of the Lorentz block each dense
// norm-bound column belongs to (deninfac.m:61)
int sdm_adenscale(sdm_int nq, sdm_int nden, const double *detd, const sdm_int *q, const sdm_int *dencols, const sdm_int *blkend,
                  double *smult) {
  SDM_TRY
  sdm_int j = 0;
  for (sdm_int k = 0; k < nq; k++) {
    const double detdk = detd[q[k]];
    while (j < nden) {
      if (dencols[j] >= blkend[k]) break;
      smult[j++] = detdk;
    }
  }
  SDM_CATCH
}

static void to_int(const sdm_int *dopiv, sdm_int nden, std::vector<int> &v) { v.resize(std::max<sdm_int>(nden, 1)); for (sdm_int k = 0; k < nden; k++) v[k] = (int)dopiv[k]; }

int sdm_fwdpr1(sdm_int m, sdm_int nrhs, sdm_int nden, const sdm_int *dzjc, const sdm_int *dzir, const sdm_int *betajc,
               const double *beta, const double *p, const sdm_int *pivperm, sdm_int npivperm, const sdm_int *dopiv,
               const double *b, double *y) {
  SDM_TRY
  std::vector<int> dp; to_int(dopiv, nden, dp);
  pr1_solve(false, m, nrhs, nden, dzjc, dzir, betajc, beta, p, pivperm, npivperm, dp.data(), b, y);
  SDM_CATCH
}
int sdm_bwdpr1(sdm_int m, sdm_int nrhs, sdm_int nden, const sdm_int *dzjc, const sdm_int *dzir, const sdm_int *betajc,
               const double *beta, const double *p, const sdm_int *pivperm, sdm_int npivperm, const sdm_int *dopiv,
               const double *b, double *y) {
  SDM_TRY
  std::vector<int> dp; to_int(dopiv, nden, dp);
  pr1_solve(true, m, nrhs, nden, dzjc, dzir, betajc, beta, p, pivperm, npivperm, dp.data(), b, y);
  SDM_CATCH
}

}  // extern "C"

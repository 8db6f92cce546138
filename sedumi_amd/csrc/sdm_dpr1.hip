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
// sequential walk of length m.  The factor kernel (k_dpr1_general) is the whole of dodpr1fact: the data-dependent parts --
// postponed pivots and their sort, dependent rows (d = 0), negative multiples (Lorentz trace columns) -- are scans,
// prefix counts and a bitonic network as well (see the kernel), so nothing of the factorisation runs on the host.
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

// ---- generic block-wide exclusive scan over i = 0 .. n-1 with an associative op (identity `id`): use(i, op-sum of the elements
// before i); returns the total.  Same shape as affine_scan (chunk per work-item, log-depth scan of the chunk sums in LDS).
template <class T, class Get, class Use, class Op>
__device__ __forceinline__ T block_scan(int n, T id, Get get, Use use, Op op, T *sh) {
  const int tid = threadIdx.x;
  const int C = (n + DT - 1) / DT, lo = min(n, tid * C), hi = min(n, lo + C);
  T acc = id;
  for (int i = lo; i < hi; i++) acc = op(acc, get(i));
  sh[tid] = acc;
  __syncthreads();
  for (int off = 1; off < DT; off <<= 1) {
    T prev = id;
    if (tid >= off) prev = sh[tid - off];
    __syncthreads();
    if (tid >= off) sh[tid] = op(prev, sh[tid]);
    __syncthreads();
  }
  const T total = sh[DT - 1];
  T run = tid == 0 ? id : sh[tid - 1];
  __syncthreads();
  for (int i = lo; i < hi; i++) { const T v = get(i); use(i, run); run = op(run, v); }
  __syncthreads();
  return total;
}

// ---- rank-1 step k of the product form, the WHOLE of dodpr1fact (dpr1fact.c:280-477) by one workgroup:
//   factor  diag(d) + tmul p p'  =  L(p, beta) diag(d_new) L(p, beta)'  with  max|L| <= maxu  by choice of the pivot order.
// The reference walks the rows once with a running t and a data-dependent accept / postpone decision per row, sorts the
// postponed rows by decreasing p_j^2 (qsort), walks those, and treats rows with d = 0 ("dependent") through a partition of
// the rows around the largest dependent one.  Here every walk is a scan:
//   * candidate order `ord` (natural order, or the partition [p_j^2 > h | idep | rest] built from prefix counts);
//   * mu_i = max(h, max_{j>i} p_j^2) -- a suffix-max scan;
//   * first round: the decision of row i depends on the decisions before it only through t_i (a prefix sum over the ACCEPTED
//     rows) and muph2_i (a prefix max over the POSTPONED rows).  Start from "all accepted", evaluate all decisions in parallel
//     from the two scans, repeat until no decision changes: after r rounds the first r decisions are the sequential ones,
//     so the fixed point reached IS the sequential result; an interior-point iteration needs one or two rounds;
//   * accepted rows are compacted to the front (prefix count), the postponed ones sorted by (p_j^2 descending, row ascending)
//     with a bitonic network (the documented order of kdsortdec; the reference's own comparator is undefined behaviour,
//     DESIGN.md section 4) and factored by a second prefix-sum scan;
//   * the dependent-row list `dep` (ascending, tail, then the removed dependencies: dpr1fact.c:392-405, findnewdep :495-512)
//     is a handful of entries and is kept by work-item 0.
// Outputs straight into the device-resident factor tables: beta (at betajc[k]), the row order (pivperm at permoff[k]) when
// rows were reordered, dopiv[k], betajc[k+1], permoff[k+1], and the updated gathered diagonal.
struct Dpr1G {
  const double *p;                  // column k (mk entries)
  double *beta_base, *dgat;         // all betas; gathered diagonal (dznnz)
  double *psq, *mu, *key;           // scratch doubles: mk, mk, pow2(mk)
  int *ord, *ord2, *acc, *post;     // scratch ints: mk, mk, mk, pow2(mk)
  int *dep, *st;                    // dependent rows (capacity + 1 entries) and {ndep, maxndep}
  int64_t *betajc, *permoff;
  int *dopiv, *pivperm;
  int k, mk;
  double tmul, maxu;
};
__global__ void __launch_bounds__(DT)
k_dpr1_general(Dpr1G G) {
  __shared__ double sa[DT], sb[DT];
  __shared__ int si[DT];
  __shared__ int s_idep, s_deldep, s_caseB, s_changed;
  __shared__ double s_h;
  const int tid = threadIdx.x, k = G.k, mk = G.mk;
  const double *p = G.p;
  double *d = G.dgat, *psq = G.psq;
  if (G.tmul == 0.0 || mk == 0) {                                    // dpr1fact.c:293-296: beta = 0, L = I
    if (tid == 0) { G.betajc[k + 1] = G.betajc[k]; G.permoff[k + 1] = G.permoff[k]; G.dopiv[k] = 0; }
    return;
  }
  const double t0 = 1.0 / G.tmul;
  for (int i = tid; i < mk; i += DT) psq[i] = p[i] * p[i];
  __syncthreads();
  // ---- dependent rows among 0 .. mk-1 (dpr1fact.c:371-405): work-item 0 keeps the list
  if (tid == 0) {
    int *dep = G.dep;
    int ndep = G.st[0];
    s_caseB = dep[0] < mk ? 1 : 0;
    s_idep = -1; s_deldep = 0; s_h = 0.0;
    if (s_caseB) {
      double psqrdep = 0.0; int jd = 0;
      for (int i = 0; dep[i] < mk; i++) if (psq[dep[i]] > psqrdep) { jd = i; psqrdep = psq[dep[i]]; }
      if (psqrdep > 0.0) {
        s_idep = dep[jd];
        if (t0 > 0.0) {
          s_deldep = 1;
          for (int q = jd; q < ndep; q++) dep[q] = dep[q + 1];        // incl. the tail dep[ndep]
          s_h = G.maxu * G.maxu * psqrdep;
          dep[ndep] = s_idep;                                         // remember the removed dependency
          G.st[0] = ndep - 1;
        } else s_h = psqrdep;                                         // D - p p' should be psd: [0, psqrdep] counts as 0
      } else s_idep = dep[0];
    }
  }
  __syncthreads();
  const int caseB = s_caseB, idep = s_idep, deldep = s_deldep;
  const double h = s_h;
  // ---- candidate order
  int n;
  if (!caseB) {
    for (int i = tid; i < mk; i += DT) G.ord[i] = i;
    n = mk;
    __syncthreads();
  } else {                                                            // perm = [find(psqr > h), idep, remainder (from the back)]
    auto big = [&](int i) { return (i != idep && psq[i] > h) ? 1 : 0; };
    auto plus = [](int a, int b) { return a + b; };
    n = block_scan<int>(mk, 0, big, [&](int i, int before) {
      if (i == idep) return;
      const int skipped = i > idep ? 1 : 0;                           // idep itself is in neither list
      if (psq[i] > h) G.ord[before] = i; else G.ord[mk - 1 - (i - skipped - before)] = i;
    }, plus, si);
    if (tid == 0) G.ord[n] = idep;
    __syncthreads();
  }
  const int *ord = G.ord;
  // ---- mu_i = max(h, max_{j in (i, n)} psqr[ord[j]])  (dpr1fact.c:320-323, 433-436): a scan from the back
  {
    auto fmx = [](double a, double b) { return fmax(a, b); };
    block_scan<double>(n, h, [&](int q) { return psq[ord[n - 1 - q]]; }, [&](int q, double before) { G.mu[n - 1 - q] = before; }, fmx, sa);
  }
  // ---- first round (dpr1fact.c:97-135 / :168-202): decisions to their fixed point
  for (int i = tid; i < n; i += DT) G.acc[i] = 1;
  __syncthreads();
  double tend = t0;
  for (int round = 0; round <= n; round++) {
    if (tid == 0) s_changed = 0;
    // t_i over the accepted rows before i -> key[] (scratch), then muph2_i over the postponed rows before i
    auto getc = [&](int i, double &a, double &b) { const int r = ord[i]; a = 1.0; b = G.acc[i] ? psq[r] / d[r] : 0.0; };
    tend = affine_scan(n, t0, getc, [&](int i, double t) { G.key[i] = t; }, sa, sb);
    auto fmx = [](double a, double b) { return fmax(a, b); };
    int changed = 0;
    block_scan<double>(n, 0.0, [&](int i) { return G.acc[i] ? 0.0 : psq[ord[i]]; }, [&](int i, double muph2) {
      const int r = ord[i];
      const double pj2 = psq[r], fij = pj2 + G.key[i] * d[r], sfi = G.maxu * fij;
      const int ok = (pj2 * fmax(muph2, G.mu[i]) <= sfi * sfi) ? 1 : 0;
      if (ok != G.acc[i]) { G.post[i] = ok; changed = 1; } else G.post[i] = G.acc[i];
    }, fmx, sa);
    if (changed) s_changed = 1;
    __syncthreads();
    const int any = s_changed;
    __syncthreads();
    if (!any) break;
    for (int i = tid; i < n; i += DT) G.acc[i] = G.post[i];
    __syncthreads();
  }
  // ---- commit the accepted rows (t_i in key[]): fi -> psq, d_new; compact: accepted to the front, postponed listed
  auto plus = [](int a, int b) { return a + b; };
  const int nacc = block_scan<int>(n, 0, [&](int i) { return G.acc[i]; }, [&](int i, int before) {
    const int r = ord[i];
    if (G.acc[i]) {
      const double t = G.key[i], dj = d[r], fij = psq[r] + t * dj;
      psq[r] = fij; d[r] = fij / t;
      G.ord2[before] = r;
    } else G.ord2[n - 1 - (i - before)] = i;                          // (position in ord, parked at the back in reverse)
  }, plus, si);
  const int npost = n - nacc;
  double *beta = G.beta_base + G.betajc[k];
  const bool natural = !caseB && npost == 0;                          // (uniform over the workgroup)
  if (natural) {                                                      // natural order, nothing reordered (dpr1fact.c:330-333)
    for (int i = tid; i < mk; i += DT) beta[i] = p[i] / psq[i];
    if (tid == 0) { G.betajc[k + 1] = G.betajc[k] + mk; G.permoff[k + 1] = G.permoff[k]; G.dopiv[k] = 0; }
  } else {
  int *perm = G.pivperm + G.permoff[k];
  for (int q = tid; q < nacc; q += DT) { const int r = G.ord2[q]; perm[q] = r; beta[q] = p[r] / psq[r]; }
  // ---- postponed rows: sort by (p_j^2 descending, row ascending), second round (ph2dpr1fact, dpr1fact.c:224-240)
  int n2 = 1;
  while (n2 < npost) n2 <<= 1;
  for (int q = tid; q < n2; q += DT) {
    if (q < npost) { const int r = ord[G.ord2[n - 1 - q]]; G.post[q] = r; G.key[q] = psq[r]; }
    else { G.post[q] = 0x7fffffff; G.key[q] = -1.0; }                 // (p_j^2 >= 0: padding sorts last)
  }
  __syncthreads();
  for (int kk = 2; kk <= n2; kk <<= 1)
    for (int j = kk >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < n2; i += DT) {
        const int l = i ^ j;
        if (l > i) {
          const double ki = G.key[i], kl = G.key[l];
          const int ri = G.post[i], rl = G.post[l];
          const bool i_first = ki > kl || (ki == kl && ri < rl);     // i belongs before l in the wanted order
          const bool up = (i & kk) == 0;
          if (up ? !i_first : i_first) { G.key[i] = kl; G.key[l] = ki; G.post[i] = rl; G.post[l] = ri; }
        }
      }
      __syncthreads();
    }
  if (npost > 0) {
    auto getp = [&](int q, double &a, double &b) { const int r = G.post[q]; a = 1.0; b = G.key[q] / d[r]; };
    // (two passes: the scan reads d of rows its use() overwrites -- each row occurs once, read by get() before its own use())
    tend = affine_scan(npost, tend, getp, [&](int q, double t) {
      const int r = G.post[q];
      const double dj = d[r], fij = G.key[q] + t * dj;
      d[r] = fij / t; perm[nacc + q] = r; beta[nacc + q] = p[r] / fij;
    }, sa, sb);
  }
  // ---- the dependent row that this update removes is pivoted on last (dpr1fact.c:466-473); the rest keep their places
  if (caseB) {
    for (int i = n + tid; i < mk; i += DT) perm[i] = ord[i];
    if (tid == 0 && deldep) { d[idep] = psq[idep] / tend; beta[n] = 1.0 / p[idep]; }
  }
  if (tid == 0) {
    G.betajc[k + 1] = G.betajc[k] + (caseB ? n + deldep : mk);
    G.permoff[k + 1] = G.permoff[k] + mk;
    G.dopiv[k] = 1;
  }
  }
  // ---- subtracting a rank-1 term (Lorentz trace columns) may bring a removed dependency back (findnewdep, dpr1fact.c:495-512;
  // prodformfact calls it after EVERY column with a negative multiple, :575-576, whichever way dodpr1fact went)
  __syncthreads();
  if (tid == 0 && G.tmul < 0.0) {
    int *dep = G.dep;
    const int ndep = G.st[0], maxndep = G.st[1];
    int i;
    for (i = ndep + 1; i <= maxndep; i++) if (d[dep[i]] <= 0.0) break;
    if (i <= maxndep) {
      const int id2 = dep[i];
      int j = 0;
      while (j < ndep && dep[j] <= id2) j++;                          // first j with dep[j] > idep
      for (int q = i; q > j; q--) dep[q] = dep[q - 1];
      dep[j] = id2;
      G.st[0] = ndep + 1;
    }
  }
}
// the dependent rows of the gathered diagonal, ascending, with the tail (dpr1fact.c:730-742): one workgroup
__global__ void __launch_bounds__(DT)
k_dpr1_deps(const double *dgat, int dznnz, int *dep, int *st, int tail) {
  __shared__ int si[DT];
  auto plus = [](int a, int b) { return a + b; };
  const int nd = block_scan<int>(dznnz, 0, [&](int i) { return dgat[i] <= 0.0 ? 1 : 0; }, [&](int i, int before) { if (dgat[i] <= 0.0) dep[before] = i; }, plus, si);
  if (threadIdx.x == 0) { dep[nd] = tail; st[0] = nd; st[1] = nd; }
}

// ---- L(p_k, beta_k)^{-1} applied to the later columns that overlap it (prodformfact, dpr1fact.c:585-597: fwipr1 / fwipr1o,
// auxfwdpr1.c:44-122): one workgroup per later column, the factor's row order and length read from the device tables
__global__ void __launch_bounds__(DT)
k_dpr1_apply(double *pall, const double *beta_base, const int64_t *poff, const int64_t *betajc, const int64_t *permoff, const int *dopiv,
             const int *pivperm, const int *later, int k, int mk) {
  __shared__ double sa[DT], sb[DT];
  const int nk = (int)(betajc[k + 1] - betajc[k]);
  if (nk < 1) return;
  const int j = later[blockIdx.x];
  const double *pk = pall + poff[k], *bk = beta_base + betajc[k];
  const int *perm = dopiv[k] ? pivperm + permoff[k] : nullptr;
  double *y = pall + poff[j];
  auto get = [&](int i, double &a, double &b) { const int r = perm ? perm[i] : i; a = 1.0 - pk[r] * bk[i]; b = y[r] * bk[i]; };
  auto use = [&](int i, double t) { const int r = perm ? perm[i] : i; y[r] -= t * pk[r]; };
  const double t = affine_scan(nk, 0.0, get, use, sa, sb);
  for (int i = nk + threadIdx.x; i < mk; i += DT) { const int r = perm ? perm[i] : i; y[r] -= t * pk[r]; }
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
// the chol-independent part of the dense-column unit: symbolic tables and the device buffers of the product-form factorisation
void dense_tables(DensePlan &D, sdm_int m, sdm_int nden, const sdm_int *dzjc, const sdm_int *dzir, const sdm_int *colperm, const sdm_int *first) {
  D.nden = nden; D.dznnz = dzjc[nden]; D.mrows = m;
  D.dzjc.assign(dzjc, dzjc + nden + 1); D.dzir.assign(dzir, dzir + D.dznnz);
  D.colperm.assign(colperm, colperm + nden); D.first.assign(first, first + nden);
  D.poff.assign(nden + 1, 0);
  for (sdm_int k = 0; k < nden; k++) D.poff[k + 1] = D.poff[k] + dzjc[k + 1];
  D.pnnz = D.poff[nden];
  // the later columns each factor is applied to (prodformfact, dpr1fact.c:585-597: first affecting pivot <= k)
  D.later_ptr.assign(nden + 1, 0);
  std::vector<int> flat;
  for (sdm_int k = 0; k < nden; k++) {
    for (sdm_int j = k + 1; j < nden; j++)
      if (first[colperm[j]] <= k) flat.push_back((int)j);
    D.later_ptr[k + 1] = (int)flat.size();
  }
  if (flat.empty()) flat.push_back(0);
  D.d_later.upload(flat);
  { std::vector<int> v(D.dzir.begin(), D.dzir.end()); if (v.empty()) v.push_back(0); D.d_dzir.upload(v); }
  { std::vector<int> v(D.colperm.begin(), D.colperm.end()); D.d_colperm.upload(v); }
  { std::vector<int64_t> v(D.dzjc.begin(), D.dzjc.end()); D.d_dzjc.upload(v); }
  D.d_poff.upload(D.poff);
  D.d_betajc.alloc(nden + 1); D.d_permoff.alloc(nden + 1); D.d_dopiv.alloc(nden); D.d_pivperm.alloc((size_t)std::max<sdm_int>(D.pnnz, 1));
  D.p.alloc((size_t)std::max<sdm_int>(D.pnnz, 1)); D.beta.alloc((size_t)std::max<sdm_int>(D.pnnz, 1));
  D.dgat.alloc((size_t)std::max<sdm_int>(D.dznnz, 1));
  size_t n2 = 1;
  while ((sdm_int)n2 < D.dznnz) n2 <<= 1;
  const size_t nz = (size_t)std::max<sdm_int>(D.dznnz, 1);
  D.w_psq.alloc(nz); D.w_mu.alloc(nz); D.w_key.alloc(n2);
  D.w_ord.alloc(nz); D.w_ord2.alloc(nz); D.w_acc.alloc(nz); D.w_post.alloc(n2); D.w_dep.alloc(nz + 2); D.w_st.alloc(2);
}

void dense_set(sdm_plan *P, sdm_int nden, const sdm_int *LADjc, const sdm_int *LADir, const sdm_int *dzjc, const sdm_int *dzir,
               const sdm_int *colperm, const sdm_int *first) {
  DensePlan &D = P->dense;
  CholPlan &C = P->chol;
  const sdm_int m = C.m;
  D = DensePlan();
  if (nden <= 0) return;
  dense_tables(D, m, nden, dzjc, dzir, colperm, first);
  D.LADjc.assign(LADjc, LADjc + nden + 1); D.LADir.assign(LADir, LADir + LADjc[nden]);
  D.ad.alloc((size_t)(m * nden)); D.lad.alloc((size_t)(m * nden)); D.wvb.alloc((size_t)(C.wsize * nden));
  D.smult.alloc(nden); D.dden.alloc(m);
  D.active = true;
}

// Lden = dpr1fact(...) on the device: D.p holds the gathered columns, D.dgat the gathered diagonal (prodformfact, dpr1fact.c:549-621).
// One launch per rank-1 step (k_dpr1_general: the whole of dodpr1fact incl. postponed pivots, dependent rows and negative
// multiples) and one for its application to the later columns; nothing is read back between them -- lengths, row orders and
// flags of the factors are device tables (d_betajc, d_permoff, d_dopiv, d_pivperm).  P (may be null): launches are timed with it.
void dense_prodformfact(sdm_plan *P, hipStream_t st, DensePlan &D, const double *smult, double maxu) {
  const int nden = (int)D.nden;
#define DPR1_LAUNCH(kernel, grid, ...)                                                            \
  do { if (P) SDM_KLAUNCH_ON(P, st, kernel, grid, dim3(DT), 0, __VA_ARGS__); else SDM_LAUNCH(kernel, grid, dim3(DT), 0, st, __VA_ARGS__); } while (0)
  SDM_HIP_CHECK(hipMemsetAsync(D.d_betajc.p, 0, sizeof(int64_t), st));
  SDM_HIP_CHECK(hipMemsetAsync(D.d_permoff.p, 0, sizeof(int64_t), st));
  DPR1_LAUNCH(k_dpr1_deps, dim3(1), (const double *)D.dgat.p, (int)D.dznnz, D.w_dep.p, D.w_st.p, (int)D.mrows);
  for (int k = 0; k < nden; k++) {
    Dpr1G G;
    G.p = D.p.p + D.poff[k]; G.beta_base = D.beta.p; G.dgat = D.dgat.p;
    G.psq = D.w_psq.p; G.mu = D.w_mu.p; G.key = D.w_key.p;
    G.ord = D.w_ord.p; G.ord2 = D.w_ord2.p; G.acc = D.w_acc.p; G.post = D.w_post.p; G.dep = D.w_dep.p; G.st = D.w_st.p;
    G.betajc = D.d_betajc.p; G.permoff = D.d_permoff.p; G.dopiv = D.d_dopiv.p; G.pivperm = D.d_pivperm.p;
    G.k = k; G.mk = (int)D.dzjc[k + 1]; G.tmul = smult[D.colperm[k]]; G.maxu = maxu;
    DPR1_LAUNCH(k_dpr1_general, dim3(1), G);
    const int nl = D.later_ptr[k + 1] - D.later_ptr[k];
    if (nl > 0 && G.tmul != 0.0)
      DPR1_LAUNCH(k_dpr1_apply, dim3((unsigned)nl), D.p.p, (const double *)D.beta.p, (const int64_t *)D.d_poff.p, (const int64_t *)D.d_betajc.p,
                  (const int64_t *)D.d_permoff.p, (const int *)D.d_dopiv.p, (const int *)D.d_pivperm.p, (const int *)(D.d_later.p + D.later_ptr[k]), k, G.mk);
  }
#undef DPR1_LAUNCH
  SDM_HIP_CHECK(hipGetLastError());
  D.tables_on_host = false;
}
// host copies of the factor tables (lengths, reordered flags) after a factorisation: for sdm_plan_lden / the stateless entry
void dense_fetch_tables(hipStream_t st, DensePlan &D) {
  if (D.tables_on_host) return;
  const sdm_int nden = D.nden;
  D.betajc.assign(nden + 1, 0); D.permoff.assign(nden + 1, 0); D.dopiv.assign(nden, 0);
  SDM_HIP_CHECK(hipMemcpyAsync(D.betajc.data(), D.d_betajc.p, (nden + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, st));
  SDM_HIP_CHECK(hipMemcpyAsync(D.permoff.data(), D.d_permoff.p, (nden + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, st));
  SDM_HIP_CHECK(hipMemcpyAsync(D.dopiv.data(), D.d_dopiv.p, nden * sizeof(int), hipMemcpyDeviceToHost, st));
  SDM_HIP_CHECK(hipStreamSynchronize(st));
  D.tables_on_host = true;
}

// smult: host, nden values in dense.cols order (deninfac.m:60-62).  The dense columns Ad are in plan buffer "ad".
void dense_factor(sdm_plan *P, const double *smult, double maxuden, int *host_fallback) {
  DensePlan &D = P->dense;
  CholPlan &C = P->chol;
  if (!D.active) throw std::runtime_error("deninfac: no dense columns set (sdm_plan_set_dense)");
  if (!P->factored) throw std::runtime_error("deninfac: no factor resident (sdm_plan_blkchol first)");
  const int m = (int)C.m, nden = (int)D.nden;
  // LAD = L \ Ad(perm,:): all dense columns in one set of launches
  solve_fw_batch(P, D.ad.p, m, D.lad.p, m, D.wvb.p, nden);
  SDM_KLAUNCH(P, k_dpr1_gather, dim3(std::max(1, std::min(64, (m + DT - 1) / DT)), nden), dim3(DT), 0, D.p.p, D.dgat.p, D.lad.p, C.d.p,
              D.d_dzir.p, D.d_dzjc.p, D.d_poff.p, D.d_colperm.p, m, nden);
  dense_prodformfact(P, P->stream, D, smult, maxuden);
  if (host_fallback) *host_fallback = 0;                               // (kept in the ABI: there is no host algorithm any more)
  SDM_KLAUNCH(P, k_dden, dim3((m + 255) / 256), dim3(256), 0, D.dden.p, C.dsolve.p, C.d.p, D.dgat.p, D.d_dzir.p, (int)D.dznnz, C.lb.p, C.pivstat.p, m, 0);
  if (D.dznnz > 0)                                                    // (dense columns without structural rows: nothing to scatter, and an empty grid is an invalid launch)
    SDM_KLAUNCH(P, k_dden, dim3(((int)D.dznnz + 255) / 256), dim3(256), 0, D.dden.p, C.dsolve.p, C.d.p, D.dgat.p, D.d_dzir.p, (int)D.dznnz, C.lb.p, C.pivstat.p, m, 1);
  SDM_KLAUNCH(P, k_dden, dim3((m + 255) / 256), dim3(256), 0, D.dden.p, C.dsolve.p, C.d.p, D.dgat.p, D.d_dzir.p, (int)D.dznnz, C.lb.p, C.pivstat.p, m, 2);
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

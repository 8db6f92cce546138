// sdm_chol.hip -- numeric supernodal LDL' (blkchol) and triangular solves
// (fwblkslv / bwblkslv) for gfx950.
//
// What the reference does (blkchol.c:157-231 spchol, blkchol2.c:464-563 blkLDL,
// :346-420 precorrect, :96-167 cholonBlk): a sequential left-looking supernodal
// LDL' built from level-1 BLAS calls, with a never-fail pivot rule.
//
// What this file does instead (MI355X-first): a level-scheduled MULTIFRONTAL
// LDL'.  Every supernode owns a dense column-major front in one HBM arena;
// independent fronts of an elimination-tree level are factored by the same
// launches; children pass their Schur complements to the parent by a
// deterministic, ownership-partitioned extend-add (no atomics).  Inside a
// front the elimination is blocked: a 32-column diagonal block is factored in
// LDS (pivot rule applied column by column), the panel below is solved one
// row per work-item, and the trailing update  C -= L21*D*L21'  runs on the FP64
// matrix cores (v_mfma_f64_16x16x4_f64, 64x64 tile per 4-wave workgroup).
// The result is the same L, d (unit diagonal stored explicitly, skipped
// columns returned as unit vectors, blkchol.c:409-414) up to rounding, and the
// pivot DECISIONS follow blkchol2.c:114-161 including the idamax quirk of
// maxabs (blkchol2.c:66-70).
#include "sdm_plan.h"
#include <algorithm>
#include <cmath>
#include <numeric>

namespace sdm {

// ============================================================ host analysis
void chol_build(sdm_plan *P, sdm_int m, const sdm_int *Ljc, const sdm_int *Lir, const sdm_int *perm,
                sdm_int nsuper, const sdm_int *xsuper, const sdm_int *ADAjc, const sdm_int *ADAir) {
  CholPlan &C = P->chol;
  C.m = m; C.nsuper = nsuper; C.nnzL = Ljc[m]; C.nnzADA = ADAjc[m];
  if (C.nnzADA >= (sdm_int)1 << 31) throw std::runtime_error("nnz(ADA) >= 2^31 not supported");
  C.Ljc.assign(Ljc, Ljc + m + 1);
  C.perm.assign(perm, perm + m);
  std::vector<int> snode(m);
  C.sn_first.resize(nsuper); C.sn_ns.resize(nsuper); C.sn_ms.resize(nsuper);
  C.sn_parent.assign(nsuper, -1); C.sn_level.assign(nsuper, 0);
  C.sn_foff.resize(nsuper); C.sn_xl.resize(nsuper); C.sn_woff.resize(nsuper); C.sn_roff.assign(nsuper, 0);
  C.sn_toff.resize(nsuper);
  int64_t foff = 0, xl = 0, toff = 0;
  C.maxms = 0; C.maxns = 0;
  for (sdm_int s = 0; s < nsuper; s++) {
    sdm_int f = xsuper[s], n = xsuper[s + 1] - f, ms = Ljc[f + 1] - Ljc[f];
    if (n <= 0 || ms < n) throw std::runtime_error("bad supernode partition");
    for (sdm_int j = f; j < f + n; j++) {
      snode[j] = (int)s;
      if (Ljc[j + 1] - Ljc[j] != ms - (j - f)) throw std::runtime_error("L.L columns are not nested within a supernode");
    }
    C.sn_first[s] = (int)f; C.sn_ns[s] = (int)n; C.sn_ms[s] = (int)ms;
    C.sn_foff[s] = foff; C.sn_xl[s] = xl; C.sn_woff[s] = xl; C.sn_toff[s] = toff;
    foff += (int64_t)ms * ms; xl += ms; toff += (int64_t)ms * n;
    C.maxms = std::max(C.maxms, (int)ms); C.maxns = std::max(C.maxns, (int)n);
  }
  C.fsize = foff; C.wsize = xl; C.tsize = toff;
  // compressed subscripts (row list of the first column of every supernode)
  std::vector<int> lindx((size_t)xl);
  for (sdm_int s = 0; s < nsuper; s++) {
    const sdm_int *r = Lir + Ljc[C.sn_first[s]];
    for (int i = 0; i < C.sn_ms[s]; i++) lindx[C.sn_xl[s] + i] = (int)r[i];
  }
  // supernodal etree: parent = supernode of the first row below the block
  for (sdm_int s = 0; s < nsuper; s++)
    if (C.sn_ms[s] > C.sn_ns[s]) C.sn_parent[s] = snode[lindx[C.sn_xl[s] + C.sn_ns[s]]];
  C.childptr.assign(nsuper + 1, 0);
  for (sdm_int s = 0; s < nsuper; s++) if (C.sn_parent[s] >= 0) C.childptr[C.sn_parent[s] + 1]++;
  for (sdm_int s = 0; s < nsuper; s++) C.childptr[s + 1] += C.childptr[s];
  C.childlist.resize(C.childptr[nsuper]);
  { std::vector<int> pos(C.childptr.begin(), C.childptr.end() - 1);
    for (sdm_int s = 0; s < nsuper; s++) if (C.sn_parent[s] >= 0) C.childlist[pos[C.sn_parent[s]]++] = (int)s; }
  // levels (parents have larger indices than children: postordered)
  int nlev = 0;
  for (sdm_int s = 0; s < nsuper; s++) {
    int p = C.sn_parent[s];
    if (p >= 0) { if (p <= s) throw std::runtime_error("supernodes not postordered"); C.sn_level[p] = std::max(C.sn_level[p], C.sn_level[s] + 1); }
    nlev = std::max(nlev, C.sn_level[s] + 1);
  }
  C.nlevels = nlev;
  C.levptr.assign(nlev + 1, 0);
  for (sdm_int s = 0; s < nsuper; s++) C.levptr[C.sn_level[s] + 1]++;
  for (int l = 0; l < nlev; l++) C.levptr[l + 1] += C.levptr[l];
  C.levlist.resize(nsuper);
  { std::vector<int> pos(C.levptr.begin(), C.levptr.end() - 1);
    for (sdm_int s = 0; s < nsuper; s++) C.levlist[pos[C.sn_level[s]]++] = (int)s; }
  for (int l = 0; l < nlev; l++)
    std::stable_sort(C.levlist.begin() + C.levptr[l], C.levlist.begin() + C.levptr[l + 1],
                     [&](int a, int b) { return C.sn_ns[a] > C.sn_ns[b]; });
  // relative indices child rows -> parent front rows
  std::vector<int> relidx;
  { std::vector<int> posmap(m, -1);
    for (sdm_int p = 0; p < nsuper; p++) {
      if (C.childptr[p + 1] == C.childptr[p]) continue;
      for (int i = 0; i < C.sn_ms[p]; i++) posmap[lindx[C.sn_xl[p] + i]] = i;
      for (int ci = C.childptr[p]; ci < C.childptr[p + 1]; ci++) {
        int c = C.childlist[ci];
        C.sn_roff[c] = (int64_t)relidx.size();
        for (int i = C.sn_ns[c]; i < C.sn_ms[c]; i++) {
          int q = posmap[lindx[C.sn_xl[c] + i]];
          if (q < 0) throw std::runtime_error("child structure not contained in parent structure");
          relidx.push_back(q);
        }
      }
    }
  }
  // permuteP map (blkchol.c:95-120): L slot -> ADA value index / front offset
  std::vector<int> asm_src((size_t)C.nnzL);
  std::vector<int64_t> asm_dst((size_t)C.nnzL), asm_dstT((size_t)C.nnzL);
  { std::vector<int> rowpos(m, -1);
    for (sdm_int j = 0; j < m; j++) {
      sdm_int jc = perm[j];
      for (sdm_int t = ADAjc[jc]; t < ADAjc[jc + 1]; t++) rowpos[ADAir[t]] = (int)t;
      int s = snode[j]; int c = (int)(j - C.sn_first[s]); int64_t ms = C.sn_ms[s];
      for (sdm_int t = Ljc[j]; t < Ljc[j + 1]; t++) {
        asm_src[t] = rowpos[perm[Lir[t]]];
        asm_dst[t] = C.sn_foff[s] + (int64_t)c * ms + c + (t - Ljc[j]);
        asm_dstT[t] = C.sn_toff[s] + (int64_t)(c + (t - Ljc[j])) * C.sn_ns[s] + c;     // L^T panel: (row r) * n_s + c
      }
      for (sdm_int t = ADAjc[jc]; t < ADAjc[jc + 1]; t++) rowpos[ADAir[t]] = -1;
    }
  }
  // factor launch schedule
  C.launches.clear(); C.lev_first_launch.assign(nlev + 1, 0); C.lev_T.assign(nlev, 1);
  for (int l = 0; l < nlev; l++) {
    C.lev_first_launch[l] = (int)C.launches.size();
    int b = C.levptr[l], e = C.levptr[l + 1];
    int maxns = C.sn_ns[C.levlist[b]], maxms = 0;
    for (int i = b; i < e; i++) maxms = std::max(maxms, C.sn_ms[C.levlist[i]]);
    C.lev_T[l] = std::max(1, std::min(128, maxms / 16));
    for (int p = 0; p * NB < maxns; p++) {
      LevelLaunch L; L.level = l; L.panel = p; L.nactive = 0; L.maxrows = 0; L.maxtiles = 0;
      for (int i = b; i < e; i++) {
        int s = C.levlist[i];
        if (C.sn_ns[s] <= p * NB) break;
        L.nactive++;
        int kb = std::min(NB, C.sn_ns[s] - p * NB);
        int rows = C.sn_ms[s] - (p * NB + kb);
        L.maxrows = std::max(L.maxrows, rows);
        int nt = (rows + TILE - 1) / TILE;
        L.maxtiles = std::max(L.maxtiles, nt * (nt + 1) / 2);
      }
      C.launches.push_back(L);
    }
  }
  C.lev_first_launch[nlev] = (int)C.launches.size();
  // upload
  C.d_first.upload(C.sn_first); C.d_ns.upload(C.sn_ns); C.d_ms.upload(C.sn_ms); C.d_parent.upload(C.sn_parent);
  C.d_childptr.upload(C.childptr); C.d_childlist.upload(C.childlist); C.d_levlist.upload(C.levlist);
  C.d_lindx.upload(lindx); C.d_relidx.upload(relidx);
  { std::vector<int> p32(m); for (sdm_int i = 0; i < m; i++) p32[i] = (int)perm[i]; C.d_perm.upload(p32); }
  C.d_foff.upload(C.sn_foff); C.d_xl.upload(C.sn_xl); C.d_woff.upload(C.sn_woff); C.d_roff.upload(C.sn_roff);
  C.d_asm_src.upload(asm_src); C.d_asm_dst.upload(asm_dst); C.d_asm_dstT.upload(asm_dstT); C.d_toff.upload(C.sn_toff);
  C.frontsT.alloc((size_t)C.tsize);
  { std::vector<int64_t> l64(C.Ljc.begin(), C.Ljc.end()); C.d_Ljc.upload(l64); }
  C.fronts.alloc((size_t)C.fsize); C.wvec.alloc((size_t)C.wsize); C.colbuf.alloc((size_t)C.wsize + (size_t)nsuper);
  C.d.alloc(m); C.dsolve.alloc(m); C.lb.alloc(m); C.pivval.alloc(m); C.pivstat.alloc(m); C.ub.alloc(2);
  P->ada_val.alloc((size_t)C.nnzADA); P->absd.alloc(m); P->lpr.alloc((size_t)C.nnzL);
  P->rhs.alloc(m); P->y.alloc(m); P->ywork.alloc(m);
  P->has_chol = true; P->factored = false;
}

// ================================================================= kernels
struct FrontTab {
  const int *first, *ns, *ms;
  const int64_t *foff, *xl, *woff, *roff, *toff;
  const int *childptr, *childlist, *lindx, *relidx;
};

// ---- permuteP: scatter tril(ADA(perm,perm)) into the (zeroed) fronts
__global__ void k_assemble(double *F, const double *ada, const int *src, const int64_t *dst, int64_t nnzL) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; t < nnzL; t += stride) { int s = src[t]; F[dst[t]] = s < 0 ? 0.0 : ada[s]; }
}
__global__ void k_extract(double *Lpr, const double *F, const int64_t *dst, int64_t nnzL) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; t < nnzL; t += stride) Lpr[t] = F[dst[t]];
}
__global__ void k_load_factor(double *F, double *FT, const double *Lpr, const int64_t *dst, const int64_t *dstT, int64_t nnzL) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; t < nnzL; t += stride) { const double v = Lpr[t]; F[dst[t]] = v; FT[dstT[t]] = v; }
}

// ---- pivot thresholds (blkchol.c:168-184): one workgroup.
//   ub = max_j P(perm_j,perm_j) / maxu^2 ;  lb_j = max(abstol, canceltol * orgd_j)
__global__ void k_prep_pivots(int m, const double *ada, const int *asm_src, const int64_t *Ljc, const int *perm,
                              const double *absd, int use_absd, double canceltol, double maxu, double abstol,
                              double *lb, double *ub, int *pivstat, double *pivval) {
  __shared__ double red[256];
  double mx = 0.0;
  for (int j = threadIdx.x; j < m; j += blockDim.x) {
    int s = asm_src[Ljc[j]];
    double dj = s < 0 ? 0.0 : ada[s];
    if (dj > mx) mx = dj;
    double org = use_absd ? absd[perm[j]] : dj;
    double v = canceltol * org;
    lb[j] = v > abstol ? v : abstol;
    pivstat[j] = 0; pivval[j] = 0.0;
  }
  red[threadIdx.x] = mx;
  __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s && red[threadIdx.x + s] > red[threadIdx.x]) red[threadIdx.x] = red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) { ub[0] = red[0] / (maxu * maxu); ub[1] = maxu; }
}

// ---- extend-add: parent front += children's Schur complements.
// grid (T, fronts of the level).  Workgroup `slice` owns the parent columns
// J with J % T == slice, so every parent entry has exactly one writer and the
// children are applied in a fixed order: deterministic, no atomics.
__global__ void k_extend_add(double *F, FrontTab tab, const int *list) {
  const int p = list[blockIdx.y];
  const int slice = blockIdx.x, T = gridDim.x;
  const int msp = tab.ms[p];
  double *Fp = F + tab.foff[p];
  for (int ci = tab.childptr[p]; ci < tab.childptr[p + 1]; ci++) {
    const int c = tab.childlist[ci];
    const int nc = tab.ns[c], mc = tab.ms[c], mu = mc - nc;
    const int *rel = tab.relidx + tab.roff[c];
    const double *Fc = F + tab.foff[c];
    for (int j = 0; j < mu; j++) {
      const int J = rel[j];
      if (J % T != slice) continue;
      const double *src = Fc + (int64_t)(nc + j) * mc + nc;
      double *dst = Fp + (int64_t)J * msp;
      for (int i = j + threadIdx.x; i < mu; i += blockDim.x) dst[rel[i]] += src[i];
    }
    __syncthreads();
  }
}

// ---- rare path of the pivot rule: value that the reference's maxabs() reads
// for column k of the current panel, i.e. x[idamax+1-based] (blkchol2.c:66-70,
// 121-131).  Column storage order = front rows below the diagonal.  All
// threads of the workgroup call this (uniform).  S = diagonal block in LDS
// (columns < k already final), rows below the block are obtained by forward
// substitution against those columns.  cb = scratch of >= ms+1 doubles.
__device__ double pivot_probe(const double (*S)[NB + 1], int k, int kb, int k0, int ns, int ms, int first,
                              const double *Fs, const double *d, double *cb, double next_raw_diag,
                              double *red_v, int *red_i) {
  SDM_FP_STRICT;   // no FMA contraction: the pivot decisions must see the reference's mul-then-subtract rounding
  const int tid = threadIdx.x, bs = blockDim.x;
  const int len = ms - (k0 + k) - 1;          // entries below the diagonal of this column
  const int nin = kb - k - 1;                 // of which inside the LDS block
  // gather the column into cb[0..len-1]; cb[len] = what lies after the column in L's storage
  for (int i = tid; i < nin; i += bs) cb[i] = S[k + 1 + i][k];
  for (int r = k0 + kb + tid; r < ms; r += bs) {
    double x[NB];
    double diagacc = 0.0;
    for (int c = 0; c <= k; c++) {
      double v = Fs[(int64_t)(k0 + c) * ms + r];
      for (int j = 0; j < c; j++) v -= x[j] * S[c][j];
      double dc = (c < k) ? d[first + k0 + c] : 1.0;
      x[c] = (dc > 0.0) ? v : 0.0;
      if (c < k && dc > 0.0) diagacc += x[c] * (x[c] / dc);
    }
    cb[nin + (r - (k0 + kb))] = x[k];
    if (r == k0 + kb && nin == 0 && k0 + k + 1 < ns)   // next column = first row below the block
      cb[len] = Fs[(int64_t)r * ms + r] - diagacc;
  }
  if (tid == 0) {
    if (k0 + k + 1 >= ns) cb[len] = next_raw_diag;      // next column lives in the next supernode: untouched so far
    else if (nin > 0) cb[len] = S[k + 1][k + 1];
  }
  __syncthreads();
  // first index of maximum |.| (Fortran IDAMAX semantics)
  double bv = -1.0; int bi = 0x7fffffff;
  for (int i = tid; i < len; i += bs) { double a = fabs(cb[i]); if (a > bv) { bv = a; bi = i; } }
  red_v[tid] = bv; red_i[tid] = bi;
  __syncthreads();
  for (int s = bs / 2; s > 0; s >>= 1) {
    if (tid < s) {
      double ov = red_v[tid + s]; int oi = red_i[tid + s];
      if (ov > red_v[tid] || (ov == red_v[tid] && oi < red_i[tid])) { red_v[tid] = ov; red_i[tid] = oi; }
    }
    __syncthreads();
  }
  const int imax = red_i[0];
  const double val = fabs(cb[imax + 1]);      // 1-based index used as 0-based: the element AFTER the max
  __syncthreads();
  return val;
}

// ---- K1: LDL' of the kb x kb diagonal block of panel p (one workgroup per front)
__global__ void __launch_bounds__(256)
k_ldl_diag(double *F, double *FT, FrontTab tab, const int *list, int panel, double *d, double *lb, const double *ubp,
           int *pivstat, double *pivval, double *colbuf, const double *ada, const int *asm_src,
           const int64_t *Ljc, int mtot) {
  SDM_FP_STRICT;   // no FMA contraction: the pivot decisions must see the reference's mul-then-subtract rounding
  __shared__ double S[NB][NB + 1];
  __shared__ double lcol[NB];
  __shared__ double red_v[256];
  __shared__ int red_i[256];
  const int s = list[blockIdx.x];
  const int ns = tab.ns[s], ms = tab.ms[s], first = tab.first[s];
  const int k0 = panel * NB, kb = min(NB, ns - k0);
  double *Fs = F + tab.foff[s];
  double *Ts = FT + tab.toff[s];
  double *cb = colbuf + tab.woff[s] + s;
  const int tid = threadIdx.x;
  const double ub = ubp[0], maxu = ubp[1];
  // ---- fast path: one wavefront, lane i owns row i of the block in registers; column k's pivot and the
  // unscaled column entries travel by v_readlane broadcasts -- no LDS traffic, no barrier in the k-loop.
  // It gives up (nothing written) as soon as a pivot needs the column probe of the never-fail rule, which
  // is then handled by the general LDS path below.
  __shared__ int fast_ok;
  if (tid < 64) {
    const int i = tid;
    double x[NB];
#pragma unroll
    for (int j = 0; j < NB; j++) x[j] = (j <= i && i < kb) ? Fs[(int64_t)(k0 + j) * ms + k0 + i] : 0.0;
    const double mylb = i < kb ? lb[first + k0 + i] : 0.0;
    double dval = 0.0, pval = 0.0;
    int stat = 0;
    bool ok = true;
#pragma unroll
    for (int k = 0; k < NB; k++) {
      if (k < kb && ok) {
        const double xkk = sdm_bcast_lane(x[k], k);
        const double lbk = sdm_bcast_lane(mylb, k);
        if (xkk > lbk) {
          if (ms - (k0 + k) > 1 && xkk < ub) {
            ok = false;
          } else {
            const double l = x[k] / xkk;               // lanes i > k: l_ik
#pragma unroll
            for (int j = k + 1; j < NB; j++) {
              if (j < kb) {
                // x(i,j) -= (x(j,k)/xkk) * x(i,k): scaled multiplier of column j times the unscaled own entry,
                // the operand order of cholonBlk (blkchol2.c:141-146) so that noise-level pivots round alike
                const double ljk = sdm_bcast_lane(l, j);
                if (i >= j) x[j] -= ljk * x[k];
              }
            }
            if (i > k) x[k] = l;
            if (i == k) { dval = xkk; stat = 0; }
          }
        } else {
          if (i > k) x[k] = 0.0;
          if (i == k) { dval = 0.0; stat = 1; pval = xkk; }
        }
      }
    }
    if (tid == 0) fast_ok = ok ? 1 : 0;
    if (ok && i < kb) {
#pragma unroll
      for (int j = 0; j < NB; j++) {
        if (j <= i) {
          const double v = (j == i) ? 1.0 : x[j];
          Fs[(int64_t)(k0 + j) * ms + k0 + i] = v;
          Ts[(int64_t)(k0 + i) * ns + k0 + j] = v;
        }
      }
      const int gk = first + k0 + i;
      d[gk] = dval;
      if (stat) { pivstat[gk] = 1; pivval[gk] = pval; }
    }
  }
  __syncthreads();
  if (fast_ok) return;
  for (int idx = tid; idx < kb * kb; idx += blockDim.x) {
    int i = idx % kb, j = idx / kb;
    if (i >= j) S[i][j] = Fs[(int64_t)(k0 + j) * ms + k0 + i];
  }
  __syncthreads();
  for (int k = 0; k < kb; k++) {
    const int gk = first + k0 + k;
    double xkk = S[k][k];
    const double lbk = lb[gk];
    const bool accept = xkk > lbk;
    if (accept) {
      const int mrem = ms - (k0 + k);
      if (mrem > 1 && xkk < ub) {
        double nraw = 0.0;
        if (k0 + k + 1 >= ns && first + ns < mtot) { int sidx = asm_src[Ljc[first + ns]]; nraw = sidx < 0 ? 0.0 : ada[sidx]; }
        const double ubk = pivot_probe(S, k, kb, k0, ns, ms, first, Fs, d, cb, nraw, red_v, red_i) / maxu;
        if (xkk < ubk) {
          if (tid == 0) { pivstat[gk] = 2; pivval[gk] = ubk - xkk; lb[gk] = ubk - xkk; }
          xkk = ubk;
        }
      }
      if (tid > k && tid < kb) lcol[tid] = S[tid][k] / xkk;
      __syncthreads();
      const int nrem = kb - k - 1;
      for (int idx = tid; idx < nrem * nrem; idx += blockDim.x) {
        int r = k + 1 + idx % nrem, i = k + 1 + idx / nrem;
        if (r >= i) S[r][i] -= lcol[i] * S[r][k];
      }
      __syncthreads();
      if (tid > k && tid < kb) S[tid][k] = lcol[tid];
      if (tid == 0) d[gk] = xkk;
    } else {
      // skipped pivot: d=0, column becomes the unit vector (blkchol2.c:157-161, blkchol.c:409-414)
      if (tid > k && tid < kb) S[tid][k] = 0.0;
      if (tid == 0) { pivstat[gk] = 1; pivval[gk] = xkk; d[gk] = 0.0; }   // S[k][k] is still being read by slower waves
    }
    __syncthreads();
  }
  for (int idx = tid; idx < kb * kb; idx += blockDim.x) {
    int i = idx % kb, j = idx / kb;
    if (i >= j) {
      const double v = (i == j) ? 1.0 : S[i][j];                                  // unit diagonal (blkchol2.c:136)
      Fs[(int64_t)(k0 + j) * ms + k0 + i] = v;
      Ts[(int64_t)(k0 + i) * ns + k0 + j] = v;                                    // L^T panel copy for the backward solve
    }
  }
}

// ---- K2: rows below the diagonal block: X = A21 * L11^-T, L21 = X * D^-1 (one row per work-item)
__global__ void __launch_bounds__(256)
k_ldl_panel(double *F, double *FT, FrontTab tab, const int *list, int panel, const double *d) {
  SDM_FP_STRICT;   // no FMA contraction: the pivot decisions must see the reference's mul-then-subtract rounding
  __shared__ double Ls[NB][NB + 1];
  __shared__ double ds[NB];
  const int s = list[blockIdx.y];
  const int ns = tab.ns[s], ms = tab.ms[s], first = tab.first[s];
  const int k0 = panel * NB, kb = min(NB, ns - k0);
  const int r0 = k0 + kb;
  if ((int)(blockIdx.x * blockDim.x) >= ms - r0) return;       // uniform per workgroup
  double *Fs = F + tab.foff[s];
  double *Ts = FT + tab.toff[s];
  const int tid = threadIdx.x;
  for (int idx = tid; idx < kb * kb; idx += blockDim.x) {
    int i = idx % kb, j = idx / kb;
    if (i > j) Ls[i][j] = Fs[(int64_t)(k0 + j) * ms + k0 + i];
  }
  if (tid < kb) ds[tid] = d[first + k0 + tid];
  __syncthreads();
  const int r = r0 + blockIdx.x * blockDim.x + tid;
  if (r < ms) {
    double x[NB];
#pragma unroll
    for (int c = 0; c < NB; c++) {
      if (c < kb) {
        double v = Fs[(int64_t)(k0 + c) * ms + r];
#pragma unroll
        for (int j = 0; j < NB; j++)
          if (j < c) v -= x[j] * Ls[c][j];
        const double dc = ds[c];
        x[c] = dc > 0.0 ? v : 0.0;
        const double l = dc > 0.0 ? v / dc : 0.0;
        Fs[(int64_t)(k0 + c) * ms + r] = l;
        Ts[(int64_t)r * ns + k0 + c] = l;
      }
    }
  }
}

// ---- K3: trailing update C -= L21 * D * L21' on the FP64 matrix cores.
// One 64x64 lower tile per workgroup (4 waves, each a 32x32 quadrant made of
// 2x2 v_mfma_f64_16x16x4_f64 tiles).  The product is formed transposed
// (D^T = B * A^T) so that the 16 consecutive lanes of a result register map to
// 16 consecutive rows of the column-major front: coalesced read-modify-write.
__global__ void __launch_bounds__(256)
k_ldl_update(double *F, FrontTab tab, const int *list, int panel, const double *d) {
  __shared__ double As[NB][TILE];   // As[k][i] = L21[I-tile row i][k]
  __shared__ double Bs[NB][TILE];   // Bs[k][j] = L21[J-tile row j][k] * d_k
  const int s = list[blockIdx.y];
  const int ns = tab.ns[s], ms = tab.ms[s], first = tab.first[s];
  const int k0 = panel * NB, kb = min(NB, ns - k0);
  const int r0 = k0 + kb, nrem = ms - r0;
  const int nt = (nrem + TILE - 1) / TILE;
  const int t = blockIdx.x;
  if (t >= nt * (nt + 1) / 2) return;
  int I = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while ((I + 1) * (I + 2) / 2 <= t) I++;
  while (I * (I + 1) / 2 > t) I--;
  const int J = t - I * (I + 1) / 2;
  double *Fs = F + tab.foff[s];
  const int tid = threadIdx.x;
  for (int idx = tid; idx < NB * TILE; idx += blockDim.x) {
    const int i = idx % TILE, k = idx / TILE;
    const int ri = r0 + I * TILE + i, rj = r0 + J * TILE + i;
    double a = 0.0, b = 0.0;
    if (k < kb) {
      if (ri < ms) a = Fs[(int64_t)(k0 + k) * ms + ri];
      if (rj < ms) b = Fs[(int64_t)(k0 + k) * ms + rj] * d[first + k0 + k];
    }
    As[k][i] = a; Bs[k][i] = b;
  }
  __syncthreads();
  const int w = tid >> 6, l = tid & 63;
  const int wi = w >> 1, wj = w & 1;
  sdm_double4 acc[2][2];
  for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) for (int r = 0; r < 4; r++) acc[a][b][r] = 0.0;
  const int lk = l >> 4, ll = l & 15;
  for (int kk = 0; kk < NB; kk += 4) {
    if (kk >= kb) break;
    double bv[2], av[2];
    for (int b = 0; b < 2; b++) bv[b] = Bs[kk + lk][wj * 32 + b * 16 + ll];
    for (int a = 0; a < 2; a++) av[a] = As[kk + lk][wi * 32 + a * 16 + ll];
    for (int a = 0; a < 2; a++)
      for (int b = 0; b < 2; b++) acc[a][b] = SDM_MFMA_F64_16x16x4(bv[b], av[a], acc[a][b]);
  }
  for (int a = 0; a < 2; a++)
    for (int b = 0; b < 2; b++)
      for (int r = 0; r < 4; r++) {
        const int jj = lk + 4 * r;                 // result row  -> J dimension (front column)
        const int gi = r0 + I * TILE + wi * 32 + a * 16 + ll;
        const int gj = r0 + J * TILE + wj * 32 + b * 16 + jj;
        if (gi < ms && gj < ms && gi >= gj) Fs[(int64_t)gj * ms + gi] -= acc[a][b][r];
      }
}

// ================================================================== solves
// One workgroup (512) per front; fronts of an etree level are independent.  The front-local vector lives in
// LDS.  Per 64-column panel: ONE wavefront does the in-block triangular solve -- lane i owns row i, its 64
// coefficients sit in registers (prefetched while the previous panel's GEMV runs) and the dependency chain is
// a v_readlane broadcast + one FMA per column (no LDS, no barrier) -- then all waves apply the panel to the
// remaining rows with fully coalesced column reads (one row per work-item, 64 independent loads in flight).
// Forward (fwblkslv.c:77-134): multifrontal, the children's update vectors are summed in a fixed order.
// Backward (bwblkslv.c:73-125): runs on the L^T panel copy written by the factor kernels, so the update of
// the earlier unknowns is the same coalesced, reduction-free GEMV as in the forward sweep.
__global__ void __launch_bounds__(512)
k_fw_level(const double *F, FrontTab tab, const int *list, double *wvec, double *y, int use_lds) {
  SDM_DYN_SMEM(smem);
  __shared__ double wb[SNB];
  const int s = list[blockIdx.x];
  const int ns = tab.ns[s], ms = tab.ms[s], first = tab.first[s];
  const double *Fs = F + tab.foff[s];
  double *wg = wvec + tab.woff[s];
  double *w = use_lds ? (double *)smem : wg;
  const int tid = threadIdx.x, bs = blockDim.x;
  for (int i = tid; i < ms; i += bs) w[i] = i < ns ? y[first + i] : 0.0;
  __syncthreads();
  for (int ci = tab.childptr[s]; ci < tab.childptr[s + 1]; ci++) {
    const int c = tab.childlist[ci];
    const int nc = tab.ns[c], mu = tab.ms[c] - nc;
    const int *rel = tab.relidx + tab.roff[c];
    const double *wc = wvec + tab.woff[c] + nc;
    for (int i = tid; i < mu; i += bs) w[rel[i]] += wc[i];
    __syncthreads();
  }
  double lr[SNB];
  if (tid < 64) {                                    // row tid of the first diagonal block
    const int kb = min(SNB, ns);
#pragma unroll
    for (int c = 0; c < SNB; c++) lr[c] = (c < tid && tid < kb) ? Fs[(int64_t)c * ms + tid] : 0.0;
  }
  for (int k0 = 0; k0 < ns; k0 += SNB) {
    const int kb = min(SNB, ns - k0);
    if (tid < 64) {
      double wi = tid < kb ? w[k0 + tid] : 0.0;
#pragma unroll
      for (int k = 0; k < SNB; k++) {
        if (k < kb) {
          const double wk = sdm_bcast_lane(wi, k);
          if (tid > k) wi -= lr[k] * wk;             // lr[k] = 0 for lanes outside the block
        }
      }
      if (tid < kb) { w[k0 + tid] = wi; wb[tid] = wi; y[first + k0 + tid] = wi; }
      const int k1 = k0 + SNB;                         // prefetch the next diagonal block's rows
      if (k1 < ns) {
        const int kbn = min(SNB, ns - k1);
#pragma unroll
        for (int c = 0; c < SNB; c++) lr[c] = (c < tid && tid < kbn) ? Fs[(int64_t)(k1 + c) * ms + k1 + tid] : 0.0;
      }
    }
    __syncthreads();
    for (int r = k0 + kb + tid; r < ms; r += bs) {
      const double *col = Fs + (int64_t)k0 * ms + r;
      double acc = 0.0;
#pragma unroll 8
      for (int c = 0; c < kb; c++) acc += col[(int64_t)c * ms] * wb[c];
      w[r] -= acc;
    }
    __syncthreads();
  }
  if (use_lds) for (int i = ns + tid; i < ms; i += bs) wg[i] = w[i];      // update vector for the parent
}

__global__ void __launch_bounds__(512)
k_bw_level(const double *FT, FrontTab tab, const int *list, double *y, int use_lds) {
  SDM_DYN_SMEM(smem);
  __shared__ double yb[SNB];
  const int s = list[blockIdx.x];
  const int ns = tab.ns[s], ms = tab.ms[s], first = tab.first[s];
  const double *T = FT + tab.toff[s];                // T[j*ns + r] = L(j, r): n_s x m_s, column-major
  const int *rows = tab.lindx + tab.xl[s];
  double *yl = use_lds ? (double *)smem : y + first;
  const int tid = threadIdx.x, bs = blockDim.x;
  if (use_lds) {
    for (int r = tid; r < ns; r += bs) yl[r] = y[first + r];
    __syncthreads();
  }
  // rows below the supernode belong to ancestors (already solved): y_s -= L21' * y[anc]
  if (ms > ns) {
    for (int r = tid; r < ns; r += bs) {
      double acc = 0.0;
      for (int j = ns; j < ms; j++) acc += T[(int64_t)j * ns + r] * y[rows[j]];
      yl[r] -= acc;
    }
    __syncthreads();
  }
  const int npan = (ns + SNB - 1) / SNB;
  double lr[SNB];
  if (tid < 64) {
    const int k0 = (npan - 1) * SNB, kb = ns - k0;
#pragma unroll
    for (int c = 0; c < SNB; c++) lr[c] = (c > tid && c < kb) ? T[(int64_t)(k0 + c) * ns + k0 + tid] : 0.0;
  }
  for (int pnl = npan - 1; pnl >= 0; pnl--) {
    const int k0 = pnl * SNB, kb = min(SNB, ns - k0);
    if (tid < 64) {
      double yi = tid < kb ? yl[k0 + tid] : 0.0;
#pragma unroll
      for (int k = SNB - 1; k >= 0; k--) {
        if (k < kb) {
          const double yk = sdm_bcast_lane(yi, k);
          if (tid < k) yi -= lr[k] * yk;             // lr[k] = L(k0+k, k0+tid)
        }
      }
      if (tid < kb) { yl[k0 + tid] = yi; yb[tid] = yi; }
      if (pnl > 0) {
        const int k1 = k0 - SNB;
#pragma unroll
        for (int c = 0; c < SNB; c++) lr[c] = (c > tid) ? T[(int64_t)(k1 + c) * ns + k1 + tid] : 0.0;
      }
    }
    __syncthreads();
    for (int r = tid; r < k0; r += bs) {
      const double *col = T + (int64_t)k0 * ns + r;
      double acc = 0.0;
#pragma unroll 8
      for (int c = 0; c < kb; c++) acc += col[(int64_t)c * ns] * yb[c];
      yl[r] -= acc;
    }
    __syncthreads();
  }
  if (use_lds) for (int r = tid; r < ns; r += bs) y[first + r] = yl[r];
}

__global__ void k_gather_perm(double *dst, const double *src, const int *perm, int m, int forward) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < m) { if (forward) dst[k] = src[perm[k]]; else dst[perm[k]] = src[k]; }
}
__global__ void k_divd(double *v, const double *d, int m) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < m) v[k] /= d[k];
}
// d for the solves: deninfac.m:89-94 with no dense columns -- skipped pivots (d=0) act as 1
__global__ void k_dsolve(double *ds, const double *d, int m) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < m) ds[k] = d[k] > 0.0 ? d[k] : 1.0;
}

// ============================================================ host drivers
static FrontTab front_tab(CholPlan &C) {
  FrontTab t;
  t.first = C.d_first.p; t.ns = C.d_ns.p; t.ms = C.d_ms.p;
  t.foff = C.d_foff.p; t.xl = C.d_xl.p; t.woff = C.d_woff.p; t.roff = C.d_roff.p; t.toff = C.d_toff.p;
  t.childptr = C.d_childptr.p; t.childlist = C.d_childlist.p; t.lindx = C.d_lindx.p; t.relidx = C.d_relidx.p;
  return t;
}
static inline int grid1d(int64_t n, int bs, int cap = 4096) {
  int64_t g = (n + bs - 1) / bs;
  return (int)std::max<int64_t>(1, std::min<int64_t>(g, cap));
}

void chol_factor(sdm_plan *P, const double canceltol, const double maxu, const double abstol, int use_absd) {
  CholPlan &C = P->chol;
  hipStream_t st = P->stream;
  FrontTab tab = front_tab(C);
  const int m = (int)C.m;
  
  SDM_HIP_CHECK(hipMemsetAsync(C.fronts.p, 0, (size_t)C.fsize * sizeof(double), st));
  SDM_KLAUNCH(P, k_assemble, dim3(grid1d(C.nnzL, 256)), dim3(256), 0, C.fronts.p, P->ada_val.p, C.d_asm_src.p,
             C.d_asm_dst.p, (int64_t)C.nnzL);
  SDM_KLAUNCH(P, k_prep_pivots, dim3(1), dim3(256), 0, m, P->ada_val.p, C.d_asm_src.p, C.d_Ljc.p, C.d_perm.p,
             P->absd.p, use_absd, canceltol, maxu, abstol, C.lb.p, C.ub.p, C.pivstat.p, C.pivval.p);
  for (int l = 0; l < C.nlevels; l++) {
    const int *list = C.d_levlist.p + C.levptr[l];
    const int nfr = C.levptr[l + 1] - C.levptr[l];
    if (l > 0) SDM_KLAUNCH(P, k_extend_add, dim3(C.lev_T[l], nfr), dim3(256), 0, C.fronts.p, tab, list);
    for (int li = C.lev_first_launch[l]; li < C.lev_first_launch[l + 1]; li++) {
      const LevelLaunch &L = C.launches[li];
      SDM_KLAUNCH(P, k_ldl_diag, dim3(L.nactive), dim3(256), 0, C.fronts.p, C.frontsT.p, tab, list, L.panel, C.d.p, C.lb.p, C.ub.p,
                 C.pivstat.p, C.pivval.p, C.colbuf.p, P->ada_val.p, C.d_asm_src.p, C.d_Ljc.p, m);
      if (L.maxrows > 0) {
        SDM_KLAUNCH(P, k_ldl_panel, dim3((L.maxrows + 255) / 256, L.nactive), dim3(256), 0, C.fronts.p, C.frontsT.p, tab, list,
                   L.panel, C.d.p);
        SDM_KLAUNCH(P, k_ldl_update, dim3(L.maxtiles, L.nactive), dim3(256), 0, C.fronts.p, tab, list, L.panel, C.d.p);
      }
    }
  }
  SDM_KLAUNCH(P, k_dsolve, dim3((m + 255) / 256), dim3(256), 0, C.dsolve.p, C.d.p, m);
  SDM_HIP_CHECK(hipGetLastError());
  P->factored = true;
}

void chol_extract(sdm_plan *P, double *d_Lpr_out) {
  CholPlan &C = P->chol;
  SDM_KLAUNCH(P, k_extract, dim3(grid1d(C.nnzL, 256)), dim3(256), 0, d_Lpr_out, C.fronts.p, C.d_asm_dst.p,
             (int64_t)C.nnzL);
}

void chol_load_factor(sdm_plan *P, const double *h_Lpr) {
  CholPlan &C = P->chol;
  DevBuf<double> tmp;
  tmp.upload(h_Lpr, (size_t)C.nnzL);
  SDM_KLAUNCH(P, k_load_factor, dim3(grid1d(C.nnzL, 256)), dim3(256), 0, C.fronts.p, C.frontsT.p, tmp.p, C.d_asm_dst.p,
              C.d_asm_dstT.p, (int64_t)C.nnzL);
  SDM_HIP_CHECK(hipStreamSynchronize(P->stream));
  P->factored = true;
}

// the front-local vector goes to LDS when the largest front of the plan fits (96 KB), else it stays in HBM
static void solve_lds(CholPlan &C, bool fw, size_t &bytes, int &use) {
  const int need = fw ? C.maxms : C.maxns;
  use = need <= SOLVE_LDS_MAX ? 1 : 0;
  bytes = use ? (size_t)need * sizeof(double) : 0;
}
void solve_fw(sdm_plan *P) {
  CholPlan &C = P->chol;
  FrontTab tab = front_tab(C);
  size_t fw_lds; int fw_use; solve_lds(C, true, fw_lds, fw_use);
#ifndef SDM_EMU
  if (fw_lds > 48 * 1024) SDM_HIP_CHECK(hipFuncSetAttribute((const void *)k_fw_level, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fw_lds));
#endif
  for (int l = 0; l < C.nlevels; l++) {
    const int nfr = C.levptr[l + 1] - C.levptr[l];
    SDM_KLAUNCH(P, k_fw_level, dim3(nfr), dim3(512), fw_lds, C.fronts.p, tab, C.d_levlist.p + C.levptr[l], C.wvec.p,
                P->ywork.p, fw_use);
  }
}
void solve_bw(sdm_plan *P) {
  CholPlan &C = P->chol;
  FrontTab tab = front_tab(C);
  size_t bw_lds; int bw_use; solve_lds(C, false, bw_lds, bw_use);
#ifndef SDM_EMU
  if (bw_lds > 48 * 1024) SDM_HIP_CHECK(hipFuncSetAttribute((const void *)k_bw_level, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bw_lds));
#endif
  for (int l = C.nlevels - 1; l >= 0; l--) {
    const int nfr = C.levptr[l + 1] - C.levptr[l];
    SDM_KLAUNCH(P, k_bw_level, dim3(nfr), dim3(512), bw_lds, C.frontsT.p, tab, C.d_levlist.p + C.levptr[l], P->ywork.p, bw_use);
  }
}
void vec_gather(sdm_plan *P, double *dst, const double *src, bool forward) {
  const int m = (int)P->chol.m;
  SDM_KLAUNCH(P, k_gather_perm, dim3((m + 255) / 256), dim3(256), 0, dst, src, P->chol.d_perm.p, m, forward ? 1 : 0);
}
void vec_divd(sdm_plan *P, double *v) {
  const int m = (int)P->chol.m;
  SDM_KLAUNCH(P, k_divd, dim3((m + 255) / 256), dim3(256), 0, v, P->chol.dsolve.p, m);
}

}  // namespace sdm

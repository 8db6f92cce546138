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
// front the elimination is blocked by 64 columns, ONE launch per panel
// (k_ldl_panel): workgroup 0 factors the diagonal block in registers / LDS (pivot
// rule applied column by column), further workgroups solve the rows below it as
// the block is published 16 columns at a time, and the trailing update
// C -= L21*D*L21' of the PREVIOUS panel (FP64 matrix cores, 64x64 tiles) rides
// along in the remaining workgroups; per-front counters in HBM order them.
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
  C.sn_first.resize(nsuper); C.sn_ns.resize(nsuper); C.sn_ms.resize(nsuper); C.sn_ld.resize(nsuper);
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
    C.sn_ld[s] = (int)(ms + (ms & 1));                        // even leading dimension: 16-byte aligned row pairs in every column
    foff += (int64_t)C.sn_ld[s] * ms; xl += ms; toff += (int64_t)((n + NB - 1) / NB) * NB * NB;
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
      int s = snode[j]; int c = (int)(j - C.sn_first[s]);
      for (sdm_int t = Ljc[j]; t < Ljc[j + 1]; t++) {
        asm_src[t] = rowpos[perm[Lir[t]]];
        asm_dst[t] = C.sn_foff[s] + (int64_t)c * C.sn_ld[s] + c + (t - Ljc[j]);
        { // transposed copy of the 64x64 diagonal blocks only: DT[panel][row in block][col in block]
          const int64_t rr = c + (t - Ljc[j]); const int pnl = c / NB;
          asm_dstT[t] = (rr < (int64_t)(pnl + 1) * NB && rr < C.sn_ns[s]) ? C.sn_toff[s] + (int64_t)pnl * NB * NB + (rr - (int64_t)pnl * NB) * NB + (c - pnl * NB) : -1;
        }
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
      LevelLaunch L; L.level = l; L.panel = p; L.nactive = 0; L.maxrows = 0; L.maxtiles = 0; L.lasttiles = 0; L.ride_wgs = 0;
      for (int i = b; i < e; i++) {
        int s = C.levlist[i];
        if (C.sn_ns[s] <= p * NB) break;
        L.nactive++;
        int kb = std::min(NB, C.sn_ns[s] - p * NB);
        int rows = C.sn_ms[s] - (p * NB + kb);
        L.maxrows = std::max(L.maxrows, rows);
        int nt = (rows + TILE - 1) / TILE;
        L.maxtiles = std::max(L.maxtiles, nt * (nt + 1) / 2);
        if (C.sn_ns[s] <= (p + 1) * NB) L.lasttiles = std::max(L.lasttiles, nt * (nt + 1) / 2);
        // workgroups of k_ldl_panel beyond the diagonal-block one (see the kernel): row solves, then pairs of update tiles
        const int nrw = rows > TRSM_ROWS ? (C.sn_ms[s] - (p * NB + NB) + ROWS_BATCH - 1) / ROWS_BATCH : 0;
        int tw = 0;
        if (p > 0) {
          const int ntp = (C.sn_ms[s] - p * NB + TILE - 1) / TILE;            // tile rows of the update of panel p-1
          tw = nrw > 0 ? ((ntp - 1) * ntp / 2 + 1) / 2 : (ntp * (ntp + 1) / 2 - 1 + 1) / 2;
        }
        L.ride_wgs = std::max(L.ride_wgs, nrw + tw);
      }
      C.launches.push_back(L);
    }
    // q0: first panel of the level whose diagonal-block launch carries the previous panel's update tiles
    // (maxtiles does not grow with p)
    int q0 = 1 << 30;
    for (int li = C.lev_first_launch[l] + 1; li < (int)C.launches.size(); li++)
      if (C.launches[li - 1].maxtiles <= FUSE_MAX_TILES) { q0 = C.launches[li].panel; break; }
    for (int li = C.lev_first_launch[l]; li < (int)C.launches.size(); li++) C.launches[li].q0 = q0;
  }
  C.lev_first_launch[nlev] = (int)C.launches.size();
  // upload
  C.d_first.upload(C.sn_first); C.d_ns.upload(C.sn_ns); C.d_ms.upload(C.sn_ms); C.d_ld.upload(C.sn_ld); C.d_parent.upload(C.sn_parent);
  C.d_childptr.upload(C.childptr); C.d_childlist.upload(C.childlist); C.d_levlist.upload(C.levlist);
  C.d_lindx.upload(lindx); C.d_relidx.upload(relidx);
  { std::vector<int> p32(m); for (sdm_int i = 0; i < m; i++) p32[i] = (int)perm[i]; C.d_perm.upload(p32); }
  C.d_foff.upload(C.sn_foff); C.d_xl.upload(C.sn_xl); C.d_woff.upload(C.sn_woff); C.d_roff.upload(C.sn_roff);
  C.d_asm_src.upload(asm_src); C.d_asm_dst.upload(asm_dst); C.d_asm_dstT.upload(asm_dstT); C.d_toff.upload(C.sn_toff);
  C.frontsT.alloc((size_t)C.tsize);
  { std::vector<int64_t> l64(C.Ljc.begin(), C.Ljc.end()); C.d_Ljc.upload(l64); }
  C.fronts.alloc((size_t)C.fsize); C.wvec.alloc((size_t)C.wsize); C.colbuf.alloc((size_t)C.wsize + (size_t)nsuper);
  C.d.alloc(m); C.dsolve.alloc(m); C.lb.alloc(m); C.pivval.alloc(m); C.pivstat.alloc(m); C.ub.alloc(3); C.upd_cnt.alloc((size_t)std::max<sdm_int>(1, C.nsuper)); C.diag_cnt.alloc((size_t)std::max<sdm_int>(1, C.nsuper));
  P->ada_val.alloc((size_t)C.nnzADA); P->absd.alloc(m); P->lpr.alloc((size_t)C.nnzL);
  P->rhs.alloc(m); P->y.alloc(m); P->ywork.alloc(m);
  P->has_chol = true; P->factored = false;
}

// ================================================================= kernels
struct FrontTab {
  const int *first, *ns, *ms, *ld;
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
  for (; t < nnzL; t += stride) { const double v = Lpr[t]; F[dst[t]] = v; if (dstT[t] >= 0) FT[dstT[t]] = v; }
}

// ---- pivot thresholds (blkchol.c:168-184), grid-stride over the columns.
//   ub = max_j P(perm_j,perm_j) / maxu^2 ;  lb_j = max(abstol, canceltol * orgd_j)
// ub[2] collects max_j as the bit pattern of a non-negative double (ordered like the unsigned integer: atomicMax is
// exact and order independent); ub[1] = maxu; k_ldl_panel forms ub from them.  ub[2] is zeroed by the host before.
__global__ void k_prep_pivots(int m, const double *ada, const int *asm_src, const int64_t *Ljc, const int *perm,
                              const double *absd, int use_absd, double canceltol, double maxu, double abstol,
                              double *lb, double *ub, int *pivstat, double *pivval, int nsuper, int *upd_cnt, int *diag_cnt) {
  __shared__ double red[256];
  const int gid = blockIdx.x * blockDim.x + threadIdx.x, gstride = gridDim.x * blockDim.x;
  for (int i = gid; i < nsuper; i += gstride) { upd_cnt[i] = 0; diag_cnt[i] = 0; }     // counters of k_ldl_panel
  double mx = 0.0;
  for (int j = gid; j < m; j += gstride) {
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
  if (threadIdx.x == 0) {
    union { double d; unsigned long long u; } b; b.d = red[0];
    atomicMax((unsigned long long *)&ub[2], b.u);
    ub[1] = maxu;
  }
}

// ---- extend-add: parent front += children's Schur complements.
// grid (T, fronts of the level).  Workgroup `slice` owns the parent columns
// J with J % T == slice, so every parent entry has exactly one writer and the
// children are applied in a fixed order: deterministic, no atomics.
__global__ void k_extend_add(double *F, FrontTab tab, const int *list) {
  const int p = list[blockIdx.y];
  const int slice = blockIdx.x, T = gridDim.x;
  const int msp = tab.ld[p];
  double *Fp = F + tab.foff[p];
  for (int ci = tab.childptr[p]; ci < tab.childptr[p + 1]; ci++) {
    const int c = tab.childlist[ci];
    const int nc = tab.ns[c], mu = tab.ms[c] - nc, mc = tab.ld[c];
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
// (unscaled, updated by the columns < k), Lc[j*NB+i] = l_ij of the finished
// columns, ds = their pivots, rows below the block are
// obtained by forward substitution against those columns.  cb = scratch of >= ms+1
// doubles.
__device__ __noinline__ double pivot_probe(const double (*S)[NB + 1], const double *Lc, int k, int kb, int k0, int ns,
                                           int ms, int ld, const double *Fs, const double *ds, double *cb,
                                           double next_raw_diag, double *red_v, int *red_i) {
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
      double v = Fs[(int64_t)(k0 + c) * ld + r];
      for (int j = 0; j < c; j++) v -= x[j] * Lc[j * NB + c];   // l_cj, scaled
      double dc = (c < k) ? ds[c] : 1.0;
      x[c] = (dc > 0.0) ? v : 0.0;
      if (c < k && dc > 0.0) diagacc += x[c] * (x[c] / dc);
    }
    cb[nin + (r - (k0 + kb))] = x[k];
    if (r == k0 + kb && nin == 0 && k0 + k + 1 < ns)   // next column = first row below the block
      cb[len] = Fs[(int64_t)r * ld + r] - diagacc;
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

// ---- K3: trailing update C -= L21 * D * L21' on the FP64 matrix cores, one 64x64 lower tile per workgroup.
// NW wavefronts share the tile: 4 (32x32 quadrants of 2x2 v_mfma_f64_16x16x4_f64 tiles) in the stand-alone kernel,
// 8 (32x16 blocks) when the update rides along with the next diagonal-block launch.  The product is formed
// transposed (D^T = B * A^T) so that the 16 consecutive lanes of a result register map to 16 consecutive rows of
// the column-major front: coalesced read-modify-write.  As[k][i] = L21[I-tile row i][k], Bs[k][j] = L21[J-tile
// row j][k] * d_k, dsh = NB doubles (all LDS).
// DIAG (tile (0,0) in the workgroup that factors the next diagonal block right away): the result also goes to LDS
// as that kernel's S / Lc arrays (which overlay As / Bs), kbn = columns of the next panel.
template <int NW, bool DIAG, bool WT = false, bool TW = false>
__device__ __forceinline__ void update_tile(double *Fs, int ld, int ms, int first, int k0, int kb, int I, int J, const double *d,
                                            double (*As)[TILE], double (*Bs)[TILE], double *dsh,
                                            double (*S)[NB + 1] = nullptr, double *Lc = nullptr, int kbn = 0,
                                            int tid = threadIdx.x, bool active = true, double *tw = nullptr) {
  // TW: the result also goes to LDS as the row solve's wave tiles (tw[(row/16)*NB*17 + col*17 + row%16], columns
  // beyond kbn zeroed) -- the workgroup that solves these rows next needs no second trip to HBM
  // tid: position inside the group of NW wavefronts that shares the tile (two groups of one workgroup may run two
  // tiles side by side: same barriers); active = false: go through the motions (barriers) without storing
  constexpr int BJ = 8 / NW;                                  // 16-column MFMA tiles per wavefront along J
  const int r0 = k0 + kb;
  SDM_PHASE_BEGIN();
  if (tid < NB) dsh[tid] = tid < kb ? d[first + k0 + tid] : 0.0;
  const int w = tid >> 6, l = tid & 63;
  const int wi = NW == 4 ? w >> 1 : w >> 2, wj = NW == 4 ? w & 1 : w & 3;
  const int cj = wj * 16 * BJ;                                // first tile column of this wavefront
  const int lk = l >> 4, ll = l & 15;
  // read-modify-write of the tile: its loads go out together with the operands' (one memory round trip for both)
  double cv[2][BJ][4];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < BJ; b++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int jj = lk + 4 * r;                 // result row  -> J dimension (front column)
        const int gi = r0 + I * TILE + wi * 32 + a * 16 + ll;
        const int gj = r0 + J * TILE + cj + b * 16 + jj;
        cv[a][b][r] = Fs[(int64_t)min(gj, ms - 1) * ld + min(gi, ms - 1)];
      }
  {
    // all loads of a work-item are issued before the first use (addresses clamped, masked afterwards): one
    // memory round trip per tile instead of one per element
    const int i = tid & 63, kq = tid >> 6;
    const int ri = r0 + I * TILE + i, rj = r0 + J * TILE + i;
    const double *pa = Fs + min(ri, ms - 1), *pb = Fs + min(rj, ms - 1);
    double av[NB / NW], bv[NB / NW];
#pragma unroll
    for (int q = 0; q < NB / NW; q++) {
      const int64_t off = (int64_t)(k0 + min(kq + NW * q, kb - 1)) * ld;
      av[q] = pa[off]; bv[q] = pb[off];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NB / NW; q++) {
      const int k = kq + NW * q;
      As[k][i] = (k < kb && ri < ms) ? av[q] : 0.0;
      Bs[k][i] = (k < kb && rj < ms) ? bv[q] * dsh[k] : 0.0;
    }
  }
  __syncthreads();
  SDM_PHASE(DIAG ? 14 : 28);
  sdm_double4 acc[2][BJ];
  for (int a = 0; a < 2; a++) for (int b = 0; b < BJ; b++) for (int r = 0; r < 4; r++) acc[a][b][r] = 0.0;
  // operands of step kk+4 are fetched from LDS while the MFMAs of step kk issue (As/Bs rows beyond kb are zero)
  double bv[BJ], av[2];
#pragma unroll
  for (int b = 0; b < BJ; b++) bv[b] = Bs[lk][cj + b * 16 + ll];
#pragma unroll
  for (int a = 0; a < 2; a++) av[a] = As[lk][wi * 32 + a * 16 + ll];
#pragma unroll
  for (int kk = 0; kk < NB; kk += 4) {
    double bn[BJ], an[2];
    const int kn = min(kk + 4, NB - 4);
#pragma unroll
    for (int b = 0; b < BJ; b++) bn[b] = Bs[kn + lk][cj + b * 16 + ll];
#pragma unroll
    for (int a = 0; a < 2; a++) an[a] = As[kn + lk][wi * 32 + a * 16 + ll];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int b = 0; b < BJ; b++) acc[a][b] = SDM_MFMA_F64_16x16x4(bv[b], av[a], acc[a][b]);
#pragma unroll
    for (int b = 0; b < BJ; b++) bv[b] = bn[b];
#pragma unroll
    for (int a = 0; a < 2; a++) av[a] = an[a];
  }
  SDM_PHASE(DIAG ? 15 : 29);
  if (TW) __syncthreads();                         // As / Bs are dead: the wave tiles overlay them
  if (DIAG) {
    __syncthreads();                               // As / Bs are dead: S and Lc overlay them
    const int tx = tid & 63, ty = tid >> 6;
    for (int j = ty; j < NB; j += NW) { S[tx][j] = (tx == j && tx >= kbn) ? 1.0 : 0.0; Lc[j * NB + tx] = 0.0; }
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < BJ; b++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int jj = lk + 4 * r;
        const int ti = wi * 32 + a * 16 + ll, tj = cj + b * 16 + jj;
        const int gi = r0 + I * TILE + ti, gj = r0 + J * TILE + tj;
        if (active && gi < ms && gj < ms && gi >= gj) {
          const double v = cv[a][b][r] - acc[a][b][r];
          if (WT) sdm_store_wt(&Fs[(int64_t)gj * ld + gi], v); else Fs[(int64_t)gj * ld + gi] = v;
          if (DIAG && ti < kbn) S[ti][tj] = v;
          if (TW) tw[(ti >> 4) * (NB * 17) + tj * 17 + (ti & 15)] = tj < kbn ? v : 0.0;
        }
      }
  SDM_PHASE(DIAG ? 31 : 30);
}
// lower tile t -> (I, J), I >= J
__device__ __forceinline__ void tile_index(int t, int &I, int &J) {
  I = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while ((I + 1) * (I + 2) / 2 <= t) I++;
  while (I * (I + 1) / 2 > t) I--;
  J = t - I * (I + 1) / 2;
}

// ---- K1 (k_ldl_panel): one workgroup per front of a level, 64-column panel p: LDL' of the kb x kb diagonal
// block; when the rows below the block fit one workgroup (<= TRSM_ROWS) they are solved here as well and the
// block is written back in place.  Otherwise the factored block goes to the transposed copy DT only and
// the row-solve workgroups of the same launch solve the rows and copy the block in place -- nobody may
// overwrite the panel while the never-fail rule's column probe of K1 can still read its raw values.
//
// Diagonal block (bit-faithful to cholonBlk, blkchol2.c:114-161: column i -= (x_ik / x_kk) * x(:,k), one multiply
// and one subtract per entry, columns in order): the 64 columns are swept SW at a time.  Wavefront 0 holds the SW
// current columns of all 64 rows in registers (lane = row) and runs the sweep -- pivots and multipliers travel by
// v_readlane, there is no LDS traffic and no barrier inside a sweep.  The sweep is a chain of dependent FP64
// divisions (~120 clocks per column measured, tools/ubench/ubench6) and it is issue bound when several wavefronts
// repeat it, so it is pipelined against the rest: while wavefront 0 first brings the NEXT SW columns up to date
// (look-ahead) and sweeps them, the other wavefronts apply the sweep before to the remaining trailing columns
// (x_rj -= l_jk * x_rk, k ascending: the same operations in the same order as the column-by-column reference).
// One barrier per sweep.
// A pivot that needs the never-fail rule's column probe (x_kk < ub) abandons this path; the block is reloaded
// and factored by the general all-work-items loop, which can call pivot_probe.
//
// Rows below the block: fronts with few rows use the faithful substitution (one row per work-item,
// x_rc = a_rc - sum_{j<c} x_rj * l_cj in ascending j, l_rc = x_rc / d_c).  Fronts with >= MFMA_MIN_ROWS rows below
// the block solve 16 rows per wavefront by blocked substitution: per 16-column block the GEMM part
// T_b = A_b - sum_{b'<b} X_b' L_bb'^T runs on the FP64 matrix cores, the 16x16 triangle is solved by substitution
// (no inverse is formed: the never-fail pivot rule allows multipliers up to maxu = 5e5); results agree with the
// plain substitution to rounding.
// 16 rows x 64 columns of the panel -> LDS wave tile Tw[col*17 + row]
__device__ __forceinline__ void rows_stage(const double *Fs, int ld, int ms, int k0, int kb, int R0, double *Tw, int lane) {
  const int li = lane & 15, lk = lane >> 4;
  double tv[NB / 4];
  const double *pr = Fs + min(R0 + li, ms - 1);
#pragma unroll
  for (int c4 = 0; c4 < NB / 4; c4++) tv[c4] = pr[(int64_t)(k0 + min(4 * c4 + lk, kb - 1)) * ld];    // 16 loads in flight
#pragma unroll
  for (int c4 = 0; c4 < NB / 4; c4++) { const int c = 4 * c4 + lk; Tw[c * 17 + li] = c < kb ? tv[c4] : 0.0; }
}
// 16-column block b of the blocked substitution on the wave tile: needs columns 0 .. 16b+15 of S (L11) and ds
__device__ __forceinline__ void rows_block(int b, const double (*S)[NB + 1], const double *ds, double *Tw, int lane) {
  const int li = lane & 15, lk = lane >> 4;
  const int cb = 16 * b;
  if (b > 0) {
    // T = A_b - sum_{b'<b} X_b' L_bb'^T on the matrix cores (D layout: lane holds rows lk+4r of column li)
    sdm_double4 acc;
    for (int r = 0; r < 4; r++) acc[r] = Tw[(cb + li) * 17 + lk + 4 * r];
    for (int bp = 0; bp < b; bp++)
      for (int q = 0; q < 4; q++) {
        const double a = Tw[(16 * bp + 4 * q + lk) * 17 + li];          // X_bp[row li][k]
        const double bv = S[cb + li][16 * bp + 4 * q + lk];             // L11[cb + j][k]
        acc = SDM_MFMA_F64_16x16x4(-a, bv, acc);
      }
    for (int r = 0; r < 4; r++) Tw[(cb + li) * 17 + lk + 4 * r] = acc[r];
    SDM_WAVE_SYNC();
  }
  // the 16x16 triangle by substitution, lane li = row (the 4 lane groups lk compute the same row redundantly),
  // column-oriented: once x_j is final, x_c -= x_j l_cj for all c > j (independent updates, one LDS round trip
  // per column of the triangle) -- no inverse of the block is formed (multipliers may be as large as maxu)
  double x[16];
#pragma unroll
  for (int c = 0; c < 16; c++) x[c] = Tw[(cb + c) * 17 + li];
  double lcol[16], dsv[16];
#pragma unroll
  for (int c = 0; c < 16; c++) { lcol[c] = c > 0 ? S[cb + c][cb] : 0.0; dsv[c] = ds[cb + c]; }
#pragma unroll
  for (int j = 0; j < 16; j++) {
    double lnext[16];                                                   // column j+1 is fetched while column j is applied
#pragma unroll
    for (int c = 0; c < 16; c++) lnext[c] = (j + 1 < 16 && c > j + 1) ? S[cb + c][cb + j + 1] : 0.0;
    if (dsv[j] <= 0.0) x[j] = 0.0;                                      // skipped pivot: column not used (blkchol2.c:157-161)
#pragma unroll
    for (int c = 0; c < 16; c++)
      if (c > j) x[c] -= x[j] * lcol[c];
#pragma unroll
    for (int c = 0; c < 16; c++) lcol[c] = lnext[c];
  }
  SDM_WAVE_SYNC();
#pragma unroll
  for (int c = 0; c < 16; c++) Tw[(cb + c) * 17 + li] = x[c];
  SDM_WAVE_SYNC();
}
// l = x / d out of the wave tile into the front
__device__ __forceinline__ void rows_store(double *Fs, int ld, int ms, int k0, int kb, int R0, const double *ds, const double *Tw, int lane) {
  const int li = lane & 15, lk = lane >> 4;
  for (int c4 = 0; c4 < NB / 4; c4++) {
    const int c = 4 * c4 + lk, row = R0 + li;
    if (c < kb && row < ms) {
      const double dc = ds[c], xv = Tw[c * 17 + li];
      Fs[(int64_t)(k0 + c) * ld + row] = dc > 0.0 ? xv / dc : 0.0;
    }
  }
}
__device__ __forceinline__ void panel_rows_mfma(double *Fs, int ld, int ms, int k0, int kb, int R0, const double (*S)[NB + 1],
                                                const double *ds, double *Tw, int lane, bool staged = false) {
  if (!staged) rows_stage(Fs, ld, ms, k0, kb, R0, Tw, lane);
  SDM_WAVE_SYNC();
  SDM_PHASE_BEGIN();
  for (int b = 0; b < NB / 16 && 16 * b < kb; b++) rows_block(b, S, ds, Tw, lane);
  SDM_PHASE(26);
  rows_store(Fs, ld, ms, k0, kb, R0, ds, Tw, lane);
}

// rows [rbeg, rend) below the diagonal block of panel k0 (at most brows = TRSM_ROWS of them per call)
__device__ __forceinline__ void panel_rows(double *Fs, int ld, int ns, int ms, int k0, int kb, int rbeg, int rend, int brows,
                                           const double (*S)[NB + 1], const double *ds, double *RB, bool staged = false) {
  SDM_FP_STRICT;
  const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6, ny = blockDim.x >> 6;
  rend = min(rend, ms);
  if (ms - min(NB, ns) >= MFMA_MIN_ROWS) {                         // per front, the same path for all its panels
    // 16 rows per wavefront at a time, blocked substitution with the GEMM part on the matrix cores
    for (int R0 = rbeg + 16 * ty; R0 < rend; R0 += 16 * ny)
      panel_rows_mfma(Fs, ld, rend, k0, kb, R0, S, ds, RB + ty * (NB * 17), tx, staged);
    return;
  }
  // few rows: faithful substitution, one row per work-item, 16-column chunks; x of earlier chunks parked in LDS
  double *Xs = RB;
  const int r = rbeg + tid;
  if (tid >= brows || r >= rend) return;
  for (int c0 = 0; c0 < kb; c0 += CHK) {
    double acc[CHK], x[CHK];
#pragma unroll
    for (int cc = 0; cc < CHK; cc++) acc[cc] = (c0 + cc < kb) ? Fs[(int64_t)(k0 + c0 + cc) * ld + r] : 0.0;
    for (int j = 0; j < c0; j++) {
      const double xj = Xs[j * brows + tid];
#pragma unroll
      for (int cc = 0; cc < CHK; cc++) acc[cc] -= xj * S[c0 + cc][j];
    }
#pragma unroll
    for (int cc = 0; cc < CHK; cc++) {
      double v = acc[cc];
#pragma unroll
      for (int jj = 0; jj < CHK; jj++)
        if (jj < cc) v -= x[jj] * S[c0 + cc][c0 + jj];
      const double dc = ds[c0 + cc];
      x[cc] = dc > 0.0 ? v : 0.0;
      if (c0 + cc < kb) Fs[(int64_t)(k0 + c0 + cc) * ld + r] = dc > 0.0 ? v / dc : 0.0;
      if (c0 + CHK < NB) Xs[(c0 + cc) * brows + tid] = x[cc];
    }
  }
}

// Workgroup 0 of k_ldl_panel before it reads rows below its diagonal block: the tiles of the previous panel's update
// that cover them (block column 0) are applied by other workgroups of the same launch -- by the row-solve workgroups
// when the panel has them (more than TRSM_ROWS rows below the block: one signal each), else by the tile workgroups
// (one signal per pair of tiles).  upd_cnt[s] counts those signals since the factorisation began (reset by
// k_prep_pivots); all work-items call this.  The spin gives up after a few seconds rather than hang the device.
__device__ __forceinline__ int panel_row_wgs(int ns, int ms, int q) {
  const int kbq = min(NB, ns - q * NB), nrows = ms - (q * NB + kbq);
  return nrows > TRSM_ROWS ? (ms - (q * NB + NB) + ROWS_BATCH - 1) / ROWS_BATCH : 0;
}
// tmo: the plan's own time-out flag (pinned host memory, CholPlan::tmo): a spin that gives up raises it; the host turns
// it into an error at the next read-back of that plan (chol_wait_timeouts)
__device__ __forceinline__ void spin_until(const int *cnt, int target, int *tmo) {
  if (threadIdx.x == 0) {
    long it = 0;
    for (; sdm_signal_load(cnt) < target && it < (1L << 21); it++) SDM_SPIN_PAUSE();
    if (it == (1L << 21)) sdm_raise_flag(tmo);
  }
  __syncthreads();
  SDM_ACQUIRE_FENCE();
}
__device__ __forceinline__ void wait_prev_update(const int *cnt, int ns, int ms, int panel, int q0, int *tmo) {
  int target = 0;                                              // launches q0 .. panel carried update tiles
  for (int q = max(q0, 1); q <= panel; q++) {
    const int nt = (ms - q * NB + TILE - 1) / TILE, nrw = panel_row_wgs(ns, ms, q);
    target += nrw > 0 ? nrw : (nt * (nt + 1) / 2) / 2;
  }
  spin_until(cnt, target, tmo);
}

__global__ void __launch_bounds__(LDL_THREADS)
k_ldl_panel(double *F, double *DT, FrontTab tab, const int *list, int panel, double *d, double *lb, const double *ubp,
            int *pivstat, double *pivval, double *colbuf, const double *ada, const int *asm_src,
            const int64_t *Ljc, int mtot, int *upd_cnt, int *diag_cnt, int q0, int phase, int *tmo) {
  SDM_FP_STRICT;   // no FMA contraction: the pivot decisions must see the reference's mul-then-subtract rounding
  SDM_DYN_SMEM(smem);
  // ONE launch per 64-column panel p.  grid = (workgroups, fronts); per front:
  //   workgroup 0        tile (0,0) of the trailing update of panel p-1 (its own diagonal block), then the LDL' of the
  //                      block, published (DT, d) for the row-solve workgroups;
  //   1 .. nrw           (fronts with more than TRSM_ROWS rows below the block) row solve of ROWS_BATCH rows each: first
  //                      the tile of the previous update that covers exactly those rows in this panel's columns, then
  //                      -- once workgroup 0 has published the factored block -- the substitution;
  //   the rest           the other tiles of the previous update, two side by side per workgroup (nobody in this launch
  //                      reads them, except in fronts without row-solve workgroups, where they signal).
  // All of this hides behind workgroup 0's dependency chain.  The emulator runs workgroups one after the other:
  // phase 1 (everything but the substitution, diagonal block last) and phase 2 (the substitution) are two launches
  // there; the GPU runs phase 0 = both.
  {
    const int s = list[blockIdx.y];
    const int ns = tab.ns[s], ms = tab.ms[s], ld = tab.ld[s], first = tab.first[s];
    const int k0c = panel * NB, kbc = min(NB, ns - k0c), nrowsc = ms - (k0c + kbc);
    const int nrw = nrowsc > TRSM_ROWS ? (ms - (k0c + NB) + ROWS_BATCH - 1) / ROWS_BATCH : 0;   // tile rows below the first
    const int kp = (panel - 1) * NB;                               // previous panel (full when there is a panel p)
    const int nt = panel > 0 ? (ms - (kp + NB) + TILE - 1) / TILE : 0;
    // Roles by "logical" index bx: 0 = diagonal block, 1..nrw = row solves, beyond = update tiles.  The hardware hands
    // out workgroups in launch order, and a workgroup that waits must wait for one handed out BEFORE it (or for one
    // that does not wait before it signals), else a full device of waiting workgroups could keep the awaited one out:
    //   fronts with row-solve workgroups:  diagonal block < row solves (wait for it) < tiles (nobody waits for them);
    //   small fronts:                      tiles (never wait) < diagonal block (its in-workgroup row solve waits for them).
    // The emulator runs them one after the other in the order  tiles, row solves (phase 1: their update tile only),
    // diagonal block.
#ifdef SDM_EMU
    const int bx = (int)gridDim.x - 1 - (int)blockIdx.x;
#else
    int bx = blockIdx.x;
    if (nrw == 0) {
      const int ntw = panel > 0 ? (nt * (nt + 1) / 2) / 2 : 0;
      if (bx > ntw) return;
      bx = bx < ntw ? 1 + bx : 0;
    }
#endif
    double (*As)[TILE] = (double (*)[TILE])smem;
    double (*Bs)[TILE] = As + NB;
    __shared__ double dsh[NB];
    if (bx > 0 && bx <= nrw) {
      // ---- row-solve workgroup b
      const int b = bx - 1;
      double *Fs = F + tab.foff[s];
      const bool mfma_rows = ms - min(NB, ns) >= MFMA_MIN_ROWS;
      if (phase != 2 && panel > 0) {
        if (mfma_rows)                                              // the result doubles as the row solve's wave tiles in LDS
          update_tile<LDL_THREADS / 64, false, true, true>(Fs, ld, ms, first, kp, NB, b + 1, 0, d, As, Bs, dsh, nullptr, nullptr, kbc,
                                                            threadIdx.x, true, (double *)smem + NB * (NB + 1));
        else
          update_tile<LDL_THREADS / 64, false, true>(Fs, ld, ms, first, kp, NB, b + 1, 0, d, As, Bs, dsh);
        SDM_STORES_DONE();
        __syncthreads();
        if (threadIdx.x == 0) sdm_signal_add(&upd_cnt[s]);
      }
      if (phase == 1) return;
      double (*S)[NB + 1] = (double (*)[NB + 1])smem;
      double *RB = (double *)smem + NB * (NB + 1);
      __shared__ double dsr[NB];
      const double *Ds = DT + tab.toff[s] + (int64_t)panel * NB * NB;    // Ds[i*NB + j] = L(k0+i, k0+j)
      const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6, ny = blockDim.x >> 6;
      const int rbeg = k0c + NB * (b + 1), rend = min(ms, k0c + NB * (b + 2));      // = tile row b+1
      if (!mfma_rows) {
        // few rows: the faithful substitution needs the whole block
        spin_until(diag_cnt + s, 4 * (panel + 1), tmo);
        constexpr int NQ = NB / (LDL_THREADS / 64);
        double sv[NQ];
#pragma unroll
        for (int q = 0; q < NQ; q++) sv[q] = Ds[min(ty + ny * q, NB - 1) * NB + tx];
#pragma unroll
        for (int q = 0; q < NQ; q++) { const int i = ty + ny * q; if (i < NB) S[i][tx] = (i < kbc && tx < i) ? sv[q] : 0.0; }
        if (tid < NB) dsr[tid] = tid < kbc ? d[first + k0c + tid] : 0.0;
        __syncthreads();
        panel_rows(Fs, ld, ns, ms, k0c, kbc, rbeg, rend, ROWS_BATCH, S, dsr, RB);
        return;
      }
      // blocked substitution, 16 rows per wavefront (4 of the 8 are busy), following the diagonal block as workgroup 0
      // publishes it 16 columns at a time
      const int R0 = rbeg + 16 * ty;
      const bool busy = R0 < rend;
      double *Tw = RB + ty * (NB * 17);
      if (!(phase == 0 && panel > 0) && busy) rows_stage(Fs, ld, rend, k0c, kbc, R0, Tw, tx);      // else staged by the update above
      for (int blk = 0; blk < NB / 16 && 16 * blk < kbc; blk++) {
        spin_until(diag_cnt + s, 4 * panel + blk + 1, tmo);            // columns 16 blk .. of L11 and their pivots are in DT / d
        for (int e = tid; e < NB * 16; e += LDL_THREADS) {
          const int i = e >> 4, j = 16 * blk + (e & 15);
          S[i][j] = (i < kbc && j < i) ? Ds[i * NB + j] : 0.0;
        }
        if (tid < 16) dsr[16 * blk + tid] = 16 * blk + tid < kbc ? d[first + k0c + 16 * blk + tid] : 0.0;
        __syncthreads();
        if (busy) rows_block(blk, S, dsr, Tw, tx);
      }
      if (busy) rows_store(Fs, ld, rend, k0c, kbc, R0, dsr, Tw, tx);
      return;
    }
    if (bx > nrw) {
      // ---- tile workgroup: two tiles side by side, 4 wavefronts each (the stand-alone kernel's shape)
      if (phase == 2 || panel == 0) return;
      const int w = bx - 1 - nrw, half = threadIdx.x >> 8, u = 2 * w + half;
      int I, J, ntl;
      bool active;
      if (nrw > 0) {                                               // block column 0 belongs to the row-solve workgroups
        ntl = (nt - 1) * nt / 2;
        if (2 * w >= ntl) return;
        active = u < ntl;
        tile_index(active ? u : 0, I, J);
        I++; J++;
      } else {
        ntl = nt * (nt + 1) / 2 - 1;                               // all tiles but (0,0)
        if (2 * w >= ntl) return;
        active = u < ntl;
        tile_index(active ? u + 1 : 1, I, J);
      }
      __shared__ double dsh2[2][NB];
      update_tile<4, false, true>(F + tab.foff[s], ld, ms, first, kp, NB, I, J, d, As + half * 2 * NB, Bs + half * 2 * NB, dsh2[half],
                                  nullptr, nullptr, 0, (int)threadIdx.x & 255, active);
      if (nrw == 0) {                                              // readers in this launch: workgroup 0's row solve / probe
        SDM_STORES_DONE();
        __syncthreads();
        if (threadIdx.x == 0) sdm_signal_add(&upd_cnt[s]);
      }
      return;
    }
    // ---- workgroup 0
    if (phase == 2) return;
    if (panel > 0)
      // tile (0,0) = this panel's diagonal block (and what lies right of / below it inside the tile): straight into S
      update_tile<LDL_THREADS / 64, true>(F + tab.foff[s], ld, ms, first, kp, NB, 0, 0, d, As, Bs, dsh,
                                          (double (*)[NB + 1])smem, (double *)smem + NB * (NB + 1), kbc);
  }
  double (*S)[NB + 1] = (double (*)[NB + 1])smem;                 // diagonal block, S[row][col]
  double *RB = (double *)smem + NB * (NB + 1);                    // Lc during the LDL', then Xs / the wave tiles of the row solve
  double *Lc = RB;                                                // Lc[k*NB+i] = l_ik
  __shared__ double ds[NB], lbs[NB], pv[NB];
  __shared__ int stt[NB];
  __shared__ int badflag, npub;
  __shared__ double red_v[LDL_THREADS];
  __shared__ int red_i[LDL_THREADS];
  const int s = list[blockIdx.y];
  const int ns = tab.ns[s], ms = tab.ms[s], ld = tab.ld[s], first = tab.first[s];
  const int k0 = panel * NB, kb = min(NB, ns - k0);
  const int r0 = k0 + kb, nrows = ms - r0;
  double *Fs = F + tab.foff[s];
  double *cb = colbuf + tab.woff[s] + s;                          // probe scratch: ms + 1 doubles per front
  const int tid = threadIdx.x, bs = blockDim.x;
  const int tx = tid & 63, ty = tid >> 6, ny = bs >> 6;
  const double maxu = ubp[1], ub = ubp[2] / (maxu * maxu);      // ubp[2] = max diagonal (k_prep_pivots)
  if (panel == 0) {
    double sv[NB / (LDL_THREADS / 64)];
    const double *pc = Fs + (int64_t)k0 * ld + k0 + min(tx, kb - 1);
#pragma unroll
    for (int q = 0; q < NB / (LDL_THREADS / 64); q++) sv[q] = pc[(int64_t)min(ty + ny * q, kb - 1) * ld];     // all loads in flight
#pragma unroll
    for (int q = 0; q < NB / (LDL_THREADS / 64); q++) {
      const int j = ty + ny * q;
      // (columns beyond a partial block: unit diagonal, so that the straight-line sweep stays finite there)
      if (j < NB) { S[tx][j] = (tx < kb && j <= tx) ? sv[q] : ((tx == j && tx >= kb) ? 1.0 : 0.0); Lc[j * NB + tx] = 0.0; }
    }
  }
  if (tid < NB) { lbs[tid] = tid < kb ? lb[first + k0 + tid] : 0.0; ds[tid] = 0.0; stt[tid] = 0; pv[tid] = 0.0; }
  if (tid == 0) { badflag = 0; npub = 0; }
  SDM_PHASE_BEGIN();
  __syncthreads();
  SDM_PHASE(16);
  // ---- LDL' of the block (see the header): wavefront 0 sweeps SW columns in registers while the other wavefronts
  // apply the previous sweep to the trailing columns.  The sweep is straight-line code: a skipped pivot gives the
  // multiplier 0, a pivot that needs the probe only raises `bad` (everything computed after it is discarded: the
  // block is redone by the general path), the bookkeeping of pivot gc lives in lane gc.
  const int nsw = (kb + SW - 1) / SW;
  if (ty == 0) {
    SDM_SETPRIO(3);
    const double mylb = lbs[tx];
    double xs[SW];                                                     // columns of the sweep just finished (unscaled)
    for (int s = -1; s < nsw - 1; s++) {
      const int c0 = s * SW, cn = c0 + SW;                             // sweep s is final; sweep columns cn .. cn+SW-1 now
      double x[SW], lsc[SW];
#pragma unroll
      for (int cc = 0; cc < SW; cc++) x[cc] = S[tx][cn + cc];
      if (s >= 0) {
        // look-ahead: the columns of the next sweep receive sweep s here (x_rj -= l_jk * x_rk, k ascending)
        // (multipliers fetched in two batches of SW/2 columns, all loads of a batch in flight before the first use)
#pragma unroll
        for (int kh = 0; kh < SW; kh += SW / 2) {
          double lj[SW / 2][SW];
#pragma unroll
          for (int k = 0; k < SW / 2; k++)
#pragma unroll
            for (int cc = 0; cc < SW; cc++) lj[k][cc] = Lc[(c0 + kh + k) * NB + cn + cc];
#pragma unroll
          for (int k = 0; k < SW / 2; k++)
#pragma unroll
            for (int cc = 0; cc < SW; cc++) SDM_PIN(lj[k][cc]);
#pragma unroll
          for (int k = 0; k < SW / 2; k++)
#pragma unroll
            for (int cc = 0; cc < SW; cc++) x[cc] -= lj[k][cc] * xs[kh + k];
        }
      }
#pragma unroll
      for (int k = 0; k < SW; k++) {
        const int gc = cn + k;
        const double xkk = sdm_bcast_lane(x[k], gc);
        const bool accept = sdm_lane_pred(x[k] > mylb, gc);           // uniform: the pivot's own lane decides (x_kk > lb_k)
        const double l = accept ? x[k] / xkk : 0.0;                    // skipped pivot: unit column
#pragma unroll
        for (int j = k + 1; j < SW; j++) x[j] -= sdm_bcast_lane(l, cn + j) * x[k];
        lsc[k] = l;
      }
      // rows above the diagonal carry don't-care values from here on (nobody reads them: every consumer of S and Lc
      // is restricted to the lower triangle), which saves the masks
#pragma unroll
      for (int k = 0; k < SW; k++) {
        Lc[(cn + k) * NB + tx] = lsc[k];
        S[tx][cn + k] = x[k];
        xs[k] = x[k];
      }
      SDM_WAVE_SYNC();
      if (tx >= cn && tx < cn + SW && tx < kb) {                       // bookkeeping of pivot tx in lane tx
        const double pval = S[tx][tx];
        const bool acc = pval > mylb;
        ds[tx] = acc ? pval : 0.0;
        if (!acc) { stt[tx] = 1; pv[tx] = pval; }
        if (acc && ms - (k0 + tx) > 1 && pval < ub) badflag = 1;       // needs the column probe: general path below
      }
      SDM_PHASE(17);
      __syncthreads();
      SDM_PHASE(19);
    }
    SDM_SETPRIO(0);
  } else if (ty < ny - 1) {
    __syncthreads();                                                   // sweep 0
    for (int s = 0; s < nsw - 1; s++) {
      const int c0 = s * SW;
      double xk[SW];
#pragma unroll
      for (int k = 0; k < SW; k++) xk[k] = S[tx][c0 + k];
      for (int j0 = c0 + 2 * SW + 4 * (ty - 1); j0 < kb; j0 += 4 * (ny - 2)) {   // 4 columns per wavefront at a time
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = S[tx][min(j0 + u, NB - 1)];
#pragma unroll
        for (int k = 0; k < SW; k++) {
          double lj[4];
#pragma unroll
          for (int u = 0; u < 4; u++) lj[u] = Lc[(c0 + k) * NB + min(j0 + u, NB - 1)];
#pragma unroll
          for (int u = 0; u < 4; u++) v[u] -= lj[u] * xk[k];
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
          if (j0 + u < kb && tx >= j0 + u) S[tx][j0 + u] = v[u];
      }
      SDM_PHASE(18);
      __syncthreads();
    }
  } else {
    // ---- the last wavefront publishes the factor as it grows: after every second sweep 16 more columns of L11 (and
    // their pivots) are final; they go to DT / d write-through and, one sweep later (the stores have been acknowledged
    // by then), the count the row-solve workgroups of this launch poll goes up by one.  Nothing is published from a
    // sweep on in which a pivot asked for the probe (the block is redone by the general path; what was published
    // before is what the general path computes again).
    double *Dsp = DT + tab.toff[s] + (int64_t)panel * NB * NB;
    int issued = 0, signalled = 0;
    __syncthreads();                                                   // sweep 0
    for (int sw = 0; sw < nsw - 1; sw++) {
      if (nrows > TRSM_ROWS) {
        if (issued > signalled) {                                      // columns stored during the previous sweep
          SDM_STORES_DONE();
          if (tx == 0) sdm_signal_add(&diag_cnt[s]);
          signalled = issued;
        }
        const int g = issued;                                          // sweeps 0 .. sw are final: columns < 8 (sw+1)
        if (SW * (sw + 1) >= 16 * (g + 1) && badflag == 0) {
#pragma unroll
          for (int c = 0; c < 16; c++) {
            const int j = 16 * g + c;
            if (tx >= j) sdm_store_wt(&Dsp[tx * NB + j], tx == j ? 1.0 : Lc[j * NB + tx]);
          }
          if (tx < 16 && 16 * g + tx < kb) sdm_store_wt(&d[first + k0 + 16 * g + tx], ds[16 * g + tx]);
          issued = g + 1;
        }
      }
      __syncthreads();
    }
    // after the last sweep: what is left of the block, right away (the epilogue below would be 2-3 us later)
    if (nrows > TRSM_ROWS && badflag == 0) {
      for (int g = issued; 16 * g < kb; g++) {
#pragma unroll
        for (int c = 0; c < 16; c++) {
          const int j = 16 * g + c;
          if (tx >= j) sdm_store_wt(&Dsp[tx * NB + j], tx == j ? 1.0 : Lc[j * NB + tx]);
        }
        if (tx < 16 && 16 * g + tx < kb) sdm_store_wt(&d[first + k0 + 16 * g + tx], ds[16 * g + tx]);
        issued = g + 1;
      }
    }
    if (issued > signalled) { SDM_STORES_DONE(); if (tx == 0) sdm_signal_add(&diag_cnt[s], issued - signalled); }
    if (tx == 0) npub = issued;
  }
  const bool bad = badflag != 0;
  const bool ok = !bad;
  if (!ok) {
    // ---- general path: one column per step by all work-items, pivot_probe available
    if (panel > 0) wait_prev_update(upd_cnt + s, ns, ms, panel, q0, tmo);  // the probe reads the rows below the block
    for (int j = ty; j < NB; j += ny) { S[tx][j] = (tx < kb && j <= tx) ? Fs[(int64_t)(k0 + j) * ld + k0 + tx] : 0.0; Lc[j * NB + tx] = 0.0; }
    if (tid < NB) { ds[tid] = 0.0; stt[tid] = 0; pv[tid] = 0.0; }
    __syncthreads();
    for (int k = 0; k < kb; k++) {
      double xkk = S[k][k];
      if (xkk > lbs[k]) {
        if (ms - (k0 + k) > 1 && xkk < ub) {                         // rare: stability probe of the never-fail rule
          double nraw = 0.0;
          if (k0 + k + 1 >= ns && first + ns < mtot) { int sidx = asm_src[Ljc[first + ns]]; nraw = sidx < 0 ? 0.0 : ada[sidx]; }
          const double ubk = pivot_probe(S, Lc, k, kb, k0, ns, ms, ld, Fs, ds, cb, nraw, red_v, red_i) / maxu;
          if (xkk < ubk) {
            if (tid == 0) { stt[k] = 2; pv[k] = ubk - xkk; lbs[k] = ubk - xkk; }
            xkk = ubk;
          }
        }
        // every work-item forms the multipliers it needs itself (same division, same rounding): one barrier per column
        const double sik = S[tx][k];
        if (tid > k && tid < kb) Lc[k * NB + tid] = sik / xkk;
        if (tid == 0) ds[k] = xkk;
        for (int i = k + 1 + ty; i < kb; i += ny)
          if (tx >= i) S[tx][i] -= (S[i][k] / xkk) * sik;
      } else {
        // skipped pivot: d = 0, the column becomes the unit vector (blkchol2.c:157-161, blkchol.c:409-414)
        if (tid == 0) { stt[k] = 1; pv[k] = xkk; ds[k] = 0.0; }
      }
      __syncthreads();
    }
  }
  SDM_PHASE(20);
  for (int j = ty; j < NB; j += ny) if (tx > j) S[tx][j] = Lc[j * NB + tx];     // scaled columns for the row solve
  __syncthreads();
  SDM_PHASE(21);
  {
    double *Ds = DT + tab.toff[s] + (int64_t)panel * NB * NB;
    // the factored block goes in place from THIS workgroup in every case: it also stored the raw updated block (tile
    // (0,0) of the previous update), and two workgroups writing the same lines in one launch may sit behind different
    // L2s whose write-back order is not defined
    const bool inplace = true;
    for (int j = ty; j < kb; j += ny)
      if (tx < kb && tx >= j) {
        const double v = (tx == j) ? 1.0 : S[tx][j];                // unit diagonal stored explicitly (blkchol2.c:136)
        if (inplace) Fs[(int64_t)(k0 + j) * ld + k0 + tx] = v;
        sdm_store_wt(&Ds[tx * NB + j], v);                          // transposed copy of the block (backward solve, row solve)
      }
    if (tid < kb) {
      const int gk = first + k0 + tid;
      sdm_store_wt(&d[gk], ds[tid]);
      if (stt[tid]) { pivstat[gk] = stt[tid]; pivval[gk] = pv[tid]; }   // pivval = amount added (what blkchol2.c:127 keeps in lb[k])
    }
  }
  if (nrows > TRSM_ROWS) {                                         // the row-solve workgroups of this launch are waiting
    if (16 * npub < kb) SDM_STORES_DONE();                         // (uniform) something of the block is still unpublished
    __syncthreads();
    if (tid == 0) sdm_signal_add(&diag_cnt[s], 4 - npub);          // 4 counts per panel: one per 16 columns of the block
    // a partial block (kb < 64, last panel of the supernode) leaves rows r0 .. k0+63 in this workgroup's own tile row:
    // they were updated by its tile (0,0) and are solved here (the row-solve workgroups own whole tile rows)
    if (kb < NB) {
      SDM_ACQUIRE_FENCE();                                       // its own tile-(0,0) stores, not a cached copy from before them
      panel_rows(Fs, ld, ns, ms, k0, kb, r0, k0 + NB, TRSM_ROWS, S, ds, RB);
    }
  }
  SDM_PHASE(22);
  if (nrows > 0 && nrows <= TRSM_ROWS) {
    if (panel > 0 && ok) wait_prev_update(upd_cnt + s, ns, ms, panel, q0, tmo);
    panel_rows(Fs, ld, ns, ms, k0, kb, r0, ms, TRSM_ROWS, S, ds, RB);
  }
  if (nrows <= TRSM_ROWS) {
    // nobody in this launch waits for this block: the count (= 4 x panels done) goes up at the very end, behind the
    // same stores-acknowledged / barrier sequence as every other publication
    SDM_STORES_DONE();
    __syncthreads();
    if (tid == 0) sdm_signal_add(&diag_cnt[s], 4);
  }
  SDM_PHASE(23);
}

// stand-alone update.  Supernodes that END with this panel: all tiles (the update of the rows beyond, passed up to the
// parent).  Supernodes with a next panel: nothing when `riding` (their tiles ride along with the diagonal-block launch
// of the next panel, k_ldl_panel), else every tile but tile 0 (which k_ldl_panel's workgroup 0 always applies itself).
__global__ void __launch_bounds__(256)
k_ldl_update(double *F, FrontTab tab, const int *list, int panel, const double *d, int riding) {
  __shared__ double As[NB][TILE];
  __shared__ double Bs[NB][TILE];
  __shared__ double dsh[NB];
  const int s = list[blockIdx.y];
  const int ns = tab.ns[s], ms = tab.ms[s], ld = tab.ld[s], first = tab.first[s];
  const int k0 = panel * NB, kb = min(NB, ns - k0);
  const bool has_next = ns > k0 + NB;
  if (has_next && riding) return;
  const int nrem = ms - (k0 + kb);
  const int nt = (nrem + TILE - 1) / TILE;
  const int t = blockIdx.x + (has_next ? 1 : 0);
  if (t >= nt * (nt + 1) / 2) return;
  int I, J;
  tile_index(t, I, J);
  update_tile<4, false>(F + tab.foff[s], ld, ms, first, k0, kb, I, J, d, As, Bs, dsh);
}

// ================================================================== solves
// One workgroup per front; fronts of an etree level are independent.  The front-local vector w lives in LDS
// (HBM scratch for fronts beyond SOLVE_LDS_MAX rows).  Per 64-column panel ONE wavefront does the in-block
// triangular solve -- lane i owns row i, its 64 coefficients sit in registers (prefetched while the previous
// panel streams) and the dependency chain is a v_readlane broadcast + one FMA per column (no LDS, no barrier)
// -- while all waves stream the panel below the block exactly once with coalesced reads:
//   forward  (fwblkslv.c:77-134): one row per work-item, 16 independent loads in flight per lane;
//   backward (bwblkslv.c:73-125): one column per wavefront at a time (contiguous reads), wave reduction.
// The transposed in-block solve of the backward sweep reads the 64x64 diagonal blocks from the compact
// transposed copy DT written by the factor (coalesced for lane = column).
// Stage the strictly lower triangle of a kb x kb diagonal block into LDS (Sd[c*64 + i] = L(k0+i, k0+c) for
// c < i < kb, 0 elsewhere) -- all work-items, coalesced along the rows.  The in-block triangular solve then
// reads its coefficient of step k with one conflict-free ds_read (address independent of the dependency chain).
__device__ __forceinline__ void stage_block(double *Sd, const double *blk, int ld, int kb) {
  for (int idx = threadIdx.x; idx < SNB * SNB; idx += blockDim.x) {
    const int i = idx & 63, c = idx >> 6;
    Sd[idx] = (c < i && i < kb) ? blk[(int64_t)c * ld + i] : 0.0;
  }
}
// the same in two halves for workgroups of SOLVE_THREADS work-items: the (clamped, unconditional) loads are issued
// early, the masked LDS stores are done after other memory traffic has been issued
constexpr int STG = SNB * SNB / SOLVE_THREADS;
__device__ __forceinline__ void stage_block_load(double (&sv)[STG], const double *blk, int ld, int kb) {
#pragma unroll
  for (int q = 0; q < STG; q++) {
    const int idx = threadIdx.x + q * SOLVE_THREADS, i = idx & 63, c = idx >> 6;
    sv[q] = blk[(int64_t)min(c, kb - 1) * ld + min(i, kb - 1)];
  }
}
__device__ __forceinline__ void stage_block_store(double *Sd, const double (&sv)[STG], int kb) {
#pragma unroll
  for (int q = 0; q < STG; q++) {
    const int idx = threadIdx.x + q * SOLVE_THREADS, i = idx & 63, c = idx >> 6;
    Sd[idx] = (c < i && i < kb) ? sv[q] : 0.0;
  }
}
// the same from the transposed copy DT (Dp[c*64 + i] = L(k0+c, k0+i)): Sd[c*64 + i] = L(k0+c, k0+i) for i < c < kb
__device__ __forceinline__ void stage_blockT(double *Sd, const double *Dp, int kb) {
  for (int idx = threadIdx.x; idx < SNB * SNB; idx += blockDim.x) {
    const int i = idx & 63, c = idx >> 6;
    Sd[idx] = (c > i && c < kb) ? Dp[idx] : 0.0;
  }
}

constexpr int TCH = 32;  // in-block solve: coefficients fetched from LDS per batch (one LDS round trip per TCH steps of the chain)

// ---- panel streaming helpers.  The panel below a diagonal block is read exactly once, with 16-byte loads (ld
// and the first row `ra` of the pair range are even), 16 loads per lane in flight.
// Forward: a row pair is shared by the 4 lanes {l, l+16, l+32, l+48} of a wavefront (g = lane >> 4), each owning
// 16 of the 64 columns, so that the 16 lanes of one g read 256 contiguous bytes of one column (lanes that are
// neighbours in a quad must not straddle columns: the L1 coalescer then handles one line per lane, measured 4x
// slower); the 4 partial sums meet in a fixed-order reduction across g (deterministic).
__device__ __forceinline__ void fw_issue(sdm_double2 (&v)[16], const double *Fs, int ld, int k0, int ra, int npair, int t, int g) {
  const sdm_double2 *col = (const sdm_double2 *)(Fs + (int64_t)(k0 + 16 * g) * ld + ra + 2 * min(t, npair - 1));
  const int ld2 = ld >> 1;
#pragma unroll
  for (int c = 0; c < 16; c++) v[c] = col[(int64_t)c * ld2];
}
__device__ __forceinline__ void fw_consume(const sdm_double2 (&v)[16], const double *wb, double *w, int ra, int npair, int t, int g) {
  double a0 = 0.0, a1 = 0.0;
#pragma unroll
  for (int c = 0; c < 16; c++) { a0 += v[c].x * wb[16 * g + c]; a1 += v[c].y * wb[16 * g + c]; }
  a0 += __shfl_xor(a0, 16); a1 += __shfl_xor(a1, 16);
  a0 += __shfl_xor(a0, 32); a1 += __shfl_xor(a1, 32);
  if (g == 0 && t < npair) { w[ra + 2 * t] -= a0; w[ra + 2 * t + 1] -= a1; }
}
// Backward: a wavefront sweeps 4 columns at a time, lanes along the row pairs, 4 pair-chunks of 64 per round.
__device__ __forceinline__ void bw_issue(sdm_double2 (&v)[16], const double *Fs, int ld, int k0, int kb, int cb0, int ra, int npair, int t0) {
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int tc = min(t0 + 64 * i, npair - 1);
#pragma unroll
    for (int q = 0; q < 4; q++)
      v[4 * i + q] = ((const sdm_double2 *)(Fs + (int64_t)(k0 + min(cb0 + q, kb - 1)) * ld + ra))[tc];
  }
}
__device__ __forceinline__ void bw_consume(const sdm_double2 (&v)[16], double (&acc)[4], const double *w, int ra, int npair, int t0) {
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int t = t0 + 64 * i;
    if (t < npair) {
      const double w0 = w[ra + 2 * t], w1 = w[ra + 2 * t + 1];
#pragma unroll
      for (int q = 0; q < 4; q++) acc[q] += v[4 * i + q].x * w0 + v[4 * i + q].y * w1;
    }
  }
}

__device__ __forceinline__ void front_fw_small(const double *Fs, int ns, int ms, int ld, double *w, double *wb, double *Sd) {
  const int tid = threadIdx.x, bs = blockDim.x;
  const int g = (tid >> 4) & 3, t0 = (tid >> 6) * 16 + (tid & 15), tstep = bs >> 2;
  SDM_PHASE_BEGIN();
  stage_block(Sd, Fs, ld, min(SNB, ns));
  __syncthreads();
  SDM_PHASE(0);
  for (int k0 = 0; k0 < ns; k0 += SNB) {
    const int kb = min(SNB, ns - k0);
    if (tid < 64) {
      double wi = tid < kb ? w[k0 + tid] : 0.0;
      // coefficients of TCH steps are pulled into registers at once (one LDS round trip per TCH steps of the
      // dependency chain instead of one per step); 0 for lanes <= k and for k >= kb
#pragma unroll
      for (int h = 0; h < SNB; h += TCH) {
        double lr[TCH];
#pragma unroll
        for (int k = 0; k < TCH; k++) lr[k] = Sd[(h + k) * SNB + tid];
#pragma unroll
        for (int k = 0; k < TCH; k++) SDM_PIN(lr[k]);
#pragma unroll
        for (int k = 0; k < TCH; k++) wi -= lr[k] * sdm_bcast_lane(wi, h + k);
      }
      if (tid < kb) { w[k0 + tid] = wi; wb[tid] = wi; }
    }
    SDM_PHASE(1);
    __syncthreads();
    SDM_PHASE(2);
    const int k1 = k0 + SNB;
    const bool fast_stage = k1 < ns && bs == SOLVE_THREADS;
    double sv[STG];
    if (fast_stage) stage_block_load(sv, Fs + (int64_t)k1 * ld + k1, ld, min(SNB, ns - k1));   // next diagonal block: loads now
    else if (k1 < ns) stage_block(Sd, Fs + (int64_t)k1 * ld + k1, ld, min(SNB, ns - k1));
    SDM_PHASE(3);
    const int rb = k0 + kb, ra = rb + (rb & 1);
    const int npair = ms > ra ? (ms - ra) >> 1 : 0;
    if (kb == SNB) {
      const int lim = (npair + 15) & ~15;                // whole quads / waves stay converged for the shuffles
      for (int t = t0; t < lim; t += tstep) {
        sdm_double2 v[16];
        fw_issue(v, Fs, ld, k0, ra, npair, t, g);
        fw_consume(v, wb, w, ra, npair, t, g);
      }
    } else {
      for (int t = tid; t < npair; t += bs) {
        const int r = ra + 2 * t;
        const sdm_double2 *col = (const sdm_double2 *)(Fs + (int64_t)k0 * ld + r);
        double a0 = 0.0, a1 = 0.0;
        for (int c = 0; c < kb; c++) { const sdm_double2 x = col[(int64_t)c * (ld >> 1)]; a0 += x.x * wb[c]; a1 += x.y * wb[c]; }
        w[r] -= a0; w[r + 1] -= a1;
      }
    }
    {  // the (at most two) unpaired rows: rb when odd, the last row when the pair range leaves one over
      int r = -1;
      if (tid == bs - 1 && (rb & 1) && rb < ms) r = rb;
      if (tid == bs - 2 && ms > ra && ((ms - ra) & 1)) r = ms - 1;
      if (r >= 0) {
        const double *col = Fs + (int64_t)k0 * ld + r;
        double acc = 0.0;
        for (int c = 0; c < kb; c++) acc += col[(int64_t)c * ld] * wb[c];
        w[r] -= acc;
      }
    }
    if (fast_stage) stage_block_store(Sd, sv, min(SNB, ns - k1));                 // ... LDS stores after the panel stream
    SDM_PHASE(4);
    __syncthreads();
    SDM_PHASE(5);
  }
}

__device__ __forceinline__ void front_bw_small(const double *Fs, const double *Ds, int ns, int ms, int ld, double *w, double *dots, double *Sd) {
  const int tid = threadIdx.x, bs = blockDim.x;
  const int wave = tid >> 6, lane = tid & 63, nw = bs >> 6;
  const int npan = (ns + SNB - 1) / SNB;
  SDM_PHASE_BEGIN();
  for (int pnl = npan - 1; pnl >= 0; pnl--) {
    const int k0 = pnl * SNB, kb = min(SNB, ns - k0);
    const int rb = k0 + kb, ra = rb + (rb & 1);
    const int npair = ms > ra ? (ms - ra) >> 1 : 0;
    stage_blockT(Sd, Ds + (int64_t)pnl * SNB * SNB, kb);
    // dots[c] = sum_{r >= rb} L(r, k0+c) * w[r]
    if (rb < ms) {
      for (int cb0 = wave * 4; cb0 < kb; cb0 += nw * 4) {
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        for (int tb = lane; tb < npair; tb += 256) {
          sdm_double2 v[16];
          bw_issue(v, Fs, ld, k0, kb, cb0, ra, npair, tb);
          bw_consume(v, acc, w, ra, npair, tb);
        }
        if (lane == 0) {
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const double *colp = Fs + (int64_t)(k0 + min(cb0 + q, kb - 1)) * ld;
            if (rb & 1) acc[q] += colp[rb] * w[rb];
            if (ms > ra && ((ms - ra) & 1)) acc[q] += colp[ms - 1] * w[ms - 1];
          }
        }
        // wave reduction of the 4 column sums with 7 shuffles instead of 24: fold the columns into the lane index
        // first (upper half wave keeps columns 2,3, then odd 16-lane groups keep the odd column), then 4 plain steps
        {
          const bool hi = lane >= 32;
          const double s0 = hi ? acc[0] : acc[2], s1 = hi ? acc[1] : acc[3];      // what the partner half keeps
          double k0v = (hi ? acc[2] : acc[0]) + __shfl_xor(s0, 32);
          double k1v = (hi ? acc[3] : acc[1]) + __shfl_xor(s1, 32);
          const bool od = (lane >> 4) & 1;
          double a = (od ? k1v : k0v) + __shfl_xor(od ? k0v : k1v, 16);
          a += __shfl_xor(a, 8); a += __shfl_xor(a, 4); a += __shfl_xor(a, 2); a += __shfl_xor(a, 1);
          const int q = (hi ? 2 : 0) + (od ? 1 : 0);                               // column held by this 16-lane group
          if ((lane & 15) == 0 && cb0 + q < kb) dots[cb0 + q] = a;
        }
      }
    }
    SDM_PHASE(8);
    __syncthreads();
    SDM_PHASE(9);
    if (tid < 64) {
      double yi = tid < kb ? w[k0 + tid] - (rb < ms ? dots[tid] : 0.0) : 0.0;
#pragma unroll
      for (int h = SNB - TCH; h >= 0; h -= TCH) {       // L(k0+k, k0+tid), 0 unless k > tid
        double lr[TCH];
#pragma unroll
        for (int k = 0; k < TCH; k++) lr[k] = Sd[(h + k) * SNB + tid];
#pragma unroll
        for (int k = 0; k < TCH; k++) SDM_PIN(lr[k]);
#pragma unroll
        for (int k = TCH - 1; k >= 0; k--) yi -= lr[k] * sdm_bcast_lane(yi, h + k);
      }
      if (tid < kb) w[k0 + tid] = yi;
    }
    SDM_PHASE(10);
    __syncthreads();
    SDM_PHASE(11);
  }
}


// ---------------------------------------------------------------- pipelined sweeps (workgroups of SOLVE_THREADS)
// The in-block triangular solve is a 64-step dependency chain (~1.1 us) on ONE wavefront; streaming the panel is
// bandwidth work for the others.  The two overlap.  The rows below a panel are split into
//  * the 64 rows right below it (the next diagonal block, needed by the next in-block solve): their update is
//    accumulated by wavefront 0 INSIDE the chain -- step k broadcasts x_k anyway, one more FMA with the coefficient
//    of the sub-diagonal block costs nothing on a latency-bound chain -- so the next solve starts right after ONE
//    barrier;
//  * the rows beyond, streamed by the other wavefronts while wavefront 0 is already solving the next block.
// The diagonal block and the sub-diagonal block of the next step are staged in LDS by the streaming wavefronts,
// alternately in the two halves of Sd2 / Sb2 (the backward sweep stores the sub-diagonal block transposed, pitch
// SBP); x_p / the dots of the rows beyond are double-buffered.
constexpr int BW_FARW = 13;              // backward: wavefronts 1..13 form the dots beyond (5 columns each), 14..15 stage
constexpr int STG15 = (SNB * SNB + (SOLVE_THREADS - 64) - 1) / (SOLVE_THREADS - 64);   // staging by wavefronts 1..15
constexpr int SBP = SNB + 1;             // row pitch of the transposed sub-diagonal block
constexpr int TCF = 8;                   // fused in-block solve: steps per batch of coefficients (two register sets in ping-pong)
constexpr int SOLVE_STAGE_DOUBLES = 2 * SNB * SNB + 2 * SNB * SBP;   // Sd2 then Sb2 at the head of the dynamic LDS

__device__ __forceinline__ void stage15_load(double (&sv)[STG15], const double *blk, int ld, int kb) {
#pragma unroll
  for (int q = 0; q < STG15; q++) {
    const int idx = min((int)threadIdx.x - 64 + q * (SOLVE_THREADS - 64), SNB * SNB - 1), i = idx & 63, c = idx >> 6;
    sv[q] = blk[(int64_t)min(c, kb - 1) * ld + min(i, kb - 1)];
  }
}
__device__ __forceinline__ void stage15_store(double *Sd, const double (&sv)[STG15], int kb) {
#pragma unroll
  for (int q = 0; q < STG15; q++) {
    const int idx = (int)threadIdx.x - 64 + q * (SOLVE_THREADS - 64), i = idx & 63, c = idx >> 6;
    if (idx < SNB * SNB) Sd[idx] = (c < i && i < kb) ? sv[q] : 0.0;
  }
}

// in-block forward solve by one wavefront (lane = row): coefficients of TCH steps are pulled into registers at once
// (one LDS round trip per TCH steps of the dependency chain); they are 0 for lanes <= k and for k >= kb
__device__ __forceinline__ double trsv_fw_block(const double *Sd, double wi, int lane) {
#pragma unroll
  for (int h = 0; h < SNB; h += TCH) {
    double lr[TCH];
#pragma unroll
    for (int k = 0; k < TCH; k++) lr[k] = Sd[(h + k) * SNB + lane];
#pragma unroll
    for (int k = 0; k < TCH; k++) SDM_PIN(lr[k]);
#pragma unroll
    for (int k = 0; k < TCH; k++) wi -= lr[k] * sdm_bcast_lane(wi, h + k);
  }
  return wi;
}
__device__ __forceinline__ double trsv_bw_block(const double *Sd, double yi, int lane) {
#pragma unroll
  for (int h = SNB - TCH; h >= 0; h -= TCH) {       // L(k0+k, k0+lane), 0 unless k > lane
    double lr[TCH];
#pragma unroll
    for (int k = 0; k < TCH; k++) lr[k] = Sd[(h + k) * SNB + lane];
#pragma unroll
    for (int k = 0; k < TCH; k++) SDM_PIN(lr[k]);
#pragma unroll
    for (int k = TCH - 1; k >= 0; k--) yi -= lr[k] * sdm_bcast_lane(yi, h + k);
  }
  return yi;
}
// the same chain with the update of the next block riding along: cacc += sum_k Sb[k*pitch + lane] * x_k.
// Coefficients travel in batches of TCF steps, two register sets in ping-pong: the loads of a batch are issued when
// the batch before it starts (SDM_ZERO_AFTER pins them there) and land while its TCF steps run.
#define SDM_FUSED_LOAD(lr, ls, h, z, PITCH)                                                       \
  _Pragma("unroll") for (int k = 0; k < TCF; k++) { lr[k] = Sd[((h) + k) * SNB + lane + (z)]; ls[k] = Sb[((h) + k) * (PITCH) + lane + (z)]; }
__device__ __forceinline__ double trsv_fw_fused(const double *Sd, const double *Sb, double wi, double &cacc, int lane) {
  double lrA[TCF], lsA[TCF], lrB[TCF], lsB[TCF];
  SDM_FUSED_LOAD(lrA, lsA, 0, 0, SNB)
#pragma unroll
  for (int h = 0; h < SNB; h += 2 * TCF) {
    { const int z = SDM_ZERO_AFTER(wi); SDM_FUSED_LOAD(lrB, lsB, h + TCF, z, SNB) }
#pragma unroll
    for (int k = 0; k < TCF; k++) { const double xk = sdm_bcast_lane(wi, h + k); wi -= lrA[k] * xk; cacc += lsA[k] * xk; }
    if (h + 2 * TCF < SNB) { const int z = SDM_ZERO_AFTER(wi); SDM_FUSED_LOAD(lrA, lsA, h + 2 * TCF, z, SNB) }
#pragma unroll
    for (int k = 0; k < TCF; k++) { const double xk = sdm_bcast_lane(wi, h + TCF + k); wi -= lrB[k] * xk; cacc += lsB[k] * xk; }
  }
  return wi;
}
__device__ __forceinline__ double trsv_bw_fused(const double *Sd, const double *Sb, double yi, double &cacc, int lane) {
  double lrA[TCF], lsA[TCF], lrB[TCF], lsB[TCF];
  SDM_FUSED_LOAD(lrA, lsA, SNB - TCF, 0, SBP)
#pragma unroll
  for (int h = SNB - TCF; h >= 0; h -= 2 * TCF) {
    { const int z = SDM_ZERO_AFTER(yi); SDM_FUSED_LOAD(lrB, lsB, h - TCF, z, SBP) }
#pragma unroll
    for (int k = TCF - 1; k >= 0; k--) { const double xk = sdm_bcast_lane(yi, h + k); yi -= lrA[k] * xk; cacc += lsA[k] * xk; }
    if (h - 2 * TCF >= 0) { const int z = SDM_ZERO_AFTER(yi); SDM_FUSED_LOAD(lrA, lsA, h - 2 * TCF, z, SBP) }
#pragma unroll
    for (int k = TCF - 1; k >= 0; k--) { const double xk = sdm_bcast_lane(yi, h - TCF + k); yi -= lrB[k] * xk; cacc += lsB[k] * xk; }
  }
  return yi;
}
// one row r below a kb-column panel (rows that do not pair up): w[r] -= L(r, k0:k0+kb) . x
__device__ __forceinline__ void fw_single_row(const double *Fs, int ld, int k0, int kb, int r, const double *x, double *w) {
  const double *col = Fs + (int64_t)k0 * ld + r;
  double acc = 0.0;
  for (int c = 0; c < kb; c++) acc += col[(int64_t)c * ld] * x[c];
  w[r] -= acc;
}
// wave reduction of 4 column sums with 7 shuffles instead of 24: fold the columns into the lane index first (upper
// half wave keeps columns 2,3, then odd 16-lane groups keep the odd column), then 4 plain steps; lanes with
// (lane & 15) == 0 end up holding column q = 2*(lane >= 32) + ((lane >> 4) & 1)
__device__ __forceinline__ double fold4(const double (&acc)[4], int lane) {
  const bool hi = lane >= 32;
  const double s0 = hi ? acc[0] : acc[2], s1 = hi ? acc[1] : acc[3];      // what the partner half keeps
  double k0v = (hi ? acc[2] : acc[0]) + __shfl_xor(s0, 32);
  double k1v = (hi ? acc[3] : acc[1]) + __shfl_xor(s1, 32);
  const bool od = (lane >> 4) & 1;
  double a = (od ? k1v : k0v) + __shfl_xor(od ? k0v : k1v, 16);
  a += __shfl_xor(a, 8); a += __shfl_xor(a, 4); a += __shfl_xor(a, 2); a += __shfl_xor(a, 1);
  return a;
}
// dots[c] = sum over the rows [ra, ms) of L(r, k0+c) w[r] for the 4 columns cb0..cb0+3 (one wavefront; ra even)
__device__ __forceinline__ void bw_far_quad(const double *Fs, int ld, int k0, int kb, int cb0, int ra, int ms, const double *w,
                                            double *dots, int lane) {
  const int npair = ms > ra ? (ms - ra) >> 1 : 0;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int tb = lane; tb < npair; tb += 256) {
    sdm_double2 v[16];
    bw_issue(v, Fs, ld, k0, kb, cb0, ra, npair, tb);
    bw_consume(v, acc, w, ra, npair, tb);
  }
  if (lane == 0 && ms > ra && ((ms - ra) & 1)) {
#pragma unroll
    for (int q = 0; q < 4; q++) acc[q] += Fs[(int64_t)(k0 + min(cb0 + q, kb - 1)) * ld + ms - 1] * w[ms - 1];
  }
  const double a = fold4(acc, lane);
  const int q = (lane >= 32 ? 2 : 0) + ((lane >> 4) & 1);
  if ((lane & 15) == 0 && cb0 + q < kb) dots[cb0 + q] = a;
}
// dots over the rows [ra, ms) for the 5 columns c0 .. c0+4 of a full panel (one wavefront; columns beyond the panel
// are clamped and dropped): BW_NCH chunks of 64 row pairs x 5 columns sixteen-byte loads per lane in flight
constexpr int BW_NCH = 3;
__device__ __forceinline__ void bw_far_five(const double *Fs, int ld, int k0, int c0, int ra, int ms, const double *w,
                                            double *dots, int lane) {
  const int npair = ms > ra ? (ms - ra) >> 1 : 0;
  double acc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  const sdm_double2 *cp = (const sdm_double2 *)(Fs + (int64_t)(k0 + c0) * ld + ra);
  const int ld2 = ld >> 1;
  for (int tb = lane; tb < npair; tb += 64 * BW_NCH) {
    sdm_double2 v[5 * BW_NCH];
#pragma unroll
    for (int i = 0; i < BW_NCH; i++) {
      const int tc = min(tb + 64 * i, npair - 1);
#pragma unroll
      for (int q = 0; q < 5; q++) v[5 * i + q] = cp[(int64_t)min(q, SNB - 1 - c0) * ld2 + tc];
    }
#pragma unroll
    for (int i = 0; i < BW_NCH; i++) {
      const int t = tb + 64 * i;
      if (t < npair) {
        const double w0 = w[ra + 2 * t], w1 = w[ra + 2 * t + 1];
#pragma unroll
        for (int q = 0; q < 5; q++) acc[q] += v[5 * i + q].x * w0 + v[5 * i + q].y * w1;
      }
    }
  }
  if (lane == 0 && ms > ra && ((ms - ra) & 1)) {
#pragma unroll
    for (int q = 0; q < 5; q++) acc[q] += Fs[(int64_t)(k0 + min(c0 + q, SNB - 1)) * ld + ms - 1] * w[ms - 1];
  }
  const double a4[4] = {acc[0], acc[1], acc[2], acc[3]};
  const double a = fold4(a4, lane);
  double e = acc[4];
  e += __shfl_xor(e, 32); e += __shfl_xor(e, 16); e += __shfl_xor(e, 8); e += __shfl_xor(e, 4); e += __shfl_xor(e, 2); e += __shfl_xor(e, 1);
  const int q = (lane >= 32 ? 2 : 0) + ((lane >> 4) & 1);
  if ((lane & 15) == 0 && c0 + q < SNB) dots[c0 + q] = a;
  if (lane == 1 && c0 + 4 < SNB) dots[c0 + 4] = e;
}

// sub-diagonal block below full panel k0 for the forward sweep: Sb[k*64 + i] = L(k0+64+i, k0+k), 0 for rows >= ms
__device__ __forceinline__ void stage15_sub_load(double (&sv)[STG15], const double *Fs, int ld, int k0, int ms) {
  const int nsub = min(SNB, ms - (k0 + SNB));
#pragma unroll
  for (int q = 0; q < STG15; q++) {
    const int idx = min((int)threadIdx.x - 64 + q * (SOLVE_THREADS - 64), SNB * SNB - 1), i = idx & 63, k = idx >> 6;
    sv[q] = nsub > 0 ? Fs[(int64_t)(k0 + k) * ld + k0 + SNB + min(i, nsub - 1)] : 0.0;
  }
}
__device__ __forceinline__ void stage15_sub_store(double *Sb, const double (&sv)[STG15], int k0, int ms) {
  const int nsub = min(SNB, ms - (k0 + SNB));
#pragma unroll
  for (int q = 0; q < STG15; q++) {
    const int idx = (int)threadIdx.x - 64 + q * (SOLVE_THREADS - 64), i = idx & 63;
    if (idx < SNB * SNB) Sb[idx] = i < nsub ? sv[q] : 0.0;
  }
}

// forward sweep of one front, workgroup of SOLVE_THREADS; wb2 = 2*SNB doubles, Sd2 = SOLVE_STAGE_DOUBLES (LDS).
// Wavefront 0 and the others run their own loops over the full panels (one barrier per panel each) -- separate loops
// keep the register live ranges of the roles apart.
__device__ __forceinline__ void front_fw_pipe(const double *Fs, int ns, int ms, int ld, double *w, double *wb2, double *Sd2) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = (tid >> 4) & 3;
  const int nfull = ns / SNB, rem = ns - nfull * SNB;
  double *Sb2 = Sd2 + 2 * SNB * SNB;
  SDM_LPHASE_BEGIN();
  stage_block(Sd2, Fs, ld, min(SNB, ns));
  if (nfull > 0) {
    const int nsub = min(SNB, ms - SNB);
    for (int idx = tid; idx < SNB * SNB; idx += SOLVE_THREADS) {
      const int i = idx & 63, k = idx >> 6;
      Sb2[idx] = i < nsub ? Fs[(int64_t)k * ld + SNB + i] : 0.0;
    }
  }
  __syncthreads();
  SDM_LPHASE(0);
  if (wave == 0) {
    // ---- in-block solves; cacc = what panel p owes to the rows of block p+1
    SDM_SETPRIO(3);                                                // the dependency chain ahead of the streaming wavefronts
    double cacc = 0.0;
    for (int p = 0; p < nfull; p++) {
      const int k0 = p * SNB;
      double cnext = 0.0;
      const double wi = trsv_fw_fused(Sd2 + (p & 1) * SNB * SNB, Sb2 + (p & 1) * SNB * SNB, w[k0 + lane] - cacc, cnext, lane);
      cacc = cnext;
      w[k0 + lane] = wi;
      wb2[(p & 1) * SNB + lane] = wi;
      SDM_LPHASE(1);
      __syncthreads();
      SDM_LPHASE(2);
    }
    if (nfull > 0 && nfull * SNB + lane < ms) w[nfull * SNB + lane] -= cacc;      // rows right below the last full panel
  } else {
    // ---- the rows beyond: panel p-1 is streamed while wavefront 0 solves block p; blocks of step p+1 staged
    for (int p = 0; p < nfull; p++) {
      const int k0 = p * SNB, k1 = k0 + SNB;
      double sv[STG15], sb[STG15];
      if (k1 < ns) stage15_load(sv, Fs + (int64_t)k1 * ld + k1, ld, min(SNB, ns - k1));      // next diagonal block: loads now
      if (p + 1 < nfull) stage15_sub_load(sb, Fs, ld, k1, ms);
      if (p > 0) {
        const int kp = k0 - SNB, ra = k0 + SNB;                                              // rows beyond block p
        const int npair = ms > ra ? (ms - ra) >> 1 : 0, lim = (npair + 15) & ~15;
        const double *wbq = wb2 + ((p - 1) & 1) * SNB;
        for (int t = (wave - 1) * 16 + (lane & 15); t < lim; t += (SOLVE_THREADS / 64 - 1) * 16) {
          sdm_double2 v[16];
          fw_issue(v, Fs, ld, kp, ra, npair, t, g);
          fw_consume(v, wbq, w, ra, npair, t, g);
        }
        if (tid == SOLVE_THREADS - 1 && ms > ra && ((ms - ra) & 1)) fw_single_row(Fs, ld, kp, SNB, ms - 1, wbq, w);
      }
      if (k1 < ns) stage15_store(Sd2 + ((p + 1) & 1) * SNB * SNB, sv, min(SNB, ns - k1));    // ... LDS stores after the stream
      if (p + 1 < nfull) stage15_sub_store(Sb2 + ((p + 1) & 1) * SNB * SNB, sb, k1, ms);
      SDM_LPHASE(4);
      __syncthreads();
    }
  }
  __syncthreads();
  // ---- common tail: what the last full panel still owes to the rows beyond, then a partial last panel
  if (nfull > 0 && ms > nfull * SNB + SNB) {
    const int kp = (nfull - 1) * SNB, ra = nfull * SNB + SNB;
    const int npair = (ms - ra) >> 1, lim = (npair + 15) & ~15;
    const double *wbq = wb2 + ((nfull - 1) & 1) * SNB;
    for (int t = wave * 16 + (lane & 15); t < lim; t += SOLVE_THREADS / 4) {
      sdm_double2 v[16];
      fw_issue(v, Fs, ld, kp, ra, npair, t, g);
      fw_consume(v, wbq, w, ra, npair, t, g);
    }
    if (tid == SOLVE_THREADS - 1 && ((ms - ra) & 1)) fw_single_row(Fs, ld, kp, SNB, ms - 1, wbq, w);
    __syncthreads();
  }
  if (rem > 0) {
    const int k0 = nfull * SNB, kb = rem, rb = ns;
    double *wbp = wb2 + (nfull & 1) * SNB;
    if (wave == 0) {
      const double wi = trsv_fw_block(Sd2 + (nfull & 1) * SNB * SNB, lane < kb ? w[k0 + lane] : 0.0, lane);
      if (lane < kb) w[k0 + lane] = wi;
      wbp[lane] = wi;
    }
    __syncthreads();
    if (rb < ms) {
      const int ra = rb + (rb & 1), npair = ms > ra ? (ms - ra) >> 1 : 0;
      for (int t = tid; t < npair; t += SOLVE_THREADS) {
        const int r = ra + 2 * t;
        const sdm_double2 *col = (const sdm_double2 *)(Fs + (int64_t)k0 * ld + r);
        double a0 = 0.0, a1 = 0.0;
        for (int c = 0; c < kb; c++) { const sdm_double2 x = col[(int64_t)c * (ld >> 1)]; a0 += x.x * wbp[c]; a1 += x.y * wbp[c]; }
        w[r] -= a0; w[r + 1] -= a1;
      }
      if (tid == SOLVE_THREADS - 1 && (rb & 1)) fw_single_row(Fs, ld, k0, kb, rb, wbp, w);
      if (tid == SOLVE_THREADS - 2 && ms > ra && ((ms - ra) & 1)) fw_single_row(Fs, ld, k0, kb, ms - 1, wbp, w);
      __syncthreads();
    }
  }
  SDM_LPHASE(5);
  SDM_LPHASE_END();
}

// backward sweep of one front, workgroup of SOLVE_THREADS; dots3 = 2*SNB doubles, Sd2 = SOLVE_STAGE_DOUBLES (LDS)
__device__ __forceinline__ void front_bw_pipe(const double *Fs, const double *Ds, int ns, int ms, int ld, double *w, double *dots3,
                                              double *Sd2) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int npan = (ns + SNB - 1) / SNB;
  double *Sb2 = Sd2 + 2 * SNB * SNB;
  SDM_LPHASE_BEGIN();
  {
    // last panel: nothing to overlap with.  Its dots over ALL rows below it; its diagonal block, and (transposed) the
    // block left of it, which the in-block solve uses to form what panel P-1 gets from the rows of block P.
    const int P = npan - 1, k0 = P * SNB, kb = ns - k0, rb = ns;
    double *dF = dots3 + (P & 1) * SNB;
    stage_blockT(Sd2 + (P & 1) * SNB * SNB, Ds + (int64_t)P * SNB * SNB, kb);
    if (P > 0) {
      double *Sn = Sb2 + (P & 1) * SNB * SBP;
      for (int idx = tid; idx < SNB * SNB; idx += SOLVE_THREADS) {
        const int i = idx & 63, c = idx >> 6;
        Sn[i * SBP + c] = k0 + i < ms ? Fs[(int64_t)(k0 - SNB + c) * ld + k0 + i] : 0.0;   // incl. rows beyond a partial block
      }
    }
    if (4 * wave < kb) {
      const int ra = rb + (rb & 1), cb0 = 4 * wave;               // rb is odd only for a partial panel
      bw_far_quad(Fs, ld, k0, kb, cb0, ra, ms, w, dF, lane);
      SDM_WAVE_SYNC();
      if (lane < 4 && cb0 + lane < kb && (rb & 1) && rb < ms) dF[cb0 + lane] += Fs[(int64_t)(k0 + cb0 + lane) * ld + rb] * w[rb];
    }
  }
  SDM_LPHASE(8);
  __syncthreads();
  if (wave == 0) {
    SDM_SETPRIO(3);
    double cacc = 0.0;                                             // what the rows of block p+1 give to panel p
    for (int p = npan - 1; p >= 0; p--) {
      const int k0 = p * SNB, kb = min(SNB, ns - k0);
      const double *dF = dots3 + (p & 1) * SNB;
      // lanes beyond a partial last block carry the (final) values of the rows below it: passive in the solve (their
      // coefficients are 0), but panel p-1 gets their contribution through the same chain
      double yi = lane < kb ? w[k0 + lane] - dF[lane] - cacc : (k0 + lane < ms ? w[k0 + lane] : 0.0);
      double cnext = 0.0;
      if (p > 0) yi = trsv_bw_fused(Sd2 + (p & 1) * SNB * SNB, Sb2 + (p & 1) * SNB * SBP, yi, cnext, lane);
      else yi = trsv_bw_block(Sd2 + (p & 1) * SNB * SNB, yi, lane);
      cacc = cnext;
      if (lane < kb) w[k0 + lane] = yi;
      SDM_LPHASE(10);
      __syncthreads();
      SDM_LPHASE(11);
    }
  } else {
    for (int p = npan - 1; p >= 0; p--) {
      const int k0 = p * SNB;
      if (p > 0) {
        // what panel p-1 (full) can already know: its rows beyond block p are final.  Wavefronts 1..13 five columns
        // each; wavefronts 14..15 stage the blocks of the next step: diagonal block p-1 (transposed copy DT) and the
        // block left of it, transposed on the way into LDS.
        const int kq = k0 - SNB, ra = k0 + SNB;
        if (wave <= BW_FARW) {
          // this wavefront's share of the block left of diagonal block p-1 (rows kq.., columns kq-64..), transposed
          // on the way into LDS: loads first, stores after the dots
          constexpr int NFT = 64 * BW_FARW, PT = (SNB * SNB / 2 + NFT - 1) / NFT;
          const int t = tid - 64;
          sdm_double2 tv[PT];
          if (p > 1) {
#pragma unroll
            for (int q = 0; q < PT; q++) {
              const int e = min(t + q * NFT, SNB * SNB / 2 - 1), ip = e & 31, c = e >> 5;   // rows kq+2ip, kq+2ip+1 of column kq-64+c
              tv[q] = ((const sdm_double2 *)(Fs + (int64_t)(kq - SNB + c) * ld + kq))[ip];
            }
          }
          bw_far_five(Fs, ld, kq, 5 * (wave - 1), ra, ms, w, dots3 + ((p - 1) & 1) * SNB, lane);
          if (p > 1) {
            double *Tn = Sb2 + ((p - 1) & 1) * SNB * SBP;
#pragma unroll
            for (int q = 0; q < PT; q++) {
              const int e = t + q * NFT, ip = e & 31, c = e >> 5;
              if (e < SNB * SNB / 2) { Tn[(2 * ip) * SBP + c] = tv[q].x; Tn[(2 * ip + 1) * SBP + c] = tv[q].y; }
            }
          }
        } else {
          constexpr int NST = SOLVE_THREADS - 64 * (1 + BW_FARW), PER = SNB * SNB / 2 / NST;
          const sdm_double2 *Dp = (const sdm_double2 *)(Ds + (int64_t)(p - 1) * SNB * SNB);
          double *Sn = Sd2 + ((p - 1) & 1) * SNB * SNB;
          const int t = tid - 64 * (1 + BW_FARW);
          sdm_double2 sv[PER];
#pragma unroll
          for (int q = 0; q < PER; q++) sv[q] = Dp[t + q * NST];
#pragma unroll
          for (int q = 0; q < PER; q++) {
            const int idx = 2 * (t + q * NST), i = idx & 63, c = idx >> 6;       // Sn[c*64 + i] = L(kq+c, kq+i) for i < c
            Sn[idx] = c > i ? sv[q].x : 0.0;
            Sn[idx + 1] = c > i + 1 ? sv[q].y : 0.0;
          }
        }
      }
      __syncthreads();
    }
  }
  SDM_LPHASE_END();
}

__device__ __forceinline__ void front_fw(const double *Fs, int ns, int ms, int ld, double *w, double *wb2, double *Sd2) {
  if (blockDim.x == SOLVE_THREADS) front_fw_pipe(Fs, ns, ms, ld, w, wb2, Sd2);
  else front_fw_small(Fs, ns, ms, ld, w, wb2, Sd2);
}
__device__ __forceinline__ void front_bw(const double *Fs, const double *Ds, int ns, int ms, int ld, double *w, double *dots3, double *Sd2) {
  if (blockDim.x == SOLVE_THREADS) front_bw_pipe(Fs, Ds, ns, ms, ld, w, dots3, Sd2);
  else front_bw_small(Fs, Ds, ns, ms, ld, w, dots3, Sd2);
}

// The sweep kernels keep the front-local vector w either in LDS or (fronts beyond SOLVE_LDS_MAX rows) in HBM.  Their
// bodies are instantiated once per case, so that every access to w is a plain LDS or a plain global instruction: a
// pointer that may be either compiles to FLAT accesses, which queue up behind the streaming loads of the other
// wavefronts -- measured 2 us per in-block solve of wavefront 0 instead of 1.2.
__device__ __forceinline__ void fw_level_body(const double *F, const FrontTab &tab, int s, double *wvec, double *y, double *w, bool copy_up,
                                              double *wb, double *Sd, const double *src, const int *perm) {
  const int ns = tab.ns[s], ms = tab.ms[s], first = tab.first[s];
  double *wg = wvec + tab.woff[s];
  const int tid = threadIdx.x, bs = blockDim.x;
  // src != null: the right-hand side is gathered through perm on the way in (fwblkslv.c:298-303) instead of by a
  // separate launch
  for (int i = tid; i < ms; i += bs) w[i] = i < ns ? (src ? src[perm[first + i]] : y[first + i]) : 0.0;
  __syncthreads();
  for (int ci = tab.childptr[s]; ci < tab.childptr[s + 1]; ci++) {   // children's update vectors, fixed order
    const int c = tab.childlist[ci];
    const int nc = tab.ns[c], mu = tab.ms[c] - nc;
    const int *rel = tab.relidx + tab.roff[c];
    const double *wc = wvec + tab.woff[c] + nc;
    for (int i = tid; i < mu; i += bs) w[rel[i]] += wc[i];
    __syncthreads();
  }
  front_fw(F + tab.foff[s], ns, ms, tab.ld[s], w, wb, Sd);
  for (int i = tid; i < ns; i += bs) y[first + i] = w[i];
  if (copy_up) for (int i = ns + tid; i < ms; i += bs) wg[i] = w[i];      // update vector for the parent
}
__global__ void __launch_bounds__(SOLVE_THREADS)
k_fw_level(const double *F, FrontTab tab, const int *list, double *wvec, double *y, int use_lds, const double *src, const int *perm) {
  SDM_DYN_SMEM(smem);
  __shared__ double wb[3 * SNB];
  double *Sd = (double *)smem;                                    // staged diagonal / sub-diagonal blocks
  const int s = list[blockIdx.x];
  // use_lds = offset of w behind the staged blocks, 0 = w in HBM
  if (use_lds) fw_level_body(F, tab, s, wvec, y, (double *)smem + use_lds, true, wb, Sd, src, perm);
  else fw_level_body(F, tab, s, wvec, y, wvec + tab.woff[s], false, wb, Sd, src, perm);
}

__device__ __forceinline__ void bw_level_body(const double *F, const double *DT, const FrontTab &tab, int s, double *y, double *w,
                                              double *dots, double *Sd, const double *dscale, double *yout, const int *perm) {
  const int ns = tab.ns[s], ms = tab.ms[s], first = tab.first[s];
  const int *rows = tab.lindx + tab.xl[s];
  const int tid = threadIdx.x, bs = blockDim.x;
  // rows below the supernode belong to ancestors, already final (bwblkslv.c:104-105 gathers them once); dscale != null:
  // the ./d between the sweeps (wrapPcg.m:57) is applied to the supernode's own entries on the way in
  for (int i = tid; i < ms; i += bs) w[i] = i < ns ? (dscale ? y[first + i] / dscale[first + i] : y[first + i]) : y[rows[i]];
  __syncthreads();
  front_bw(F + tab.foff[s], DT + tab.toff[s], ns, ms, tab.ld[s], w, dots, Sd);
  for (int i = tid; i < ns; i += bs) {
    y[first + i] = w[i];                                           // descendants read it from here
    if (yout) yout[perm[first + i]] = w[i];                        // y(perm) = ... (bwblkslv.c:272-278) without a scatter launch
  }
}
__global__ void __launch_bounds__(SOLVE_THREADS)
k_bw_level(const double *F, const double *DT, FrontTab tab, const int *list, double *wvec, double *y, int use_lds,
           const double *dscale, double *yout, const int *perm) {
  SDM_DYN_SMEM(smem);
  __shared__ double dots[3 * SNB];
  double *Sd = (double *)smem;                                    // staged diagonal / sub-diagonal blocks
  const int s = list[blockIdx.x];
  if (use_lds) bw_level_body(F, DT, tab, s, y, (double *)smem + use_lds, dots, Sd, dscale, yout, perm);
  else bw_level_body(F, DT, tab, s, y, wvec + tab.woff[s], dots, Sd, dscale, yout, perm);
}

// The whole  y(perm) = L' \ ((L \ rhs(perm)) ./ d)  of wrapPcg.m:56-59 in ONE launch when the factor is a single
// front (the dense shortcut of symbchol.m:75-77 -- every shipped example): gather, forward sweep, diagonal
// scaling, backward sweep and scatter without leaving the CU.
__device__ __forceinline__ void ldl_single_body(const double *F, const double *DT, int m, const int *perm, const double *dsolve,
                                                const double *rhs, double *yout, double *w, int mode, double *wb, double *Sd) {
  const int tid = threadIdx.x, bs = blockDim.x;
  // mode bits: 1 forward sweep, 2 divide by d, 4 backward sweep; rhs is permuted on the way in iff forward,
  // the result on the way out iff backward (fwblkslv.c:298-303, bwblkslv.c:272-278)
  for (int i = tid; i < m; i += bs) w[i] = (mode & 1) ? rhs[perm[i]] : rhs[i];
  __syncthreads();
  if (mode & 1) front_fw(F, m, m, m + (m & 1), w, wb, Sd);
  if (mode & 2) { for (int i = tid; i < m; i += bs) w[i] /= dsolve[i]; __syncthreads(); }
  if (mode & 4) front_bw(F, DT, m, m, m + (m & 1), w, wb, Sd);
  for (int i = tid; i < m; i += bs) { if (mode & 4) yout[perm[i]] = w[i]; else yout[i] = w[i]; }
}
__global__ void __launch_bounds__(SOLVE_THREADS)
k_ldl_single(const double *F, const double *DT, int m, const int *perm, const double *dsolve, const double *rhs,
             double *yout, double *wglob, int use_lds, int mode) {
  SDM_DYN_SMEM(smem);
  __shared__ double wb[3 * SNB];
  double *Sd = (double *)smem;                                    // staged diagonal / sub-diagonal blocks
  if (use_lds) ldl_single_body(F, DT, m, perm, dsolve, rhs, yout, (double *)smem + use_lds, mode, wb, Sd);
  else ldl_single_body(F, DT, m, perm, dsolve, rhs, yout, wglob, mode, wb, Sd);
}

// ---- big single fronts (m >= BIG_FRONT): one CU cannot stream the factor fast enough (~100 GB/s) and a launch per
// 64-column panel is launch bound (>= 4 us per dependent launch here), so the sweeps are cut into SUPER-panels of
// BIGW = 256 columns, one launch each; the launches of a sweep are stream-ordered, there is no inter-workgroup
// synchronisation inside a launch.
// Forward launch P (P = -1 .. nsb-2): x_P (super-block P of the solution) is final.  Workgroup b owns 256 rows:
// b = 0 the rows of super-block P+1, b >= 1 the rows (P+2)*BIGW + (b-1)*256 ...  It applies the 256 columns of
// super-panel P to its rows (four 64-column panels through the same streaming code as front_fw); workgroup 0 then
// solves the diagonal super-block P+1 with front_fw on that sub-front, so that x_{P+1} is final for the next launch.
__global__ void __launch_bounds__(SOLVE_THREADS)
k_big_fw(const double *Fs, int m, int ld, int P, double *w) {
  SDM_DYN_SMEM(smem);
  __shared__ double wl[BIGW], xb[BIGW], wb[3 * SNB];
  double *Sd = (double *)smem;                                    // staged blocks (workgroup 0)
  const int tid = threadIdx.x, bs = blockDim.x;
  const int rbeg = (P + 1) * BIGW + (blockIdx.x == 0 ? 0 : BIGW + ((int)blockIdx.x - 1) * 256);
  const int rend = min(m, rbeg + 256);
  for (int i = tid; i < 256; i += bs) wl[i] = rbeg + i < rend ? w[rbeg + i] : 0.0;
  if (P >= 0)
    for (int i = tid; i < BIGW; i += bs) xb[i] = w[P * BIGW + i];
  __syncthreads();
  if (P >= 0) {
    const int g = (tid >> 4) & 3, t0 = (tid >> 6) * 16 + (tid & 15), tstep = bs >> 2;
    const int npair = (rend - rbeg) >> 1, lim = (npair + 15) & ~15;
    double *wadj = wl - rbeg;                                    // fw_consume indexes by front row
    for (int sub = 0; sub < BIGW / SNB; sub++) {
      const int k0 = P * BIGW + sub * SNB;
      for (int t = t0; t < lim; t += tstep) {
        sdm_double2 v[16];
        fw_issue(v, Fs, ld, k0, rbeg, npair, t, g);
        fw_consume(v, xb + sub * SNB, wadj, rbeg, npair, t, g);
      }
      if (tid == bs - 1 && ((rend - rbeg) & 1)) {                // unpaired last row of the front
        const int r = rend - 1;
        double acc = 0.0;
        for (int c = 0; c < SNB; c++) acc += Fs[(int64_t)(k0 + c) * ld + r] * xb[sub * SNB + c];
        wl[r - rbeg] -= acc;
      }
      __syncthreads();
    }
  }
  if (blockIdx.x == 0) {
    const int ns = rend - rbeg;                                  // diagonal super-block P+1 as a front of its own
    front_fw(Fs + (int64_t)rbeg * ld + rbeg, ns, ns, ld, wl, wb, Sd);
  }
  for (int i = tid; i < rend - rbeg; i += bs) w[rbeg + i] = wl[i];
}
// Backward launch P (P = nsb .. 1): x_P final.  Workgroup b owns the 256 columns of super-block q = P-1-b: it
// applies the rows of super-block P, y_q -= L(P,q)' x_P, and workgroup 0 then solves the transposed diagonal
// super-block P-1 with front_bw.  Launch nsb only solves the last super-block.
__global__ void __launch_bounds__(SOLVE_THREADS)
k_big_bw(const double *Fs, const double *DT, int m, int ld, int P, int nsb, double *w) {
  SDM_DYN_SMEM(smem);
  __shared__ double yl[BIGW], xb[BIGW], dots[BIGW];
  double *Sd = (double *)smem;                                    // staged blocks (workgroup 0)
  const int tid = threadIdx.x, bs = blockDim.x;
  const int wave = tid >> 6, lane = tid & 63, nw = bs >> 6;
  const int q = P - 1 - (int)blockIdx.x, cbeg = q * BIGW, ncol = min(BIGW, m - cbeg);
  for (int i = tid; i < BIGW; i += bs) yl[i] = i < ncol ? w[cbeg + i] : 0.0;
  if (P < nsb) {
    const int pbeg = P * BIGW, np_ = min(BIGW, m - pbeg);
    for (int i = tid; i < BIGW; i += bs) xb[i] = i < np_ ? w[pbeg + i] : 0.0;
    __syncthreads();
    const int npair = np_ >> 1;
    const double *xadj = xb - pbeg;                              // bw_consume indexes by front row
    for (int cb0 = wave * 4; cb0 < ncol; cb0 += nw * 4) {
      double acc[4] = {0.0, 0.0, 0.0, 0.0};
      for (int tb = lane; tb < npair; tb += 256) {
        sdm_double2 v[16];
        bw_issue(v, Fs, ld, cbeg, ncol, cb0, pbeg, npair, tb);
        bw_consume(v, acc, xadj, pbeg, npair, tb);
      }
      if (lane == 0 && (np_ & 1))
#pragma unroll
        for (int u = 0; u < 4; u++) acc[u] += Fs[(int64_t)(cbeg + min(cb0 + u, ncol - 1)) * ld + pbeg + np_ - 1] * xb[np_ - 1];
      {
        const bool hi = lane >= 32;
        const double s0 = hi ? acc[0] : acc[2], s1 = hi ? acc[1] : acc[3];
        double k0v = (hi ? acc[2] : acc[0]) + __shfl_xor(s0, 32);
        double k1v = (hi ? acc[3] : acc[1]) + __shfl_xor(s1, 32);
        const bool od = (lane >> 4) & 1;
        double a = (od ? k1v : k0v) + __shfl_xor(od ? k0v : k1v, 16);
        a += __shfl_xor(a, 8); a += __shfl_xor(a, 4); a += __shfl_xor(a, 2); a += __shfl_xor(a, 1);
        const int u = (hi ? 2 : 0) + (od ? 1 : 0);
        if ((lane & 15) == 0 && cb0 + u < ncol) dots[cb0 + u] = a;
      }
    }
    __syncthreads();
    for (int i = tid; i < ncol; i += bs) yl[i] -= dots[i];
  }
  __syncthreads();
  if (blockIdx.x == 0)
    front_bw(Fs + (int64_t)cbeg * ld + cbeg, DT + (int64_t)(cbeg / SNB) * SNB * SNB, ncol, ncol, ld, yl, dots, Sd);
  __syncthreads();
  for (int i = tid; i < ncol; i += bs) w[cbeg + i] = yl[i];
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
  t.first = C.d_first.p; t.ns = C.d_ns.p; t.ms = C.d_ms.p; t.ld = C.d_ld.p;
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
  
#ifndef SDM_EMU
  SDM_HIP_CHECK(hipFuncSetAttribute((const void *)k_ldl_panel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PANEL_LDS_RIDE));
#endif
  SDM_HIP_CHECK(hipMemsetAsync(C.fronts.p, 0, (size_t)C.fsize * sizeof(double), st));
  SDM_KLAUNCH(P, k_assemble, dim3(grid1d(C.nnzL, 256)), dim3(256), 0, C.fronts.p, P->ada_val.p, C.d_asm_src.p,
             C.d_asm_dst.p, (int64_t)C.nnzL);
  SDM_HIP_CHECK(hipMemsetAsync(C.ub.p, 0, 3 * sizeof(double), st));
  SDM_KLAUNCH(P, k_prep_pivots, dim3(grid1d(m, 256, 64)), dim3(256), 0, m, P->ada_val.p, C.d_asm_src.p, C.d_Ljc.p, C.d_perm.p,
             P->absd.p, use_absd, canceltol, maxu, abstol, C.lb.p, C.ub.p, C.pivstat.p, C.pivval.p, (int)C.nsuper, C.upd_cnt.p, C.diag_cnt.p);
  for (int l = 0; l < C.nlevels; l++) {
    const int *list = C.d_levlist.p + C.levptr[l];
    const int nfr = C.levptr[l + 1] - C.levptr[l];
    if (l > 0) SDM_KLAUNCH(P, k_extend_add, dim3(C.lev_T[l], nfr), dim3(256), 0, C.fronts.p, tab, list);
    for (int li = C.lev_first_launch[l]; li < C.lev_first_launch[l + 1]; li++) {
      const LevelLaunch &L = C.launches[li];
      // ONE launch per panel: diagonal block (+ tile 0 of the previous panel's update), the row solves and the rest of
      // the previous update (see k_ldl_panel).  The emulator runs it in two phases (workgroups are sequential there).
#ifdef SDM_EMU
      for (int phase = 1; phase <= 2; phase++)
#else
      const int phase = 0;
#endif
        SDM_KLAUNCH(P, k_ldl_panel, dim3(1 + L.ride_wgs, L.nactive), dim3(LDL_THREADS), PANEL_LDS_RIDE, C.fronts.p, C.frontsT.p, tab, list,
                    L.panel, C.d.p, C.lb.p, C.ub.p, C.pivstat.p, C.pivval.p, C.colbuf.p, P->ada_val.p, C.d_asm_src.p,
                    C.d_Ljc.p, m, C.upd_cnt.p, C.diag_cnt.p, 1, phase, C.tmo.dev());
      if (L.lasttiles > 0)                                           // supernodes that end with this panel and have rows beyond
        SDM_KLAUNCH(P, k_ldl_update, dim3(L.lasttiles, L.nactive), dim3(256), 0, C.fronts.p, tab, list, L.panel, C.d.p, 1);
    }
  }
  SDM_KLAUNCH(P, k_dsolve, dim3((m + 255) / 256), dim3(256), 0, C.dsolve.p, C.d.p, m);
  SDM_HIP_CHECK(hipGetLastError());
  P->factored = true;
}

// Non-zero when a workgroup gave up waiting for another one inside a launch (never expected; the results of that
// factorisation are then unusable: the plan is marked "not factored", so the solves refuse to run on it).  Reads and
// clears the plan's own flag (pinned host memory the kernels of THIS plan write to); call after a stream synchronise.
int chol_wait_timeouts(sdm_plan *P) {
  CholPlan &C = P->chol;
  if (!C.tmo.host) return 0;
  const int n = *(volatile int *)C.tmo.host;
  if (n) { *(volatile int *)C.tmo.host = 0; P->factored = false; }
  return n;
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

// dynamic LDS of the sweep kernels: two staged diagonal blocks, then the front-local vector when the largest front
// of the plan fits (SOLVE_LDS_MAX doubles), else that vector stays in HBM
constexpr size_t SOLVE_LDS_BLOCKS = (size_t)SOLVE_STAGE_DOUBLES * sizeof(double);
// use = 0: the front-local vector stays in HBM; else its offset (doubles) behind the staged blocks -- plans without a
// front on the look-ahead schedule only ever stage one diagonal block
static void solve_cfg(CholPlan &C, size_t &bytes, int &use) {
  const int stage = C.maxms >= PIPE_MIN_ROWS ? SOLVE_STAGE_DOUBLES : SNB * SNB;
  use = C.maxms <= SOLVE_LDS_MAX ? stage : 0;
  bytes = (size_t)(stage + (use ? C.maxms : 0)) * sizeof(double);
}
static int level_threads(const CholPlan &C, int l) {
  int mx = 0;
  for (int i = C.levptr[l]; i < C.levptr[l + 1]; i++) mx = std::max(mx, C.sn_ms[C.levlist[i]]);
  if (mx >= PIPE_MIN_ROWS) return SOLVE_THREADS;              // full workgroup: look-ahead schedule (front_fw_pipe / front_bw_pipe)
  return std::min(SOLVE_THREADS - 64, std::max(64, (mx + 63) / 64 * 64));
}
void solve_fw(sdm_plan *P, const double *src) {
  CholPlan &C = P->chol;
  FrontTab tab = front_tab(C);
  size_t lds; int use; solve_cfg(C, lds, use);
#ifndef SDM_EMU
  if (lds > 48 * 1024) SDM_HIP_CHECK(hipFuncSetAttribute((const void *)k_fw_level, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
#endif
  for (int l = 0; l < C.nlevels; l++) {
    const int nfr = C.levptr[l + 1] - C.levptr[l];
    SDM_KLAUNCH(P, k_fw_level, dim3(nfr), dim3(level_threads(C, l)), lds, C.fronts.p, tab, C.d_levlist.p + C.levptr[l], C.wvec.p,
                P->ywork.p, use, src, C.d_perm.p);
  }
}
void solve_bw(sdm_plan *P, bool divide, double *yout) {
  CholPlan &C = P->chol;
  FrontTab tab = front_tab(C);
  size_t lds; int use; solve_cfg(C, lds, use);
#ifndef SDM_EMU
  if (lds > 48 * 1024) SDM_HIP_CHECK(hipFuncSetAttribute((const void *)k_bw_level, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
#endif
  for (int l = C.nlevels - 1; l >= 0; l--) {
    const int nfr = C.levptr[l + 1] - C.levptr[l];
    SDM_KLAUNCH(P, k_bw_level, dim3(nfr), dim3(level_threads(C, l)), lds, C.fronts.p, C.frontsT.p, tab, C.d_levlist.p + C.levptr[l],
                C.wvec.p, P->ywork.p, use, divide ? (const double *)C.dsolve.p : (const double *)nullptr, yout, C.d_perm.p);
  }
}
// single-front factor: the complete solve (mode bits 1 fw | 2 ./d | 4 bw) in one launch, rhs -> yout
bool solve_single(sdm_plan *P, const double *rhs, double *yout, int mode) {
  CholPlan &C = P->chol;
  if (C.nsuper != 1) return false;
  if (C.m >= BIG_FRONT) {
    // big front: one launch per 256-column super-panel and sweep (k_big_fw / k_big_bw), w = ywork in HBM
    const int m = (int)C.m, ld = C.sn_ld[0], nsb = (m + BIGW - 1) / BIGW;
#ifndef SDM_EMU
    SDM_HIP_CHECK(hipFuncSetAttribute((const void *)k_big_fw, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SOLVE_LDS_BLOCKS));
    SDM_HIP_CHECK(hipFuncSetAttribute((const void *)k_big_bw, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SOLVE_LDS_BLOCKS));
#endif
    if (mode & 1) vec_gather(P, P->ywork.p, rhs, true);
    else SDM_HIP_CHECK(hipMemcpyAsync(P->ywork.p, rhs, (size_t)m * sizeof(double), hipMemcpyDeviceToDevice, P->stream));
    if (mode & 1)
      for (int p = -1; p <= nsb - 2; p++) {
        const int below = m - (p + 2) * BIGW;                     // rows beyond super-block p+1
        SDM_KLAUNCH(P, k_big_fw, dim3(p < 0 ? 1 : 1 + std::max(0, (below + 255) / 256)), dim3(SOLVE_THREADS), SOLVE_LDS_BLOCKS, C.fronts.p, m, ld,
                    p, P->ywork.p);
      }
    if (mode & 2) vec_divd(P, P->ywork.p);
    if (mode & 4)
      for (int p = nsb; p >= 1; p--)
        SDM_KLAUNCH(P, k_big_bw, dim3(p == nsb ? 1 : p), dim3(SOLVE_THREADS), SOLVE_LDS_BLOCKS, C.fronts.p, C.frontsT.p, m, ld, p, nsb, P->ywork.p);
    if (mode & 4) vec_gather(P, yout, P->ywork.p, false);
    else SDM_HIP_CHECK(hipMemcpyAsync(yout, P->ywork.p, (size_t)m * sizeof(double), hipMemcpyDeviceToDevice, P->stream));
    return true;
  }
  size_t lds; int use; solve_cfg(C, lds, use);
#ifndef SDM_EMU
  if (lds > 48 * 1024) SDM_HIP_CHECK(hipFuncSetAttribute((const void *)k_ldl_single, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
#endif
  SDM_KLAUNCH(P, k_ldl_single, dim3(1), dim3(level_threads(C, 0)), lds, C.fronts.p, C.frontsT.p, (int)C.m, C.d_perm.p, C.dsolve.p,
              rhs, yout, C.wvec.p, use, mode);
  return true;
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

#if defined(SDM_PHASES) && !defined(SDM_EMU)
// tools-only build (python -m sedumi_amd.build --phases): read / reset the in-kernel phase clocks of this file
extern "C" int sdm_debug_phases_chol(unsigned long long *out32, int reset) {
  if (out32 && hipMemcpyFromSymbol(out32, HIP_SYMBOL(sdm_phase_acc), 32 * sizeof(unsigned long long)) != hipSuccess) return 1;
  if (reset) { unsigned long long z[32] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(sdm_phase_acc), z, sizeof(z)) != hipSuccess) return 1; }
  return 0;
}
#endif

// sdm_ada.hip -- forming the normal-equations matrix ADA' on gfx950.
//
// Reference (getada1.c:89-152, getada2.c:74-118, getada3.c:253-361 with
// spscale.c:249-305 sprealdxd): a sequential sweep over constraints in three
// different "sparsest first" orders; for the PSD part D*A_j*D is evaluated
// only on an incrementally growing pattern (Aord.dz) and each routine fills
// one triangle, getada3 finally symmetrising (spmakesym, getada3.c:151-180).
//
// Here the same numbers are produced data-parallel:
//   * LP / Lorentz parts: one workgroup per ADA column, one wavefront per
//     pattern entry, sparse-sparse dot by binary search (no dense scratch).
//   * PSD part, stage 1: one workgroup per (constraint j, PSD block k) task
//     computes z_jk = (D_k sym(A_jk) D_k) restricted to U_k, the union pattern
//     of block k over all constraints (what Aord.dz enumerates incrementally):
//     Y = D*X(:,cols) is staged in LDS, targets are evaluated with the same
//     two-dot formula as spscale.c:283-304.
//   * PSD part, stage 2: one workgroup per ADA column gathers
//     ADA(i,j) += a_i[psd]' z_j  for every pattern entry (getada3.c:333-351),
//     absd fused on the diagonal entry.
// The triangular bookkeeping of the reference (which triangle each routine
// writes) is reproduced for the stand-alone MEX-equivalent calls through an
// inverse-permutation mask; the fused resident path writes the symmetric
// matrix directly.
#include "sdm_plan.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace sdm {

// ============================================================ host analysis
void ada_build(sdm_plan *P, sdm_int N, sdm_int m, const sdm_int *Ajc, const sdm_int *Air, const double *Apr,
               const sdm_int *Ajc_psd, sdm_int lpN, sdm_int lorN, const sdm_int *lorNL, sdm_int sdpN,
               sdm_int rsdpN, const sdm_int *sdpNL, const sdm_int *qblkstart, const sdm_int *psd_blkstart,
               const sdm_int *Qjc, const sdm_int *Qir, const sdm_int *ADAjc, const sdm_int *ADAir) {
  AdaPlan &A = P->ada;
  A.N = N; A.m = m; A.nnzA = Ajc[m]; A.lpN = lpN; A.lorN = lorN; A.sdpN = sdpN; A.rsdpN = rsdpN;
  A.ic_n.release(); A.ufac.release();                                // invcholfac tables belong to the old cone
  if (A.nnzA >= (sdm_int)1 << 31 || N >= (sdm_int)1 << 31) throw std::runtime_error("At too large for 32-bit row indices");
  (void)lorNL;
  A.nlq = sdpN > 0 ? psd_blkstart[0] : N;
  if (lorN > 0 && qblkstart[lorN] != A.nlq && sdpN > 0) throw std::runtime_error("qblkstart / psd_blkstart mismatch");
  // ---- dsqr source codes (getada1.c:106-118): -1 -> dl[r];  -2-k -> -ddet[k];  k>=0 -> ddet[k]
  {
    std::vector<int> code((size_t)A.nlq, -1);
    for (sdm_int r = lpN; r < lpN + lorN && r < A.nlq; r++) code[r] = (int)(-2 - (r - lpN));
    for (sdm_int k = 0; k < lorN; k++)
      for (sdm_int r = qblkstart[k]; r < qblkstart[k + 1] && r < A.nlq; r++) code[r] = (int)k;
    A.dsqr_code.upload(code);
  }
  // ---- PSD blocks
  A.psd_n.assign(sdpNL, sdpNL + sdpN);
  A.psd_start.assign(psd_blkstart, psd_blkstart + sdpN + (sdpN > 0 ? 1 : 0));
  A.psd_udoff.assign(sdpN + 1, 0);
  A.maxn = 0;
  for (sdm_int k = 0; k < sdpN; k++) {
    sdm_int n = sdpNL[k];
    sdm_int len = (k < rsdpN ? 1 : 2) * n * n;
    if (psd_blkstart[k + 1] - psd_blkstart[k] != len) throw std::runtime_error("PSD block size / blkstart mismatch");
    A.psd_udoff[k + 1] = A.psd_udoff[k] + len;
    A.maxn = std::max<int>(A.maxn, (int)n);
  }
  A.lenud = A.psd_udoff[sdpN];
  // ---- per PSD nonzero: block id and position in the union pattern U_k
  std::vector<int> Ablk((size_t)A.nnzA, -1), Aupos((size_t)A.nnzA, 0);
  std::vector<std::vector<int>> U(sdpN);
  for (sdm_int j = 0; j < m && sdpN > 0; j++) {
    sdm_int k = 0;
    for (sdm_int t = Ajc_psd[j]; t < Ajc[j + 1]; t++) {
      sdm_int r = Air[t];
      if (r < A.nlq) throw std::runtime_error("Ajc_psd points into the LP/Lorentz part");
      while (k < sdpN && r >= psd_blkstart[k + 1]) k++;
      if (k >= sdpN) throw std::runtime_error("At row index beyond the PSD blocks");
      Ablk[t] = (int)k;
      U[k].push_back((int)(r - psd_blkstart[k]));
    }
  }
  std::vector<int64_t> uoff(sdpN + 1, 0);
  for (sdm_int k = 0; k < sdpN; k++) {
    std::sort(U[k].begin(), U[k].end());
    U[k].erase(std::unique(U[k].begin(), U[k].end()), U[k].end());
    uoff[k + 1] = uoff[k] + (int64_t)U[k].size();
  }
  std::vector<int> upos_all((size_t)uoff[sdpN]);
  for (sdm_int k = 0; k < sdpN; k++) std::copy(U[k].begin(), U[k].end(), upos_all.begin() + uoff[k]);
  sdm_int psdnnz = 0;
  for (sdm_int j = 0; j < m && sdpN > 0; j++)
    for (sdm_int t = Ajc_psd[j]; t < Ajc[j + 1]; t++) {
      int k = Ablk[t];
      int q = (int)(Air[t] - psd_blkstart[k]);
      Aupos[t] = (int)(std::lower_bound(U[k].begin(), U[k].end(), q) - U[k].begin());
      psdnnz++;
    }
  A.thread_per_row = (m > 0 && psdnnz / (double)m < 48.0);     // short rows: one pattern entry per work-item, else per wavefront
  A.nnz_lq = A.nnzA - psdnnz;                                      // LP + Lorentz nonzeros of At
  A.lq_maxcol = 0;
  for (sdm_int j = 0; j < m; j++) A.lq_maxcol = std::max<int64_t>(A.lq_maxcol, (sdpN > 0 ? Ajc_psd[j] : Ajc[j + 1]) - Ajc[j]);
  // ---- stage-1 tasks and slots
  std::vector<int> t_col, t_blk, t_n, t_nslot, t_ulen, t_herm, s_col;
  std::vector<int64_t> t_slotptr, t_udoff, t_uoff, t_zoff, s_nzptr, c_taskptr(m + 1, 0);
  int64_t zlen = 0;
  for (sdm_int j = 0; j < m; j++) {
    sdm_int t = sdpN > 0 ? Ajc_psd[j] : Ajc[j + 1];
    while (t < Ajc[j + 1]) {
      int k = Ablk[t];
      sdm_int te = t;
      while (te < Ajc[j + 1] && Ablk[te] == k) te++;
      const sdm_int n = A.psd_n[k];
      const bool herm = k >= rsdpN;
      t_col.push_back((int)j); t_blk.push_back(k); t_n.push_back((int)n); t_herm.push_back(herm ? 1 : 0);
      t_slotptr.push_back((int64_t)s_col.size());
      t_udoff.push_back(A.psd_udoff[k]); t_uoff.push_back(uoff[k]); t_ulen.push_back((int)U[k].size());
      t_zoff.push_back(zlen); zlen += (int64_t)U[k].size();
      // slots: distinct columns of X_jk (real part first, then imaginary part for Hermitian blocks)
      int nslot = 0; sdm_int prevcol = -1; int prevpart = -1;
      for (sdm_int u = t; u < te; u++) {
        sdm_int q = Air[u] - psd_blkstart[k];
        int part = q >= n * n ? 1 : 0;
        sdm_int col = (q - part * n * n) / n;
        if (col != prevcol || part != prevpart) {
          s_col.push_back((int)(col + part * n)); s_nzptr.push_back((int64_t)u);
          nslot++; prevcol = col; prevpart = part;
        }
      }
      t_nslot.push_back(nslot);
      t = te;
    }
    c_taskptr[j + 1] = (int64_t)t_col.size();
  }
  s_nzptr.push_back(A.nnzA);   // sentinel (only used through per-task end pointers)
  { // dispatch order of the stage-1 tasks: decreasing cost (nonzeros x order + slots x order^2), so that the few heavy
    // constraints do not form the tail of the launch
    std::vector<int> order(t_col.size());
    std::vector<double> cost(t_col.size());
    for (size_t t = 0; t < t_col.size(); t++) {
      order[t] = (int)t;
      const double nz = (double)((t + 1 < t_slotptr.size() ? s_nzptr[t_slotptr[t + 1]] : A.nnzA) - s_nzptr[t_slotptr[t]]);
      cost[t] = nz * t_n[t] + (double)t_nslot[t] * t_n[t] * t_n[t];
    }
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cost[a] > cost[b]; });
    A.t_order.upload(order);
    // the generic kernel's dispatch order when there are many blocks (64 x 200): the hardware deals consecutive workgroups to the 8 XCDs
    // in turn, and every task re-reads rows of its block's D_k through its XCD's L2 -- with the constraints in natural order every D_k
    // was fetched by all eight L2s (246 MB of HBM traffic per launch against the 20 MB of the D_k).  So: block k's tasks go to XCD k % 8,
    // workgroup 8 s + x takes the s-th task of XCD x's list (heaviest first inside a list); lists that run out leave their turns to the rest.
    A.t_order_xcd.release();
    if (sdpN >= 16 && !t_col.empty()) {
      constexpr int NX = 8;
      std::vector<std::vector<int>> lst(NX);
      for (int t : order) lst[t_blk[t] % NX].push_back(t);
      std::vector<int> ox; ox.reserve(order.size());
      std::vector<size_t> pos(NX, 0);
      while (ox.size() < order.size())
        for (int x = 0; x < NX; x++) if (pos[x] < lst[x].size()) ox.push_back(lst[x][pos[x]++]);
      A.t_order_xcd.upload(ox);
    } }
  A.s1_maxnz = 0;
  for (size_t t = 0; t < t_col.size(); t++)
    A.s1_maxnz = std::max<int64_t>(A.s1_maxnz, (t + 1 < t_slotptr.size() ? s_nzptr[t_slotptr[t + 1]] : A.nnzA) - s_nzptr[t_slotptr[t]]);
  A.ntask = (sdm_int)t_col.size(); A.zlen = zlen;
  A.one_task_per_col = true; A.s1_maxulen = 0;
  for (sdm_int j = 0; j < m; j++) if (c_taskptr[j + 1] - c_taskptr[j] > 1) A.one_task_per_col = false;
  for (size_t t = 0; t < t_ulen.size(); t++) A.s1_maxulen = std::max(A.s1_maxulen, t_ulen[t]);
  A.h_taskptr = c_taskptr; A.col0 = 0; A.col1 = m;
  { std::vector<int64_t> czl(m + 1, 0);                       // length of z_j (all tasks of constraint j)
    A.zmaxj = 0;
    for (sdm_int j = 0; j < m; j++) {
      for (int64_t t = c_taskptr[j]; t < c_taskptr[j + 1]; t++) czl[j] += t_ulen[t];
      A.zmaxj = std::max<int64_t>(A.zmaxj, czl[j]);
    }
    A.c_zlen.upload(czl); }
  // per task: end of its last slot = start of next task's first nonzero; store explicit end pointers in s_nzptr
  // by giving every slot an (begin) and using the next slot's begin inside a task, and the task end via t_end:
  std::vector<int64_t> t_end(A.ntask);
  { sdm_int ti = 0;
    for (sdm_int j = 0; j < m; j++) {
      sdm_int t = sdpN > 0 ? Ajc_psd[j] : Ajc[j + 1];
      while (t < Ajc[j + 1]) { int k = Ablk[t]; sdm_int te = t; while (te < Ajc[j + 1] && Ablk[te] == k) te++; t_end[ti++] = te; t = te; }
    } }
  // ---- stage-2 fast path (dense-ish ADA patterns): the PSD nonzeros of At re-packed for one-row-per-lane sweeps.
  // Rows (constraints) are sorted by their number of PSD nonzeros and cut into groups of 64; a group stores its
  // nonzeros interleaved (entry t of all 64 rows contiguous) and padded to the longest row of the group, so that
  // every load of the sweep is one coalesced 512-byte line and no cross-lane reduction is needed.  All wavefronts of
  // a workgroup share every group (interleaved slices of the entry range).
  {
    A.ell_ok = false;
    // z_j is staged in LDS at FULL length (all blocks, zeros where constraint j has no nonzero): an entry of the ELL
    // copy then carries its final position uoff[k] + upos and the sweep needs one LDS gather per entry and column
    const int64_t zmax = uoff[sdpN];
    A.zmax = zmax;
    const double dens = m > 0 ? (double)ADAjc[m] / ((double)m * (double)m) : 0.0;
    const size_t lds = (size_t)zmax * sizeof(double);
    if (sdpN > 0 && psdnnz > 0 && dens >= 0.2 && lds <= 96 * 1024 && !A.thread_per_row) {   // very short rows: one pattern entry per work-item instead
      std::vector<int> order(m);
      for (sdm_int j = 0; j < m; j++) order[j] = (int)j;
      std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return Ajc[a + 1] - Ajc_psd[a] > Ajc[b + 1] - Ajc_psd[b]; });
      const int ng = (int)((m + 63) / 64);
      std::vector<int> grow((size_t)ng * 64, -1), glen(ng);
      std::vector<int64_t> goff(ng + 1, 0);
      for (int g = 0; g < ng; g++) {
        int len = 0;
        for (int l = 0; l < 64 && g * 64 + l < m; l++) { const int i = order[g * 64 + l]; grow[g * 64 + l] = i; len = std::max<int>(len, (int)(Ajc[i + 1] - Ajc_psd[i])); }
        glen[g] = len; goff[g + 1] = goff[g] + len;
      }
      std::vector<double> gval((size_t)goff[ng] * 64, 0.0);
      std::vector<int> gbu((size_t)goff[ng] * 64, 0);           // position in the full-length z vector (padding: 0 with value 0)
      for (int g = 0; g < ng; g++)
        for (int l = 0; l < 64; l++) {
          const int i = grow[g * 64 + l];
          if (i < 0) continue;
          for (sdm_int t = Ajc_psd[i]; t < Ajc[i + 1]; t++) {
            const size_t pos = (size_t)(goff[g] + (t - Ajc_psd[i])) * 64 + l;
            gval[pos] = Apr[t]; gbu[pos] = (int)(uoff[Ablk[t]] + Aupos[t]);
          }
        }
      A.ell_ng = ng;
      A.g_row.upload(grow); A.g_len.upload(glen); A.g_off.upload(goff); A.g_val.upload(gval); A.g_bu.upload(gbu);
      { std::vector<int> pos((size_t)m); for (sdm_int p = 0; p < m; p++) pos[(size_t)order[p]] = (int)p; A.ell_pos.upload(pos); }
      A.ell_order.upload(order);
      A.ell_full = (ADAjc[m] == (sdm_int)m * m);                      // every column of the pattern full: (j, i) sits at ADAjc[i] + j
      A.d_uoff.upload(uoff);
      { std::vector<int> zp((size_t)A.nnzA, 0);                       // position of a PSD nonzero's entry in the full-length z vector
        for (sdm_int i = 0; i < m; i++) for (sdm_int t = Ajc_psd[i]; t < Ajc[i + 1]; t++) zp[(size_t)t] = (int)(uoff[Ablk[t]] + Aupos[t]);
        A.d_Azpos.upload(zp);
        std::vector<int> zd(t_blk.size());                            // and of a task's block
        for (size_t t = 0; t < t_blk.size(); t++) zd[t] = (int)uoff[t_blk[t]];
        A.t_zdst.upload(zd);
        // one record per column for the prologue of k_psd_stage2_ell, indexed by column (cdc*) and by ELL position (cdp*):
        //   cd64[8 x + ..] = first task, first PSD nonzero, end of the column, source offset in zbuf of its first four tasks
        //   cd32[16 x + ..] = the column, number of tasks, destination offset in z of the first four, their lengths
        for (int byp = 0; byp < 2; byp++) {
          std::vector<long long> c64((size_t)std::max<sdm_int>(m, 1) * 8, 0);
          std::vector<int> c32((size_t)std::max<sdm_int>(m, 1) * 16, 0);
          for (sdm_int x = 0; x < m; x++) {
            const sdm_int j = byp ? order[(size_t)x] : x;
            const int64_t tb = c_taskptr[j], te = c_taskptr[j + 1];
            c64[8 * x] = tb; c64[8 * x + 1] = Ajc_psd[j]; c64[8 * x + 2] = Ajc[j + 1];
            c32[16 * x] = (int)j; c32[16 * x + 1] = (int)(te - tb);
            for (int sg = 0; sg < 4 && tb + sg < te; sg++) {
              c64[8 * x + 3 + sg] = t_zoff[(size_t)(tb + sg)]; c32[16 * x + 2 + sg] = zd[(size_t)(tb + sg)]; c32[16 * x + 6 + sg] = t_ulen[(size_t)(tb + sg)];
            }
          }
          if (byp) { A.cdp64.upload(c64); A.cdp32.upload(c32); } else { A.cdc64.upload(c64); A.cdc32.upload(c32); }
        } }
      A.ell_ok = true;
    }
  }
  // ---- transposed-entry map of the ADA pattern
  std::vector<int> adaT((size_t)ADAjc[m], -1);
  { std::vector<sdm_int> nxt(ADAjc, ADAjc + m);
    // for entry e=(i,j): find (j,i) by binary search in column i
    for (sdm_int j = 0; j < m; j++)
      for (sdm_int e = ADAjc[j]; e < ADAjc[j + 1]; e++) {
        sdm_int i = ADAir[e];
        const sdm_int *b = ADAir + ADAjc[i], *en = ADAir + ADAjc[i + 1];
        const sdm_int *f = std::lower_bound(b, en, j);
        if (f != en && *f == j) adaT[e] = (int)(f - ADAir);
      } }
  // ---- upload
  { std::vector<int64_t> v(Ajc, Ajc + m + 1); A.d_Ajc.upload(v); }
  { std::vector<int64_t> v(Ajc_psd, Ajc_psd + m); A.d_Ajc_psd.upload(v); }
  { std::vector<int> v((size_t)A.nnzA); for (sdm_int t = 0; t < A.nnzA; t++) v[t] = (int)Air[t]; A.d_Air.upload(v); }
  A.d_Apr.upload(Apr, (size_t)A.nnzA);
  A.d_Ablk.upload(Ablk); A.d_Aupos.upload(Aupos);
  A.nnzQ = lorN > 0 ? Qjc[m] : 0;
  A.q_maxcol = 0;
  for (sdm_int j = 0; j < m && lorN > 0; j++) A.q_maxcol = std::max<int64_t>(A.q_maxcol, Qjc[j + 1] - Qjc[j]);
  { std::vector<int64_t> v(m + 1, 0); if (lorN > 0) v.assign(Qjc, Qjc + m + 1); A.d_Qjc.upload(v); }
  { std::vector<int> v((size_t)A.nnzQ); for (sdm_int t = 0; t < A.nnzQ; t++) v[t] = (int)Qir[t]; A.d_Qir.upload(v); }
  { std::vector<int64_t> v(ADAjc, ADAjc + m + 1); A.d_ADAjc.upload(v); }
  { std::vector<int> v((size_t)ADAjc[m]); for (sdm_int t = 0; t < ADAjc[m]; t++) v[t] = (int)ADAir[t]; A.d_ADAir.upload(v); }
  A.d_ADAT.upload(adaT);
  A.u_pos.upload(upos_all);
  { std::vector<int> urc(upos_all.size(), 0);                        // (r << 16) | c of a real block's target (k_psd_stage1_mfma: no division per target)
    for (sdm_int k = 0; k < std::min(sdpN, rsdpN); k++) {
      const int n = (int)A.psd_n[k];
      if (n >= 65536) continue;
      for (size_t u = 0; u < U[k].size(); u++) { const int q = U[k][u], c = q / n, r = q - c * n; urc[(size_t)uoff[k] + u] = (r << 16) | c; }
    }
    A.u_rc.upload(urc); }
  A.t_col.upload(t_col); A.t_blk.upload(t_blk); A.t_n.upload(t_n); A.t_nslot.upload(t_nslot); A.t_ulen.upload(t_ulen);
  A.t_herm.upload(t_herm);
  A.t_slotptr.upload(t_slotptr); A.t_udoff.upload(t_udoff); A.t_uoff.upload(t_uoff); A.t_zoff.upload(t_zoff);
  A.s_col.upload(s_col); A.s_nzptr.upload(s_nzptr); A.c_taskptr.upload(c_taskptr);
  A.t_end.upload(t_end);
  { std::vector<int64_t> v(A.psd_start.begin(), A.psd_start.end()); if (v.empty()) v.push_back(0); A.d_psd_start.upload(v); }
  A.zbuf.alloc((size_t)std::max<int64_t>(zlen, 1));
  A.dsqr.alloc((size_t)std::max<sdm_int>(A.nlq, 1));
  A.dl.alloc((size_t)std::max<sdm_int>(lpN, 1)); A.ddet.alloc((size_t)std::max<sdm_int>(lorN, 1));
  A.qpr.alloc((size_t)std::max<sdm_int>(A.nnzQ, 1)); A.udsqr.alloc((size_t)std::max<sdm_int>(A.lenud, 1));
  { std::vector<int64_t> v(lorN + 1, A.nlq); for (sdm_int k = 0; k <= lorN && lorN > 0; k++) v[k] = qblkstart[k]; A.d_qblk.upload(v); }
  A.q1.alloc((size_t)std::max<sdm_int>(lorN, 1));
  A.q2.alloc((size_t)std::max<sdm_int>(lorN > 0 ? qblkstart[lorN] - qblkstart[0] : 0, 1));
  A.symtmp.alloc((size_t)std::max<sdm_int>(ADAjc[m], 1));
  // ---- dense-column form of the LP / Lorentz part.  A sparse-sparse dot per ADA' entry (k_ada_spdot) is the right
  // tool for sparse columns; when the columns are dense-ish (nb.mat: 66 %) the same sums are a weighted Gram matrix
  // A' diag(dsqr) A, i.e. GEMM-shaped work for the matrix cores.  Static data (At) is expanded once here.
  {
    sdm_int nz = 0;
    for (sdm_int j = 0; j < m; j++) nz += Ajc_psd[j] - Ajc[j];
    const double cells = (double)A.nlq * (double)m;
    A.lq_dense = A.nlq > 0 && m > 1 && nz > 0 && (double)nz >= 0.10 * cells && cells * 8.0 <= 1.0e9;
    A.q_dense = lorN > 0 && A.nnzQ > 0 && (double)A.nnzQ >= 0.10 * (double)lorN * (double)m && (double)lorN * m * 8.0 <= 1.0e9;
    if (A.lq_dense) {
      std::vector<double> D((size_t)A.nlq * (size_t)m, 0.0);
      for (sdm_int j = 0; j < m; j++)
        for (sdm_int t = Ajc[j]; t < Ajc_psd[j]; t++) D[(size_t)j * (size_t)A.nlq + (size_t)Air[t]] = Apr[t];
      A.Alq_d.upload(D);
    } else A.Alq_d.release();
    if (A.q_dense) {
      std::vector<int64_t> dst((size_t)A.nnzQ);
      for (sdm_int j = 0; j < m; j++)
        for (sdm_int t = Qjc[j]; t < Qjc[j + 1]; t++) dst[(size_t)t] = (int64_t)j * lorN + Qir[t];
      A.q_dst.upload(dst);
      A.Q_d.alloc((size_t)lorN * (size_t)m);
      if (A.lq_dense && (double)lorN * (double)m <= 1.6e7) {          // inverse map for the fused form (ada_lq_q)
        std::vector<int> src((size_t)lorN * (size_t)m, -1);
        for (sdm_int t = 0; t < A.nnzQ; t++) src[(size_t)dst[(size_t)t]] = (int)t;
        A.q_src.upload(src);
      } else A.q_src.release();
    } else { A.Q_d.release(); A.q_dst.release(); A.q_src.release(); }
    if (A.lq_dense || A.q_dense) {
      const int nt = (int)((m + TILE - 1) / TILE), T = nt * (nt + 1) / 2;
      const sdm_int rows = std::max(A.lq_dense ? A.nlq : 0, A.q_dense ? lorN : 0);
      A.gram_split = (int)std::max<sdm_int>(1, std::min<sdm_int>((rows + TILE - 1) / TILE, std::max(1, 512 / T)));
      A.gram_part.alloc((size_t)A.gram_split * (size_t)m * (size_t)m);
    } else A.gram_part.release();
  }
  // LDS budget for stage 1: Y chunk of CC slots x n rows
  A.stage1_lds = 96 * 1024;
  { const size_t need = (size_t)(sdpN > rsdpN ? 4 : 2) * (size_t)A.maxn * sizeof(double);     // one slot: Y (+Yi) and D(col,:) (+Im)
    if (need > A.stage1_lds) A.stage1_lds = need; }
  if (A.stage1_lds > 136 * 1024) throw std::runtime_error("PSD block too large for the LDS-staged D*A*D kernel (n > 8700)");
  P->has_ada = true;
}

// ================================================================= kernels
__global__ void k_dsqr(double *dsqr, const int *code, const double *dl, const double *ddet, int nlq) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < nlq) { int c = code[r]; dsqr[r] = c == -1 ? dl[r] : (c >= 0 ? ddet[c] : -ddet[-2 - c]); }
}
__global__ void k_fill(double *x, double v, int64_t n) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; t < n; t += stride) x[t] = v;
}

// sparse weighted dot of two At columns restricted to [beg,end) ranges; one wavefront per ADA entry.
//   val(i,j) = sum_r  M(r,i) * w(r) * M(r,j),   w = dsqr (getada1) or 1 (getada2)
constexpr int SPDOT_CAP = 2048;      // entries of a column of M that k_ada_spdot keeps in LDS (24 KB)
// LPE = lanes per pattern entry: 16 (four entries per wavefront: short columns) or 64
template <int LPE>
__global__ void __launch_bounds__(256)
k_ada_spdot(double *ada, const int64_t *ADAjc, const int *ADAir, const int64_t *Mbeg, const int64_t *Mend,
            const int *Mir, const double *Mpr, const double *wgt, const int *invperm, int accumulate, int jbase, int cap) {
  // Column j of M (its row indices and its values times the weights) goes to LDS once per workgroup when it fits `cap` entries: the
  // binary searches of all the column's pattern entries then walk LDS (~100 clocks a step) instead of being chains of dependent loads
  // from L2 (each entry of a nearly dense ADA' was 8 steps x 4 rounds of them: 4 ms on the LP with m = 2000, 36 us on arch0.mat).
  SDM_DYN_SMEM(smem);
  double *jv = (double *)smem;
  int *jr = (int *)(jv + cap);
  const int j = blockIdx.x + jbase;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const int64_t jb = Mbeg[j], je = Mend[j];
  const int nj = (int)(je - jb);
  const bool inlds = nj <= cap;
  if (inlds)
    for (int t = threadIdx.x; t < nj; t += blockDim.x) { const int r = Mir[jb + t]; jr[t] = r; jv[t] = (wgt ? wgt[r] : 1.0) * Mpr[jb + t]; }
  __syncthreads();
  const int ipj = invperm ? invperm[j] : 0;
  // short columns: four pattern entries per wavefront, 16 lanes each -- the per-entry chain (row index -> column range -> nonzeros) is
  // three dependent round trips however short the column, so it is the number of entries in flight that counts (arch0.mat: 36 -> 12 us);
  // long columns keep the whole wavefront on one entry (nb.mat's DAt.q at the identity scaling: 34 us, 49 with four entries each)
  constexpr int EPW = 64 / LPE;
  const int g = lane / LPE, l16 = lane % LPE;
  const int64_t e1 = ADAjc[j + 1];
  for (int64_t e0 = ADAjc[j] + EPW * wave; e0 < e1; e0 += EPW * nw) {
    const int64_t e = e0 + g;
    const bool on = e < e1;
    const int i = ADAir[on ? e : e1 - 1];
    const bool skip = !on || (invperm && invperm[i] > ipj);
    double acc = 0.0;
    if (je > jb && !skip) {
      for (int64_t t = Mbeg[i] + l16; t < Mend[i]; t += LPE) {
        const int r = Mir[t];
        if (inlds) {
          int lo = 0, hi = nj;                        // first index with jr >= r
          while (lo < hi) { const int mid = (lo + hi) >> 1; if (jr[mid] < r) lo = mid + 1; else hi = mid; }
          if (lo < nj && jr[lo] == r) acc += Mpr[t] * jv[lo];
        } else {
          int64_t lo = jb, hi = je;
          while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (Mir[mid] < r) lo = mid + 1; else hi = mid; }
          if (lo < je && Mir[lo] == r) acc += Mpr[t] * ((wgt ? wgt[r] : 1.0) * Mpr[lo]);
        }
      }
    }
    for (int off = LPE / 2; off > 0; off >>= 1) acc += __shfl_down(acc, off);       // (lane 0 of each group ends with its own group's sum)
    if (l16 == 0 && !skip) { if (accumulate) ada[e] += acc; else ada[e] = acc; }
  }
}

// ---- dense-column form: G = M' diag(w) M for a dense column-major R x m matrix M (rows = LP/Lorentz rows of At, or
// the rows of DAt.q with w = 1).  grid (lower 64x64 tiles, K splits): workgroup (t, s) forms tile t over the rows of
// split s on v_mfma_f64_16x16x4_f64 and writes it to part[s] (m x m, lower tiles); k_gram_scatter adds the splits in
// fixed order and stores into the ADA' pattern (same masks / accumulate semantics as k_ada_spdot).
__global__ void __launch_bounds__(256)
k_gram_tile(const double *M, const double *wgt, int R, int m, int nsplit, double *part, const double *M2 = nullptr, int R2 = 0) {
  // M2 != null: G = M' diag(w) M + M2' M2 in the same launch (LP / Lorentz-det rows, then the rows of DAt.q: ada_lq_q)
  __shared__ double As[TILE][TILE], Bs[TILE][TILE];
  const int t = blockIdx.x, sp = blockIdx.y;
  int I = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while ((I + 1) * (I + 2) / 2 <= t) I++;
  while (I * (I + 1) / 2 > t) I--;
  const int J = t - I * (I + 1) / 2;
  const int nch1 = (R + TILE - 1) / TILE, nch = nch1 + (M2 ? (R2 + TILE - 1) / TILE : 0);
  const int c0 = (int)((int64_t)nch * sp / nsplit), c1 = (int)((int64_t)nch * (sp + 1) / nsplit);
  const double *M1 = M; const int R1 = R;
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
  const int wi = w >> 1, wj = w & 1, lk = l >> 4, ll = l & 15;
  sdm_double4 acc[2][2];
  for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) for (int r = 0; r < 4; r++) acc[a][b][r] = 0.0;
  for (int chg = c0; chg < c1; chg++) {
    const bool second = chg >= nch1;                                 // (uniform) this chunk of 64 rows comes from M2
    const int ch = second ? chg - nch1 : chg;
    M = second ? M2 : M1; R = second ? R2 : R1;
    {
      const int tt = tid & 63, cq = tid >> 6, r = ch * TILE + tt;
      double av[TILE / 4], bv[TILE / 4];
      const double wr = (!second && wgt && r < R) ? wgt[r] : 1.0;
#pragma unroll
      for (int q = 0; q < TILE / 4; q++) {
        const int ci = I * TILE + cq + 4 * q, cj = J * TILE + cq + 4 * q;
        av[q] = M[(int64_t)min(ci, m - 1) * R + min(r, R - 1)];
        bv[q] = M[(int64_t)min(cj, m - 1) * R + min(r, R - 1)];
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < TILE / 4; q++) {
        const int ci = I * TILE + cq + 4 * q, cj = J * TILE + cq + 4 * q;
        As[tt][cq + 4 * q] = (ci < m && r < R) ? av[q] : 0.0;
        Bs[tt][cq + 4 * q] = (cj < m && r < R) ? bv[q] * wr : 0.0;
      }
    }
    __syncthreads();
#pragma unroll 4
    for (int kk = 0; kk < TILE; kk += 4) {
      double ar[2], br[2];
#pragma unroll
      for (int a = 0; a < 2; a++) ar[a] = As[kk + lk][wi * 32 + a * 16 + ll];
#pragma unroll
      for (int b = 0; b < 2; b++) br[b] = Bs[kk + lk][wj * 32 + b * 16 + ll];
#pragma unroll
      for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) acc[a][b] = SDM_MFMA_F64_16x16x4(br[b], ar[a], acc[a][b]);
    }
    __syncthreads();
  }
  double *out = part + (int64_t)sp * m * m;
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int i = I * TILE + wi * 32 + a * 16 + ll, j = J * TILE + wj * 32 + b * 16 + lk + 4 * r;
        if (i < m && j < m) out[(int64_t)j * m + i] = acc[a][b][r];       // G(i,j), tiles I >= J (diagonal tiles in full)
      }
}
__global__ void k_gram_scatter(double *ada, const int64_t *ADAjc, const int *ADAir, const double *part, int nsplit, int m,
                               const int *invperm, int accumulate, int jbase, double *absd = nullptr) {
  // absd != null (no PSD part follows): absd(j) = ADA(j,j)  (getada.m:40)
  const int j = blockIdx.x + jbase;
  const int ipj = invperm ? invperm[j] : 0;
  for (int64_t e = ADAjc[j] + threadIdx.x; e < ADAjc[j + 1]; e += blockDim.x) {
    const int i = ADAir[e];
    if (invperm && invperm[i] > ipj) continue;
    // tile (I,J) with I >= J holds G(i,j) at [j*m + i]; inside a diagonal tile both orders are present
    const int a = (i / TILE >= j / TILE) ? i : j, b = (i / TILE >= j / TILE) ? j : i;
    double v = 0.0;
    for (int s = 0; s < nsplit; s++) v += part[(int64_t)s * m * m + (int64_t)b * m + a];
    if (accumulate) ada[e] += v; else ada[e] = v;
    if (absd && i == j) absd[j] = accumulate ? ada[e] : v;
  }
}
// dsqr (k_dsqr) and the dense copy of DAt.q (zero fill included, through the inverse map) in one launch
__global__ void k_lq_q_prep(double *dsqr, const int *code, const double *dl, const double *ddet, int nlq, double *Qd, const double *qpr,
                            const int *qsrc, int64_t nq) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < nlq) { const int c = code[t]; dsqr[t] = c == -1 ? dl[t] : (c >= 0 ? ddet[c] : -ddet[-2 - c]); }
  if (t < nq) { const int sidx = qsrc[t]; Qd[t] = sidx < 0 ? 0.0 : qpr[sidx]; }
}
__global__ void k_q_densify(double *Qd, const double *qpr, const int64_t *dst, int64_t nnz) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < nnz) Qd[dst[t]] = qpr[t];
}

// ---- DAt.q (getDAtm.m:39-44): q(k,j) = d.q1(k) * A(trace row of cone k, j) + d.q2(cone k)' * A(norm-bound rows of
// cone k, j)  [extractA + spdiags product + ddot.c:66-160], one work-item per entry of the pattern, sums in row order
__global__ void k_datq(double *qpr, const int64_t *Qjc, const int *Qir, const int64_t *Ajc, const int64_t *Aend, const int *Air,
                       const double *Apr, const double *q1, const double *q2, const int64_t *qblk, int lpN) {
  const int j = blockIdx.x;
  const int64_t cb = Ajc[j], ce = Aend[j];
  for (int64_t e = Qjc[j] + threadIdx.x; e < Qjc[j + 1]; e += blockDim.x) {
    const int k = Qir[e];
    double v = 0.0;
    {
      const int rt = lpN + k;                                   // trace entry of cone k
      int64_t lo = cb, hi = ce;
      while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (Air[mid] < rt) lo = mid + 1; else hi = mid; }
      if (lo < ce && Air[lo] == rt) v = q1[k] * Apr[lo];
    }
    {
      const int64_t r0 = qblk[k], r1 = qblk[k + 1], base = qblk[0];
      int64_t lo = cb, hi = ce;
      while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (Air[mid] < r0) lo = mid + 1; else hi = mid; }
      double acc = 0.0;
      for (; lo < ce && Air[lo] < r1; lo++) acc += q2[Air[lo] - base] * Apr[lo];
      v += acc;
    }
    qpr[e] = v;
  }
}

// ---- stage 1: z_jk = (D_k sym(X_jk) D_k)[U_k]    (spscale.c:249-305)
struct Stage1Tab {
  const int *t_n, *t_nslot, *t_ulen, *t_herm, *s_col, *u_pos, *u_rc, *Air;
  const int64_t *t_slotptr, *t_udoff, *t_uoff, *t_zoff, *t_end, *s_nzptr;
  const double *Apr;
  const int *t_blk;
  const int64_t *psd_start;
};
// what the generic stage-1 kernel needs to finish column j itself (stage 2 riding in the task): null ada = two-stage form
struct Stage2Ride {
  double *ada, *absd;
  const int64_t *ADAjc, *Ajc, *Ajc_psd;
  const int *ADAir, *Ablk, *Aupos, *t_col, *invperm;
  const double *Apr;
};
__global__ void __launch_bounds__(512, 4)
k_psd_stage1(Stage1Tab T, const double *udsqr, double *zbuf, int ldsY, int task0, int nzcap, Stage2Ride R2, const int *order) {
  SDM_DYN_SMEM(smem);
  double *Y = (double *)smem;                       // Y[slot][row], chunk of CC slots (Hermitian: Re then Im plane)
  const int task = order ? order[blockIdx.x] : blockIdx.x + task0;      // (order: the XCD-aware dispatch of a launch over all tasks)
  const int n = T.t_n[task], nslot = T.t_nslot[task], ulen = T.t_ulen[task];
  const int herm = T.t_herm[task];
  const int64_t slot0 = T.t_slotptr[task], tend = T.t_end[task];
  const double *D = udsqr + T.t_udoff[task];
  const double *Di = D + (int64_t)n * n;            // imaginary part of a Hermitian D_k (udsqr = [Re; Im], A.8)
  const int *U = T.u_pos + T.t_uoff[task];
  double *z = zbuf + T.t_zoff[task];
  const int64_t rowbase = T.psd_start[T.t_blk[task]];
  const int tid = threadIdx.x, bs = blockDim.x;
  const int wave = tid >> 6, lane = tid & 63, nw = bs >> 6;
  // LDS: Y (and Yi) plus the rows D(col_t,:) of the chunk's slots (Dl; Hermitian: Re and Im, D(ct,j) = conj(D(j,ct)))
  int CC = ldsY / (n * (herm ? 4 : 2)); if (CC < 1) CC = 1; if (CC > nslot) CC = nslot;
  double *Yi = Y + (int64_t)CC * n;
  double *Dl = Y + (int64_t)(herm ? 2 : 1) * CC * n;
  double *Dli = Dl + (int64_t)CC * n;
  // the task's nonzeros (value, offset inside the block) behind the Y / D-row area: nzcap of them, sized by the launch for
  // its largest task (<= S1_NZ) -- as a static 18 KB array they cost every task of 23 nonzeros a workgroup slot per CU
  double *nzx = Y + ldsY;
  int *nzr = (int *)(nzx + nzcap);
  __shared__ int scol[256];
  const int64_t nzb = T.s_nzptr[slot0];
  const bool staged = tend - nzb <= nzcap;
  if (staged)
    for (int64_t u = nzb + tid; u < tend; u += bs) { nzx[u - nzb] = T.Apr[u]; nzr[u - nzb] = (int)(T.Air[u] - rowbase); }
  // Real blocks: the first S1_TREG targets of every work-item -- their indices and their sums over the chunks of slots -- live in
  // registers for the whole task: read once, written once.  (As a loop over `z[u] += v` behind `q = U[u]` every target of every chunk
  // was two dependent round trips to L2: ~11 targets per work-item and chunk on 64 blocks of order 200, 15 of the 20 us of a chunk.)
  constexpr int S1_TREG = 6;
  int rreg[S1_TREG], creg[S1_TREG];
  double zreg[S1_TREG];
  const int nreg = herm ? 0 : min(ulen, S1_TREG * bs);              // targets [0, nreg) are register targets
  if (!herm) {
#pragma unroll
    for (int k = 0; k < S1_TREG; k++) {
      const int q = U[min(tid + k * bs, max(ulen - 1, 0))];
      creg[k] = q / n; rreg[k] = q - creg[k] * n; zreg[k] = 0.0;
    }
  }
  for (int c0 = 0; c0 < nslot; c0 += CC) {
    const int cc = min(CC, nslot - c0);
    for (int t = tid; t < min(cc, 256); t += bs) scol[t] = T.s_col[slot0 + c0 + t];
    __syncthreads();
    // (1) Y[:,t] = sum_{nz in slot} x * D[:, row(nz)]          (realdmulx, spscale.c:73-107; cpxdmulx :128-224:
    //     a nonzero of the imaginary plane contributes (i x) * d_row:  Re -= x Im(d),  Im += x Re(d))
    // slots x rows flattened over the whole workgroup when there are fewer slots than wavefronts (MAXCUT: one slot
    // of 4000 rows), else one slot per wavefront at a time
    const bool flat = cc < nw;
    for (int t = flat ? 0 : wave; t < cc; t += flat ? 1 : nw) {
      const int64_t sb = T.s_nzptr[slot0 + c0 + t];
      const int64_t se = (c0 + t + 1 < nslot) ? T.s_nzptr[slot0 + c0 + t + 1] : tend;
      const int sc = t < 256 ? scol[t] : T.s_col[slot0 + c0 + t];
      const int part = sc >= n ? 1 : 0, col = sc - part * n;
      const int sub0 = part * n * n + col * n;                   // offset of the slot's column inside the block
      if (!herm && !flat && n <= 256) {
        // the four row groups of the slot side by side: all loads of a nonzero (and of the slot's D row) in flight together
        double a4[4] = {0.0, 0.0, 0.0, 0.0}, d4[4];
        int i4[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { i4[k] = min(lane + 64 * k, n - 1); d4[k] = D[(int64_t)col * n + i4[k]]; }
        for (int64_t u = sb; u < se; u++) {
          const int rx = (staged ? nzr[u - nzb] : (int)(T.Air[u] - rowbase)) - sub0;
          const double x = staged ? nzx[u - nzb] : T.Apr[u];
          double dr[4];
#pragma unroll
          for (int k = 0; k < 4; k++) dr[k] = D[(int64_t)rx * n + i4[k]];
#pragma unroll
          for (int k = 0; k < 4; k++) a4[k] += x * dr[k];
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
          if (lane + 64 * k < n) { Y[t * n + lane + 64 * k] = a4[k]; Dl[t * n + lane + 64 * k] = d4[k]; }
        continue;
      }
      for (int i = flat ? tid : lane; i < n; i += flat ? bs : 64) {
        double ar = 0.0, ai = 0.0;
        for (int64_t u = sb; u < se; u++) {
          const int rx = (staged ? nzr[u - nzb] : (int)(T.Air[u] - rowbase)) - sub0;
          const double x = staged ? nzx[u - nzb] : T.Apr[u], dr = D[(int64_t)rx * n + i];
          if (!herm) ar += x * dr;
          else {
            const double di = Di[(int64_t)rx * n + i];
            if (part) { ar -= x * di; ai += x * dr; } else { ar += x * dr; ai += x * di; }
          }
        }
        Y[t * n + i] = ar;
        Dl[t * n + i] = D[(int64_t)col * n + i];                  // Re D(col, i) (symmetric)
        if (herm) { Yi[t * n + i] = ai; Dli[t * n + i] = -Di[(int64_t)col * n + i]; }    // Im D(col, i) = -Im D(i, col)
      }
    }
    __syncthreads();
    // (2) targets: with DXD_rc = sum_t Y[r,t] D[col_t,c],
    //     real plane  z(r,c) (+)= Re(DXD_rc + DXD_cr) / 2     (spscale.c:283-304, :379-397)
    //     imag plane  z(r,c) (+)= Im(DXD_rc - DXD_cr) / 2     (spscale.c:411-434)
    if (!herm) {
      double sreg[S1_TREG];
#pragma unroll
      for (int k = 0; k < S1_TREG; k++) sreg[k] = 0.0;
#pragma unroll 1
      for (int t = 0; t < cc; t++) {
        const double *Yt = Y + t * n, *Dt = Dl + t * n;
#pragma unroll
        for (int k = 0; k < S1_TREG; k++) sreg[k] += Yt[rreg[k]] * Dt[creg[k]] + Yt[creg[k]] * Dt[rreg[k]];
      }
#pragma unroll
      for (int k = 0; k < S1_TREG; k++) zreg[k] += sreg[k] / 2;
    }
    for (int u = nreg + tid; u < ulen; u += bs) {
      int q = U[u];
      const int zpart = q >= n * n ? 1 : 0;
      q -= zpart * n * n;
      const int c = q / n, r = q - c * n;
      double v;
      if (!herm) {
        double a1 = 0.0, a2 = 0.0;
        for (int t = 0; t < cc; t++) {
          a1 += Y[t * n + r] * Dl[t * n + c];
          a2 += Y[t * n + c] * Dl[t * n + r];
        }
        v = (a1 + a2) / 2;
      } else {
        double rrc = 0.0, irc = 0.0, rcr = 0.0, icr = 0.0;      // Re/Im of DXD_rc and DXD_cr
        for (int t = 0; t < cc; t++) {
          const double drc = Dl[t * n + c], dic = Dli[t * n + c], drr = Dl[t * n + r], dir = Dli[t * n + r];
          const double yr_r = Y[t * n + r], yi_r = Yi[t * n + r], yr_c = Y[t * n + c], yi_c = Yi[t * n + c];
          rrc += yr_r * drc - yi_r * dic; irc += yr_r * dic + yi_r * drc;
          rcr += yr_c * drr - yi_c * dir; icr += yr_c * dir + yi_c * drr;
        }
        v = zpart ? (irc - icr) / 2 : (rrc + rcr) / 2;
      }
      if (c0 == 0) z[u] = v; else z[u] += v;
    }
    __syncthreads();
  }
  if (!herm && R2.ada && ulen <= S1_TREG * bs) {
    // Stage 2 of this constraint right here (it touches no other PSD block: z_j is exactly what this task has in registers): z_j goes
    // to LDS instead of zbuf -- 230 MB written and read again per unit on 64 blocks of order 200 -- and the pattern entries of column j
    // are formed one per work-item as k_psd_stage2 does (getada3.c:333-351, absd :341-347).
    double *zl = Y;                                                  // (Y / Dl are dead: behind the last barrier of the chunk loop)
#pragma unroll
    for (int k = 0; k < S1_TREG; k++) if (tid + k * bs < ulen) zl[tid + k * bs] = zreg[k];
    __syncthreads();
    const int j = R2.t_col[task], kblk = T.t_blk[task];
    const int ipj = R2.invperm ? R2.invperm[j] : 0;
    for (int64_t e = R2.ADAjc[j] + tid; e < R2.ADAjc[j + 1]; e += bs) {
      const int i = R2.ADAir[e];
      if (R2.invperm && R2.invperm[i] > ipj) continue;
      double acc = 0.0, aabs = 0.0;
      int64_t p = R2.Ajc_psd[i];
      const int64_t pe = R2.Ajc[i + 1];
      for (; p + 4 <= pe; p += 4) {                                  // 4 nonzeros of a_i in flight
        double x[4]; int b[4], u[4];
#pragma unroll
        for (int q = 0; q < 4; q++) { x[q] = R2.Apr[p + q]; b[q] = R2.Ablk[p + q]; u[q] = R2.Aupos[p + q]; }
#pragma unroll
        for (int q = 0; q < 4; q++) if (b[q] == kblk) { const double term = x[q] * zl[u[q]]; acc += term; aabs += fabs(term); }
      }
      for (; p < pe; p++) if (R2.Ablk[p] == kblk) { const double term = R2.Apr[p] * zl[R2.Aupos[p]]; acc += term; aabs += fabs(term); }
      const double base = R2.ada[e];
      R2.ada[e] = base + acc;
      if (i == j) R2.absd[j] = base + aabs;
    }
    return;
  }
  if (!herm) {
#pragma unroll
    for (int k = 0; k < S1_TREG; k++) if (tid + k * bs < ulen) z[tid + k * bs] = zreg[k];
  }
}

// ---- stage 1 on the FP64 matrix cores (blocks of order n <= S1_MAXN): per task Z = Y * D(cols,:) is a dense
// n x n x nslot GEMM (Y = D X(:,cols) as above), accumulated over chunks of S1_KC slots; every wavefront owns a
// fixed set of 16x16 tiles of Z in registers.  The targets are read off the finished Z in LDS:
// z(r,c) = (Z[r][c] + Z[c][r]) / 2  -- the same two sums as spscale.c:283-304.
__global__ void __launch_bounds__(64 * S1_WAVES, 4)
k_psd_stage1_mfma(Stage1Tab T, const double *udsqr, double *zbuf, int task0, const int *order, double *zero_ptr, long long zero_n) {
  SDM_DYN_SMEM(smem);
  // (a zero LP / Lorentz part of ADA': cleared here, by everybody a slice, instead of by a memset launch of its own in front of
  // this one -- stage 2, the first reader, is a launch later)
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < zero_n; i += (long long)gridDim.x * blockDim.x) zero_ptr[i] = 0.0;
  const int task = order ? order[blockIdx.x] : blockIdx.x + task0;      // heaviest tasks first (full-range launches)
  const int n = T.t_n[task], nslot = T.t_nslot[task], ulen = T.t_ulen[task];
  const int np = (n + 15) & ~15, nt = np >> 4, ntile = nt * nt;
  double *Yl = (double *)smem;                      // Yl[t*np + i], t < S1_KC
  double *Dl = Yl + S1_KC * np;                     // Dl[t*np + j] = D[col_t][j]
  double *Zl = (double *)smem;                      // Zl[i*np + j]: written after the last GEMM chunk, aliases Yl/Dl
                                                    // (LDS per task = max(2*16*np, np*np) doubles -> several tasks per CU)
  const int64_t slot0 = T.t_slotptr[task], tend = T.t_end[task];
  const double *D = udsqr + T.t_udoff[task];
  double *z = zbuf + T.t_zoff[task];
  const int64_t rowbase = T.psd_start[T.t_blk[task]];
  const int tid = threadIdx.x, bs = blockDim.x;
  const int wave = SDM_UNIFORM_INT(tid >> 6), lane = tid & 63, nw = bs >> 6;      // (the wavefront index in a scalar register)
  const int li = lane & 15, lk = lane >> 4;
  sdm_double4 acc[S1_MAXT];
  for (int x = 0; x < S1_MAXT; x++) for (int r = 0; r < 4; r++) acc[x][r] = 0.0;
  __shared__ double nzx[S1_NZ];                     // the task's nonzeros: value ...
  __shared__ int nzr[S1_NZ];                        // ... and row offset inside the block
  __shared__ int sbeg[S1_MAXN + 1], scol[S1_MAXN];  // slot -> first nonzero (relative), column of X_jk
  SDM_PHASE_BEGIN();
  // ONE round trip brings the whole task (nonzeros + slot table) into LDS: every later load of D depends on LDS only
  const int64_t nzb = T.s_nzptr[slot0];
  const bool staged = tend - nzb <= S1_NZ && nslot <= S1_MAXN;
  if (staged) {
    for (int64_t u = nzb + tid; u < tend; u += bs) { nzx[u - nzb] = T.Apr[u]; nzr[u - nzb] = (int)(T.Air[u] - rowbase); }
    for (int t = tid; t <= nslot; t += bs) sbeg[t] = (int)((t < nslot ? T.s_nzptr[slot0 + t] : tend) - nzb);
    for (int t = tid; t < nslot; t += bs) scol[t] = T.s_col[slot0 + t];
  }
  __syncthreads();
  SDM_PHASE(0);
  for (int c0 = 0; c0 < nslot; c0 += S1_KC) {
    const int cc = min(S1_KC, nslot - c0);
    for (int t = wave; t < S1_KC; t += nw) {
      if (t < cc) {
        int sb, se, col;
        if (staged) { sb = sbeg[c0 + t]; se = sbeg[c0 + t + 1]; col = scol[c0 + t]; }
        else {
          sb = (int)(T.s_nzptr[slot0 + c0 + t] - nzb);
          se = (int)(((c0 + t + 1 < nslot) ? T.s_nzptr[slot0 + c0 + t + 1] : tend) - nzb);
          col = T.s_col[slot0 + c0 + t];
        }
        for (int i = lane; i < np; i += 64) {
          double a = 0.0;
          if (i < n) {
            if (staged) {
              int u = sb;
              for (; u + 4 <= se; u += 4) {               // 4 independent column loads of D in flight
                const double d0 = D[(int64_t)(nzr[u] - col * n) * n + i], d1 = D[(int64_t)(nzr[u + 1] - col * n) * n + i];
                const double d2 = D[(int64_t)(nzr[u + 2] - col * n) * n + i], d3 = D[(int64_t)(nzr[u + 3] - col * n) * n + i];
                a += nzx[u] * d0; a += nzx[u + 1] * d1; a += nzx[u + 2] * d2; a += nzx[u + 3] * d3;
              }
              for (; u < se; u++) a += nzx[u] * D[(int64_t)(nzr[u] - col * n) * n + i];
            } else {
              for (int64_t u = nzb + sb; u < nzb + se; u++) a += T.Apr[u] * D[(int64_t)((int)(T.Air[u] - rowbase) - col * n) * n + i];
            }
          }
          Yl[t * np + i] = a;
          Dl[t * np + i] = i < n ? D[(int64_t)col * n + i] : 0.0;
        }
      } else {
        for (int i = lane; i < np; i += 64) { Yl[t * np + i] = 0.0; Dl[t * np + i] = 0.0; }
      }
    }
    SDM_PHASE(1);
    __syncthreads();
    SDM_PHASE(2);
    for (int x = 0; x < S1_MAXT; x++) {
      const int tile = wave + x * nw;
      if (tile < ntile) {                               // wave-uniform
        const int I = tile / nt, J = tile - I * nt;
#pragma unroll 4
        for (int q = 0; q < S1_KC / 4; q++) {
          const double a = Yl[(4 * q + lk) * np + I * 16 + li];
          const double b = Dl[(4 * q + lk) * np + J * 16 + li];
          acc[x] = SDM_MFMA_F64_16x16x4(a, b, acc[x]);
        }
      }
    }
    SDM_PHASE(3);
    __syncthreads();
    SDM_PHASE(4);
  }
  for (int x = 0; x < S1_MAXT; x++) {
    const int tile = wave + x * nw;
    if (tile < ntile) {
      const int I = tile / nt, J = tile - I * nt;
      for (int r = 0; r < 4; r++) Zl[(I * 16 + lk + 4 * r) * np + J * 16 + li] = acc[x][r];
    }
  }
  __syncthreads();
  SDM_PHASE(5);
  // (the targets' (r, c) packed at set-up and eight of them in flight per work-item: one look-up per target behind the other and a division
  // by the block's order each made this loop 7.6 us of a task's 25 -- in-kernel clocks, profiles/r08x_stage1_mfma_phases.txt)
  const int *RC = T.u_rc + T.t_uoff[task];
  for (int u0 = 0; u0 < ulen; u0 += 8 * bs) {
    int rc[8];
#pragma unroll
    for (int x = 0; x < 8; x++) rc[x] = RC[min(u0 + x * bs + tid, ulen - 1)];
#pragma unroll
    for (int x = 0; x < 8; x++) {
      const int u = u0 + x * bs + tid;
      const int r = rc[x] >> 16, c = rc[x] & 0xffff;
      if (u < ulen) z[u] = (Zl[r * np + c] + Zl[c * np + r]) / 2;
    }
  }
  SDM_PHASE(6);
#if defined(SDM_PHASES) && !defined(SDM_EMU)
  if (tid == 0) { atomicAdd(&sdm_phase_acc[30], 1ull); if (n > 40) atomicAdd(&sdm_phase_acc[31], 1ull); }
#endif
}

// ---- stage 2: ADA(i,j) += a_i[psd]' z_j ; absd fused (getada3.c:333-351).  Sparse ADA' patterns: one workgroup per
// column j walks the pattern entries of the column.  Short constraint rows: one entry per work-item, z_j staged in
// LDS when it fits (all entries of the column gather from it); long rows: one entry per wavefront.
__global__ void __launch_bounds__(256)
k_psd_stage2(double *ada, double *absd, const int64_t *ADAjc, const int *ADAir, const int64_t *Ajc,
             const int64_t *Ajc_psd, const double *Apr, const int *Ablk, const int *Aupos,
             const int64_t *c_taskptr, const int *t_blk, const int64_t *t_zoff, const int64_t *c_zlen, const double *zbuf,
             const int *invperm, int nblk, int thread_per_row, int jbase, int zlds) {
  SDM_DYN_SMEM(smem);
  long long *zo = (long long *)smem;                // block -> offset of z_jk relative to z_j, or -1
  double *zl = (double *)(zo + nblk);               // z_j staged in LDS (zlds doubles available)
  const int j = blockIdx.x + jbase;
  const int tid = threadIdx.x, bs = blockDim.x;
  const int64_t tb0 = c_taskptr[j], te0 = c_taskptr[j + 1];
  const bool jhas = te0 > tb0;
  const int64_t z0 = jhas ? t_zoff[tb0] : 0;        // the tasks of one constraint are consecutive in zbuf
  const int64_t zlen = c_zlen[j];
  for (int k = tid; k < nblk; k += bs) zo[k] = -1;
  __syncthreads();
  for (int64_t t = tb0 + tid; t < te0; t += bs) zo[t_blk[t]] = t_zoff[t] - z0;
  const bool staged = thread_per_row && jhas && zlen <= zlds;
  if (staged)
    for (int64_t u = tid; u < zlen; u += bs) zl[u] = zbuf[z0 + u];
  __syncthreads();
  const double *zsrc = staged ? zl : zbuf + z0;
  const int ipj = invperm ? invperm[j] : 0;
  if (thread_per_row) {
    for (int64_t e = ADAjc[j] + tid; e < ADAjc[j + 1]; e += bs) {
      const int i = ADAir[e];
      if (invperm && invperm[i] > ipj) continue;
      double acc = 0.0, aabs = 0.0;
      if (jhas) {
        int64_t t = Ajc_psd[i];
        const int64_t te = Ajc[i + 1];
        for (; t + 4 <= te; t += 4) {                    // 4 nonzeros of a_i in flight
          double x[4]; int b[4], u[4];
#pragma unroll
          for (int q = 0; q < 4; q++) { x[q] = Apr[t + q]; b[q] = Ablk[t + q]; u[q] = Aupos[t + q]; }
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const long long off = zo[b[q]];
            if (off >= 0) { const double term = x[q] * zsrc[off + u[q]]; acc += term; aabs += fabs(term); }
          }
        }
        for (; t < te; t++) {
          const long long off = zo[Ablk[t]];
          if (off >= 0) { const double term = Apr[t] * zsrc[off + Aupos[t]]; acc += term; aabs += fabs(term); }
        }
      }
      const double base = ada[e];
      ada[e] = base + acc;
      if (i == j) absd[j] = jhas ? base + aabs : 0.0;
    }
  } else {
    const int wave = tid >> 6, lane = tid & 63, nw = bs >> 6;
    for (int64_t e = ADAjc[j] + wave; e < ADAjc[j + 1]; e += nw) {
      const int i = ADAir[e];
      if (invperm && invperm[i] > ipj) continue;
      double acc = 0.0, aabs = 0.0;
      if (jhas)
        for (int64_t t = Ajc_psd[i] + lane; t < Ajc[i + 1]; t += 64) {
          const long long off = zo[Ablk[t]];
          if (off >= 0) { const double term = Apr[t] * zsrc[off + Aupos[t]]; acc += term; aabs += fabs(term); }
        }
      for (int off = 32; off > 0; off >>= 1) { acc += __shfl_down(acc, off); aabs += __shfl_down(aabs, off); }
      if (lane == 0) {
        const double base = ada[e];
        ada[e] = base + acc;
        if (i == j) absd[j] = jhas ? base + aabs : 0.0;
      }
    }
  }
}

// ---- PSD part WITHOUT the z_j detour, for constraints of at most S1_DIRECT_MAXNZ nonzeros per PSD block (MAXCUT: A_i = e_i e_i').  With
// so few nonzeros the pairwise form   a_i' z_j = sum_{(r,c) in A_i} a_i(r,c) * 1/2 sum_{(s,t) in A_j} x_j(s,t) (D(r,s) D(t,c) + D(c,s) D(t,r))
// costs a handful of D entries per pattern entry, while the two-stage form writes and re-reads every z_j on the whole union pattern (MAXCUT-4000:
// 128 MB each way, 715 MB of HBM traffic and two launches of 180 + 200 us).  One workgroup per column j, one pattern entry per work-item as
// in k_psd_stage2; the symmetry of D_k puts the work-item's own index last in every D access (coalesced).  absd as in getada3.c:341-347.
constexpr int S1_DIRECT_MAXNZ = 2;
// ---- the same for FULL columns of ADA' (every row present: a dense ADA' pattern, MAXCUT), JB columns per workgroup.  Round 6: k_psd_direct re-reads
// the description of constraint i (Ajc_psd[i], Ajc[i+1], Ablk / Air / Apr of its nonzeros: 36+ bytes) for every entry (i, j) -- on MAXCUT-4000
// 16 M entries x 10 load instructions, ~1.2 GB through the L2s for 256 MB of D_k and ADA': 133 us, 1.9 TB/s.  Here a work-item owns ROW i of the
// workgroup's JB columns: constraint i is looked up once, the entry's own index is i (no ADAir), and only the D_k terms and the store are per
// column.  Per entry the same terms in the same order as k_psd_direct: the same bits.
constexpr int DIRECT_JB = 8;
__global__ void __launch_bounds__(256)
k_psd_direct_cols(double *ada, double *absd, const int64_t *ADAjc, const int64_t *Ajc, const int64_t *Ajc_psd, const double *Apr,
                  const int *Air, const int *Ablk, const int64_t *c_taskptr, const int *t_blk, const int *t_n, const int64_t *t_udoff,
                  const int64_t *t_slotptr, const int64_t *s_nzptr, const int64_t *t_end, const int64_t *psd_start, const double *udsqr,
                  int nblk, int jbase, int jend, int m, int base_zero) {
  SDM_DYN_SMEM(smem);
  // per column jj of the workgroup and block k: the tables of k_psd_direct
  const int per_i = 2 * nblk + 2 * S1_DIRECT_MAXNZ * nblk + 2, per_ll = 2 * nblk, per_d = S1_DIRECT_MAXNZ * nblk;
  int *bn_all = (int *)smem;                                                              // [JB][per_i]
  long long *ll_all = (long long *)(bn_all + DIRECT_JB * (per_i + (per_i & 1)));          // [JB][per_ll]  (8-byte aligned)
  double *bx_all = (double *)(ll_all + DIRECT_JB * per_ll);                               // [JB][per_d]
  __shared__ int jhas_s[DIRECT_JB];
  int *bnk = (int *)(bx_all + DIRECT_JB * per_d);                                         // [nblk] order of block k if one of the columns touches it, else 0
  const int tid = threadIdx.x, bs = blockDim.x;
  const int j0 = jbase + DIRECT_JB * blockIdx.x;
  const int stride_i = per_i + (per_i & 1);
  for (int jj = 0; jj < DIRECT_JB; jj++) {
    int *bn = bn_all + jj * stride_i, *bnz = bn + nblk, *brc = bnz + nblk;
    long long *bud = ll_all + jj * per_ll, *bst = bud + nblk;
    double *bx = bx_all + jj * per_d;
    for (int k = tid; k < nblk; k += bs) {
      bn[k] = 0; bnz[k] = 0; bst[k] = psd_start[k]; bud[k] = 0;
      for (int w = 0; w < S1_DIRECT_MAXNZ; w++) { brc[(k * S1_DIRECT_MAXNZ + w) * 2] = 0; brc[(k * S1_DIRECT_MAXNZ + w) * 2 + 1] = 0; bx[k * S1_DIRECT_MAXNZ + w] = 0.0; }
    }
  }
  __syncthreads();
  for (int jj = 0; jj < DIRECT_JB; jj++) {
    const int j = j0 + jj;
    if (j >= jend) { if (tid == 0) jhas_s[jj] = 0; continue; }
    int *bn = bn_all + jj * stride_i, *bnz = bn + nblk, *brc = bnz + nblk;
    long long *bud = ll_all + jj * per_ll;
    double *bx = bx_all + jj * per_d;
    const int64_t tb0 = c_taskptr[j], te0 = c_taskptr[j + 1];
    if (tid == 0) jhas_s[jj] = te0 > tb0 ? 1 : 0;
    for (int64_t t = tb0 + tid; t < te0; t += bs) {
      const int k = t_blk[t], n = t_n[t];
      const int64_t u0 = s_nzptr[t_slotptr[t]], u1 = t_end[t];
      bn[k] = n; bud[k] = t_udoff[t]; bnz[k] = (int)(u1 - u0);
      for (int64_t u = u0; u < u1 && u - u0 < S1_DIRECT_MAXNZ; u++) {
        const int q = (int)(Air[u] - psd_start[k]);
        brc[(k * S1_DIRECT_MAXNZ + (int)(u - u0)) * 2] = q % n; brc[(k * S1_DIRECT_MAXNZ + (int)(u - u0)) * 2 + 1] = q / n;
        bx[k * S1_DIRECT_MAXNZ + (int)(u - u0)] = Apr[u];
      }
    }
  }
  __syncthreads();
  for (int k = tid; k < nblk; k += bs) {
    int nk = 0;
    for (int jj = 0; jj < DIRECT_JB; jj++) nk = max(nk, (bn_all + jj * stride_i)[k]);
    bnk[k] = nk;
  }
  __syncthreads();
  const long long *bst0 = ll_all + nblk;                                                  // (the blocks' first rows: the same for every column)
  // (the rows are split over gridDim.y workgroups: with all m rows in one workgroup the launch had 2 wavefronts per SIMD and ran at the
  // latency of its dependent loads -- 175 us against the 133 of k_psd_direct)
  const int rows_per = ((m + (int)gridDim.y - 1) / (int)gridDim.y + 1) & ~1;
  const int i_end = min(m, ((int)blockIdx.y + 1) * rows_per);
  for (int i = (int)blockIdx.y * rows_per + tid; i < i_end; i += bs) {
    double acc[DIRECT_JB], aabs[DIRECT_JB];
#pragma unroll
    for (int jj = 0; jj < DIRECT_JB; jj++) { acc[jj] = 0.0; aabs[jj] = 0.0; }
    for (int64_t p = Ajc_psd[i]; p < Ajc[i + 1]; p++) {
      const int k = Ablk[p];
      const int n = bnk[k];                                                               // (the block's order: one division per nonzero of a_i, not per column)
      if (n == 0) continue;
      const int q = (int)(Air[p] - bst0[k]);
      const double xp = Apr[p];
      const int c = q / n, r = q - c * n;
#pragma unroll
      for (int jj = 0; jj < DIRECT_JB; jj++) {
        const int *bn = bn_all + jj * stride_i;
        if (bn[k] == 0 || !jhas_s[jj]) continue;
        const int *bnz = bn + nblk, *brc = bnz + nblk;
        const double *D = udsqr + (ll_all + jj * per_ll)[k];
        const double *bx = bx_all + jj * per_d;
        double zv = 0.0;
        for (int w = 0; w < bnz[k]; w++) {
          const int s2 = brc[(k * S1_DIRECT_MAXNZ + w) * 2], t2 = brc[(k * S1_DIRECT_MAXNZ + w) * 2 + 1];
          const double *Ds = D + (int64_t)s2 * n, *Dt = D + (int64_t)t2 * n;
          zv += bx[k * S1_DIRECT_MAXNZ + w] * (Ds[r] * Dt[c] + Ds[c] * Dt[r]);
        }
        const double term = xp * (zv / 2);
        acc[jj] += term; aabs[jj] += fabs(term);
      }
    }
#pragma unroll
    for (int jj = 0; jj < DIRECT_JB; jj++) {
      const int j = j0 + jj;
      if (j >= jend) continue;
      const int64_t e = ADAjc[j] + i;                                                     // (full column: row i is its i-th entry)
      const double base = base_zero ? 0.0 : ada[e];
      ada[e] = base + acc[jj];
      if (i == j) absd[j] = jhas_s[jj] ? base + aabs[jj] : 0.0;
    }
  }
}

constexpr size_t S1_DIRECT_LDS_MAX = 48 * 1024;   // its per-block tables (56 bytes per PSD block) must fit the default dynamic-LDS limit
__global__ void __launch_bounds__(256)
k_psd_direct(double *ada, double *absd, const int64_t *ADAjc, const int *ADAir, const int64_t *Ajc, const int64_t *Ajc_psd, const double *Apr,
             const int *Air, const int *Ablk, const int64_t *c_taskptr, const int *t_blk, const int *t_n, const int64_t *t_udoff,
             const int64_t *t_slotptr, const int64_t *s_nzptr, const int64_t *t_end, const int64_t *psd_start, const double *udsqr,
             const int *invperm, int nblk, int jbase, int base_zero) {
  // base_zero: the ADA' handed in is the zero matrix ada_lq left uncleared (ada_zero_flush): nothing is read, every entry of the
  // column is written -- the memset and the read of the zeros were 256 of this stage's 512 MB at n = 4000
  SDM_DYN_SMEM(smem);
  // per block of column j: order n (0 = no nonzero of A_j there), offsets of D_k and of the block's rows, and the nonzeros of A_jk
  int *bn = (int *)smem;                                   // [nblk]
  int *bnz = bn + nblk;                                    // [nblk] number of nonzeros
  int *brc = bnz + nblk;                                   // [nblk][MAXNZ][2] (row, column)
  long long *bud = (long long *)(brc + 2 * S1_DIRECT_MAXNZ * nblk);     // [nblk] (8-byte aligned: an even number of ints before it)
  long long *bst = bud + nblk;                             // [nblk]
  double *bx = (double *)(bst + nblk);                     // [nblk][MAXNZ]
  const int j = blockIdx.x + jbase;
  const int tid = threadIdx.x, bs = blockDim.x;
  const int64_t tb0 = c_taskptr[j], te0 = c_taskptr[j + 1];
  const bool jhas = te0 > tb0;
  for (int k = tid; k < nblk; k += bs) {
    bn[k] = 0; bnz[k] = 0; bst[k] = psd_start[k]; bud[k] = 0;
    for (int w = 0; w < S1_DIRECT_MAXNZ; w++) { brc[(k * S1_DIRECT_MAXNZ + w) * 2] = 0; brc[(k * S1_DIRECT_MAXNZ + w) * 2 + 1] = 0; bx[k * S1_DIRECT_MAXNZ + w] = 0.0; }
  }
  __syncthreads();
  for (int64_t t = tb0 + tid; t < te0; t += bs) {
    const int k = t_blk[t], n = t_n[t];
    const int64_t u0 = s_nzptr[t_slotptr[t]], u1 = t_end[t];
    bn[k] = n; bud[k] = t_udoff[t]; bnz[k] = (int)(u1 - u0);
    for (int64_t u = u0; u < u1 && u - u0 < S1_DIRECT_MAXNZ; u++) {
      const int q = (int)(Air[u] - psd_start[k]);
      brc[(k * S1_DIRECT_MAXNZ + (int)(u - u0)) * 2] = q % n; brc[(k * S1_DIRECT_MAXNZ + (int)(u - u0)) * 2 + 1] = q / n;
      bx[k * S1_DIRECT_MAXNZ + (int)(u - u0)] = Apr[u];
    }
  }
  __syncthreads();
  const int ipj = invperm ? invperm[j] : 0;
  // (four pattern entries per work-item in flight with straight-line terms measured no faster: 172 against 164 us on MAXCUT-4000)
  for (int64_t e = ADAjc[j] + tid; e < ADAjc[j + 1]; e += bs) {
    const int i = ADAir[e];
    if (invperm && invperm[i] > ipj) continue;
    double acc = 0.0, aabs = 0.0;
    if (jhas)
      for (int64_t p = Ajc_psd[i]; p < Ajc[i + 1]; p++) {
        const int k = Ablk[p], n = bn[k];
        if (n == 0) continue;
        const double *D = udsqr + bud[k];
        const int q = (int)(Air[p] - bst[k]);
        const int c = q / n, r = q - c * n;
        double zv = 0.0;
        for (int w = 0; w < bnz[k]; w++) {
          const int s2 = brc[(k * S1_DIRECT_MAXNZ + w) * 2], t2 = brc[(k * S1_DIRECT_MAXNZ + w) * 2 + 1];
          const double *Ds = D + (int64_t)s2 * n, *Dt = D + (int64_t)t2 * n;
          zv += bx[k * S1_DIRECT_MAXNZ + w] * (Ds[r] * Dt[c] + Ds[c] * Dt[r]);
        }
        const double term = Apr[p] * (zv / 2);
        acc += term; aabs += fabs(term);
      }
    const double base = base_zero ? 0.0 : ada[e];
    ada[e] = base + acc;
    if (i == j) absd[j] = jhas ? base + aabs : 0.0;
  }
}

// ---- stage 2, dense-ish patterns: one workgroup per JB consecutive ADA columns.  Their z_j (all blocks touched
// by constraint j) are staged in LDS once; the wavefronts then sweep the groups of 64 rows, ONE ROW PER LANE, over
// the interleaved (ELL) copy of the PSD nonzeros: coalesced loads (each feeding JB columns), LDS gathers, no
// cross-lane reduction, one writer per entry.
template <int JB>
__global__ void __launch_bounds__(64 * ELL_WAVES)
k_psd_stage2_ell(double *ada, double *absd, const int64_t *ADAjc, const int *ADAir, const int64_t *Ajc,
                 const int64_t *Ajc_psd, const double *Apr, const int *Ablk, const int *Aupos,
                 const int64_t *c_taskptr, const int *t_blk, const int *t_ulen, const int64_t *t_zoff, const double *zbuf,
                 const int64_t *uoff, const int *g_row, const int *g_len, const int64_t *g_off, const double *g_val,
                 const int *g_pos, int ngroups, const int *invperm, int zmax, int m, int jbase, int jend, const int *ell_pos, const int *ell_order,
                 const int *Azpos, const int *t_zdst, int base_zero, const long long *cd64, const int *cd32) {
  // ell_order != null (full ADA' on a full pattern): the workgroup's columns are order[jbase + ...] -- neighbours in the ELL row
  // order -- and only the row groups from its columns' own group on are swept: an entry (i, j) with group(i) > group(j) is
  // computed once, at column j, and added to (j, i) as well (the PSD part is symmetric: <a_i, z_j> = <a_j, z_i>); pairs inside
  // one group are computed at both columns.  Half the sweep.
  SDM_DYN_SMEM(smem);
  double *zl = (double *)smem;                      // JB x z_j at full length (all blocks; zero where j has no nonzero)
  __shared__ double absred[JB][ELL_WAVES];
  __shared__ double part[JB][ELL_WAVES][64];
  const int j0 = jbase + blockIdx.x * JB;
  const int tid = threadIdx.x, bs = blockDim.x;
  const int wave = tid >> 6, lane = tid & 63, nw = bs >> 6;
  int gq[JB], gstart = 0;                                  // ELL group of each of the workgroup's columns
  if (ell_order) {                                         // (the column at ELL position x sits in group x / 64: nothing to look up)
    gstart = ngroups;
#pragma unroll
    for (int q = 0; q < JB; q++) { gq[q] = min(j0 + q, jend - 1) >> 6; gstart = min(gstart, gq[q]); }
  } else {
#pragma unroll
    for (int q = 0; q < JB; q++) gq[q] = 0;
  }
  // Round 6 (in-kernel clocks, profiles/r08x_stage2_ell_phases.txt): the prologue was 44 % of a workgroup's 65 us on control07 -- z_j staged
  // task by task, each behind three dependent look-ups and a copy loop with one load in flight per work-item (15.7 us), absd and the diagonal's
  // base value in ALL workgroups of a column set although one owns the diagonal (12.3 us) -- and a round trip under this launch's load is 3 - 5
  // us, so what counts is their NUMBER in front of the first product.  Now two: (1) ONE record per column (cd64 / cd32, built at set-up, indexed
  // by ELL position on the symmetric path: column, task range, PSD range, up to four (source, destination, length) segments of z_j) and, beside
  // it, the first row group's header; (2) every segment's entries in flight together.  (The sweep's first eight entries fetched ahead as well, beside
  // the segments: 35.0 us against 31.5 without -- more registers held across the prologue, nothing hidden.)
  constexpr int SEG_MAX = 4 * JB;
  __shared__ long long sg_src[SEG_MAX];
  __shared__ int sg_dst[SEG_MAX], sg_cum[SEG_MAX + 1];
  __shared__ long long tb_s[JB], te_s[JB], ab_s[JB], ae_s[JB];
  __shared__ int jc_s[JB];
  for (int k = tid; k < zmax * JB; k += bs) zl[k] = 0.0;
  if (tid < SEG_MAX) {
    const int q = tid >> 2, sgi = tid & 3;
    const int x = min(j0 + q, jend - 1);
    const bool in = j0 + q < jend;
    const int nt = cd32[16 * x + 1];
    if (sgi == 0) {
      jc_s[q] = cd32[16 * x];
      const long long tb = cd64[8 * x];
      tb_s[q] = in ? tb : 0; te_s[q] = in ? tb + nt : 0;
      ab_s[q] = in ? cd64[8 * x + 1] : 0; ae_s[q] = in ? cd64[8 * x + 2] : 0;
    }
    const bool live = in && sgi < nt;
    sg_src[tid] = cd64[8 * x + 3 + sgi]; sg_dst[tid] = q * zmax + cd32[16 * x + 2 + sgi]; sg_cum[tid + 1] = live ? cd32[16 * x + 6 + sgi] : 0;
  }
  if (tid == 0) sg_cum[0] = 0;
  // the first row group of this workgroup: its header now (it does not depend on z_j)
  const int g0 = gstart + (int)blockIdx.y;
  int i0 = -1, len0 = 0;
  long long off0 = 0;
  if (g0 < ngroups) { i0 = g_row[g0 * 64 + lane]; len0 = g_len[g0]; off0 = g_off[g0]; }
  __syncthreads();
#define ELLCOL(x) (jc_s[(x) - j0 < JB ? (x) - j0 : JB - 1])           /* the workgroup's columns only: from the record */
  bool jhas[JB], own[JB], owns = false;
  bool inl = true;                                                   // every column's segments are in its record
#pragma unroll
  for (int q = 0; q < JB; q++) {
    jhas[q] = te_s[q] > tb_s[q];
    inl = inl && te_s[q] - tb_s[q] <= 4;
    // (the row groups are split over gridDim.y workgroups: the one that owns row j's group is the only writer of entry (j,j) and of absd(j))
    const int grp = ell_order ? gq[q] : (ell_pos[min(j0 + q, jend - 1)] >> 6);
    own[q] = j0 + q < jend && (int)blockIdx.y == (grp - gstart) % (int)gridDim.y;
    owns = owns || own[q];
  }
  if (inl) {
    if (tid == 0) for (int x = 0; x < SEG_MAX; x++) sg_cum[x + 1] += sg_cum[x];
    __syncthreads();
    const int ntot = sg_cum[SEG_MAX];
    for (int base = 0; base < ntot; base += 12 * bs) {               // 12 entries per work-item in flight (two columns of control07: one trip)
      double v[12]; int dd[12];
#pragma unroll
      for (int x = 0; x < 12; x++) {
        const int idx = base + x * bs + tid;
        dd[x] = -1; v[x] = 0.0;
        if (idx < ntot) {
          int sgi = 0;
          while (idx >= sg_cum[sgi + 1]) sgi++;
          v[x] = zbuf[sg_src[sgi] + (idx - sg_cum[sgi])]; dd[x] = sg_dst[sgi] + (idx - sg_cum[sgi]);
        }
      }
#pragma unroll
      for (int x = 0; x < 12; x++) if (dd[x] >= 0) zl[dd[x]] = v[x];
    }
  } else {
#pragma unroll
    for (int q = 0; q < JB; q++) {
      double *zq = zl + (size_t)q * zmax;
      for (int64_t t = tb_s[q]; t < te_s[q]; t++) {                  // the blocks touched by constraint j
        const double *src = zbuf + t_zoff[t];
        double *dst = zq + t_zdst[t];
        const int ul = t_ulen[t];
        for (int u = tid; u < ul; u += bs) dst[u] = src[u];
      }
    }
  }
  __syncthreads();
  if (owns) {                                                        // (uniform for the workgroup)
    // absd(j) = ADA_jj (LP/Lorentz part so far) + sum |a_j[psd] .* z_j|   (getada3.c:341-347); 0 without PSD nonzeros
#pragma unroll
    for (int q = 0; q < JB; q++) {
      double aabs = 0.0;
      if (own[q] && jhas[q]) {
        const double *zq = zl + (size_t)q * zmax;
        for (int64_t t = ab_s[q] + tid; t < ae_s[q]; t += bs) aabs += fabs(Apr[t] * zq[Azpos[t]]);
      }
      for (int off = 32; off > 0; off >>= 1) aabs += __shfl_down(aabs, off);
      if (lane == 0) absred[q][wave] = aabs;
    }
    __syncthreads();
    // (the owner reads the entry's LP/Lorentz value here, before it adds to it)
    if (tid < JB && own[tid]) {                                      // own[] / jhas[] are indexed by a per-lane value only in these JB lanes
      const int j = ELLCOL(j0 + tid);
      double basev = 0.0;
      int64_t lo = ADAjc[j], hi = ADAjc[j + 1];
      const int64_t ce = hi;
      if (ce - lo == m) lo += j;                           // full column: no search
      else while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (ADAir[mid] < j) lo = mid + 1; else hi = mid; }
      if (lo < ce && ADAir[lo] == j) basev = ada[lo];
      double a = 0.0;
      for (int w = 0; w < nw; w++) a += absred[tid][w];
      absd[j] = jhas[tid] ? basev + a : 0.0;
    }
    __syncthreads();
  }
  // Every group of 64 rows is swept by ALL wavefronts: wave w takes the entries t = w, w+nw, ... of the 64 rows (one
  // row per lane), so the longest row costs len/nw dependent memory round trips instead of len; the nw partial
  // sums of a row meet in LDS and are added in wave order (deterministic).
  for (int g = g0; g < ngroups; g += gridDim.y) {
    const bool first = g == g0;
    const int i = first ? i0 : g_row[g * 64 + lane];
    const int len = first ? len0 : g_len[g];
    const int64_t goff = first ? off0 : g_off[g];
    const double *gv = g_val + goff * 64 + lane;
    const int *gp = g_pos + goff * 64 + lane;
    double acc[JB];
#pragma unroll
    for (int q = 0; q < JB; q++) acc[q] = 0.0;
    int t = wave;
    for (; t + 7 * nw < len; t += 8 * nw) {             // 8 entries per lane in flight
      double v[8]; int pos[8];
#pragma unroll
      for (int x = 0; x < 8; x++) { const int64_t tt = t + x * nw; v[x] = gv[tt * 64]; pos[x] = gp[tt * 64]; }
#pragma unroll
      for (int x = 0; x < 8; x++)
#pragma unroll
        for (int q = 0; q < JB; q++) acc[q] += v[x] * zl[(size_t)q * zmax + pos[x]];
    }
    for (; t < len; t += nw) {
      const double v = gv[(int64_t)t * 64];
      const int pos = gp[(int64_t)t * 64];
#pragma unroll
      for (int q = 0; q < JB; q++) acc[q] += v * zl[(size_t)q * zmax + pos];
    }
#pragma unroll
    for (int q = 0; q < JB; q++) part[q][wave][lane] = acc[q];
    __syncthreads();
    if (wave < JB && i >= 0 && j0 + wave < jend && g >= gq[wave]) {      // wave q finishes column q of this group
      const int q = wave, j = ELLCOL(j0 + q);
      if (jhas[q] && !(invperm && invperm[i] > invperm[j])) {
        int64_t lo = 0, hi = 0, e = -1;
        if (ell_order) e = (int64_t)j * m + i;            // (the full pattern: nothing to look up)
        else { lo = ADAjc[j]; hi = ADAjc[j + 1]; }
        const int64_t ce = hi;
        if (ell_order) {}
        else if (ce - lo == m) e = lo + i;                // full column: no search
        else {
          while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (ADAir[mid] < i) lo = mid + 1; else hi = mid; }
          if (lo < ce && ADAir[lo] == i) e = lo;
        }
        if (e >= 0) {
          double a = 0.0;
          for (int w2 = 0; w2 < nw; w2++) a += part[q][w2][lane];
          // (base_zero: the LP / Lorentz part is the zero matrix the stage-1 launch has just written -- nothing to read; 0.0 + a: the bits of the sum)
          if (base_zero) ada[e] = 0.0 + a; else ada[e] += a;
          if (ell_order && g > gq[q]) {                                // (j, i): not computed anywhere else (full columns: it sits at i m + j)
            const int64_t em = (int64_t)i * m + j;
            if (base_zero) ada[em] = 0.0 + a; else ada[em] += a;
          }
        }
      }
    }
    __syncthreads();
  }
}

// spmakesym (getada3.c:151-180): out(i,j) = in(i,j) + in(j,i) for i != j
// (the transposed entries in[(j, i)] of a column are a stride of a column apart: a 128-byte line serves 16 neighbouring columns.  Workgroup b runs
// on XCD b % 8: groups of 128 columns are dealt 16 neighbours to an XCD, so that the line is fetched into ONE L2 and used 16 times there --
// MAXCUT-4000: 1.76 GB per launch for a 128 MB matrix with column j = b, profiles/r08u_maxcut4000_pmc_traffic.json)
__global__ void k_symmetrize(double *out, const double *in, const int64_t *ADAjc, const int *ADAir, const int *adaT, int m) {
  int j = blockIdx.x;
  if (j < (m / 128) * 128) { const int r = j & 127; j = (j & ~127) + (r & 7) * 16 + (r >> 3); }
  for (int64_t e = ADAjc[j] + threadIdx.x; e < ADAjc[j + 1]; e += blockDim.x) {
    const int i = ADAir[e];
    double v = in[e];
    if (i != j) { const int et = adaT[e]; if (et >= 0) v += in[et]; }
    out[e] = v;
  }
}
// cpspdiag (getada3.c:119-135): absd = diag(ADA) when there are no PSD blocks
__global__ void k_diag(double *absd, const double *ada, const int64_t *ADAjc, const int *ADAir, int m) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  int64_t lo = ADAjc[j], hi = ADAjc[j + 1];            // sorted row indices: binary search for the diagonal
  const int64_t ce = hi;
  while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (ADAir[mid] < j) lo = mid + 1; else hi = mid; }
  absd[j] = (lo < ce && ADAir[lo] == j) ? ada[lo] : 0.0;
}

// ============================================================ host drivers
// Column range [A.col0, A.col1) of ADA' (all columns by default): every stage below only touches those columns,
// which is what lets the ranks of a job form disjoint column panels of one ADA' (sdm_plan_getada_cols).
// the zero LP / Lorentz part of ADA' that ada_lq left to the next stage (AdaPlan::zero_ptr): cleared now, by a memset, unless
// the stage-1 launch of ada_psd has taken it over.  Everything that reads or adds to ADA' calls this first.
void ada_zero_flush(sdm_plan *P) {
  AdaPlan &A = P->ada;
  if (A.zero_ptr && A.zero_n > 0) SDM_HIP_CHECK(hipMemsetAsync(A.zero_ptr, 0, (size_t)A.zero_n * sizeof(double), P->stream));
  A.zero_ptr = nullptr; A.zero_n = 0;
}
void ada_lq(sdm_plan *P, double *ada, const int *d_invperm, bool accumulate) {
  AdaPlan &A = P->ada;
  if (A.col1 <= A.col0) return;
  if (A.nnz_lq == 0 && !accumulate && !d_invperm) {
    // no LP / Lorentz nonzeros at all (e.g. MAXCUT): the LP part of ADA' is the zero matrix -- one memset of the
    // column panel instead of nnz(ADA') empty sparse dot products
    const sdm_int e0 = P->ada_jc[A.col0], e1 = P->ada_jc[A.col1];
    ada_zero_flush(P);
    if (e1 > e0) { A.zero_ptr = ada + e0; A.zero_n = (long long)(e1 - e0); }
    if (!A.zero_defer) ada_zero_flush(P);                            // (only sdm_plan_getada's own sequence lets the next stage take it over)
    return;
  }
  if (A.nlq > 0)
    SDM_KLAUNCH(P, k_dsqr, dim3((unsigned)((A.nlq + 255) / 256)), dim3(256), 0, A.dsqr.p, A.dsqr_code.p, A.dl.p, A.ddet.p, (int)A.nlq);
  if (A.lq_dense) {
    const int m = (int)A.m, nt = (m + TILE - 1) / TILE;
    SDM_KLAUNCH(P, k_gram_tile, dim3(nt * (nt + 1) / 2, A.gram_split), dim3(256), 0, A.Alq_d.p, A.dsqr.p, (int)A.nlq, m, A.gram_split,
                A.gram_part.p);
    SDM_KLAUNCH(P, k_gram_scatter, dim3((unsigned)(A.col1 - A.col0)), dim3(128), 0, ada, A.d_ADAjc.p, A.d_ADAir.p, A.gram_part.p,
                A.gram_split, m, d_invperm, accumulate ? 1 : 0, (int)A.col0);
    return;
  }
  if (A.lq_maxcol <= 256)                                            // (measured: 200 nonzeros per column are still faster four entries at a time;
                                                                     //  ONE column of 793 is not: its single entry is the whole launch)
    SDM_KLAUNCH(P, k_ada_spdot<16>, dim3((unsigned)(A.col1 - A.col0)), dim3(256), (size_t)SPDOT_CAP * 12, ada, A.d_ADAjc.p, A.d_ADAir.p, A.d_Ajc.p, A.d_Ajc_psd.p,
               A.d_Air.p, A.d_Apr.p, A.dsqr.p, d_invperm, accumulate ? 1 : 0, (int)A.col0, SPDOT_CAP);
  else
    SDM_KLAUNCH(P, k_ada_spdot<64>, dim3((unsigned)(A.col1 - A.col0)), dim3(256), (size_t)SPDOT_CAP * 12, ada, A.d_ADAjc.p, A.d_ADAir.p, A.d_Ajc.p, A.d_Ajc_psd.p,
               A.d_Air.p, A.d_Apr.p, A.dsqr.p, d_invperm, accumulate ? 1 : 0, (int)A.col0, SPDOT_CAP);
}
void ada_datq(sdm_plan *P) {
  AdaPlan &A = P->ada;
  if (A.lorN == 0 || A.nnzQ == 0) return;
  SDM_KLAUNCH(P, k_datq, dim3((unsigned)A.m), dim3(128), 0, A.qpr.p, A.d_Qjc.p, A.d_Qir.p, A.d_Ajc.p, A.d_Ajc_psd.p, A.d_Air.p, A.d_Apr.p,
              A.q1.p, A.q2.p, A.d_qblk.p, (int)A.lpN);
}
void ada_q(sdm_plan *P, double *ada, const int *d_invperm, bool accumulate) {
  AdaPlan &A = P->ada;
  if (A.lorN == 0 || A.nnzQ == 0 || A.col1 <= A.col0) return;
  ada_zero_flush(P);
  if (A.q_dense) {
    const int m = (int)A.m, nt = (m + TILE - 1) / TILE;
    SDM_HIP_CHECK(hipMemsetAsync(A.Q_d.p, 0, (size_t)A.lorN * (size_t)m * sizeof(double), P->stream));
    SDM_KLAUNCH(P, k_q_densify, dim3((unsigned)((A.nnzQ + 255) / 256)), dim3(256), 0, A.Q_d.p, A.qpr.p, A.q_dst.p, (int64_t)A.nnzQ);
    SDM_KLAUNCH(P, k_gram_tile, dim3(nt * (nt + 1) / 2, A.gram_split), dim3(256), 0, A.Q_d.p, (const double *)nullptr, (int)A.lorN, m,
                A.gram_split, A.gram_part.p);
    SDM_KLAUNCH(P, k_gram_scatter, dim3((unsigned)(A.col1 - A.col0)), dim3(128), 0, ada, A.d_ADAjc.p, A.d_ADAir.p, A.gram_part.p,
                A.gram_split, m, d_invperm, accumulate ? 1 : 0, (int)A.col0);
    return;
  }
  if (A.q_maxcol <= 256)
    SDM_KLAUNCH(P, k_ada_spdot<16>, dim3((unsigned)(A.col1 - A.col0)), dim3(256), (size_t)SPDOT_CAP * 12, ada, A.d_ADAjc.p, A.d_ADAir.p, A.d_Qjc.p, A.d_Qjc.p + 1,
               A.d_Qir.p, A.qpr.p, (const double *)nullptr, d_invperm, accumulate ? 1 : 0, (int)A.col0, SPDOT_CAP);
  else
    SDM_KLAUNCH(P, k_ada_spdot<64>, dim3((unsigned)(A.col1 - A.col0)), dim3(256), (size_t)SPDOT_CAP * 12, ada, A.d_ADAjc.p, A.d_ADAir.p, A.d_Qjc.p, A.d_Qjc.p + 1,
               A.d_Qir.p, A.qpr.p, (const double *)nullptr, d_invperm, accumulate ? 1 : 0, (int)A.col0, SPDOT_CAP);
}
// LP + Lorentz-det part and Lorentz rank-1 part of a FULL ADA' in three launches when both take the dense Gram form
// (nb.mat's shape): prep (dsqr, dense DAt.q), one Gram launch over the rows of both operands, one scatter -- instead of
// eight operations.  Returns false when the case does not apply (the caller then runs ada_lq + ada_q).
bool ada_lq_q(sdm_plan *P, double *ada) {
  AdaPlan &A = P->ada;
  const int m = (int)A.m;
  if (!(A.lq_dense && A.q_dense && A.q_src.n && A.col0 == 0 && A.col1 == A.m && A.lorN > 0 && A.nnzQ > 0)) return false;
  const int64_t nq = (int64_t)A.lorN * m, nprep = std::max<int64_t>(nq, A.nlq);
  const int nt = (m + TILE - 1) / TILE;
  SDM_KLAUNCH(P, k_lq_q_prep, dim3((unsigned)((nprep + 255) / 256)), dim3(256), 0, A.dsqr.p, A.dsqr_code.p, A.dl.p, A.ddet.p, (int)A.nlq,
              A.Q_d.p, A.qpr.p, A.q_src.p, nq);
  SDM_KLAUNCH(P, k_gram_tile, dim3(nt * (nt + 1) / 2, A.gram_split), dim3(256), 0, A.Alq_d.p, A.dsqr.p, (int)A.nlq, m, A.gram_split,
              A.gram_part.p, A.Q_d.p, (int)A.lorN);
  SDM_KLAUNCH(P, k_gram_scatter, dim3((unsigned)m), dim3(128), 0, ada, A.d_ADAjc.p, A.d_ADAir.p, A.gram_part.p, A.gram_split, m,
              (const int *)nullptr, 0, 0, A.sdpN == 0 ? P->absd.p : (double *)nullptr);
  return true;
}
void ada_psd(sdm_plan *P, double *ada, const int *d_invperm, bool sym_input, bool absd_done) {
  AdaPlan &A = P->ada;
  hipStream_t st = P->stream;
  const int m = (int)A.m;
  if (A.sdpN == 0 || sym_input || A.col1 <= A.col0) ada_zero_flush(P);
  if (A.sdpN == 0) {
    if (sym_input) {
      SDM_KLAUNCH(P, k_symmetrize, dim3(m), dim3(128), 0, A.symtmp.p, ada, A.d_ADAjc.p, A.d_ADAir.p, A.d_ADAT.p, m);
      SDM_HIP_CHECK(hipMemcpyAsync(ada, A.symtmp.p, A.symtmp.n * sizeof(double), hipMemcpyDeviceToDevice, st));
    }
    if (!absd_done) SDM_KLAUNCH(P, k_diag, dim3((m + 255) / 256), dim3(256), 0, P->absd.p, ada, A.d_ADAjc.p, A.d_ADAir.p, m);
    return;
  }
  if (A.col1 <= A.col0) return;
  const int ncols = (int)(A.col1 - A.col0), jbase = (int)A.col0;
  const int task0 = (int)A.h_taskptr[A.col0], ntask = (int)(A.h_taskptr[A.col1] - A.h_taskptr[A.col0]);
  // constraints of one or two nonzeros per PSD block on a pattern swept one entry per work-item: the pairwise form, no z_j (k_psd_direct)
  // (its per-block tables live in LDS, 56 bytes per PSD block: problems of more than ~850 blocks keep the two-stage path, whose
  // stage 2 needs 8 bytes per block and raises its LDS attribute itself)
  const size_t direct_lds = (size_t)(2 * A.sdpN + 2 * S1_DIRECT_MAXNZ * A.sdpN + 2) * sizeof(int) + (size_t)(2 * A.sdpN) * sizeof(long long) +
                            (size_t)(S1_DIRECT_MAXNZ * A.sdpN) * sizeof(double);
  const bool direct = A.thread_per_row && !A.ell_ok && A.sdpN == A.rsdpN && A.s1_maxnz <= S1_DIRECT_MAXNZ && direct_lds <= S1_DIRECT_LDS_MAX;
  // (the matrix-core stage 1 takes the clearing over; the pairwise form writes every entry of the panel itself when it has no skipped triangle)
  const bool direct_zero = direct && !sym_input && !d_invperm && A.zero_ptr == ada + P->ada_jc[A.col0] &&
                           A.zero_n == (long long)(P->ada_jc[A.col1] - P->ada_jc[A.col0]);
  if (direct_zero) { A.zero_ptr = nullptr; A.zero_n = 0; }
  if (direct || ntask <= 0 || !(A.maxn <= S1_MAXN && A.sdpN == A.rsdpN)) ada_zero_flush(P);
  if (direct) {
    if (sym_input) {
      SDM_KLAUNCH(P, k_symmetrize, dim3(m), dim3(128), 0, A.symtmp.p, ada, A.d_ADAjc.p, A.d_ADAir.p, A.d_ADAT.p, m);
      SDM_HIP_CHECK(hipMemcpyAsync(ada, A.symtmp.p, A.symtmp.n * sizeof(double), hipMemcpyDeviceToDevice, st));
    }
    const int nb = (int)A.sdpN;
    const size_t lds = direct_lds;
    // full columns (a dense ADA' pattern) and no skipped triangle: DIRECT_JB columns per workgroup, constraint i looked up once per row
    const size_t lds_cols = (size_t)DIRECT_JB * (direct_lds + 16) + (size_t)A.sdpN * sizeof(int);
    if (!d_invperm && (int64_t)(P->ada_jc[A.col1] - P->ada_jc[A.col0]) == (int64_t)ncols * m && lds_cols <= S1_DIRECT_LDS_MAX) {
      const int ysplit = std::max(1, std::min(16, (int)((m + 511) / 512)));             // two rows per work-item
      SDM_KLAUNCH(P, k_psd_direct_cols, dim3((ncols + DIRECT_JB - 1) / DIRECT_JB, ysplit), dim3(256), lds_cols, ada, P->absd.p, A.d_ADAjc.p, A.d_Ajc.p, A.d_Ajc_psd.p, A.d_Apr.p,
                  A.d_Air.p, A.d_Ablk.p, A.c_taskptr.p, A.t_blk.p, A.t_n.p, A.t_udoff.p, A.t_slotptr.p, A.s_nzptr.p, A.t_end.p, A.d_psd_start.p, A.udsqr.p,
                  nb, jbase, jbase + ncols, m, direct_zero ? 1 : 0);
      SDM_HIP_CHECK(hipGetLastError());
      return;
    }
    SDM_KLAUNCH(P, k_psd_direct, dim3(ncols), dim3(256), lds, ada, P->absd.p, A.d_ADAjc.p, A.d_ADAir.p, A.d_Ajc.p, A.d_Ajc_psd.p, A.d_Apr.p, A.d_Air.p,
                A.d_Ablk.p, A.c_taskptr.p, A.t_blk.p, A.t_n.p, A.t_udoff.p, A.t_slotptr.p, A.s_nzptr.p, A.t_end.p, A.d_psd_start.p, A.udsqr.p,
                d_invperm, nb, jbase, direct_zero ? 1 : 0);
    SDM_HIP_CHECK(hipGetLastError());
    return;
  }
  bool stage1_cleared_all = false;                                   // the LP / Lorentz part of the panel is zero and stage 1 has written those zeros
  if (ntask > 0) {
    Stage1Tab T;
    T.t_n = A.t_n.p; T.t_nslot = A.t_nslot.p; T.t_ulen = A.t_ulen.p; T.t_herm = A.t_herm.p; T.s_col = A.s_col.p;
    T.u_pos = A.u_pos.p; T.u_rc = A.u_rc.p; T.Air = A.d_Air.p; T.t_slotptr = A.t_slotptr.p; T.t_udoff = A.t_udoff.p; T.t_uoff = A.t_uoff.p;
    T.t_zoff = A.t_zoff.p; T.t_end = A.t_end.p; T.s_nzptr = A.s_nzptr.p; T.Apr = A.d_Apr.p; T.t_blk = A.t_blk.p;
    T.psd_start = A.d_psd_start.p;
#ifndef SDM_EMU
    SDM_HIP_CHECK(hipFuncSetAttribute((const void *)k_psd_stage1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)A.stage1_lds));
#endif
    if (A.maxn <= S1_MAXN && A.sdpN == A.rsdpN) {
      const int np = (A.maxn + 15) & ~15;
      const size_t lds = (size_t)std::max(2 * S1_KC * np, np * np) * sizeof(double);
#ifndef SDM_EMU
      if (lds > 48 * 1024) SDM_HIP_CHECK(hipFuncSetAttribute((const void *)k_psd_stage1_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
#endif
      double *zp = A.zero_ptr; const long long zn = A.zero_n;         // (taken over from ada_lq: see ada_zero_flush)
      A.zero_ptr = nullptr; A.zero_n = 0;
      stage1_cleared_all = zp == ada + P->ada_jc[A.col0] && zn == (long long)(P->ada_jc[A.col1] - P->ada_jc[A.col0]);
      SDM_KLAUNCH(P, k_psd_stage1_mfma, dim3((unsigned)ntask), dim3(64 * S1_WAVES), lds, T, A.udsqr.p, A.zbuf.p, task0,
                  (ntask == (int)A.ntask) ? (const int *)A.t_order.p : (const int *)nullptr, zp, zn);
    } else
    {
      // LDS per task: at least one slot (Y and D row, x2 for Hermitian); beyond that S1_GEN_LDS -- several tasks per CU
      // hide each other's latencies better than one task with all its slots resident
      const size_t one = (size_t)(A.sdpN > A.rsdpN ? 4 : 2) * (size_t)A.maxn * sizeof(double);
      const size_t ldsy = std::max(one, std::min(A.stage1_lds, (size_t)S1_GEN_LDS));
      const int nzcap = (int)std::min<int64_t>(S1_NZ, (A.s1_maxnz + 1) & ~(int64_t)1);
      const size_t lds = ldsy + (size_t)nzcap * (sizeof(double) + sizeof(int));
#ifndef SDM_EMU
      SDM_HIP_CHECK(hipFuncSetAttribute((const void *)k_psd_stage1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
#endif
      // every constraint touches at most one PSD block and the pattern is swept one entry per work-item: stage 2 rides in the task
      const bool ride = A.one_task_per_col && A.thread_per_row && !A.ell_ok && A.sdpN == A.rsdpN && A.s1_maxulen <= 6 * 512 &&
                        (int64_t)A.s1_maxulen * (int64_t)sizeof(double) <= (int64_t)ldsy;
      Stage2Ride R2 = {};
      if (ride) {
        if (sym_input) {
          SDM_KLAUNCH(P, k_symmetrize, dim3(m), dim3(128), 0, A.symtmp.p, ada, A.d_ADAjc.p, A.d_ADAir.p, A.d_ADAT.p, m);
          SDM_HIP_CHECK(hipMemcpyAsync(ada, A.symtmp.p, A.symtmp.n * sizeof(double), hipMemcpyDeviceToDevice, st));
        }
        SDM_HIP_CHECK(hipMemsetAsync(P->absd.p + jbase, 0, (size_t)ncols * sizeof(double), st));      // (constraints without PSD nonzeros: absd = 0)
        R2.ada = ada; R2.absd = P->absd.p; R2.ADAjc = A.d_ADAjc.p; R2.Ajc = A.d_Ajc.p; R2.Ajc_psd = A.d_Ajc_psd.p; R2.ADAir = A.d_ADAir.p;
        R2.Ablk = A.d_Ablk.p; R2.Aupos = A.d_Aupos.p; R2.t_col = A.t_col.p; R2.invperm = d_invperm; R2.Apr = A.d_Apr.p;
      }
      SDM_KLAUNCH(P, k_psd_stage1, dim3((unsigned)ntask), dim3(512), lds, T, A.udsqr.p, A.zbuf.p, (int)(ldsy / sizeof(double)), task0, nzcap, R2,
                  (ntask == (int)A.ntask && A.t_order_xcd.n == (size_t)ntask) ? (const int *)A.t_order_xcd.p : (const int *)nullptr);
      if (ride) { SDM_HIP_CHECK(hipGetLastError()); return; }
    }
  }
  // the reference first adds the PSD part on one triangle and symmetrises at the very end; summing the
  // transposed partial sums of getada1/2 first and adding the (symmetric) PSD part afterwards is the same sum.
  if (sym_input) {
    SDM_KLAUNCH(P, k_symmetrize, dim3(m), dim3(128), 0, A.symtmp.p, ada, A.d_ADAjc.p, A.d_ADAir.p, A.d_ADAT.p, m);
    SDM_HIP_CHECK(hipMemcpyAsync(ada, A.symtmp.p, A.symtmp.n * sizeof(double), hipMemcpyDeviceToDevice, st));
  }
  if (A.ell_ok) {
    auto lds_of = [&](int jb) { return (size_t)jb * (size_t)A.zmax * sizeof(double); };
#define SDM_STAGE2_ELL(JB)                                                                                              \
    do {                                                                                                                 \
      const size_t lds = lds_of(JB);                                                                                    \
      SDM_STAGE2_ATTR(JB, lds);                                                                                         \
      SDM_KLAUNCH(P, k_psd_stage2_ell<JB>, dim3((ncols + JB - 1) / JB, gsplit), dim3(64 * ELL_WAVES), lds, ada, P->absd.p, A.d_ADAjc.p, A.d_ADAir.p, \
                  A.d_Ajc.p, A.d_Ajc_psd.p, A.d_Apr.p, A.d_Ablk.p, A.d_Aupos.p, A.c_taskptr.p, A.t_blk.p, A.t_ulen.p,    \
                  A.t_zoff.p, A.zbuf.p, A.d_uoff.p, A.g_row.p, A.g_len.p, A.g_off.p, A.g_val.p, A.g_bu.p, A.ell_ng, d_invperm, \
                  (int)A.zmax, m, jbase, jbase + ncols, A.ell_pos.p, sym ? (const int *)A.ell_order.p : (const int *)nullptr,     \
                  A.d_Azpos.p, A.t_zdst.p, stage1_cleared_all && !sym_input ? 1 : 0,                                             \
                  sym ? (const long long *)A.cdp64.p : (const long long *)A.cdc64.p, sym ? (const int *)A.cdp32.p : (const int *)A.cdc32.p); \
    } while (0)
#ifndef SDM_EMU
#define SDM_STAGE2_ATTR(JB, lds) if ((lds) > 48 * 1024) SDM_HIP_CHECK(hipFuncSetAttribute((const void *)k_psd_stage2_ell<JB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds)))
#else
#define SDM_STAGE2_ATTR(JB, lds) (void)(lds)
#endif
    // as many columns per workgroup as fit 96 KB of LDS (each coalesced load of the ELL copy then feeds JB columns)
    // The row groups of a column set (distinct rows = distinct output entries, nothing to combine) can be dealt to
    // several workgroups (gridDim.y).  Measured on the bench workload with 3: 65 -> 62 us, but 2.3x the HBM traffic
    // (z_j staged once per workgroup) -- the sweep is bound by its LDS gathers and L2 re-reads, not by the chain of
    // groups inside one workgroup.  Off by default.
    // symmetric half-sweep: the whole of a full-pattern ADA' is being formed and nothing restricts the entries touched
    const bool sym = A.ell_full && !d_invperm && jbase == 0 && ncols == m;
    const int gsplit = sym ? 3 : 1;                                     // half-sweep: the long sweeps (early groups) split in three (measured 1: 52.7, 2: 41.6, 3: 38.9, 4: 41.2 us)
    if (lds_of(4) <= 64 * 1024 && m >= 1024) SDM_STAGE2_ELL(4);
    else if (lds_of(2) <= 64 * 1024 && m >= 512) SDM_STAGE2_ELL(2);
    else SDM_STAGE2_ELL(1);
#undef SDM_STAGE2_ELL
#undef SDM_STAGE2_ATTR
    SDM_HIP_CHECK(hipGetLastError());
    return;
  }
  {
    // z_j in LDS for the one-entry-per-work-item variant when the longest z_j fits 64 KB beside the block table
    const int64_t zl = (A.thread_per_row && A.zmaxj * 8 <= 64 * 1024) ? A.zmaxj : 0;
    const size_t lds = (size_t)A.sdpN * 8 + (size_t)zl * 8;
#ifndef SDM_EMU
    if (lds > 48 * 1024) SDM_HIP_CHECK(hipFuncSetAttribute((const void *)k_psd_stage2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
#endif
    SDM_KLAUNCH(P, k_psd_stage2, dim3(ncols), dim3(256), lds, ada, P->absd.p, A.d_ADAjc.p, A.d_ADAir.p, A.d_Ajc.p,
               A.d_Ajc_psd.p, A.d_Apr.p, A.d_Ablk.p, A.d_Aupos.p, A.c_taskptr.p, A.t_blk.p, A.t_zoff.p, A.c_zlen.p, A.zbuf.p,
               d_invperm, (int)A.sdpN, A.thread_per_row ? 1 : 0, jbase, (int)zl);
  }
  SDM_HIP_CHECK(hipGetLastError());
}

}  // namespace sdm

#if defined(SDM_PHASES) && !defined(SDM_EMU)
// tools-only build (python -m sedumi_amd.build --phases): read / reset the in-kernel phase clocks of this file
extern "C" int sdm_debug_phases_ada(unsigned long long *out32, int reset) {
  if (out32 && hipMemcpyFromSymbol(out32, HIP_SYMBOL(sdm_phase_acc), 32 * sizeof(unsigned long long)) != hipSuccess) return 1;
  if (reset) { unsigned long long z[32] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(sdm_phase_acc), z, sizeof(z)) != hipSuccess) return 1; }
  return 0;
}
#endif

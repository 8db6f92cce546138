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
#include "sdm_follow.h"
#include <cstring>
#include <algorithm>
#include <cmath>
#include <numeric>

namespace sdm {

#ifndef SDM_EMU
struct FollowArgs { double *S, *STr; unsigned long long *sb_g; int nfront; };   // (k_ldl_front: the follower's workgroups behind the front's)
__global__ void k_ldl_front(double *F, double *DT, FrontTab tab, const int *list, double *d, double *lb, const double *ubp, int *pivstat,
                            double *pivval, const PanelCtx *ctx, int *front_cnt, int *diag_cnt, int phase, int step, int tile_wg0, int *tmo, FollowArgs fa);
#else
struct FollowArgs { double *S, *STr; unsigned long long *sb_g; int nfront; };
#endif
// ============================================================ schedule of the trailing updates of a big front (host + device)
// Panel p's rank-64 update of the trailing matrix, applied in the launch after it (the "eager" schedule), is a read-modify-write of every
// trailing tile per panel.  The panels are therefore taken UPD_G at a time (a GROUP g = panels G g .. G g + G - 1):
//   * tile columns up to G g + 2 G - 1 (those factored before the group's deferred update can have reached them) get every panel of the
//     group eagerly, as before: K = 64 in the launch after the panel;
//   * tile columns from G g + 2 G on get the whole group in ONE read-modify-write with K = 64 G once its last panel is final: columns
//     G g + 2 G .. G g + 3 G - 1 (the next group's eager window) in launch G g + G, the rest dealt over the launches G g + G .. G g + 2 G - 1.
// Every tile still receives the panels in ascending order, each as  c <- c - (product over the panel's 64 columns, accumulated from zero):
// the same operations in the same order as the eager schedule, i.e. the SAME BITS; only the trips of c through memory are saved.
// A group is deferred only if all its launches exist and have row-solve workgroups (NP >= G g + 2 G panels, T >= G g + 2 G + 4 tile rows).
// The unit of work is a MACRO TILE of 2 x 2 tiles (128 x 128; panel_role_tiles_stream says why): a region = the tiles (I, J) with
// J0 <= J < J0 + JW, J <= I < nt, cut into macro tiles from (J0, J0) on; tiles of a macro tile outside the region are masked.
constexpr int UPD_G = 2;
__host__ __device__ inline bool upd_group_deferred(int ns, int ms, int g) {
  return g >= 0 && (ns + NB - 1) / NB >= UPD_G * g + 2 * UPD_G && (ms + TILE - 1) / TILE >= UPD_G * g + 2 * UPD_G + 4;
}
// macro tiles of the region (nt, J0, JW): macro columns MJ < MC, macro rows MJ <= MI < MR
__host__ __device__ inline int macro_count(int nt, int J0, int JW) {
  const int R = nt - J0;
  if (R <= 0 || JW <= 0) return 0;
  const int C = JW < R ? JW : R, MC = (C + 1) / 2, MR = (R + 1) / 2;
  return MC * MR - MC * (MC - 1) / 2;
}
__host__ __device__ inline void macro_index(int t, int nt, int J0, int JW, int &MI, int &MJ) {   // column by column
  const int R = nt - J0, C = JW < R ? JW : R, MC = (C + 1) / 2, MR = (R + 1) / 2;
  MJ = 0;
  while (MJ + 1 < MC && t >= MR - MJ) { t -= MR - MJ; MJ++; }
  MI = MJ + t;
}
// of N items, those dealt to launch r of the group's UPD_G launches: t % 10 in [cut[r], cut[r+1])  (the first launch carries the
// group's first columns as well and gets less)
__host__ __device__ inline void share_range(int r, int &lo, int &hi) {
  lo = r == 0 ? 0 : 3 + (r - 1) * 7 / (UPD_G - 1 > 0 ? UPD_G - 1 : 1);
  hi = r == UPD_G - 1 ? 10 : 3 + r * 7 / (UPD_G - 1 > 0 ? UPD_G - 1 : 1);
  if (UPD_G == 1) { lo = 0; hi = 10; }
}
__host__ __device__ inline int share_count(int N, int r) {
  int lo, hi; share_range(r, lo, hi);
  const int rem = N % 10 - lo;
  return (N / 10) * (hi - lo) + (rem < 0 ? 0 : (rem > hi - lo ? hi - lo : rem));
}
__host__ __device__ inline int share_item(int k, int r) { int lo, hi; share_range(r, lo, hi); return 10 * (k / (hi - lo)) + lo + k % (hi - lo); }
// what the tile workgroups of launch q (the launch that factors panel q) of a front do, in macro tiles: NE of the eager region (panel q-1
// into the columns 1 .. JE relative to tile column q; column 0 is the row-solve workgroups'), NH + NR of the deferred group g2 (in its
// first launch the columns G .. 2G-1, and this launch's share of the triangle beyond them -- tile column G g2 + 3 G of the front = column
// 2 G - r relative to this launch; `all`: the whole triangle at once, when the next group is not deferred and its eager updates would
// otherwise meet these tiles in the launches to come)
struct TileSched { int nt, JE, NE, NH, NR, g2, r, all; };
__host__ __device__ inline TileSched tile_sched(int ns, int ms, int q) {
  TileSched S;
  S.nt = (ms - q * NB + TILE - 1) / TILE;
  const int g = (q - 1) / UPD_G;
  S.JE = upd_group_deferred(ns, ms, g) ? min(UPD_G * g + 2 * UPD_G - 1 - q, S.nt - 1) : S.nt - 1;
  S.NE = macro_count(S.nt, 1, S.JE);
  S.g2 = q >= UPD_G ? q / UPD_G - 1 : -1; S.r = q % UPD_G; S.NH = 0; S.NR = 0; S.all = 0;
  if (upd_group_deferred(ns, ms, S.g2)) {
    const int J0 = 2 * UPD_G - S.r, N = macro_count(S.nt, J0, S.nt);   // (the same triangle in all the group's launches: from tile column G g2 + 3 G)
    S.all = upd_group_deferred(ns, ms, S.g2 + 1) ? 0 : 1;
    if (S.r == 0) S.NH = macro_count(S.nt, UPD_G, UPD_G);
    S.NR = S.all ? (S.r == 0 ? N : 0) : share_count(N, S.r);
  } else S.g2 = -1;
  return S;
}
__host__ __device__ inline int tile_sched_items(const TileSched &S) { return S.NE + S.NH + S.NR; }
// item u of the launch's schedule: first tile (I, J) of its macro tile (relative to tile column q), which of its 2 x 2 tiles are the
// item's (bit 2a+b: tile (I+a, J+b)), first panel and number of panels it applies
__host__ __device__ inline void tile_sched_item(const TileSched &sc, int q, int u, int &I, int &J, int &act, int &p0, int &np) {
  int J0, JW, x = u;
  if (u < sc.NE) { J0 = 1; JW = sc.JE; np = 1; p0 = q - 1; }                                                  // eager: the panel before
  else {
    np = UPD_G; p0 = UPD_G * sc.g2;
    if (u < sc.NE + sc.NH) { x = u - sc.NE; J0 = UPD_G; JW = UPD_G; }                                        // the deferred group's first columns
    else { x = u - sc.NE - sc.NH; if (!sc.all) x = share_item(x, sc.r); J0 = 2 * UPD_G - sc.r; JW = sc.nt; }  // this launch's share of the triangle beyond them
  }
  int MI, MJ;
  macro_index(x, sc.nt, J0, JW, MI, MJ);
  I = J0 + 2 * MI; J = J0 + 2 * MJ;
  const int Jend = min(J0 + JW, sc.nt);
  act = 0;
  for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) if (I + a < sc.nt && J + b < Jend && I + a >= J + b) act |= 1 << (2 * a + b);
}

// ============================================================ block-column-cyclic ownership (several ranks factor ONE dense front: sedumi_amd.dist.BlockCyclicFactor)
// `own` = world | rank << 8 | blk << 16 (0: the plan owns everything).  Tile column c of the front belongs to rank (c / blk) % world.  The owner of
// tile column q factors panel q (diagonal block + row solves of launch q); EVERY update of a tile is applied by the owner of the tile's column, in
// the launch the single-plan schedule applies it in: per tile the same operations in the same order, i.e. the same bits (blkchol2.c:346-420 applied
// column by column; the relink rule of blkchol2.c:550-554 becomes "panel q goes to everybody once it is final": the caller broadcasts it).
__host__ __device__ inline bool owns_col(int own, int c) {
  const int world = own & 255;
  if (world <= 1) return true;
  const int blk = own >> 16;
  return (c / (blk > 0 ? blk : 1)) % world == ((own >> 8) & 255);
}

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
  if (!C.sn_active.empty() && (sdm_int)C.sn_active.size() != nsuper) throw std::runtime_error("active-supernode mask does not match the supernode partition");
  auto active = [&](sdm_int s) { return C.sn_active.empty() || C.sn_active[s] != 0; };
  C.levptr.assign(nlev + 1, 0);
  for (sdm_int s = 0; s < nsuper; s++) if (active(s)) C.levptr[C.sn_level[s] + 1]++;
  for (int l = 0; l < nlev; l++) C.levptr[l + 1] += C.levptr[l];
  C.levlist.resize(C.levptr[nlev]);
  { std::vector<int> pos(C.levptr.begin(), C.levptr.end() - 1);
    for (sdm_int s = 0; s < nsuper; s++) if (active(s)) C.levlist[pos[C.sn_level[s]]++] = (int)s; }
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
  int tile_wg_cap = C.tile_wgs_req;                                  // sdm_plan_set_tile_workgroups: tests (a small number makes every workgroup loop)
  if (tile_wg_cap <= 0) {
#ifdef SDM_EMU
    tile_wg_cap = 1 << 20;
#else
    SDM_HIP_CHECK(hipDeviceGetAttribute(&tile_wg_cap, hipDeviceAttributeMultiprocessorCount, P->device));
#endif
  }
  C.launches.clear(); C.lev_first_launch.assign(nlev + 1, 0); C.lev_T.assign(nlev, 1);
  for (int l = 0; l < nlev; l++) {
    C.lev_first_launch[l] = (int)C.launches.size();
    int b = C.levptr[l], e = C.levptr[l + 1];
    if (b == e) continue;                                            // (no active supernode on this level)
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
          tw = nrw > 0 ? tile_sched_items(tile_sched(C.sn_ns[s], C.sn_ms[s], p)) : (ntp * (ntp + 1) / 2 - 1 + 1) / 2;
          // big fronts: the tiles are dealt to as many workgroups as the device holds beside the chain and the row solves (one workgroup
          // per compute unit at this launch's LDS footprint, a few compute units left free: a workgroup that finds none starts when the
          // first one has finished); each works through its tiles as a pipeline (panel_role_tiles_stream)
          if (nrw > 0) tw = std::min(tw, std::max(16, (tile_wg_cap - (C.tile_wgs_req > 0 ? 0 : 8)) / (e - b) - 1 - nrw));
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
  // levels whose fronts are all of the k_ldl_front kind: blocked row solves (MFMA_MIN_ROWS rule of panel_rows), at most
  // FRONT_MAXT tile rows, no partial last panel with rows below it, and few enough workgroups to be resident together
  C.lev_persist.assign(nlev, 0); C.lev_maxT.assign(nlev, 0); C.lev_ntw.assign(nlev, 0);
  std::vector<int> fslot(std::max<sdm_int>(1, C.nsuper), 0);
  int nslot = 0;
  {
    // sdm_plan_set_one_launch_fronts(p, 0): the comparison switch of tests and tools; front_disabled: a launch of this plan timed out
    // before (chol_wait_timeouts) -- the plan stays on the launch-per-panel path across later set_chol calls too, and with no
    // one-launch level follow_decide (sdm_solve.hip) plans no inverse behind the factor either
    const bool off = C.front_off_req || C.front_disabled;
    const int maxT_allowed = FRONT_MAXT;
    // k_ldl_front's workgroups wait for each other in both directions (a row workgroup for its tile workgroups and vice
    // versa): they must all be resident, one per compute unit (135 KB of LDS each).  A device -- or a partition of one --
    // with fewer compute units than the level needs keeps the launch-per-panel path.
    int ncu = 0;
#ifdef SDM_EMU
    ncu = 1 << 20;
#else
    SDM_HIP_CHECK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, P->device));
    {
      // the launch is sized from what the runtime says fits: workgroups of k_ldl_front per compute unit at its LDS and register
      // footprint (1 on gfx950: 135 KB of LDS) -- 0 means the kernel cannot be resident on this device at all (a partition with
      // less LDS, a debugger's reservation): no level takes the one-launch path then
      int per_cu = 0;
      SDM_HIP_CHECK(hipFuncSetAttribute((const void *)k_ldl_front, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FRONT_LDS));
      SDM_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k_ldl_front, LDL_THREADS, FRONT_LDS));
      if (per_cu < 1) ncu = 0;
    }
#endif
    const int wg_budget = std::min(224, ncu - ncu / 8);                // leave an eighth of the device to whatever else is running
    for (int l = 0; l < nlev; l++) {
      bool ok = !off;
      int maxT = 0;
      const int nfr = C.levptr[l + 1] - C.levptr[l];
      for (int i = C.levptr[l]; i < C.levptr[l + 1] && ok; i++) {
        const int s = C.levlist[i], ns = C.sn_ns[s], ms = C.sn_ms[s], T = (ms + TILE - 1) / TILE;
        if (ms - std::min(NB, ns) < MFMA_MIN_ROWS || T > maxT_allowed || (ns % NB != 0 && ms != ns)) ok = false;
        maxT = std::max(maxT, T);
      }
      if (!ok || nfr == 0) continue;
      // one row workgroup per tile row and ONE tile workgroup per tile (r, c), c >= 2: only levels that fit the device that way
      // qualify (single fronts of up to 21 tile rows on a whole MI355X).  Tile workgroups that own several tiles were built and
      // measured in round 2 (MAXCUT-4000's front, 63 tile rows: 5.56 ms against 2.26 ms for the 63 panel launches -- a tile
      // update costs ~20 us of a 147 KB workgroup, so with several tiles each they fall far behind the chain) and removed.
      const int pool = FRONT_POOL;
      const int ntiles = (maxT - 1) * (maxT - 2) / 2;
      const int ntw = std::min(ntiles, wg_budget / nfr - maxT);
      if (ntw < 0 || (int64_t)ntw * pool < ntiles) continue;
      C.lev_persist[l] = 1; C.lev_maxT[l] = maxT; C.lev_ntw[l] = ntw;
      for (int i = C.levptr[l]; i < C.levptr[l + 1]; i++) fslot[C.levlist[i]] = nslot++;
    }
  }
  C.front_cnt.alloc((size_t)std::max(1, nslot) * FRONT_CNT);
  C.d_fslot.upload(fslot);
  // upload
  C.d_first.upload(C.sn_first); C.d_ns.upload(C.sn_ns); C.d_ms.upload(C.sn_ms); C.d_ld.upload(C.sn_ld); C.d_parent.upload(C.sn_parent);
  C.d_childptr.upload(C.childptr); C.d_childlist.upload(C.childlist); C.d_levlist.upload(C.levlist);
  C.d_lindx.upload(lindx); C.d_relidx.upload(relidx);
  { std::vector<int> p32(m); for (sdm_int i = 0; i < m; i++) p32[i] = (int)perm[i]; C.d_perm.upload(p32); }
  C.d_foff.upload(C.sn_foff); C.d_xl.upload(C.sn_xl); C.d_woff.upload(C.sn_woff); C.d_roff.upload(C.sn_roff);
  if (C.fsize <= ASM_FULL_MAX) {                                    // inverse map for k_assemble_full
    std::vector<int> fsrc((size_t)C.fsize, -1);
    for (sdm_int t = 0; t < C.nnzL; t++) fsrc[(size_t)asm_dst[t]] = asm_src[t];
    C.d_asm_fsrc.upload(fsrc);
  } else C.d_asm_fsrc.release();
  C.d_asm_src.upload(asm_src); C.d_asm_dst.upload(asm_dst); C.d_asm_dstT.upload(asm_dstT); C.d_toff.upload(C.sn_toff);
  C.frontsT.alloc((size_t)C.tsize);
  { std::vector<int64_t> l64(C.Ljc.begin(), C.Ljc.end()); C.d_Ljc.upload(l64); }
  C.fronts.alloc((size_t)C.fsize + 128);                          // + padding: k_sinv128 reads up to 63 rows past a partial block
  C.wvec.alloc((size_t)C.wsize); C.colbuf.alloc((size_t)C.wsize + (size_t)nsuper);
  C.d.alloc(m); C.dsolve.alloc(m); C.lb.alloc(m); C.pivval.alloc(m); C.pivstat.alloc(m); C.ub.alloc(3); C.upd_cnt.alloc((size_t)std::max<sdm_int>(1, C.nsuper)); C.diag_cnt.alloc((size_t)std::max<sdm_int>(1, C.nsuper));
  P->ada_val.alloc((size_t)C.nnzADA); P->absd.alloc(m); P->lpr.alloc((size_t)C.nnzL);
  P->rhs.alloc(m); P->y.alloc(m); P->ywork.alloc(m);
  P->has_chol = true; P->factored = false;
  solve_build(P);
}

// ================================================================= kernels

// ---- permuteP: scatter tril(ADA(perm,perm)) into the (zeroed) fronts
// the same with the zero fill of the fronts folded in (arenas of up to ASM_FULL_MAX entries): one work-item per ENTRY of
// the arena through the inverse map (entry -> ADA value index, -1 = structural zero / padding), one launch less per
// factorisation than memset + scatter
__global__ void k_assemble_full(double *F, const double *ada, const int *fsrc, int64_t fsize, double *ub) {
  if (ub && blockIdx.x == 0 && threadIdx.x < 3) ub[threadIdx.x] = 0.0;
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; t < fsize; t += stride) { const int s = fsrc[t]; F[t] = s < 0 ? 0.0 : ada[s]; }
}
__global__ void k_assemble(double *F, const double *ada, const int *src, const int64_t *dst, int64_t nnzL, double *ub) {
  // ub[0..2] (pivot thresholds of this factorisation, filled by k_prep_pivots -- the next launch on the stream) start at 0
  if (ub && blockIdx.x == 0 && threadIdx.x < 3) ub[threadIdx.x] = 0.0;
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

constexpr unsigned long long DT_SENTINEL = 0x7ff8dead5ed00001ull;     // what DT holds until a diagonal block is published (diag_group_fetch): a quiet NaN no computation produces
// ---- pivot thresholds (blkchol.c:168-184), grid-stride over the columns.
//   ub = max_j P(perm_j,perm_j) / maxu^2 ;  lb_j = max(abstol, canceltol * orgd_j)
// ub[2] collects max_j as the bit pattern of a non-negative double (ordered like the unsigned integer: atomicMax is
// exact and order independent); ub[1] = maxu; k_ldl_panel forms ub from them.  ub[2] is zeroed by the host before.
struct PrepArgs {
  int m; const double *ada; const int *asm_src; const int64_t *Ljc; const int *perm; const double *absd; int use_absd;
  double canceltol, maxu, abstol; double *lb, *ub; int *pivstat; double *pivval; int nsuper; int *upd_cnt, *diag_cnt;
  unsigned long long *sb_g; int nsbg; int *front_cnt; int nfc; unsigned long long *DTbits; int64_t ndt;
};
// the work of workgroup `bx` of `nb`: re-arms counters / tags, thresholds of its columns; returns the largest diagonal entry it saw
__device__ __forceinline__ double prep_pivots_part(const PrepArgs &A, int bx, int nb, double *red) {
  const int gid = bx * blockDim.x + threadIdx.x, gstride = nb * blockDim.x;
  for (int64_t i = gid; i < A.ndt; i += gstride) A.DTbits[i] = DT_SENTINEL;             // the data-tagged hand-over of the diagonal blocks (k_ldl_front)
  for (int i = gid; i < A.nsuper; i += gstride) { A.upd_cnt[i] = 0; A.diag_cnt[i] = 0; }   // counters of k_ldl_panel
  for (int i = gid; i < A.nfc; i += gstride) A.front_cnt[i] = 0;                        // counters of k_ldl_front
  for (int i = gid; i < A.nsbg; i += gstride) A.sb_g[i] = 0ull;                         // growth records of the solve inverses (sdm_solve.hip)
  double mx = 0.0;
  for (int j = gid; j < A.m; j += gstride) {
    int s = A.asm_src[A.Ljc[j]];
    double dj = s < 0 ? 0.0 : A.ada[s];
    if (dj > mx) mx = dj;
    double org = A.use_absd ? A.absd[A.perm[j]] : dj;
    double v = A.canceltol * org;
    A.lb[j] = v > A.abstol ? v : A.abstol;
    A.pivstat[j] = 0; A.pivval[j] = 0.0;
  }
  red[threadIdx.x] = mx;
  __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s && red[threadIdx.x + s] > red[threadIdx.x]) red[threadIdx.x] = red[threadIdx.x + s];
    __syncthreads();
  }
  return red[0];
}
__global__ void k_prep_pivots(PrepArgs A) {
  __shared__ double red[256];
  const double mx = prep_pivots_part(A, blockIdx.x, gridDim.x, red);
  if (threadIdx.x == 0) {
    union { double d; unsigned long long u; } b; b.d = mx;
    atomicMax((unsigned long long *)&A.ub[2], b.u);
    A.ub[1] = A.maxu;
  }
}
// k_assemble_full and k_prep_pivots as ONE launch (fronts small enough for the inverse map: the bench workloads): workgroups
// [0, nprep) do the pivot bounds, the rest assemble.  The largest diagonal entry cannot be an atomicMax into a cell that this same
// launch would have to clear first: every bounds workgroup leaves its maximum in `part`, the last one to finish (a ticket that
// resets itself) takes the maximum of them -- order independent, like the atomicMax.
__global__ void k_begin_factor(PrepArgs A, int nprep, double *part, int *ticket, double *F, const int *fsrc, int64_t fsize) {
  if ((int)blockIdx.x >= nprep) {
    int64_t t = (int64_t)((int)blockIdx.x - nprep) * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)((int)gridDim.x - nprep) * blockDim.x;
    for (; t < fsize; t += stride) { const int s = fsrc[t]; F[t] = s < 0 ? 0.0 : A.ada[s]; }
    return;
  }
  __shared__ double red[256];
  __shared__ int last;
  const double mx = prep_pivots_part(A, blockIdx.x, nprep, red);
  // (write-through store, acknowledged, then the ticket; the last workgroup reads with L2-bypassing loads: the workgroups sit on
  // different XCDs -- with plain stores and a fence the soak of round 4 met a stale maximum twice in 3471 fronts)
  if (threadIdx.x == 0) sdm_store_wt(&part[blockIdx.x], mx);
  SDM_STORES_DONE();
  if (threadIdx.x == 0) last = sdm_ticket_take(ticket) == nprep - 1;
  __syncthreads();
  if (!last) return;
  double m2 = 0.0;
  for (int i = threadIdx.x; i < nprep; i += blockDim.x) { const double v = sdm_load_wt(&part[i]); if (v > m2) m2 = v; }
  __syncthreads();
  red[threadIdx.x] = m2;
  __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s && red[threadIdx.x + s] > red[threadIdx.x]) red[threadIdx.x] = red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) { A.ub[0] = 0.0; A.ub[1] = A.maxu; A.ub[2] = red[0]; sdm_signal_add(ticket, -nprep); }   // (the ticket back to 0, where the tickets are counted)
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
                                           int ms, int ld, SDM_GP(const double) Fs_, const double *ds, SDM_GP(double) cb_,
                                           double next_raw_diag, double *red_v, int *red_i) {
  SDM_FP_STRICT;   // no FMA contraction: the pivot decisions must see the reference's mul-then-subtract rounding
  const double *Fs = (const double *)Fs_;
  double *cb = (double *)cb_;
  const int tid = threadIdx.x, bs = LDL_THREADS;              // (blockDim.x inside a called function is two dependent loads from the dispatch packet)
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
                                            double (*As)[UTP], double (*Bs)[UTP], double *dsh,
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
template <bool WT = false>
__device__ __forceinline__ void rows_stage(const double *Fs, int ld, int ms, int k0, int kb, int R0, double *Tw, int lane) {
  const int li = lane & 15, lk = lane >> 4;
  double tv[NB / 4];
  const double *pr = Fs + min(R0 + li, ms - 1);
#pragma unroll
  for (int c4 = 0; c4 < NB / 4; c4++) {                                                                // 16 loads in flight
    const double *a = &pr[(int64_t)(k0 + min(4 * c4 + lk, kb - 1)) * ld];
    tv[c4] = WT ? sdm_load_wt(a) : *a;
  }
#pragma unroll
  for (int c4 = 0; c4 < NB / 4; c4++) { const int c = 4 * c4 + lk; Tw[c * 17 + li] = c < kb ? tv[c4] : 0.0; }
}
// 16-column block b of the blocked substitution on the wave tile, in two halves: the product part needs the columns
// 0 .. 16b-1 of L11 only (rows 16b .. 16b+15 of them), the triangle its columns 16b .. 16b+15 and their pivots -- the
// row-solve workgroups run the first half BEFORE they wait for the publication of the block's own 16 columns
__device__ __forceinline__ void rows_block_gemm(int b, const double (*S)[NB + 1], double *Tw, int lane) {
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
}
__device__ __forceinline__ void rows_block_tri(int b, const double (*S)[NB + 1], const double *ds, double *Tw, int lane) {
  const int li = lane & 15;
  const int cb = 16 * b;
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
__device__ __forceinline__ void rows_block(int b, const double (*S)[NB + 1], const double *ds, double *Tw, int lane) {
  rows_block_gemm(b, S, Tw, lane);
  rows_block_tri(b, S, ds, Tw, lane);
}
// l = x / d out of the wave tile into the front
template <bool WT = false>
__device__ __forceinline__ void rows_store(double *Fs, int ld, int ms, int k0, int kb, int R0, const double *ds, const double *Tw, int lane) {
  const int li = lane & 15, lk = lane >> 4;
  for (int c4 = 0; c4 < NB / 4; c4++) {
    const int c = 4 * c4 + lk, row = R0 + li;
    if (c < kb && row < ms) {
      const double dc = ds[c], xv = Tw[c * 17 + li];
      const double v = dc > 0.0 ? xv / dc : 0.0;
      if (WT) sdm_store_wt(&Fs[(int64_t)(k0 + c) * ld + row], v); else Fs[(int64_t)(k0 + c) * ld + row] = v;
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
// ONLY: 0 both paths compiled in, 1 the blocked (MFMA) path alone, 2 the few-rows path alone (callers that have chosen already)
template <int ONLY = 0>
__device__ __forceinline__ void panel_rows(double *Fs, int ld, int ns, int ms, int k0, int kb, int rbeg, int rend, int brows,
                                           const double (*S)[NB + 1], const double *ds, double *RB, bool staged = false) {
  SDM_FP_STRICT;
  const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6, ny = LDL_THREADS >> 6;
  rend = min(rend, ms);
  if (ONLY != 2 && (ONLY == 1 || ms - min(NB, ns) >= MFMA_MIN_ROWS)) {   // per front, the same path for all its panels
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
// fence = false: the caller reads what it waited for with sdm_load_wt only (no acquire fence needed, 1.7 us less)
__device__ __forceinline__ void spin_until(const int *cnt, int target, int *tmo, bool fence = true) {
  if (threadIdx.x == 0) {
    for (long it = 0; sdm_signal_load(cnt) < target; it++) { if (sdm_spin_giveup(it, tmo)) break; SDM_SPIN_PAUSE(); }
  }
  __syncthreads();
  if (fence) SDM_ACQUIRE_FENCE();
}
__device__ __forceinline__ void wait_prev_update(const int *cnt, int ns, int ms, int panel, int q0, int *tmo) {
  int target = 0;                                              // launches q0 .. panel carried update tiles
  for (int q = max(q0, 1); q <= panel; q++) {
    const int nt = (ms - q * NB + TILE - 1) / TILE, nrw = panel_row_wgs(ns, ms, q);
    target += nrw > 0 ? nrw : (nt * (nt + 1) / 2) / 2;
  }
  spin_until(cnt, target, tmo);
}

// ---- LDL' of the 64-column diagonal block of panel `panel` of front s by ALL work-items of the calling workgroup (the
// header of k_ldl_panel describes the method).  S (= smem) holds the block on entry unless load_block (then it is read
// from the front), Lc = zero.  publish: other workgroups wait for the factored block -- it goes to DT / d 16 columns at
// a time as it becomes final, diag_cnt[s] counts those publications (*npub of the 4 are out on return; the caller
// signals the rest once the write-back below has been acknowledged).  On return: S = unit lower factor (scaled columns),
// ds = pivots (LDS), the block written in place and to DT, d / pivstat / pivval stored.  Returns false when the block
// went through the general path (a pivot asked for the never-fail rule's column probe).
// PERSIST (k_ldl_front): upd_cnt = the front's per-tile-row counters of finished update steps.
// (k_ldl_front) every tile row below `panel` has applied the updates of the panels before it
__device__ __forceinline__ void front_wait_updates(const int *upd_done, int panel, int T, int *tmo) {
  for (int r = panel + 1; r < T; r++) spin_until(upd_done + r, panel, tmo);
  // the probe of the block's last column also looks at the first diagonal entry of the next block (what lies behind the column in
  // L's storage): that tile's updates q <= panel - 1 are its tile workgroup's, counted in tile_cnt (behind upd_done)
  if (panel + 1 < T && panel + 1 >= 2) spin_until(upd_done + FRONT_MAXT + (panel + 1) * FRONT_MAXT + panel + 1, panel, tmo);
}
// ---- the two inner pieces of the diagonal block's LDL' (ldl_diag_block describes the method; k_ldl_front's chain
// workgroup runs the same pieces with a different cast of wavefronts).
// Wavefront 0, one sweep: sweep s (columns c0 = s*SW ..) is final and sits in xs (unscaled); the next SW columns cn .. are
// brought up to date with it (look-ahead), swept in registers (lane = row; pivots and multipliers by v_readlane), written
// back to S / Lc, and the bookkeeping of their pivots is done in the pivots' own lanes.
__device__ __forceinline__ void diag_sweep_w0(double (*S)[NB + 1], double *Lc, int s, double (&xs)[SW], double mylb, int tx, int kb, int k0, int ms,
                                              double ub, double *ds, int *stt, double *pv, int *badflag_p) {
  SDM_FP_STRICT;
  int &badflag = *badflag_p;
      const int c0 = s * SW, cn = c0 + SW;                             // sweep s is final; sweep columns cn .. cn+SW-1 now
      double x[SW], lsc[SW];
      SDM_PHASE_BEGIN();
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
      SDM_PHASE(6);
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
      SDM_PHASE(7);
      // rows above the diagonal carry don't-care values from here on (nobody reads them: every consumer of S and Lc
      // is restricted to the lower triangle), which saves the masks
#pragma unroll
      for (int k = 0; k < SW; k++) {
        Lc[(cn + k) * NB + tx] = lsc[k];
        S[tx][cn + k] = x[k];
        xs[k] = x[k];
      }
      if (tx >= cn && tx < cn + SW && tx < kb) {                       // bookkeeping of pivot tx in lane tx (its own register copy of x_tt)
        double pval = x[0];
#pragma unroll
        for (int k = 1; k < SW; k++) pval = (tx == cn + k) ? x[k] : pval;
        const bool acc = pval > mylb;
        ds[tx] = acc ? pval : 0.0;
        if (!acc) { stt[tx] = 1; pv[tx] = pval; }
        if (acc && ms - (k0 + tx) > 1 && pval < ub) badflag = 1;       // needs the column probe: general path below
      }
      SDM_PHASE(8);
}
// One of nw helper wavefronts (widx = 0 .. nw-1), one sweep: sweep s goes into the trailing columns from c0 + 2 SW on
// (x_rj -= l_jk * x_rk, k ascending), 4 columns per wavefront at a time.
__device__ __forceinline__ void diag_trail(double (*S)[NB + 1], const double *Lc, int s, int kb, int tx, int widx, int nw) {
  SDM_FP_STRICT;
      const int c0 = s * SW;
      double xk[SW];
#pragma unroll
      for (int k = 0; k < SW; k++) xk[k] = S[tx][c0 + k];
      for (int j0 = c0 + 2 * SW + 4 * widx; j0 < kb; j0 += 4 * nw) {   // 4 columns per wavefront at a time
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
}

// columns 16 g .. 16 g + 15 of the factored block into its transposed copy DT, one wavefront, lane = row: a row's 16 entries are
// contiguous there (128 bytes), so they go out as eight 16-byte write-through stores -- full fabric writes -- instead of one
// 8-byte write per lane and column (the publication lagged the sweeps by 4-5 us per group that way: profiles/r03k).  The pivot
// travels in the diagonal slot; what lies above the diagonal is not read by anybody (zeros).
__device__ __forceinline__ void publish_group(double *Dsp, const double *Lc, const double *ds, int g, int tx) {
  if (tx < 16 * g) return;
  double v[16];
#pragma unroll
  for (int c = 0; c < 16; c++) {
    const int j = 16 * g + c;
    v[c] = tx > j ? Lc[j * NB + tx] : (tx == j ? ds[j] : 0.0);
  }
#if defined(SDM_PUB8)
#pragma unroll
  for (int c = 0; c < 16; c++) sdm_store_wt(&Dsp[tx * NB + 16 * g + c], v[c]);
#else
#pragma unroll
  for (int p2 = 0; p2 < 8; p2++) sdm_store_wt2(&Dsp[tx * NB + 16 * g + 2 * p2], v[2 * p2], v[2 * p2 + 1]);
#endif
}
// what ldl_diag_block needs of a front's descriptor, fetched ONCE per workgroup (every read of the tables in HBM is a dependent
// load of a microsecond, and the noinline stages would each repeat them on the chain).  It lives in LDS and is handed on BY
// ADDRESS: a struct passed by value to a called function travels through the stack (scratch memory) behind a pointer -- two
// dependent memory round trips at the top of every diagonal block.  For the same reason the function's other arguments are
// kept to the 32 registers the calling convention has: what only the rare general path needs comes through PanelCtx.
struct FrontDesc { int ns, ms, ld, first; int64_t foff, toff, woff; double maxu, ub; int s, pad; };
template <bool PERSIST>
__device__ __forceinline__ bool ldl_diag_block(char *smem, double *F, double *DT, const FrontDesc &fd, int panel, double *d, double *lb,
                                               int *pivstat, double *pivval, const PanelCtx *ctx, int *upd_cnt, int *diag_cnt, int q0, int *tmo,
                                               bool load_block, bool publish, double *ds, int *npub, bool raw_in_lds = false, int pub_skip = 0,
                                               const double *lbs_pre = nullptr) {
  // ctx: what only the general path reads (probe scratch, the next supernode's raw diagonal); fd.maxu / fd.ub are read behind the first barrier
  // lbs_pre (k_ldl_front): the block's pivot thresholds, fetched into LDS when the workgroup started (one global round trip off the chain)
  // pub_skip (k_ldl_front's chain workgroup redoing a block on the general path): 16-column groups of this block already counted in diag_cnt
  // raw_in_lds (k_ldl_front): the raw block is not in the front but in LDS behind the wave tiles (front_rows_diag)
  SDM_FP_STRICT;   // no FMA contraction: the pivot decisions must see the reference's mul-then-subtract rounding
  double (*S)[NB + 1] = (double (*)[NB + 1])smem;                 // diagonal block, S[row][col]
  double *RB = (double *)smem + NB * (NB + 1);                    // Lc during the LDL', then Xs / the wave tiles of the row solve
  double *Lc = RB;                                                // Lc[k*NB+i] = l_ik
  __shared__ double lbs[NB], pv[NB];
  __shared__ int stt[NB];
  __shared__ int badflag;
  __shared__ double red_v[LDL_THREADS];
  __shared__ int red_i[LDL_THREADS];
  const int ns = fd.ns, ms = fd.ms, ld = fd.ld, first = fd.first;
  const int64_t toff_s = fd.toff;
  const int k0 = panel * NB, kb = min(NB, ns - k0);
  double *Fs = F + fd.foff;
  const int s = fd.s;
  const int tid = threadIdx.x, bs = LDL_THREADS;                    // (both kernels launch LDL_THREADS work-items; blockDim.x inside a called function is two dependent loads)
  const int tx = tid & 63, ty = tid >> 6, ny = bs >> 6;
  if (load_block) {
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
  if (tid < NB) { lbs[tid] = lbs_pre ? lbs_pre[tid] : (tid < kb ? lb[first + k0 + tid] : 0.0); ds[tid] = 0.0; stt[tid] = 0; pv[tid] = 0.0; }
  if (tid == 0) { badflag = 0; *npub = 0; }
  SDM_PHASE_BEGIN();
  __syncthreads();
  const double ub = fd.ub;                                           // max diagonal (k_prep_pivots) / maxu^2; (k_ldl_panel: written just before this block)
  SDM_PHASE(16);
  if (PERSIST) SDM_TRACE(16 * panel + 0);                              // D: sweeps start
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
      diag_sweep_w0(S, Lc, s, xs, mylb, tx, kb, k0, ms, ub, ds, stt, pv, &badflag);
      SDM_PHASE(17);
      __syncthreads();
      SDM_PHASE(19);
    }
    SDM_SETPRIO(0);
    if (PERSIST) SDM_TRACE(16 * panel + 1);                            // D: sweeps end
  } else if (ty < ny - 1) {
    __syncthreads();                                                   // sweep 0
    for (int s = 0; s < nsw - 1; s++) {
      diag_trail(S, Lc, s, kb, tx, ty - 1, ny - 2);
      SDM_PHASE(18);
      __syncthreads();
    }
  } else {
    // ---- the last wavefront publishes the factor as it grows: after every second sweep 16 more columns of L11 (and
    // their pivots) are final; they go to DT / d write-through and, one sweep later (the stores have been acknowledged
    // by then), the count the row-solve workgroups of this launch poll goes up by one.  Nothing is published from a
    // sweep on in which a pivot asked for the probe (the block is redone by the general path; what was published
    // before is what the general path computes again).
    double *Dsp = DT + toff_s + (int64_t)panel * NB * NB;
    int issued = pub_skip, signalled = pub_skip;
    __syncthreads();                                                   // sweep 0
    for (int sw = 0; sw < nsw - 1; sw++) {
      if (publish) {
        // (k_ldl_front: the row workgroups read the data-tagged DT itself; the count is for consumers off the chain -- the follower,
        // the column probe -- and goes up behind the last sweep: no acknowledgement wait inside the sweeps, whose barrier it would hold)
        if (!PERSIST && issued > signalled) {                          // columns stored during the previous sweep
          SDM_STORES_DONE();
          if (tx == 0) sdm_signal_add(&diag_cnt[s]);
          signalled = issued;
        }
        const int g = issued;                                          // sweeps 0 .. sw are final: columns < 8 (sw+1)
        if (SW * (sw + 1) >= 16 * (g + 1) && badflag == 0) {
          publish_group(Dsp, Lc, ds, g, tx);
          if (tx < 16 && 16 * g + tx < kb) sdm_store_wt(&d[first + k0 + 16 * g + tx], ds[16 * g + tx]);
          issued = g + 1;
        }
      }
      __syncthreads();
    }
    // after the last sweep: what is left of the block, right away (the epilogue below would be 2-3 us later)
    if (publish && badflag == 0) {
      for (int g = issued; 16 * g < kb; g++) {
        publish_group(Dsp, Lc, ds, g, tx);
        if (tx < 16 && 16 * g + tx < kb) sdm_store_wt(&d[first + k0 + 16 * g + tx], ds[16 * g + tx]);
        issued = g + 1;
      }
    }
    if (issued > signalled) { SDM_STORES_DONE(); if (tx == 0) sdm_signal_add(&diag_cnt[s], issued - signalled); }
    if (tx == 0) *npub = issued;
  }
  const bool bad = badflag != 0;
  const bool ok = !bad;
  if (!ok) {
    // ---- general path: one column per step by all work-items, pivot_probe available
    if (panel > 0) {                                                 // the probe reads the rows below the block
      if (PERSIST) front_wait_updates(upd_cnt, panel, (ms + TILE - 1) / TILE, tmo);
      else wait_prev_update(upd_cnt + s, ns, ms, panel, q0, tmo);
    }
    for (int j = ty; j < NB; j += ny) {
      const double raw = raw_in_lds ? ((const double *)smem)[FRONT_CV_OFF + j * TILE + tx] : Fs[(int64_t)(k0 + min(j, kb - 1)) * ld + k0 + min(tx, kb - 1)];
      S[tx][j] = (tx < kb && j <= tx) ? raw : 0.0; Lc[j * NB + tx] = 0.0;
    }
    if (tid < NB) { ds[tid] = 0.0; stt[tid] = 0; pv[tid] = 0.0; }
    __syncthreads();
    for (int k = 0; k < kb; k++) {
      double xkk = S[k][k];
      if (xkk > lbs[k]) {
        if (ms - (k0 + k) > 1 && xkk < ub) {                         // rare: stability probe of the never-fail rule
          double nraw = 0.0;
          const double maxu = fd.maxu;
          double *cb = ctx->colbuf + fd.woff + s;                    // probe scratch: ms + 1 doubles per front
          if (k0 + k + 1 >= ns && first + ns < ctx->mtot) { int sidx = ctx->asm_src[ctx->Ljc[first + ns]]; nraw = sidx < 0 ? 0.0 : ctx->ada[sidx]; }
          const double ubk = pivot_probe(S, Lc, k, kb, k0, ns, ms, ld, (SDM_GP(const double))Fs, ds, (SDM_GP(double))cb, nraw, red_v, red_i) / maxu;
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
    double *Ds = DT + toff_s + (int64_t)panel * NB * NB;
    // the factored block goes in place from THIS workgroup in every case: it also stored the raw updated block (tile
    // (0,0) of the previous update), and two workgroups writing the same lines in one launch may sit behind different
    // L2s whose write-back order is not defined
    const bool inplace = true;
    for (int j = ty; j < kb; j += ny)
      if (tx < kb && tx >= j) {
        const double v = (tx == j) ? 1.0 : S[tx][j];                // unit diagonal stored explicitly (blkchol2.c:136)
        if (inplace) Fs[(int64_t)(k0 + j) * ld + k0 + tx] = v;
        sdm_store_wt(&Ds[tx * NB + j], tx == j ? ds[j] : v);        // transposed copy of the block for the row solves; its diagonal slots carry the pivots
      }
    if (tid < kb) {
      const int gk = first + k0 + tid;
      sdm_store_wt(&d[gk], ds[tid]);
      if (stt[tid]) { pivstat[gk] = stt[tid]; pivval[gk] = pv[tid]; }   // pivval = amount added (what blkchol2.c:127 keeps in lb[k])
    }
  }
  return ok;
}

#ifdef SDM_EMU
#define SDM_NOINLINE
#else
// (a callable function does not inherit the kernel's launch bounds: the register allocator budgets it for the translation unit's
// default workgroup size -- 1024 work-items = 128 VGPRs unless the file is compiled with --gpu-max-threads-per-block=512, as
// sedumi_amd/build.py does for this file; with 128 the roles spill inside although their kernels may use 256)
// not_tail_called: a call the optimizer marks as a tail call (it does whenever no stack object of the caller is passed along) makes
// the callee ineligible for the "no callee-saved registers" treatment of internal functions, and it then saves and restores every
// callee-saved VGPR it touches through scratch on each call: 72 of them in the diagonal-block stage, 117 in panel_diag_rows.
#define SDM_NOINLINE __noinline__ __attribute__((not_tail_called))
#endif
#ifndef SDM_NI_ROWS
#define SDM_NI_ROWS __forceinline__       // (measured: +1 us per launch as a call -- while calls still saved callee-saved registers, see SDM_NOINLINE)
#endif
#ifndef SDM_NI_TILES
#define SDM_NI_TILES __forceinline__      // (the throughput role of big fronts: as a call it saved and reloaded 40 callee-saved VGPRs per workgroup, +4 us per launch on MAXCUT-4000 -- before not_tail_called; not measured again)
#endif
// The roles of a k_ldl_panel workgroup are REAL function calls (as in k_ldl_front below): inlined into one body they shared one
// register allocation and the kernel spilled 167 VGPRs (round 2); each role alone fits.
// ---- row-solve workgroup b of panel `panel` (see k_ldl_panel)
__device__ SDM_NI_ROWS SDM_NORETURN void panel_role_rows(char *smem, double *Fs, const double *Ds, double *d, int ns, int ms, int ld, int first, int panel, int b,
                                             int *upd_cnt_s, const int *diag_cnt_s, int phase, int *tmo) {
  SDM_FP_STRICT;
  double (*As)[UTP] = (double (*)[UTP])smem;
  double (*Bs)[UTP] = As + NB;
  __shared__ double dsh[NB];
  const int k0c = panel * NB, kbc = min(NB, ns - k0c);
  const int kp = (panel - 1) * NB;
  const bool mfma_rows = ms - min(NB, ns) >= MFMA_MIN_ROWS;
  if (phase != 2 && panel > 0) {
    if (mfma_rows)                                              // the result doubles as the row solve's wave tiles in LDS
      update_tile<LDL_THREADS / 64, false, true, true>(Fs, ld, ms, first, kp, NB, b + 1, 0, d, As, Bs, dsh, nullptr, nullptr, kbc,
                                                        threadIdx.x, true, (double *)smem + NB * (NB + 1));
    else
      update_tile<LDL_THREADS / 64, false, true>(Fs, ld, ms, first, kp, NB, b + 1, 0, d, As, Bs, dsh);
    SDM_STORES_DONE();
    __syncthreads();
    if (threadIdx.x == 0) sdm_signal_add(upd_cnt_s);
  }
  if (phase == 1) SDM_ENDPGM();
  double (*S)[NB + 1] = (double (*)[NB + 1])smem;
  double *RB = (double *)smem + NB * (NB + 1);
  __shared__ double dsr[NB];
  const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6, ny = LDL_THREADS >> 6;
  const int rbeg = k0c + NB * (b + 1), rend = min(ms, k0c + NB * (b + 2));      // = tile row b+1
  if (!mfma_rows) {
    // few rows: the faithful substitution needs the whole block
    spin_until(diag_cnt_s, 4 * (panel + 1), tmo);                  // (with the fence: panel_rows re-reads this workgroup's own updated rows)
    constexpr int NQ = NB / (LDL_THREADS / 64);
    double sv[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) sv[q] = sdm_load_wt(&Ds[min(ty + ny * q, NB - 1) * NB + tx]);
#pragma unroll
    for (int q = 0; q < NQ; q++) { const int i = ty + ny * q; if (i < NB) S[i][tx] = (i < kbc && tx < i) ? sv[q] : 0.0; }
    if (tid < NB) dsr[tid] = tid < kbc ? sdm_load_wt(&d[first + k0c + tid]) : 0.0;
    __syncthreads();
    panel_rows(Fs, ld, ns, ms, k0c, kbc, rbeg, rend, ROWS_BATCH, S, dsr, RB);
    SDM_ENDPGM();
  }
  // blocked substitution, 16 rows per wavefront (4 of the 8 are busy), following the diagonal block as workgroup 0
  // publishes it 16 columns at a time
  const int R0 = rbeg + 16 * ty;
  const bool busy = R0 < rend;
  double *Tw = RB + ty * (NB * 17);
  if (!(phase == 0 && panel > 0) && busy) rows_stage(Fs, ld, rend, k0c, kbc, R0, Tw, tx);      // else staged by the update above
  for (int blk = 0; blk < NB / 16 && 16 * blk < kbc; blk++) {
    if (busy) rows_block_gemm(blk, S, Tw, tx);                   // needs earlier columns only: off the tail of the launch
    spin_until(diag_cnt_s, 4 * panel + blk + 1, tmo, false);       // columns 16 blk .. of L11 and their pivots are in DT / d (sc1 loads: no fence)
    for (int e = tid; e < NB * 16; e += LDL_THREADS) {
      const int i = e >> 4, j = 16 * blk + (e & 15);
      S[i][j] = (i < kbc && j < i) ? sdm_load_wt(&Ds[i * NB + j]) : 0.0;
    }
    if (tid < 16) dsr[16 * blk + tid] = 16 * blk + tid < kbc ? sdm_load_wt(&d[first + k0c + 16 * blk + tid]) : 0.0;
    __syncthreads();
    if (busy) rows_block_tri(blk, S, dsr, Tw, tx);
  }
  if (busy) rows_store(Fs, ld, rend, k0c, kbc, R0, dsr, Tw, tx);
  SDM_ENDPGM();                                                  // (the kernel has nothing left to do for this workgroup)
}
// ---- tile workgroup w: two tiles of the previous panel's update side by side, 4 wavefronts each
__device__ SDM_NI_TILES void panel_role_tiles(char *smem, double *Fs, const double *d, int ms, int ld, int first, int panel, int w, int nrw, int nt,
                                              int *upd_cnt_s, int own) {
  SDM_FP_STRICT;
  double (*As)[UTP] = (double (*)[UTP])smem;
  double (*Bs)[UTP] = As + NB;
  const int kp = (panel - 1) * NB;
  const int half = threadIdx.x >> 8, u = 2 * w + half;
  int I, J, ntl;
  bool active;
  if (nrw > 0) {                                               // block column 0 belongs to the row-solve workgroups
    ntl = (nt - 1) * nt / 2;
    if (2 * w >= ntl) SDM_ENDPGM();
    active = u < ntl;
    tile_index(active ? u : 0, I, J);
    I++; J++;
  } else {
    ntl = nt * (nt + 1) / 2 - 1;                               // all tiles but (0,0)
    if (2 * w >= ntl) SDM_ENDPGM();
    active = u < ntl;
    tile_index(active ? u + 1 : 1, I, J);
  }
  if (!owns_col(own, panel + J)) active = false;               // (another rank's tile column; the signal below is still given)
  __shared__ double dsh2[2][NB];
  update_tile<4, false, true>(Fs, ld, ms, first, kp, NB, I, J, d, As + half * 2 * NB, Bs + half * 2 * NB, dsh2[half],
                              nullptr, nullptr, 0, (int)threadIdx.x & 255, active);
  if (nrw == 0) {                                              // readers in this launch: workgroup 0's row solve / probe
    SDM_STORES_DONE();
    __syncthreads();
    if (threadIdx.x == 0) sdm_signal_add(upd_cnt_s);
  }
  SDM_ENDPGM();
}
// ---- the same for big fronts (those with row-solve workgroups: nobody in the launch reads these tiles): workgroup w of ntw works
// through the MACRO tiles w, w + ntw, ... of the launch's schedule (tile_sched: eager macro tiles of the panel before, K = 64; deferred
// ones of a whole group of panels, K = 64 UPD_G) as one pipeline over steps of half a chunk (32 columns).
// What shaped it, each step measured on MAXCUT-4000 (profiles/r05_factor_update_variants.txt, r05l / r05m / r05s_phase_tiles_*, r05r_ubench9_*):
//   * one pair of 64 x 64 tiles per workgroup and launch (round 4, panel_role_tiles): ~10 us per pair -- dispatch, a memory round trip,
//     staging, 3.4 us of MFMAs, acknowledged write-through stores, one after the other with nothing else on the compute unit;
//   * the same tiles streamed through persistent workgroups with prefetched operands, one or two LDS buffers, lock step or the wavefront
//     halves out of phase, 8- or 16-byte loads: 4.2 us per tile step whatever the arrangement -- a step fetches 64 KB of operands, and what
//     the workgroups get delivered together is ~3 TB/s (16 GB/s each), not what the matrix cores could take (2.0 us per step);
//   * hence 2 x 2 MACRO tiles: the operand blocks A(I), A(I+1), B(J), B(J+1) of a chunk are fetched once (all four in LDS: the launch's
//     whole LDS footprint) for FOUR tile products -- half the bytes per flop -- and the macro tile is ONE 128 x 128 product, 64 x 32 per
//     wavefront (8 accumulators, 6 operand reads from LDS per 8 MFMAs: the loop runs at the matrix rate, 3.97 us per half chunk against
//     3.95 for the loop alone, tools/ubench/ubench9; as four 64 x 64 products of 32 x 16 per wavefront it ran at 55 %);
//   * steps of HALF a chunk: the 32 prefetch registers more of a whole chunk spilled, and the reloads' s_waitcnt vmcnt(0) in front of
//     the MFMA loop waited for the prefetch itself (16 us per chunk instead of 8); with half chunks the LDS blocks double-buffer in place
//     (MFMAs on one half of their columns while the other half is written) and a step needs one barrier;
//   * a macro tile's own values first, operand loads behind them (the counter of outstanding loads retires in order), plain stores for the
//     finished tile (32 write-through stores took 8 us to issue; nobody reads these tiles before the launch ends);
//   * groups of UPD_G = 2 panels: a deferred macro tile of four panels is 32 us of MFMAs on one compute unit -- more than the 26 us of the
//     chain the tiles are meant to hide behind.
// What is left: a step is 4.0 us of MFMAs + 5.6 us of everything else (waiting for operands and tile values -- a macro tile moves 512 KB for
// its 15.8 us of MFMAs, the workgroups together ask for 3 TB/s --, LDS writes, stores, barrier) that the lock step of a barrier per step
// puts in front of the MFMAs instead of beside them.  The launches from the 15th on are bound by the chain of the diagonal blocks.
// A tile's own values make one trip through the registers per macro tile and group.  Per tile: the same operands, the same instructions on
// the same accumulators, the same order as the eager schedule -- the same bits.
struct TileItem { int I, J, nch, kp0, ds0, act; };               // first tile of the macro tile (relative to tile column q); act: bit 2a+b = tile (I+a, J+b) is the item's
__device__ SDM_NI_TILES void panel_role_tiles_stream(char *smem, double *Fs, const double *d, int ns, int ms, int ld, int first, int panel, int w, int ntw, int own) {
  SDM_FP_STRICT;
  const int tid = threadIdx.x;
  double (*Ab)[UTP] = (double (*)[UTP])smem;                   // A(I) = Ab, A(I+1) = Ab + NB, B(J) = Ab + 2 NB, B(J+1) = Ab + 3 NB   (4 NB UTP doubles = PANEL_LDS_RIDE)
  __shared__ double dshs[1 + UPD_G][NB];                       // pivots: row 0 the panel before, rows 1 .. G the panels of the deferred group
  const TileSched sc = tile_sched(ns, ms, panel);
  const int nitems = tile_sched_items(sc);
  if (w >= nitems) SDM_ENDPGM();
  const int r0 = panel * NB;
  for (int e = tid; e < (1 + UPD_G) * NB; e += LDL_THREADS) {
    const int row = e / NB, k = e % NB;
    dshs[row][k] = row == 0 ? d[first + (panel - 1) * NB + k] : (sc.g2 >= 0 ? d[first + (UPD_G * sc.g2 + row - 1) * NB + k] : 0.0);
  }
  // the macro tile as ONE 128 x 128 product: wavefront wv owns the block of 64 rows x 32 columns (4 x 2 MFMA accumulators: 6 operand reads
  // from LDS per 8 MFMAs; as four 64 x 64 products of 32 x 16 per wavefront it was 3 reads per 2 MFMAs and the LDS kept the matrix pipes at
  // 55 %: 3.6 us per product against 2.0) -- inside tile (qa, qb) of the macro tile
  const int wv = tid >> 6, l = tid & 63;
  const int qa = wv >> 2, qb = (wv & 3) >> 1, cj = (wv & 1) * 32, lk = l >> 4, ll = l & 15;
  const int i2 = (tid & 31) * 2, kq = tid >> 5;                // operand staging: a work-item takes the row pair i2, i2 + 1 of the columns kq + 16 q
  auto locate = [&](int u, TileItem &t) {
    int p0, np;
    tile_sched_item(sc, panel, u, t.I, t.J, t.act, p0, np);
    t.nch = np; t.kp0 = p0 * NB; t.ds0 = np == 1 ? 0 : 1;
    if (own) for (int b = 0; b < 2; b++) if (!owns_col(own, panel + t.J + b)) t.act &= ~((1 << b) | (1 << (2 + b)));   // (another rank's tile column)
  };
  // A step covers HALF a chunk: 32 columns of the four operand blocks (64 KB: 8 sixteen-byte loads per work-item, two rows each).  The LDS
  // blocks hold a whole chunk; while the MFMAs read one half of them, the next step's operands go from the registers into the other
  // half -- one barrier per step -- and the loads of the step after that are issued.  (With whole chunks per step the 32 prefetch registers
  // more spilled, and the reloads' s_waitcnt vmcnt(0) in front of the MFMA loop waited for the prefetch itself: 16 us per chunk, not 8.)
  sdm_double2 ov[4][2];
  const int rowcap = (ms - 1) & ~1;                            // (the row pair stays inside the column: ld is even)
  auto fetch_operands = [&](const TileItem &t, int kp, int h) {
#pragma unroll
    for (int blk = 0; blk < 4; blk++) {
      const int tr = (blk < 2 ? t.I : t.J) + (blk & 1);
      const double *pp = Fs + min(r0 + tr * TILE + i2, rowcap);
#pragma unroll
      for (int q = 0; q < 2; q++) ov[blk][q] = *(const sdm_double2 *)(pp + (int64_t)(kp + 32 * h + kq + 16 * q) * ld);
    }
  };
  auto stage = [&](const TileItem &t, int ch, int h) {         // registers -> LDS, B scaled by the pivots
    const double *dsh = dshs[t.ds0 + ch];
#pragma unroll
    for (int blk = 0; blk < 4; blk++) {
      const int row = r0 + ((blk < 2 ? t.I : t.J) + (blk & 1)) * TILE + i2;
      double (*Xs)[UTP] = Ab + blk * NB;
#pragma unroll
      for (int q = 0; q < 2; q++) {
        const int k = 32 * h + kq + 16 * q;
        const double dk = blk < 2 ? 1.0 : dsh[k];
        sdm_double2 x;
        x.x = row < ms ? ov[blk][q].x * dk : 0.0; x.y = row + 1 < ms ? ov[blk][q].y * dk : 0.0;
        *(sdm_double2 *)&Xs[k][i2] = x;
      }
    }
  };
  double c[4][2][4];                                           // this wavefront's 64 x 32 block: [16-row block][16-column block][register]
  auto tile_rw = [&](const TileItem &t, bool store) {
    if (t.act >> (2 * qa + qb) & 1) {                          // (uniform per wavefront)
#pragma unroll
      for (int ra = 0; ra < 4; ra++)
#pragma unroll
        for (int cb = 0; cb < 2; cb++)
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int gi = r0 + (t.I + qa) * TILE + ra * 16 + ll, gj = r0 + (t.J + qb) * TILE + cj + cb * 16 + lk + 4 * r;
            double *p = &Fs[(int64_t)min(gj, ms - 1) * ld + min(gi, ms - 1)];
            if (!store) c[ra][cb][r] = *p;
            else if (gi < ms && gj < ms && gi >= gj) *p = c[ra][cb][r];   // (plain stores: nobody reads these tiles before the launch ends; write-through ones took 8 us to issue)
          }
    }
  };
  // step = (macro tile, chunk of 64 columns, half of it); `it`/`ch`/`h` the step in LDS, `nx`/`nch`/`nh` the one in the registers
  int item = w, ch = 0, h = 0;
  TileItem it, nx;
  locate(item, it);
  fetch_operands(it, it.kp0, 0);
  __syncthreads();                                             // dshs
  stage(it, 0, 0);
  int nitem = item, nch = 0, nh = 1;
  bool more = true;
  nx = it;
  fetch_operands(nx, nx.kp0, 1);
  __syncthreads();
#ifdef SDM_PHASES
  long long tp_ = wall_clock64();
  // (free slots of this file's phase array: 9 product, 10 finished tile out, 11 barrier, 12 operands -> LDS (incl. the wait for them), 0 next tile + loads issued, 24 tile values; 13 = steps)
#define SDM_TP(n) do { const long long t_ = wall_clock64(); if (threadIdx.x == 0) atomicAdd(&sdm_phase_acc[(n)], (unsigned long long)(t_ - tp_)); tp_ = t_; } while (0)
#else
#define SDM_TP(n) do {} while (0)
#endif
  sdm_double4 acc[4][2];
  for (;;) {
    // (a new macro tile's own values first: the counter of outstanding loads retires in order, and  c -= acc  must not wait for operands behind them)
    if (ch == 0 && h == 0) tile_rw(it, false);
    SDM_TP(24);
    // ---- the next step: its operands (in flight since the step before) into the other half of the blocks, then the loads of the step after it
    TileItem n2 = nx;
    int n2item = nitem, n2ch = nch, n2h = nh ^ 1;
    bool more2 = more;
    if (more) {
      stage(nx, nch, nh);
      SDM_TP(12);
      if (n2h == 0) {                                          // (the step after the next starts a new chunk, or a new macro tile)
        n2ch = nch + 1;
        if (n2ch == nx.nch) { n2item = nitem + ntw; n2ch = 0; more2 = n2item < nitems; if (more2) locate(n2item, n2); }
      }
      if (more2) fetch_operands(n2, n2.kp0 + n2ch * NB, n2h);
      SDM_TP(0);
    }
    // ---- the product of this step: this wavefront's block of tile (qa, qb) = A(I+qa) B(J+qb)', columns 32 h .. 32 h + 31 of the chunk
    if (it.act >> (2 * qa + qb) & 1) {
      double (*As)[UTP] = Ab + qa * NB, (*Bs)[UTP] = Ab + (2 + qb) * NB;
      if (h == 0) for (int ra = 0; ra < 4; ra++) for (int cb = 0; cb < 2; cb++) for (int r = 0; r < 4; r++) acc[ra][cb][r] = 0.0;
      const int k0 = 32 * h;
      double ao[4], bo[2];
#pragma unroll
      for (int ra = 0; ra < 4; ra++) ao[ra] = As[k0 + lk][ra * 16 + ll];
#pragma unroll
      for (int cb = 0; cb < 2; cb++) bo[cb] = Bs[k0 + lk][cj + cb * 16 + ll];
#pragma unroll
      for (int kk = 0; kk < 32; kk += 4) {
        const int kn = k0 + min(kk + 4, 28);
        double an[4], bn[2];
#pragma unroll
        for (int ra = 0; ra < 4; ra++) an[ra] = As[kn + lk][ra * 16 + ll];
#pragma unroll
        for (int cb = 0; cb < 2; cb++) bn[cb] = Bs[kn + lk][cj + cb * 16 + ll];
#pragma unroll
        for (int ra = 0; ra < 4; ra++)
#pragma unroll
          for (int cb = 0; cb < 2; cb++) acc[ra][cb] = SDM_MFMA_F64_16x16x4(bo[cb], ao[ra], acc[ra][cb]);
#pragma unroll
        for (int ra = 0; ra < 4; ra++) ao[ra] = an[ra];
#pragma unroll
        for (int cb = 0; cb < 2; cb++) bo[cb] = bn[cb];
      }
      if (h == 1) {
#pragma unroll
        for (int ra = 0; ra < 4; ra++)
#pragma unroll
          for (int cb = 0; cb < 2; cb++)
#pragma unroll
            for (int r = 0; r < 4; r++) c[ra][cb][r] = c[ra][cb][r] - acc[ra][cb][r];
      }
    }
    SDM_TP(9);
    if (h == 1 && ch == it.nch - 1) tile_rw(it, true);         // the macro tile is finished
    SDM_TP(10);
    if (!more) break;
    __syncthreads();                                           // the other half of the blocks is complete, this one is free
    SDM_TP(11);
#ifdef SDM_PHASES
    if (threadIdx.x == 0) atomicAdd(&sdm_phase_acc[13], 1ull);
#endif
    it = nx; item = nitem; ch = nch; h = nh;
    nx = n2; nitem = n2item; nch = n2ch; nh = n2h; more = more2;
  }
  SDM_ENDPGM();
}
// ---- workgroup 0 = the dependency chain of the launch, as a CHAIN OF STAGES that never return: the role's prologue calls the update
// of its diagonal tile, which calls the LDL' of the block, which calls the rows left to this workgroup -- each call the last thing its
// caller does.  Each stage is a function for the sake of its own register allocation (inlined into one body they spilled 167 VGPRs,
// round 2); as calls that RETURN they made their caller park what it needed afterwards -- the work-item id, a dozen addresses --
// in scratch around every call, because a called stage uses the whole register file and saves nothing (SDM_NOINLINE).  A caller
// with nothing left to do has nothing to park: what the later stages need travels along as arguments, in registers.
// (The emulator's functions do return: every stage ends with SDM_ENDPGM = return there.)

// (its few-rows form: fronts whose rows below the first block are fewer than MFMA_MIN_ROWS -- the small fronts of arch0, nb)
__device__ SDM_NOINLINE SDM_NORETURN void panel_stage_rows_few(char *smem, SDM_GP(double) Fs_, int ld, int ns, int ms, int k0, int kb, int rbeg, int rend, const double *ds) {
  panel_rows<2>((double *)Fs_, ld, ns, ms, k0, kb, rbeg, rend, TRSM_ROWS, (const double (*)[NB + 1])smem, ds, (double *)smem + NB * (NB + 1));
  SDM_ENDPGM();
}
__device__ SDM_NOINLINE SDM_NORETURN void panel_stage_rows_blocked(char *smem, SDM_GP(double) Fs_, int ld, int ns, int ms, int k0, int kb, int rbeg, int rend, const double *ds) {
  panel_rows<1>((double *)Fs_, ld, ns, ms, k0, kb, rbeg, rend, TRSM_ROWS, (const double (*)[NB + 1])smem, ds, (double *)smem + NB * (NB + 1));
  SDM_ENDPGM();
}
// last stage: the rows of the block column that are this workgroup's own
__device__ SDM_NOINLINE SDM_NORETURN void panel_stage_rows(char *smem, const FrontDesc *fdp, int panel, SDM_GP(double) F_, SDM_GP(int) upd_cnt_, int q0,
                                                          SDM_GP(int) tmo_, const double *ds, bool ok) {
  double *F = (double *)F_;
  int *upd_cnt = (int *)upd_cnt_;
  int *tmo = (int *)tmo_;
  const FrontDesc &fd = *fdp;
  const int s = fd.s, ns = fd.ns, ms = fd.ms, ld = fd.ld;
  double *Fs = F + fd.foff;
  const int k0 = panel * NB, kb = min(NB, ns - k0);
  const int r0 = k0 + kb, nrows = ms - r0;
  SDM_PHASE_BEGIN();
  int rend = r0;                                                   // rows r0 .. rend-1 are solved here (one call site: panel_rows is big)
  if (nrows > TRSM_ROWS) {
    // a partial block (kb < 64, last panel of the supernode) leaves rows r0 .. k0+63 in this workgroup's own tile row:
    // they were updated by its tile (0,0) and are solved here (the row-solve workgroups own whole tile rows)
    if (kb < NB) {
      SDM_ACQUIRE_FENCE();                                         // its own tile-(0,0) stores, not a cached copy from before them
      rend = k0 + NB;
    }
  } else if (nrows > 0) {
    if (panel > 0 && ok) wait_prev_update(upd_cnt + s, ns, ms, panel, q0, tmo);
    rend = ms;
  }
  SDM_PHASE(22);
  if (rend > r0) {
    // (the two forms of the row solve as two stages: together in one function the few-rows form spilled inside its chunk loop)
    if (ms - min(NB, ns) >= MFMA_MIN_ROWS) panel_stage_rows_blocked(smem, (SDM_GP(double))Fs, ld, ns, ms, k0, kb, r0, rend, ds);
    else panel_stage_rows_few(smem, (SDM_GP(double))Fs, ld, ns, ms, k0, kb, r0, rend, ds);
  }
  SDM_PHASE(23);
  SDM_ENDPGM();
}
// middle stage: the LDL' of the block and its publication
__device__ SDM_NOINLINE SDM_NORETURN void panel_stage_block(char *smem, SDM_GP(double) F_, SDM_GP(double) DT_, const FrontDesc *fdp, int panel, SDM_GP(double) d_, SDM_GP(double) lb_,
                                                           SDM_GP(int) pivstat_, SDM_GP(double) pivval_, SDM_GP(const PanelCtx) ctx_, SDM_GP(int) upd_cnt_, SDM_GP(int) diag_cnt_,
                                                           int q0, SDM_GP(int) tmo_) {
  double *F = (double *)F_;
  double *DT = (double *)DT_;
  double *d = (double *)d_;
  double *lb = (double *)lb_;
  int *pivstat = (int *)pivstat_;
  double *pivval = (double *)pivval_;
  const PanelCtx *ctx = (const PanelCtx *)ctx_;
  int *upd_cnt = (int *)upd_cnt_;
  int *diag_cnt = (int *)diag_cnt_;
  int *tmo = (int *)tmo_;
  const FrontDesc &fd = *fdp;
  const int k0 = panel * NB, kb = min(NB, fd.ns - k0);
  const int nrows = fd.ms - (k0 + kb);
  __shared__ double ds[NB];
  __shared__ int npub;
  const bool ok = ldl_diag_block<false>(smem, F, DT, fd, panel, d, lb, pivstat, pivval, ctx, upd_cnt, diag_cnt, q0, tmo, panel == 0, nrows > TRSM_ROWS, ds, &npub);
  if (nrows > TRSM_ROWS) {                                         // the row-solve workgroups of this launch are waiting
    if (16 * npub < kb) SDM_STORES_DONE();                         // (uniform) something of the block is still unpublished
    __syncthreads();
    if (threadIdx.x == 0) sdm_signal_add(&diag_cnt[fd.s], 4 - npub);   // 4 counts per panel: one per 16 columns of the block
    if (kb == NB) SDM_ENDPGM();                                    // nothing of the block column is left to this workgroup
  } else {
    // nobody in this launch waits for this block; the launches to come find the count (= 4 x panels done) in place -- what they
    // read of THIS launch's rows is ordered by the launch boundary, so the count need not wait for the rows below
    SDM_STORES_DONE();
    __syncthreads();
    if (threadIdx.x == 0) sdm_signal_add(&diag_cnt[fd.s], 4);
    if (nrows <= 0) SDM_ENDPGM();
  }
  panel_stage_rows(smem, fdp, panel, (SDM_GP(double))F, (SDM_GP(int))upd_cnt, q0, (SDM_GP(int))tmo, ds, ok);
  SDM_ENDPGM();
}
// first stage (panels after the first): tile (0,0) of the previous update = this panel's diagonal block, straight into S
__device__ SDM_NOINLINE SDM_NORETURN void panel_stage_update(char *smem, SDM_GP(double) F_, SDM_GP(double) DT_, const FrontDesc *fdp, int panel, SDM_GP(double) d_, SDM_GP(double) lb_,
                                                            SDM_GP(int) pivstat_, SDM_GP(double) pivval_, SDM_GP(const PanelCtx) ctx_, SDM_GP(int) upd_cnt_, SDM_GP(int) diag_cnt_,
                                                            int q0, SDM_GP(int) tmo_) {
  SDM_FP_STRICT;
  {
    const FrontDesc &fd = *fdp;
    double *Fs = (double *)F_ + fd.foff;
    const double *d = (const double *)d_;
    double (*As)[UTP] = (double (*)[UTP])smem;
    double (*Bs)[UTP] = As + NB;
    __shared__ double dsh[NB];
    update_tile<LDL_THREADS / 64, true>(Fs, fd.ld, fd.ms, fd.first, (panel - 1) * NB, NB, 0, 0, d, As, Bs, dsh,
                                        (double (*)[NB + 1])smem, (double *)smem + NB * (NB + 1), min(NB, fd.ns - panel * NB));
  }
  panel_stage_block(smem, F_, DT_, fdp, panel, d_, lb_, pivstat_, pivval_, ctx_, upd_cnt_, diag_cnt_, q0, tmo_);
  SDM_ENDPGM();
}
// the role's prologue
__device__ SDM_NOINLINE SDM_NORETURN void panel_role_diag(char *smem, SDM_GP(double) F_, SDM_GP(double) DT_, FrontDesc *fdp, int panel, SDM_GP(double) d_,
                                                         SDM_GP(const PanelCtx) ctx_, SDM_GP(int) upd_cnt_, SDM_GP(int) diag_cnt_, int q0, SDM_GP(int) tmo_) {
  const PanelCtx *ctx = (const PanelCtx *)ctx_;
  // (uniform loads: one scalar round trip, issued before the tile of the previous update is fetched)
  double *lb = ctx->lb; int *pivstat = ctx->pivstat; double *pivval = ctx->pivval;
  {
    const double *ubp = ctx->ubp;
    const double maxu = ubp[1], ub = ubp[2] / (maxu * maxu);
    if (threadIdx.x == 0) { fdp->maxu = maxu; fdp->ub = ub; }      // (read behind the first barrier of the diagonal block)
  }
  if (panel > 0) panel_stage_update(smem, F_, DT_, fdp, panel, d_, (SDM_GP(double))lb, (SDM_GP(int))pivstat, (SDM_GP(double))pivval, ctx_, upd_cnt_, diag_cnt_, q0, tmo_);
  else panel_stage_block(smem, F_, DT_, fdp, panel, d_, (SDM_GP(double))lb, (SDM_GP(int))pivstat, (SDM_GP(double))pivval, ctx_, upd_cnt_, diag_cnt_, q0, tmo_);
  SDM_ENDPGM();
}

__global__ void __launch_bounds__(LDL_THREADS)
k_ldl_panel(double *F, double *DT, FrontTab tab, const int *list, int panel, double *d, const PanelCtx *ctx,
            int *upd_cnt, int *diag_cnt, int q0, int phase, int *tmo, int own) {
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
  const int s = list[blockIdx.y];
  const int ns = tab.ns[s], ms = tab.ms[s], ld = tab.ld[s], first = tab.first[s];
  double *Fs = F + tab.foff[s];
  const int k0 = panel * NB, kb = min(NB, ns - k0);
  const int r0 = k0 + kb, nrows = ms - r0;
  const int nrw = nrows > TRSM_ROWS ? (ms - (k0 + NB) + ROWS_BATCH - 1) / ROWS_BATCH : 0;   // tile rows below the first
  const int nt = panel > 0 ? (ms - k0 + TILE - 1) / TILE : 0;    // tile rows of the previous panel's update
  // Roles by "logical" index bx: 0 = diagonal block, 1..nrw = row solves, beyond = update tiles.  The hardware hands
  // out workgroups in launch order, and a workgroup that waits must wait for one handed out BEFORE it (or for one
  // that does not wait before it signals), else a full device of waiting workgroups could keep the awaited one out:
  //   fronts with row-solve workgroups:  diagonal block < row solves (wait for it) < tiles (nobody waits for them);
  //   small fronts:                      tiles (never wait) < diagonal block (its in-workgroup row solve waits for them).
  // The emulator runs them one after the other in the order  tiles, row solves (phase 1: their update tile only),
  // diagonal block.
  int bx = blockIdx.x;
#ifdef SDM_EMU
  if (phase != 0) bx = (int)gridDim.x - 1 - (int)blockIdx.x;        // (phase 0 in the emulator: its workgroups run as concurrent processes, roles as on the device)
  else
#endif
  if (nrw == 0) {
    const int ntw = panel > 0 ? (nt * (nt + 1) / 2) / 2 : 0;
    if (bx > ntw) return;
    bx = bx < ntw ? 1 + bx : 0;
  }
  if (bx <= nrw && !owns_col(own, panel)) return;                  // (block-cyclic ranks: the panel is factored by the owner of its tile column alone)
  if (bx > 0 && bx <= nrw) {
    panel_role_rows(smem, Fs, DT + tab.toff[s] + (int64_t)panel * NB * NB, d, ns, ms, ld, first, panel, bx - 1, upd_cnt + s, diag_cnt + s, phase, tmo);
    return;
  }
  if (bx > nrw) {
    if (phase == 2 || panel == 0) return;
    if (nrw > 0) panel_role_tiles_stream(smem, Fs, d, ns, ms, ld, first, panel, bx - 1 - nrw, (int)gridDim.x - 1 - nrw, own);
    else panel_role_tiles(smem, Fs, d, ms, ld, first, panel, bx - 1 - nrw, nrw, nt, upd_cnt + s, own);
    return;
  }
  // ---- workgroup 0
  if (phase == 2) return;
  __shared__ FrontDesc fd;                                          // (every work-item writes the same values: no barrier needed before it reads them back)
  fd.ns = ns; fd.ms = ms; fd.ld = ld; fd.first = first; fd.foff = tab.foff[s]; fd.toff = tab.toff[s]; fd.woff = tab.woff[s]; fd.s = s;
  panel_role_diag(smem, (SDM_GP(double))F, (SDM_GP(double))DT, &fd, panel, (SDM_GP(double))d, (SDM_GP(const PanelCtx))ctx, (SDM_GP(int))upd_cnt, (SDM_GP(int))diag_cnt, q0,
                  (SDM_GP(int))tmo);
}

// ---- hand-over of a factored diagonal block to the workgroups of k_ldl_front that solve rows against it: DATA-TAGGED.  The
// transposed copy DT of every block starts a factorisation filled with a sentinel (k_prep_pivots); the block's workgroup
// stores each 16-column group write-through as it becomes final -- the pivots in the diagonal slots of DT, which nobody else
// reads -- and the consumers poll the 8-byte words they need until none of them is the sentinel.  No counter, no
// acknowledgement wait, no second round trip between "it is there" and "here it is": the flag-then-load form (store,
// s_waitcnt, counter, poll, sc1 read) was 5.5 us of the 22.8 us per panel of control07's chain.  diag_cnt
// is still counted for the consumers that are not on the chain (k_ldl_panel's row solves, k_sinv_follow, the column probe).
__device__ __forceinline__ bool is_dt_sentinel(double v) { union { double d; unsigned long long u; } b; b.d = v; return b.u == DT_SENTINEL; }
__device__ __forceinline__ double dt_tagged_load(const double *a, int *tmo) {
  double v = sdm_load_wt(a);
  for (long it = 0; is_dt_sentinel(v); it++) { if (sdm_spin_giveup(it, tmo)) break; SDM_SPIN_PAUSE(); v = sdm_load_wt(a); }
  return v;
}
// columns 16 blk .. 16 blk + 15 of the block (strictly lower part, rows < kb) into S, their pivots into dsr; all work-items.
// Every load of a work-item (two entries, for 16 of them a pivot) is in flight before the first one is looked at: ONE memory
// round trip per group when the data is there, not one per word.
__device__ __forceinline__ void diag_group_fetch(const double *Ds, int blk, int kb, double (*S)[NB + 1], double *dsr, int *tmo) {
  const int tid = threadIdx.x;
  constexpr int NE = NB * 16 / LDL_THREADS;
  const double *a[NE + 1];
  double v[NE + 1];
  bool need[NE + 1];
#pragma unroll
  for (int t = 0; t < NE; t++) {
    const int e = tid + LDL_THREADS * t, i = e >> 4, j = 16 * blk + (e & 15);
    need[t] = i < kb && j < i;
    a[t] = &Ds[i * NB + j];
  }
  need[NE] = tid < 16 && 16 * blk + tid < kb;
  a[NE] = &Ds[(16 * blk + (tid & 15)) * NB + 16 * blk + (tid & 15)];
#pragma unroll
  for (int t = 0; t <= NE; t++) v[t] = need[t] ? sdm_load_wt(a[t]) : 0.0;
#pragma unroll
  for (int t = 0; t <= NE; t++)
    if (need[t])
      for (long it = 0; is_dt_sentinel(v[t]); it++) { if (sdm_spin_giveup(it, tmo)) break; SDM_SPIN_PAUSE(); v[t] = sdm_load_wt(a[t]); }
#pragma unroll
  for (int t = 0; t < NE; t++) {
    const int e = tid + LDL_THREADS * t, i = e >> 4, j = 16 * blk + (e & 15);
    S[i][j] = v[t];
  }
  if (tid < 16) dsr[16 * blk + tid] = v[NE];
  __syncthreads();
}

// k_ldl_front calls its three stages through real function calls: each gets a register allocation of its own.  Inlined
// into one body they share 256 VGPRs with the sweep code of the diagonal block and spill inside the store loops -- and a
// scratch reload between two write-through stores waits for the first one's acknowledgement (vmcnt counts in order):
// every stored tile then costs eight memory round trips instead of one.
__device__ SDM_NOINLINE bool front_diag(char *smem, SDM_GP(double) F_, SDM_GP(double) DT_, const FrontDesc *fd, int panel, SDM_GP(double) d_, SDM_GP(double) lb_,
                                        SDM_GP(int) pivstat_, SDM_GP(double) pivval_, SDM_GP(const PanelCtx) ctx_, SDM_GP(int) upd_done_, SDM_GP(int) diag_cnt_, SDM_GP(int) tmo_,
                                        bool load_block, bool publish, double *ds, int *npub, bool raw_in_lds, const double *lbs_pre) {
  double *F = (double *)F_;
  double *DT = (double *)DT_;
  double *d = (double *)d_;
  double *lb = (double *)lb_;
  int *pivstat = (int *)pivstat_;
  double *pivval = (double *)pivval_;
  const PanelCtx *ctx = (const PanelCtx *)ctx_;
  int *upd_done = (int *)upd_done_;
  int *diag_cnt = (int *)diag_cnt_;
  int *tmo = (int *)tmo_;
  return ldl_diag_block<true>(smem, F, DT, *fd, panel, d, lb, pivstat, pivval, ctx, upd_done, diag_cnt, 0, tmo, load_block, publish, ds, npub, raw_in_lds, 0, lbs_pre);
}
// kind 0: plain (result to the front only), 2: also the wave tiles of the next row solve (RB)
__device__ SDM_NOINLINE void front_update(int kind, SDM_GP(double) Fs_, int ld, int ms, int first, int k0, int I, int J, SDM_GP(const double) d_, char *smem,
                                          double *dsh, int kbn) {
  double *Fs = (double *)Fs_;
  const double *d = (const double *)d_;
  double (*As)[UTP] = (double (*)[UTP])smem;
  double (*Bs)[UTP] = As + NB;
  double *RB = (double *)smem + NB * (NB + 1);
  if (kind == 2) update_tile<LDL_THREADS / 64, false, true, true>(Fs, ld, ms, first, k0, NB, I, J, d, As, Bs, dsh, nullptr, nullptr, kbn, threadIdx.x, true, RB);
  else update_tile<LDL_THREADS / 64, false, true>(Fs, ld, ms, first, k0, NB, I, J, d, As, Bs, dsh);
}
// R: rows of tile row r against the diagonal block of panel q as it is published; have_tw: the tile is in the wave tiles already
__device__ SDM_NOINLINE void front_rows(SDM_GP(double) Fs_, SDM_GP(const double) Ds_, int ld, int ms, int q, int kb, int r, char *smem,
                                        double *dsr, SDM_GP(int) tmo_, bool have_tw, bool defer_ack) {
  double *Fs = (double *)Fs_;
  const double *Ds = (const double *)Ds_;
  int *tmo = (int *)tmo_;
  double (*S)[NB + 1] = (double (*)[NB + 1])smem;
  double *RB = (double *)smem + NB * (NB + 1);
  const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
  const int k0 = q * NB, rbeg = r * TILE, rend = min(ms, rbeg + TILE);
  const int R0 = rbeg + 16 * ty;
  const bool busy = R0 < rend;
  double *Tw = RB + ty * (NB * 17);
  if (!have_tw && busy) rows_stage(Fs, ld, rend, k0, kb, R0, Tw, tx);
  for (int blk = 0; blk < NB / 16 && 16 * blk < kb; blk++) {
    if (busy) rows_block_gemm(blk, S, Tw, tx);
    diag_group_fetch(Ds, blk, kb, S, dsr, tmo);                      // polls the data itself (DT starts as a sentinel)
    if (busy) rows_block_tri(blk, S, dsr, Tw, tx);
  }
  if (busy) rows_store<true>(Fs, ld, rend, k0, kb, R0, dsr, Tw, tx);
  if (defer_ack) return;                                           // the caller counts the rows after its next piece of work
  SDM_STORES_DONE();
  __syncthreads();
}
// The two of them fused for the workgroup on the chain (r = q + 1): the rows are solved 16 columns at a time as before, but
// every finished 16-column block is scaled (l = x / d, as rows_store does) and kept in LDS (Lt, the wave tiles the four idle
// wavefronts do not use; it goes to the front write-through when the LAST block has arrived -- the column probe of the diagonal
// block's workgroup must find the rows unsolved until then), and the update of the workgroup's own diagonal tile reads its operands
// from there -- A operand l, B operand l * d, the products in the same k order as update_tile, so the same bits.  The
// K = 48 part of the update runs while the last 16 columns of the diagonal block are still being factored; after they
// arrive only their triangle, 4 of the 16 k-steps and the epilogue are left.  The raw updated block stays in LDS (cvl:
// the general path of the LDL' reloads it from there); it reaches the front as the factored block.
__device__ SDM_NOINLINE void front_rows_diag(SDM_GP(double) Fs_, SDM_GP(const double) Ds_, int ld, int ms, int q, int r, char *smem,
                                             double *dsr, SDM_GP(int) tmo_, bool have_tw, int kbn) {
  double *Fs = (double *)Fs_;
  const double *Ds = (const double *)Ds_;
  int *tmo = (int *)tmo_;
  double (*S)[NB + 1] = (double (*)[NB + 1])smem;
  double *RB = (double *)smem + NB * (NB + 1);
  double *cvl = (double *)smem + FRONT_CV_OFF;
  constexpr int NW = LDL_THREADS / 64;
  const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
  const int k0 = q * NB, rbeg = r * TILE, rend = min(ms, rbeg + TILE);
  const int R0 = rbeg + 16 * ty;
  const bool busy = R0 < rend && ty < 4;
  double *Tw = RB + ty * (NB * 17);
  double *Lt = RB + 4 * (NB * 17);                                    // Lt[w * NB*17 + c*17 + li] = l(row 16w + li, column c)
  const int li = tx & 15, lk = tx >> 4;
  const int wi = ty >> 2, wj = ty & 3;                                // update_tile's NW = 8 mapping: rows 32 wi + 16 a + li, columns 16 wj + ..
  sdm_double4 acc[2];
  for (int a = 0; a < 2; a++) for (int x = 0; x < 4; x++) acc[a][x] = 0.0;
  if (!have_tw && busy) rows_stage(Fs, ld, rend, k0, NB, R0, Tw, tx);
  for (int blk = 0; blk < NB / 16; blk++) {
    if (busy) rows_block_gemm(blk, S, Tw, tx);
    diag_group_fetch(Ds, blk, NB, S, dsr, tmo);                      // polls the data itself (DT starts as a sentinel)
    SDM_TRACE(16 * q + 2 + blk);                                     // R of the chain workgroup: group blk has arrived
    if (busy) {
      rows_block_tri(blk, S, dsr, Tw, tx);
      // l = x / d of the block's 16 columns: to the front (write-through) and to Lt (rows beyond the front: 0)
#pragma unroll
      for (int c4 = 0; c4 < 4; c4++) {
        const int c = 16 * blk + 4 * c4 + lk, row = R0 + li;
        const double dc = dsr[c], xv = Tw[c * 17 + li];
        const double l = (row < rend && dc > 0.0) ? xv / dc : 0.0;
        if (blk == 3 && row < rend) sdm_store_wt(&Fs[(int64_t)(k0 + c) * ld + row], l);
        Lt[ty * (NB * 17) + c * 17 + li] = l;
      }
    } else if (blk == 3 && ty >= 4) {
      // the rows of the first three groups go to the front only NOW, when the last group has arrived.  Until then workgroup q may
      // still take the general path for a later column of its block, and the column probe reads these rows' unsolved values
      // from the front (a store per finished group -- the first form of this function -- replaced them with multipliers under
      // the probe's eyes: wrong pivot decisions on rank-deficient fronts, profiles/r03ap_soak_def.txt).  The four wavefronts
      // without rows of their own write them from Lt, beside the last group's substitution.  (Keeping the store per group and
      // giving the probe a shadow copy of the unsolved rows is no faster: profiles/r03at.  The loop must NOT be unrolled: with
      // 12 loads and addresses in flight the function needs callee-saved VGPRs and saves 76 of them through scratch on every
      // call, 1.4 us per panel -- tests/test_abi.py checks.)
      const int w = ty - 4, rowb = rbeg + 16 * w + li;
#pragma unroll 1
      for (int c4 = 0; c4 < 12; c4++) {
        const int c = 4 * c4 + lk;
        if (rowb < rend) sdm_store_wt(&Fs[(int64_t)(k0 + c) * ld + rowb], Lt[w * (NB * 17) + c * 17 + li]);
      }
    }
    if (blk < 2) continue;
    __syncthreads();                                                  // Lt of the columns up to 16 blk + 15 is complete
    // k-steps of the update whose columns are final: 0 .. 11 after block 2, 12 .. 15 after block 3
    for (int kk = (blk == 2 ? 0 : 48); kk < (blk == 2 ? 48 : 64); kk += 4) {
      const int k = kk + lk;
      const double bv = Lt[wj * (NB * 17) + k * 17 + li] * dsr[k];
#pragma unroll
      for (int a = 0; a < 2; a++) acc[a] = SDM_MFMA_F64_16x16x4(bv, Lt[(2 * wi + a) * (NB * 17) + k * 17 + li], acc[a]);
    }
  }
  __syncthreads();                                                    // S, the wave tiles and Lt are dead: S / Lc of the LDL' overlay them
  double *Lc = RB;
  for (int j = ty; j < NB; j += NW) { S[tx][j] = (tx == j && tx >= kbn) ? 1.0 : 0.0; Lc[j * NB + tx] = 0.0; }
  __syncthreads();
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int x = 0; x < 4; x++) {
      const int ti = wi * 32 + a * 16 + li, tj = 16 * wj + lk + 4 * x;
      if (rbeg + ti < ms && rbeg + tj < ms && ti >= tj) {
        const double v = cvl[tj * TILE + ti] - acc[a][x];
        cvl[tj * TILE + ti] = v;                                      // the raw updated block (general path of the LDL')
        if (ti < kbn) S[ti][tj] = v;
      }
    }
  SDM_TRACE(16 * q + 7);                                               // chain workgroup: its diagonal tile is in the LDS arrays of the LDL'
}

// ---- the whole LDL' of a front in ONE launch (fronts of FRONT_MINMS <= m_s <= 64 FRONT_MAXT rows; chol_build decides per
// level).  All workgroups resident.  One ROW workgroup per 64-row tile row r of the front; for the panels q = 0, 1, ... it does
//   q <  r   R: rows of tile (r, q) against the factored diagonal block of panel q (published by workgroup q, 16 columns
//               at a time, exactly as in k_ldl_panel), result stored write-through and counted in row_cnt[r];
//            U: the LAST update of the tile in the next panel's column, (r, q+1) -= L(r, q) D_q L(q+1, q)': it stays in LDS
//               as the next R's input, or -- r = q+1 -- goes straight into the LDS arrays of the LDL';
//   q == r   D: LDL' of the diagonal block (ldl_diag_block), then the workgroup is done.
// The TILE workgroups apply the other updates q = 0 .. c-2 of the tiles (r, c), 2 <= c <= r, as soon as row_cnt says L(r, q)
// and L(c, q) are there, and count them in tile_cnt (the row workgroup waits for it before the tile's last update): a row
// workgroup never has more than one tile update between two row solves.  One workgroup per tile by default (chol_build's
// FRONT_POOL rule); with fewer, workgroup w owns the tiles w, w + ntw, ... and goes through them panel by panel.
// The chain per panel is D -> (hand-over) -> last block of R in workgroup q+1 -> its diagonal tile -> D: no launch
// boundary, no wait for the slowest row workgroup.  Arithmetic and its order per entry are those of the launch-per-panel
// path (same device functions), so both produce the same bits.  upd_done[r] counts the U steps finished (the rare column
// probe of a later diagonal block waits for them).  The emulator runs workgroups one after the other: there the host
// loops over (step, phase 1 = D, 2 = R, 3 = U) and nothing is carried in LDS.
// (scalar arguments: they travel in the 32 argument registers of the convention; a FrontTab by reference made the kernel keep its copy in scratch)
__device__ SDM_NOINLINE void front_follow_stage(char *smem, int bx, int fs, int ns, int ld, int sld, int slot, int sboff, int64_t foff, int64_t soff, int64_t toff,
                                                 SDM_GP(const double) F_, SDM_GP(const double) DT_, SDM_GP(double) S_, SDM_GP(double) STr_, SDM_GP(int) front_cnt_,
                                                 SDM_GP(const int) diag_cnt_, SDM_GP(unsigned long long) sb_g_, SDM_GP(int) tmo_) {
  FollowDesc fd;
  fd.s = fs; fd.ns = ns; fd.ld = ld; fd.sld = sld; fd.slot = slot; fd.sboff = sboff; fd.foff = foff; fd.soff = soff; fd.toff = toff;
  sinv_follow_body(smem, bx, fd, (const double *)F_, (const double *)DT_, (double *)S_, (double *)STr_, (int *)front_cnt_, (const int *)diag_cnt_,
                   (unsigned long long *)sb_g_, (int *)tmo_);
}
__global__ void __launch_bounds__(LDL_THREADS)
k_ldl_front(double *F, double *DT, FrontTab tab, const int *list, double *d, double *lb, const double *ubp, int *pivstat,
            double *pivval, const PanelCtx *ctx, int *front_cnt, int *diag_cnt, int phase, int step, int tile_wg0, int *tmo, FollowArgs fa) {
  SDM_FP_STRICT;
  SDM_DYN_SMEM(smem);
  if (fa.S && (int)blockIdx.x >= fa.nfront) {
    // ---- the workgroups that build the front's inverse for the solves BEHIND its factorisation (sdm_follow.h; they poll the counters
    // the workgroups below count).  They used to be a launch of their own on a second stream: forking to it and joining it again cost
    // 6.8 + 11.5 us of every factorisation (event record, cross-queue wait: profiles/r04p_*, r04q_*); as the last workgroups of this
    // launch -- dispatched after every workgroup they wait for -- they cost nothing.  256 of the 512 work-items do the work: a
    // wavefront that has ended is not waited for by the barriers of the others.
    if (threadIdx.x >= ST) return;
    const FollowDesc fd = follow_desc(tab, list, (int)blockIdx.y);
    front_follow_stage(smem, (int)blockIdx.x - fa.nfront, fd.s, fd.ns, fd.ld, fd.sld, fd.slot, fd.sboff, fd.foff, fd.soff, fd.toff, (SDM_GP(const double))F,
                       (SDM_GP(const double))DT, (SDM_GP(double))fa.S, (SDM_GP(double))fa.STr, (SDM_GP(int))front_cnt, (SDM_GP(const int))diag_cnt,
                       (SDM_GP(unsigned long long))fa.sb_g, (SDM_GP(int))tmo);
    return;
  }
  const int s = list[blockIdx.y];
  const int ns = tab.ns[s], ms = tab.ms[s], ld = tab.ld[s], first = tab.first[s];
  const int T = (ms + TILE - 1) / TILE, NP = (ns + NB - 1) / NB;
  double *Fs = F + tab.foff[s];
  int *row_cnt = front_cnt + (int64_t)tab.fslot[s] * FRONT_CNT, *upd_done = row_cnt + FRONT_MAXT, *tile_cnt = upd_done + FRONT_MAXT;
  __shared__ double dsh[NB], ds[NB], dsr[NB];
  __shared__ int npub;
  const int tid = threadIdx.x;
  if ((int)blockIdx.x >= tile_wg0) {
    // ---- tile workgroup: owner of the tiles t = w, w + ntw, ... (tiles (rt, ct), 2 <= ct <= rt, counted column by column)
    // for the updates of the panels q <= ct - 2 (the last update of a tile, q = ct - 1, belongs to the row workgroup: its
    // result is the next R's / D's input).  Panel by panel, and within a panel its tiles left to right: the tiles the
    // chain needs next (column q + 2) come first.  Everything it waits for (the rows of panel q) depends on tile updates
    // of EARLIER panels only, which this loop has finished by then.
    if (phase != 0 && phase != 3) return;
    const int w = (int)blockIdx.x - tile_wg0, ntw = fa.nfront - tile_wg0;   // (the grid carries the follower workgroups behind the fa.nfront of the factorisation)
    if (w >= (T - 1) * (T - 2) / 2) return;
    for (int q = (phase == 0 ? 0 : step); q < (phase == 0 ? NP : step + 1) && q <= T - 3; q++) {
      int ct = 2, c0 = 0;                                              // c0 = index of tile (ct, ct)
      for (int t = w; t < (T - 1) * (T - 2) / 2; t += ntw) {
        while (t >= c0 + (T - ct)) { c0 += T - ct; ct++; }
        const int rt = ct + (t - c0);
        if (q > ct - 2) continue;
        spin_until(row_cnt + rt, q + 1, tmo);
        if (ct < rt) spin_until(row_cnt + ct, q + 1, tmo);
        front_update(0, (SDM_GP(double))Fs, ld, ms, first, q * NB, rt - (q + 1), ct - (q + 1), (SDM_GP(const double))d, smem, dsh, 0);
        SDM_STORES_DONE();
        __syncthreads();
        if (tid == 0) sdm_signal_add(&tile_cnt[rt * FRONT_MAXT + ct]);
      }
    }
    return;
  }
  const int r = blockIdx.x;
  if (r >= T) return;
  const bool carry = phase == 0;
  __shared__ FrontDesc fd;                                          // (every work-item writes the same values: no barrier needed before it reads them back)
  fd.ns = ns; fd.ms = ms; fd.ld = ld; fd.first = first; fd.foff = tab.foff[s]; fd.toff = tab.toff[s]; fd.woff = tab.woff[s]; fd.s = s;
  { const double maxu = ubp[1]; fd.maxu = maxu; fd.ub = ubp[2] / (maxu * maxu); }
  __shared__ double lbs_pre[NB];
  if (carry && r < NP && tid < NB) lbs_pre[tid] = r * NB + tid < ns ? lb[first + r * NB + tid] : 0.0;      // thresholds of the block this workgroup will factor
  bool have_S = false, have_tw = false, raw_in_lds = false;
  for (int q = carry ? 0 : step; q < (carry ? NP : step + 1) && q <= r; q++) {
    const int k0 = q * NB, kb = min(NB, ns - k0);
    if (q == r) {
      if (phase == 0 || phase == 1) {
        const int nrows = ms - (k0 + kb);
        front_diag(smem, (SDM_GP(double))F, (SDM_GP(double))DT, &fd, q, (SDM_GP(double))d, (SDM_GP(double))lb, (SDM_GP(int))pivstat, (SDM_GP(double))pivval, (SDM_GP(const PanelCtx))ctx,
                   (SDM_GP(int))upd_done, (SDM_GP(int))diag_cnt, (SDM_GP(int))tmo, !have_S, nrows > 0, ds, &npub, raw_in_lds, carry ? lbs_pre : nullptr);
        if (nrows > 0) {
          if (16 * npub < kb) SDM_STORES_DONE();
          __syncthreads();
          if (tid == 0) sdm_signal_add(&diag_cnt[s], 4 - npub);
        } else {                                                      // nobody in this launch waits for the last block; k_sinv_follow does
          SDM_STORES_DONE();
          __syncthreads();
          if (tid == 0) sdm_signal_add(&diag_cnt[s], 4);
        }
      }
      break;
    }
    const int rbeg = r * TILE;
#if defined(SDM_PHASES) && !defined(SDM_EMU)
    const bool crit = r == q + 1;
    long long fph_ = wall_clock64();
#define SDM_FPHASE(n) do { const long long t_ = wall_clock64(); if (crit && threadIdx.x == 0) atomicAdd(&sdm_phase_acc[n], (unsigned long long)(t_ - fph_)); fph_ = t_; } while (0)
#else
#define SDM_FPHASE(n) do {} while (0)
#endif
    const bool crit_lds = carry && r == q + 1 && q + 1 < NP;         // the next diagonal block's workgroup: its tile update runs from LDS
    if (crit_lds) {
      // the tile's current values into LDS while the diagonal block of panel q is still being factored
      if (r >= 2) spin_until(tile_cnt + r * FRONT_MAXT + r, q, tmo);
      else SDM_ACQUIRE_FENCE();
      double *cvl = (double *)smem + FRONT_CV_OFF;
      double t8[NB * TILE / LDL_THREADS];
#pragma unroll
      for (int j = 0; j < NB * TILE / LDL_THREADS; j++) {
        const int e = tid + LDL_THREADS * j, ti = e & 63, tj = e >> 6;
        t8[j] = Fs[(int64_t)min(rbeg + tj, ms - 1) * ld + min(rbeg + ti, ms - 1)];
      }
#pragma unroll
      for (int j = 0; j < NB * TILE / LDL_THREADS; j++) cvl[tid + LDL_THREADS * j] = t8[j];
    }
    if (phase == 0 || phase == 2) {
      // ---- R: 16 rows per wavefront (4 of the 8 busy), following the diagonal block as workgroup q publishes it
      if (crit_lds) front_rows_diag((SDM_GP(double))Fs, (SDM_GP(const double))(DT + tab.toff[s] + (int64_t)q * NB * NB), ld, ms, q, r, smem, dsr, (SDM_GP(int))tmo, have_tw,
                                    min(NB, ns - (q + 1) * NB));
      else front_rows((SDM_GP(double))Fs, (SDM_GP(const double))(DT + tab.toff[s] + (int64_t)q * NB * NB), ld, ms, q, kb, r, smem, dsr, (SDM_GP(int))tmo, have_tw, false);
      SDM_FPHASE(1);
      have_tw = false;
      if (crit_lds) {
        have_S = true; raw_in_lds = true;
        SDM_FPHASE(4);
        // (counting these from inside the next LDL', behind its first sweep, takes the acknowledgement wait off this workgroup's
        // path -- and puts 2 us on the path of tile row r + 1, whose last update before ITS turn waits for exactly this count:
        // measured slower, profiles/r03l)
        SDM_STORES_DONE();
        __syncthreads();
        if (tid == 0) { sdm_signal_add(&row_cnt[r]); sdm_signal_add(&upd_done[r]); }
        SDM_FPHASE(5);
        SDM_TRACE(16 * q + 8);                                         // its rows acknowledged and counted
        continue;
      }
      if (tid == 0) sdm_signal_add(&row_cnt[r]);
      SDM_FPHASE(2);                                                  // rows stored, acknowledged, counted
    }
    if (phase == 0 || phase == 3) {
      // ---- U: the LAST update of the tile in the next panel's column, (r, q+1); its earlier ones came from the tile's
      // own workgroup (tile_cnt), the others of this row are theirs altogether
      const int c1 = q + 1;
      if (c1 <= r) {
        SDM_ACQUIRE_FENCE();                                          // this workgroup's own L(r, q), not a cached copy from before
        if (c1 >= 2) spin_until(tile_cnt + r * FRONT_MAXT + c1, q, tmo);
        if (c1 < r) spin_until(row_cnt + c1, q + 1, tmo);
        SDM_FPHASE(3);                                                // fence + counters
        const int I = r - c1, J = 0;
        const int kbn = min(NB, ns - c1 * NB);                        // columns of the next panel (<= 0: none)
        if (c1 < r && carry && c1 < NP) { front_update(2, (SDM_GP(double))Fs, ld, ms, first, k0, I, J, (SDM_GP(const double))d, smem, dsh, kbn); have_tw = true; }
        else front_update(0, (SDM_GP(double))Fs, ld, ms, first, k0, I, J, (SDM_GP(const double))d, smem, dsh, kbn);
      }
      SDM_FPHASE(4);                                                  // the tile update
      SDM_STORES_DONE();
      __syncthreads();
      if (tid == 0) sdm_signal_add(&upd_done[r]);
      SDM_FPHASE(5);
    }
  }
}

// stand-alone update.  Supernodes that END with this panel: all tiles (the update of the rows beyond, passed up to the
// parent).  Supernodes with a next panel: nothing when `riding` (their tiles ride along with the diagonal-block launch
// of the next panel, k_ldl_panel), else every tile but tile 0 (which k_ldl_panel's workgroup 0 always applies itself).
__global__ void __launch_bounds__(256)
k_ldl_update(double *F, FrontTab tab, const int *list, int panel, const double *d, int riding) {
  __shared__ double As[NB][UTP];
  __shared__ double Bs[NB][UTP];
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
// The triangular solves live in sdm_solve.hip (explicit inverses of the diagonal super-blocks, one GEMV launch per
// super-block column).  Only the small vector helpers they share with the factor remain here.
__global__ void k_gather_perm(double *dst, const double *src, const int *perm, int m, int forward) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < m) { if (forward) dst[k] = src[perm[k]]; else dst[perm[k]] = src[k]; }
}
// ./d of wrapPcg.m:57 with deninfac.m:89-94 folded in: skipped pivots (d = 0) act as 1
__global__ void k_divd(double *v, const double *d, int m) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < m) { const double dk = d[k]; v[k] /= dk > 0.0 ? dk : 1.0; }
}

// ============================================================ host drivers
void chol_forget_plan(sdm_plan *P) { PersistTurn::forget(P); }
FrontTab front_tab(CholPlan &C) {
  FrontTab t;
  t.soff = C.d_soff.p; t.sld = C.d_sld.p; t.sboff = C.d_sboff.p; t.ltoff = C.d_ltoff.p;
  t.first = C.d_first.p; t.ns = C.d_ns.p; t.ms = C.d_ms.p; t.ld = C.d_ld.p;
  t.foff = C.d_foff.p; t.xl = C.d_xl.p; t.woff = C.d_woff.p; t.roff = C.d_roff.p; t.toff = C.d_toff.p; t.fslot = C.d_fslot.p;
  t.childptr = C.d_childptr.p; t.childlist = C.d_childlist.p; t.lindx = C.d_lindx.p; t.relidx = C.d_relidx.p;
  return t;
}
static inline int grid1d(int64_t n, int bs, int cap = 4096) {
  int64_t g = (n + bs - 1) / bs;
  return (int)std::max<int64_t>(1, std::min<int64_t>(g, cap));
}

void chol_begin(sdm_plan *P, const double canceltol, const double maxu, const double abstol, int use_absd) {
  CholPlan &C = P->chol;
  hipStream_t st = P->stream;
  const int m = (int)C.m;
  P->factored = false;               // until the launches below have all been issued: a factorisation that throws leaves no factor behind
#ifndef SDM_EMU
  SDM_HIP_CHECK(hipFuncSetAttribute((const void *)k_ldl_panel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PANEL_LDS_RIDE));
  SDM_HIP_CHECK(hipFuncSetAttribute((const void *)k_ldl_front, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FRONT_LDS));
#endif
  // (DT needs its sentinel only where k_ldl_front runs; the launch-per-panel levels wait on counters)
  bool any_persist = false;
  for (int l = 0; l < C.nlevels; l++) any_persist = any_persist || C.lev_persist[l];
  const int64_t ndt = any_persist ? C.tsize : 0;
  PrepArgs pa = {m, P->ada_val.p, C.d_asm_src.p, C.d_Ljc.p, C.d_perm.p, P->absd.p, use_absd, canceltol, maxu, abstol, C.lb.p, C.ub.p, C.pivstat.p, C.pivval.p,
                 (int)C.nsuper, C.upd_cnt.p, C.diag_cnt.p, C.sb_g.p, (int)C.sb_g.n, C.front_cnt.p, (int)C.front_cnt.n, (unsigned long long *)C.frontsT.p, ndt};
  const int nprep = grid1d(std::max<int64_t>(m, ndt / 4), 256, 256);
  if (C.d_asm_fsrc.n) {
    if (!C.prep_part.p) { C.prep_part.alloc(256); C.prep_ticket.alloc(1); SDM_HIP_CHECK(hipMemsetAsync(C.prep_ticket.p, 0, sizeof(int), st)); }
    SDM_KLAUNCH(P, k_begin_factor, dim3(nprep + grid1d(C.fsize, 256)), dim3(256), 0, pa, nprep, C.prep_part.p, C.prep_ticket.p, C.fronts.p, C.d_asm_fsrc.p, (int64_t)C.fsize);
  } else {
    SDM_HIP_CHECK(hipMemsetAsync(C.fronts.p, 0, (size_t)C.fsize * sizeof(double), st));
    SDM_KLAUNCH(P, k_assemble, dim3(grid1d(C.nnzL, 256)), dim3(256), 0, C.fronts.p, P->ada_val.p, C.d_asm_src.p,
               C.d_asm_dst.p, (int64_t)C.nnzL, C.ub.p);
    SDM_KLAUNCH(P, k_prep_pivots, dim3(nprep), dim3(256), 0, pa);
  }
  C.pars_canceltol = canceltol; C.pars_maxu = maxu; C.pars_abstol = abstol; C.pars_use_absd = use_absd;
  {
    PanelCtx c = {};
    c.lb = C.lb.p; c.ubp = C.ub.p; c.pivstat = C.pivstat.p; c.pivval = C.pivval.p; c.colbuf = C.colbuf.p; c.ada = P->ada_val.p;
    c.asm_src = C.d_asm_src.p; c.Ljc = C.d_Ljc.p; c.mtot = m;
    if (!C.panel_ctx.p || memcmp(&c, &C.panel_ctx_host, sizeof(c)) != 0) {
      if (!C.panel_ctx.p) C.panel_ctx.alloc(1);
      C.panel_ctx_host = c;
      SDM_HIP_CHECK(hipMemcpyAsync(C.panel_ctx.p, &C.panel_ctx_host, sizeof(c), hipMemcpyHostToDevice, st));
    }
  }
  SDM_HIP_CHECK(hipGetLastError());
}
// levels l0 .. l1-1: children's update matrices into the fronts of the level (extend-add), then -- unless extend_only -- its LDL'
void chol_levels(sdm_plan *P, int l0, int l1, bool extend_only, int pan0, int pan1) {
  // pan0 .. pan1-1: of every level's panel launches only these (sdm_plan_blkchol_panels: ranks that factor one front block-cyclically
  // exchange the finished panel between two launches); the extend-add of a level runs with its first launch
  CholPlan &C = P->chol;
  hipStream_t st = P->stream; (void)st;                              // (used by the emulator's launches of the follower only)
  FrontTab tab = front_tab(C);
  const bool follow = C.follow;
  for (int l = std::max(l0, 0); l < std::min(l1, C.nlevels); l++) {
    const int *list = C.d_levlist.p + C.levptr[l];
    const int nfr = C.levptr[l + 1] - C.levptr[l];
    if (nfr == 0) continue;
    if (l > 0 && pan0 <= 0) SDM_KLAUNCH(P, k_extend_add, dim3(C.lev_T[l], nfr), dim3(256), 0, C.fronts.p, tab, list);
    if (extend_only) continue;
    if (C.lev_persist[l] && !C.front_disabled) {                     // the whole level in one launch (k_ldl_front)
      PersistTurn turn(P);
#ifdef SDM_EMU
      int maxnp = 0;
      for (int i = C.levptr[l]; i < C.levptr[l + 1]; i++) maxnp = std::max(maxnp, (C.sn_ns[C.levlist[i]] + NB - 1) / NB);
      if (emu_concurrent()) {
        emu_group_begin();
        // the kernel exactly as the GPU runs it (phase 0: everything carried in LDS, the fused row solve of the chain workgroup,
        // the data-tagged hand-over): one process per workgroup, all at once (tests/hipemu: emu_launch_concurrent)
        SDM_KLAUNCH_CONCURRENT(P, k_ldl_front, dim3(C.lev_maxT[l] + C.lev_ntw[l], nfr), dim3(LDL_THREADS), FRONT_LDS, C.fronts.p, C.frontsT.p, tab, list, C.d.p,
                               C.lb.p, C.ub.p, C.pivstat.p, C.pivval.p, C.panel_ctx.p, C.front_cnt.p,
                               C.diag_cnt.p, 0, 0, C.lev_maxT[l], C.tmo.dev(), FollowArgs{nullptr, nullptr, nullptr, C.lev_maxT[l] + C.lev_ntw[l]});
        if (follow) solve_follow(P, l, st);                          // beside it, polling its counters -- as the last workgroups of the launch do on the device
        emu_group_end();
        continue;
      }
      for (int step = 0; step < maxnp; step++)
        for (int phase = 1; phase <= 3; phase++)
          SDM_KLAUNCH(P, k_ldl_front, dim3(C.lev_maxT[l] + C.lev_ntw[l], nfr), dim3(LDL_THREADS), FRONT_LDS, C.fronts.p, C.frontsT.p, tab, list, C.d.p,
                      C.lb.p, C.ub.p, C.pivstat.p, C.pivval.p, C.panel_ctx.p, C.front_cnt.p,
                      C.diag_cnt.p, phase, step, C.lev_maxT[l], C.tmo.dev(), FollowArgs{nullptr, nullptr, nullptr, C.lev_maxT[l] + C.lev_ntw[l]});
      if (follow) solve_follow(P, l, st);                            // (workgroups run one after the other here: behind = after)
      if (emu_take_injected_timeout()) *(volatile int *)C.tmo.host = 1;   // (tests: as if a workgroup of this launch had given up waiting)
#else
      // the inverse of the level's fronts is built BEHIND their factorisation by the last workgroups of the same launch (k_ldl_front,
      // sdm_follow.h: they poll the progress counters; both kinds of workgroup fit the device together: solve_build)
      FollowArgs fa = {nullptr, nullptr, nullptr, C.lev_maxT[l] + C.lev_ntw[l]};
      if (follow) { fa.S = C.S.p; fa.STr = C.ST.p; fa.sb_g = C.sb_g.p; C.growth_used = C.growth_max; }
      SDM_KLAUNCH(P, k_ldl_front, dim3(C.lev_maxT[l] + C.lev_ntw[l] + (follow ? C.lev_followT[l] : 0), nfr), dim3(LDL_THREADS), FRONT_LDS, C.fronts.p, C.frontsT.p, tab, list, C.d.p,
                  C.lb.p, C.ub.p, C.pivstat.p, C.pivval.p, C.panel_ctx.p, C.front_cnt.p,
                  C.diag_cnt.p, 0, 0, C.lev_maxT[l], C.tmo.dev(), fa);
#endif
      continue;
    }
    for (int li = C.lev_first_launch[l]; li < C.lev_first_launch[l + 1]; li++) {
      const LevelLaunch &L = C.launches[li];
      if (L.panel < pan0 || L.panel >= pan1) continue;
      // ONE launch per panel: diagonal block (+ tile 0 of the previous panel's update), the row solves and the rest of
      // the previous update (see k_ldl_panel).  The emulator runs it in two phases (workgroups are sequential there).
#ifdef SDM_EMU
      if (emu_concurrent() && (1 + L.ride_wgs) * L.nactive <= 200)   // as on the device: one launch, the roles wait for each other (one process per workgroup)
        SDM_KLAUNCH_CONCURRENT(P, k_ldl_panel, dim3(1 + L.ride_wgs, L.nactive), dim3(LDL_THREADS), PANEL_LDS_RIDE, C.fronts.p, C.frontsT.p, tab, list,
                               L.panel, C.d.p, C.panel_ctx.p, C.upd_cnt.p, C.diag_cnt.p, 1, 0, C.tmo.dev(), C.own);
      else
      for (int phase = 1; phase <= 2; phase++)
#else
      const int phase = 0;
#endif
        SDM_KLAUNCH(P, k_ldl_panel, dim3(1 + L.ride_wgs, L.nactive), dim3(LDL_THREADS), PANEL_LDS_RIDE, C.fronts.p, C.frontsT.p, tab, list,
                    L.panel, C.d.p, C.panel_ctx.p, C.upd_cnt.p, C.diag_cnt.p, 1, phase, C.tmo.dev(), C.own);
      if (L.lasttiles > 0)                                           // supernodes that end with this panel and have rows beyond
        SDM_KLAUNCH(P, k_ldl_update, dim3(L.lasttiles, L.nactive), dim3(256), 0, C.fronts.p, tab, list, L.panel, C.d.p, 1);
    }
  }
  SDM_HIP_CHECK(hipGetLastError());
}
// ---- the record of a finished panel that travels between block-cyclic ranks beside the panel's columns of the front: d, lb, pivval, pivstat
// (as doubles) of its 64 columns, the front's two progress counters, the transposed copy DT of its diagonal block  (3 NB + NB + 2 + NB NB doubles)
__global__ void k_panel_record(double *rec, double *d, double *lb, double *pivval, int *pivstat, int *upd_cnt, int *diag_cnt, double *DT, int first, int k0, int kb,
                               int s, int64_t toff, int unpack) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
  for (int e = tid; e < 4 * NB + 2 + NB * NB; e += nth) {
    double *slot = nullptr; int *islot = nullptr;
    if (e < NB) { if (e < kb) slot = d + first + k0 + e; }
    else if (e < 2 * NB) { if (e - NB < kb) slot = lb + first + k0 + e - NB; }
    else if (e < 3 * NB) { if (e - 2 * NB < kb) slot = pivval + first + k0 + e - 2 * NB; }
    else if (e < 4 * NB) { if (e - 3 * NB < kb) islot = pivstat + first + k0 + e - 3 * NB; }
    else if (e == 4 * NB) islot = upd_cnt + s;
    else if (e == 4 * NB + 1) islot = diag_cnt + s;
    else slot = DT + toff + (int64_t)(k0 / NB) * NB * NB + (e - 4 * NB - 2);
    if (slot) { if (unpack) *slot = rec[e]; else rec[e] = *slot; }
    else if (islot) { if (unpack) *islot = (int)rec[e]; else rec[e] = (double)*islot; }
    else if (!unpack) rec[e] = 0.0;
  }
}
void chol_panel_record(sdm_plan *P, int panel, int unpack) {
  CholPlan &C = P->chol;
  if (C.nsuper != 1) throw std::runtime_error("panel records: block-cyclic factorisation is for ONE dense front");
  if (C.panelrec.n < (size_t)(4 * NB + 2 + NB * NB)) C.panelrec.alloc((size_t)(4 * NB + 2 + NB * NB));
  const int ns = C.sn_ns[0], k0 = panel * NB, kb = std::min(NB, ns - k0);
  if (k0 >= ns) throw std::runtime_error("panel record: no such panel");
  SDM_KLAUNCH(P, k_panel_record, dim3(8), dim3(256), 0, C.panelrec.p, C.d.p, C.lb.p, C.pivval.p, C.pivstat.p, C.upd_cnt.p, C.diag_cnt.p, C.frontsT.p, C.sn_first[0], k0, kb, 0,
              C.sn_toff[0], unpack);
}
void chol_end(sdm_plan *P) {
  if (!P->chol.follow) solve_prepare(P, /*sb_g_is_zero=*/true);      // inverses of the diagonal super-blocks for the solves (else: built behind the levels)
  SDM_HIP_CHECK(hipGetLastError());
  P->factored = true;
}
void chol_factor(sdm_plan *P, const double canceltol, const double maxu, const double abstol, int use_absd) {
  chol_begin(P, canceltol, maxu, abstol, use_absd);
  chol_levels(P, 0, P->chol.nlevels, false, 0, 1 << 30);
  chol_end(P);
}

// Non-zero when a workgroup gave up waiting for another one inside a launch (never expected; the results of that
// factorisation are then unusable: the plan is marked "not factored", so the solves refuse to run on it).  Reads and
// clears the plan's own flag (pinned host memory the kernels of THIS plan write to); call after a stream synchronise.
int chol_wait_timeouts(sdm_plan *P) {
  CholPlan &C = P->chol;
  if (!C.tmo.host) return 0;
  const int n = *(volatile int *)C.tmo.host;
  if (n) {
    *(volatile int *)C.tmo.host = 0; P->factored = false;
    // k_ldl_front needs ALL its workgroups resident and they wait for each other: another process on the same device (or a partition
    // smaller than the one the plan was built on) can keep some of them out until the bounded waits give up.  The launch-per-panel
    // path only ever waits for workgroups dispatched earlier, so this plan's later factorisations take that one.
    C.front_disabled = true; C.follow = false;
  }
  return n;
}

void chol_extract(sdm_plan *P, double *d_Lpr_out) {
  CholPlan &C = P->chol;
  SDM_KLAUNCH(P, k_extract, dim3(grid1d(C.nnzL, 256)), dim3(256), 0, d_Lpr_out, C.fronts.p, C.d_asm_dst.p,
             (int64_t)C.nnzL);
}

void chol_load_factor(sdm_plan *P, const double *h_Lpr, const double *h_d) {
  CholPlan &C = P->chol;
  if (h_d) {                                                         // an externally computed factor with its d: no pivot report
    const int m = (int)C.m;
    SDM_HIP_CHECK(hipMemcpyAsync(C.d.p, h_d, (size_t)m * sizeof(double), hipMemcpyHostToDevice, P->stream));
    SDM_HIP_CHECK(hipMemsetAsync(C.pivstat.p, 0, (size_t)m * sizeof(int), P->stream));
    SDM_HIP_CHECK(hipMemsetAsync(C.lb.p, 0, (size_t)m * sizeof(double), P->stream));
    SDM_HIP_CHECK(hipStreamSynchronize(P->stream));                  // h_d may be pageable
  }
  DevBuf<double> tmp;
  tmp.upload(h_Lpr, (size_t)C.nnzL);
  SDM_HIP_CHECK(hipMemsetAsync(C.fronts.p, 0, (size_t)C.fsize * sizeof(double), P->stream));    // padding rows / Schur parts: defined
  SDM_HIP_CHECK(hipMemsetAsync(C.frontsT.p, 0, (size_t)C.tsize * sizeof(double), P->stream));
  SDM_KLAUNCH(P, k_load_factor, dim3(grid1d(C.nnzL, 256)), dim3(256), 0, C.fronts.p, C.frontsT.p, tmp.p, C.d_asm_dst.p,
              C.d_asm_dstT.p, (int64_t)C.nnzL);
  solve_prepare(P, false);
  SDM_HIP_CHECK(hipStreamSynchronize(P->stream));
  P->factored = true;
}

void vec_gather(sdm_plan *P, double *dst, const double *src, bool forward) {
  const int m = (int)P->chol.m;
  SDM_KLAUNCH(P, k_gather_perm, dim3((m + 255) / 256), dim3(256), 0, dst, src, P->chol.d_perm.p, m, forward ? 1 : 0);
}
void vec_divd(sdm_plan *P, double *v) {
  const int m = (int)P->chol.m;
  SDM_KLAUNCH(P, k_divd, dim3((m + 255) / 256), dim3(256), 0, v, solve_d(P), m);
}

}  // namespace sdm

#if defined(SDM_PHASES) && !defined(SDM_EMU)
// tools-only build (python -m sedumi_amd.build --phases): read / reset the in-kernel phase clocks of this file
extern "C" int sdm_debug_trace_chol(long long *out2048) {
  return hipMemcpyFromSymbol(out2048, HIP_SYMBOL(sdm_trace_buf), 2048 * sizeof(long long)) != hipSuccess;
}
#endif
// (tests) the update work of panel launch q of a front with ns columns and ms rows that has row-solve workgroups: per item five ints
// {I, J, act, first panel, panels} (tile_sched_item; tile coordinates relative to tile column q); returns the number of items, -1 if the
// launch has no row-solve workgroups (its tiles are the eager ones of panel_role_tiles: every tile of the trailing matrix but those of column 0)
extern "C" int sdm_debug_tile_items(int ns, int ms, int q, int *out, int cap) {
  using namespace sdm;
  const int kb = std::min(NB, ns - q * NB), nrows = ms - (q * NB + kb);
  if (nrows <= TRSM_ROWS) return -1;
  const TileSched sc = tile_sched(ns, ms, q);
  const int n = tile_sched_items(sc);
  for (int u = 0; u < n && u < cap; u++) tile_sched_item(sc, q, u, out[5 * u], out[5 * u + 1], out[5 * u + 2], out[5 * u + 3], out[5 * u + 4]);
  return n;
}
#if defined(SDM_PHASES) && !defined(SDM_EMU)
extern "C" int sdm_debug_phases_chol(unsigned long long *out32, int reset) {
  if (out32 && hipMemcpyFromSymbol(out32, HIP_SYMBOL(sdm_phase_acc), 32 * sizeof(unsigned long long)) != hipSuccess) return 1;
  if (reset) { unsigned long long z[32] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(sdm_phase_acc), z, sizeof(z)) != hipSuccess) return 1; }
  return 0;
}
#endif

// sdm_solve.hip -- triangular solves  y = L \ b(perm),  y(perm) = L' \ b  (fwblkslv.c:77-134, bwblkslv.c:73-125)
// for gfx950, built around EXPLICIT INVERSES of wide diagonal super-blocks.
//
// The reference substitutes column by column -- a chain of m dependent steps.  A single right-hand side leaves a
// GPU nothing to batch over (the four solves of an IPM iteration depend on each other, wrapPcg.m:56-59), so the chain
// itself has to go.  The columns of every front are cut into super-blocks of W = CholPlan::sbw columns (256 .. 2048:
// the power of two that covers the widest front, so control07's 666-column front is ONE block and MAXCUT-4000's front
// is two), and after every factorisation (solve_prepare) the diagonal block L_PP of every super-block is inverted
// explicitly into the arena S:
//   * 128-column leaves: 32x32 by substitution in registers, then two levels of  X21 = -inv(C) B inv(A)  on the FP64
//     matrix cores, all inside one workgroup (k_sinv128);
//   * combine levels 256, 512, ... W: the same identity on 64x64 product tiles (k_stile), two dependent stages per
//     level (T = B inv(A), then X21 = -inv(C) T); small problems run all of it as ONE launch with completion
//     counters between the stages (k_sprep).
// Then, per etree level and super-block P,
//   forward   y_P = inv(L_PP) t_P                       one GEMV launch (k_sfw_diag), no dependency inside it,
//             t_R -= L(R, P) y_P  for the rows R beyond  one GEMV launch (k_sfw_step), read from the factor itself
//                                                        (own rows and the rows of the ancestors alike),
//   backward  x_Q = inv(L_QQ)' v_Q ,  v_C -= L(Q, C)' x_Q  for the columns C left of Q: the mirror image,
// i.e. 2 nsb - 1 dependent launches per sweep of a front of nsb super-blocks -- ONE for a front that fits a block --
// each a plain HBM-streaming matrix-vector product over many workgroups (16 rows or columns x up to 1024 columns or
// rows in flight per workgroup).
//
// Never-fail pivoting admits multipliers up to maxu = 5e5 (blkchol2.c:114-161), so an explicit inverse can be
// ill-conditioned.  The growth of every super-block is therefore measured when it is inverted
// (max|inv(L_PP)| * max|L_PP|); a block beyond CholPlan::growth_max is solved by substitution against the factor by
// one workgroup of the same launch -- per block, decided on the device, no host round trip.  (Measured on whole
// runs of control07.mat: the growth of the 666-wide block stays within 1.6x of that of its 256-wide sub-blocks in
// all 40 iterations, and both pass 1e4 in the last six)  Everything is deterministic (fixed summation
// orders, no atomics on data).
#include "sdm_follow.h"
#include <algorithm>

namespace sdm {

constexpr int SPREP_MAX_ITEMS = 256;                                  // k_sprep: one workgroup per item, all resident (one per CU)
constexpr int MC_N = 32, MC_STRIDE = 32, MC_SET = MC_N * MC_STRIDE;   // counters of the merged sweep launches (merged_count below)
constexpr int GRPW = 1024;       // columns (forward) / rows (backward) of a slab product in flight at a time: 32 16-byte loads per work-item

// ---------------------------------------------------------------- host tables
static void follow_decide(sdm_plan *P);
#ifndef SDM_EMU
__global__ void k_sinv_follow(const double *F, const double *DT, double *S, double *STr, FrontTab tab, const int *list, int *front_cnt, const int *diag_cnt,
                              unsigned long long *sb_g, int *tmo);
#endif
void solve_build(sdm_plan *P) {
  CholPlan &C = P->chol;
  const int nsuper = (int)C.nsuper;
  int W = C.sbw_req;
  if (W == 0) { W = SBW_MIN; while (W < C.maxns && W < SBW_MAX) W *= 2; }
  C.sbw = W;
  // a new solve: no ill-conditioned block met yet (sweeps of the previous symbolic factor still in flight would write their notes
  // after this reset: drained first -- set_chol happens once per solve)
  if (C.noted.host) SDM_HIP_CHECK(hipStreamSynchronize(P->stream));
  C.noted.ensure(); C.noted.host[0] = C.noted.host[1] = 0; C.refine_on = false; C.sweep_seq = 0;
  C.sn_soff.assign(nsuper, 0); C.sn_sld.assign(nsuper, 0); C.sn_sboff.assign(nsuper, 0);
  std::vector<int> i128;
  std::vector<std::vector<int>> stage(2 * SINV_MAXLEV);              // combine tiles per stage st = 2 * level + (0: T, 1: X)
  std::vector<std::vector<int>> stage_w(2 * SINV_MAXLEV);            // and the number of K steps of each
  int64_t soff = 0; int sb = 0;
  for (int s = 0; s < nsuper; s++) {
    const int ns = C.sn_ns[s];
    // leading dimension of the front's inverse blocks: a multiple of 16 (whole 128-byte lines per 16-row slab), never a
    // multiple of 256 doubles (columns 2 KB-aligned to each other would land on the same memory channels)
    int sld = (std::min(ns, W) + 15) & ~15;
    if (sld % 256 == 0) sld += 16;
    C.sn_soff[s] = soff; C.sn_sld[s] = sld; C.sn_sboff[s] = sb;
    soff += (int64_t)sld * ns;
    const int nsb = (ns + W - 1) / W;
    const bool act = C.sn_active.empty() || C.sn_active[s] != 0;     // (supernodes of other ranks: a place in the arena, no work)
    for (int h = 0; act && 128 * h < ns; h++) { i128.push_back(s); i128.push_back(h); i128.push_back(0); i128.push_back(0); }
    for (int Pb = 0; act && Pb < nsb; Pb++) {
      const int nb = std::min(W, ns - Pb * W);
      int prev = (nb + 127) / 128;                                    // what stage 0 waits for: the leaves of this super-block
      for (int lev = 0; lev < SINV_MAXLEV; lev++) {
        const int h = 128 << lev;
        if (h >= nb) break;
        int cnt = 0;
        for (int t = 0; t < 2; t++) {
          std::vector<int> &dst = stage[2 * lev + t];
          for (int pi = 0; pi * 2 * h + h < nb; pi++) {
            const int nc = std::min(h, nb - pi * 2 * h - h);
            for (int I = 0; 64 * I < nc; I++)
              for (int J = 0; 64 * J < h; J++) {
                const int it[8] = {s, Pb, lev, pi, I, J, t, prev};
                dst.insert(dst.end(), it, it + 8);
                stage_w[2 * lev + t].push_back(t == 0 ? h / 64 - J : std::min(I + 1, (nc + 63) / 64));   // its K steps (stile_body)
                if (t == 0) cnt++;
              }
          }
          prev = cnt;                                                  // T and X stages of a level have the same tiles
        }
      }
    }
    sb += nsb;
  }
  C.ssize = soff; C.nsbtot = sb;
  std::vector<int> items;
  C.stage_ptr.assign(2 * SINV_MAXLEV + 1, 0);
  for (int st = 0; st < 2 * SINV_MAXLEV; st++) {
    C.stage_ptr[st] = (int)items.size() / 8;
    // Longest items first.  The products are triangular (1 .. h/64 K steps per tile) and a launch of more tiles than fit the device
    // at once lasts as long as whatever is dispatched last: in the order the tiles are generated (long ones last in stage X) the two
    // 496-tile stages of MAXCUT-4000 took 107 and 103 us, sorted 90 and 89 (profiles/r07_inverse_tile_variants.txt).  Measured and
    // not kept: workgroups taking the tiles in pairs, longest with shortest (88 us, and the small stages slower: a workgroup alone on
    // its CU needs 4.6 us per K step, two on a CU 6.4 us each), and operand blocks two K steps ahead in registers (a few us, at the
    // price of the second workgroup per CU) -- a K step is 64 KB of operands at the ~16 GB/s a CU gets when all CUs stream.
    const std::vector<int> &w = stage_w[st];
    std::vector<int> ord(w.size());
    for (size_t i = 0; i < ord.size(); i++) ord[i] = (int)i;
    std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return w[a] > w[b]; });
    for (int i : ord) items.insert(items.end(), stage[st].begin() + 8 * (size_t)i, stage[st].begin() + 8 * (size_t)i + 8);
  }
  C.stage_ptr[2 * SINV_MAXLEV] = (int)items.size() / 8;
  C.n_i128 = (int)i128.size() / 4; C.n_items = (int)items.size() / 8;
  C.l_i128.upload(i128); C.l_items.upload(items);
  C.d_soff.upload(C.sn_soff); C.d_sld.upload(C.sn_sld); C.d_sboff.upload(C.sn_sboff);
  const size_t sz = (size_t)std::max<int64_t>(soff, 1);
  C.S.alloc(sz); C.ST.alloc(sz); C.Tarena.alloc(C.n_items ? sz : 1);
  SDM_HIP_CHECK(hipMemset(C.S.p, 0, sz * sizeof(double)));            // upper triangles stay zero for good
  SDM_HIP_CHECK(hipMemset(C.ST.p, 0, sz * sizeof(double)));           // (here: the lower ones)
  // fronts of several super-blocks: transposed copy of the rows of L below each super-block (forward step launches)
  {
    C.sn_ltoff.assign(nsuper, 0);
    std::vector<int> lt;
    int64_t ltoff = 0;
    for (int s = 0; s < nsuper; s++) {
      const int ns = C.sn_ns[s], nsb = (ns + W - 1) / W;
      C.sn_ltoff[s] = ltoff;
      if (!(C.sn_active.empty() || C.sn_active[s] != 0)) continue;
      for (int Pb = 0; Pb + 1 < nsb; Pb++) {
        const int nr = ns - (Pb + 1) * W;
        for (int I = 0; 64 * I < nr; I++)
          for (int J = 0; 64 * J < W; J++) { lt.push_back(s); lt.push_back(Pb); lt.push_back(I); lt.push_back(J); }
        ltoff += (int64_t)nr * W;
      }
    }
    C.n_lt = (int)lt.size() / 4;
    C.l_lt.upload(lt); C.d_ltoff.upload(C.sn_ltoff);
    C.LT.alloc((size_t)std::max<int64_t>(ltoff, 1));
  }
  C.xfin.alloc((size_t)std::max<sdm_int>(C.m, 1)); C.zdiv.alloc((size_t)std::max<sdm_int>(C.m, 1));
  // growth records (2 words per super-block), then the completion counters of k_sprep (SPREP_NCNT ints per super-block)
  const size_t gw = (size_t)std::max(sb, 1) * (2 + SPREP_NCNT / 2);
  C.sb_g.alloc(gw);
  SDM_HIP_CHECK(hipMemset(C.sb_g.p, 0, gw * sizeof(unsigned long long)));
  C.sweep_cnt.alloc(2 * MC_SET);                                    // (the merged sweep launches' counters: merged_count)
  SDM_HIP_CHECK(hipMemset(C.sweep_cnt.p, 0, 2 * MC_SET * sizeof(int)));
  // levels
  C.slev.assign(C.nlevels, SolveLevel());
  for (int l = 0; l < C.nlevels; l++) {
    SolveLevel &L = C.slev[l];
    L.nfronts = C.levptr[l + 1] - C.levptr[l];
    for (int i = C.levptr[l]; i < C.levptr[l + 1]; i++) {
      const int s = C.levlist[i], ns = C.sn_ns[s], ms = C.sn_ms[s];
      L.maxns = std::max(L.maxns, ns); L.maxms = std::max(L.maxms, ms);
      if (C.childptr[s + 1] > C.childptr[s]) L.children = true;
      if (ms > ns) L.below = true;
    }
    L.nsb = (L.maxns + W - 1) / W;
    L.slabs_fw.assign(L.nsb, 0);
    for (int i = C.levptr[l]; i < C.levptr[l + 1]; i++) {
      const int s = C.levlist[i], ns = C.sn_ns[s], ms = C.sn_ms[s];
      for (int Pb = 0; Pb * W < ns; Pb++)                              // (slabs of the rows BELOW the supernode; its own later rows: k_sfw_rows)
        if (ms > ns) L.slabs_fw[Pb] = std::max(L.slabs_fw[Pb], (ms - (ns & ~1) + SROWS - 1) / SROWS);
    }
  }
  follow_decide(P);
}

// Can the inverses be built BEHIND the factorisation (k_sinv_follow)?  Every level must be a k_ldl_front level, every front
// one super-block, and the workgroups of both kernels of a level must fit the device together, one per compute unit
// (whichever of the two the hardware dispatches first, nobody may be kept out by workgroups that wait).
static void follow_decide(sdm_plan *P) {
  CholPlan &C = P->chol;
  C.follow = false;
  C.lev_followT.assign(C.nlevels, 0);
  if (C.nlevels == 0 || C.maxns > C.sbw || C.front_disabled) return;
  int ncu = 1 << 20;
#ifndef SDM_EMU
  SDM_HIP_CHECK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, P->device));
  {
    // (one workgroup of either kernel per compute unit is what the count below assumes: the follower must fit at least that)
    int per_cu = 0;
    SDM_HIP_CHECK(hipFuncSetAttribute((const void *)k_sinv_follow, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TILE_LDS));
    SDM_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k_sinv_follow, ST, TILE_LDS));
    if (per_cu < 1) return;
  }
#endif
  for (int l = 0; l < C.nlevels; l++) {
    if (!C.lev_persist[l]) return;
    const int nfr = C.levptr[l + 1] - C.levptr[l];
    int Tn = 0;
    for (int i = C.levptr[l]; i < C.levptr[l + 1]; i++) Tn = std::max(Tn, (C.sn_ns[C.levlist[i]] + 63) / 64);
    C.lev_followT[l] = Tn * (Tn + 1) / 2;
    if ((int64_t)nfr * (C.lev_followT[l] + C.lev_maxT[l] + C.lev_ntw[l]) > ncu - ncu / 8) return;
  }
  C.follow = true;
}

// Leaves.  One workgroup per 128-column block h of a front, bottom-up, everything in LDS / registers:
//   32x32  each of the four wavefronts inverts one 32x32 unit lower triangular diagonal block by columns (lane j owns
//          column j of the inverse in registers; the entries of L come as broadcast LDS reads at compile-time offsets);
//   64x64  X10 = -inv(A11) (A10 inv(A00)) for the two 64-column blocks A and C (matrix cores, two wavefronts each);
//   128    X21 = -inv(C) (B inv(A)) on the FP64 matrix cores, B = L(C rows, A columns) requested at the very start.
// Results go to S; max|inv| and max|L| to sb_g (growth check).
template <bool WT>
__device__ __forceinline__ void sinv128_body(char *smem, const double *__restrict__ F, double *__restrict__ S, double *__restrict__ STr, const FrontTab &tab,
                                             const int *it, unsigned long long *sb_g, int W) {
  double *bufA = (double *)smem, *bufC = bufA + 64 * TP, *bufB = bufC + 64 * TP, *bufT = bufB + 64 * TP;
  const int s = it[0], h = it[1];
  const int ns = tab.ns[s], ld = tab.ld[s], sld = tab.sld[s];
  const double *Fs = F + tab.foff[s];
  const int k0 = 128 * h, nbA = min(64, ns - k0), nbC = max(0, min(64, ns - k0 - 64));
  const int Pb = k0 / W, kl = k0 - Pb * W;                          // super-block of the leaf, its first column inside it
  double *Ss = S + tab.soff[s] + (int64_t)Pb * W * sld;
  double *Ts = STr + tab.soff[s] + (int64_t)Pb * W * sld;           // the transposed copy: Ts[r*sld + c] = inverse(r, c)
  unsigned long long *gP = sb_g + 2 * (tab.sboff[s] + Pb);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  double vB[SPT];
  SDM_PHASE_BEGIN();
  if (nbC > 0) stage_colmajor_load(vB, Fs + (int64_t)k0 * ld + k0 + 64, ld, nbC, 64, tid);     // B(row, k) = L(k0+64+row, k0+k)
  // raw strictly lower triangles, column-major: rawA[k*TP + i] = L(k0+i, k0+k) (bufT), rawC likewise (bufB); the
  // destination buffers start as zero
  double *rawA = bufT, *rawC = bufB;
  double lmx = 0.0;
  {
    double va[SPT], vc[SPT];
    const int i = tid & 63, kq = tid >> 6;
#pragma unroll
    for (int j = 0; j < SPT; j++) {
      const int k = kq + (ST / 64) * j;
      va[j] = Fs[(int64_t)(k0 + min(k, nbA - 1)) * ld + k0 + min(i, nbA - 1)];
      vc[j] = nbC > 0 ? Fs[(int64_t)(k0 + 64 + min(k, nbC - 1)) * ld + k0 + 64 + min(i, nbC - 1)] : 0.0;
    }
#pragma unroll
    for (int j = 0; j < SPT; j++) {
      const int k = kq + (ST / 64) * j;
      const double a = (i > k && i < nbA) ? va[j] : 0.0, c = (i > k && i < nbC) ? vc[j] : 0.0;
      rawA[k * TP + i] = a; rawC[k * TP + i] = c;
      bufA[k * TP + i] = 0.0; bufC[k * TP + i] = 0.0;
      lmx = fmax(lmx, fmax(fabs(a), fabs(c)));
    }
  }
  __syncthreads();
  SDM_PHASE(0);
  inv64_pair(rawA, rawC, bufA, bufC, wave, lane, gP);
  SDM_PHASE(3);
  __syncthreads();                                                  // bufA = inv(A) (B operand), bufC = inv(C) (A operand); raw buffers free
  if (nbC > 0) lmx = fmax(lmx, stage_colmajor_store(bufB, vB, nbC, 64, tid));
  wave_atomic_max(gP + 1, lmx, lane);
  // inverses to S (lower triangles incl. the unit diagonal; the upper triangles of S are zero and stay zero)
  for (int e = tid; e < 64 * 64; e += ST) {
    const int i = e & 63, j = e >> 6;
    if (i >= j && i < nbA) { if (WT) sdm_store_wt(&Ss[(int64_t)(kl + j) * sld + kl + i], bufA[i * TP + j]); else Ss[(int64_t)(kl + j) * sld + kl + i] = bufA[i * TP + j]; }
    if (i >= j && i < nbC) { if (WT) sdm_store_wt(&Ss[(int64_t)(kl + 64 + j) * sld + kl + 64 + i], bufC[j * TP + i]); else Ss[(int64_t)(kl + 64 + j) * sld + kl + 64 + i] = bufC[j * TP + i]; }
  }
  for (int e = tid; e < 64 * 64; e += ST) {                           // transposed copy: consecutive work-items on consecutive columns j
    const int j = e & 63, i = e >> 6;
    if (i >= j && i < nbA) { if (WT) sdm_store_wt(&Ts[(int64_t)(kl + i) * sld + kl + j], bufA[i * TP + j]); else Ts[(int64_t)(kl + i) * sld + kl + j] = bufA[i * TP + j]; }
    if (i >= j && i < nbC) { if (WT) sdm_store_wt(&Ts[(int64_t)(kl + 64 + i) * sld + kl + 64 + j], bufC[j * TP + i]); else Ts[(int64_t)(kl + 64 + i) * sld + kl + 64 + j] = bufC[j * TP + i]; }
  }
  SDM_PHASE(4);
  if (nbC <= 0) return;
  __syncthreads();
  Acc22 acc;
  acc_zero(acc);
  mma_block(acc, bufB, bufA, wave, lane);                           // T = B inv(A)
  acc_to_lds_rowmajor(acc, bufT, wave, lane, 1.0);                  // bufT[k*TP + col] = T(k, col): a B operand
  __syncthreads();
  acc_zero(acc);
  mma_block(acc, bufC, bufT, wave, lane);                           // inv(C) T
  __syncthreads();                                                  // bufB is free: stage the result for coalesced stores
  acc_to_lds_rowmajor(acc, bufB, wave, lane, -1.0);
  __syncthreads();
  SDM_PHASE(5);
  const double gm = store_tile<WT>(Ss + (int64_t)kl * sld + kl + 64, sld, bufB, nbC, 64, tid);
  store_tile_T<WT>(Ts + (int64_t)(kl + 64) * sld + kl, sld, bufB, nbC, 64, tid);
  wave_atomic_max(gP, gm, lane);
  SDM_PHASE(6);
}
__global__ void __launch_bounds__(ST)
k_sinv128(const double *__restrict__ F, double *__restrict__ S, double *__restrict__ STr, FrontTab tab, const int *items, unsigned long long *sb_g, int W) {
  SDM_DYN_SMEM(smem);
  sinv128_body<false>(smem, F, S, STr, tab, items + 4 * blockIdx.x, sb_g, W);
}

// Combine levels.  Level lev joins the inverses of neighbouring column ranges of half width h = 128 << lev inside a
// super-block:  inv([A 0; B C]) = [inv(A) 0; -inv(C) B inv(A), inv(C)]  with A = columns a0 .. a0+h-1, C = the nc <= h
// columns behind them.  One 64x64 tile of one of the two products per item {s, Pb, lev, pair, I, J, stage, wait}:
//   stage 0  T(I, J)   =   sum_{K >= J} B(I, K) inv(A)(K, J)        B = L(C rows, A columns) from the factor; T into the scratch arena
//   stage 1  X21(I, J) = - sum_{K <= I} inv(C)(I, K) T(K, J)        into S
// (both triangular in K: only the 64-blocks that can be non-zero are multiplied).
template <bool WT>
__device__ __forceinline__ void stile_body(char *smem, const double *F, double *S, double *STr, double *T, const FrontTab &tab, const int *it,
                                           unsigned long long *sb_g, int W) {
  double *As = (double *)smem, *Bs = As + 64 * TP;
  const int s = it[0], Pb = it[1], lev = it[2], pi = it[3], I = it[4], J = it[5], stage = it[6];
  const int ns = tab.ns[s], ld = tab.ld[s], sld = tab.sld[s];
  const int P0 = Pb * W, nb = min(W, ns - P0);
  const int h = 128 << lev, a0 = pi * 2 * h, nc = min(h, nb - a0 - h);
  const double *Fs = F + tab.foff[s];
  double *Sb = S + tab.soff[s] + (int64_t)P0 * sld, *Tb = T + tab.soff[s] + (int64_t)P0 * sld;
  unsigned long long *gP = sb_g + 2 * (tab.sboff[s] + Pb);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const double *Ap, *Bp; double *Cp;
  int64_t lda, ldb;
  const int arows = min(64, nc - 64 * I);
  int kvalid;
  if (stage == 0) {
    kvalid = h - 64 * J;
    Ap = Fs + (int64_t)(P0 + a0 + 64 * J) * ld + P0 + a0 + h + 64 * I; lda = ld;
    Bp = Sb + (int64_t)(a0 + 64 * J) * sld + a0 + 64 * J; ldb = sld;
    Cp = Tb + (int64_t)(a0 + 64 * J) * sld + a0 + h + 64 * I;
  } else {
    kvalid = min(64 * (I + 1), nc);
    Ap = Sb + (int64_t)(a0 + h) * sld + a0 + h + 64 * I; lda = sld;
    Bp = Tb + (int64_t)(a0 + 64 * J) * sld + a0 + h; ldb = sld;
    Cp = Sb + (int64_t)(a0 + 64 * J) * sld + a0 + h + 64 * I;
  }
  Acc22 acc;
  acc_zero(acc);
  double lmx = 0.0;
  double va[SPT], vb[SPT];
  SDM_PHASE_BEGIN();
  stage_colmajor_load<WT>(va, Ap, lda, arows, kvalid, tid);
  stage_transposed_load<WT>(vb, Bp, ldb, kvalid, 64, tid);
  for (int kb = 0; kb < kvalid; kb += 64) {
    lmx = fmax(lmx, stage_colmajor_store(As, va, arows, kvalid - kb, tid));
    stage_transposed_store(Bs, vb, kvalid - kb, 64, tid);
    __syncthreads();
    if (kb + 64 < kvalid) {                                          // next K block: loads in flight during the products
      stage_colmajor_load<WT>(va, Ap + (int64_t)(kb + 64) * lda, lda, arows, kvalid - kb - 64, tid);
      stage_transposed_load<WT>(vb, Bp + kb + 64, ldb, kvalid - kb - 64, 64, tid);
    }
    mma_block(acc, As, Bs, wave, lane);
    __syncthreads();
  }
  SDM_PHASE(8 + 4 * stage);
  if (stage == 0) wave_atomic_max(gP + 1, lmx, lane);                // max |L| over the off-diagonal blocks of the super-block
  acc_to_lds_rowmajor(acc, As, wave, lane, stage == 0 ? 1.0 : -1.0);
  __syncthreads();
  const double gm = store_tile<WT>(Cp, sld, As, arows, 64, tid);
  if (stage == 1) {
    store_tile_T<WT>(STr + tab.soff[s] + (int64_t)P0 * sld + (int64_t)(a0 + h + 64 * I) * sld + a0 + 64 * J, sld, As, arows, 64, tid);
    wave_atomic_max(gP, gm, lane);                                   // max |inverse|
  }
  SDM_PHASE(9 + 4 * stage);
}
__global__ void __launch_bounds__(ST)
k_stile(const double *F, double *S, double *STr, double *T, FrontTab tab, const int *items, unsigned long long *sb_g, int W) {
  SDM_DYN_SMEM(smem);
  stile_body<false>(smem, F, S, STr, T, tab, items + 8 * blockIdx.x, sb_g, W);
}

// ---- all of the above in ONE launch for problems whose items fit the device at once (k_sprep): workgroups take the
// items in the order leaves, level 0 stage T, level 0 stage X, level 1 stage T, ... and wait on per-super-block completion
// counters instead of on launch boundaries.  Producers store write-through and count after their stores are
// acknowledged; consumers poll relaxed and read with sc1 loads.
// cnt[SPREP_NCNT * sb + 0] = finished leaves, [1 + st] = finished tiles of stage st (zeroed with sb_g by k_prep_pivots).
__global__ void __launch_bounds__(ST)
k_sprep(const double *F, double *S, double *STr, double *T, FrontTab tab, const int *l_i128, int n_i128, const int *l_items,
        unsigned long long *sb_g, int *cnt, int W, int *tmo) {
  SDM_DYN_SMEM(smem);
  const int b = blockIdx.x;
  if (b < n_i128) {
    const int *it = l_i128 + 4 * b;
    sinv128_body<true>(smem, F, S, STr, tab, it, sb_g, W);
    prep_done(cnt + SPREP_NCNT * (tab.sboff[it[0]] + (128 * it[1]) / W));
    return;
  }
  const int *it = l_items + 8 * (b - n_i128);
  const int st = 2 * it[2] + it[6];
  int *c = cnt + SPREP_NCNT * (tab.sboff[it[0]] + it[1]);
  prep_wait(c + st, it[7], tmo);
  stile_body<true>(smem, F, S, STr, T, tab, it, sb_g, W);
  prep_done(c + st + 1);
}

// ---- the inverse of a whole front BEHIND its factorisation: the body is sinv_follow_body (sdm_follow.h); this kernel runs it where
// the k_ldl_front launch does not carry the follower's workgroups itself (the emulator; captured graphs of older plans)
__global__ void __launch_bounds__(ST)
k_sinv_follow(const double *F, const double *DT, double *S, double *STr, FrontTab tab, const int *list, int *front_cnt, const int *diag_cnt,
              unsigned long long *sb_g, int *tmo) {
  SDM_DYN_SMEM(smem);
  const FollowDesc fd = follow_desc(tab, list, (int)blockIdx.y);
  sinv_follow_body(smem, (int)blockIdx.x, fd, F, DT, S, STr, front_cnt, diag_cnt, sb_g, tmo);
}

// ================================================================ substitution fallback for one super-block
// Rare path (growth check failed): L_PP y = r  /  L_PP' x = v  in place on the nb entries w (LDS) by ONE workgroup.
// Fs = front, (k0, k0) = position of the block.  Sd = 64*TP doubles of LDS.
constexpr int BSC = 32;          // columns per step of the substitution fallback (its LDS tile: BSC x (BSC + 1) doubles)
constexpr int BSP = BSC + 1;
// (the merged launches: W = 2048 and the tile of 32 are 24.8 KB -- six workgroups per CU; with a tile of 16, 18.6 KB, all eight the streaming role lives on)
constexpr int BSC_LEAN = 16;
#define SDM_MERGED_SMEM(W) ((size_t)((W) + BSC_LEAN * (BSC_LEAN + 1)) * sizeof(double))
#define SDM_DIAG_SMEM(W) ((size_t)((W) + BSC * BSP) * sizeof(double))   // (measured: the allocation costs the launch nothing, profiles/r08*)
template <int BSC>
__device__ __forceinline__ void block_solve_fw_inl(const double *Fs, int ld, int k0, int nb, double *w, double *Sd) {
  constexpr int BSP = BSC + 1;
  const int tid = threadIdx.x, lane = tid & 63;
  for (int kk = 0; kk < nb; kk += BSC) {
    const int kb = min(BSC, nb - kk);
    for (int e = tid; e < BSC * BSC; e += ST) {
      const int i = e % BSC, c = e / BSC;
      Sd[c * BSP + i] = (i > c && i < kb) ? Fs[(int64_t)(k0 + kk + c) * ld + k0 + kk + i] : 0.0;
    }
    __syncthreads();
    if (tid < 64) {
      double wi = lane < kb ? w[kk + lane] : 0.0;
      for (int k = 0; k < BSC; k++) wi -= (lane < BSC ? Sd[k * BSP + min(lane, BSC - 1)] : 0.0) * sdm_bcast_lane(wi, k);
      if (lane < kb) w[kk + lane] = wi;
    }
    __syncthreads();
    for (int r = kk + kb + tid; r < nb; r += ST) {
      double acc = 0.0;
      for (int c = 0; c < kb; c++) acc += Fs[(int64_t)(k0 + kk + c) * ld + k0 + r] * w[kk + c];
      w[r] -= acc;
    }
    __syncthreads();
  }
}
// (the merged sweep launches inline the substitutions -- no call: a kernel is allotted the registers of the fattest function it may call --,
// the diagonal-block launches call them)
__device__ __noinline__ void block_solve_fw(const double *Fs, int ld, int k0, int nb, double *w, double *Sd) { block_solve_fw_inl<BSC>(Fs, ld, k0, nb, w, Sd); }
template <int BSC>
__device__ __forceinline__ void block_solve_bw_inl(const double *Fs, int ld, int k0, int nb, double *w, double *Sd) {
  constexpr int BSP = BSC + 1;
  const int tid = threadIdx.x, lane = tid & 63;
  for (int kk = ((nb - 1) / BSC) * BSC; kk >= 0; kk -= BSC) {
    const int kb = min(BSC, nb - kk);
    for (int e = tid; e < BSC * BSC; e += ST) {
      const int i = e % BSC, c = e / BSC;
      Sd[c * BSP + i] = (i > c && i < kb) ? Fs[(int64_t)(k0 + kk + c) * ld + k0 + kk + i] : 0.0;
    }
    __syncthreads();
    if (tid < 64) {
      double xj = lane < kb ? w[kk + lane] : 0.0;
      for (int k = BSC - 1; k >= 0; k--) xj -= (lane < BSC ? Sd[min(lane, BSC - 1) * BSP + k] : 0.0) * sdm_bcast_lane(xj, k);      // L(k, lane), zero unless k > lane
      if (lane < kb) w[kk + lane] = xj;
    }
    __syncthreads();
    for (int c = tid; c < kk; c += ST) {
      double acc = 0.0;
      for (int r = 0; r < kb; r++) acc += Fs[(int64_t)(k0 + c) * ld + k0 + kk + r] * w[kk + r];
      w[c] -= acc;
    }
    __syncthreads();
  }
}
__device__ __noinline__ void block_solve_bw(const double *Fs, int ld, int k0, int nb, double *w, double *Sd) { block_solve_bw_inl<BSC>(Fs, ld, k0, nb, w, Sd); }

// ================================================================ slab products
// Every kernel below issues its matrix loads FIRST (registers), then fetches the vector it multiplies with (written by the
// previous launch) into LDS, then multiplies: the two memory latencies overlap.
// Forward: sum_c M(r, c) xs[c] for the SROWS rows rbase .. of one slab over ncols columns.  Work-item (p = tid & 7,
// g = tid >> 3) owns row pair p and the columns g, g+32, ...; 16-byte loads, NL of them in flight; fixed-order
// reduction over g.  rbase even; rows are clamped to the last valid pair (rlast = last valid row).
template <int NL>
__device__ __forceinline__ void slab_issue(sdm_double2 (&v)[NL], const double *M, int64_t ldm, int ncols, int rbase, int rlast) {
  const int tid = threadIdx.x, p = tid & 7, g = tid >> 3;
  const int r = min(rbase + 2 * p, rlast & ~1);
  const sdm_double2 *col = (const sdm_double2 *)(M + r);
  const int64_t ld2 = ldm >> 1;
#pragma unroll
  for (int j = 0; j < NL; j++) v[j] = col[(int64_t)min(g + 32 * j, ncols - 1) * ld2];
}
template <int NL>
__device__ __forceinline__ void slab_accum(const sdm_double2 (&v)[NL], int ncols, const double *xs, double &a0, double &a1) {
  const int g = threadIdx.x >> 3;
#pragma unroll
  for (int j = 0; j < NL; j++) {
    const int c = g + 32 * j;
    const double xc = c < ncols ? xs[c] : 0.0;
    a0 += v[j].x * xc; a1 += v[j].y * xc;
  }
}
// the whole product of a slab: groups of up to GRPW columns; fill() loads xs[0 .. ncols) (its barrier is ours) after the
// first group's matrix loads have been issued.  Result for row rbase + t in work-items t < SROWS.
template <class Fill>
__device__ __forceinline__ double fw_product(const double *M, int64_t ldm, int ncols, int rbase, int rlast, const double *xs, double *red, Fill fill) {
  const int tid = threadIdx.x, p = tid & 7, g = tid >> 3;
  double a0 = 0.0, a1 = 0.0;
  bool filled = false;
  for (int c0 = 0; c0 < ncols; c0 += GRPW) {
    const int n = min(GRPW, ncols - c0);
    const double *Mg = M + (int64_t)c0 * ldm;
    if (n <= 256) {
      sdm_double2 v[8]; slab_issue<8>(v, Mg, ldm, n, rbase, rlast);
      if (!filled) { fill(); __syncthreads(); filled = true; }
      slab_accum<8>(v, n, xs + c0, a0, a1);
    } else if (n <= 512) {
      sdm_double2 v[16]; slab_issue<16>(v, Mg, ldm, n, rbase, rlast);
      if (!filled) { fill(); __syncthreads(); filled = true; }
      slab_accum<16>(v, n, xs + c0, a0, a1);
    } else {
      sdm_double2 v[32]; slab_issue<32>(v, Mg, ldm, n, rbase, rlast);
      if (!filled) { fill(); __syncthreads(); filled = true; }
      slab_accum<32>(v, n, xs + c0, a0, a1);
    }
  }
  red[g * SROWS + 2 * p] = a0; red[g * SROWS + 2 * p + 1] = a1;
  __syncthreads();
  double sum = 0.0;
  if (tid < SROWS) {
#pragma unroll
    for (int q = 0; q < ST / 8; q++) sum += red[q * SROWS + tid];
  }
  return sum;
}
// Backward: sum_r M(rbase + r, c) xs[r] for the SROWS columns cbase .. of one slab over nrows rows: wavefront w owns 4
// columns, lanes run down the row pairs (contiguous 16-byte loads, 4 * NH in flight), wave reduction in a fixed order.
// rbase even.  Result for column cbase + 4*w + q in every lane of wavefront w as out[q].
template <int NH>
__device__ __forceinline__ void slabT_issue(sdm_double2 (&v)[4 * NH], const double *M, int64_t ldm, int cbase, int ncols, int rbase, int nrows) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int plast = max(((nrows + 1) >> 1) - 1, 0);
#pragma unroll
  for (int h = 0; h < NH; h++)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int c = min(cbase + 4 * wave + q, cbase + ncols - 1);
      v[4 * h + q] = ((const sdm_double2 *)(M + (int64_t)c * ldm + rbase))[min(lane + 64 * h, plast)];
    }
}
template <int NH>
__device__ __forceinline__ void slabT_accum(const sdm_double2 (&v)[4 * NH], int nrows, const double *xs, double (&acc)[4]) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int h = 0; h < NH; h++) {
    const int pi = lane + 64 * h;
    // rows beyond the range are dropped by selects on BOTH factors: a padding row of the front may hold anything
    const bool in0 = 2 * pi < nrows, in1 = 2 * pi + 1 < nrows;
    const double x0 = in0 ? xs[2 * pi] : 0.0, x1 = in1 ? xs[2 * pi + 1] : 0.0;
#pragma unroll
    for (int q = 0; q < 4; q++) acc[q] += (in0 ? v[4 * h + q].x : 0.0) * x0 + (in1 ? v[4 * h + q].y : 0.0) * x1;
  }
}
template <class Fill>
__device__ __forceinline__ void bw_product(const double *M, int64_t ldm, int cbase, int ncols, int rbase, int nrows, const double *xs, double (&out)[4], Fill fill) {
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  bool filled = false;
  for (int r0 = 0; r0 < nrows; r0 += GRPW) {
    const int n = min(GRPW, nrows - r0);
    if (n <= 256) {
      sdm_double2 v[8]; slabT_issue<2>(v, M, ldm, cbase, ncols, rbase + r0, n);
      if (!filled) { fill(); __syncthreads(); filled = true; }
      slabT_accum<2>(v, n, xs + r0, acc);
    } else if (n <= 512) {
      sdm_double2 v[16]; slabT_issue<4>(v, M, ldm, cbase, ncols, rbase + r0, n);
      if (!filled) { fill(); __syncthreads(); filled = true; }
      slabT_accum<4>(v, n, xs + r0, acc);
    } else {
      sdm_double2 v[32]; slabT_issue<8>(v, M, ldm, cbase, ncols, rbase + r0, n);
      if (!filled) { fill(); __syncthreads(); filled = true; }
      slabT_accum<8>(v, n, xs + r0, acc);
    }
  }
#pragma unroll
  for (int q = 0; q < 4; q++) {
    double a = acc[q];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    out[q] = a;
  }
}

// ================================================================ row products
// Every product of the sweeps that involves an inverse block, or a block of L inside a supernode, is a set of independent
// DOT PRODUCTS ALONG CONTIGUOUS MEMORY: rows of the transposed inverse (forward diagonal block), rows of the transposed copy of
// L (forward step), columns of the inverse (backward diagonal block), columns of L (backward step).  One wavefront per row,
// four rows per workgroup: 16-byte loads, up to eight in flight per lane (the vector entries next to them, straight from
// L2), no LDS, no barrier, and -- rows being what they are -- as many workgroups as rows / 4, however long the rows are
// (the 16-row slabs of the first version streamed up to 256 KB per workgroup at ~25 GB/s: 12 us per launch on MAXCUT-4000).
// sum_{j = jlo}^{n-1} M[j] x[j];  M 16-byte aligned, readable up to index n rounded up to even; fixed summation order.
// GATHER: x[j] = xg[px[j]].  Result in every lane.  (Eight 16-byte loads per lane in flight, the vector entries next to them
// straight from L2; the variant with the vector staged in LDS and sixteen loads in flight measured 5-10 % slower on the long
// rows of MAXCUT-4000 -- the barrier costs more than the second round trip: profiles/r03n.)
// LPR = lanes per row: 64 (one row per wavefront) or 16 (four rows per wavefront, lane = position inside its group of 16: fronts of
// up to 256 columns -- rows of at most 128 pairs are eight loads per lane of a 16-lane group, and a launch over many small fronts
// needs a quarter of the workgroups: blockdiag 64 x 150 columns: 2432 -> 640, 27 -> 19 us per solve).
template <bool GATHER, int LPR = 64>
__device__ __forceinline__ double row_dot(const double *__restrict__ M, const double *__restrict__ x, const int *__restrict__ px, int n, int jlo, int lane) {
  double a0 = 0.0, a1 = 0.0;
  const sdm_double2 *M2 = (const sdm_double2 *)M;
  const int npair = (n + 1) >> 1;
  for (int p0 = 0; p0 < npair; p0 += 8 * LPR) {
    sdm_double2 v[8];
    double x0[8], x1[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int pi = p0 + lane + LPR * k, pc = min(pi, npair - 1);
      v[k] = M2[pc];
      const int j0 = 2 * pc, j1 = min(2 * pc + 1, n - 1);
      x0[k] = GATHER ? x[px[j0]] : x[j0];
      x1[k] = GATHER ? x[px[j1]] : x[j1];
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int pi = p0 + lane + LPR * k;
      const bool in0 = pi < npair && 2 * pi >= jlo, in1 = pi < npair && 2 * pi + 1 < n && 2 * pi + 1 >= jlo;
      // (selects on both factors: what lies outside the range may be anything)
      a0 += (in0 ? v[k].x : 0.0) * (in0 ? x0[k] : 0.0);
      a1 += (in1 ? v[k].y : 0.0) * (in1 ? x1[k] : 0.0);
    }
  }
  double a = a0 + a1;
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) a += __shfl_xor(a, o);
  return a;
}
// Round 6 (tools/ubench/ubench11, profiles/r08c_*): a launch of a sweep is a BURST, not a stream -- every workgroup of it is resident at once,
// so what a launch takes is its boundary + the longest wavefront's chain of round trips; a wavefront that loops twice over 8 KB of a 16 KB row pays
// two of them (triangle of 2048: 6.7 us against 5.5 with every wavefront exactly one trip, full rows 8.0 -> 7.5).  row_seg = the pairs
// [plo, phi) of row_dot's sum, NL 16-byte loads per lane and trip; tri_task deals the rows of a triangular block so that a wavefront never loops:
// a row of up to SEGN entries to one wavefront (four rows per workgroup), a longer one to TWO wavefronts of one workgroup (two rows per workgroup)
// whose partial sums meet in LDS and are added in a fixed order.
template <bool GATHER, int NL>
__device__ __forceinline__ double row_seg(const double *__restrict__ M, const double *__restrict__ x, const int *__restrict__ px, int n, int jlo, int plo, int phi, int lane) {
  double a0 = 0.0, a1 = 0.0;
  const sdm_double2 *M2 = (const sdm_double2 *)M;
  for (int p0 = plo; p0 < phi; p0 += 64 * NL) {
    sdm_double2 v[NL];
    double x0[NL], x1[NL];
#pragma unroll
    for (int k = 0; k < NL; k++) {
      const int pi = p0 + lane + 64 * k, pc = min(pi, phi - 1);
      v[k] = M2[pc];
      const int j0 = 2 * pc, j1 = min(2 * pc + 1, n - 1);
      x0[k] = GATHER ? x[px[j0]] : x[j0];
      x1[k] = GATHER ? x[px[j1]] : x[j1];
    }
#pragma unroll
    for (int k = 0; k < NL; k++) {
      const int pi = p0 + lane + 64 * k;
      const bool in0 = pi < phi && 2 * pi >= jlo, in1 = pi < phi && 2 * pi + 1 < n && 2 * pi + 1 >= jlo;
      // (fma spelled out: left to the compiler's contraction, two inlined copies of this loop -- the diagonal-block launch and the merged launch's
      // diagonal role -- came out one fused, one not, and their results an ulp apart: profiles/r08s_probe.txt)
      a0 = fma(in0 ? v[k].x : 0.0, in0 ? x0[k] : 0.0, a0);
      a1 = fma(in1 ? v[k].y : 0.0, in1 ? x1[k] : 0.0, a1);
    }
  }
  double a = a0 + a1;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
  return a;
}
constexpr int SEGN = 1024;                                           // entries of a row one wavefront takes in one trip (8 loads of 16 bytes per lane)
// workgroup b, wavefront `wave` of a triangular block of nb rows -> t = the row's rank by length (its row has t + 1 entries; -1: none), seg of nseg
__device__ __forceinline__ void tri_task(int nb, int b, int wave, int &t, int &seg, int &nseg) {
  const int n1 = min(nb, SEGN), wg1 = (n1 + 3) >> 2;
  if (b < wg1) { t = 4 * b + wave; seg = 0; nseg = 1; if (t >= n1) t = -1; }
  else { t = n1 + 2 * (b - wg1) + (wave >> 1); seg = wave & 1; nseg = 2; if (t >= nb) t = -1; }
}
__host__ __device__ inline int tri_grid(int nb) { const int n1 = nb < SEGN ? nb : SEGN; return (n1 + 3) / 4 + (nb - n1 + 1) / 2; }
// the two partial sums of a split row -> its first wavefront (every wavefront of the workgroup calls this)
__device__ __forceinline__ double seg_combine(double *part, double a, int wave, int lane, int nseg) {
  if (nseg == 1) return a;                                            // (uniform for the workgroup)
  if (lane == 0) part[wave] = a;
  __syncthreads();
  return part[wave & ~1] + part[wave | 1];
}
// transposed copy of the rows of L below super-block Pb of a front (64x64 tiles through LDS): LT[r*W + c] = L((Pb+1) W + r, Pb W + c)
__device__ __forceinline__ int64_t lt_boff(int ns, int W, int Pb) { return (int64_t)W * ((int64_t)Pb * ns - (int64_t)W * Pb * (Pb + 1) / 2); }
__global__ void __launch_bounds__(ST)
k_ltrans(const double *__restrict__ F, double *__restrict__ LT, FrontTab tab, const int *items, int W) {
  __shared__ double t[64][65];
  const int *it = items + 4 * blockIdx.x;
  const int s = it[0], Pb = it[1], I = it[2], J = it[3];
  const int ns = tab.ns[s], ld = tab.ld[s];
  const int R0 = (Pb + 1) * W, nr = ns - R0;
  const double *src = F + tab.foff[s] + (int64_t)(Pb * W + 64 * J) * ld + R0 + 64 * I;       // (row i, column c) at src[c*ld + i]
  double *dst = LT + tab.ltoff[s] + lt_boff(ns, W, Pb) + (int64_t)(64 * I) * W + 64 * J;
  const int tid = threadIdx.x, a = tid & 63, b = tid >> 6;
  const int nri = min(64, nr - 64 * I);
  for (int c = b; c < 64; c += ST / 64) t[c][a] = a < nri ? src[(int64_t)c * ld + a] : 0.0;
  __syncthreads();
  for (int i = b; i < 64; i += ST / 64) if (i < nri) dst[(int64_t)i * W + a] = t[a][i];
}

// ================================================================ forward sweep
// several right-hand sides side by side (blockIdx.z): element strides of the right-hand sides, of the result and of the
// update-vector scratch (all 0 for a single right-hand side)
struct FwBatch { int64_t src, y, wv; };
// assembly of a front's right-hand side (levels above the leaves): own entries through perm, children's update vectors
__global__ void __launch_bounds__(ST)
k_sfw_init(FrontTab tab, const int *list, double *wv, const double *src, const int *perm, const double *y, FwBatch bt) {
  const int s = list[blockIdx.x];
  const int ns = tab.ns[s], ms = tab.ms[s], first = tab.first[s];
  wv += (int64_t)blockIdx.z * bt.wv; y += (int64_t)blockIdx.z * bt.y; if (src) src += (int64_t)blockIdx.z * bt.src;
  double *a = wv + tab.woff[s];
  const int tid = threadIdx.x;
  for (int i = tid; i < ms; i += ST) a[i] = i < ns ? (src ? src[perm[first + i]] : y[first + i]) : 0.0;
  __syncthreads();
  for (int ci = tab.childptr[s]; ci < tab.childptr[s + 1]; ci++) {   // fixed order: deterministic
    const int c = tab.childlist[ci];
    const int nc = tab.ns[c], mu = tab.ms[c] - nc;
    const int *rel = tab.relidx + tab.roff[c];
    const double *wc = wv + tab.woff[c] + nc;
    for (int i = tid; i < mu; i += ST) a[rel[i]] += wc[i];
    __syncthreads();
  }
}

#define FT(field) (tab.one ? tab.o_##field : tab.field[s])      // front descriptor: kernel argument (one-front level) or table
// y_P = inv(L_PP) t_P for super-block Pb of every front of a level that has one; t_P = the right-hand side gathered
// through perm (Pb = 0 of a leaf level: gather0) or the front's assembled / updated vector a.  With zdiv the ./d copy
// (wrapPcg.m:57; skipped pivots act as 1, deninfac.m:89-94) is written as well: y_P is final here.  One wavefront per row of
// the TRANSPOSED inverse (row r of the inverse: r + 1 contiguous entries).  A block that failed the growth check is
// substituted against the factor by workgroup 0.
__device__ __forceinline__ void sfw_diag_body(char *smem, double *part, int bx, const double *__restrict__ F, const double *__restrict__ STr, const FrontTab &tab, const int *list,
           const double *wv, const double *src, const int *perm, double *y, const unsigned long long *sb_g, double thr, int Pb, int gather0, FwBatch bt,
           double *zdiv, const double *dscale, int W, int mode, double thr2, const double *resid, int *noted, int seq, int mark, int *rearm) {
  const bool WIDE = W > 256;                                           // (uniform for the launch: the grid is sized accordingly)
  const bool gather = gather0 && Pb == 0 && mode != 2;
  // mode 0: the sweep as planned for well-conditioned factors (blocks beyond the bound are substituted by workgroup 0);
  // mode 1: the first of the refinement launches: blocks of kind 1 are applied as their inverse like the good ones;
  // mode 2: blocks of kind 1 only:  y_P += inv(L_PP) r_P  with the residual r_P = t_P - L_PP y_P of k_sfw_resid (`resid`)
  // (smem: the rare substitution fallback only: W + BSC * BSP doubles of dynamic LDS -- as static
  double *xs = (double *)smem, *Sd = xs + W;                          // arrays sized for the widest block they cost every launch 49 KB per workgroup)
  // (the first diagonal-block launch of a sweep tells the host that the sweep before it has run: CholPlan::noted)
  if (mark >= 0 && noted && bx == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) sdm_host_note(noted + 1, mark);
  // (the level's first diagonal-block launch clears both counter sets of the merged launches that follow it: merged_count)
  if (rearm && bx == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x < 2 * MC_N) sdm_signal_reset(rearm + (int)threadIdx.x * MC_STRIDE);
  const int s = tab.one ? tab.o_s : list[blockIdx.y];
  const int ns = FT(ns), first = FT(first);
  const int c0 = Pb * W;
  if (c0 >= ns) return;
  const int nb = min(W, ns - c0);
  if (!WIDE ? 16 * bx >= nb : bx >= tri_grid(nb)) return;
  wv += (int64_t)blockIdx.z * bt.wv; y += (int64_t)blockIdx.z * bt.y; if (src) src += (int64_t)blockIdx.z * bt.src;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int sld = FT(sld);
  const double *a = wv + FT(woff) + c0;
  int cls = -1;
  if (mode == 2) {                                                     // (the refinement launches: blocks within the bound leave them at once)
    cls = sb_class(sb_g, FT(sboff) + Pb, thr, thr2);
    if (cls != 1) return;
    a = resid + first + c0;
  }
  const int *pp = perm + first + c0;
  // the product of the planned path FIRST: its loads do not wait for the block's class (a scalar load and a branch in front of them was a
  // round trip of its own); a block beyond the bound throws the sum away below
  double sum = 0.0;
  int r = -1, l16 = lane & 15;
  if (!WIDE) {                                                         // 16 rows per workgroup (the grid is sized accordingly)
    r = 16 * bx + 4 * wave + (lane >> 4);
    const int rc = min(r, nb - 1);
    const double *M = STr + FT(soff) + (int64_t)c0 * sld + (int64_t)rc * sld;
    sum = gather ? row_dot<true, 16>(M, src, pp, rc + 1, 0, l16) : row_dot<false, 16>(M, a, nullptr, rc + 1, 0, l16);
    if (l16 != 0 || r >= nb) r = -1;
  } else {
    int seg, nseg;
    tri_task(nb, bx, wave, r, seg, nseg);
    if (r >= 0) {
      const double *M = STr + FT(soff) + (int64_t)c0 * sld + (int64_t)r * sld;
      const int npair = (r + 2) >> 1, plo = seg * (SEGN / 2), phi = nseg == 1 ? npair : min(npair, plo + SEGN / 2);
      sum = gather ? row_seg<true, 8>(M, src, pp, r + 1, 0, plo, phi, lane) : row_seg<false, 8>(M, a, nullptr, r + 1, 0, plo, phi, lane);
    }
    sum = seg_combine(part, sum, wave, lane, nseg);
    if (lane != 0 || seg != 0) r = -1;
  }
  if (cls < 0) cls = sb_class(sb_g, FT(sboff) + Pb, thr, thr2);
  // (a block of kind 1 met by a sweep: the host plans the refinement launches while such blocks keep turning up -- solve_refines)
  if (cls == 1 && mode != 2 && bx == 0 && tid == 0 && noted) sdm_host_note(noted, seq);
  if (cls == 2 || (cls == 1 && mode == 0)) {
    if (bx != 0) return;
    for (int c = tid; c < nb; c += ST) xs[c] = gather ? src[pp[c]] : a[c];
    __syncthreads();
    block_solve_fw(F + FT(foff), FT(ld), c0, nb, xs, Sd);
    for (int i = tid; i < nb; i += ST) {
      const double yv = xs[i];
      y[first + c0 + i] = yv;
      if (zdiv) { const double dk = dscale[first + c0 + i]; zdiv[first + c0 + i] = yv / (dk > 0.0 ? dk : 1.0); }
    }
    return;
  }
  if (r >= 0) {
    const double yv = mode == 2 ? y[first + c0 + r] + sum : sum;
    y[first + c0 + r] = yv;
    if (zdiv) { const double dk = dscale[first + c0 + r]; zdiv[first + c0 + r] = yv / (dk > 0.0 ? dk : 1.0); }
  }
}
__global__ void __launch_bounds__(ST)
k_sfw_diag(const double *__restrict__ F, const double *__restrict__ STr, FrontTab tab, const int *list, const double *wv, const double *src,
           const int *perm, double *y, const unsigned long long *sb_g, double thr, int Pb, int gather0, FwBatch bt,
           double *zdiv, const double *dscale, int W, int mode, double thr2, const double *resid, int *noted, int seq, int mark, int *rearm) {
  SDM_DYN_SMEM(smem);
  __shared__ double part[ST / 64];
  sfw_diag_body(smem, part, (int)blockIdx.x, F, STr, tab, list, wv, src, perm, y, sb_g, thr, Pb, gather0, bt, zdiv, dscale, W, mode, thr2, resid, noted, seq, mark, rearm);
}

// r_P = t_P - L_PP y_P for the super-blocks of kind 1 (iterative refinement of y_P = inv(L_PP) t_P against the factor itself: the
// explicit inverse of an ill-conditioned block loses accuracy in proportion to its growth, the refined result has the accuracy of
// the substitution it replaces).  L is stored by columns: a workgroup takes 64 rows (one per lane), its four wavefronts the
// columns j = w, w + 4, ... left of the diagonal (512-byte coalesced reads), the partial sums meet in LDS.
constexpr int RT = 1024;          // work-items of k_sfw_resid: sixteen wavefronts keep 64 KB of a 64-row slab in flight (with four, the last slab of a
                                  // 2048-wide block was 64 dependent round trips: 55 us)
__global__ void __launch_bounds__(RT)
k_sfw_resid(const double *__restrict__ F, FrontTab tab, const int *list, const double *wv, const double *src, const int *perm, const double *y,
            double *resid, const unsigned long long *sb_g, double thr, double thr2, int Pb, int gather0, int W) {
  __shared__ double ys[SBW_MAX], red[RT / 64][64];
  const int s = tab.one ? tab.o_s : list[blockIdx.y];
  const int ns = FT(ns), first = FT(first), ld = FT(ld);
  const int c0 = Pb * W;
  if (c0 >= ns) return;
  const int nb = min(W, ns - c0);
  const int i0 = 64 * blockIdx.x;
  if (i0 >= nb) return;
  if (sb_class(sb_g, FT(sboff) + Pb, thr, thr2) != 1) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int jend = min(i0 + 64, nb);
  for (int c = tid; c < jend; c += RT) ys[c] = y[first + c0 + c];
  __syncthreads();
  const int i = i0 + lane, ic = min(i, nb - 1);
  const double *Fi = F + FT(foff) + (int64_t)c0 * ld + c0 + ic;        // L(c0 + i, c0 + j) = Fi[j * ld]
  double acc = 0.0;
  for (int j0 = wave; j0 < jend; j0 += 8 * (RT / 64)) {
    double v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = Fi[(int64_t)min(j0 + (RT / 64) * k, jend - 1) * ld];
#pragma unroll
    for (int k = 0; k < 8; k++) { const int j = j0 + (RT / 64) * k; if (j < i && j < jend) acc += v[k] * ys[j]; }
  }
  red[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && i < nb) {
    double sum = 0.0;
    for (int w = 0; w < RT / 64; w++) sum += red[w][lane];
    const bool gather = gather0 && Pb == 0;
    const double t = gather ? src[perm[first + c0 + i]] : wv[FT(woff) + c0 + i];
    resid[first + c0 + i] = (t - ys[i]) - sum;
  }
}

// step Pb, the front's OWN rows of later super-blocks:  - L(r, P) y_P  along the rows of the transposed copy LT
// assign0 (Pb = 0 of a leaf level): the vector a has not been initialised: a = right-hand side - sum.
#ifndef SDM_ROWS_NL
#define SDM_ROWS_NL 4            // loads of 16 bytes per lane and trip in the full-row launches (ubench11: two wavefronts per 2048-entry row, 4 loads x 2 trips)
#endif
// bx = the workgroup's index among the row workgroups; urgent_wt: rows of the NEXT super-block are stored write-through (the merged launch: its
// diagonal role reads them in the same launch)
__device__ __forceinline__ void sfw_rows_body(double *part, int bx, const double *__restrict__ LT, const FrontTab &tab, const int *list, double *wv, const double *src,
                                              const int *perm, const double *y, int Pb, int assign0, FwBatch bt, int W, bool urgent_wt) {
  // W > SEGN: two wavefronts per row (two rows per workgroup), else one (four rows per workgroup): uniform for the launch, the grid is sized accordingly
  const int s = tab.one ? tab.o_s : list[blockIdx.y];
  const int ns = FT(ns), first = FT(first);
  const int R0 = (Pb + 1) * W;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int nseg = W > SEGN ? 2 : 1, seg = nseg == 2 ? (wave & 1) : 0;
  const int rl = nseg == 2 ? 2 * bx + (wave >> 1) : 4 * bx + wave;
  if ((nseg == 2 ? 2 : 4) * bx + R0 >= ns) return;                    // (the whole workgroup)
  wv += (int64_t)blockIdx.z * bt.wv; y += (int64_t)blockIdx.z * bt.y; if (src) src += (int64_t)blockIdx.z * bt.src;
  const bool live = R0 + rl < ns;
  double sum = 0.0;
  if (live) {
    const double *M = LT + FT(ltoff) + lt_boff(ns, W, Pb) + (int64_t)rl * W;
    const int npair = W >> 1, per = npair / nseg;
    sum = row_seg<false, SDM_ROWS_NL>(M, y + first + Pb * W, nullptr, W, 0, seg * per, (seg + 1) * per, lane);
  }
  sum = seg_combine(part, sum, wave, lane, nseg);
  if (lane == 0 && seg == 0 && live) {
    double *a = wv + FT(woff);
    const int r = R0 + rl;
    const double v = (assign0 ? src[perm[first + r]] : a[r]) - sum;
    if (urgent_wt && rl < W) sdm_store_wt(&a[r], v); else a[r] = v;
  }
}
__global__ void __launch_bounds__(ST)
k_sfw_rows(const double *__restrict__ LT, FrontTab tab, const int *list, double *wv, const double *src, const int *perm, const double *y,
           int Pb, int assign0, FwBatch bt, int W) {
  __shared__ double part[ST / 64];
  sfw_rows_body(part, (int)blockIdx.x, LT, tab, list, wv, src, perm, y, Pb, assign0, bt, W, false);
}

// ---- Round 6: a row launch and the NEXT diagonal-block launch as ONE (profiles/r08r_solve_trace_m16000.txt: the 16 diagonal-block launches of a
// solve at m = 16000 are 26 % of its time for 13 % of its bytes -- 7.9 us each for 16.8 MB, a burst of loads behind a launch boundary).  Row r of
// y_{P+1} = inv(L_{P+1,P+1}) t_{P+1} needs t_{P+1}[0 .. r]: the rows of the NEXT super-block (the URGENT rows of this launch) up to r, nothing of
// the later rows of the front.  Grid order = dependency order: the urgent row workgroups in row order, up to PRE_WGS of the others, the diagonal
// role's workgroups in row order (tri_task), the rest.  The urgent workgroups store their entries of t past the caches and count per CHUNK of 64
// rows; a diagonal workgroup waits for the chunks its rows reach -- all handed out before it, none of which waits for anything --, copies that
// much of t to LDS with L1-bypassing loads (once per workgroup: read that way per row, 2 M loads on the same 16 KB, the role took twice the
// separate launch's time -- profiles/r08s_*) and runs its rows against the copy.  Its
// loads ride in the stream of the launch, its first rows a few microseconds behind the first urgent ones.  One right-hand side, one front in the
// level, W > 256, no refinement launches (solve_fw_batch).
// The counters: 1024 increments of ONE address cost a launch 19 us (same-address atomics resolve one after the other at the device's coherence
// point): one counter per chunk, each on a 128-byte line of its own (32 or 16 increments each).  Two sets: the launches of a level use them in
// turn (set = super-block & 1), a launch clears the set the NEXT one counts on, and the level's first diagonal-block launch -- never merged --
// clears both: no re-arming count that every diagonal workgroup would have to pass through, nothing carried from sweep to sweep (a replayed
// graph finds what it was captured with).
constexpr int PRE_WGS = 1024;          // (SEDUMI_HIP_SWEEP_PRE overrides: 0 ... 512 lose 8 - 11 %, 1024 ... 4096 are level -- profiles/r08x_merge_pre_sweep.txt)
#ifdef SDM_EMU
#define SDM_EIGHT_WAVES
#else
#define SDM_EIGHT_WAVES __attribute__((amdgpu_waves_per_eu(8, 8)))      // <= 64 vector registers: eight workgroups of 256 per CU
#endif
// role of workgroup b of a merged launch: nurg urgent workgroups, npre others of the streaming role ahead of the ndiag of the diagonal role, the rest behind.  Returns true
// for the diagonal role; bx = the index inside the role
__device__ __forceinline__ bool merged_role(int b, int nurg, int npre, int ndiag, int &bx) {
  const int pre = nurg + npre;
  if (b < pre) { bx = b; return false; }
  if (b < pre + ndiag) { bx = b - pre; return true; }
  bx = b - ndiag;
  return false;
}
// urgent workgroups per chunk of 64 rows (two rows per workgroup when W > SEGN, else four: sfw_rows_body)
__device__ __forceinline__ int merged_per(int W) { return W > SEGN ? 32 : 16; }
__device__ __forceinline__ void merged_count(int *cnt, int set, int bx, int W) { sdm_signal_add(cnt + set * MC_SET + (bx / merged_per(W)) * MC_STRIDE); }
__device__ __forceinline__ void merged_clear(int *cnt, int set) { if (threadIdx.x < MC_N) sdm_signal_reset(cnt + set * MC_SET + (int)threadIdx.x * MC_STRIDE); }
// until the first nch chunks of the urgent rows are out (work-item k polls chunk k's counter), then a barrier
__device__ __forceinline__ void merged_wait(const int *cnt, int set, int nch, int nurg, int W, int *tmo) {
  const int k = threadIdx.x, per = merged_per(W);
  if (k < 64) {
    // (the chunks come out roughly in order: ONE work-item polls the last one needed, then the others look at theirs -- with every one of them
    // polling from the start, 12 000 pollers on 31 lines, an all-urgent launch took 22 us for 14 of separate launches)
    for (int ph = 0; ph < 2; ph++)
      if ((ph == 0 ? k == nch - 1 : k < nch - 1) && k * per < nurg) {
        const int target = min(per, nurg - k * per);
        const int *c = cnt + set * MC_SET + k * MC_STRIDE;
        for (long it = 0; sdm_signal_load(c) < target; it++) { if (sdm_spin_giveup(it, tmo)) break; SDM_SPIN_PAUSE(); SDM_SPIN_PAUSE(); }
      }
  }
  __syncthreads();
}
// the largest row rank among the tasks of workgroup b (tri_task)
__device__ __forceinline__ int tri_tmax(int nb, int b) {
  const int n1 = min(nb, SEGN), wg1 = (n1 + 3) >> 2;
  return min(nb - 1, b < wg1 ? 4 * b + 3 : n1 + 2 * (b - wg1) + 1);
}
// the diagonal role of the merged launch: sfw_diag_body for W > 256, mode 0, no gather -- written out on its own, lean: the kernel must fit 64
// vector registers, or its streaming role loses the eight workgroups per CU it lives on (sfw_diag_body with its W <= 256 and gather paths takes
// 111; a second inlined copy of it also crashes the device compiler of ROCm 7.2, clang-22's CGSCC inliner)
__device__ __forceinline__ void sfw_diag_lean(char *smem, double *part, int bx, const double *__restrict__ F, const double *__restrict__ STr, const FrontTab &tab,
                                              const double *wv, double *y, const unsigned long long *sb_g, double thr, int Pb, double *zdiv, const double *dscale, int W,
                                              double thr2, int *noted, int seq, const int *cnt, int set, int nurg, int *tmo) {
  const int ns = tab.o_ns, first = tab.o_first, c0 = Pb * W, nb = min(W, ns - c0);   // (one-front levels only)
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const double *a = wv + tab.o_woff + c0;
  double *xs = (double *)smem, *Sd = xs + W;
  // (the block's class first: its round trip passes while the counters are polled)
  const int cls = sb_class(sb_g, tab.o_sboff + Pb, thr, thr2);
  if (cls == 1 && bx == 0 && tid == 0 && noted) sdm_host_note(noted, seq);
  if (cls != 0 && bx != 0) return;                                    // beyond the bound: substituted against the factor by workgroup 0 (as sfw_diag_body, mode 0)
  const int need = cls != 0 ? nb : tri_tmax(nb, bx) + 1;
  merged_wait(cnt, set, (need + 63) >> 6, nurg, W, tmo);
  for (int c = tid; c < need; c += ST) xs[c] = sdm_load_wt(&a[c]);
  __syncthreads();
  if (cls != 0) {
    block_solve_fw_inl<BSC_LEAN>(F + tab.o_foff, tab.o_ld, c0, nb, xs, Sd);
    for (int i = tid; i < nb; i += ST) {
      const double yv = xs[i];
      y[first + c0 + i] = yv;
      if (zdiv) { const double dk = dscale[first + c0 + i]; zdiv[first + c0 + i] = yv / (dk > 0.0 ? dk : 1.0); }
    }
    return;
  }
  int r, seg, nseg;
  tri_task(nb, bx, wave, r, seg, nseg);
  double sum = 0.0;
  if (r >= 0) {
    const double *M = STr + tab.o_soff + (int64_t)c0 * tab.o_sld + (int64_t)r * tab.o_sld;
    const int npair = (r + 2) >> 1, plo = seg * (SEGN / 2), phi = nseg == 1 ? npair : min(npair, plo + SEGN / 2);
    sum = row_seg<false, SDM_ROWS_NL>(M, xs, nullptr, r + 1, 0, plo, phi, lane);
  }
  sum = seg_combine(part, sum, wave, lane, nseg);
  if (lane == 0 && seg == 0 && r >= 0) {
    y[first + c0 + r] = sum;
    if (zdiv) { const double dk = dscale[first + c0 + r]; zdiv[first + c0 + r] = sum / (dk > 0.0 ? dk : 1.0); }
  }
}
__global__ void __launch_bounds__(ST) SDM_EIGHT_WAVES
k_sfw_rows_diag(const double *__restrict__ LT, const double *__restrict__ F, const double *__restrict__ STr, FrontTab tab, const int *list, double *wv, const double *src,
                const int *perm, double *y, const unsigned long long *sb_g, double thr, int Pb, int assign0, FwBatch bt, double *zdiv, const double *dscale, int W,
                double thr2, int *noted, int seq, int *cnt, int nurg, int npre, int ndiag, int *tmo) {
  SDM_DYN_SMEM(smem);
  __shared__ double part[ST / 64];
  int bx;
  if (!merged_role((int)blockIdx.x, nurg, npre, ndiag, bx)) {
    sfw_rows_body(part, bx, LT, tab, list, wv, src, perm, y, Pb, assign0, bt, W, true);
    if (bx < nurg) {                                                   // an urgent workgroup: its rows of t_{P+1} are out (write-through), count it
      SDM_STORES_DONE();
      __syncthreads();
      if (threadIdx.x == 0) merged_count(cnt, Pb & 1, bx, W);
      if (bx == 0) merged_clear(cnt, (Pb + 1) & 1);
    }
    return;
  }
  sfw_diag_lean(smem, part, bx, F, STr, tab, wv, y, sb_g, thr, Pb + 1, zdiv, dscale, W, thr2, noted, seq, cnt, Pb & 1, nurg, tmo);
}

// step Pb, the rows BELOW the supernode (they belong to its ancestors; their sums are the update vector passed to the parent):
//  - L(r, P) y_P  read from the factor itself, 16-row slabs.  assign0: the vector has not been initialised: 0 - sum.
__global__ void __launch_bounds__(ST)
k_sfw_step(const double *__restrict__ F, FrontTab tab, const int *list, double *wv, const double *y, int Pb, int assign0, FwBatch bt, int W) {
  __shared__ double xs[SBW_MAX], red[(ST / 8) * SROWS];
  const int s = tab.one ? tab.o_s : list[blockIdx.y];
  const int ns = FT(ns), ms = FT(ms), first = FT(first), ld = FT(ld);
  const int c0 = Pb * W;
  if (c0 >= ns || ms <= ns) return;
  const int nb = min(W, ns - c0);
  const int r0 = (ns & ~1) + SROWS * blockIdx.x;
  if (r0 >= ms) return;
  wv += (int64_t)blockIdx.z * bt.wv; y += (int64_t)blockIdx.z * bt.y;
  const int tid = threadIdx.x;
  const double *yp = y + first + c0;
  const double sum = fw_product(F + FT(foff) + (int64_t)c0 * ld, ld, nb, r0, ms - 1, xs, red,
                                [&]() { for (int c = tid; c < nb; c += ST) xs[c] = yp[c]; });
  double *a = wv + FT(woff);
  const int r = r0 + tid;
  if (tid < SROWS && r >= ns && r < ms) a[r] = (assign0 ? 0.0 : a[r]) - sum;
}

// ================================================================ backward sweep
// v = z ./ d  -  (rows below the supernode)' x_ancestors   for every column of every front of a level
__global__ void __launch_bounds__(ST)
k_sbw_init(const double *__restrict__ F, FrontTab tab, const int *list, double *y, const double *xfin, const double *dscale) {
  __shared__ double xs[256];
  const int s = tab.one ? tab.o_s : list[blockIdx.y];
  const int ns = FT(ns), ms = FT(ms), first = FT(first), ld = FT(ld);
  const int c0 = SROWS * blockIdx.x;
  if (c0 >= ns) return;
  const int ncols = min(SROWS, ns - c0);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const double *Fs = F + FT(foff);
  const int *rows = tab.lindx + FT(xl);
  double tot[4] = {0.0, 0.0, 0.0, 0.0};
  const int ebase = ns & ~1;
  if (ms > ns) {
    for (int rb = ebase; rb < ms; rb += 256) {                      // ancestors' entries, gathered 256 at a time
      const int nr = min(256, ms - rb);
      sdm_double2 v[8];
      slabT_issue<2>(v, Fs, ld, c0, ncols, rb, nr);
      __syncthreads();
      for (int i = tid; i < nr; i += ST) xs[i] = rb + i >= ns ? xfin[rows[rb + i]] : 0.0;
      __syncthreads();
      slabT_accum<2>(v, nr, xs, tot);
    }
  }
#pragma unroll
  for (int q = 0; q < 4; q++) {
    double a = tot[q];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    tot[q] = a;
  }
  if (lane == 0) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int c = c0 + 4 * wave + q;
      if (c < ns) {
        const double z = y[first + c];
        double dk = dscale ? dscale[first + c] : 1.0;
        dk = dk > 0.0 ? dk : 1.0;                                    // skipped pivots act as 1 (deninfac.m:89-94)
        y[first + c] = (dscale ? z / dk : z) - tot[q];
      }
    }
  }
}

// x_Q = inv(L_QQ)' v_Q for super-block Q of every front of a level that has one; the result goes to xfin (descendants
// and the step launch read it) and, scattered through perm, to yout.  One wavefront per COLUMN of the inverse (column c: the
// nb - c contiguous entries from the diagonal down).  A block that failed the growth check is substituted against the factor
// by workgroup 0.
__device__ __forceinline__ void sbw_diag_body(char *smem, double *part, int bx, const double *__restrict__ F, const double *__restrict__ S, const FrontTab &tab, const int *list,
           const double *y, double *xfin, double *yout, const int *perm, const unsigned long long *sb_g, double thr, int Q, int W, int mode, double thr2, const double *wv,
           int *noted, int seq, int mark, int *rearm) {
  // mode 0 / 1 / 2 as in k_sfw_diag; mode 2:  x_Q += inv(L_QQ)' r_Q  with the residual of k_sbw_resid (in the front's slice of wv)
  double *xs = (double *)smem, *Sd = xs + W;
  if (mark >= 0 && noted && bx == 0 && blockIdx.y == 0 && threadIdx.x == 0) sdm_host_note(noted + 1, mark);
  if (rearm && bx == 0 && blockIdx.y == 0 && threadIdx.x < 2 * MC_N) sdm_signal_reset(rearm + (int)threadIdx.x * MC_STRIDE);
  const int s = tab.one ? tab.o_s : list[blockIdx.y];
  const int ns = FT(ns), first = FT(first);
  const int rb = Q * W;
  if (rb >= ns) return;
  const int nb = min(W, ns - rb);
  if (W <= 256 ? 16 * bx >= nb : bx >= tri_grid(nb)) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  int cls = -1;
  if (mode == 2) {
    cls = sb_class(sb_g, FT(sboff) + Q, thr, thr2);
    if (cls != 1) return;
  }
  const double *vp = mode == 2 ? wv + FT(woff) + rb : y + first + rb;
  const int sld = FT(sld);
  // (the planned path's product before the block's class is looked at: see k_sfw_diag)
  double sum = 0.0;
  int c = -1;
  if (W <= 256) {                                                     // 16 columns per workgroup (see k_sfw_diag)
    c = 16 * bx + 4 * wave + (lane >> 4);
    const int l16 = lane & 15, cc = min(c, nb - 1), ce = cc & ~1;
    const double *M = S + FT(soff) + (int64_t)rb * sld + (int64_t)cc * sld + ce;
    sum = row_dot<false, 16>(M, vp + ce, nullptr, nb - ce, cc - ce, l16);
    if (l16 != 0 || c >= nb) c = -1;
  } else {
    int t, seg, nseg;
    tri_task(nb, bx, wave, t, seg, nseg);                     // column nb - 1 - t has t + 1 entries from the diagonal down
    if (t >= 0) {
      c = nb - 1 - t;
      const int ce = c & ~1;                                          // 16-byte aligned start (the entry above the diagonal is skipped: jlo)
      const double *M = S + FT(soff) + (int64_t)rb * sld + (int64_t)c * sld + ce;
      const int n = nb - ce, npair = (n + 1) >> 1, plo = seg * (SEGN / 2), phi = nseg == 1 ? npair : min(npair, plo + SEGN / 2);
      sum = row_seg<false, 8>(M, vp + ce, nullptr, n, c - ce, plo, phi, lane);
    }
    sum = seg_combine(part, sum, wave, lane, nseg);
    if (lane != 0 || seg != 0) c = -1;
  }
  if (cls < 0) cls = sb_class(sb_g, FT(sboff) + Q, thr, thr2);
  if (cls == 1 && mode != 2 && bx == 0 && tid == 0 && noted) sdm_host_note(noted, seq);
  if (cls == 2 || (cls == 1 && mode == 0)) {
    if (bx != 0) return;
    for (int i = tid; i < nb; i += ST) xs[i] = vp[i];
    __syncthreads();
    block_solve_bw(F + FT(foff), FT(ld), rb, nb, xs, Sd);
    for (int i = tid; i < nb; i += ST) { xfin[first + rb + i] = xs[i]; if (yout) yout[perm[first + rb + i]] = xs[i]; }
    return;
  }
  if (c >= 0) { const double xv = mode == 2 ? xfin[first + rb + c] + sum : sum; xfin[first + rb + c] = xv; if (yout) yout[perm[first + rb + c]] = xv; }
}
__global__ void __launch_bounds__(ST)
k_sbw_diag(const double *__restrict__ F, const double *__restrict__ S, FrontTab tab, const int *list, const double *y, double *xfin, double *yout,
           const int *perm, const unsigned long long *sb_g, double thr, int Q, int W, int mode, double thr2, const double *wv, int *noted, int seq, int mark, int *rearm) {
  SDM_DYN_SMEM(smem);
  __shared__ double part[ST / 64];
  sbw_diag_body(smem, part, (int)blockIdx.x, F, S, tab, list, y, xfin, yout, perm, sb_g, thr, Q, W, mode, thr2, wv, noted, seq, mark, rearm);
}
// the diagonal role of the merged launch k_sbw_step_diag (as sfw_diag_lean).  Super-block Q is a full one (only a front's last block is not);
// the urgent workgroups take its columns from the last one down (sbw_step_body, rev): chunk k = columns W - 64 k - 64 .. W - 64 k - 1, and the
// column of rank t (tri_task) needs the entries from its own -- W - 1 - t, an even start -- to the block's end: chunks 0 .. (t + 1) / 64
__device__ __forceinline__ void sbw_diag_lean(char *smem, double *part, int bx, const double *__restrict__ F, const double *__restrict__ S, const FrontTab &tab,
                                              const double *y, double *xfin, double *yout, const int *perm, const unsigned long long *sb_g, double thr, int Q, int W,
                                              double thr2, int *noted, int seq, const int *cnt, int set, int nurg, int *tmo) {
  const int first = tab.o_first, rb = Q * W, nb = W;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const double *vp = y + first + rb;
  double *xs = (double *)smem, *Sd = xs + W;
  const int cls = sb_class(sb_g, tab.o_sboff + Q, thr, thr2);
  if (cls == 1 && bx == 0 && tid == 0 && noted) sdm_host_note(noted, seq);
  if (cls != 0 && bx != 0) return;
  const int lo = cls != 0 ? 0 : (nb - 1 - tri_tmax(nb, bx)) & ~1;
  merged_wait(cnt, set, (nb - lo + 63) >> 6, nurg, W, tmo);
  for (int i = lo + tid; i < nb; i += ST) xs[i] = sdm_load_wt(&vp[i]);
  __syncthreads();
  if (cls != 0) {
    block_solve_bw_inl<BSC_LEAN>(F + tab.o_foff, tab.o_ld, rb, nb, xs, Sd);
    for (int i = tid; i < nb; i += ST) { xfin[first + rb + i] = xs[i]; if (yout) yout[perm[first + rb + i]] = xs[i]; }
    return;
  }
  int t, seg, nseg, c = -1;
  tri_task(nb, bx, wave, t, seg, nseg);                               // column nb - 1 - t has t + 1 entries from the diagonal down
  double sum = 0.0;
  if (t >= 0) {
    c = nb - 1 - t;
    const int ce = c & ~1;
    const double *M = S + tab.o_soff + (int64_t)rb * tab.o_sld + (int64_t)c * tab.o_sld + ce;
    const int n = nb - ce, npair = (n + 1) >> 1, plo = seg * (SEGN / 2), phi = nseg == 1 ? npair : min(npair, plo + SEGN / 2);
    sum = row_seg<false, SDM_ROWS_NL>(M, xs + ce, nullptr, n, c - ce, plo, phi, lane);
  }
  sum = seg_combine(part, sum, wave, lane, nseg);
  if (lane == 0 && seg == 0 && c >= 0) { xfin[first + rb + c] = sum; if (yout) yout[perm[first + rb + c]] = sum; }
}

// r_Q = v_Q - L_QQ' x_Q for the super-blocks of kind 1 (see k_sfw_resid): a column of L is contiguous, one wavefront per column;
// the residual goes to the front's slice of the forward sweep's update vectors (idle during the backward sweep)
__global__ void __launch_bounds__(ST)
k_sbw_resid(const double *__restrict__ F, FrontTab tab, const int *list, const double *y, const double *xfin, double *wv,
            const unsigned long long *sb_g, double thr, double thr2, int Q, int W) {
  const int s = tab.one ? tab.o_s : list[blockIdx.y];
  const int ns = FT(ns), first = FT(first), ld = FT(ld);
  const int rb = Q * W;
  if (rb >= ns) return;
  const int nb = min(W, ns - rb);
  const int c = 4 * blockIdx.x + (threadIdx.x >> 6);
  if (c >= nb) return;
  if (sb_class(sb_g, FT(sboff) + Q, thr, thr2) != 1) return;
  const int lane = threadIdx.x & 63;
  const int s0 = rb + c + 1, sa = s0 & ~1, send = rb + nb;           // rows below the diagonal, from a 16-byte aligned start
  double sum = 0.0;
  if (s0 < send) sum = row_dot<false>(F + FT(foff) + (int64_t)(rb + c) * ld + sa, xfin + first + sa, nullptr, send - sa, s0 - sa, lane);
  if (lane == 0) wv[FT(woff) + rb + c] = (y[first + rb + c] - xfin[first + rb + c]) - sum;
}

// step Q: x_Q is final; every column left of super-block Q receives  - L(Q rows, c)' x_Q , read from the factor (a column
// of L is contiguous there), one wavefront per column
// rev (the merged launch): workgroup bx takes the columns from rb - 1 DOWN -- those of super-block Q - 1 first, stored write-through
__device__ __forceinline__ void sbw_step_body(double *part, int bx, const double *__restrict__ F, const FrontTab &tab, const int *list, double *y, const double *xfin, int Q, int W,
                                              bool rev) {
  // (W > SEGN: two wavefronts per column, as in k_sfw_rows)
  const int s = tab.one ? tab.o_s : list[blockIdx.y];
  const int ns = FT(ns), first = FT(first), ld = FT(ld);
  const int rb = Q * W;
  if (rb >= ns) return;
  const int nbq = min(W, ns - rb);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int nseg = W > SEGN ? 2 : 1, seg = nseg == 2 ? (wave & 1) : 0;
  const int ci = nseg == 2 ? 2 * bx + (wave >> 1) : 4 * bx + wave;                                // < rb by the grid
  const int c = rev ? rb - 1 - ci : ci;
  const int npair = (nbq + 1) >> 1, per = nseg == 2 ? ((npair + 1) >> 1) : npair;
  double sum = row_seg<false, SDM_ROWS_NL>(F + FT(foff) + (int64_t)c * ld + rb, xfin + first + rb, nullptr, nbq, 0, seg * per, min(npair, (seg + 1) * per), lane);
  sum = seg_combine(part, sum, wave, lane, nseg);
  if (lane == 0 && seg == 0) {
    const double v = y[first + c] - sum;
    if (rev && ci < W) sdm_store_wt(&y[first + c], v); else y[first + c] = v;
  }
}
__global__ void __launch_bounds__(ST)
k_sbw_step(const double *__restrict__ F, FrontTab tab, const int *list, double *y, const double *xfin, int Q, int W) {
  __shared__ double part[ST / 64];
  sbw_step_body(part, (int)blockIdx.x, F, tab, list, y, xfin, Q, W, false);
}
// step Q and the diagonal block Q - 1 as ONE launch (k_sfw_rows_diag's counterpart: the columns of super-block Q - 1 are the urgent ones)
__global__ void __launch_bounds__(ST) SDM_EIGHT_WAVES
k_sbw_step_diag(const double *__restrict__ F, const double *__restrict__ S, FrontTab tab, const int *list, double *y, double *xfin, double *yout, const int *perm,
                const unsigned long long *sb_g, double thr, int Q, int W, double thr2, int *noted, int seq, int *cnt, int nurg, int npre, int ndiag, int *tmo) {
  SDM_DYN_SMEM(smem);
  __shared__ double part[ST / 64];
  int bx;
  if (!merged_role((int)blockIdx.x, nurg, npre, ndiag, bx)) {
    sbw_step_body(part, bx, F, tab, list, y, xfin, Q, W, true);
    if (bx < nurg) {
      SDM_STORES_DONE();
      __syncthreads();
      if (threadIdx.x == 0) merged_count(cnt, Q & 1, bx, W);
      if (bx == 0) merged_clear(cnt, (Q + 1) & 1);
    }
    return;
  }
  sbw_diag_lean(smem, part, bx, F, S, tab, y, xfin, yout, perm, sb_g, thr, Q - 1, W, thr2, noted, seq, cnt, Q & 1, nurg, tmo);
}
#undef FT

// ================================================================ host drivers
const double *solve_d(sdm_plan *P) { return P->dense.factored ? (const double *)P->chol.dsolve.p : (const double *)P->chol.d.p; }

// the level's table for the solve kernels: with the descriptor of its only front filled in when there is just one
static FrontTab level_tab(const CholPlan &C, FrontTab t, int l) {
  if (C.levptr[l + 1] - C.levptr[l] != 1) return t;
  const int s = C.levlist[C.levptr[l]];
  t.one = 1; t.o_s = s; t.o_ns = C.sn_ns[s]; t.o_ms = C.sn_ms[s]; t.o_ld = C.sn_ld[s]; t.o_first = C.sn_first[s];
  t.o_sld = C.sn_sld[s]; t.o_sboff = C.sn_sboff[s]; t.o_foff = C.sn_foff[s]; t.o_soff = C.sn_soff[s]; t.o_woff = C.sn_woff[s]; t.o_ltoff = C.sn_ltoff[s];
  t.o_xl = C.sn_xl[s];
  return t;
}
static void solve_attrs() {
#ifndef SDM_EMU
  static bool attr = false;
  if (!attr) {
    SDM_HIP_CHECK(hipFuncSetAttribute((const void *)k_sinv128, hipFuncAttributeMaxDynamicSharedMemorySize, (int)INV_LDS));
    SDM_HIP_CHECK(hipFuncSetAttribute((const void *)k_stile, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TILE_LDS));
    SDM_HIP_CHECK(hipFuncSetAttribute((const void *)k_sprep, hipFuncAttributeMaxDynamicSharedMemorySize, (int)INV_LDS));
    SDM_HIP_CHECK(hipFuncSetAttribute((const void *)k_sinv_follow, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TILE_LDS));
    attr = true;
  }
#endif
}
// the inverses of the fronts of level l behind their factorisation: launched on stream st right after (next to) k_ldl_front
void solve_follow(sdm_plan *P, int l, hipStream_t st) {
  CholPlan &C = P->chol;
  solve_attrs();
  C.growth_used = C.growth_max;
  const int nfr = C.levptr[l + 1] - C.levptr[l];
  SDM_KLAUNCH_ON(P, st, k_sinv_follow, dim3(C.lev_followT[l], nfr), dim3(ST), TILE_LDS, C.fronts.p, C.frontsT.p, C.S.p, C.ST.p, front_tab(C),
                 C.d_levlist.p + C.levptr[l], C.front_cnt.p, C.diag_cnt.p, C.sb_g.p, C.tmo.dev());
}
void solve_prepare(sdm_plan *P, bool sb_g_is_zero) {
  CholPlan &C = P->chol;
  FrontTab tab = front_tab(C);
  solve_attrs();
  C.growth_used = C.growth_max;                                     // the solves decide with the bound in force here
  const size_t gw = (size_t)std::max(C.nsbtot, 1) * (2 + SPREP_NCNT / 2);
  if (!sb_g_is_zero)                                                // (a factorisation zeroes them in k_prep_pivots)
    SDM_HIP_CHECK(hipMemsetAsync(C.sb_g.p, 0, gw * sizeof(unsigned long long), P->stream));
  const int W = C.sbw;
  if (C.n_lt) SDM_KLAUNCH(P, k_ltrans, dim3(C.n_lt), dim3(ST), 0, C.fronts.p, C.LT.p, tab, C.l_lt.p, W);
  if (C.n_i128 == 0) return;
  if (C.n_i128 + C.n_items <= SPREP_MAX_ITEMS && !C.sprep_off) {    // everything resident at once: one launch, counters instead of boundaries
#ifdef SDM_EMU
    if (emu_concurrent() && C.n_i128 + C.n_items <= 200) {          // as on the device: its workgroups wait for each other's counters (one process each)
      SDM_KLAUNCH_CONCURRENT(P, k_sprep, dim3(C.n_i128 + C.n_items), dim3(ST), INV_LDS, C.fronts.p, C.S.p, C.ST.p, C.Tarena.p, tab, C.l_i128.p, C.n_i128,
                             C.l_items.p, C.sb_g.p, (int *)(C.sb_g.p + 2 * std::max(C.nsbtot, 1)), W, C.tmo.dev());
      return;
    }
#endif
    SDM_KLAUNCH(P, k_sprep, dim3(C.n_i128 + C.n_items), dim3(ST), INV_LDS, C.fronts.p, C.S.p, C.ST.p, C.Tarena.p, tab, C.l_i128.p, C.n_i128,
                C.l_items.p, C.sb_g.p, (int *)(C.sb_g.p + 2 * std::max(C.nsbtot, 1)), W, C.tmo.dev());
    return;
  }
  SDM_KLAUNCH(P, k_sinv128, dim3(C.n_i128), dim3(ST), INV_LDS, C.fronts.p, C.S.p, C.ST.p, tab, C.l_i128.p, C.sb_g.p, W);
  for (int st = 0; st < 2 * SINV_MAXLEV; st++) {
    const int n = C.stage_ptr[st + 1] - C.stage_ptr[st];
    if (n > 0) SDM_KLAUNCH(P, k_stile, dim3(n), dim3(ST), TILE_LDS, C.fronts.p, C.S.p, C.ST.p, C.Tarena.p, tab, C.l_items.p + 8 * (size_t)C.stage_ptr[st], C.sb_g.p, W);
  }
}

// growth statistics of the last solve_prepare (host read-back; tests and bench reporting)
void solve_stats(sdm_plan *P, sdm_int *nblocks, sdm_int *nbad, double *max_growth) {
  CholPlan &C = P->chol;
  std::vector<unsigned long long> g((size_t)std::max(2 * C.nsbtot, 2));
  SDM_HIP_CHECK(hipStreamSynchronize(P->stream));
  SDM_HIP_CHECK(hipMemcpy(g.data(), C.sb_g.p, g.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  sdm_int bad = 0; double mx = 0.0;
  for (int i = 0; i < C.nsbtot; i++) {
    union { unsigned long long u; double d; } a, b; a.u = g[2 * i]; b.u = g[2 * i + 1];
    const double gr = a.d * b.d;
    if (!(gr <= C.growth_used)) bad++;
    if (gr > mx || gr != gr) mx = gr;
  }
  if (nblocks) *nblocks = C.nsbtot;
  if (nbad) *nbad = bad;
  if (max_growth) *max_growth = mx;
}

// do the sweeps run the refinement launches for blocks beyond the growth bound?  (CholPlan::refine_mode; the note is read as it is
// now, without waiting for the device)
constexpr int REFINE_STEPS = 2;
// merged sweep launches (k_sfw_rows_diag): 0 never, 1 where rows beyond the next super-block exist, 2 every row launch of a one-front level
static int sweep_merge_pre() { const char *e = getenv("SEDUMI_HIP_SWEEP_PRE"); return e ? std::max(0, atoi(e)) : PRE_WGS; }
static int sweep_merge_level() {
  const char *e = getenv("SEDUMI_HIP_SWEEP_MERGE");               // (read per sweep: the tests switch it)
  return e ? atoi(e) : 1;
}
static bool solve_refines(CholPlan &C) {
  if (C.refine_mode != 1) return C.refine_mode == 2;
  if (C.noted.host) {
    const int done = ((volatile int *)C.noted.host)[1], bad = ((volatile int *)C.noted.host)[0];   // (done first: a sweep's own marks precede the next sweep's)
    if (bad > 0 && bad >= done) C.refine_on = true;                  // the latest sweep known to have run (or the one in flight) met such a block
    else if (done - bad >= 2) C.refine_on = false;                   // two sweeps have run since the last one that did
  }
  return C.refine_on;
}
// forward sweeps of nrhs right-hand sides side by side (grid.z): rhs + z*rhs_stride -> y + z*y_stride (permuted order);
// wv = update-vector scratch of wsize doubles per right-hand side
void solve_fw_batch(sdm_plan *P, const double *rhs, int64_t rhs_stride, double *y, int64_t y_stride, double *wv, int nrhs,
                    double *zdiv, const double *dscale, int l0, int l1, int what) {
  // levels l0 .. l1-1 (l1 < 0: all); what: 1 the assembly launches only (k_sfw_init), 2 everything but them, 3 both
  CholPlan &C = P->chol;
  const double thr = C.growth_used;
  const int W = C.sbw;
  FwBatch bt;
  bt.src = nrhs > 1 ? rhs_stride : 0; bt.y = nrhs > 1 ? y_stride : 0; bt.wv = nrhs > 1 ? C.wsize : 0;
  const FrontTab tab0 = front_tab(C);
  if (l1 < 0) l1 = C.nlevels;
  const bool refine = nrhs == 1 && solve_refines(C);
  // (inside a graph capture the sweep numbers would be baked into the kernel arguments and replayed stale: a captured sweep leaves
  // no notes and keeps the mode decided at capture; blocks beyond the bound it was not planned for are substituted, as accurate)
  int *noted = C.refine_mode == 1 && !P->capturing ? C.noted.dev() : nullptr;
  const int seq = P->capturing ? C.sweep_seq : ++C.sweep_seq;
  bool marked = false;
  for (int l = std::max(l0, 0); l < std::min(l1, C.nlevels); l++) {
    const SolveLevel &L = C.slev[l];
    if (L.nfronts == 0) continue;
    const int *list = C.d_levlist.p + C.levptr[l];
    const FrontTab tab = level_tab(C, tab0, l);
    const int gather = L.children ? 0 : 1;
    if (!gather && (what & 1)) SDM_KLAUNCH(P, k_sfw_init, dim3(L.nfronts, 1, nrhs), dim3(ST), 0, tab, list, wv, rhs, C.d_perm.p, y, bt);
    if (!(what & 2)) continue;
    bool diag_done = false;                                          // (this super-block's diagonal role ran inside the row launch before it)
    const int merge = sweep_merge_level();
    const bool may_merge = merge && W > 256 && nrhs == 1 && !refine && tab.one && L.nsb >= 2;
    for (int Pb = 0; Pb < L.nsb; Pb++) {
      const int nbmax = std::min(W, L.maxns - Pb * W);
      const dim3 gdiag(W <= 256 ? (nbmax + 15) / 16 : tri_grid(nbmax), L.nfronts, nrhs);
      if (!diag_done) SDM_KLAUNCH(P, k_sfw_diag, gdiag, dim3(ST), SDM_DIAG_SMEM(W), C.fronts.p, C.ST.p, tab, list, wv, rhs,
                  C.d_perm.p, y, C.sb_g.p, thr, Pb, gather, bt, zdiv, dscale, W, refine ? 1 : 0, C.refine_max, (const double *)nullptr, noted, seq, marked ? -1 : seq - 1, Pb == 0 && may_merge ? C.sweep_cnt.p : (int *)nullptr);
      marked = true;
      for (int it = 0; refine && it < REFINE_STEPS; it++) {            // (blocks within the bound leave these launches at once)
        SDM_KLAUNCH(P, k_sfw_resid, dim3((nbmax + 63) / 64, L.nfronts), dim3(RT), 0, C.fronts.p, tab, list, wv, rhs, C.d_perm.p, y, C.xfin.p, C.sb_g.p, thr,
                    C.refine_max, Pb, gather, W);
        SDM_KLAUNCH(P, k_sfw_diag, gdiag, dim3(ST), SDM_DIAG_SMEM(W), C.fronts.p, C.ST.p, tab, list, wv, rhs,
                    C.d_perm.p, y, C.sb_g.p, thr, Pb, gather, bt, zdiv, dscale, W, 2, C.refine_max, (const double *)C.xfin.p, (int *)nullptr, seq, -1, (int *)nullptr);
      }
      const int assign0 = (gather && Pb == 0) ? 1 : 0;
      diag_done = false;
      if (may_merge && L.maxns > (Pb + (merge >= 2 ? 1 : 2)) * W) {
        const int nrw = W > SEGN ? (L.maxns - (Pb + 1) * W + 1) / 2 : (L.maxns - (Pb + 1) * W + 3) / 4;
        const int nurg = std::min(nrw, W > SEGN ? W / 2 : W / 4), ndiag = tri_grid(std::min(W, L.maxns - (Pb + 1) * W));
#ifdef SDM_EMU
        // (tests/test_emu_concurrent.py: one process per workgroup, the diagonal role really waits for the urgent rows)
        if (emu_concurrent() && nrw + ndiag <= 256) SDM_KLAUNCH_CONCURRENT(P, k_sfw_rows_diag, dim3(nrw + ndiag), dim3(ST), SDM_MERGED_SMEM(W), C.LT.p, C.fronts.p, C.ST.p, tab, list, wv, rhs, C.d_perm.p, y, C.sb_g.p, thr, Pb,
                    assign0, bt, zdiv, dscale, W, C.refine_max, noted, seq, C.sweep_cnt.p, nurg, std::min(nrw - nurg, sweep_merge_pre()), ndiag, C.tmo.dev());
        else
#endif
        SDM_KLAUNCH(P, k_sfw_rows_diag, dim3(nrw + ndiag), dim3(ST), SDM_MERGED_SMEM(W), C.LT.p, C.fronts.p, C.ST.p, tab, list, wv, rhs, C.d_perm.p, y, C.sb_g.p, thr, Pb,
                    assign0, bt, zdiv, dscale, W, C.refine_max, noted, seq, C.sweep_cnt.p, nurg, std::min(nrw - nurg, sweep_merge_pre()), ndiag, C.tmo.dev());
        diag_done = true;
      } else if (L.maxns > (Pb + 1) * W)                             // the fronts' own rows of later super-blocks
        SDM_KLAUNCH(P, k_sfw_rows, dim3(W > SEGN ? (L.maxns - (Pb + 1) * W + 1) / 2 : (L.maxns - (Pb + 1) * W + 3) / 4, L.nfronts, nrhs), dim3(ST), 0, C.LT.p, tab, list, wv, rhs, C.d_perm.p, y, Pb,
                    assign0, bt, W);
      if (L.slabs_fw[Pb] > 0)                                        // the rows below the supernodes
        SDM_KLAUNCH(P, k_sfw_step, dim3(L.slabs_fw[Pb], L.nfronts, nrhs), dim3(ST), 0, C.fronts.p, tab, list, wv, y, Pb, assign0, bt, W);
    }
  }
}

// backward sweep in place on the vector y (permuted order).  dscale != null: ./d on the way in (k_sbw_init);
// skip_plain_init: the levels whose fronts have no rows below their own columns need no k_sbw_init at all (the ./d
// was already applied by the forward sweep's final writes)
static void solve_bw_inplace(sdm_plan *P, double *y, double *yout, const double *dscale, bool skip_plain_init, int l0 = 0, int l1 = -1) {
  CholPlan &C = P->chol;
  const double thr = C.growth_used;
  const int W = C.sbw;
  const FrontTab tab0 = front_tab(C);
  if (l1 < 0) l1 = C.nlevels;
  const bool refine = solve_refines(C);
  int *noted = C.refine_mode == 1 && !P->capturing ? C.noted.dev() : nullptr;
  const int seq = P->capturing ? C.sweep_seq : ++C.sweep_seq;
  bool marked = false;
  for (int l = std::min(l1, C.nlevels) - 1; l >= std::max(l0, 0); l--) {
    const SolveLevel &L = C.slev[l];
    if (L.nfronts == 0) continue;
    const int *list = C.d_levlist.p + C.levptr[l];
    const FrontTab tab = level_tab(C, tab0, l);
    if (!(skip_plain_init && !L.below))
      SDM_KLAUNCH(P, k_sbw_init, dim3((L.maxns + SROWS - 1) / SROWS, L.nfronts), dim3(ST), 0, C.fronts.p, tab, list, y, C.xfin.p, dscale);
    bool diag_done = false;
    const int merge = sweep_merge_level();
    const bool may_merge = merge && W > 256 && !refine && tab.one && L.nsb >= 2;
    for (int Q = L.nsb - 1; Q >= 0; Q--) {
      const int nbmax = std::min(W, L.maxns - Q * W);
      const dim3 gdiag(W <= 256 ? (nbmax + 15) / 16 : tri_grid(nbmax), L.nfronts);
      if (!diag_done) SDM_KLAUNCH(P, k_sbw_diag, gdiag, dim3(ST), SDM_DIAG_SMEM(W), C.fronts.p, C.S.p, tab, list, y, C.xfin.p, yout,
                  C.d_perm.p, C.sb_g.p, thr, Q, W, refine ? 1 : 0, C.refine_max, (const double *)nullptr, noted, seq, marked ? -1 : seq - 1,
                  Q == L.nsb - 1 && may_merge ? C.sweep_cnt.p : (int *)nullptr);
      marked = true;
      for (int it = 0; refine && it < REFINE_STEPS; it++) {
        SDM_KLAUNCH(P, k_sbw_resid, dim3((nbmax + 3) / 4, L.nfronts), dim3(ST), 0, C.fronts.p, tab, list, y, C.xfin.p, C.wvec.p, C.sb_g.p, thr, C.refine_max, Q, W);
        SDM_KLAUNCH(P, k_sbw_diag, gdiag, dim3(ST), SDM_DIAG_SMEM(W), C.fronts.p, C.S.p, tab, list, y, C.xfin.p, yout,
                    C.d_perm.p, C.sb_g.p, thr, Q, W, 2, C.refine_max, (const double *)C.wvec.p, (int *)nullptr, seq, -1, (int *)nullptr);
      }
      diag_done = false;
      const int nst = W > SEGN ? Q * (W / 2) : Q * (W / 4);
      if (may_merge && Q >= (merge >= 2 ? 1 : 2)) {
        const int nurg = W > SEGN ? W / 2 : W / 4, ndiag = tri_grid(W);
#ifdef SDM_EMU
        if (emu_concurrent() && nst + ndiag <= 256) SDM_KLAUNCH_CONCURRENT(P, k_sbw_step_diag, dim3(nst + ndiag), dim3(ST), SDM_MERGED_SMEM(W), C.fronts.p, C.S.p, tab, list, y, C.xfin.p, yout, C.d_perm.p, C.sb_g.p, thr, Q, W,
                    C.refine_max, noted, seq, C.sweep_cnt.p, nurg, std::min(nst - nurg, sweep_merge_pre()), ndiag, C.tmo.dev());
        else
#endif
        SDM_KLAUNCH(P, k_sbw_step_diag, dim3(nst + ndiag), dim3(ST), SDM_MERGED_SMEM(W), C.fronts.p, C.S.p, tab, list, y, C.xfin.p, yout, C.d_perm.p, C.sb_g.p, thr, Q, W,
                    C.refine_max, noted, seq, C.sweep_cnt.p, nurg, std::min(nst - nurg, sweep_merge_pre()), ndiag, C.tmo.dev());
        diag_done = true;
      } else if (Q > 0) SDM_KLAUNCH(P, k_sbw_step, dim3(nst, L.nfronts), dim3(ST), 0, C.fronts.p, tab, list, y, C.xfin.p, Q, W);
    }
  }
}

// the fw, ./d, bw solve of a right-hand side in P->rhs level by level (sdm_plan_solve_levels; the multi-GPU layer reduces the
// assembled vectors of the separator fronts between `what` 1 and 2 and broadcasts their solution before `what` 4)
void solve_levels(sdm_plan *P, int what, int l0, int l1) {
  CholPlan &C = P->chol;
  if (what & 3) solve_fw_batch(P, P->rhs.p, 0, P->ywork.p, 0, C.wvec.p, 1, C.zdiv.p, solve_d(P), l0, l1, what & 3);
  if (what & 4) solve_bw_inplace(P, C.zdiv.p, P->y.p, nullptr, true, l0, l1);
}
void solve_run(sdm_plan *P, const double *rhs, double *yout, int mode) {
  CholPlan &C = P->chol;
  const size_t mb = (size_t)C.m * sizeof(double);
  double *y = P->ywork.p;
  // with a resident dense-column factor (sdm_plan_deninfac) the complete solve is wrapPcg.m:56-59:
  //   p = fwdpr1(Lden, L \ r(perm)) ;  y = p ./ L.d ;  y(perm) = L' \ bwdpr1(Lden, y)
  const bool dense = (mode == 7) && P->dense.factored;
  // fw, ./d, bw in one call without dense columns: the forward sweep writes the ./d copy of every block as it becomes
  // final (zdiv), the backward sweep runs on that copy and needs k_sbw_init only where rows below a supernode exist
  const bool fold = (mode == 7) && !dense;
  if (mode & 1) {
    solve_fw_batch(P, rhs, 0, y, 0, C.wvec.p, 1, fold ? C.zdiv.p : nullptr, fold ? solve_d(P) : nullptr);
    if (!(mode & 4)) {
      if (mode & 2) vec_divd(P, y);
      SDM_HIP_CHECK(hipMemcpyAsync(yout, y, mb, hipMemcpyDeviceToDevice, P->stream));
      return;
    }
  } else {
    SDM_HIP_CHECK(hipMemcpyAsync(y, rhs, mb, hipMemcpyDeviceToDevice, P->stream));
  }
  if (dense) {
    dense_prodform(P, y, /*with_divide=*/true);                      // fwdpr1, ./ Ld, bwdpr1 in one launch
    solve_bw_inplace(P, y, yout, nullptr, false);
  } else if (fold) {
    solve_bw_inplace(P, C.zdiv.p, yout, nullptr, true);
  } else {
    solve_bw_inplace(P, y, yout, (mode & 2) ? solve_d(P) : (const double *)nullptr, false);
  }
}

}  // namespace sdm

#if defined(SDM_PHASES) && !defined(SDM_EMU)
// tools-only build (python -m sedumi_amd.build --phases): read / reset the in-kernel phase clocks of this file
extern "C" int sdm_debug_phases_solve(unsigned long long *out32, int reset) {
  if (out32 && hipMemcpyFromSymbol(out32, HIP_SYMBOL(sdm_phase_acc), 32 * sizeof(unsigned long long)) != hipSuccess) return 1;
  if (reset) { unsigned long long z[32] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(sdm_phase_acc), z, sizeof(z)) != hipSuccess) return 1; }
  return 0;
}
#endif

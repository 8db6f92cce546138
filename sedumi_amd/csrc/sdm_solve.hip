// sdm_solve.hip -- triangular solves  y = L \ b(perm),  y(perm) = L' \ b  (fwblkslv.c:77-134, bwblkslv.c:73-125)
// for gfx950, built around EXPLICIT INVERSES of the diagonal super-blocks.
//
// The reference substitutes column by column -- a chain of m dependent steps.  A single right-hand side leaves a
// GPU nothing to batch over (the four solves of an IPM iteration depend on each other, wrapPcg.m:56-59), so the chain
// itself has to go.  After every factorisation (solve_prepare) each front gets a second array S (ns x ns):
//   * the diagonal super-blocks of SBW = 256 columns are inverted explicitly:  S_PP = inv(L_PP)  (unit lower
//     triangular; 64x64 blocks by substitution in registers, then two levels of  X21 = -inv(C) B inv(A)  on the FP64
//     matrix cores),
//   * the block rows left of them are premultiplied:  S_PQ = inv(L_PP) L_PQ  (FP64 matrix cores),
// i.e. L = D~ L~ with D~ = blockdiag(L_PP) and L~ = S with identity diagonal super-blocks.  Then
//   forward   y_P = inv(L_PP) b_P - sum_{Q<P} S_PQ y_Q          : one GEMV launch for all inv(L_PP) b_P, then one GEMV
//                                                                  launch per super-block column P ("step"),
//   backward  v_P = z_P - sum_{Q>P} S_QP' v_Q ,  x_P = inv(L_PP)' v_P : the mirror image,
// every launch a plain HBM-streaming matrix-vector product over MANY workgroups (16 rows or columns each) with no
// dependency inside it: m/256 dependent steps per sweep instead of m.  Rows below a supernode's own columns (they
// belong to its ancestors) are never premultiplied and are read from the factor itself.
//
// Never-fail pivoting admits multipliers up to maxu = 5e5 (blkchol2.c:114-161), so an explicit inverse can be
// ill-conditioned.  The growth of every super-block is therefore measured when it is inverted
// (max|inv(L_PP)| * max|L_PP|); a block beyond CholPlan::growth_max keeps its rows unpremultiplied and is solved by
// substitution by the workgroup that completes its right-hand side (an arrival ticket among the workgroups that
// update it) -- per block, decided on the device, no host round trip.  Everything is deterministic (fixed summation
// orders, no atomics on data).
#include "sdm_plan.h"
#include <algorithm>

namespace sdm {

constexpr int ST = 256;          // work-items per workgroup of every kernel in this file
constexpr int TP = 65;           // LDS pitch of staged 64-wide operand blocks (conflict-free transposing stores)
constexpr size_t INV_LDS = (size_t)4 * 64 * TP * sizeof(double);      // k_sinv128: four staged 64x64 blocks
constexpr int SPREP_MAX_ITEMS = 256;                                  // k_sprep: one workgroup per item, all resident (one per CU)
constexpr size_t TILE_LDS = (size_t)2 * 64 * TP * sizeof(double);     // k_stile: one A and one B operand block
constexpr int SFRONT_MAX_WGS = 512;                                   // k_solve_front: all workgroups resident (three fit a compute unit)

// ---------------------------------------------------------------- host tables
void solve_build(sdm_plan *P) {
  CholPlan &C = P->chol;
  const int nsuper = (int)C.nsuper;
  C.sn_soff.assign(nsuper, 0); C.sn_sld.assign(nsuper, 0); C.sn_sboff.assign(nsuper, 0);
  std::vector<int> i128, t3, pm;
  int64_t soff = 0; int sb = 0, tslots = 0;
  for (int s = 0; s < nsuper; s++) {
    const int ns = C.sn_ns[s], sld = ns + (ns & 1);
    C.sn_soff[s] = soff; C.sn_sld[s] = sld; C.sn_sboff[s] = sb;
    soff += (int64_t)sld * ns;
    const int nsb = (ns + SBW - 1) / SBW;
    for (int h = 0; 128 * h < ns; h++) { i128.push_back(s); i128.push_back(h); i128.push_back(0); i128.push_back(0); }
    for (int Pb = 0; Pb < nsb; Pb++) {
      const int k0 = Pb * SBW, nb = std::min(SBW, ns - k0);
      if (nb > 128) {                                              // level 3:  X = -inv(C2) B2 inv(A2), via T = B2 inv(A2)
        const int nc = nb - 128;
        for (int I = 0; 64 * I < nc; I++)
          for (int J = 0; J < 2; J++) { t3.push_back(s); t3.push_back(Pb); t3.push_back(2 * I + J); t3.push_back(tslots); }
        tslots++;
      }
      if (Pb > 0)
        for (int I = 0; 64 * I < nb; I++)
          for (int J = 0; J < 4 * Pb; J++) { pm.push_back(s); pm.push_back(Pb); pm.push_back(I); pm.push_back(J); }
    }
    sb += nsb;
  }
  C.ssize = soff; C.nsbtot = sb;
  C.n_i128 = (int)i128.size() / 4; C.n_t3 = (int)t3.size() / 4; C.n_pm = (int)pm.size() / 4;
  C.l_i128.upload(i128); C.l_t3.upload(t3); C.l_pm.upload(pm);
  C.d_soff.upload(C.sn_soff); C.d_sld.upload(C.sn_sld); C.d_sboff.upload(C.sn_sboff);
  C.S.alloc((size_t)std::max<int64_t>(soff, 1));
  SDM_HIP_CHECK(hipMemset(C.S.p, 0, (size_t)std::max<int64_t>(soff, 1) * sizeof(double)));   // upper triangles stay zero for good
  C.xfin.alloc((size_t)std::max<sdm_int>(C.m, 1)); C.zdiv.alloc((size_t)std::max<sdm_int>(C.m, 1));
  C.ttmp.alloc((size_t)std::max(tslots, 1) * 128 * 128);
  // growth records (2 per super-block), then the completion counters of k_sprep (4 ints = 2 words per super-block)
  C.sb_g.alloc((size_t)std::max(4 * sb, 4)); C.sb_cnt.alloc((size_t)std::max(sb, 1));
  SDM_HIP_CHECK(hipMemset(C.sb_g.p, 0, (size_t)std::max(4 * sb, 4) * sizeof(unsigned long long)));
  SDM_HIP_CHECK(hipMemset(C.sb_cnt.p, 0, (size_t)std::max(sb, 1) * sizeof(int)));
  C.sfront_cnt.alloc((size_t)2 * std::max(sb, 1) + 4);
  SDM_HIP_CHECK(hipMemset(C.sfront_cnt.p, 0, C.sfront_cnt.n * sizeof(int)));
  // fw, ./d, bw of a one-front factor without rows below as ONE launch (k_solve_front): opt-in until it has been timed
  C.solve_fused = getenv("SDM_SOLVE_FUSED") != nullptr && nsuper == 1 && C.sn_ms[0] == C.sn_ns[0] &&
                  (C.sn_ns[0] + SROWS - 1) / SROWS <= SFRONT_MAX_WGS;
  // levels
  C.slev.assign(C.nlevels, SolveLevel());
  for (int l = 0; l < C.nlevels; l++) {
    SolveLevel &L = C.slev[l];
    L.nfronts = C.levptr[l + 1] - C.levptr[l];
    int nsteps = 0;
    for (int i = C.levptr[l]; i < C.levptr[l + 1]; i++) {
      const int s = C.levlist[i], ns = C.sn_ns[s], ms = C.sn_ms[s];
      L.maxns = std::max(L.maxns, ns); L.maxms = std::max(L.maxms, ms);
      if (C.childptr[s + 1] > C.childptr[s]) L.children = true;
      if (ms > ns) L.below = true;
      const int nsb = (ns + SBW - 1) / SBW;
      nsteps = std::max(nsteps, nsb - 1 + (ms > ns ? 1 : 0));
    }
    L.nsb = (L.maxns + SBW - 1) / SBW;
    L.maxslab_fw.assign(nsteps, 0);
    for (int i = C.levptr[l]; i < C.levptr[l + 1]; i++) {
      const int s = C.levlist[i], ns = C.sn_ns[s], ms = C.sn_ms[s];
      for (int Pb = 0; Pb < nsteps && Pb * SBW < ns; Pb++) {
        const int ra = (Pb + 1) * SBW;
        const int slabsA = ns > ra ? (ns - ra + SROWS - 1) / SROWS : 0;
        const int slabsB = ms > ns ? (ms - (ns & ~1) + SROWS - 1) / SROWS : 0;
        L.maxslab_fw[Pb] = std::max(L.maxslab_fw[Pb], slabsA + slabsB);
      }
    }
  }
}

// ================================================================ device helpers
__device__ __forceinline__ double bits_to_double(unsigned long long u) { union { unsigned long long u; double d; } b; b.u = u; return b.d; }
__device__ __forceinline__ unsigned long long double_to_bits(double d) { union { unsigned long long u; double d; } b; b.d = d; return b.u; }
// growth check of super-block sb: max|inv| * max|L| within bounds (NaN counts as bad)
__device__ __forceinline__ bool sb_is_bad(const unsigned long long *g, int sb, double thr) {
  return !(bits_to_double(g[2 * sb]) * bits_to_double(g[2 * sb + 1]) <= thr);
}
// max over the wavefront, then one order-independent atomicMax on the bit pattern of a non-negative double
__device__ __forceinline__ void wave_atomic_max(unsigned long long *dst, double v, int lane) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const double t = __shfl_xor(v, o); v = (t > v || t != t) ? t : v; }      // NaN wins
  if (lane == 0 && v > 0.0) atomicMax(dst, double_to_bits(v));
  if (lane == 0 && !(v == v)) atomicMax(dst, 0x7ff8000000000000ull);   // NaN: larger than every finite pattern
}

// ---- 64x64 (x K) product tiles on the FP64 matrix cores.  Workgroup of 256: wavefront w owns the 32x32 quadrant
// (w & 1, w >> 1) = 2 x 2 tiles of v_mfma_f64_16x16x4_f64.  Operands staged in LDS as As[k*TP + row], Bs[k*TP + col].
struct Acc22 { sdm_double4 t[2][2]; };
__device__ __forceinline__ void acc_zero(Acc22 &a) {
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 4; r++) a.t[i][j][r] = 0.0;
}
__device__ __forceinline__ void mma_block(Acc22 &acc, const double *As, const double *Bs, int wave, int lane) {
  const int rb = 32 * (wave & 1) + (lane & 15), cb = 32 * (wave >> 1) + (lane & 15), kq = lane >> 4;
  // the operands of step kk+4 are fetched from LDS while the four products of step kk issue
  double a0 = As[kq * TP + rb], a1 = As[kq * TP + rb + 16], b0 = Bs[kq * TP + cb], b1 = Bs[kq * TP + cb + 16];
#pragma unroll
  for (int kk = 0; kk < 64; kk += 4) {
    const int kn = min(kk + 4, 60) + kq;
    const double na0 = As[kn * TP + rb], na1 = As[kn * TP + rb + 16], nb0 = Bs[kn * TP + cb], nb1 = Bs[kn * TP + cb + 16];
    acc.t[0][0] = SDM_MFMA_F64_16x16x4(a0, b0, acc.t[0][0]);
    acc.t[0][1] = SDM_MFMA_F64_16x16x4(a0, b1, acc.t[0][1]);
    acc.t[1][0] = SDM_MFMA_F64_16x16x4(a1, b0, acc.t[1][0]);
    acc.t[1][1] = SDM_MFMA_F64_16x16x4(a1, b1, acc.t[1][1]);
    a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
  }
}
// accumulator -> LDS as Cs[row*TP + col] (the layout of a B operand whose k index is the row) scaled by sgn
__device__ __forceinline__ void acc_to_lds_rowmajor(const Acc22 &acc, double *Cs, int wave, int lane, double sgn) {
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = 32 * (wave & 1) + 16 * i + (lane >> 4) + 4 * r, col = 32 * (wave >> 1) + 16 * j + (lane & 15);
        Cs[row * TP + col] = sgn * acc.t[i][j][r];
      }
}
// Operand staging in two halves -- all global loads of a 64x64 block first (16 per work-item, addresses clamped,
// unconditional), the LDS stores later -- so that one memory latency is paid per block, not one per element.
// column-major operand: dst[k*TP + r] = src[k*ld + r] for r < nr, k < nk, zero elsewhere
constexpr int SPT = 64 * 64 / ST;      // elements per work-item
// WT: sc1 loads (what another workgroup of the SAME launch stored write-through is read without an acquire fence)
template <bool WT = false>
__device__ __forceinline__ void stage_colmajor_load(double (&v)[SPT], const double *src, int64_t ld, int nr, int nk, int tid) {
  const int r = min(tid & 63, nr - 1), kq = tid >> 6;
#pragma unroll
  for (int j = 0; j < SPT; j++) {
    const double *a = &src[(int64_t)min(kq + (ST / 64) * j, nk - 1) * ld + r];
    v[j] = WT ? sdm_load_wt(a) : *a;
  }
}
__device__ __forceinline__ double stage_colmajor_store(double *dst, const double (&v)[SPT], int nr, int nk, int tid) {
  const int r = tid & 63, kq = tid >> 6;
  double mx = 0.0;
#pragma unroll
  for (int j = 0; j < SPT; j++) {
    const int k = kq + (ST / 64) * j;
    const double x = (r < nr && k < nk) ? v[j] : 0.0;
    dst[k * TP + r] = x;
    mx = fabs(x) > mx ? fabs(x) : mx;
  }
  return mx;
}
// the same transposed: dst[k*TP + c] = src[c*ld + k] for k < nk, c < nc, zero elsewhere
template <bool WT = false>
__device__ __forceinline__ void stage_transposed_load(double (&v)[SPT], const double *src, int64_t ld, int nk, int nc, int tid) {
  const int k = min(tid & 63, nk - 1), cq = tid >> 6;
#pragma unroll
  for (int j = 0; j < SPT; j++) {
    const double *a = &src[(int64_t)min(cq + (ST / 64) * j, nc - 1) * ld + k];
    v[j] = WT ? sdm_load_wt(a) : *a;
  }
}
__device__ __forceinline__ void stage_transposed_store(double *dst, const double (&v)[SPT], int nk, int nc, int tid) {
  const int k = tid & 63, cq = tid >> 6;
#pragma unroll
  for (int j = 0; j < SPT; j++) {
    const int c = cq + (ST / 64) * j;
    dst[k * TP + c] = (k < nk && c < nc) ? v[j] : 0.0;
  }
}
// Cs[row*TP + col] (LDS) -> column-major destination, rows < nr, cols < nc; returns max |value| written
// WT: write-through stores (the tile is read by other workgroups of the SAME launch, k_sprep)
template <bool WT = false>
__device__ __forceinline__ double store_tile(double *dst, int64_t ld, const double *Cs, int nr, int nc, int tid) {
  const int r = tid & 63, cq = tid >> 6;
  double mx = 0.0;
#pragma unroll 4
  for (int c = cq; c < 64; c += ST / 64)
    if (r < nr && c < nc) {
      const double v = Cs[r * TP + c];
      if (WT) sdm_store_wt(&dst[(int64_t)c * ld + r], v); else dst[(int64_t)c * ld + r] = v;
      mx = fabs(v) > mx ? fabs(v) : mx;
    }
  return mx;
}

// ================================================================ inversion of the diagonal super-blocks
// One workgroup per 128-column block h of a front, bottom-up, everything in LDS / registers:
//   32x32  each of the four wavefronts inverts one 32x32 unit lower triangular diagonal block by columns (lane j owns
//          column j of the inverse in registers; the entries of L come as broadcast LDS reads at compile-time offsets);
//   64x64  X10 = -inv(A11) (A10 inv(A00)) for the two 64-column blocks A and C (matrix cores, two wavefronts each);
//   128    X21 = -inv(C) (B inv(A)) on the FP64 matrix cores, B = L(C rows, A columns) requested at the very start.
// Results go to S; max|inv| and max|L| to sb_g (growth check).
template <bool WT>
__device__ __forceinline__ void sinv128_body(char *smem, const double *__restrict__ F, double *__restrict__ S, const FrontTab &tab,
                                             const int *it, unsigned long long *sb_g) {
  double *bufA = (double *)smem, *bufC = bufA + 64 * TP, *bufB = bufC + 64 * TP, *bufT = bufB + 64 * TP;
  const int s = it[0], h = it[1];
  const int ns = tab.ns[s], ld = tab.ld[s], sld = tab.sld[s];
  const double *Fs = F + tab.foff[s];
  double *Ss = S + tab.soff[s];
  const int k0 = 128 * h, nbA = min(64, ns - k0), nbC = max(0, min(64, ns - k0 - 64));
  unsigned long long *gP = sb_g + 2 * (tab.sboff[s] + k0 / SBW);
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
  const int blk = wave >> 1, q = wave & 1;                          // wavefront -> (64-block A / C, 32-block inside it)
  const double *raw = blk == 0 ? rawA : rawC;
  double *dst = blk == 0 ? bufA : bufC;
  {
    // ---- 32x32 by columns: lane j owns column j of the inverse in registers, X(i, j) = delta_ij - sum_{k<i} L(i, k) X(k, j);
    // L(i, k) is the same for every lane: one broadcast LDS read at a compile-time offset per term, no cross-lane traffic
    // (the earlier row form spent 2 v_readlane + 1 FMA per term on the chain: 6.8 us; this one 496 pipelined reads + FMAs)
    const int j = lane & 31;
    const double *Lb = raw + (32 * q) * TP + 32 * q;                 // Lb[k*TP + i] = L(i, k) of this 32-block
    double X[32];
#pragma unroll
    for (int i = 0; i < 32; i++) {
      // row i's coefficients are read one row ahead of their use; the dependence on X[i-2] keeps the compiler from hoisting ALL
      // 496 reads to the top (which it does otherwise -- and then spills them: 79 us instead of 3)
      double lrow[32];
      const int z = i >= 2 ? SDM_ZERO_AFTER(X[i - 2]) : 0;           // an opaque 0: row i's reads cannot be issued before row i-2 is done
#pragma unroll
      for (int k = 0; k < i; k++) lrow[k] = Lb[k * TP + i + z];
      double a0 = (i == j) ? 1.0 : 0.0, a1 = 0.0;                    // two partial sums: half the dependent FMA chain
#pragma unroll
      for (int k = 0; k < i; k++) {
        if (k & 1) a1 -= lrow[k] * X[k]; else a0 -= lrow[k] * X[k];
      }
      X[i] = a0 + a1;
    }
    double gm = 0.0;
    if (lane < 32) {
#pragma unroll
      for (int i = 0; i < 32; i++) {
        gm = fmax(gm, fabs(X[i]));
        // inv(A) is kept as a B operand [k*TP + col] = inv(k, col); inv(C) as an A operand [k*TP + row] = inv(row, k)
        if (blk == 0) dst[(32 * q + i) * TP + 32 * q + j] = X[i]; else dst[(32 * q + j) * TP + 32 * q + i] = X[i];
      }
    }
    wave_atomic_max(gP, gm, lane);
  }
  SDM_PHASE(1);
  __syncthreads();
  SDM_PHASE(2);
  {
    // ---- 64x64: X10 = -inv11 (L10 inv00) per 64-block on the FP64 matrix cores: two wavefronts per block, wavefront q owns
    // the two 16x16 tiles of output columns 16q .. 16q+15 of each product, K = 32 = 8 steps of v_mfma_f64_16x16x4_f64.
    // T goes to the unused upper right quadrant of the raw buffer (rows 32.., columns < 32 of raw[k*TP + i] hold zeros).
    const int li = lane & 15, lk = lane >> 4;
    double *Ts = (blk == 0 ? rawA : rawC) + 32 * TP;                 // Ts[r*TP + c]
    sdm_double4 acc[2];
    for (int t = 0; t < 2; t++) for (int r = 0; r < 4; r++) acc[t][r] = 0.0;
#pragma unroll
    for (int s4 = 0; s4 < 8; s4++) {
      const int k = 4 * s4 + lk;
      const double b = blk == 0 ? dst[k * TP + 16 * q + li] : dst[(16 * q + li) * TP + k];          // inv00(k, 16q + li)
#pragma unroll
      for (int t = 0; t < 2; t++) acc[t] = SDM_MFMA_F64_16x16x4(raw[k * TP + 32 + 16 * t + li], b, acc[t]);   // L10(16t + li, k)
    }
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int r = 0; r < 4; r++) Ts[(16 * t + lk + 4 * r) * TP + 16 * q + li] = acc[t][r];
    __syncthreads();
    for (int t = 0; t < 2; t++) for (int r = 0; r < 4; r++) acc[t][r] = 0.0;
#pragma unroll
    for (int s4 = 0; s4 < 8; s4++) {
      const int k = 4 * s4 + lk;
      const double b = Ts[k * TP + 16 * q + li];                                                    // T(k, 16q + li)
#pragma unroll
      for (int t = 0; t < 2; t++) {
        const int rr = 16 * t + li;
        const double a = blk == 0 ? dst[(32 + rr) * TP + 32 + k] : dst[(32 + k) * TP + 32 + rr];    // inv11(rr, k)
        acc[t] = SDM_MFMA_F64_16x16x4(a, b, acc[t]);
      }
    }
    double gm = 0.0;
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int rr = 16 * t + lk + 4 * r, c = 16 * q + li;
        const double v = -acc[t][r];
        gm = fmax(gm, fabs(v));
        if (blk == 0) dst[(32 + rr) * TP + c] = v; else dst[c * TP + 32 + rr] = v;
      }
    wave_atomic_max(gP, gm, lane);
  }
  SDM_PHASE(3);
  __syncthreads();                                                  // bufA = inv(A) (B operand), bufC = inv(C) (A operand); raw buffers free
  if (nbC > 0) lmx = fmax(lmx, stage_colmajor_store(bufB, vB, nbC, 64, tid));
  wave_atomic_max(gP + 1, lmx, lane);
  // inverses to S (lower triangles incl. the unit diagonal; the upper triangles of S are zero and stay zero)
  for (int e = tid; e < 64 * 64; e += ST) {
    const int i = e & 63, j = e >> 6;
    if (i >= j && i < nbA) { if (WT) sdm_store_wt(&Ss[(int64_t)(k0 + j) * sld + k0 + i], bufA[i * TP + j]); else Ss[(int64_t)(k0 + j) * sld + k0 + i] = bufA[i * TP + j]; }
    if (i >= j && i < nbC) { if (WT) sdm_store_wt(&Ss[(int64_t)(k0 + 64 + j) * sld + k0 + 64 + i], bufC[j * TP + i]); else Ss[(int64_t)(k0 + 64 + j) * sld + k0 + 64 + i] = bufC[j * TP + i]; }
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
  const double gm = store_tile<WT>(Ss + (int64_t)k0 * sld + k0 + 64, sld, bufB, nbC, 64, tid);
  wave_atomic_max(gP, gm, lane);
  SDM_PHASE(6);
}
__global__ void __launch_bounds__(ST)
k_sinv128(const double *__restrict__ F, double *__restrict__ S, FrontTab tab, const int *items, unsigned long long *sb_g) {
  SDM_DYN_SMEM(smem);
  sinv128_body<false>(smem, F, S, tab, items + 4 * blockIdx.x, sb_g);
}

// Generic product tile  C(64x64) = sgn * sum_k A(:,k) B(k,:)  for the remaining stages:
//   mode 0  T(I,J)  = sum_{K>=J} B2(I,K) inv(A2)(K,J)          (level 3, first half; into the scratch ttmp)
//   mode 1  X(I,J)  = - sum_{K<=I} inv(C2)(I,K) T(K,J)         (level 3, second half; into S)
//   mode 2  S_PQ tile (I,J) = sum_{K<=I} inv(L_PP)(I,K) L(P rows K, columns J)     (premultiplication; into S)
template <bool WT>
__device__ __forceinline__ void stile_body(char *smem, const double *F, double *S, double *ttmp, const FrontTab &tab, const int *it,
                                           unsigned long long *sb_g, int mode, double thr) {
  double *As = (double *)smem, *Bs = As + 64 * TP;
  const int s = it[0], Pb = it[1];
  const int ns = tab.ns[s], ld = tab.ld[s], sld = tab.sld[s];
  const double *Fs = F + tab.foff[s];
  double *Ss = S + tab.soff[s];
  const int k0 = Pb * SBW, nb = min(SBW, ns - k0);
  unsigned long long *gP = sb_g + 2 * (tab.sboff[s] + Pb);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const double *Ap, *Bp; double *Cp;
  int64_t lda, ldb, ldc;
  int arows, kvalid;
  double sgn = 1.0;
  bool track_l = false, track_g = false;
  if (mode == 0) {
    const int I = it[2] >> 1, J = it[2] & 1;
    arows = min(64, nb - 128 - 64 * I); kvalid = 128 - 64 * J;
    Ap = Fs + (int64_t)(k0 + 64 * J) * ld + k0 + 128 + 64 * I; lda = ld;
    Bp = Ss + (int64_t)(k0 + 64 * J) * sld + k0 + 64 * J; ldb = sld;
    Cp = ttmp + (int64_t)it[3] * 128 * 128 + (int64_t)(64 * J) * 128 + 64 * I; ldc = 128;
    track_l = true;
  } else if (mode == 1) {
    const int I = it[2] >> 1, J = it[2] & 1;
    arows = min(64, nb - 128 - 64 * I); kvalid = min(64 * (I + 1), nb - 128);
    Ap = Ss + (int64_t)(k0 + 128) * sld + k0 + 128 + 64 * I; lda = sld;
    Bp = ttmp + (int64_t)it[3] * 128 * 128 + (int64_t)(64 * J) * 128; ldb = 128;
    Cp = Ss + (int64_t)(k0 + 64 * J) * sld + k0 + 128 + 64 * I; ldc = sld;
    sgn = -1.0; track_g = true;
  } else {
    if (WT ? !(bits_to_double(sdm_load_wt_u64(&sb_g[2 * (tab.sboff[s] + Pb)])) * bits_to_double(sdm_load_wt_u64(&sb_g[2 * (tab.sboff[s] + Pb) + 1])) <= thr)
           : sb_is_bad(sb_g, tab.sboff[s] + Pb, thr)) return;       // this block row stays unpremultiplied
    const int I = it[2], J = it[3];
    arows = min(64, nb - 64 * I); kvalid = min(64 * (I + 1), nb);
    Ap = Ss + (int64_t)k0 * sld + k0 + 64 * I; lda = sld;
    Bp = Fs + (int64_t)(64 * J) * ld + k0; ldb = ld;
    Cp = Ss + (int64_t)(64 * J) * sld + k0 + 64 * I; ldc = sld;
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
  SDM_PHASE(8 + 4 * mode);
  if (track_l) wave_atomic_max(gP + 1, lmx, lane);
  acc_to_lds_rowmajor(acc, As, wave, lane, sgn);
  __syncthreads();
  const double gm = store_tile<WT>(Cp, ldc, As, arows, 64, tid);
  if (track_g) wave_atomic_max(gP, gm, lane);
  SDM_PHASE(9 + 4 * mode);
#if defined(SDM_PHASES) && !defined(SDM_EMU)
  if (tid == 0) atomicAdd(&sdm_phase_acc[10 + 4 * mode], 1ull);
#endif
}
__global__ void __launch_bounds__(ST)
k_stile(const double *F, double *S, double *ttmp, FrontTab tab, const int *items, unsigned long long *sb_g, int mode, double thr) {
  SDM_DYN_SMEM(smem);
  stile_body<false>(smem, F, S, ttmp, tab, items + 4 * blockIdx.x, sb_g, mode, thr);
}

// ---- all of the above in ONE launch for problems whose items fit the device at once (k_sprep): workgroups take the
// items in the order 128-blocks, level-3 first halves, level-3 second halves, premultiplication tiles, and wait on
// per-super-block completion counters instead of on launch boundaries (three of them, ~4 us each, and their tails).
// Producers store write-through, count after their stores are acknowledged; consumers poll with acquire loads.
// cnt[4*sb + 0/1/2] = finished 128-blocks / first halves / second halves of super-block sb (zeroed with sb_g).
__device__ __forceinline__ void prep_wait(const int *cnt, int target, int *tmo) {
  if (threadIdx.x == 0) {
    long it = 0;
    for (; sdm_signal_load(cnt) < target && it < (1L << 21); it++) SDM_SPIN_PAUSE();
    if (it == (1L << 21)) sdm_raise_flag(tmo);
  }
  __syncthreads();                                                  // no acquire fence: everything waited for is read with sc1 loads
}
__device__ __forceinline__ void prep_done(int *cnt) {
  SDM_STORES_DONE();
  __syncthreads();
  if (threadIdx.x == 0) sdm_signal_add(cnt);
}
__global__ void __launch_bounds__(ST)
k_sprep(const double *F, double *S, double *ttmp, FrontTab tab, const int *l_i128, int n_i128, const int *l_t3, int n_t3,
        const int *l_pm, int n_pm, unsigned long long *sb_g, int *cnt, double thr, int *tmo) {
  SDM_DYN_SMEM(smem);
  int b = blockIdx.x;
  if (b < n_i128) {
    const int *it = l_i128 + 4 * b;
    sinv128_body<true>(smem, F, S, tab, it, sb_g);
    // the growth maxima (atomicMax, device scope) are read by the premultiplication tiles of this launch as well
    prep_done(cnt + 4 * (tab.sboff[it[0]] + (128 * it[1]) / SBW));
    return;
  }
  b -= n_i128;
  const int mode = b < n_t3 ? 0 : (b < 2 * n_t3 ? 1 : 2);
  const int *it = mode == 2 ? l_pm + 4 * (b - 2 * n_t3) : l_t3 + 4 * (b - mode * n_t3);
  const int s = it[0], Pb = it[1];
  const int nb = min(SBW, tab.ns[s] - Pb * SBW);
  const int n128 = (nb + 127) / 128, nt = nb > 128 ? 2 * ((nb - 128 + 63) / 64) : 0;
  int *c = cnt + 4 * (tab.sboff[s] + Pb);
  prep_wait(c, n128, tmo);
  if (mode == 1) prep_wait(c + 1, nt, tmo);
  if (mode == 2 && nt > 0) prep_wait(c + 2, nt, tmo);
  stile_body<true>(smem, F, S, ttmp, tab, it, sb_g, mode, thr);
  if (mode < 2) prep_done(c + 1 + mode);
}

// ================================================================ substitution fallback for one super-block
// Rare path (growth check failed): L_PP y = r  /  L_PP' x = v  in place on nb <= SBW entries at yp, by ONE workgroup.
// Fs = front, (k0, k0) = position of the block.  Sd = 64*TP doubles, w = SBW doubles of LDS.
__device__ __noinline__ void block_solve_fw(const double *Fs, int ld, int k0, int nb, double *yp, double *w, double *Sd) {
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < nb; i += ST) w[i] = yp[i];
  __syncthreads();
  for (int kk = 0; kk < nb; kk += 64) {
    const int kb = min(64, nb - kk);
    for (int e = tid; e < 64 * 64; e += ST) {
      const int i = e & 63, c = e >> 6;
      Sd[c * TP + i] = (i > c && i < kb) ? Fs[(int64_t)(k0 + kk + c) * ld + k0 + kk + i] : 0.0;
    }
    __syncthreads();
    if (tid < 64) {
      double wi = lane < kb ? w[kk + lane] : 0.0;
      for (int k = 0; k < 64; k++) wi -= Sd[k * TP + lane] * sdm_bcast_lane(wi, k);
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
  for (int i = tid; i < nb; i += ST) yp[i] = w[i];
}
__device__ __noinline__ void block_solve_bw(const double *Fs, int ld, int k0, int nb, double *yp, double *w, double *Sd) {
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < nb; i += ST) w[i] = yp[i];
  __syncthreads();
  for (int kk = ((nb - 1) / 64) * 64; kk >= 0; kk -= 64) {
    const int kb = min(64, nb - kk);
    for (int e = tid; e < 64 * 64; e += ST) {
      const int i = e & 63, c = e >> 6;
      Sd[c * TP + i] = (i > c && i < kb) ? Fs[(int64_t)(k0 + kk + c) * ld + k0 + kk + i] : 0.0;
    }
    __syncthreads();
    if (tid < 64) {
      double xj = lane < kb ? w[kk + lane] : 0.0;
      for (int k = 63; k >= 0; k--) xj -= Sd[lane * TP + k] * sdm_bcast_lane(xj, k);      // L(k, lane), zero unless k > lane
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
  for (int i = tid; i < nb; i += ST) yp[i] = w[i];
}
// The nsl workgroups that complete the right-hand side of a BAD super-block each call this after their stores; the
// last arriver solves the block in place.  (All stores acknowledged, barrier, agent-scope release by one work-item,
// ticket; the winner acquires.)
template <bool FW>
__device__ __forceinline__ void bad_block_arrive(const double *Fs, int ld, int k0, int nb, double *yp, int *cnt, int nsl, double *w,
                                                 double *Sd, double *zp = nullptr, const double *dp = nullptr, bool also_bw = false) {
  __shared__ int last;
  SDM_STORES_DONE();
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    const int t = atomicAdd(cnt, 1);
    last = t == nsl - 1;
    if (last) atomicExch(cnt, 0);                                   // ready for the next solve
  }
  __syncthreads();
  if (!last) return;
  SDM_ACQUIRE_FENCE();
  if (FW) block_solve_fw(Fs, ld, k0, nb, yp, w, Sd); else block_solve_bw(Fs, ld, k0, nb, yp, w, Sd);
  if (FW && zp) {                                                    // the block is final: its ./d copy for the backward sweep
    __syncthreads();
    for (int i = threadIdx.x; i < nb; i += ST) { const double dk = dp[i]; zp[i] = yp[i] / (dk > 0.0 ? dk : 1.0); }
    if (also_bw) {                                                   // last block of a front without rows below: x = L_PP' \ z right away
      SDM_STORES_DONE();
      __syncthreads();
      block_solve_bw(Fs, ld, k0, nb, zp, w, Sd);
    }
  }
}

// ================================================================ forward sweep
// several right-hand sides side by side (blockIdx.z): element strides of the right-hand sides, of the result, of the
// update-vector scratch and of the fallback tickets (all 0 for a single right-hand side)
// fold_bw: the level's fronts have no rows below their own columns and the ./d copy is being written: a BAD last
// super-block is then also substituted backward right where its forward value becomes final (the backward sweep of
// such a level starts without k_sbw_init)
struct FwBatch { int64_t src, y, wv, cnt; int fold_bw; };
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
// ---- slab products.  Every kernel below issues its matrix loads FIRST (registers), then fetches the vector it
// multiplies with (written by the previous launch) into LDS, then multiplies: the two memory latencies overlap.
// Forward: sum_c M(r, cbase + c) xs[c] for the SROWS rows rbase .. of one slab.  Work-item (p = tid & 7, g = tid >> 3)
// owns row pair p and the columns g, g+32, ... (ncols <= SBW = 8 x 32); 16-byte loads, 8 in flight; fixed-order
// reduction over g.  rbase even; rows are clamped to the last valid pair (rlast = last valid row).
constexpr int NLD = SBW / 32;
__device__ __forceinline__ void slab_issue(sdm_double2 (&v)[NLD], const double *M, int64_t ldm, int cbase, int ncols, int rbase, int rlast) {
  const int tid = threadIdx.x, p = tid & 7, g = tid >> 3;
  const int r = min(rbase + 2 * p, rlast & ~1);
  const sdm_double2 *col = (const sdm_double2 *)(M + (int64_t)cbase * ldm + r);
  const int64_t ld2 = ldm >> 1;
#pragma unroll
  for (int j = 0; j < NLD; j++) v[j] = col[(int64_t)min(g + 32 * j, ncols - 1) * ld2];
}
// result for row rbase + t in work-items t < SROWS
__device__ __forceinline__ double slab_consume(const sdm_double2 (&v)[NLD], int ncols, const double *xs, double *red) {
  const int tid = threadIdx.x, p = tid & 7, g = tid >> 3;
  double a0 = 0.0, a1 = 0.0;
#pragma unroll
  for (int j = 0; j < NLD; j++) {
    const int c = g + 32 * j;
    const double xc = c < ncols ? xs[c] : 0.0;
    a0 += v[j].x * xc; a1 += v[j].y * xc;
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

// y_P = inv(L_PP) a_P for every super-block of every front of a level (bad blocks: copy, then substitution)
__global__ void __launch_bounds__(ST)
k_sfw_diag(const double *__restrict__ F, const double *__restrict__ S, FrontTab tab, const int *list, const double *wv, const double *src,
           const int *perm, double *y, const unsigned long long *sb_g, int *sb_cnt, double thr, int gather, FwBatch bt,
           double *zdiv, const double *dscale) {
  __shared__ double xs[SBW], red[(ST / 8) * SROWS], wsub[SBW];
  __shared__ double Sd[64 * TP];
  const int s = tab.one ? tab.o_s : list[blockIdx.y];
  const int ns = FT(ns), first = FT(first);
  const int r0 = SROWS * blockIdx.x;
  if (r0 >= ns) return;
  wv += (int64_t)blockIdx.z * bt.wv; y += (int64_t)blockIdx.z * bt.y; if (src) src += (int64_t)blockIdx.z * bt.src;
  sb_cnt += (int64_t)blockIdx.z * bt.cnt;
  const int Pb = r0 / SBW, c0 = Pb * SBW, nb = min(SBW, ns - c0);
  const int sb = FT(sboff) + Pb;
  const int tid = threadIdx.x;
  const int ncols = min(nb, r0 + SROWS - c0);                       // lower triangular: columns up to the slab's last row
  sdm_double2 v[NLD];
  slab_issue(v, S + FT(soff), FT(sld), c0, ncols, r0, ns - 1);
  const bool bad = sb_is_bad(sb_g, sb, thr);
  const double *a = wv + FT(woff);
  // the block's right-hand side: gathered through perm (leaves) or taken from the assembled vector
  for (int c = tid; c < nb; c += ST) xs[c] = gather ? src[perm[first + c0 + c]] : a[c0 + c];
  __syncthreads();
  if (bad) {
    if (tid < SROWS && r0 + tid < ns) y[first + r0 + tid] = xs[r0 - c0 + tid];
    // block 0 has nothing left of it: its right-hand side is complete here; later blocks are completed by step Pb-1
    if (Pb == 0) bad_block_arrive<true>(F + FT(foff), FT(ld), c0, nb, y + first + c0, sb_cnt + sb, (nb + SROWS - 1) / SROWS, wsub, Sd,
                                         zdiv ? zdiv + first + c0 : nullptr, zdiv ? dscale + first + c0 : nullptr,
                                         bt.fold_bw && ns <= SBW);
    return;
  }
  const double sum = slab_consume(v, ncols, xs, red);
  if (tid < SROWS && r0 + tid < ns) {
    y[first + r0 + tid] = sum;
    // block 0 is final here: its ./d copy (wrapPcg.m:57; skipped pivots act as 1, deninfac.m:89-94) for the backward sweep
    if (zdiv && Pb == 0) { const double dk = dscale[first + r0 + tid]; zdiv[first + r0 + tid] = sum / (dk > 0.0 ? dk : 1.0); }
  }
}

// step P: y_P is final; every row beyond super-block P receives  - M(r, P) y_P   (M = S for the front's own rows
// -- F for the rows of a bad super-block -- and F for the rows below the supernode, whose sums are the update
// vector passed to the parent)
__global__ void __launch_bounds__(ST)
k_sfw_step(const double *__restrict__ F, const double *__restrict__ S, FrontTab tab, const int *list, double *wv, double *y,
           const unsigned long long *sb_g, int *sb_cnt, double thr, int Pb, int first_assign, FwBatch bt, double *zdiv,
           const double *dscale) {
  __shared__ double xs[SBW], red[(ST / 8) * SROWS], wsub[SBW];
  __shared__ double Sd[64 * TP];
  const int s = tab.one ? tab.o_s : list[blockIdx.y];
  const int ns = FT(ns), ms = FT(ms), first = FT(first), ld = FT(ld);
  const int c0 = Pb * SBW;
  if (c0 >= ns) return;
  wv += (int64_t)blockIdx.z * bt.wv; y += (int64_t)blockIdx.z * bt.y; sb_cnt += (int64_t)blockIdx.z * bt.cnt;
  const int nb = min(SBW, ns - c0), ra = c0 + SBW;
  const int slabsA = ns > ra ? (ns - ra + SROWS - 1) / SROWS : 0;
  const int ebase = ns & ~1;
  const int slabsB = ms > ns ? (ms - ebase + SROWS - 1) / SROWS : 0;
  const int bx = blockIdx.x;
  if (bx >= slabsA + slabsB) return;
  const int tid = threadIdx.x;
  const double *Fs = F + FT(foff);
  const bool regA = bx < slabsA;
  const int r0 = regA ? ra + SROWS * bx : ebase + SROWS * (bx - slabsA);
  sdm_double2 v[NLD];
  // the front's own rows come from S (issued before the growth flag of their block is known: S is valid memory
  // either way); the rows below the supernode from the factor itself
  if (regA) slab_issue(v, S + FT(soff), FT(sld), c0, nb, r0, ns - 1);
  else slab_issue(v, Fs, ld, c0, nb, r0, ms - 1);
  const int Pr = r0 / SBW, sbr = FT(sboff) + Pr;
  const bool bad = regA && sb_is_bad(sb_g, sbr, thr);
  for (int c = tid; c < nb; c += ST) xs[c] = y[first + c0 + c];
  __syncthreads();
  if (bad) slab_issue(v, Fs, ld, c0, nb, r0, ns - 1);               // rare: the rows of a bad block were not premultiplied
  const double sum = slab_consume(v, nb, xs, red);
  if (regA) {
    if (tid < SROWS && r0 + tid < ns) {
      const double yv = y[first + r0 + tid] - sum;
      y[first + r0 + tid] = yv;
      // block Pb+1 receives its last contribution in this step: final (unless it still has to be solved by substitution)
      if (zdiv && Pr == Pb + 1 && !bad) { const double dk = dscale[first + r0 + tid]; zdiv[first + r0 + tid] = yv / (dk > 0.0 ? dk : 1.0); }
    }
    if (bad && Pr == Pb + 1) {                                      // this step completes the right-hand side of block Pr
      const int nbr = min(SBW, ns - Pr * SBW);
      bad_block_arrive<true>(Fs, ld, Pr * SBW, nbr, y + first + Pr * SBW, sb_cnt + sbr, (nbr + SROWS - 1) / SROWS, wsub, Sd,
                             zdiv ? zdiv + first + Pr * SBW : nullptr, zdiv ? dscale + first + Pr * SBW : nullptr,
                             bt.fold_bw && (Pr + 1) * SBW >= ns);
    }
  } else {
    double *u = wv + FT(woff);
    const int r = r0 + tid;
    if (tid < SROWS && r >= ns && r < ms) u[r] = first_assign ? -sum : u[r] - sum;
  }
}

// ================================================================ backward sweep
// sum_r M(rbase + r, c) xs[r] for the SROWS columns cbase .. of one slab over at most SBW rows: wavefront w owns 4
// columns, lanes run down the row pairs (contiguous 16-byte loads, 8 in flight), wave reduction in a fixed order.
// rbase even.  Result for column cbase + 4*w + q in every lane of wavefront w as out[q].
__device__ __forceinline__ void slabT_issue(sdm_double2 (&v)[8], const double *M, int64_t ldm, int cbase, int ncols, int rbase, int nrows) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int plast = max(((nrows + 1) >> 1) - 1, 0);
#pragma unroll
  for (int h = 0; h < 2; h++)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int c = min(cbase + 4 * wave + q, cbase + ncols - 1);
      v[4 * h + q] = ((const sdm_double2 *)(M + (int64_t)c * ldm + rbase))[min(lane + 64 * h, plast)];
    }
}
__device__ __forceinline__ void slabT_consume(const sdm_double2 (&v)[8], int nrows, const double *xs, double (&out)[4]) {
  const int lane = threadIdx.x & 63;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const int pi = lane + 64 * h;
    // rows beyond the range are dropped by selects on BOTH factors: a padding row of the front may hold anything
    const bool in0 = 2 * pi < nrows, in1 = 2 * pi + 1 < nrows;
    const double x0 = in0 ? xs[2 * pi] : 0.0, x1 = in1 ? xs[2 * pi + 1] : 0.0;
#pragma unroll
    for (int q = 0; q < 4; q++) acc[q] += (in0 ? v[4 * h + q].x : 0.0) * x0 + (in1 ? v[4 * h + q].y : 0.0) * x1;
  }
#pragma unroll
  for (int q = 0; q < 4; q++) {
    double a = acc[q];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    out[q] = a;
  }
}

// v = z ./ d  -  (rows below the supernode)' x_ancestors   for every column of every front of a level
__global__ void __launch_bounds__(ST)
k_sbw_init(const double *__restrict__ F, FrontTab tab, const int *list, double *y, const double *xfin, const double *dscale,
           const unsigned long long *sb_g, int *sb_cnt, double thr) {
  __shared__ double xs[SBW], wsub[SBW];
  __shared__ double Sd[64 * TP];
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
    for (int rb = ebase; rb < ms; rb += SBW) {                      // ancestors' entries, gathered SBW at a time
      const int nr = min(SBW, ms - rb);
      sdm_double2 v[8];
      slabT_issue(v, Fs, ld, c0, ncols, rb, nr);
      __syncthreads();
      for (int i = tid; i < nr; i += ST) xs[i] = rb + i >= ns ? xfin[rows[rb + i]] : 0.0;
      __syncthreads();
      double part[4];
      slabT_consume(v, nr, xs, part);
#pragma unroll
      for (int q = 0; q < 4; q++) tot[q] += part[q];
    }
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
  // the last super-block of a front has nothing above it in this sweep: if bad, it is solved once all its columns are set
  const int nsb = (ns + SBW - 1) / SBW, Pl = nsb - 1;
  if (c0 / SBW == Pl && sb_is_bad(sb_g, FT(sboff) + Pl, thr)) {
    const int nbl = ns - Pl * SBW;
    bad_block_arrive<false>(Fs, ld, Pl * SBW, nbl, y + first + Pl * SBW, sb_cnt + FT(sboff) + Pl, (nbl + SROWS - 1) / SROWS, wsub, Sd);
  }
}

// step Q: v_Q (x_Q for a bad block) is final; every column left of super-block Q receives  - M(Q rows, c)' v_Q
__global__ void __launch_bounds__(ST)
k_sbw_step(const double *__restrict__ F, const double *__restrict__ S, FrontTab tab, const int *list, double *y,
           const unsigned long long *sb_g, int *sb_cnt, double thr, int Q) {
  __shared__ double xs[SBW], wsub[SBW];
  __shared__ double Sd[64 * TP];
  const int s = tab.one ? tab.o_s : list[blockIdx.y];
  const int ns = FT(ns), first = FT(first), ld = FT(ld);
  const int rb = Q * SBW;
  if (rb >= ns) return;
  const int nbq = min(SBW, ns - rb);
  const int c0 = SROWS * blockIdx.x;                                // < rb by the grid
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const double *Fs = F + FT(foff);
  sdm_double2 v[8];
  slabT_issue(v, S + FT(soff), FT(sld), c0, SROWS, rb, nbq);  // before the growth flag is known (S is valid memory either way)
  const bool badq = sb_is_bad(sb_g, FT(sboff) + Q, thr);
  for (int i = tid; i < nbq; i += ST) xs[i] = y[first + rb + i];
  __syncthreads();
  if (badq) slabT_issue(v, Fs, ld, c0, SROWS, rb, nbq);             // rare: the rows of a bad block were not premultiplied
  double part[4];
  slabT_consume(v, nbq, xs, part);
  if (lane == 0) {
#pragma unroll
    for (int q = 0; q < 4; q++) y[first + c0 + 4 * wave + q] -= part[q];
  }
  const int Pc = c0 / SBW;
  if (Pc == Q - 1 && sb_is_bad(sb_g, FT(sboff) + Pc, thr))      // this step completes v of block Q-1
    bad_block_arrive<false>(Fs, ld, Pc * SBW, SBW, y + first + Pc * SBW, sb_cnt + FT(sboff) + Pc, SBW / SROWS, wsub, Sd);
}

// x_P = inv(L_PP)' v_P ; the result goes to xfin (descendants read it) and, scattered through perm, to yout
__global__ void __launch_bounds__(ST)
k_sbw_diag(const double *__restrict__ S, FrontTab tab, const int *list, const double *y, double *xfin, double *yout, const int *perm,
           const unsigned long long *sb_g, double thr) {
  __shared__ double xs[SBW];
  const int s = tab.one ? tab.o_s : list[blockIdx.y];
  const int ns = FT(ns), first = FT(first);
  const int c0 = SROWS * blockIdx.x;
  if (c0 >= ns) return;
  const int Pb = c0 / SBW, rb = Pb * SBW, nb = min(SBW, ns - rb);
  const int ncols = min(SROWS, ns - c0);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int nr = rb + nb - c0;                                       // rows c0 .. end of the block (upper part of S is zero)
  sdm_double2 v[8];
  slabT_issue(v, S + FT(soff), FT(sld), c0, ncols, c0, nr);
  const bool bad = sb_is_bad(sb_g, FT(sboff) + Pb, thr);
  for (int i = tid; i < nr; i += ST) xs[i] = y[first + c0 + i];
  __syncthreads();
  double part[4];
  if (bad) {
#pragma unroll
    for (int q = 0; q < 4; q++) { const int c = c0 + 4 * wave + q; part[q] = c < ns ? xs[c - c0] : 0.0; }
  } else {
    slabT_consume(v, nr, xs, part);
  }
  if (lane == 0) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int c = c0 + 4 * wave + q;
      if (c < ns) { xfin[first + c] = part[q]; if (yout) yout[perm[first + c]] = part[q]; }
    }
  }
}

// ================================================================ the whole solve of a one-front factor in ONE launch
// fw, ./d, bw (wrapPcg.m:56-59 without dense columns) for factors that are a single front without rows below (the dense
// Schur complements of SDPs: control07, MAXCUT).  The launch-per-step sweeps above are chains of 2 (nsb - 1) + 2 dependent
// launches; here workgroup b OWNS rows 16 b .. of the forward sweep and the same columns of the backward sweep and
// accumulates their value in registers, in the order of the step launches (same slab functions: same bits):
//   forward   diagonal-block product, then for Q = 0 .. P-1:  - M(rows, Q) y_Q  as soon as block Q is final (doneF[Q] =
//             slabs of block Q final; the matrix slab is in flight before the wait); y and z = y ./ d stored write-through,
//             counted in doneF[P];
//   backward  v = z, then for Q = nsb-1 .. P+1:  - M(Q, columns)' v_Q  as soon as block Q is final (doneB[Q]); v stored (vb, an
//             array of its own), counted in doneB[P]; once the whole block is there: x = inv(L_PP)' v  ->  xfin, yout(perm).
// Blocks on the substitution fallback: rows / columns stay unpremultiplied (F instead of S), the last slab to arrive
// solves the block (block_solve_fw / _bw) and counts for all of them.  Forward waits are for lower workgroups, backward
// waits for higher ones: all workgroups must be resident (SFRONT_MAX_WGS; PersistTurn on the host).  The last workgroup
// to leave re-arms the counters.  The emulator (workgroups one after the other) runs phase 1 = forward, phase 2 = the
// backward accumulation with the workgroups in reverse order, phase 3 = the diagonal-block products; the GPU runs
// phase 0 = all of it.
__device__ __forceinline__ bool arrive_is_last(int *cnt, int nsl) {
  __shared__ int last;
  SDM_STORES_DONE();
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    const int t = atomicAdd(cnt, 1);
    last = t == nsl - 1;
    if (last) atomicExch(cnt, 0);                                     // ready for the next solve
  }
  __syncthreads();
  return last != 0;
}
__global__ void __launch_bounds__(ST)
k_solve_front(const double *__restrict__ F, const double *__restrict__ S, FrontTab tab, const double *src, const int *perm, double *y,
              double *zdiv, double *vb, const double *dscale, double *xfin, double *yout, const unsigned long long *sb_g, int *sb_cnt, int *done,
              double thr, int phase, int *tmo) {
  __shared__ double xs[SBW], red[(ST / 8) * SROWS], wsub[SBW];
  __shared__ double Sd[64 * TP];
  const int ns = tab.o_ns, first = tab.o_first, ld = tab.o_ld, sld = tab.o_sld;
  const int nsb = (ns + SBW - 1) / SBW;
  const int nwg = gridDim.x;
#ifdef SDM_EMU
  const int bx = phase == 2 ? nwg - 1 - (int)blockIdx.x : (int)blockIdx.x;
#else
  const int bx = blockIdx.x;
#endif
  const int r0 = SROWS * bx;                                          // rows (forward) = columns (backward) of this workgroup
  const int Pb = r0 / SBW, c0 = Pb * SBW, nb = min(SBW, ns - c0);
  const int sb = tab.o_sboff + Pb;
  const int nsl = (nb + SROWS - 1) / SROWS;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const double *Fs = F + tab.o_foff, *Ss = S + tab.o_soff;
  int *doneF = done, *doneB = done + nsb, *fin = done + 2 * nsb;
  const bool bad = sb_is_bad(sb_g, sb, thr);
  if (phase <= 1) {
    // ---------------------------------------------------------------- forward
    sdm_double2 v[NLD];
    const int ncols = min(nb, r0 + SROWS - c0);                       // lower triangular: columns up to the slab's last row
    slab_issue(v, Ss, sld, c0, ncols, r0, ns - 1);
    for (int c = tid; c < nb; c += ST) xs[c] = src[perm[first + c0 + c]];
    __syncthreads();
    double acc;
    if (bad) acc = (tid < SROWS && r0 + tid < ns) ? xs[r0 - c0 + tid] : 0.0;     // the right-hand side itself: solved by substitution below
    else acc = slab_consume(v, ncols, xs, red);
    for (int Q = 0; Q < Pb; Q++) {
      const int cq = Q * SBW;
      if (bad) slab_issue(v, Fs, ld, cq, SBW, r0, ns - 1);            // the rows of a bad block were not premultiplied
      else slab_issue(v, Ss, sld, cq, SBW, r0, ns - 1);
      prep_wait(doneF + Q, SBW / SROWS, tmo);                         // y_Q is final (and everybody is done with xs / red)
      for (int c = tid; c < SBW; c += ST) xs[c] = sdm_load_wt(&y[first + cq + c]);
      __syncthreads();
      acc -= slab_consume(v, SBW, xs, red);
    }
    if (tid < SROWS && r0 + tid < ns) {
      sdm_store_wt(&y[first + r0 + tid], acc);
      if (!bad) { const double dk = dscale[first + r0 + tid]; sdm_store_wt(&zdiv[first + r0 + tid], acc / (dk > 0.0 ? dk : 1.0)); }
    }
    if (!bad) {
      prep_done(doneF + Pb);
    } else if (arrive_is_last(sb_cnt + sb, nsl)) {
      SDM_ACQUIRE_FENCE();
      block_solve_fw(Fs, ld, c0, nb, y + first + c0, wsub, Sd);
      __syncthreads();
      for (int i = tid; i < nb; i += ST) { const double dk = dscale[first + c0 + i]; zdiv[first + c0 + i] = y[first + c0 + i] / (dk > 0.0 ? dk : 1.0); }
      if (Pb == nsb - 1) {                                            // nothing above the last block: x = L_PP' \ z right away
        SDM_STORES_DONE();
        __syncthreads();
        block_solve_bw(Fs, ld, c0, nb, zdiv + first + c0, wsub, Sd);
      }
      SDM_STORES_DONE();
      __syncthreads();
      if (tid == 0) { __threadfence(); sdm_signal_add(doneF + Pb, nsl); }
    }
    if (phase == 1) return;
  }
  // ------------------------------------------------------------------ backward (z from zdiv, v in vb)
  if (phase == 0 || phase == 2) {
  prep_wait(doneF + Pb, nsl, tmo);                                    // this block's z (x for a bad last block) is complete
  double val[4];
#pragma unroll
  for (int q = 0; q < 4; q++) { const int c = r0 + 4 * wave + q; val[q] = c < ns ? sdm_load_wt(&zdiv[first + c]) : 0.0; }
  for (int Q = nsb - 1; Q > Pb; Q--) {
    const int rb = Q * SBW, nbq = min(SBW, ns - rb);
    const bool badq = sb_is_bad(sb_g, tab.o_sboff + Q, thr);
    sdm_double2 vt[8];
    if (badq) slabT_issue(vt, Fs, ld, r0, SROWS, rb, nbq); else slabT_issue(vt, Ss, sld, r0, SROWS, rb, nbq);
    prep_wait(doneB + Q, (nbq + SROWS - 1) / SROWS, tmo);             // v_Q (x_Q of a bad block) is final
    for (int i = tid; i < nbq; i += ST) xs[i] = sdm_load_wt(&vb[first + rb + i]);
    __syncthreads();
    double part[4];
    slabT_consume(vt, nbq, xs, part);
#pragma unroll
    for (int q = 0; q < 4; q++) val[q] -= part[q];
  }
  // v goes to an array of its own: every line another workgroup reads after a wait is one it has not read before in this
  // launch (z and v in the same place would be read twice through the same L2)
  if (lane == 0) {
#pragma unroll
    for (int q = 0; q < 4; q++) { const int c = r0 + 4 * wave + q; if (c < ns) sdm_store_wt(&vb[first + c], val[q]); }
  }
  if (!bad || Pb == nsb - 1) {                                        // (a bad LAST block was substituted by the forward part: val is x)
    prep_done(doneB + Pb);
  } else if (arrive_is_last(sb_cnt + sb, nsl)) {
    SDM_ACQUIRE_FENCE();
    block_solve_bw(Fs, ld, c0, nb, vb + first + c0, wsub, Sd);
    SDM_STORES_DONE();
    __syncthreads();
    if (tid == 0) { __threadfence(); sdm_signal_add(doneB + Pb, nsl); }
  }
  if (phase == 2) return;
  }
  {
    const int nr = c0 + nb - r0, ncols = min(SROWS, ns - r0);         // rows r0 .. end of the block (upper part of S is zero)
    sdm_double2 vt[8];
    slabT_issue(vt, Ss, sld, r0, ncols, r0, nr);
    prep_wait(doneB + Pb, nsl, tmo);
    for (int i = tid; i < nr; i += ST) xs[i] = sdm_load_wt(&vb[first + r0 + i]);
    __syncthreads();
    double part[4];
    if (bad) {
#pragma unroll
      for (int q = 0; q < 4; q++) { const int c = r0 + 4 * wave + q; part[q] = c < ns ? xs[c - r0] : 0.0; }
    } else {
      slabT_consume(vt, nr, xs, part);
    }
    if (lane == 0) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int c = r0 + 4 * wave + q;
        if (c < ns) { xfin[first + c] = part[q]; if (yout) yout[perm[first + c]] = part[q]; }
      }
    }
  }
  __syncthreads();
  if (tid == 0) {                                                      // the last one out re-arms the counters
    const int t = atomicAdd(fin, 1);
    if (t == nwg - 1) { for (int i = 0; i < 2 * nsb; i++) atomicExch(done + i, 0); atomicExch(fin, 0); }
  }
}

// ================================================================ host drivers
const double *solve_d(sdm_plan *P) { return P->dense.factored ? (const double *)P->chol.dsolve.p : (const double *)P->chol.d.p; }

#undef FT
// the level's table for the solve kernels: with the descriptor of its only front filled in when there is just one
static FrontTab level_tab(const CholPlan &C, FrontTab t, int l) {
  if (C.levptr[l + 1] - C.levptr[l] != 1) return t;
  const int s = C.levlist[C.levptr[l]];
  t.one = 1; t.o_s = s; t.o_ns = C.sn_ns[s]; t.o_ms = C.sn_ms[s]; t.o_ld = C.sn_ld[s]; t.o_first = C.sn_first[s];
  t.o_sld = C.sn_sld[s]; t.o_sboff = C.sn_sboff[s]; t.o_foff = C.sn_foff[s]; t.o_soff = C.sn_soff[s]; t.o_woff = C.sn_woff[s];
  t.o_xl = C.sn_xl[s];
  return t;
}
void solve_prepare(sdm_plan *P, bool sb_g_is_zero) {
  CholPlan &C = P->chol;
  FrontTab tab = front_tab(C);
#ifndef SDM_EMU
  static bool attr = false;
  if (!attr) {
    SDM_HIP_CHECK(hipFuncSetAttribute((const void *)k_sinv128, hipFuncAttributeMaxDynamicSharedMemorySize, (int)INV_LDS));
    SDM_HIP_CHECK(hipFuncSetAttribute((const void *)k_stile, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TILE_LDS));
    SDM_HIP_CHECK(hipFuncSetAttribute((const void *)k_sprep, hipFuncAttributeMaxDynamicSharedMemorySize, (int)INV_LDS));
    attr = true;
  }
#endif
  C.growth_used = C.growth_max;                                     // the solves decide with the bound the premultiplication saw
  if (!sb_g_is_zero)                                                // (a factorisation zeroes them in k_prep_pivots)
    SDM_HIP_CHECK(hipMemsetAsync(C.sb_g.p, 0, (size_t)std::max(4 * C.nsbtot, 4) * sizeof(unsigned long long), P->stream));
  const int nitems = C.n_i128 + 2 * C.n_t3 + C.n_pm;
  static const bool fused_off = getenv("SDM_SPREP_OFF") != nullptr;     // tuning override (tools only)
  if (nitems <= SPREP_MAX_ITEMS && C.n_i128 > 0 && !fused_off) {                   // everything resident at once: one launch, counters instead of boundaries
    SDM_KLAUNCH(P, k_sprep, dim3(nitems), dim3(ST), INV_LDS, C.fronts.p, C.S.p, C.ttmp.p, tab, C.l_i128.p, C.n_i128, C.l_t3.p, C.n_t3,
                C.l_pm.p, C.n_pm, C.sb_g.p, (int *)(C.sb_g.p + 2 * C.nsbtot), C.growth_used, C.tmo.dev());
    return;
  }
  if (C.n_i128) SDM_KLAUNCH(P, k_sinv128, dim3(C.n_i128), dim3(ST), INV_LDS, C.fronts.p, C.S.p, tab, C.l_i128.p, C.sb_g.p);
  if (C.n_t3) {
    SDM_KLAUNCH(P, k_stile, dim3(C.n_t3), dim3(ST), TILE_LDS, C.fronts.p, C.S.p, C.ttmp.p, tab, C.l_t3.p, C.sb_g.p, 0, C.growth_used);
    SDM_KLAUNCH(P, k_stile, dim3(C.n_t3), dim3(ST), TILE_LDS, C.fronts.p, C.S.p, C.ttmp.p, tab, C.l_t3.p, C.sb_g.p, 1, C.growth_used);
  }
  if (C.n_pm) SDM_KLAUNCH(P, k_stile, dim3(C.n_pm), dim3(ST), TILE_LDS, C.fronts.p, C.S.p, C.ttmp.p, tab, C.l_pm.p, C.sb_g.p, 2, C.growth_used);
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

// forward sweeps of nrhs right-hand sides side by side (grid.z): rhs + z*rhs_stride -> y + z*y_stride (permuted order);
// wv = update-vector scratch of wsize doubles per right-hand side
void solve_fw_batch(sdm_plan *P, const double *rhs, int64_t rhs_stride, double *y, int64_t y_stride, double *wv, int nrhs,
                    double *zdiv, const double *dscale) {
  CholPlan &C = P->chol;
  FrontTab tab = front_tab(C);
  const double thr = C.growth_used;
  if ((size_t)C.nsbtot * (size_t)nrhs > C.sb_cnt.n) throw std::runtime_error("solve_fw_batch: ticket array too small for this many right-hand sides");
  FwBatch bt;
  bt.src = nrhs > 1 ? rhs_stride : 0; bt.y = nrhs > 1 ? y_stride : 0; bt.wv = nrhs > 1 ? C.wsize : 0; bt.cnt = nrhs > 1 ? C.nsbtot : 0;
  const FrontTab tab0 = tab;
  for (int l = 0; l < C.nlevels; l++) {
    const SolveLevel &L = C.slev[l];
    bt.fold_bw = (zdiv && !L.below) ? 1 : 0;
    const int *list = C.d_levlist.p + C.levptr[l];
    tab = level_tab(C, tab0, l);
    const int gather = L.children ? 0 : 1;
    if (!gather) SDM_KLAUNCH(P, k_sfw_init, dim3(L.nfronts, 1, nrhs), dim3(ST), 0, tab, list, wv, rhs, C.d_perm.p, y, bt);
    SDM_KLAUNCH(P, k_sfw_diag, dim3((L.maxns + SROWS - 1) / SROWS, L.nfronts, nrhs), dim3(ST), 0, C.fronts.p, C.S.p, tab, list, wv, rhs,
                C.d_perm.p, y, C.sb_g.p, C.sb_cnt.p, thr, gather, bt, zdiv, dscale);
    for (int Pb = 0; Pb < (int)L.maxslab_fw.size(); Pb++)
      if (L.maxslab_fw[Pb] > 0)
        SDM_KLAUNCH(P, k_sfw_step, dim3(L.maxslab_fw[Pb], L.nfronts, nrhs), dim3(ST), 0, C.fronts.p, C.S.p, tab, list, wv, y, C.sb_g.p,
                    C.sb_cnt.p, thr, Pb, (gather && Pb == 0) ? 1 : 0, bt, zdiv, dscale);
  }
}

// backward sweep in place on the vector y (permuted order).  dscale != null: ./d on the way in (k_sbw_init);
// skip_plain_init: the levels whose fronts have no rows below their own columns need no k_sbw_init at all (the ./d
// was already applied by the forward sweep's final writes)
static void solve_bw_inplace(sdm_plan *P, double *y, double *yout, const double *dscale, bool skip_plain_init) {
  CholPlan &C = P->chol;
  FrontTab tab = front_tab(C);
  const double thr = C.growth_used;
  const FrontTab tab0 = tab;
  for (int l = C.nlevels - 1; l >= 0; l--) {
    const SolveLevel &L = C.slev[l];
    const int *list = C.d_levlist.p + C.levptr[l];
    tab = level_tab(C, tab0, l);
    const int ncs = (L.maxns + SROWS - 1) / SROWS;
    // (a bad last super-block of such a level was substituted backward by the forward sweep: FwBatch::fold_bw)
    if (!(skip_plain_init && !L.below))
      SDM_KLAUNCH(P, k_sbw_init, dim3(ncs, L.nfronts), dim3(ST), 0, C.fronts.p, tab, list, y, C.xfin.p, dscale, C.sb_g.p, C.sb_cnt.p, thr);
    for (int Q = L.nsb - 1; Q >= 1; Q--)
      SDM_KLAUNCH(P, k_sbw_step, dim3(Q * (SBW / SROWS), L.nfronts), dim3(ST), 0, C.fronts.p, C.S.p, tab, list, y, C.sb_g.p, C.sb_cnt.p, thr, Q);
    SDM_KLAUNCH(P, k_sbw_diag, dim3(ncs, L.nfronts), dim3(ST), 0, C.S.p, tab, list, y, C.xfin.p, yout, C.d_perm.p, C.sb_g.p, thr);
  }
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
  if (fold && C.solve_fused) {
    const FrontTab tab = level_tab(C, front_tab(C), 0);
    const int nwg = (C.sn_ns[0] + SROWS - 1) / SROWS;
    PersistTurn turn(P);
#ifdef SDM_EMU
    for (int phase = 1; phase <= 3; phase++)
#else
    const int phase = 0;
#endif
      SDM_KLAUNCH(P, k_solve_front, dim3(nwg), dim3(ST), 0, C.fronts.p, C.S.p, tab, rhs, C.d_perm.p, y, C.zdiv.p, C.wvec.p, solve_d(P), C.xfin.p, yout,
                  C.sb_g.p, C.sb_cnt.p, C.sfront_cnt.p, C.growth_used, phase, C.tmo.dev());
    return;
  }
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

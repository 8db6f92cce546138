// sdm_follow.h -- device code shared by sdm_solve.hip (the inverses of the diagonal super-blocks) and sdm_chol.hip (whose one-launch
// factor kernel carries the workgroups that build a front's inverse BEHIND its factorisation): matrix-core tile helpers, the 64x64
// inversion, and the follower's body.  Device functions only; the kernels stay in their sources.
#pragma once
#include "sdm_plan.h"

namespace sdm {

constexpr int ST = 256;          // work-items per workgroup of every kernel of sdm_solve.hip
constexpr int TP = 65;           // LDS pitch of staged 64-wide operand blocks (conflict-free transposing stores)
constexpr size_t INV_LDS = (size_t)4 * 64 * TP * sizeof(double);      // k_sinv128: four staged 64x64 blocks
constexpr size_t TILE_LDS = (size_t)2 * 64 * TP * sizeof(double);     // k_stile / the follower: one A and one B operand block

// ================================================================ device helpers
__device__ __forceinline__ double bits_to_double(unsigned long long u) { union { unsigned long long u; double d; } b; b.u = u; return b.d; }
__device__ __forceinline__ unsigned long long double_to_bits(double d) { union { unsigned long long u; double d; } b; b.d = d; return b.u; }
// growth check of super-block sb: max|inv| * max|L| within bounds (NaN counts as bad)
__device__ __forceinline__ bool sb_is_bad(const unsigned long long *g, int sb, double thr) {
  return !(bits_to_double(g[2 * sb]) * bits_to_double(g[2 * sb + 1]) <= thr);
}
// the three kinds of super-block: 0 within the bound (applied as its explicit inverse), 1 beyond it but no further than `thr2`
// (inverse + iterative refinement against the factor when the solve runs its refinement launches, substitution otherwise),
// 2 beyond thr2 or not a number (always substituted)
__device__ __forceinline__ int sb_class(const unsigned long long *g, int sb, double thr, double thr2) {
  const double gr = bits_to_double(g[2 * sb]) * bits_to_double(g[2 * sb + 1]);
  return gr <= thr ? 0 : (gr <= thr2 ? 1 : 2);
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

// the same tile into the TRANSPOSED arena: dst[r*ld + c] = Cs[r*TP + c] (rows < nr, columns < nc), consecutive work-items on
// consecutive columns
template <bool WT = false>
__device__ __forceinline__ void store_tile_T(double *dst, int64_t ld, const double *Cs, int nr, int nc, int tid) {
  const int c = tid & 63, rq = tid >> 6;
#pragma unroll 4
  for (int r = rq; r < 64; r += ST / 64)
    if (r < nr && c < nc) {
      const double v = Cs[r * TP + c];
      if (WT) sdm_store_wt(&dst[(int64_t)r * ld + c], v); else dst[(int64_t)r * ld + c] = v;
    }
}

// ================================================================ inversion of the diagonal super-blocks
// The two 64x64 unit lower triangular blocks A and C of a leaf, all four wavefronts: rawA / rawC hold their strictly lower
// triangles column-major (raw[k*TP + i] = L(i, k)), bufA / bufC start as zero and receive the inverses -- inv(A) as a B
// operand (bufA[k*TP + col] = inv(k, col)), inv(C) as an A operand (bufC[k*TP + row] = inv(row, k)).  One barrier inside;
// the caller synchronises before it reads the results.  max |inverse| goes to gP.
// ONLYA: there is no block C (rawC / bufC are not touched; wavefronts 2 and 3 only keep the barriers company).
template <bool ONLYA = false>
__device__ __forceinline__ void inv64_pair(double *rawA, double *rawC, double *bufA, double *bufC, int wave, int lane, unsigned long long *gP) {
  const int blk = wave >> 1, q = wave & 1;                          // wavefront -> (64-block A / C, 32-block inside it)
  const bool work = !(ONLYA && blk == 1);
  const double *raw = blk == 0 ? rawA : rawC;
  double *dst = blk == 0 ? bufA : bufC;
  if (work) {
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
  __syncthreads();
  // ---- 64x64: X10 = -inv11 (L10 inv00) per 64-block on the FP64 matrix cores: two wavefronts per block, wavefront q owns
  // the two 16x16 tiles of output columns 16q .. 16q+15 of each product, K = 32 = 8 steps of v_mfma_f64_16x16x4_f64.
  // T goes to the unused upper right quadrant of the raw buffer (rows 32.., columns < 32 of raw[k*TP + i] hold zeros).
  const int li = lane & 15, lk = lane >> 4;
  double *Ts = (blk == 0 ? rawA : rawC) + 32 * TP;                   // Ts[r*TP + c]
  sdm_double4 acc[2];
  if (work) {
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
  }
  __syncthreads();
  if (work) {
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
}

__device__ __forceinline__ void prep_wait(const int *cnt, int target, int *tmo) {
  if (threadIdx.x == 0) {
    for (long it = 0; sdm_signal_load(cnt) < target; it++) { if (sdm_spin_giveup(it, tmo)) break; SDM_SPIN_PAUSE(); }
  }
  __syncthreads();                                                  // no acquire fence: everything waited for is read with sc1 loads
}
__device__ __forceinline__ void prep_done(int *cnt) {
  SDM_STORES_DONE();
  __syncthreads();
  if (threadIdx.x == 0) sdm_signal_add(cnt);
}
// ---- the inverse of a whole front BEHIND its factorisation (k_sinv_follow): fronts factored by ONE k_ldl_front launch
// (sdm_chol.hip) whose columns are one super-block.  Run by the LAST workgroups of that launch (until round 4: by a launch of its
// own on a second stream -- fork and join cost 18 us per factorisation), it follows the factor's progress counters and builds
// X = inv(L) by block rows of 64:
//   X(r, r) = inv(L(r, r))                                  as soon as panel r's diagonal block is published (diag_cnt),
//   X(r, c) = - X(r, r) sum_{q = c}^{r-1} L(r, q) X(q, c)   the sum as the rows L(r, q) are published (row_cnt) and the
//                                                            tiles X(q, c) arrive (xcnt), the product once X(r, r) is there,
// one workgroup per 64x64 tile (diagonal tiles first, then the tiles row by row: a workgroup only waits for lower ones --
// or for the factorisation).  The factor chain takes ~20 us per panel; a tile needs one 64^3 product per panel, so the
// inverse is complete a few microseconds after the factor instead of a chain of six dependent tile stages later (146 us
// for control07's 666 columns, r03a).  What it reads was stored write-through by k_ldl_front (DT, d, the rows of L) and
// is read with sc1 loads; its own tiles likewise.  The emulator launches it after the factorisation.
// xcnt[r * FRONT_MAXT + c] = 1 once X(r, c) is in S (c = r: the diagonal tile); zeroed with front_cnt by k_prep_pivots.
__device__ __forceinline__ void follow_wait(const int *cnt, int target, int *tmo) {
  if (threadIdx.x == 0) {
    for (long it = 0; sdm_signal_load(cnt) < target; it++) { if (sdm_spin_giveup(it, tmo)) break; SDM_SPIN_PAUSE(); }
  }
  __syncthreads();
}
// the front a follower workgroup works on: scalars only, so that a called stage can take them in registers (sdm_chol.hip)
struct FollowDesc { int ns, ld, sld, slot, sboff, s; int64_t foff, soff, toff; };
__device__ __forceinline__ FollowDesc follow_desc(const FrontTab &tab, const int *list, int by) {
  FollowDesc d;
  d.s = list[by];
  d.ns = tab.ns[d.s]; d.ld = tab.ld[d.s]; d.sld = tab.sld[d.s]; d.slot = tab.fslot[d.s]; d.sboff = tab.sboff[d.s];
  d.foff = tab.foff[d.s]; d.soff = tab.soff[d.s]; d.toff = tab.toff[d.s];
  return d;
}
// the work of workgroup bx of a front's follower grid (ST work-items; smem: TILE_LDS bytes of LDS): called by k_sinv_follow and, on
// the device, by the extra workgroups of the k_ldl_front launch itself (sdm_chol.hip)
__device__ __forceinline__ void sinv_follow_body(char *smem, int bx, const FollowDesc &fd, const double *F, const double *DT, double *S, double *STr,
                                                 int *front_cnt, const int *diag_cnt, unsigned long long *sb_g, int *tmo) {
  const int s = fd.s;
  const int ns = fd.ns, ld = fd.ld, sld = fd.sld;
  const int T = (ns + 63) / 64;
  const int b = bx;
  if (b >= T * (T + 1) / 2) return;
  const double *Fs = F + fd.foff;
  double *Ss = S + fd.soff, *Ts = STr + fd.soff;
  const int slot = fd.slot;
  const int *row_cnt = front_cnt + (int64_t)slot * FRONT_CNT;
  int *xcnt = front_cnt + (int64_t)slot * FRONT_CNT + FRONT_XCNT_OFF;
  unsigned long long *gP = sb_g + 2 * fd.sboff;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (b < T) {
    // ---- diagonal tile r: inv(L(r, r)) from the transposed copy DT of the factored block
    const int r = b, nb = min(64, ns - 64 * r);
    double *bufA = (double *)smem, *rawA = bufA + 64 * TP;          // (TILE_LDS: the two arrays of the tile role)
    const double *Ds = DT + fd.toff + (int64_t)r * NB * NB;      // Ds[i*NB + k] = L(64r + i, 64r + k)
    follow_wait(diag_cnt + s, 4 * (r + 1), tmo);
    double lmx = 0.0;
    {
      double va[SPT];
      const int i = tid & 63, kq = tid >> 6;
#pragma unroll
      for (int j = 0; j < SPT; j++) va[j] = sdm_load_wt(&Ds[min(i, nb - 1) * NB + min(kq + (ST / 64) * j, nb - 1)]);
#pragma unroll
      for (int j = 0; j < SPT; j++) {
        const int k = kq + (ST / 64) * j;
        const double a = (i > k && i < nb) ? va[j] : 0.0;
        rawA[k * TP + i] = a;
        bufA[k * TP + i] = 0.0;
        lmx = fmax(lmx, fabs(a));
      }
    }
    __syncthreads();
    inv64_pair<true>(rawA, nullptr, bufA, nullptr, wave, lane, gP);
    __syncthreads();
    wave_atomic_max(gP + 1, lmx, lane);
    for (int e = tid; e < 64 * 64; e += ST) {
      const int i = e & 63, j = e >> 6;
      if (i >= j && i < nb) sdm_store_wt(&Ss[(int64_t)(64 * r + j) * sld + 64 * r + i], bufA[i * TP + j]);
    }
    for (int e = tid; e < 64 * 64; e += ST) {
      const int j = e & 63, i = e >> 6;
      if (i >= j && i < nb) Ts[(int64_t)(64 * r + i) * sld + 64 * r + j] = bufA[i * TP + j];       // (the transposed copy is read by the solves only: plain stores)
    }
    prep_done(xcnt + r * FRONT_MAXT + r);
    return;
  }
  // ---- tile (r, c), c < r
  int r = 1, c = b - T;
  while (c >= r) { c -= r; r++; }
  double *As = (double *)smem, *Bs = As + 64 * TP;
  const int arows = min(64, ns - 64 * r);
  Acc22 acc;
  acc_zero(acc);
  double lmx = 0.0;
  double va[SPT], vb[SPT];
  for (int q = c; q < r; q++) {
    follow_wait(row_cnt + r, q + 1, tmo);                            // L(r, q) is in the front
    follow_wait(xcnt + q * FRONT_MAXT + c, 1, tmo);                  // X(q, c) is in S
    stage_colmajor_load<true>(va, Fs + (int64_t)(64 * q) * ld + 64 * r, ld, arows, 64, tid);
    stage_transposed_load<true>(vb, Ss + (int64_t)(64 * c) * sld + 64 * q, sld, 64, 64, tid);
    lmx = fmax(lmx, stage_colmajor_store(As, va, arows, 64, tid));
    stage_transposed_store(Bs, vb, 64, 64, tid);
    __syncthreads();
    mma_block(acc, As, Bs, wave, lane);
    __syncthreads();
  }
  wave_atomic_max(gP + 1, lmx, lane);
  follow_wait(xcnt + r * FRONT_MAXT + r, 1, tmo);                    // X(r, r)
  stage_colmajor_load<true>(va, Ss + (int64_t)(64 * r) * sld + 64 * r, sld, arows, arows, tid);
  acc_to_lds_rowmajor(acc, Bs, wave, lane, 1.0);                     // the sum as a B operand: Bs[k*TP + col]
  stage_colmajor_store(As, va, arows, arows, tid);
  __syncthreads();
  acc_zero(acc);
  mma_block(acc, As, Bs, wave, lane);
  __syncthreads();
  acc_to_lds_rowmajor(acc, As, wave, lane, -1.0);
  __syncthreads();
  const double gm = store_tile<true>(Ss + (int64_t)(64 * c) * sld + 64 * r, sld, As, arows, 64, tid);
  store_tile_T<false>(Ts + (int64_t)(64 * r) * sld + 64 * c, sld, As, arows, 64, tid);
  wave_atomic_max(gP, gm, lane);
  prep_done(xcnt + r * FRONT_MAXT + c);
}

}  // namespace sdm

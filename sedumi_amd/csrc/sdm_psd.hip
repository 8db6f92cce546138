// sdm_psd.hip -- per-iteration PSD-block helpers next to the ADA' path:
//   y = invcholfac(u, K, perm)        invcholfac.c:59-168, triuaux.c:175-221 (utmulx / prpiutmulx), :61-70 (invmatperm)
// For every PSD block k of order n:  Y_k(perm,perm) = U_k' U_k  with U_k = triu(u_k) (the strict lower triangle of the
// stored n x n array is ignored, like utmulx only reads u(0:i, i)); Hermitian blocks ([Re; Im] planes) form U^H U
// with Im diag(U) taken as 0 (prpiutmulx: "assumes that diag(imag(U)) == 0").  This is the `udsqr` argument of getada3
// (sedumi.m:452): forming it on the device keeps it resident for sdm_plan_getada.
//
// One 64x64 tile of Z = U'U per workgroup (4 wavefronts, 32x32 quadrants of v_mfma_f64_16x16x4_f64 tiles), K-loop over
// the rows t <= min(i,j) in chunks of 64 staged in LDS with the triangular mask applied on the way in; only tiles
// I >= J are computed, both (i,j) and (j,i) are written through perm (the product is symmetric / Hermitian by
// construction, which is what triu2sym / triu2herm restore in the reference).
#include "sdm_plan.h"
#include "sdm_rt.h"

namespace sdm {

struct PsdBlocks {           // device arrays, one entry per PSD block
  const int *n;              // order
  const int64_t *off;        // offset of the block in u / y (doubles)
  const int *poff;           // offset into perm
  int rsdpN;                 // blocks >= rsdpN are Hermitian
};

template <bool HERM>
__device__ __forceinline__ void invchol_tile(const double *u, double *y, const int *perm, int n, int I, int J,
                                             double (*As)[TILE], double (*Bs)[TILE], double (*Ai)[TILE], double (*Bi)[TILE]) {
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
  const int wi = w >> 1, wj = w & 1, lk = l >> 4, ll = l & 15;
  const int64_t nn = (int64_t)n * n;
  sdm_double4 accr[2][2], acci[2][2];
  for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) for (int r = 0; r < 4; r++) { accr[a][b][r] = 0.0; acci[a][b][r] = 0.0; }
  const int tend = min(n, (J + 1) * TILE);                     // rows t <= min(i,j) <= last column of the J tile
  for (int t0 = 0; t0 < tend; t0 += TILE) {
    {
      // As[t][i] = U(t0+t, I*64+i), Bs[t][j] = U(t0+t, J*64+j); loads first, masked LDS stores after
      const int tt = tid & 63, cq = tid >> 6;
      double av[TILE / 4], bv[TILE / 4], ai[HERM ? TILE / 4 : 1], bi[HERM ? TILE / 4 : 1];
      const int t = t0 + tt;
#pragma unroll
      for (int q = 0; q < TILE / 4; q++) {
        const int ci = I * TILE + cq + 4 * q, cj = J * TILE + cq + 4 * q;
        const int64_t oi = (int64_t)min(ci, n - 1) * n + min(t, n - 1), oj = (int64_t)min(cj, n - 1) * n + min(t, n - 1);
        av[q] = u[oi]; bv[q] = u[oj];
        if (HERM) { ai[q] = u[nn + oi]; bi[q] = u[nn + oj]; }
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < TILE / 4; q++) {
        const int ci = I * TILE + cq + 4 * q, cj = J * TILE + cq + 4 * q;
        As[tt][cq + 4 * q] = (ci < n && t <= ci) ? av[q] : 0.0;
        Bs[tt][cq + 4 * q] = (cj < n && t <= cj) ? bv[q] : 0.0;
        if (HERM) {
          Ai[tt][cq + 4 * q] = (ci < n && t < ci) ? ai[q] : 0.0;      // Im u(i,i) is taken as 0
          Bi[tt][cq + 4 * q] = (cj < n && t < cj) ? bi[q] : 0.0;
        }
      }
    }
    __syncthreads();
#pragma unroll 4
    for (int kk = 0; kk < TILE; kk += 4) {
      double ar[2], br[2], aim[2], bim[2];
#pragma unroll
      for (int a = 0; a < 2; a++) { ar[a] = As[kk + lk][wi * 32 + a * 16 + ll]; if (HERM) aim[a] = Ai[kk + lk][wi * 32 + a * 16 + ll]; }
#pragma unroll
      for (int b = 0; b < 2; b++) { br[b] = Bs[kk + lk][wj * 32 + b * 16 + ll]; if (HERM) bim[b] = Bi[kk + lk][wj * 32 + b * 16 + ll]; }
#pragma unroll
      for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) {
          // result register r of lane (lk, ll): row (J dimension) lk+4r, column (I dimension) ll -- see update_tile
          accr[a][b] = SDM_MFMA_F64_16x16x4(br[b], ar[a], accr[a][b]);
          if (HERM) {
            accr[a][b] = SDM_MFMA_F64_16x16x4(bim[b], aim[a], accr[a][b]);
            acci[a][b] = SDM_MFMA_F64_16x16x4(bim[b], ar[a], acci[a][b]);        // + ur_ti * ui_tj
            acci[a][b] = SDM_MFMA_F64_16x16x4(-br[b], aim[a], acci[a][b]);       // - ui_ti * ur_tj
          }
        }
    }
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int i = I * TILE + wi * 32 + a * 16 + ll, j = J * TILE + wj * 32 + b * 16 + lk + 4 * r;
        if (i < n && j < n) {
          const int pi = perm ? perm[i] : i, pj = perm ? perm[j] : j;
          const double zr = accr[a][b][r];                     // Z(i,j)
          y[(int64_t)pj * n + pi] = zr;
          if (I != J) y[(int64_t)pi * n + pj] = zr;
          if (HERM) {
            const double zi = (i == j) ? 0.0 : acci[a][b][r];  // Im Z(i,j); Z(j,i) = conj
            y[nn + (int64_t)pj * n + pi] = zi;
            if (I != J) y[nn + (int64_t)pi * n + pj] = -zi;
          }
        }
      }
}

__global__ void __launch_bounds__(256)
k_invcholfac(const double *u, double *y, const int *perm, PsdBlocks B) {
  __shared__ double As[TILE][TILE], Bs[TILE][TILE];
  SDM_DYN_SMEM(smem);                                          // Hermitian blocks: the imaginary planes
  const int k = blockIdx.y, n = B.n[k];
  const int nt = (n + TILE - 1) / TILE;
  const int t = blockIdx.x;
  if (t >= nt * (nt + 1) / 2) return;
  int I = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while ((I + 1) * (I + 2) / 2 <= t) I++;
  while (I * (I + 1) / 2 > t) I--;
  const int J = t - I * (I + 1) / 2;
  const int *pk = perm ? perm + B.poff[k] : nullptr;
  if (k < B.rsdpN) invchol_tile<false>(u + B.off[k], y + B.off[k], pk, n, I, J, As, Bs, nullptr, nullptr);
  else invchol_tile<true>(u + B.off[k], y + B.off[k], pk, n, I, J, As, Bs, (double (*)[TILE])smem, (double (*)[TILE])smem + TILE);
}

// u, y: device, lenud doubles; perm: device int32 (0-based, concatenated per block) or null
void psd_invcholfac(hipStream_t st, const double *u, double *y, const int *perm, const std::vector<int> &ns, int rsdpN,
                    DevBuf<int> &d_n, DevBuf<int64_t> &d_off, DevBuf<int> &d_poff, bool tables_ready) {
  const int nb = (int)ns.size();
  if (nb == 0) return;
  std::vector<int64_t> off(nb); std::vector<int> poff(nb);
  int64_t o = 0; int po = 0, maxt = 1; bool herm = false;
  for (int k = 0; k < nb; k++) {
    off[k] = o; poff[k] = po;
    o += (int64_t)ns[k] * ns[k] * (k < rsdpN ? 1 : 2); po += ns[k];
    const int nt = (ns[k] + TILE - 1) / TILE;
    maxt = std::max(maxt, nt * (nt + 1) / 2);
    if (k >= rsdpN) herm = true;
  }
  if (!tables_ready) { d_n.upload(ns); d_off.upload(off); d_poff.upload(poff); }
  PsdBlocks B{d_n.p, d_off.p, d_poff.p, rsdpN};
  const size_t lds = herm ? 2 * sizeof(double) * TILE * TILE : 0;
#ifndef SDM_EMU
  if (lds > 0) SDM_HIP_CHECK(hipFuncSetAttribute((const void *)k_invcholfac, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
#endif
  SDM_LAUNCH(k_invcholfac, dim3(maxt, nb), dim3(256), lds, st, u, y, perm, B);
  SDM_HIP_CHECK(hipGetLastError());
}

}  // namespace sdm

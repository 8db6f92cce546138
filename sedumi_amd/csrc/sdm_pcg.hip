// sdm_pcg.hip -- the other half of every normal-equations solve: the operators wrapPcg.m / loopPcg.m
// apply around the triangular solves, on the resident plan, so that a whole wrapPcg body can stay on the device:
//   Amul      y = At' x   /   x = At y  (+ dense columns)            Amul.m:43-56
//   vecsym    Yk = (Xk + Xk')/2 per PSD block (Hermitian: Re symmetric, Im skew)   vecsym.c:50-125
//   psdscale  y[k] = vec(Ldk' Xk Ldk)  /  vec(Udk' Xk Udk)  with the pivot order ud.perm      psdscale.m:76-119
// Vectors live in plan buffers: "xN" (a cone-space vector, N doubles), "psd" (lenud doubles), "rhs" / "y" (m doubles):
//   amul(0): "xN" -> "rhs" ; ldlsolve: "rhs" -> "y" ; amul(1): "y" -> "xN" ; vecsym: "xN" in place ;
//   psdscale: PSD part of "xN" -> "psd"   -- the order in which wrapPcg.m:47-66 chains them.
#include "sdm_plan.h"
#include <algorithm>

namespace sdm {

// ---------------------------------------------------------------- Amul
// y = (x' At)' : one wavefront per constraint column (coalesced over its nonzeros), fixed-order reduction
__global__ void __launch_bounds__(256)
k_amul_cols(double *y, const double *x, const int64_t *Ajc, const int *Air, const double *Apr, int m) {
  const int lane = threadIdx.x & 63, w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
  for (int j = w; j < m; j += nw) {
    double a = 0.0;
    for (int64_t t = Ajc[j] + lane; t < Ajc[j + 1]; t += 64) a += Apr[t] * x[Air[t]];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if (lane == 0) y[j] = a;
  }
}
// x = At y : the transposed copy (row = cone variable), one work-item per row, entries in column order: deterministic
__global__ void k_amul_rows(double *x, const double *y, const int64_t *Tjc, const int *Tir, const double *Tpr, int64_t N) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= N) return;
  double a = 0.0;
  for (int64_t t = Tjc[r]; t < Tjc[r + 1]; t++) a += Tpr[t] * y[Tir[t]];
  x[r] = a;
}
// dense columns (Amul.m:50-56): y += dense.A x(dense.cols)   /   x(dense.cols) = dense.A' y
__global__ void k_amul_dense(double *y, double *x, const double *Aden, const int *cols, int m, int nden, int transp) {
  if (!transp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    double a = 0.0;
    for (int k = 0; k < nden; k++) a += Aden[(int64_t)k * m + i] * x[cols[k]];
    y[i] += a;
  } else {
    __shared__ double red[256];
    const int k = blockIdx.x;
    double a = 0.0;
    for (int i = threadIdx.x; i < m; i += blockDim.x) a += Aden[(int64_t)k * m + i] * y[i];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int s = blockDim.x >> 1; s > 0; s >>= 1) { if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0) x[cols[k]] = red[0];
  }
}

// ---------------------------------------------------------------- vecsym
__global__ void k_vecsym(double *x, const int *bn, const int64_t *boff, const int *bherm, int nblk) {
  const int b = blockIdx.y;
  const int n = bn[b];
  double *X = x + boff[b];
  const int64_t nn = (int64_t)n * n;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nn; e += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(e / n), j = (int)(e % n);                    // column i, row j; handle the pairs with j > i, and the diagonal
    if (j > i) {
      const double v = (X[e] + X[(int64_t)j * n + i]) / 2;           // symproj (vecsym.c:50-66)
      X[e] = v; X[(int64_t)j * n + i] = v;
      if (bherm[b]) {
        double *Xi = X + nn;
        const double s = (Xi[e] - Xi[(int64_t)j * n + i]) / 2;       // skewproj (vecsym.c:73-89)
        Xi[e] = s; Xi[(int64_t)j * n + i] = -s;
      }
    } else if (j == i && bherm[b]) X[nn + e] = 0.0;
  }
}

// ---------------------------------------------------------------- psdscale
// 64x64 output tiles on the FP64 matrix cores, operands fetched through the accessors of the two passes:
//   pass 1  W = Xp T      Xp(r,k) = X(pp(r), pp(k)) when the pivot order applies on the way in (psdscale.m:96-101),
//                         T = tril(U) (transp = 0) or triu(U) (transp = 1)        (psdscale.m:85-89)
//   pass 2  Y = T^H W     written through the pivot order on the way out when transp (psdscale.m:106-111)
// Hermitian blocks carry [Re; Im] planes; the complex products are sums of real ones.
constexpr int PT = 256, PP = 65;
struct PsdBlk { const int *n; const int64_t *off; const int *herm, *poff; };
struct PsdAcc { sdm_double4 t[2][2]; };

__device__ __forceinline__ void psd_mma(PsdAcc &acc, const double *As, const double *Bs, int wave, int lane) {
  const int rb = 32 * (wave & 1) + (lane & 15), cb = 32 * (wave >> 1) + (lane & 15), kq = lane >> 4;
#pragma unroll 4
  for (int kk = 0; kk < 64; kk += 4) {
    const double a0 = As[(kk + kq) * PP + rb], a1 = As[(kk + kq) * PP + rb + 16];
    const double b0 = Bs[(kk + kq) * PP + cb], b1 = Bs[(kk + kq) * PP + cb + 16];
    acc.t[0][0] = SDM_MFMA_F64_16x16x4(a0, b0, acc.t[0][0]);
    acc.t[0][1] = SDM_MFMA_F64_16x16x4(a0, b1, acc.t[0][1]);
    acc.t[1][0] = SDM_MFMA_F64_16x16x4(a1, b0, acc.t[1][0]);
    acc.t[1][1] = SDM_MFMA_F64_16x16x4(a1, b1, acc.t[1][1]);
  }
}

template <int PASS>
__global__ void __launch_bounds__(PT)
k_psdscale(double *out, const double *in, const double *u, PsdBlk B, const int *items, const int *perm, int transp, int use_perm) {
  __shared__ double As[64 * PP], Bs[64 * PP];
  const int b = items[4 * blockIdx.x], I = items[4 * blockIdx.x + 1], J = items[4 * blockIdx.x + 2];
  const int n = B.n[b], herm = B.herm[b];
  const int64_t nn = (int64_t)n * n;
  const double *U = u + B.off[b], *Xin = in + B.off[b];
  double *O = out + B.off[b];
  const int *pp = (use_perm && perm) ? perm + B.poff[b] : nullptr;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r0 = 64 * I, c0 = 64 * J;
  // T(k, c) masked to the triangle, plane pl
  auto Tval = [&](int k, int c, int pl) -> double {
    if (k >= n || c >= n) return 0.0;
    if (transp ? (k > c) : (k < c)) return 0.0;
    return U[(int64_t)pl * nn + (int64_t)c * n + k];
  };
  const int nplanes = herm ? 2 : 1;
  for (int opl = 0; opl < nplanes; opl++) {
    PsdAcc acc;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int r = 0; r < 4; r++) acc.t[i][j][r] = 0.0;
    // terms of the (complex) product contributing to output plane opl: (plane of A, plane of B, sign)
    //   pass 1  W = X T      : Wre = Xre Tre - Xim Tim ; Wim = Xre Tim + Xim Tre
    //   pass 2  Y = T^H W    : Yre = Tre' Wre + Tim' Wim ; Yim = Tre' Wim - Tim' Wre
    const int nterm = herm ? 2 : 1;
    for (int term = 0; term < nterm; term++) {
      int apl, bpl; double sgn;
      if (PASS == 1) { apl = term; bpl = opl ^ term; sgn = (opl == 0 && term == 1) ? -1.0 : 1.0; }
      else { apl = term; bpl = opl ^ term; sgn = (opl == 1 && term == 1) ? -1.0 : 1.0; }
      for (int kb = 0; kb < n; kb += 64) {
        __syncthreads();
        for (int e = tid; e < 64 * 64; e += PT) {
          const int rr = e & 63, k = e >> 6;                        // As[k][row], Bs[k][col]
          const int gr = r0 + rr, gk = kb + k, gc = c0 + rr;
          double a, bv;
          if (PASS == 1) {
            a = (gr < n && gk < n) ? Xin[(int64_t)apl * nn + (int64_t)(pp ? pp[gk] : gk) * n + (pp ? pp[gr] : gr)] : 0.0;   // Xp(gr, gk)
            bv = Tval(gk, gc, bpl);
          } else {
            a = Tval(gk, gr, apl);                                    // T^H(gr, gk) = conj(T(gk, gr)): the sign is in sgn
            bv = (gk < n && gc < n) ? Xin[(int64_t)bpl * nn + (int64_t)gc * n + gk] : 0.0;                                 // W(gk, gc)
          }
          As[k * PP + rr] = sgn * a;
          Bs[k * PP + rr] = bv;
        }
        __syncthreads();
        psd_mma(acc, As, Bs, wave, lane);
      }
    }
    // write the tile
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int row = r0 + 32 * (wave & 1) + 16 * i + (lane >> 4) + 4 * r, col = c0 + 32 * (wave >> 1) + 16 * j + (lane & 15);
          if (row < n && col < n) {
            double v = acc.t[i][j][r];
            if (PASS == 2 && opl == 1 && row == col) v = 0.0;         // psdscale.m:116: Im diag = 0
            const int orow = (PASS == 2 && pp) ? pp[row] : row, ocol = (PASS == 2 && pp) ? pp[col] : col;
            O[(int64_t)opl * nn + (int64_t)ocol * n + orow] = v;
          }
        }
  }
}

// =========================================================================== host
static void pcg_prepare(sdm_plan *P) {
  AdaPlan &A = P->ada;
  if (A.pcg_ready) return;
  if (A.h_Ajc.empty()) throw std::runtime_error("Amul / psdscale / vecsym need the problem data of sdm_plan_set_ada");
  const sdm_int N = A.N, m = A.m;
  // transposed copy of At (rows = cone variables) for x = At y
  std::vector<int64_t> Tjc((size_t)N + 1, 0);
  for (sdm_int t = 0; t < A.nnzA; t++) Tjc[(size_t)A.h_Air[t] + 1]++;
  for (sdm_int i = 0; i < N; i++) Tjc[i + 1] += Tjc[i];
  std::vector<int> Tir((size_t)std::max<sdm_int>(A.nnzA, 1));
  std::vector<double> Tpr((size_t)std::max<sdm_int>(A.nnzA, 1));
  { std::vector<int64_t> nxt(Tjc.begin(), Tjc.end() - 1);
    for (sdm_int j = 0; j < m; j++)
      for (sdm_int t = A.h_Ajc[j]; t < A.h_Ajc[j + 1]; t++) { const int64_t q = nxt[A.h_Air[t]]++; Tir[q] = (int)j; Tpr[q] = A.h_Apr[t]; } }
  A.d_Tjc.upload(Tjc); A.d_Tir.upload(Tir); A.d_Tpr.upload(Tpr);
  A.xN.alloc((size_t)std::max<sdm_int>(N, 1));
  SDM_HIP_CHECK(hipMemset(A.xN.p, 0, (size_t)std::max<sdm_int>(N, 1) * sizeof(double)));
  A.psd.alloc((size_t)std::max<sdm_int>(A.lenud, 1)); A.psdtmp.alloc((size_t)std::max<sdm_int>(A.lenud, 1));
  // PSD block tables and the tile list of the psdscale passes
  const int nb = (int)A.psd_n.size();
  std::vector<int> bn(std::max(nb, 1), 0), bh(std::max(nb, 1), 0), bp(std::max(nb, 1), 0), items;
  std::vector<int64_t> bo(std::max(nb, 1), 0);
  int po = 0;
  for (int k = 0; k < nb; k++) {
    bn[k] = (int)A.psd_n[k]; bo[k] = A.psd_udoff[k]; bh[k] = k >= A.rsdpN ? 1 : 0; bp[k] = po; po += bn[k];
    const int nt = (bn[k] + 63) / 64;
    for (int I = 0; I < nt; I++) for (int J = 0; J < nt; J++) { items.push_back(k); items.push_back(I); items.push_back(J); items.push_back(0); }
  }
  A.pb_n.upload(bn); A.pb_off.upload(bo); A.pb_herm.upload(bh); A.pb_poff.upload(bp);
  A.pcg_ntiles = (int)items.size() / 4;
  if (items.empty()) items.assign(4, 0);
  A.pb_items.upload(items);
  A.pcg_ready = true;
}

void pcg_amul(sdm_plan *P, int transp) {
  pcg_prepare(P);
  AdaPlan &A = P->ada;
  const int m = (int)A.m;
  if (!transp) {
    SDM_KLAUNCH(P, k_amul_cols, dim3(std::max(1, std::min(1024, (m + 3) / 4))), dim3(256), 0, P->rhs.p, A.xN.p, A.d_Ajc.p, A.d_Air.p, A.d_Apr.p, m);
    if (A.aden_n > 0) SDM_KLAUNCH(P, k_amul_dense, dim3((m + 255) / 256), dim3(256), 0, P->rhs.p, A.xN.p, A.aden.p, A.aden_cols.p, m, A.aden_n, 0);
  } else {
    SDM_KLAUNCH(P, k_amul_rows, dim3((unsigned)((A.N + 255) / 256)), dim3(256), 0, A.xN.p, P->y.p, A.d_Tjc.p, A.d_Tir.p, A.d_Tpr.p, (int64_t)A.N);
    if (A.aden_n > 0) SDM_KLAUNCH(P, k_amul_dense, dim3(A.aden_n), dim3(256), 0, P->y.p, A.xN.p, A.aden.p, A.aden_cols.p, m, A.aden_n, 1);
  }
}

void pcg_set_dense(sdm_plan *P, sdm_int nden, const sdm_int *cols, const double *Aden) {
  pcg_prepare(P);
  AdaPlan &A = P->ada;
  A.aden_n = (int)nden;
  if (nden <= 0) return;
  std::vector<int> c((size_t)nden);
  for (sdm_int k = 0; k < nden; k++) { if (cols[k] < 0 || cols[k] >= A.N) throw std::runtime_error("dense.cols out of range"); c[k] = (int)cols[k]; }
  A.aden_cols.upload(c);
  A.aden.upload(Aden, (size_t)(A.m * nden));
}

void pcg_vecsym(sdm_plan *P) {
  pcg_prepare(P);
  AdaPlan &A = P->ada;
  const int nb = (int)A.psd_n.size();
  if (nb == 0) return;
  const int64_t base = A.N - A.lenud;                               // PSD part = the tail of a cone-space vector
  SDM_KLAUNCH(P, k_vecsym, dim3(64, nb), dim3(256), 0, A.xN.p + base, A.pb_n.p, A.pb_off.p, A.pb_herm.p, nb);
}

// perm: device int32, per block concatenated, 0-based (the one sdm_plan_invcholfac uploaded) or null
void pcg_psdscale(sdm_plan *P, int transp, bool with_perm) {
  pcg_prepare(P);
  AdaPlan &A = P->ada;
  if (A.pcg_ntiles == 0) return;
  if (A.ufac.n < (size_t)A.lenud) throw std::runtime_error("psdscale: upload buffer \"u\" (d.u) first");
  if (with_perm && !A.ic_has_perm) throw std::runtime_error("psdscale: no pivot order resident (sdm_plan_invcholfac with perm uploads it)");
  PsdBlk B; B.n = A.pb_n.p; B.off = A.pb_off.p; B.herm = A.pb_herm.p; B.poff = A.pb_poff.p;
  const int64_t base = A.N - A.lenud;
  const int *perm = with_perm ? A.ic_perm.p : nullptr;
  // prep = ~transp (the pivot order on the way in), postp = transp (on the way out)   psdscale.m:67-69
  SDM_KLAUNCH(P, k_psdscale<1>, dim3(A.pcg_ntiles), dim3(PT), 0, A.psdtmp.p, A.xN.p + base, A.ufac.p, B, A.pb_items.p, perm, transp,
              (with_perm && !transp) ? 1 : 0);
  SDM_KLAUNCH(P, k_psdscale<2>, dim3(A.pcg_ntiles), dim3(PT), 0, A.psd.p, A.psdtmp.p, A.ufac.p, B, A.pb_items.p, perm, transp,
              (with_perm && transp) ? 1 : 0);
}

}  // namespace sdm

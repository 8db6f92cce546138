/*
 * sedumi_hip.h -- C ABI of libsedumi_hip.so, the MI355X (gfx950) implementation
 * of SeDuMi's per-iteration normal-equations hot path.
 *
 * Two tiers:
 *
 *  (1) MEX-equivalent entry points  sdm_<mexname>(...)
 *      One function per reference MEX gateway on the hot path.  Arguments are
 *      plain pointers / sizes carrying exactly what the gateway unpacks from
 *      its mxArray inputs (CSC triples with 0-based int64 indices, doubles);
 *      outputs are written into caller-allocated host buffers.  These are what
 *      a mexFunction shim (sedumi_amd/mex/) or any other
 *      FFI binds; sedumi.m stays unchanged.
 *
 *  (2) Resident "plan" API  sdm_plan_*
 *      The same kernels with problem data, ADA', the factor and work vectors
 *      kept in HBM across the 4-6 calls of an IPM iteration,
 *      used by bench.py and by a MATLAB-free driver.
 *
 * Conventions: all indices 0-based int64 (sdm_int) unless stated; sparse
 * matrices are CSC (jc[n+1], ir[nnz], pr[nnz]); every function returns 0 on
 * success, non-zero on error (sdm_last_error() gives the message).  There is
 * NO CPU fallback: without a usable HIP device every compute entry fails.
 *
 * Reference citations are relative to the SeDuMi 1.3.7 tree (/root/reference).
 */
#ifndef SEDUMI_HIP_H
#define SEDUMI_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int64_t sdm_int;

/* ----------------------------------------------------------------- general */
const char *sdm_last_error(void);
/* "hip-gfx950" for the product library ("emu" for the tests/hipemu build). */
const char *sdm_backend(void);
int sdm_device_count(void);
int sdm_set_device(int dev);

/* Cholesky pivot parameters: pars.chol of checkpars.m:144-168 as read by the
 * blkchol gateway (blkchol.c:289-311). */
typedef struct {
  double canceltol; /* 1e-12 */
  double maxu;      /* 5e5 (gateway default 5e2 when pars omitted) */
  double abstol;    /* 1e-20 */
} sdm_cholpars;

/* Cone description: what conepars (sdmauxCone.c:48-134) and the getada
 * gateways read from K after pretransfo (pretransfo.m:531-542). */
typedef struct {
  sdm_int lpN;             /* K.l (includes the artificial x0 row)            */
  sdm_int lorN;            /* length(K.q)                                     */
  const sdm_int *lorNL;    /* K.q, length lorN                                */
  sdm_int sdpN;            /* length(K.s)                                     */
  sdm_int rsdpN;           /* K.rsdpN: first rsdpN PSD blocks real symmetric  */
  const sdm_int *sdpNL;    /* K.s, length sdpN                                */
} sdm_cone;

/* ================================================================ tier (1) */

/* --- symbolic (host, integer; bit-exact with the reference) --------------- */

/* perm = ordmmdmex(X)            ordmmdmex.c:75-139 -> ordmmd.c:51 (GENMMD)
 * X m x m symmetric pattern incl. diagonal.  perm[m] out, 0-based. */
int sdm_ordmmd(sdm_int m, const sdm_int *Xjc, const sdm_int *Xir, sdm_int *perm);

/* L = symfctmex(X, perm)         symfctmex.c:127-272 -> symfct.c sfinit_/symfct_
 * Two-call protocol: first call with Ljc=Lir=NULL returns *nsuper and *nnzl;
 * second call fills perm_out[m] (post-ordered), xsuper[nsuper+1], Ljc[m+1],
 * Lir[nnzl] (every column carries its full sorted row list, diagonal first). */
int sdm_symfct(sdm_int m, const sdm_int *Xjc, const sdm_int *Xir, const sdm_int *perm_in,
               sdm_int *perm_out, sdm_int *nsuper, sdm_int *xsuper, sdm_int *nnzl,
               sdm_int *Ljc, sdm_int *Lir);

/* tmpsiz = choltmpsiz(L)         choltmpsiz.c:57-101 */
int sdm_choltmpsiz(sdm_int m, const sdm_int *Ljc, const sdm_int *Lir, sdm_int nsuper,
                   const sdm_int *xsuper, sdm_int *tmpsiz);
/* split = cholsplit(L, cachsz)   cholsplit.c:59-111 (split[m] out) */
int sdm_cholsplit(sdm_int m, const sdm_int *Ljc, sdm_int nsuper, const sdm_int *xsuper,
                  double cachsz_kb, sdm_int *split);

/* [perm, dz] = incorder(At, Ajc1, ifirst)          incorder.c:216-330 -> :140-209  (structure prep of getada3,
 * sedumi.m:378, and of the dense columns, symbcholden.m:50).  Bit-exact with the reference's greedy
 * scan, in O((nnz + m) log m) instead of O(m^2).  perm[m] 0-based; dzjc[m+1]; dzir (row subscripts of At, in the order
 * the columns list them) needs min(N - first, nonzeros in range) entries.  Ajc1 NULL = whole columns (first = 0). */
int sdm_incorder(sdm_int N, sdm_int m, const sdm_int *Atjc, const sdm_int *Atir, const sdm_int *Ajc1, sdm_int first,
                 sdm_int *perm, sdm_int *dzjc, sdm_int *dzir);

/* --- ADA' ------------------------------------------------------------------ */

/* ADA = getada1(ADA, A, Ajc2, perm, d, blkstart)        getada1.c:161-261
 * ADApr[nnz(ADA)] out: values on triu(ADA(perm,perm)) at original positions,
 * zero elsewhere.  Ajc2[m] = absolute offset (into Air/Apr) one past the last
 * LP/Lorentz nonzero of each column (= Ablkjc(:,3)); dl[lpN], ddet[lorN];
 * qblkstart[lorN+1] = 0-based K.qblkstart. */
int sdm_getada1(sdm_int m, const sdm_int *ADAjc, const sdm_int *ADAir,
                sdm_int N, const sdm_int *Ajc, const sdm_int *Air, const double *Apr,
                const sdm_int *Ajc2, const sdm_int *perm,
                sdm_int lpN, const double *dl, sdm_int lorN, const double *ddet,
                const sdm_int *qblkstart, double *ADApr);

/* ADA = getada2(ADA, DAt, Aord, K)                      getada2.c:127-214
 * ADApr in/out (copy of the input values updated on triu(ADA(qperm,qperm))).
 * DAt.q is lorN x m CSC. */
int sdm_getada2(sdm_int m, const sdm_int *ADAjc, const sdm_int *ADAir, double *ADApr,
                sdm_int lorN, const sdm_int *Qjc, const sdm_int *Qir, const double *Qpr,
                const sdm_int *qperm);

/* [ADA,absd] = getada3(ADA, A, Ajc1, Aord, udsqr, K)    getada3.c:370-569
 * ADApr in/out (symmetric on return), absd[m] out.  Ajc1[m] = absolute offset
 * of the first PSD nonzero of each column; sperm = Aord.sperm (only its
 * triangular bookkeeping is reproduced -- the result is order independent);
 * udsqr = concatenated full D_k (Hermitian blocks as [Re;Im]); psd_blkstart
 * [sdpN+1] = 0-based row offsets of the PSD blocks in At (K.sblkstart-1). */
int sdm_getada3(sdm_int m, const sdm_int *ADAjc, const sdm_int *ADAir, double *ADApr,
                sdm_int N, const sdm_int *Ajc, const sdm_int *Air, const double *Apr,
                const sdm_int *Ajc1, const sdm_int *sperm, const double *udsqr,
                const sdm_cone *K, const sdm_int *psd_blkstart, double *absd);

/* absd = getada(A, K, d, DAt)  [global ADA_sedumi_]      getada.m:13-40, called by sedumi.m:446-448 when sum(K.s)==0
 * The whole ADA' of an LP / SOCP problem:  DAt.q' DAt.q + Alq' diag([d.l; -d.det; d.det(k) per norm-bound row]) Alq,
 * Alq = A(1:K.mainblks(3)-1,:).  ADApr[nnz(ADA)] out on the pattern of the global ADA_sedumi_ (symmetric, both
 * triangles; entries MATLAB's sparse() would drop come out as explicit zeros), absd[m] = diag(ADA) out.
 * DAt.q lorN x m CSC (ignored when lorN == 0); qblkstart[lorN+1] 0-based. */
int sdm_getada(sdm_int m, const sdm_int *ADAjc, const sdm_int *ADAir,
               sdm_int N, const sdm_int *Ajc, const sdm_int *Air, const double *Apr,
               sdm_int lpN, const double *dl, sdm_int lorN, const double *ddet, const sdm_int *qblkstart,
               const sdm_int *Qjc, const sdm_int *Qir, const double *Qpr, double *ADApr, double *absd);

/* --- numeric factor / solves ---------------------------------------------- */

/* [L.L,L.d,L.skip,L.add] = blkchol(L, X, pars, absd)    blkchol.c:239-440
 * In: symbolic L (Ljc,Lir,perm,xsuper), X = ADA (full symmetric CSC), absd[m]
 * or NULL (then diag(X(perm,perm)) is used, blkchol.c:376-381).
 * Out: Lpr[nnz(L)], d[m], skip/add as (count, idx[m], val[m]) sorted by index
 * -- exactly the sparse m x 1 outputs of the gateway. */
int sdm_blkchol(sdm_int m, const sdm_int *Ljc, const sdm_int *Lir, const sdm_int *perm,
                sdm_int nsuper, const sdm_int *xsuper,
                const sdm_int *Xjc, const sdm_int *Xir, const double *Xpr,
                const sdm_cholpars *pars, const double *absd,
                double *Lpr, double *d,
                sdm_int *nskip, sdm_int *skip_idx, double *skip_val,
                sdm_int *nadd, sdm_int *add_idx, double *add_val);

/* y = fwblkslv(L, b)     dense b (m x nrhs, column major)   fwblkslv.c:193-320
 * y = L.L \ b(L.perm,:) */
int sdm_fwblkslv(sdm_int m, const sdm_int *Ljc, const sdm_int *Lir, const double *Lpr,
                 const sdm_int *perm, sdm_int nsuper, const sdm_int *xsuper,
                 sdm_int nrhs, const double *b, double *y);
/* y = bwblkslv(L, b)     y(L.perm,:) = L.L' \ b              bwblkslv.c:182-298 */
int sdm_bwblkslv(sdm_int m, const sdm_int *Ljc, const sdm_int *Lir, const double *Lpr,
                 const sdm_int *perm, sdm_int nsuper, const sdm_int *xsuper,
                 sdm_int nrhs, const double *b, double *y);
/* sparse-RHS variants: y = fwblkslv(L,b,ysymb) / bwblkslv(L,b,ysymb)
 * (fwblkslv.c:305-317 selfwsolve, bwblkslv.c:279-291 selbwsolve).  b is m x n
 * CSC, the pattern of y (Yjc,Yir) comes from symbfwblk; Ypr[nnz(Y)] out.
 * use_perm: 1 for the forward variant (b is mapped through invperm), 0 for the
 * backward variant (no permutation, as the reference). */
int sdm_fwblkslv_sparse(sdm_int m, const sdm_int *Ljc, const sdm_int *Lir, const double *Lpr,
                        const sdm_int *perm, sdm_int nsuper, const sdm_int *xsuper,
                        sdm_int n, const sdm_int *Bjc, const sdm_int *Bir, const double *Bpr,
                        const sdm_int *Yjc, const sdm_int *Yir, double *Ypr);
int sdm_bwblkslv_sparse(sdm_int m, const sdm_int *Ljc, const sdm_int *Lir, const double *Lpr,
                        sdm_int nsuper, const sdm_int *xsuper,
                        sdm_int n, const sdm_int *Bjc, const sdm_int *Bir, const double *Bpr,
                        const sdm_int *Yjc, const sdm_int *Yir, double *Ypr);

/* --- dense columns (product-form rank-1 updates, deninfac.m:58-94) --------- */

/* x = symbfwblk(L, b)            symbfwblk.c:270-377
 * Pattern of L.L \ b(L.perm,:) for sparse b (m x n CSC pattern).  Two-call protocol: with Xir == NULL only
 * Xjc[n+1] is written (Xjc[n] = nnz), the second call fills Xir[nnz] (row indices in the factor's order). */
int sdm_symbfwblk(sdm_int m, const sdm_int *Ljc, const sdm_int *Lir, const sdm_int *perm,
                  sdm_int nsuper, const sdm_int *xsuper,
                  sdm_int n, const sdm_int *Bjc, const sdm_int *Bir, sdm_int *Xjc, sdm_int *Xir);

/* Lden = finsymbden(LAD, perm, dz, firstq)      finsymbden.c:112-214
 * LAD m x n pattern (dense columns after the forward solve), perm[nperm] 0-based column order (nperm <= n:
 * the n - nperm Lorentz trace columns are attached behind their block column), dz m x nperm with CUMULATIVE
 * column pointers dzjc[nperm+1] over the row list dzir, firstq 0-based.  Out: perm_out[n], dzjc_out[n+1]
 * (cumulative, duplicated for trace columns), first_out[n] (first affecting pivot, n = none), all 0-based. */
int sdm_finsymbden(sdm_int m, sdm_int n, const sdm_int *LADjc, const sdm_int *LADir,
                   sdm_int nperm, const sdm_int *perm, const sdm_int *dzjc, const sdm_int *dzir,
                   sdm_int firstq, sdm_int *perm_out, sdm_int *dzjc_out, sdm_int *first_out);

/* [Lden, Ld] = dpr1fact(x, d, Lsymb, smult, maxu)      dpr1fact.c:630-848 (prodformfact :549-621)
 * x m x n CSC (dense columns, already forward-solved and scaled), d[m] in/out (L.d), Lsymb.dz as (dzjc[n+1]
 * cumulative, dzir), Lsymb.perm -> colperm[n], Lsymb.first -> first[n] (0-based), smult[n], maxu.
 * Out (caller allocates pnnz = sum_k dzjc[k+1] entries for beta, p, pivperm): betajc[n+1] 0-based, beta,
 * p[pnnz], pivperm[*npivperm] (0-based row order of the reordered columns, full length each), dopiv[n]. */
int sdm_dpr1fact(sdm_int m, sdm_int n, const sdm_int *Xjc, const sdm_int *Xir, const double *Xpr, double *d,
                 const sdm_int *dzjc, const sdm_int *dzir, const sdm_int *colperm, const sdm_int *first,
                 const double *smult, double maxu,
                 sdm_int *betajc, double *beta, double *p, sdm_int *pivperm, sdm_int *npivperm, sdm_int *dopiv);

/* y = fwdpr1(Lden, b)   y = PROD_k L(p_k,beta_k) \ b          fwdpr1.c:101-202, auxfwdpr1.c:44-122
 * y = bwdpr1(Lden, b)   y = (PROD_k L(p_k,beta_k))' \ b       bwdpr1.c:170-275
 * b, y m x nrhs column major; Lden as returned by sdm_dpr1fact plus dz (dzjc cumulative, dzir).
 * nden == 0: y = b (fwdpr1.c:132-135). */
int sdm_fwdpr1(sdm_int m, sdm_int nrhs, sdm_int nden, const sdm_int *dzjc, const sdm_int *dzir,
               const sdm_int *betajc, const double *beta, const double *p,
               const sdm_int *pivperm, sdm_int npivperm, const sdm_int *dopiv, const double *b, double *y);
int sdm_bwdpr1(sdm_int m, sdm_int nrhs, sdm_int nden, const sdm_int *dzjc, const sdm_int *dzir,
               const sdm_int *betajc, const double *beta, const double *p,
               const sdm_int *pivperm, sdm_int npivperm, const sdm_int *dopiv, const double *b, double *y);

/* --- next row: the input of getada3 ------------------------ */

/* y = invcholfac(u, K, perm)        invcholfac.c:59-168 (utmulx / prpiutmulx triuaux.c:175-221, invmatperm :61-70)
 * For every PSD block k of order n_k: Y_k(perm_k, perm_k) = U_k' U_k with U_k = triu(u_k); Hermitian blocks are
 * [Re; Im] planes, Im diag(U) taken as 0.  u, y: lenud = sum n_k^2 (2 n_k^2 Hermitian) doubles, blocks in K.s order;
 * perm: 0-based, concatenated per block (sum n_k entries), NULL = identity.  y is the `udsqr` of sdm_getada3. */
int sdm_invcholfac(const sdm_cone *K, const double *u, const sdm_int *perm, double *y);

/* ================================================================ tier (2) */
typedef struct sdm_plan sdm_plan;

/* Create a plan on `device` (HIP ordinal).  stream: a hipStream_t passed as
 * void* (NULL = the plan creates its own).  All numeric plan calls are
 * stream-ordered and asynchronous unless they return host data. */
sdm_plan *sdm_plan_create(int device, void *stream);
void sdm_plan_destroy(sdm_plan *p);
int sdm_plan_sync(sdm_plan *p);

/* Symbolic factor + ADA pattern (uploaded once per solve). */
int sdm_plan_set_chol(sdm_plan *p, sdm_int m, const sdm_int *Ljc, const sdm_int *Lir,
                      const sdm_int *perm, sdm_int nsuper, const sdm_int *xsuper,
                      const sdm_int *ADAjc, const sdm_int *ADAir);

/* Problem data for ADA' (uploaded once per solve): At with its LP/Lorentz |
 * PSD split (Ajc_psd[m] = absolute offset of the first PSD nonzero of each
 * column), cone, 0-based qblkstart[lorN+1] and psd_blkstart[sdpN+1], and the
 * (fixed) pattern of DAt.q (lorN x m). */
int sdm_plan_set_ada(sdm_plan *p, sdm_int N, sdm_int m, const sdm_int *Ajc, const sdm_int *Air,
                     const double *Apr, const sdm_int *Ajc_psd, const sdm_cone *K,
                     const sdm_int *qblkstart, const sdm_int *psd_blkstart,
                     const sdm_int *Qjc, const sdm_int *Qir);

/* Device buffers owned by the plan (for callers that fill them from torch /
 * other HIP code without a host round trip).  name is one of
 * "ada","absd","d","y","rhs","udsqr","dl","ddet","qpr","lpr". */
void *sdm_plan_devptr(sdm_plan *p, const char *name, sdm_int *nelem);
/* Host <-> plan buffer copies (synchronous). */
int sdm_plan_upload(sdm_plan *p, const char *name, const double *src, sdm_int nelem);
int sdm_plan_download(sdm_plan *p, const char *name, double *dst, sdm_int nelem);

/* ADA' = getada1+getada2+getada3 fused (sedumi.m:450-452) from the resident
 * scaling buffers dl, ddet, qpr (values of DAt.q), udsqr.  Result in "ada"
 * (values in ADA pattern order, symmetric) and "absd". */
int sdm_plan_getada(sdm_plan *p);
/* udsqr = invcholfac(u, K, perm) on the device: reads plan buffer "u" (sdm_plan_upload), writes plan buffer "udsqr",
 * which sdm_plan_getada then uses -- the scaled blocks never travel.  perm: host, 0-based per block, or NULL. */
int sdm_plan_invcholfac(sdm_plan *p, const sdm_int *perm);
/* qpr = values of DAt.q = diag(d.q1) * A(trace rows, :) + ddot(d.q2, A, K.qblkstart, Ablkjc)  (getDAtm.m:39-44, ddot.c:66-160;
 * SURVEY 8f N3) in the order of the Qjc/Qir pattern given to sdm_plan_set_ada: reads plan buffers "q1" (lorN doubles)
 * and "q2" (sum(K.q)-lorN doubles), writes "qpr" -- the Lorentz input of sdm_plan_getada stays on the device. */
int sdm_plan_getdatq(sdm_plan *p);
/* The same restricted to the columns j0 <= j < j1 of ADA' (and absd[j0:j1]); the other entries of "ada" are left
 * untouched.  Columns are independent given the scaling data, so the ranks of a multi-GPU job each form a panel
 * and exchange panels (sedumi_amd/dist.py does the all-gather over RCCL). */
int sdm_plan_getada_cols(sdm_plan *p, sdm_int j0, sdm_int j1);
/* Device-to-device copy between a plan buffer (elements [offset, offset+nelem)) and memory owned by the caller
 * (e.g. a torch tensor used as an RCCL send/receive buffer): to_plan != 0 copies devptr -> plan. */
int sdm_plan_copy(sdm_plan *p, const char *name, void *devptr, sdm_int offset, sdm_int nelem, int to_plan);
/* blkchol (sedumi.m:458) on the resident "ada"/"absd".  use_absd=0 takes
 * diag(ADA(perm,perm)) as in blkchol.c:380-381. */
int sdm_plan_blkchol(sdm_plan *p, const sdm_cholpars *pars, int use_absd);
/* The same, waited for, with ONE recovery: if a workgroup of a one-launch level gave up waiting for another one (the device
 * is shared with another process and not all of the launch's workgroups became resident), the factorisation is repeated
 * once on the launch-per-panel path and the plan stays on that path.  What blkchol.mex and sdm_blkchol call (they hand
 * the factor back, so they wait anyway); the asynchronous form above reports such a time-out as an error at the plan's
 * next read-back instead. */
int sdm_plan_blkchol_wait(sdm_plan *p, const sdm_cholpars *pars, int use_absd);
/* pivot report of the last factor: counts (host, synchronises) and lists. */
int sdm_plan_pivots(sdm_plan *p, sdm_int *nskip, sdm_int *skip_idx, double *skip_val,
                    sdm_int *nadd, sdm_int *add_idx, double *add_val);
/* Solves on the resident factor: "y" <- op("rhs").
 *  fw:  y = L \ rhs(perm)         bw:  y(perm) = L' \ rhs
 *  ldl: y(perm) = L' \ ( (L \ rhs(perm)) ./ d )   (wrapPcg.m:56-59, no dense cols) */
int sdm_plan_fwsolve(sdm_plan *p);
int sdm_plan_bwsolve(sdm_plan *p);
int sdm_plan_ldlsolve(sdm_plan *p);

/* hipGraph capture of a sequence of asynchronous plan calls (getada, blkchol, fw/bw/ldlsolve -- no uploads, downloads,
 * pivots or timers inside): begin, issue the calls once (they are recorded, not executed), end -> graph id;
 * sdm_plan_graph_launch replays the whole sequence with one launch on the plan's stream.  The captured sequence keeps
 * the parameters (pivot tolerances, buffers) it was recorded with. */
int sdm_plan_graph_begin(sdm_plan *p);
int sdm_plan_graph_end(sdm_plan *p, int *graph_id);
int sdm_plan_graph_launch(sdm_plan *p, int graph_id);

/* Timing of named kernels with HIP events on the plan's stream (bench.py):
 * begin/end bracket a region; *_ms returns the elapsed milliseconds. */
int sdm_plan_timer_begin(sdm_plan *p, int slot);
int sdm_plan_timer_end(sdm_plan *p, int slot);
int sdm_plan_timer_ms(sdm_plan *p, int slot, float *ms);

/* Per-kernel timing: while enabled every kernel launch of the plan is bracketed by HIP events on the
 * plan's stream.  kprof_get returns the number of launches and the summed duration of one kernel (by its
 * unqualified name, e.g. "k_bw_level"); kprof_summary writes "name:calls:ms;" for all kernels seen. */
int sdm_plan_kprof_enable(sdm_plan *p, int on);
int sdm_plan_kprof_get(sdm_plan *p, const char *kernel, sdm_int *calls, double *total_ms);
int sdm_plan_kprof_summary(sdm_plan *p, char *buf, sdm_int buflen);

/* Dense Lorentz columns of getDAtm.m:45 / deninfac.m:61 (host, O(nnz of the dense columns)):
 * Ad = adendotd(dense, d, sparAd, Ablk, blkstart)   adendotd.c:74-127   values adpr on the pattern (adjc, adir) of Ablk
 * smult = adenscale(dense, d, blkstart)             adenscale.c:62-80
 * aden = dense.A(:, dense.l+1:end) as CSC with adenjc[nq+nden+1] absolute offsets; q, dencols 0-based; blkend[k] =
 * blkstart(q(k)+2)-1; d2 = d.q2 indexed by (global subscript - firstQ), firstQ = blkstart(1)-1. */
int sdm_adendotd(sdm_int m, sdm_int nq, sdm_int nden, const sdm_int *adjc, const sdm_int *adir, double *adpr,
                 const sdm_int *sjc, const sdm_int *sir, const double *spr, const sdm_int *adenjc, const sdm_int *adenir,
                 const double *adenpr, const double *d1, const double *d2, sdm_int firstQ, const sdm_int *q,
                 const sdm_int *dencols, const sdm_int *blkend);
int sdm_adenscale(sdm_int nq, sdm_int nden, const double *detd, const sdm_int *q, const sdm_int *dencols,
                  const sdm_int *blkend, double *smult);

/* The operators wrapPcg.m:47-66 / loopPcg.m apply around the solves, on the resident plan.  Work
 * vectors are plan buffers: "xN" (a cone-space vector, N doubles), "psd" (lenud doubles), "rhs" / "y" (m doubles).
 *   pcg_init   (optional) dense columns of Amul.m:50-56: dense_cols[nden] 0-based rows of At, denseA m x nden column major
 *   amul(0)    "rhs" = At' "xN" (+ dense.A xN(dense.cols))         Amul.m:46,52
 *   amul(1)    "xN"  = At "y"   (xN(dense.cols) = dense.A' y)      Amul.m:48,54
 *   vecsym     PSD part of "xN" symmetrised in place                vecsym.c:50-125
 *   psdscale   "psd" = psdscale(ud, PSD part of "xN", K, transp): vec(Ld' X Ld) / vec(Ud' X Ud), ud.u = plan buffer "u",
 *              ud.perm = the pivot order last given to sdm_plan_invcholfac when use_perm != 0   psdscale.m:76-119 */
int sdm_plan_pcg_init(sdm_plan *p, sdm_int nden, const sdm_int *dense_cols, const double *denseA);
int sdm_plan_amul(sdm_plan *p, int transp);
int sdm_plan_vecsym(sdm_plan *p);
int sdm_plan_psdscale(sdm_plan *p, int transp, int use_perm);

/* Make a factor computed elsewhere resident: Lpr[nnz(L)] on the pattern given to sdm_plan_set_chol, d[m] = L.d
 * (NULL: only the solves without ./d are meaningful).  No pivot report (skip / add) is attached to it. */
int sdm_plan_load_factor(sdm_plan *p, const double *Lpr, const double *d);

/* Resident dense-column unit (deninfac.m:58-94; SURVEY.md 8(d): "+ sparse fwblkslv + dpr1fact + 4 x (fwdpr1+bwdpr1)").
 * set_dense (once per solve, after set_chol): the symbolic data of symbcholden.m:43-55 -- pattern of LAD = symLden.LAD
 * (m x nden CSC), dz (cumulative dzjc[nden+1], dzir), colperm = symLden.perm-1, first = symLden.first-1.
 * Every iteration: upload the dense columns Ad (m x nden column major, deninfac.m:58-59) into plan buffer "ad", then
 * sdm_plan_deninfac(smult[nden] (deninfac.m:60-62), maxuden): LAD = L \ Ad(perm,:) for all columns in one set of
 * launches, then dpr1fact on the device (the recurrences as scans; postponed pivots, dependent rows and negative multiples
 * included: *host_fallback is always 0, the argument is kept for callers of earlier versions).  Afterwards sdm_plan_ldlsolve
 * is the whole wrapPcg.m:56-59 body: fwblkslv, fwdpr1, ./Ld, bwdpr1, bwblkslv.  sdm_plan_lden downloads the factors
 * (layout of sdm_dpr1fact; buffers sized pnnz = sum_k dzjc[k+1], Ld[m]). */
int sdm_plan_set_dense(sdm_plan *p, sdm_int nden, const sdm_int *LADjc, const sdm_int *LADir, const sdm_int *dzjc,
                       const sdm_int *dzir, const sdm_int *colperm, const sdm_int *first);
int sdm_plan_deninfac(sdm_plan *p, const double *smult, double maxuden, int *host_fallback);
int sdm_plan_lden(sdm_plan *p, sdm_int *betajc, double *beta, double *pv, sdm_int *pivperm, sdm_int *npivperm,
                  sdm_int *dopiv, double *Ld);

/* The solves apply the diagonal super-blocks of L (256 .. 2048 columns: the power of two that covers the widest front,
 * sdm_solve.hip) as explicit inverses unless a block's growth  max|inv(L_PP)| * max|L_PP|  exceeds growth_max (default
 * 1e4): such a block is solved by substitution like fwblkslv.c:109-114 / bwblkslv.c:113-122.  growth_max = 0 forces
 * substitution everywhere.  Takes effect at the next factorisation (sdm_plan_blkchol).  solve_stats reports the blocks
 * of the last factorisation (synchronises).  set_solve_width (0 = automatic, or a power of two in 256 .. 2048) fixes
 * the super-block width of the NEXT sdm_plan_set_chol: narrower blocks cost less to invert after every factorisation
 * and more dependent launches per solve. */
int sdm_plan_set_growth_max(sdm_plan *p, double growth_max);
/* Super-blocks whose growth lies between growth_max and refine_max (default 1e10) -- the late iterations of an interior-point
 * run -- need not fall back to substitution (one workgroup, hundreds of microseconds): the solve can apply the explicit inverse
 * and refine the block's result twice against the factor itself (r = t - L_PP y, y += inv(L_PP) r: four short launches per
 * block and sweep), which restores the accuracy of the substitution.  mode 1 (default): the refinement launches are planned
 * while such blocks keep turning up: every sweep is numbered, a launch that meets such a block leaves the sweep's number in
 * pinned host memory, the first launch of every sweep the number of the sweep before it; the host reads both when it
 * enqueues the next sweep (no synchronisation) and switches the refinement launches on when the latest sweep known to have
 * run met such a block, off when two sweeps have run since the last one that did.  So only the first sweeps that meet such a
 * block -- those enqueued before the news arrives -- substitute.  mode 0: never (always substitute); mode 2: always.
 * Blocks beyond refine_max, or with a growth that is not a number, are always substituted. */
int sdm_plan_set_refinement(sdm_plan *p, int mode, double refine_max);
/* on = 0: the NEXT sdm_plan_set_chol plans every front on the launch-per-panel path (k_ldl_panel) -- no level is factored
 * by the one-launch kernel (k_ldl_front) and no inverse is built beside it.  Default 1.  The comparison switch of the tests
 * and tools (both paths produce the same bits); also what a caller sharing the device between processes wants (all
 * workgroups of a one-launch level must be resident at once). */
int sdm_plan_set_one_launch_fronts(sdm_plan *p, int on);
/* n > 0: the NEXT sdm_plan_set_chol deals the trailing-update tiles of a big front's panel launch to at most n workgroups beside the
 * chain and the row solves (each works through its tile pairs as a pipeline); 0 = as many as the device has compute units. */
int sdm_plan_set_tile_workgroups(sdm_plan *p, int n);
/* on = 0: the inverses of the diagonal super-blocks are built by one launch per stage (k_sinv128, k_stile) whatever the size of the
 * problem; default 1: problems whose work items fit the device at once take ONE launch with completion counters (k_sprep).  Takes
 * effect at the next factorisation.  The comparison switch of the tests: both paths give the same bits. */
int sdm_plan_set_one_launch_inverse(sdm_plan *p, int on);
int sdm_plan_set_solve_width(sdm_plan *p, sdm_int width);
int sdm_plan_get_solve_width(sdm_plan *p, sdm_int *width);      /* the width in force (after sdm_plan_set_chol) */
int sdm_plan_solve_stats(sdm_plan *p, sdm_int *nblocks, sdm_int *nbad, double *max_growth);

/* ---- Separator fronts across GPUs (sedumi_amd/dist.py SeparatorShardedSolver).  The reference relinks a finished
 * supernode to its parent snode[lindx[xlindx[s] + n_s]] (blkchol2.c:550-554) and pulls its update into it (precorrect,
 * blkchol2.c:346-420); here a rank factors the subtrees it owns and every rank holds the SAME front arena, so the update
 * matrices of the separator fronts at the top of the tree are sums of equally laid out slices -- one reduce per etree level.
 *   set_active_supernodes (before set_chol): the supernodes this plan works on (its subtrees + the top of the tree);
 *   front_layout: etree level, arena slice ("fronts": foff, fsize), update-vector slice ("wvec": woff, ms), columns (first, ns);
 *   blkchol_begin: permuteP + pivot thresholds ("ub"[2] = max diagonal, to be max-reduced across the ranks);
 *   blkchol_levels(l0, l1, extend_only): extend-add into the fronts of levels l0 .. l1-1 and (unless extend_only) their LDL';
 *   blkchol_end: the inverses for the solves;
 *   solve_levels(what, l0, l1) on the right-hand side in "rhs": what = 1 assembly of the fronts' right-hand sides (own entries +
 *     children's update vectors, in "wvec"), 2 the forward sweep of the levels without that assembly, 3 both, 4 the backward
 *     sweep of levels l1-1 .. l0 (reads the ancestors' solution from "xfin", writes "y").
 * The plan buffers "fronts", "wvec", "xfin", "ub", "panelrec" are reachable through sdm_plan_devptr / sdm_plan_copy. */
int sdm_plan_set_active_supernodes(sdm_plan *p, const int *active, sdm_int nsuper);
int sdm_plan_front_layout(sdm_plan *p, sdm_int *nlevels, sdm_int *level, sdm_int *foff, sdm_int *fsize, sdm_int *woff, sdm_int *ms, sdm_int *first, sdm_int *ns);
int sdm_plan_blkchol_begin(sdm_plan *p, const sdm_cholpars *pars, int use_absd);
int sdm_plan_blkchol_levels(sdm_plan *p, sdm_int l0, sdm_int l1, int extend_only);
int sdm_plan_blkchol_end(sdm_plan *p);
int sdm_plan_solve_levels(sdm_plan *p, int what, sdm_int l0, sdm_int l1);

/* ---- One front across GPUs (SURVEY.md 8e row blkchol, "block-cyclic dense LDL' with panel broadcasts"; sedumi_amd.dist.BlockCyclicFactor).
 * The ranks hold the same plan of ONE dense front (nsuper = 1, e.g. MAXCUT) on the launch-per-panel path (sdm_plan_set_one_launch_fronts(p, 0)
 * before set_chol) and the same ADA' values.  Tile column c (64 columns) of the front belongs to rank (c / blk) % world:
 *   set_column_owner(world, rank, blk)   before blkchol_begin; world = 1 gives the plan everything back;
 *   blkchol_panels(l0, l1, pan0, pan1)   the panel launches pan0 .. pan1-1 of the levels: the owner of tile column q factors panel q
 *                                        (cholonBlk, blkchol2.c:96-167, + the rows below it), every rank applies the trailing updates that
 *                                        are due (precorrect, blkchol2.c:346-420) to ITS tile columns -- per tile the operations and their
 *                                        order are those of the single plan: the same bits;
 *   panel_record(panel, unpack, &off, &n) after launch `panel`: its owner packs d, lb, the pivot decisions, the front's progress counters and the
 *                                        transposed diagonal block into the plan buffer "panelrec" (unpack = 0); the others store a received
 *                                        record (unpack = 1); unpack < 0 only answers.  off / n = the slice of "fronts" that holds the panel's columns.  The caller
 *                                        broadcasts both from the owner (what blkLDL's relinking, blkchol2.c:550-554, turns into when the
 *                                        supernode is spread over ranks) before anybody launches panel + 1;
 *   blkchol_end                           as above (every rank then holds the whole factor; the inverses for the solves are built by each). */
int sdm_plan_set_column_owner(sdm_plan *p, int world, int rank, int blk);
int sdm_plan_blkchol_panels(sdm_plan *p, sdm_int l0, sdm_int l1, sdm_int pan0, sdm_int pan1);
int sdm_plan_panel_record(sdm_plan *p, sdm_int panel, int unpack, sdm_int *front_offset, sdm_int *front_nelem);

/* ---- process-wide resident state behind the mexFunction shims (csrc/sdm_mexcache.hip).  Every .mex
 * binary is its own shared object, so the cache lives in this library.  The sdm_mexcache_<gateway> functions take the
 * arguments of their stateless counterparts sdm_<gateway> and give the same results, but
 *   - the device-side analysis of the problem data (At, K, ADA pattern: ada_build; symbolic factor: chol_build) is done
 *     once per solve and reused while the data presented is the same (content fingerprints, see the source file);
 *   - a value array that one gateway returned and the next one receives (ADA: getada1 -> getada2 -> getada3 -> blkchol;
 *     L.L: blkchol -> fwblkslv / bwblkslv) is not uploaded again: the device still holds it.
 * What sedumi.m:450-462 / wrapPcg.m:56-59 call every iteration therefore costs the per-iteration scaling data up and the
 * results down.  ADApr_in / ADApr: values of the input array and of the fresh array the shim returns (may alias).
 * sdm_mexcache_solve: fw != 0 forward.  sdm_mexcache_stats: counters (ada_build calls, reuses, ADA uploads, ADA taken
 * from the device, chol builds, reuses, X uploads, X resident, resident solves, stateless solves, At uploads). */
int sdm_mexcache_getada1(sdm_int m, const sdm_int *ADAjc, const sdm_int *ADAir, sdm_int N, const sdm_int *Ajc, const sdm_int *Air,
                         const double *Apr, const sdm_int *Ajc2, const sdm_int *perm, sdm_int lpN, const double *dl, sdm_int lorN,
                         const double *ddet, const sdm_int *qblkstart, double *ADApr, double token_in, double *token_out);
int sdm_mexcache_getada2(sdm_int m, const sdm_int *ADAjc, const sdm_int *ADAir, const double *ADApr_in, double *ADApr, sdm_int lorN,
                         const sdm_int *Qjc, const sdm_int *Qir, const double *Qpr, const sdm_int *qperm, double token_in, double *token_out);
int sdm_mexcache_getada3(sdm_int m, const sdm_int *ADAjc, const sdm_int *ADAir, const double *ADApr_in, double *ADApr, sdm_int N,
                         const sdm_int *Ajc, const sdm_int *Air, const double *Apr, const sdm_int *Ajc1, const double *udsqr,
                         const sdm_cone *K, const sdm_int *psd_blkstart, double *absd, double token_in, double *token_out);
int sdm_mexcache_getada(sdm_int m, const sdm_int *ADAjc, const sdm_int *ADAir, sdm_int N, const sdm_int *Ajc, const sdm_int *Air,
                        const double *Apr, sdm_int lpN, const double *dl, sdm_int lorN, const double *ddet, const sdm_int *qblkstart,
                        const sdm_int *Qjc, const sdm_int *Qir, const double *Qpr, double *ADApr, double *absd);
int sdm_mexcache_blkchol(sdm_int m, const sdm_int *Ljc, const sdm_int *Lir, const sdm_int *perm, sdm_int nsuper, const sdm_int *xsuper,
                         const sdm_int *Xjc, const sdm_int *Xir, const double *Xpr, const sdm_cholpars *pars, const double *absd,
                         double *Lpr, double *d, sdm_int *nskip, sdm_int *skip_idx, double *skip_val, sdm_int *nadd, sdm_int *add_idx,
                         double *add_val, double token_in);
int sdm_mexcache_solve(int fw, sdm_int m, const sdm_int *Ljc, const sdm_int *Lir, const double *Lpr, const sdm_int *perm, sdm_int nsuper,
                       const sdm_int *xsuper, sdm_int nrhs, const double *b, double *y);
void sdm_mexcache_stats(sdm_int *out16, sdm_int n);   /* [11] words checksummed on the host since the last clear, [12] the epoch */
/* Residency is decided by content (a checksum of every word of the host array against that of the resident data).  Arrays of up to
 * `full_below` words (default 65536) are checksummed at every presentation; a larger one once per address and epoch (= between two
 * blkchol calls), later presentations at that address in that epoch pass on length + a 512-word sampled hash: an in-place edit of
 * such an array between two calls of one iteration, at words the sample does not touch, goes unnoticed (sedumi.m never does that).
 * set_strict(1): no shortcut, every presentation is checksummed completely. */
void sdm_mexcache_set_strict(int on);
void sdm_mexcache_set_full_below(sdm_int words);
/* Lazy intermediates, opt-in (environment SEDUMI_HIP_LAZY = 1 | 2, or set_lazy): getada1.mex / getada2.mex (level 2: getada3.mex too) return
 * a TOKEN instead of ADA' -- an m x m sparse matrix whose one nonzero (1,1) is sdm_mexcache_token_base() + a serial number -- and leave the
 * values on the device; the next gateway recognises the current token (token_in of the functions above; token_out != NULL asks for one) and
 * fails loudly on any other.  sedumi.m:450-458 never looks at these arrays; default (level 0): every gateway returns the reference's array. */
void sdm_mexcache_set_lazy(int level);
int sdm_mexcache_lazy(void);
int sdm_mexcache_token_info(double token, sdm_int m, sdm_int *nnz);
int sdm_mexcache_token_pattern(double token, sdm_int m, sdm_int *jc_out, sdm_int *ir_out);
double sdm_mexcache_token_base(void);
void sdm_mexcache_set_threads(int n);   /* host threads of a checksum from 128K words on: -1 automatic (4; 8 from 1M words), 1 none */
unsigned long long sdm_mexcache_checksum(const void *words, sdm_int n);
/* dst = src (n 8-byte words) and src's content checksum in one pass, noted for the sdm_mexcache_<gateway> call that follows (the shims copy
 * the pattern of their input into the array they return: the cache then does not read the input's pattern a second time) */
unsigned long long sdm_mexcache_copy_words(void *dst, const void *src, sdm_int n);
void sdm_mexcache_forget_notes(void);   /* the shims call it first thing: notes of a gateway call that never reached the cache are dropped */   /* the content checksum of n 8-byte host words */
/* The cached plan of the factorisation for callers that drive it themselves: sdm_mexcache_plan returns the plan of the
 * symbolic factor (L.{L pattern, perm, xsuper}) and ADA pattern given, creating it on a miss (NULL + sdm_last_error on
 * failure); sdm_mexcache_remember_factor records the L.L values a factorisation returned; sdm_mexcache_factor_plan gives
 * the plan back only if the values presented ARE that factor, else NULL. */
sdm_plan *sdm_mexcache_plan(sdm_int m, const sdm_int *Ljc, const sdm_int *Lir, const sdm_int *perm, sdm_int nsuper,
                            const sdm_int *xsuper, const sdm_int *Xjc, const sdm_int *Xir);
void sdm_mexcache_remember_factor(const double *Lpr_host, sdm_int nnz);
sdm_plan *sdm_mexcache_factor_plan(sdm_int m, const sdm_int *Ljc, const sdm_int *Lir, const double *Lpr,
                                   const sdm_int *perm, sdm_int nsuper, const sdm_int *xsuper);
void sdm_mexcache_clear(void);

#ifdef __cplusplus
}
#endif
#endif /* SEDUMI_HIP_H */

// mexcommon.h -- helpers shared by the mexFunction shims.
// Each shim unpacks the mxArray arguments exactly like the reference gateway it replaces, calls the C ABI of
// libsedumi_hip.so and packs the outputs like the reference.  Compiles against MATLAB's / Octave's mex.h.
#pragma once
#include "mex.h"
#include "sedumi_hip.h"
#include <string.h>
#include <vector>

typedef std::vector<sdm_int> ivec;

inline void sdm_check(int rc) { if (rc) mexErrMsgTxt(sdm_last_error()); }
inline const mxArray *need_field(const mxArray *s, const char *name, const char *msg) {
  const mxArray *f = mxGetField(s, 0, name);
  if (!f) mexErrMsgTxt(msg);
  return f;
}
inline ivec idx_from_mw(const mwIndex *p, size_t n) { ivec v(n); for (size_t i = 0; i < n; i++) v[i] = (sdm_int)p[i]; return v; }
// MATLAB index vectors are doubles, usually 1-based: convert with an offset (e.g. -1)
inline ivec idx_from_dbl(const mxArray *a, sdm_int offset) {
  size_t n = mxGetM(a) * mxGetN(a); const double *p = mxGetPr(a);
  ivec v(n); for (size_t i = 0; i < n; i++) v[i] = (sdm_int)p[i] + offset; return v;
}
inline size_t numel(const mxArray *a) { return mxGetM(a) * mxGetN(a); }

// Index arrays of a sparse mxArray as the C ABI wants them (0-based int64): mwIndex is a 64-bit unsigned integer in MATLAB and
// in Octave builds with 64-bit indexing -- the same bits, so the array itself is handed over (the per-iteration gateways are
// called with patterns of 10^5 .. 10^7 entries: a converted copy per call costs as much as the device work); other widths copy.
struct IdxView {
  const sdm_int *p = nullptr;
  ivec own;
  IdxView() {}
  IdxView(const mwIndex *q, size_t n) {
    if (sizeof(mwIndex) == sizeof(sdm_int)) p = reinterpret_cast<const sdm_int *>(q);
    else { own = idx_from_mw(q, n); p = own.data(); }
  }
  const sdm_int *data() const { return own.empty() ? p : own.data(); }
  sdm_int operator[](size_t i) const { return data()[i]; }
};
inline IdxView jc_of(const mxArray *a) { return IdxView(mxGetJc(a), mxGetN(a) + 1); }
inline IdxView ir_of(const mxArray *a) { return IdxView(mxGetIr(a), (size_t)mxGetJc(a)[mxGetN(a)]); }
// a sparse array with the pattern of `src` and UNINITIALISED values (the caller fills them): what getada1.c:222-225 builds with
// mxCreateSparse + memcpy and getada2/3 with mxDuplicateArray, without the zero fill / value copy of nnz entries
inline mxArray *sparse_like(const mxArray *src) {
  const mwSize m = mxGetM(src), n = mxGetN(src);
  const mwIndex nnz = mxGetJc(src)[n], cap = nnz > 0 ? nnz : 1;
  mxArray *a = mxCreateSparse(m, n, 1, mxREAL);
  mwIndex *ir = (mwIndex *)mxMalloc(cap * sizeof(mwIndex));
  double *pr = (double *)mxMalloc(cap * sizeof(double));
  mxFree(mxGetIr(a)); mxFree(mxGetPr(a));
  mxSetIr(a, ir); mxSetPr(a, pr); mxSetNzmax(a, cap);
  memcpy(mxGetJc(a), mxGetJc(src), (n + 1) * sizeof(mwIndex));
  if (nnz) {                                                           // (the cache checks this pattern: copied and checksummed in one pass)
    if (sizeof(mwIndex) == 8) sdm_mexcache_copy_words(ir, mxGetIr(src), (sdm_int)nnz);
    else memcpy(ir, mxGetIr(src), nnz * sizeof(mwIndex));
  }
  if (!nnz) pr[0] = 0.0;
  return a;
}

// ---- lazy intermediates (sdm_mexcache.hip, SEDUMI_HIP_LAZY): the token of an ADA' whose values stayed on the device is an m x m
// sparse matrix with the single nonzero (1,1) = token; 0 if `a` is not one
inline double lazy_token_of(const mxArray *a) {
  if (!mxIsSparse(a) || mxGetM(a) < 2 || mxGetJc(a)[mxGetN(a)] != 1 || mxGetJc(a)[1] != 1 || mxGetIr(a)[0] != 0) return 0.0;
  const double t = mxGetPr(a)[0], b = sdm_mexcache_token_base();
  return (t > b && t < b + 4294967296.0) ? t : 0.0;
}
inline mxArray *make_token(mwSize m, double tok) {
  mxArray *a = mxCreateSparse(m, m, 1, mxREAL);
  mwIndex *jc = mxGetJc(a);
  jc[0] = 0;
  for (mwSize j = 1; j <= m; j++) jc[j] = 1;
  mxGetIr(a)[0] = 0; mxGetPr(a)[0] = tok;
  return a;
}
// the full-pattern ADA' array behind a token, values uninitialised (getada3.mex materialising at level 1)
inline mxArray *sparse_of_token(double tok, sdm_int m) {
  sdm_int nnz = 0;
  sdm_check(sdm_mexcache_token_info(tok, m, &nnz));
  const mwIndex cap = nnz > 0 ? (mwIndex)nnz : 1;
  mxArray *a = mxCreateSparse(m, m, 1, mxREAL);
  mwIndex *ir = (mwIndex *)mxMalloc(cap * sizeof(mwIndex));
  double *pr = (double *)mxMalloc(cap * sizeof(double));
  mxFree(mxGetIr(a)); mxFree(mxGetPr(a));
  mxSetIr(a, ir); mxSetPr(a, pr); mxSetNzmax(a, cap);
  if (sizeof(mwIndex) == sizeof(sdm_int)) sdm_check(sdm_mexcache_token_pattern(tok, m, (sdm_int *)mxGetJc(a), (sdm_int *)ir));
  else {
    ivec jc(m + 1), iv(cap);
    sdm_check(sdm_mexcache_token_pattern(tok, m, jc.data(), iv.data()));
    for (sdm_int j = 0; j <= m; j++) mxGetJc(a)[j] = (mwIndex)jc[j];
    for (sdm_int k = 0; k < nnz; k++) ir[k] = (mwIndex)iv[k];
  }
  return a;
}

struct SymbL {                       // L.{L,perm,xsuper} as the numeric gateways read it (blkchol.c:266-286)
  sdm_int m, nsuper;
  IdxView jc, ir;
  ivec perm, xsuper;
  const double *pr;
};
SymbL read_L(const mxArray *L, bool want_perm);

// K -> sdm_cone (conepars, sdmauxCone.c:48-134); vectors keep the storage alive
struct ConeK { sdm_cone K; ivec q, s; };
void read_cone(const mxArray *mxK, ConeK &out);

// the process-wide cache inside libsedumi_hip.so (sdm_mexcache_*) is torn down at mexAtExit (INTEGRATION.md "Keeping data on the device")
void cache_teardown_at_exit(void);

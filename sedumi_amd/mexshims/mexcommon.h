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

struct SymbL {                       // L.{L,perm,xsuper} as the numeric gateways read it (blkchol.c:266-286)
  sdm_int m, nsuper;
  ivec jc, ir, perm, xsuper;
  const double *pr;
};
SymbL read_L(const mxArray *L, bool want_perm);

// K -> sdm_cone (conepars, sdmauxCone.c:48-134); vectors keep the storage alive
struct ConeK { sdm_cone K; ivec q, s; };
void read_cone(const mxArray *mxK, ConeK &out);

// one resident plan per symbolic factor, torn down at mexAtExit (INTEGRATION.md "Keeping data on the device")
sdm_plan *cached_plan(const SymbL &L, const mwIndex *Xjc, const mwIndex *Xir);
void remember_factor(const double *Lpr_host, size_t nnz);
sdm_plan *plan_for_factor(const SymbL &L);     // non-null iff the values of L.L are the factor the last blkchol left resident

#include "mexcommon.h"
#include <stdlib.h>

SymbL read_L(const mxArray *L, bool want_perm) {
  if (!mxIsStruct(L)) mexErrMsgTxt("Parameter `L' should be a structure.");
  SymbL S;
  const mxArray *f = need_field(L, "L", "Missing field L.L.");
  if (!mxIsSparse(f)) mexErrMsgTxt("L.L should be sparse.");
  S.m = (sdm_int)mxGetM(f);
  if (S.m != (sdm_int)mxGetN(f)) mexErrMsgTxt("Size L.L mismatch.");
  S.jc = jc_of(f);
  S.ir = ir_of(f);
  S.pr = mxGetPr(f);
  if (want_perm) {
    const mxArray *p = need_field(L, "perm", "Missing field L.perm.");
    if ((sdm_int)numel(p) != S.m) mexErrMsgTxt("perm size mismatch");
    S.perm = idx_from_dbl(p, -1);
  }
  const mxArray *x = need_field(L, "xsuper", "Missing field L.xsuper.");
  S.xsuper = idx_from_dbl(x, -1);
  S.nsuper = (sdm_int)S.xsuper.size() - 1;
  if (S.nsuper > S.m) mexErrMsgTxt("Size L.xsuper mismatch.");
  return S;
}

void read_cone(const mxArray *mxK, ConeK &o) {
  if (!mxIsStruct(mxK)) mexErrMsgTxt("Parameter `K' should be a structure.");
  const mxArray *f;
  o.K.lpN = (f = mxGetField(mxK, 0, "l")) ? (sdm_int)mxGetScalar(f) : 0;
  o.q.clear(); o.s.clear();
  if ((f = mxGetField(mxK, 0, "q")) && !(numel(f) == 1 && mxGetPr(f)[0] == 0.0)) o.q = idx_from_dbl(f, 0);
  if ((f = mxGetField(mxK, 0, "s")) && !(numel(f) == 1 && mxGetPr(f)[0] == 0.0)) o.s = idx_from_dbl(f, 0);
  o.K.lorN = (sdm_int)o.q.size(); o.K.lorNL = o.q.empty() ? NULL : o.q.data();
  o.K.sdpN = (sdm_int)o.s.size(); o.K.sdpNL = o.s.empty() ? NULL : o.s.data();
  o.K.rsdpN = (f = mxGetField(mxK, 0, "rsdpN")) ? (sdm_int)mxGetScalar(f) : o.K.sdpN;
}

// ------------------------------------------------------------------ plan cache
// The cache itself lives inside libsedumi_hip.so (sdm_mexcache_*, sdm_mexcache.hip): every .mex binary is its own
// shared object, statics here would not be shared between getada3.mex, blkchol.mex and fwblkslv.mex.
namespace {
bool atexit_set = false;
void teardown(void) { sdm_mexcache_clear(); }
}  // namespace
void cache_teardown_at_exit(void) {
  if (!atexit_set) { mexAtExit(teardown); atexit_set = true; }
  sdm_mexcache_forget_notes();   // (every gateway calls this first: a checksum noted by a call that ended in an error before it reached the cache dies here)
}

#include "mexcommon.h"
#include <stdlib.h>

SymbL read_L(const mxArray *L, bool want_perm) {
  if (!mxIsStruct(L)) mexErrMsgTxt("Parameter `L' should be a structure.");
  SymbL S;
  const mxArray *f = need_field(L, "L", "Missing field L.L.");
  if (!mxIsSparse(f)) mexErrMsgTxt("L.L should be sparse.");
  S.m = (sdm_int)mxGetM(f);
  if (S.m != (sdm_int)mxGetN(f)) mexErrMsgTxt("Size L.L mismatch.");
  S.jc = idx_from_mw(mxGetJc(f), S.m + 1);
  S.ir = idx_from_mw(mxGetIr(f), (size_t)S.jc[S.m]);
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
namespace {
struct Cache {
  sdm_plan *plan = NULL;
  sdm_int m = -1, nnzL = -1, nsuper = -1, nnzX = -1;
  unsigned long long hperm = 0, hxs = 0;
  const double *last_Lpr = NULL;
  size_t last_nnz = 0;
  bool atexit_set = false;
} g;
unsigned long long hash(const ivec &v) {
  unsigned long long h = 1469598103934665603ull;
  for (size_t i = 0; i < v.size(); i++) { h ^= (unsigned long long)v[i]; h *= 1099511628211ull; }
  return h;
}
void teardown(void) { if (g.plan) sdm_plan_destroy(g.plan); g.plan = NULL; g.last_Lpr = NULL; }
}  // namespace

sdm_plan *cached_plan(const SymbL &L, const mwIndex *Xjc, const mwIndex *Xir) {
  const sdm_int nnzX = (sdm_int)Xjc[L.m];
  const unsigned long long hp = hash(L.perm), hx = hash(L.xsuper);
  if (g.plan && g.m == L.m && g.nnzL == L.jc[L.m] && g.nsuper == L.nsuper && g.nnzX == nnzX && g.hperm == hp && g.hxs == hx)
    return g.plan;
  teardown();
  const char *dev = getenv("SEDUMI_HIP_DEVICE");
  g.plan = sdm_plan_create(dev ? atoi(dev) : 0, NULL);
  if (!g.plan) mexErrMsgTxt(sdm_last_error());
  if (!g.atexit_set) { mexAtExit(teardown); g.atexit_set = true; }
  ivec xjc = idx_from_mw(Xjc, L.m + 1), xir = idx_from_mw(Xir, (size_t)nnzX);
  if (sdm_plan_set_chol(g.plan, L.m, L.jc.data(), L.ir.data(), L.perm.data(), L.nsuper, L.xsuper.data(), xjc.data(), xir.data())) {
    teardown(); mexErrMsgTxt(sdm_last_error());
  }
  g.m = L.m; g.nnzL = L.jc[L.m]; g.nsuper = L.nsuper; g.nnzX = nnzX; g.hperm = hp; g.hxs = hx;
  return g.plan;
}
void remember_factor(const double *Lpr_host, size_t nnz) { g.last_Lpr = Lpr_host; g.last_nnz = nnz; }
sdm_plan *plan_for_factor(const SymbL &L) {
  if (g.plan && g.last_Lpr == L.pr && g.last_nnz == (size_t)L.jc[L.m] && g.m == L.m && g.hxs == hash(L.xsuper)) return g.plan;
  return NULL;
}

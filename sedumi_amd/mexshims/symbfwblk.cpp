// x = symbfwblk(L,b)  -- replaces the symbfwblk.c gateway (symbfwblk.c:270-377)
#include "mexcommon.h"
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
  if (nrhs < 2) mexErrMsgTxt("symbfwblk requires more input arguments");
  if (!mxIsSparse(prhs[1])) mexErrMsgTxt("B must be sparse");
  SymbL L = read_L(prhs[0], true);
  const mxArray *B = prhs[1];
  if ((sdm_int)mxGetM(B) != L.m) mexErrMsgTxt("L.perm size mismatches B");
  const sdm_int m = L.m, n = (sdm_int)mxGetN(B);
  ivec Bjc = idx_from_mw(mxGetJc(B), n + 1), Bir = idx_from_mw(mxGetIr(B), mxGetJc(B)[n]);
  ivec Xjc(n + 1);
  sdm_check(sdm_symbfwblk(m, L.jc.data(), L.ir.data(), L.perm.data(), L.nsuper, L.xsuper.data(), n, Bjc.data(), Bir.data(), Xjc.data(), NULL));
  const sdm_int nnz = Xjc[n];
  ivec Xir(nnz > 0 ? nnz : 1);
  sdm_check(sdm_symbfwblk(m, L.jc.data(), L.ir.data(), L.perm.data(), L.nsuper, L.xsuper.data(), n, Bjc.data(), Bir.data(), Xjc.data(), Xir.data()));
  plhs[0] = mxCreateSparse(m, n, nnz > 0 ? nnz : 1, mxREAL);
  for (sdm_int j = 0; j <= n; j++) mxGetJc(plhs[0])[j] = (mwIndex)Xjc[j];
  for (sdm_int t = 0; t < nnz; t++) { mxGetIr(plhs[0])[t] = (mwIndex)Xir[t]; mxGetPr(plhs[0])[t] = 1.0; }
}

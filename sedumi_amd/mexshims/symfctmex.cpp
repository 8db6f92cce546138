// L = symfctmex(X,perm) -> L.{L,perm,xsuper}  -- replaces symfctmex.c:127-272
#include "mexcommon.h"
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
  if (nrhs < 2) mexErrMsgTxt("symfctmex requires more input arguments");
  const mxArray *X = prhs[0];
  if (!mxIsSparse(X)) mexErrMsgTxt("X must be sparse");
  const sdm_int m = (sdm_int)mxGetM(X);
  if (m != (sdm_int)mxGetN(X)) mexErrMsgTxt("X must be square");
  if ((sdm_int)numel(prhs[1]) != m) mexErrMsgTxt("perm size mismatch");
  ivec jc = idx_from_mw(mxGetJc(X), m + 1), ir = idx_from_mw(mxGetIr(X), mxGetJc(X)[m]), pin = idx_from_dbl(prhs[1], -1);
  sdm_int nsuper = 0, nnzl = 0;
  sdm_check(sdm_symfct(m, jc.data(), ir.data(), pin.data(), NULL, &nsuper, NULL, &nnzl, NULL, NULL));
  ivec pout(m > 0 ? m : 1), xs(nsuper + 1), Ljc(m + 1), Lir(nnzl > 0 ? nnzl : 1);
  sdm_check(sdm_symfct(m, jc.data(), ir.data(), pin.data(), pout.data(), &nsuper, xs.data(), &nnzl, Ljc.data(), Lir.data()));
  const char *names[] = {"L", "perm", "xsuper"};
  plhs[0] = mxCreateStructMatrix(1, 1, 3, names);
  mxArray *LL = mxCreateSparse(m, m, nnzl, mxREAL);
  for (sdm_int j = 0; j <= m; j++) mxGetJc(LL)[j] = (mwIndex)Ljc[j];
  for (sdm_int t = 0; t < nnzl; t++) { mxGetIr(LL)[t] = (mwIndex)Lir[t]; mxGetPr(LL)[t] = 1.0; }
  mxArray *P = mxCreateDoubleMatrix(m, 1, mxREAL), *XS = mxCreateDoubleMatrix(nsuper + 1, 1, mxREAL);
  for (sdm_int i = 0; i < m; i++) mxGetPr(P)[i] = (double)(pout[i] + 1);
  for (sdm_int i = 0; i <= nsuper; i++) mxGetPr(XS)[i] = (double)(xs[i] + 1);
  mxSetField(plhs[0], 0, "L", LL); mxSetField(plhs[0], 0, "perm", P); mxSetField(plhs[0], 0, "xsuper", XS);
}

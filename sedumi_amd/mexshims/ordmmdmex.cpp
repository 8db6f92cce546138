// perm = ordmmdmex(X)  -- replaces ordmmdmex.c:75-139
#include "mexcommon.h"
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
  if (nrhs != 1) mexErrMsgTxt("ordmmd requires 1 input argument.");
  const mxArray *X = prhs[0];
  if (!mxIsSparse(X)) mexErrMsgTxt("Input matrix must be sparse");
  const sdm_int m = (sdm_int)mxGetM(X);
  if (m != (sdm_int)mxGetN(X)) mexErrMsgTxt("X should be square.");
  ivec jc = idx_from_mw(mxGetJc(X), m + 1), ir = idx_from_mw(mxGetIr(X), mxGetJc(X)[m]), perm(m > 0 ? m : 1);
  sdm_check(sdm_ordmmd(m, jc.data(), ir.data(), perm.data()));
  plhs[0] = mxCreateDoubleMatrix(m, 1, mxREAL);
  for (sdm_int i = 0; i < m; i++) mxGetPr(plhs[0])[i] = (double)(perm[i] + 1);
}

// split = cholsplit(L,cachsz)  -- replaces cholsplit.c:118-184
#include "mexcommon.h"
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
  if (nrhs < 2) mexErrMsgTxt("cholsplit requires more input arguments");
  SymbL L = read_L(prhs[0], false);
  ivec split(L.m > 0 ? L.m : 1);
  sdm_check(sdm_cholsplit(L.m, L.jc.data(), L.nsuper, L.xsuper.data(), mxGetScalar(prhs[1]), split.data()));
  plhs[0] = mxCreateDoubleMatrix(L.m, 1, mxREAL);
  for (sdm_int i = 0; i < L.m; i++) mxGetPr(plhs[0])[i] = (double)split[i];
}

// tmpsiz = choltmpsiz(L)  -- replaces choltmpsiz.c:110-173
#include "mexcommon.h"
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
  if (nrhs < 1) mexErrMsgTxt("choltmpsiz requires more input arguments");
  SymbL L = read_L(prhs[0], false);
  sdm_int t = 0;
  sdm_check(sdm_choltmpsiz(L.m, L.jc.data(), L.ir.data(), L.nsuper, L.xsuper.data(), &t));
  plhs[0] = mxCreateDoubleMatrix(1, 1, mxREAL);
  mxGetPr(plhs[0])[0] = (double)t;
}

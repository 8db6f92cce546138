// y = invcholfac(u,K,perm)  -- replaces invcholfac.c:59-168 (the udsqr argument of getada3, sedumi.m:452)
#include "mexcommon.h"
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
  if (nrhs < 2) mexErrMsgTxt("invcholfac requires at least 2 input arguments.");
  ConeK ck; read_cone(prhs[1], ck);
  sdm_int lenud = 0, plen = 0;
  for (sdm_int k = 0; k < ck.K.sdpN; k++) { lenud += (k < ck.K.rsdpN ? 1 : 2) * ck.K.sdpNL[k] * ck.K.sdpNL[k]; plen += ck.K.sdpNL[k]; }
  if ((sdm_int)numel(prhs[0]) != lenud) mexErrMsgTxt("u size mismatch");
  const bool isperm = nrhs >= 3 && numel(prhs[2]) > 0;             // invcholfac.c:79-82
  ivec perm;
  if (isperm) {
    if ((sdm_int)numel(prhs[2]) != plen) mexErrMsgTxt("perm size mismatch");
    perm = idx_from_dbl(prhs[2], -1);
  }
  plhs[0] = mxCreateDoubleMatrix(lenud, 1, mxREAL);
  sdm_check(sdm_invcholfac(&ck.K, mxGetPr(prhs[0]), isperm ? perm.data() : nullptr, mxGetPr(plhs[0])));
}

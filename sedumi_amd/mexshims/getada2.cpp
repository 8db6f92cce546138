// ADA = getada2(ADA,DAt,Aord,K)  -- replaces getada2.c:127-214
#include "mexcommon.h"
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
  if (nrhs < 4) mexErrMsgTxt("getADA requires more input arguments.");
  const mxArray *ADA = prhs[0];
  if (!mxIsSparse(ADA)) mexErrMsgTxt("ADA should be sparse.");
  plhs[0] = mxDuplicateArray(ADA);                                  // getada2.c:153
  ConeK ck; read_cone(prhs[3], ck);
  if (ck.K.lorN == 0) return;                                       // getada2.c:154-155
  const sdm_int m = (sdm_int)mxGetM(ADA);
  const mxArray *Q = need_field(prhs[1], "q", "Missing field DAt.q.");
  if (!mxIsSparse(Q) || (sdm_int)mxGetM(Q) != ck.K.lorN || (sdm_int)mxGetN(Q) != m) mexErrMsgTxt("Size mismatch DAt.q.");
  const mxArray *qp = need_field(prhs[2], "qperm", "Missing field Aord.qperm.");
  if ((sdm_int)numel(qp) != m) mexErrMsgTxt("Aord.qperm size mismatch");
  ivec jc = idx_from_mw(mxGetJc(ADA), m + 1), ir = idx_from_mw(mxGetIr(ADA), mxGetJc(ADA)[m]);
  ivec Qjc = idx_from_mw(mxGetJc(Q), m + 1), Qir = idx_from_mw(mxGetIr(Q), mxGetJc(Q)[m]), qperm = idx_from_dbl(qp, -1);
  sdm_check(sdm_getada2(m, jc.data(), ir.data(), mxGetPr(plhs[0]), ck.K.lorN, Qjc.data(), Qir.data(), mxGetPr(Q), qperm.data()));
}

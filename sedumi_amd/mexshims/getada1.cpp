// ADA = getada1(ADA,A,Ajc2,perm,d,blkstart)  -- replaces getada1.c:161-261
#include "mexcommon.h"
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
  if (nrhs < 6) mexErrMsgTxt("getADA requires more input arguments.");
  const mxArray *ADA = prhs[0], *A = prhs[1];
  if (!mxIsSparse(A)) mexErrMsgTxt("At should be sparse.");
  if (!mxIsSparse(ADA)) mexErrMsgTxt("ADA should be sparse.");
  const sdm_int m = (sdm_int)mxGetN(A);
  if ((sdm_int)mxGetM(ADA) != m || (sdm_int)mxGetN(ADA) != m) mexErrMsgTxt("Size mismatch ADA.");
  if ((sdm_int)numel(prhs[2]) != m) mexErrMsgTxt("Size mismatch Ajc2.");
  if ((sdm_int)numel(prhs[3]) != m) mexErrMsgTxt("Size mismatch perm.");
  if (!mxIsStruct(prhs[4])) mexErrMsgTxt("Parameter `d' should be a structure.");
  const mxArray *dl = need_field(prhs[4], "l", "Field d.l missing."), *ddet = need_field(prhs[4], "det", "Field d.det missing.");
  ivec qb = idx_from_dbl(prhs[5], -1);                              // K.qblkstart
  const sdm_int lorN = (sdm_int)qb.size() - 1;
  if ((sdm_int)numel(ddet) != lorN) mexErrMsgTxt("Size d.det mismatch");
  IdxView Ajc = jc_of(A), Air = ir_of(A);
  ivec Ajc2 = idx_from_dbl(prhs[2], 0), perm = idx_from_dbl(prhs[3], -1);
  cache_teardown_at_exit();
  // lazy intermediates (SEDUMI_HIP_LAZY >= 1): the result is a token, the values stay on the device; at level 2 the input is getada3's token
  const double tin = lazy_token_of(ADA);
  const bool lazy = sdm_mexcache_lazy() >= 1 && m >= 2;           // (a 1 x 1 token would be indistinguishable from a genuine 1 x 1 ADA')
  IdxView jc, ir;
  if (tin == 0.0) { jc = jc_of(ADA); ir = ir_of(ADA); }
  double tout = 0.0;
  if (!lazy) plhs[0] = tin == 0.0 ? sparse_like(ADA) : sparse_of_token(tin, m);   // getada1.c:222-225
  sdm_check(sdm_mexcache_getada1(m, tin == 0.0 ? jc.data() : NULL, tin == 0.0 ? ir.data() : NULL, (sdm_int)mxGetM(A), Ajc.data(), Air.data(), mxGetPr(A),
                                 Ajc2.data(), perm.data(), (sdm_int)numel(dl), mxGetPr(dl), lorN, mxGetPr(ddet), qb.data(), lazy ? NULL : mxGetPr(plhs[0]),
                                 tin, lazy ? &tout : NULL));
  if (lazy) plhs[0] = make_token(m, tout);
}

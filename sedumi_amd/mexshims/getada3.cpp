// [ADA,absd] = getada3(ADA,A,Ajc1,Aord,udsqr,K)  -- replaces getada3.c:370-569
#include "mexcommon.h"
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
  if (nrhs < 6) mexErrMsgTxt("getADA requires more input arguments.");
  const mxArray *ADA = prhs[0], *A = prhs[1];
  if (!mxIsSparse(A)) mexErrMsgTxt("At should be sparse.");
  if (!mxIsSparse(ADA)) mexErrMsgTxt("ADA should be sparse.");
  ConeK ck; read_cone(prhs[5], ck);
  const sdm_int m = (sdm_int)mxGetN(A);
  if ((sdm_int)mxGetM(ADA) != m || (sdm_int)mxGetN(ADA) != m) mexErrMsgTxt("Size mismatch ADA.");
  if ((sdm_int)numel(prhs[2]) != m) mexErrMsgTxt("Ajc1 size mismatch");
  const mxArray *bs = need_field(prhs[5], "blkstart", "Missing K.blkstart.");
  if ((sdm_int)numel(bs) != 2 + ck.K.lorN + ck.K.sdpN) mexErrMsgTxt("Size mismatch K.blkstart.");
  ivec blk = idx_from_dbl(bs, -1);
  ivec psd(blk.begin() + ck.K.lorN + 1, blk.end());
  const mxArray *sp = need_field(prhs[3], "sperm", "Missing field Aord.sperm.");
  if ((sdm_int)numel(sp) != m) mexErrMsgTxt("Aord.sperm size mismatch");      // (only its triangular bookkeeping exists in the reference: the sum is order independent)
  IdxView Ajc = jc_of(A), Air = ir_of(A);
  ivec Ajc1 = idx_from_dbl(prhs[2], 0);
  cache_teardown_at_exit();
  // lazy intermediates: the input may be a token (level >= 1); at level 2 the result is one, too (absd is always real)
  const double tin = lazy_token_of(ADA);
  const bool lazy_out = sdm_mexcache_lazy() >= 2 && m >= 2;       // (a 1 x 1 token would be indistinguishable from a genuine 1 x 1 ADA')
  IdxView jc, ir;
  if (tin == 0.0) { jc = jc_of(ADA); ir = ir_of(ADA); }
  double tout = 0.0;
  mxArray *out0 = lazy_out ? NULL : (tin == 0.0 ? sparse_like(ADA) : sparse_of_token(tin, m));   // getada3.c:452 (the values come back from the device)
  mxArray *out1 = mxCreateDoubleMatrix(m, 1, mxREAL);
  sdm_check(sdm_mexcache_getada3(m, tin == 0.0 ? jc.data() : NULL, tin == 0.0 ? ir.data() : NULL, tin == 0.0 ? mxGetPr(ADA) : NULL, lazy_out ? NULL : mxGetPr(out0),
                                 (sdm_int)mxGetM(A), Ajc.data(), Air.data(), mxGetPr(A), Ajc1.data(), mxGetPr(prhs[4]), &ck.K, psd.data(), mxGetPr(out1),
                                 tin, lazy_out ? &tout : NULL));
  if (lazy_out) out0 = make_token(m, tout);
  plhs[0] = out0;
  if (nlhs > 1) plhs[1] = out1; else mxDestroyArray(out1);           // getada3.c:565-568
}

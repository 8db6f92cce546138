// ADA = getada2(ADA,DAt,Aord,K)  -- replaces getada2.c:127-214
#include "mexcommon.h"
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
  if (nrhs < 4) mexErrMsgTxt("getADA requires more input arguments.");
  const mxArray *ADA = prhs[0];
  if (!mxIsSparse(ADA)) mexErrMsgTxt("ADA should be sparse.");
  ConeK ck; read_cone(prhs[3], ck);
  const sdm_int m = (sdm_int)mxGetM(ADA);
  cache_teardown_at_exit();
  if (ck.K.lorN == 0) {                                              // getada2.c:153-155: the copy, nothing added
    plhs[0] = mxDuplicateArray(ADA);                                // (the same content: getada3 recognises it by that)
    return;
  }
  const mxArray *Q = need_field(prhs[1], "q", "Missing field DAt.q.");
  if (!mxIsSparse(Q) || (sdm_int)mxGetM(Q) != ck.K.lorN || (sdm_int)mxGetN(Q) != m) mexErrMsgTxt("Size mismatch DAt.q.");
  const mxArray *qp = need_field(prhs[2], "qperm", "Missing field Aord.qperm.");
  if ((sdm_int)numel(qp) != m) mexErrMsgTxt("Aord.qperm size mismatch");
  IdxView Qjc = jc_of(Q), Qir = ir_of(Q);
  ivec qperm = idx_from_dbl(qp, -1);
  const double tin = lazy_token_of(ADA);                             // (lazy intermediates: token in -> token out)
  IdxView jc, ir;
  if (tin == 0.0) { jc = jc_of(ADA); ir = ir_of(ADA); }
  const bool lazy = sdm_mexcache_lazy() >= 1 && m >= 2;           // (a 1 x 1 token would be indistinguishable from a genuine 1 x 1 ADA')
  double tout = 0.0;
  if (!lazy) plhs[0] = tin == 0.0 ? sparse_like(ADA) : sparse_of_token(tin, m);   // getada2.c:153 (the values come back from the device)
  sdm_check(sdm_mexcache_getada2(m, tin == 0.0 ? jc.data() : NULL, tin == 0.0 ? ir.data() : NULL, tin == 0.0 ? mxGetPr(ADA) : NULL, lazy ? NULL : mxGetPr(plhs[0]),
                                 ck.K.lorN, Qjc.data(), Qir.data(), mxGetPr(Q), qperm.data(), tin, lazy ? &tout : NULL));
  if (lazy) plhs[0] = make_token(m, tout);
}

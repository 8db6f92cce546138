// absd = getada(A,K,d,DAt)  -- shadows getada.m:1-40 (sedumi.m:446-448 sends every problem with sum(K.s)==0 here), so
// that LP / SOCP problems reach the HIP path under an unmodified sedumi.m.  Like the .m file it reads the pattern of
// the GLOBAL ADA_sedumi_ and writes the new ADA' back into that global; absd = diag(ADA').
#include "mexcommon.h"
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
  if (nrhs < 4) mexErrMsgTxt("getada requires more input arguments.");
  if (nlhs > 1) mexErrMsgTxt("getada produces less output arguments.");
  const mxArray *G = mexGetVariablePtr("global", "ADA_sedumi_");
  if (!G) mexErrMsgTxt("global ADA_sedumi_ does not exist.");
  if (!mxIsSparse(G)) mexErrMsgTxt("ADA_sedumi_ should be sparse.");
  const mxArray *A = prhs[0];
  if (!mxIsSparse(A)) mexErrMsgTxt("A should be sparse.");
  const sdm_int m = (sdm_int)mxGetN(A), N = (sdm_int)mxGetM(A);
  if ((sdm_int)mxGetM(G) != m || (sdm_int)mxGetN(G) != m) mexErrMsgTxt("Size mismatch ADA_sedumi_.");
  ConeK ck; read_cone(prhs[1], ck);
  if (ck.K.sdpN > 0) {
    sdm_int tot = 0;
    for (sdm_int k = 0; k < ck.K.sdpN; k++) tot += ck.s[k];
    if (tot > 0) mexErrMsgTxt("getada is the path of problems without PSD blocks (sedumi.m:446).");
  }
  const mxArray *dl = need_field(prhs[2], "l", "Missing field d.l."), *ddet = need_field(prhs[2], "det", "Missing field d.det.");
  if ((sdm_int)numel(dl) != ck.K.lpN || (sdm_int)numel(ddet) != ck.K.lorN) mexErrMsgTxt("Size mismatch d.l / d.det.");
  ivec qb(1, 0), Qjc_own(m + 1, 0), Qir_own(1, 0);
  std::vector<double> Qpr(1, 0.0);
  const double *qpr = Qpr.data();
  IdxView Qjc_v, Qir_v;
  const sdm_int *Qjc = Qjc_own.data(), *Qir = Qir_own.data();
  if (ck.K.lorN > 0) {
    qb = idx_from_dbl(need_field(prhs[1], "qblkstart", "Missing field K.qblkstart."), -1);
    if ((sdm_int)qb.size() != ck.K.lorN + 1) mexErrMsgTxt("Size mismatch K.qblkstart.");
    const mxArray *Q = need_field(prhs[3], "q", "Missing field DAt.q.");
    if ((sdm_int)mxGetM(Q) != ck.K.lorN || (sdm_int)mxGetN(Q) != m) mexErrMsgTxt("Size mismatch DAt.q.");
    if (mxIsSparse(Q)) {
      Qjc_v = jc_of(Q); Qir_v = ir_of(Q);
      Qjc = Qjc_v.data(); Qir = Qir_v.data();
      qpr = mxGetPr(Q);
    } else {                                                         // a full DAt.q: every entry
      const sdm_int nq = ck.K.lorN;
      Qir_own.resize((size_t)(nq * m));
      for (sdm_int j = 0; j <= m; j++) Qjc_own[j] = j * nq;
      for (sdm_int t = 0; t < nq * m; t++) Qir_own[t] = t % nq;
      Qjc = Qjc_own.data(); Qir = Qir_own.data();
      qpr = mxGetPr(Q);
    }
  }
  cache_teardown_at_exit();
  mxArray *out = sparse_like(G);                                     // same pattern as the global (getsymbada.m covers every product)
  plhs[0] = mxCreateDoubleMatrix(m, 1, mxREAL);
  IdxView jc = jc_of(G), ir = ir_of(G), Ajc = jc_of(A), Air = ir_of(A);
  int rc = sdm_mexcache_getada(m, jc.data(), ir.data(), N, Ajc.data(), Air.data(), mxGetPr(A), ck.K.lpN, mxGetPr(dl), ck.K.lorN, mxGetPr(ddet),
                               qb.data(), Qjc, Qir, qpr, mxGetPr(out), mxGetPr(plhs[0]));
  if (rc) { mxDestroyArray(out); mexErrMsgTxt(sdm_last_error()); }
  if (mexPutVariable("global", "ADA_sedumi_", out)) { mxDestroyArray(out); mexErrMsgTxt("could not update global ADA_sedumi_."); }
  mxDestroyArray(out);                                               // mexPutVariable stored a copy
}

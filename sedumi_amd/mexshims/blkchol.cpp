// [L.L,L.d,L.skip,L.add] = blkchol(L,X,pars,absd)  -- replaces blkchol.c:239-440
#include "mexcommon.h"
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
  if (nrhs < 2) mexErrMsgTxt("blkchol requires more input arguments");
  if (nlhs > 4) mexErrMsgTxt("blkchol produces less output arguments");
  const mxArray *X = prhs[1];
  if (!mxIsSparse(X)) mexErrMsgTxt("P must be sparse");
  SymbL L = read_L(prhs[0], true);
  if ((sdm_int)mxGetM(X) != L.m || (sdm_int)mxGetN(X) != L.m) mexErrMsgTxt("P must be square");
  sdm_cholpars pars = {1e-12, 5e2, 1e-20};                           // blkchol.c:292-294
  const double *absd = NULL;
  if (nrhs >= 3) {
    const mxArray *P = prhs[2], *f;
    if (!mxIsStruct(P)) mexErrMsgTxt("Parameter `pars' should be a structure.");
    if ((f = mxGetField(P, 0, "canceltol"))) pars.canceltol = mxGetScalar(f);
    if ((f = mxGetField(P, 0, "maxu"))) pars.maxu = mxGetScalar(f);
    if ((f = mxGetField(P, 0, "abstol"))) { pars.abstol = mxGetScalar(f); if (pars.abstol < 0) pars.abstol = 0; }
    if (nrhs >= 4) { if ((sdm_int)numel(prhs[3]) != L.m) mexErrMsgTxt("absd size mismatch"); absd = mxGetPr(prhs[3]); }
  }
  const sdm_int m = L.m, nnzL = L.jc[m];
  mxArray *out[4];
  out[0] = mxCreateSparse(m, m, nnzL, mxREAL);
  memcpy(mxGetJc(out[0]), mxGetJc(mxGetField(prhs[0], 0, "L")), (m + 1) * sizeof(mwIndex));
  memcpy(mxGetIr(out[0]), mxGetIr(mxGetField(prhs[0], 0, "L")), nnzL * sizeof(mwIndex));
  out[1] = mxCreateDoubleMatrix(m, 1, mxREAL);
  sdm_plan *p = cached_plan(L, mxGetJc(X), mxGetIr(X));
  remember_factor(NULL, 0);          // the resident factor is about to be overwritten: whatever the solves are handed before this call has returned is not it
  sdm_check(sdm_plan_upload(p, "ada", mxGetPr(X), (sdm_int)mxGetJc(X)[m]));
  if (absd) sdm_check(sdm_plan_upload(p, "absd", absd, m));
  sdm_check(sdm_plan_blkchol_wait(p, &pars, absd ? 1 : 0));      // (waited for, repeated once on the launch-per-panel path after a time-out)
  sdm_check(sdm_plan_download(p, "lpr", mxGetPr(out[0]), nnzL));
  sdm_check(sdm_plan_download(p, "d", mxGetPr(out[1]), m));
  ivec sidx(m > 0 ? m : 1), aidx(m > 0 ? m : 1);
  std::vector<double> sval(m > 0 ? m : 1), aval(m > 0 ? m : 1);
  sdm_int ns = 0, na = 0;
  sdm_check(sdm_plan_pivots(p, &ns, sidx.data(), sval.data(), &na, aidx.data(), aval.data()));
  for (int k = 0; k < 2; k++) {                                       // sparse m x 1 outputs (blkchol.c:396-421)
    const sdm_int n = k ? na : ns;
    out[2 + k] = mxCreateSparse(m, 1, n > 0 ? n : 1, mxREAL);
    mwIndex *jc = mxGetJc(out[2 + k]), *ir = mxGetIr(out[2 + k]); double *pr = mxGetPr(out[2 + k]);
    jc[0] = 0; jc[1] = (mwIndex)n;
    for (sdm_int i = 0; i < n; i++) { ir[i] = (mwIndex)(k ? aidx[i] : sidx[i]); pr[i] = k ? aval[i] : sval[i]; }
  }
  remember_factor(mxGetPr(out[0]), (size_t)nnzL);
  int keep = nlhs > 1 ? nlhs : 1;
  for (int i = 0; i < keep; i++) plhs[i] = out[i];
  for (int i = keep; i < 4; i++) mxDestroyArray(out[i]);              // unrequested outputs (blkchol.c:436-439)
}

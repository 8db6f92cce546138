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
    // pars.delay = 1 (blkchol.c:307-308, 402-414: skipped columns handed back as they were before their pivot, for a pivot-delaying
    // caller; no .m file of SeDuMi sets it) is not implemented: said loudly instead of returning unit columns under that name
    if ((f = mxGetField(P, 0, "delay")) && (char)mxGetScalar(f) == 1) mexErrMsgTxt("blkchol: pars.delay = 1 is not supported by libsedumi_hip (skipped columns are returned as unit vectors, pars.delay = 0).");
    if (nrhs >= 4) { if ((sdm_int)numel(prhs[3]) != L.m) mexErrMsgTxt("absd size mismatch"); absd = mxGetPr(prhs[3]); }
  }
  const sdm_int m = L.m, nnzL = L.jc[m];
  const mxArray *LLin = mxGetField(prhs[0], 0, "L");
  mxArray *out[4];
  cache_teardown_at_exit();
  out[0] = sparse_like(LLin);                                          // L.L keeps the symbolic pattern (blkchol.c:391-395)
  out[1] = mxCreateDoubleMatrix(m, 1, mxREAL);
  ivec sidx(m > 0 ? m : 1), aidx(m > 0 ? m : 1);
  std::vector<double> sval(m > 0 ? m : 1), aval(m > 0 ? m : 1);
  sdm_int ns = 0, na = 0;
  const double tin = lazy_token_of(X);                               // lazy intermediates, level 2: X is getada3's token, ADA' is on the device
  IdxView Xjc, Xir;
  if (tin == 0.0) { Xjc = jc_of(X); Xir = ir_of(X); }
  // the factor stays resident in the library's cache for fwblkslv / bwblkslv; X is taken from the device when it is the
  // array getada3 just returned (sdm_mexcache.hip)
  sdm_check(sdm_mexcache_blkchol(m, L.jc.data(), L.ir.data(), L.perm.data(), L.nsuper, L.xsuper.data(), tin == 0.0 ? Xjc.data() : NULL,
                                 tin == 0.0 ? Xir.data() : NULL, tin == 0.0 ? mxGetPr(X) : NULL, &pars, absd, mxGetPr(out[0]), mxGetPr(out[1]), &ns,
                                 sidx.data(), sval.data(), &na, aidx.data(), aval.data(), tin));
  (void)nnzL;
  for (int k = 0; k < 2; k++) {                                       // sparse m x 1 outputs (blkchol.c:396-421)
    const sdm_int n = k ? na : ns;
    out[2 + k] = mxCreateSparse(m, 1, n > 0 ? n : 1, mxREAL);
    mwIndex *jc = mxGetJc(out[2 + k]), *ir = mxGetIr(out[2 + k]); double *pr = mxGetPr(out[2 + k]);
    jc[0] = 0; jc[1] = (mwIndex)n;
    for (sdm_int i = 0; i < n; i++) { ir[i] = (mwIndex)(k ? aidx[i] : sidx[i]); pr[i] = k ? aval[i] : sval[i]; }
  }
  int keep = nlhs > 1 ? nlhs : 1;
  for (int i = 0; i < keep; i++) plhs[i] = out[i];
  for (int i = keep; i < 4; i++) mxDestroyArray(out[i]);              // unrequested outputs (blkchol.c:436-439)
}

// [Lden,Ld] = dpr1fact(x,d,Lsymb,smult,maxu)  -- replaces the dpr1fact.c gateway (dpr1fact.c:630-848)
#include "mexcommon.h"
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
  if (nrhs < 5) mexErrMsgTxt("dpr1fact requires more input arguments");
  const mxArray *X = prhs[0];
  if (!mxIsSparse(X)) mexErrMsgTxt("x should be sparse.");
  const sdm_int m = (sdm_int)mxGetM(X), n = (sdm_int)mxGetN(X);
  if ((sdm_int)numel(prhs[1]) != m) mexErrMsgTxt("Size mismatch d.");
  if ((sdm_int)numel(prhs[3]) != n) mexErrMsgTxt("Size mismatch smult.");
  if (!mxIsStruct(prhs[2])) mexErrMsgTxt("Lsymb should be a structure.");
  const mxArray *DZ = need_field(prhs[2], "dz", "Missing field Lsymb.dz.");
  if (!mxIsSparse(DZ)) mexErrMsgTxt("Lsymb.dz must be sparse.");
  if ((sdm_int)mxGetM(DZ) != m || (sdm_int)mxGetN(DZ) != n) mexErrMsgTxt("Lsymb.dz size mismatch.");
  ivec colperm = idx_from_dbl(need_field(prhs[2], "perm", "Missing field Lsymb.perm."), -1);
  ivec first = idx_from_dbl(need_field(prhs[2], "first", "Missing field Lsymb.first."), -1);
  if ((sdm_int)colperm.size() != n) mexErrMsgTxt("Size mismatch Lsymb.perm.");
  if ((sdm_int)first.size() != n) mexErrMsgTxt("Size mismatch Lsymb.first.");
  ivec Xjc = idx_from_mw(mxGetJc(X), n + 1), Xir = idx_from_mw(mxGetIr(X), mxGetJc(X)[n]);
  ivec dzjc = idx_from_mw(mxGetJc(DZ), n + 1), dzir = idx_from_mw(mxGetIr(DZ), mxGetJc(DZ)[n]);
  sdm_int pnnz = 0;
  for (sdm_int i = 1; i <= n; i++) pnnz += dzjc[i];
  mxArray *out[2];
  out[1] = mxDuplicateArray(prhs[1]);                               // Ld = copy of d, updated in place
  std::vector<double> beta(pnnz > 0 ? pnnz : 1), p(pnnz > 0 ? pnnz : 1);
  ivec betajc(n + 1), pivperm(pnnz > 0 ? pnnz : 1), dopiv(n > 0 ? n : 1);
  sdm_int npp = 0;
  sdm_check(sdm_dpr1fact(m, n, Xjc.data(), Xir.data(), mxGetPr(X), mxGetPr(out[1]), dzjc.data(), dzir.data(), colperm.data(), first.data(),
                         mxGetPr(prhs[3]), mxGetScalar(prhs[4]), betajc.data(), beta.data(), p.data(), pivperm.data(), &npp, dopiv.data()));
  const char *names[] = {"betajc", "beta", "p", "pivperm", "dopiv"};
  out[0] = mxCreateStructMatrix(1, 1, 5, names);
  mxArray *f = mxCreateDoubleMatrix(n + 1, 1, mxREAL);
  for (sdm_int i = 0; i <= n; i++) mxGetPr(f)[i] = (double)betajc[i] + 1.0;
  mxSetField(out[0], 0, "betajc", f);
  f = mxCreateDoubleMatrix(betajc[n], 1, mxREAL);
  if (betajc[n]) memcpy(mxGetPr(f), beta.data(), betajc[n] * sizeof(double));
  mxSetField(out[0], 0, "beta", f);
  f = mxCreateDoubleMatrix(pnnz, 1, mxREAL);
  if (pnnz) memcpy(mxGetPr(f), p.data(), pnnz * sizeof(double));
  mxSetField(out[0], 0, "p", f);
  f = mxCreateDoubleMatrix(npp, 1, mxREAL);
  for (sdm_int i = 0; i < npp; i++) mxGetPr(f)[i] = (double)pivperm[i];     // C-form (0-based), as the reference
  mxSetField(out[0], 0, "pivperm", f);
  f = mxCreateDoubleMatrix(n, 1, mxREAL);
  for (sdm_int i = 0; i < n; i++) mxGetPr(f)[i] = (double)dopiv[i];
  mxSetField(out[0], 0, "dopiv", f);
  const int want = nlhs > 1 ? nlhs : 1;
  for (int i = 0; i < 2; i++) { if (i < want) plhs[i] = out[i]; else mxDestroyArray(out[i]); }
}

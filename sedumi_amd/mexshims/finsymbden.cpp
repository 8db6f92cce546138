// Lden = finsymbden(LAD,perm,dz,firstq)  -- replaces the finsymbden.c gateway (finsymbden.c:112-214)
#include "mexcommon.h"
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
  if (nrhs < 4) mexErrMsgTxt("finsymbden requires more input arguments");
  const mxArray *LAD = prhs[0], *DZ = prhs[2];
  if (!mxIsSparse(LAD)) mexErrMsgTxt("LAD must be sparse");
  const sdm_int m = (sdm_int)mxGetM(LAD), n = (sdm_int)mxGetN(LAD);
  ivec perm = idx_from_dbl(prhs[1], -1);
  const sdm_int nperm = (sdm_int)perm.size();
  if ((sdm_int)mxGetM(DZ) != m || (sdm_int)mxGetN(DZ) != nperm) mexErrMsgTxt("dz size mismatch");
  ivec LADjc = idx_from_mw(mxGetJc(LAD), n + 1), LADir = idx_from_mw(mxGetIr(LAD), mxGetJc(LAD)[n]);
  ivec dzjc = idx_from_mw(mxGetJc(DZ), nperm + 1), dzir = idx_from_mw(mxGetIr(DZ), mxGetJc(DZ)[nperm]);
  const sdm_int firstq = (sdm_int)mxGetScalar(prhs[3]) - 1;
  ivec po(n > 0 ? n : 1), jo(n + 1), fo(n > 0 ? n : 1);
  sdm_check(sdm_finsymbden(m, n, LADjc.data(), LADir.data(), nperm, perm.data(), dzjc.data(), dzir.data(), firstq, po.data(), jo.data(), fo.data()));
  const char *names[] = {"LAD", "perm", "dz", "first"};
  plhs[0] = mxCreateStructMatrix(1, 1, 4, names);
  mxSetField(plhs[0], 0, "LAD", mxDuplicateArray(LAD));
  mxArray *f = mxCreateDoubleMatrix(n, 1, mxREAL);
  for (sdm_int i = 0; i < n; i++) mxGetPr(f)[i] = (double)po[i] + 1.0;
  mxSetField(plhs[0], 0, "perm", f);
  const sdm_int nz = jo[n];
  f = mxCreateSparse(m, n, nz > 0 ? nz : 1, mxREAL);          // dz with the new (cumulative) column pointers
  for (sdm_int j = 0; j <= n; j++) mxGetJc(f)[j] = (mwIndex)jo[j];
  for (sdm_int t = 0; t < nz; t++) { mxGetIr(f)[t] = mxGetIr(DZ)[t]; mxGetPr(f)[t] = mxGetPr(DZ)[t]; }
  mxSetField(plhs[0], 0, "dz", f);
  f = mxCreateDoubleMatrix(n, 1, mxREAL);
  for (sdm_int i = 0; i < n; i++) mxGetPr(f)[i] = (double)fo[i] + 1.0;
  mxSetField(plhs[0], 0, "first", f);
}

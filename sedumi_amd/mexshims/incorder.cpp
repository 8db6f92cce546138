// [perm,dz] = incorder(At[,Ajc1,ifirst])  -- replaces incorder.c:216-330 (same outputs bit for bit, O((nnz+m) log m))
#include "mexcommon.h"
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
  if (nrhs < 1) mexErrMsgTxt("incorder requires more input arguments.");
  if (nlhs > 2) mexErrMsgTxt("incorder produces less output arguments.");
  const mxArray *At = prhs[0];
  if (!mxIsSparse(At)) mexErrMsgTxt("At must be a sparse matrix.");
  const sdm_int N = (sdm_int)mxGetM(At), m = (sdm_int)mxGetN(At);
  ivec jc = idx_from_mw(mxGetJc(At), m + 1), ir = idx_from_mw(mxGetIr(At), mxGetJc(At)[m]), a1;
  sdm_int first = 0, nin = jc[m];
  if (nrhs >= 3) {
    if ((sdm_int)numel(prhs[1]) < m) mexErrMsgTxt("Ajc1 size mismatch");
    a1 = idx_from_dbl(prhs[1], 0);
    first = (sdm_int)mxGetScalar(prhs[2]) - 1;
    nin = 0;
    for (sdm_int j = 0; j < m; j++) nin += jc[j + 1] - a1[j];
  }
  const sdm_int lenud = N - first, cap = lenud < nin ? lenud : nin;
  ivec perm(m > 0 ? m : 1), dzjc(m + 1), dzir(cap > 0 ? cap : 1);
  sdm_check(sdm_incorder(N, m, jc.data(), ir.data(), nrhs >= 3 ? a1.data() : NULL, first, perm.data(), dzjc.data(), dzir.data()));
  mxArray *out0 = mxCreateDoubleMatrix(m, 1, mxREAL);
  for (sdm_int i = 0; i < m; i++) mxGetPr(out0)[i] = (double)(perm[i] + 1);
  mxArray *out1 = mxCreateSparse(N, m, lenud > 0 ? lenud : 1, mxREAL);          // DZ = sparse(lenfull, m, lenud)  (incorder.c:291)
  for (sdm_int j = 0; j <= m; j++) mxGetJc(out1)[j] = (mwIndex)dzjc[j];
  for (sdm_int t = 0; t < dzjc[m]; t++) { mxGetIr(out1)[t] = (mwIndex)dzir[t]; mxGetPr(out1)[t] = 1.0; }
  plhs[0] = out0;
  if (nlhs > 1) plhs[1] = out1; else mxDestroyArray(out1);
}

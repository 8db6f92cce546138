// smult = adenscale(dense, d, blkstart)  -- replaces adenscale.c:87-160
#include "mexcommon.h"
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
  if (nrhs < 3) mexErrMsgTxt("adenscale requires more input arguments");
  const mxArray *dense = prhs[0], *d = prhs[1];
  const sdm_int nl = (sdm_int)mxGetScalar(need_field(dense, "l", "Missing field dense.l."));
  ivec q = idx_from_dbl(need_field(dense, "q", "Missing field dense.q."), -1);
  const mxArray *cols = need_field(dense, "cols", "Missing field dense.cols.");
  const sdm_int nq = (sdm_int)q.size(), nden = (sdm_int)numel(cols) - nl - nq;
  if (nden < 0) mexErrMsgTxt("dense.cols size mismatch.");
  ivec dencols((size_t)(nden > 0 ? nden : 1), 0);
  for (sdm_int i = 0; i < nden; i++) dencols[i] = (sdm_int)mxGetPr(cols)[nl + nq + i] - 1;
  const mxArray *det = need_field(d, "det", "Missing field d.det.");
  if ((sdm_int)numel(prhs[2]) != (sdm_int)numel(det) + 1) mexErrMsgTxt("blkstart size mismatch");
  const double *bs = mxGetPr(prhs[2]);
  ivec blkend((size_t)(nq > 0 ? nq : 1), 0);
  for (sdm_int i = 0; i < nq; i++) blkend[i] = (sdm_int)bs[q[i] + 1] - 1;
  plhs[0] = mxCreateDoubleMatrix(nden, 1, mxREAL);
  if (q.empty()) q.push_back(0);
  sdm_check(sdm_adenscale(nq, nden, mxGetPr(det), q.data(), dencols.data(), blkend.data(), mxGetPr(plhs[0])));
}

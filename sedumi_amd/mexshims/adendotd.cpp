// Ad = adendotd(dense, d, sparAd, Ablk, blkstart)  -- replaces adendotd.c:135-232
#include "mexcommon.h"
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
  if (nrhs < 5) mexErrMsgTxt("adendotd requires more input arguments");
  const mxArray *dense = prhs[0], *d = prhs[1], *S = prhs[2], *B = prhs[3];
  const sdm_int nl = (sdm_int)mxGetScalar(need_field(dense, "l", "Missing field dense.l."));
  ivec q = idx_from_dbl(need_field(dense, "q", "Missing field dense.q."), -1);
  const mxArray *cols = need_field(dense, "cols", "Missing field dense.cols.");
  const sdm_int nq = (sdm_int)q.size(), nden = (sdm_int)numel(cols) - nl - nq;
  if (nden < 0) mexErrMsgTxt("dense.q size mismatch.");
  ivec dencols((size_t)(nden > 0 ? nden : 1), 0);
  for (sdm_int i = 0; i < nden; i++) dencols[i] = (sdm_int)mxGetPr(cols)[nl + nq + i] - 1;
  const mxArray *A = need_field(dense, "A", "Missing field dense.A.");
  if (!mxIsSparse(A)) mexErrMsgTxt("dense.A must be sparse");
  const sdm_int m = (sdm_int)mxGetM(A);
  if ((sdm_int)mxGetN(A) - nl != nq + nden) mexErrMsgTxt("dense.A size mismatch");
  const mxArray *q1 = need_field(d, "q1", "Missing field d.q1."), *q2 = need_field(d, "q2", "Missing field d.q2.");
  const sdm_int lorN = (sdm_int)numel(q1);
  if (!mxIsSparse(S) || (sdm_int)mxGetN(S) != nq) mexErrMsgTxt("Size mismatch sparAD");
  if ((sdm_int)numel(prhs[4]) != lorN + 1) mexErrMsgTxt("blkstart size mismatch");
  const double *bs = mxGetPr(prhs[4]);
  const sdm_int firstQ = (sdm_int)bs[0] - 1;
  ivec blkend((size_t)(nq > 0 ? nq : 1), 0);
  for (sdm_int i = 0; i < nq; i++) blkend[i] = (sdm_int)bs[q[i] + 1] - 1;
  plhs[0] = mxDuplicateArray(B);                                     // Ad = Ablk (adendotd.c:218)
  ivec bjc = idx_from_mw(mxGetJc(B), nq + 1), bir = idx_from_mw(mxGetIr(B), mxGetJc(B)[nq]);
  ivec sjc = idx_from_mw(mxGetJc(S), nq + 1), sir = idx_from_mw(mxGetIr(S), mxGetJc(S)[nq]);
  const sdm_int ncolA = (sdm_int)mxGetN(A);
  ivec ajc = idx_from_mw(mxGetJc(A) + nl, ncolA - nl + 1), air = idx_from_mw(mxGetIr(A), mxGetJc(A)[ncolA]);
  if (q.empty()) q.push_back(0);
  sdm_check(sdm_adendotd(m, nq, nden, bjc.data(), bir.data(), mxGetPr(plhs[0]), sjc.data(), sir.data(), mxGetPr(S), ajc.data(), air.data(),
                         mxGetPr(A), mxGetPr(q1), mxGetPr(q2), firstQ, q.data(), dencols.data(), blkend.data()));
}

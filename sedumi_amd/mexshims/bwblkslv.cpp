// y = bwblkslv(L,b[,ysymb])  -- replaces bwblkslv.c gateway
#include "mexcommon.h"
#define IS_FW 0
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
  if (nrhs < 2) mexErrMsgTxt("bwblkslv requires more input arguments.");
  SymbL L = read_L(prhs[0], true);
  const mxArray *B = prhs[1];
  if ((sdm_int)mxGetM(B) != L.m) mexErrMsgTxt("Size mismatch b.");
  const sdm_int m = L.m, n = (sdm_int)mxGetN(B);
  if (!mxIsSparse(B)) {
    plhs[0] = mxCreateDoubleMatrix(m, n, mxREAL);
    sdm_plan *p = plan_for_factor(L);                               // factor still resident from the last blkchol?
    if (p) {
      for (sdm_int c = 0; c < n; c++) {
        sdm_check(sdm_plan_upload(p, "rhs", mxGetPr(B) + c * m, m));
        sdm_check(IS_FW ? sdm_plan_fwsolve(p) : sdm_plan_bwsolve(p));
        sdm_check(sdm_plan_download(p, "y", mxGetPr(plhs[0]) + c * m, m));
      }
    } else {
      sdm_check((IS_FW ? sdm_fwblkslv : sdm_bwblkslv)(m, L.jc.data(), L.ir.data(), L.pr, L.perm.data(), L.nsuper, L.xsuper.data(),
                                                     n, mxGetPr(B), mxGetPr(plhs[0])));
    }
    return;
  }
  if (nrhs < 3) mexErrMsgTxt("bwblkslv requires more inputs in case of sparse b.");
  const mxArray *Y = prhs[2];
  if ((sdm_int)mxGetM(Y) != m || (sdm_int)mxGetN(Y) != n) mexErrMsgTxt("Size mismatch y.");
  if (!mxIsSparse(Y)) mexErrMsgTxt("y should be sparse.");
  const mwIndex *yjc = mxGetJc(Y), *yir = mxGetIr(Y);
  plhs[0] = mxCreateSparse(m, n, yjc[n], mxREAL);
  memcpy(mxGetJc(plhs[0]), yjc, (n + 1) * sizeof(mwIndex));
  memcpy(mxGetIr(plhs[0]), yir, yjc[n] * sizeof(mwIndex));
  ivec Bjc = idx_from_mw(mxGetJc(B), n + 1), Bir = idx_from_mw(mxGetIr(B), mxGetJc(B)[n]);
  ivec Yjc = idx_from_mw(yjc, n + 1), Yir = idx_from_mw(yir, yjc[n]);
#if IS_FW
  sdm_check(sdm_fwblkslv_sparse(m, L.jc.data(), L.ir.data(), L.pr, L.perm.data(), L.nsuper, L.xsuper.data(), n, Bjc.data(),
                                Bir.data(), mxGetPr(B), Yjc.data(), Yir.data(), mxGetPr(plhs[0])));
#else
  sdm_check(sdm_bwblkslv_sparse(m, L.jc.data(), L.ir.data(), L.pr, L.nsuper, L.xsuper.data(), n, Bjc.data(), Bir.data(),
                                mxGetPr(B), Yjc.data(), Yir.data(), mxGetPr(plhs[0])));
#endif
}

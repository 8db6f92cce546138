// y = fwdpr1(Lden,b)  -- replaces the fwdpr1.c gateway
#include "mexcommon.h"
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
  if (nrhs < 2) mexErrMsgTxt("fwdpr1 requires more input arguments.");
  const mxArray *B = prhs[1];
  if (mxIsSparse(B)) mexErrMsgTxt("b should be full");
  if (!mxIsStruct(prhs[0])) mexErrMsgTxt("Parameter `Lden' should be a structure.");
  const sdm_int m = (sdm_int)mxGetM(B), nrhs_b = (sdm_int)mxGetN(B);
  const mxArray *BJ = need_field(prhs[0], "betajc", "Missing field Lden.betajc.");
  const sdm_int nden = (sdm_int)numel(BJ) - 1;
  plhs[0] = mxDuplicateArray(B);
  if (nden <= 0) return;                                            // no dense columns: y = b
  ivec betajc = idx_from_dbl(BJ, -1);
  const mxArray *P = need_field(prhs[0], "p", "Missing field Lden.p.");
  const mxArray *DP = need_field(prhs[0], "dopiv", "Missing field Lden.dopiv.");
  if ((sdm_int)numel(DP) != nden) mexErrMsgTxt("Size mismatch Lden.dopiv.");
  const mxArray *PP = need_field(prhs[0], "pivperm", "Missing field Lden.pivperm.");
  const mxArray *BE = need_field(prhs[0], "beta", "Missing field Lden.beta.");
  const mxArray *DZ = need_field(prhs[0], "dz", "Missing field Lden.dz.");
  if ((sdm_int)mxGetM(DZ) != m || (sdm_int)mxGetN(DZ) != nden) mexErrMsgTxt("Lden.dz size mismatch.");
  if (!mxIsSparse(DZ)) mexErrMsgTxt("Lden.dz must be sparse.");
  if ((sdm_int)numel(BE) != betajc[nden]) mexErrMsgTxt("Size mismatch Lden.beta.");
  ivec dopiv = idx_from_dbl(DP, 0), pivperm = idx_from_dbl(PP, 0);
  ivec dzjc = idx_from_mw(mxGetJc(DZ), nden + 1), dzir = idx_from_mw(mxGetIr(DZ), mxGetJc(DZ)[nden]);
  if (pivperm.empty()) pivperm.push_back(0);
  sdm_check(sdm_fwdpr1(m, nrhs_b, nden, dzjc.data(), dzir.data(), betajc.data(), mxGetPr(BE), mxGetPr(P), pivperm.data(),
                       (sdm_int)numel(PP), dopiv.data(), mxGetPr(B), mxGetPr(plhs[0])));
}

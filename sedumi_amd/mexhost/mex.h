/*
 * mex.h -- the MEX C API as far as SeDuMi's hot-path gateways use it, for running mexFunction binaries WITHOUT MATLAB or
 * Octave (sedumi_amd/mexhost/mexhost.c implements it; sedumi_amd/mexhost.py drives it).  On a machine with MATLAB / Octave
 * the shims of sedumi_amd/mexshims are compiled against the real mex.h instead and this file is not used.
 * libsedumi_hip.so itself (sedumi_amd/csrc, include/) never includes it.  The oracle compiles the unmodified reference
 * sources against it as well (oracle/Makefile).
 */
#ifndef SDM_MEXHOST_MEX_H
#define SDM_MEXHOST_MEX_H

#include <stddef.h>
#include <stdlib.h>
#include <stdbool.h>
#include <string.h>
#include <math.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef size_t mwSize;
typedef size_t mwIndex;
typedef ptrdiff_t mwSignedIndex;

typedef enum { mxREAL = 0, mxCOMPLEX = 1 } mxComplexity;

enum { MEXHOST_DOUBLE = 0, MEXHOST_SPARSE = 1, MEXHOST_STRUCT = 2 };

typedef struct mxArray_tag {
  int kind;            /* MEXHOST_DOUBLE / MEXHOST_SPARSE / MEXHOST_STRUCT */
  size_t m, n;
  double *pr;          /* values (full: m*n, sparse: nzmax) */
  size_t *ir, *jc;     /* sparse only */
  size_t nzmax;
  int nfields;         /* struct only (1x1 structs) */
  char **fnames;
  struct mxArray_tag **fvals;
} mxArray;

/* --- queries --- */
double *mxGetPr(const mxArray *a);
mwIndex *mxGetJc(const mxArray *a);
mwIndex *mxGetIr(const mxArray *a);
size_t mxGetM(const mxArray *a);
size_t mxGetN(const mxArray *a);
double mxGetScalar(const mxArray *a);
mxArray *mxGetField(const mxArray *a, mwIndex idx, const char *name);
bool mxIsSparse(const mxArray *a);
bool mxIsStruct(const mxArray *a);
/* --- constructors / destructors --- */
mxArray *mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity c);
mxArray *mxCreateSparse(mwSize m, mwSize n, mwSize nzmax, mxComplexity c);
mxArray *mxCreateStructMatrix(mwSize m, mwSize n, int nfields, const char **names);
mxArray *mxDuplicateArray(const mxArray *a);
void mxDestroyArray(mxArray *a);
/* --- setters --- */
void mxSetField(mxArray *a, mwIndex idx, const char *name, mxArray *v);
void mxSetPr(mxArray *a, double *pr);
void mxSetIr(mxArray *a, mwIndex *ir);
void mxSetJc(mxArray *a, mwIndex *jc);
void mxSetM(mxArray *a, mwSize m);
void mxSetN(mxArray *a, mwSize n);
void mxSetNzmax(mxArray *a, mwSize nzmax);
/* --- memory --- */
void *mxCalloc(size_t n, size_t sz);
void *mxMalloc(size_t sz);
void *mxRealloc(void *p, size_t sz);
void mxFree(void *p);
/* --- errors --- */
void mexErrMsgTxt(const char *msg);
void mexWarnMsgTxt(const char *msg);
int mexPrintf(const char *fmt, ...);
int mexAtExit(void (*fn)(void));
/* --- workspace variables (only the "global" workspace exists here; sedumi.m keeps ADA_sedumi_ there) --- */
const mxArray *mexGetVariablePtr(const char *workspace, const char *name);
int mexPutVariable(const char *workspace, const char *name, const mxArray *value);   /* stores a copy; 0 = ok */

#ifdef NDEBUG
#define mxAssert(c, msg) ((void)0)
#else
#define mxAssert(c, msg) do { if (!(c)) mexErrMsgTxt("mxAssert failed: " #c); } while (0)
#endif

/* entry point every MEX source defines */
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]);

/* --- host side: what sedumi_amd/mexhost.py calls --- */
typedef void (*mexhost_mexfun_t)(int, mxArray **, int, const mxArray **);
int mexhost_call(mexhost_mexfun_t f, int nlhs, mxArray **plhs, int nrhs, const mxArray **prhs);
const char *mexhost_last_error(void);
int mexhost_kind(const mxArray *a);
size_t mexhost_nzmax(const mxArray *a);
int mexhost_nfields(const mxArray *a);
const char *mexhost_fieldname(const mxArray *a, int i);
mxArray *mexhost_fieldval(const mxArray *a, int i);
mxArray *mexhost_new_struct(void);
void mexhost_set_global(const char *name, const mxArray *value);   /* copy in (NULL clears) */
const mxArray *mexhost_get_global(const char *name);

#ifdef __cplusplus
}
#endif
#endif

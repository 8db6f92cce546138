/*
 * mex.h -- TEST INFRASTRUCTURE ONLY (part of oracle/).
 *
 * A minimal, self-written stand-in for the MATLAB/Octave MEX C API, just big
 * enough to compile the *unmodified* SeDuMi reference sources that live under
 * /root/reference into oracle/_ref/ *.so (see oracle/Makefile), and to compile
 * our own mexFunction shims (sedumi_amd/mex/) for a syntax/link check in a
 * container that has neither MATLAB nor Octave.
 *
 * Nothing in the product path (sedumi_amd/, include/) includes this file.
 */
#ifndef SDM_ORACLE_MEX_H
#define SDM_ORACLE_MEX_H

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

enum { SHIM_DOUBLE = 0, SHIM_SPARSE = 1, SHIM_STRUCT = 2 };

typedef struct mxArray_tag {
  int kind;            /* SHIM_DOUBLE / SHIM_SPARSE / SHIM_STRUCT */
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

/* --- shim-only helpers used by the python/ctypes driver --- */
typedef void (*shim_mexfun_t)(int, mxArray **, int, const mxArray **);
int shim_call(shim_mexfun_t f, int nlhs, mxArray **plhs, int nrhs, const mxArray **prhs);
const char *shim_last_error(void);
int shim_kind(const mxArray *a);
size_t shim_nzmax(const mxArray *a);
int shim_nfields(const mxArray *a);
const char *shim_fieldname(const mxArray *a, int i);
mxArray *shim_fieldval(const mxArray *a, int i);
mxArray *shim_new_struct(void);
void shim_set_global(const char *name, const mxArray *value);   /* copy in (NULL clears) */
const mxArray *shim_get_global(const char *name);

#ifdef __cplusplus
}
#endif
#endif

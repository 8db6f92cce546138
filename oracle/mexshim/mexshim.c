/*
 * mexshim.c -- TEST INFRASTRUCTURE ONLY (part of oracle/).
 *
 * Self-written implementation of the small slice of the MEX C API and of the
 * five BLAS-1 routines that the SeDuMi reference hot path needs (SURVEY.md
 * section 8c).  Built into oracle/_ref/libmexshim.so; every compiled reference MEX
 * (oracle/_ref/<name>.so) links against it, and oracle/refmex.py drives them
 * through ctypes.  Not part of the product.
 */
#include "mex.h"
#include "blas.h"
#include <setjmp.h>
#include <stdio.h>
#include <stdarg.h>

static jmp_buf g_jmp;
static int g_jmp_armed = 0;
static char g_errmsg[1024];

/* ------------------------------------------------------------------ queries */
double *mxGetPr(const mxArray *a) { return a ? a->pr : NULL; }
mwIndex *mxGetJc(const mxArray *a) { return a ? a->jc : NULL; }
mwIndex *mxGetIr(const mxArray *a) { return a ? a->ir : NULL; }
size_t mxGetM(const mxArray *a) { return a ? a->m : 0; }
size_t mxGetN(const mxArray *a) { return a ? a->n : 0; }
double mxGetScalar(const mxArray *a) {
  if (!a || !a->pr) return 0.0;
  if (a->kind == SHIM_SPARSE && a->jc[a->n] == 0) return 0.0;
  return a->pr[0];
}
bool mxIsSparse(const mxArray *a) { return a && a->kind == SHIM_SPARSE; }
bool mxIsStruct(const mxArray *a) { return a && a->kind == SHIM_STRUCT; }

mxArray *mxGetField(const mxArray *a, mwIndex idx, const char *name) {
  int i;
  (void)idx;
  if (!a || a->kind != SHIM_STRUCT) return NULL;
  for (i = 0; i < a->nfields; i++)
    if (strcmp(a->fnames[i], name) == 0) return a->fvals[i];
  return NULL;
}

/* ------------------------------------------------------------ constructors */
static mxArray *new_array(int kind, size_t m, size_t n) {
  mxArray *a = (mxArray *)calloc(1, sizeof(mxArray));
  a->kind = kind; a->m = m; a->n = n;
  return a;
}

mxArray *mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity c) {
  mxArray *a = new_array(SHIM_DOUBLE, m, n);
  size_t len = m * n;
  (void)c;
  a->pr = (double *)calloc(len ? len : 1, sizeof(double));
  return a;
}

mxArray *mxCreateSparse(mwSize m, mwSize n, mwSize nzmax, mxComplexity c) {
  mxArray *a = new_array(SHIM_SPARSE, m, n);
  (void)c;
  if (nzmax < 1) nzmax = 1;
  a->nzmax = nzmax;
  a->pr = (double *)calloc(nzmax, sizeof(double));
  a->ir = (size_t *)calloc(nzmax, sizeof(size_t));
  a->jc = (size_t *)calloc(n + 1, sizeof(size_t));
  return a;
}

mxArray *shim_new_struct(void) { return new_array(SHIM_STRUCT, 1, 1); }

mxArray *mxCreateStructMatrix(mwSize m, mwSize n, int nfields, const char **names) {
  mxArray *a = new_array(SHIM_STRUCT, m, n);
  int i;
  for (i = 0; i < nfields; i++) mxSetField(a, 0, names[i], NULL);
  return a;
}

void mxSetField(mxArray *a, mwIndex idx, const char *name, mxArray *v) {
  int i;
  (void)idx;
  if (!a || a->kind != SHIM_STRUCT) return;
  for (i = 0; i < a->nfields; i++)
    if (strcmp(a->fnames[i], name) == 0) { a->fvals[i] = v; return; }
  a->fnames = (char **)realloc(a->fnames, (a->nfields + 1) * sizeof(char *));
  a->fvals = (mxArray **)realloc(a->fvals, (a->nfields + 1) * sizeof(mxArray *));
  a->fnames[a->nfields] = strdup(name);
  a->fvals[a->nfields] = v;
  a->nfields++;
}

mxArray *mxDuplicateArray(const mxArray *s) {
  mxArray *a;
  int i;
  if (!s) return NULL;
  if (s->kind == SHIM_DOUBLE) {
    a = mxCreateDoubleMatrix(s->m, s->n, mxREAL);
    memcpy(a->pr, s->pr, s->m * s->n * sizeof(double));
  } else if (s->kind == SHIM_SPARSE) {
    a = mxCreateSparse(s->m, s->n, s->nzmax, mxREAL);
    memcpy(a->jc, s->jc, (s->n + 1) * sizeof(size_t));
    memcpy(a->ir, s->ir, s->jc[s->n] * sizeof(size_t));
    memcpy(a->pr, s->pr, s->jc[s->n] * sizeof(double));
  } else {
    a = new_array(SHIM_STRUCT, s->m, s->n);
    for (i = 0; i < s->nfields; i++)
      mxSetField(a, 0, s->fnames[i], mxDuplicateArray(s->fvals[i]));
  }
  return a;
}

void mxDestroyArray(mxArray *a) {
  int i;
  if (!a) return;
  if (a->kind == SHIM_STRUCT) {
    for (i = 0; i < a->nfields; i++) { mxDestroyArray(a->fvals[i]); free(a->fnames[i]); }
    free(a->fnames); free(a->fvals);
  } else {
    free(a->pr); free(a->ir); free(a->jc);
  }
  free(a);
}

/* ----------------------------------------------------------------- setters */
void mxSetPr(mxArray *a, double *pr) { a->pr = pr; }
void mxSetIr(mxArray *a, mwIndex *ir) { a->ir = ir; }
void mxSetJc(mxArray *a, mwIndex *jc) { a->jc = jc; }
void mxSetM(mxArray *a, mwSize m) { a->m = m; }
void mxSetN(mxArray *a, mwSize n) { a->n = n; }
void mxSetNzmax(mxArray *a, mwSize nzmax) { a->nzmax = nzmax; }

/* ------------------------------------------------------------------ memory */
void *mxCalloc(size_t n, size_t sz) { return calloc(n ? n : 1, sz ? sz : 1); }
void *mxMalloc(size_t sz) { return malloc(sz ? sz : 1); }
void *mxRealloc(void *p, size_t sz) { return realloc(p, sz ? sz : 1); }
void mxFree(void *p) { free(p); }

/* ------------------------------------------------------------------ errors */
void mexErrMsgTxt(const char *msg) {
  snprintf(g_errmsg, sizeof g_errmsg, "%s", msg ? msg : "(null)");
  if (g_jmp_armed) longjmp(g_jmp, 1);
  fprintf(stderr, "mexErrMsgTxt outside shim_call: %s\n", g_errmsg);
  abort();
}
void mexWarnMsgTxt(const char *msg) { fprintf(stderr, "mex warning: %s\n", msg); }
int mexPrintf(const char *fmt, ...) {
  va_list ap; int r;
  va_start(ap, fmt); r = vfprintf(stdout, fmt, ap); va_end(ap);
  return r;
}

/* exit handlers are run when the shim library is unloaded */
static void (*g_atexit[16])(void); static int g_natexit = 0;
int mexAtExit(void (*fn)(void)) { if (g_natexit < 16) g_atexit[g_natexit++] = fn; return 0; }
__attribute__((destructor)) static void run_atexit(void) { int i; for (i = 0; i < g_natexit; i++) g_atexit[i](); }

int shim_call(shim_mexfun_t f, int nlhs, mxArray **plhs, int nrhs, const mxArray **prhs) {
  g_errmsg[0] = 0;
  if (setjmp(g_jmp)) { g_jmp_armed = 0; return 1; }
  g_jmp_armed = 1;
  f(nlhs, plhs, nrhs, prhs);
  g_jmp_armed = 0;
  return 0;
}
const char *shim_last_error(void) { return g_errmsg; }
int shim_kind(const mxArray *a) { return a->kind; }
size_t shim_nzmax(const mxArray *a) { return a->nzmax; }
int shim_nfields(const mxArray *a) { return a->nfields; }
const char *shim_fieldname(const mxArray *a, int i) { return a->fnames[i]; }
mxArray *shim_fieldval(const mxArray *a, int i) { return a->fvals[i]; }

/* ------------------------------------------------- "global" workspace */
#define SHIM_NGLOB 8
static struct { char name[64]; mxArray *val; } g_glob[SHIM_NGLOB];
const mxArray *shim_get_global(const char *name) {
  int i;
  for (i = 0; i < SHIM_NGLOB; i++) if (g_glob[i].val && !strcmp(g_glob[i].name, name)) return g_glob[i].val;
  return NULL;
}
void shim_set_global(const char *name, const mxArray *value) {
  int i, slot = -1;
  for (i = 0; i < SHIM_NGLOB; i++) if (g_glob[i].val && !strcmp(g_glob[i].name, name)) slot = i;
  if (slot < 0) for (i = 0; i < SHIM_NGLOB && slot < 0; i++) if (!g_glob[i].val) slot = i;
  if (slot < 0) return;
  if (g_glob[slot].val) mxDestroyArray(g_glob[slot].val);
  g_glob[slot].val = value ? mxDuplicateArray(value) : NULL;
  strncpy(g_glob[slot].name, name, sizeof g_glob[slot].name - 1);
}
const mxArray *mexGetVariablePtr(const char *workspace, const char *name) {
  if (strcmp(workspace, "global")) return NULL;
  return shim_get_global(name);
}
int mexPutVariable(const char *workspace, const char *name, const mxArray *value) {
  if (strcmp(workspace, "global")) return 1;
  shim_set_global(name, value);
  return 0;
}

/* ------------------------------------------------- BLAS-1, Fortran semantics
 * Two back-ends behind the names the reference links against (blksdp.h:43-62, non-OCTAVE branch):
 *   naive   plain sequential loops (deterministic; the default, used by every parity test), and
 *   an optimised host BLAS bound at run time by shim_use_blas(path) -- the ILP64 Fortran entry points
 *   (ptrdiff_t integers, like the reference's calls) of the OpenBLAS that ships inside numpy's wheel
 *   (scipy_ddot_64_ ...), for the second CPU baseline of bench.py. */
#include <dlfcn.h>
static double naive_ddot(const ptrdiff_t *n, const double *x, const ptrdiff_t *incx, const double *y, const ptrdiff_t *incy) {
  ptrdiff_t i; double s = 0.0;
  for (i = 0; i < *n; i++) s += x[i * *incx] * y[i * *incy];
  return s;
}
static void naive_daxpy(const ptrdiff_t *n, const double *a, const double *x, const ptrdiff_t *incx, double *y, const ptrdiff_t *incy) {
  ptrdiff_t i;
  for (i = 0; i < *n; i++) y[i * *incy] += *a * x[i * *incx];
}
static void naive_dscal(const ptrdiff_t *n, const double *a, double *x, const ptrdiff_t *incx) {
  ptrdiff_t i;
  for (i = 0; i < *n; i++) x[i * *incx] *= *a;
}
static void naive_dcopy(const ptrdiff_t *n, const double *x, const ptrdiff_t *incx, double *y, const ptrdiff_t *incy) {
  ptrdiff_t i;
  for (i = 0; i < *n; i++) y[i * *incy] = x[i * *incx];
}
/* Fortran IDAMAX: 1-based index of the FIRST element of maximum |x|; 0 if n<1. */
static ptrdiff_t naive_idamax(const ptrdiff_t *n, const double *x, const ptrdiff_t *incx) {
  ptrdiff_t i, imax = 0; double vmax;
  if (*n < 1) return 0;
  vmax = fabs(x[0]);
  for (i = 1; i < *n; i++)
    if (fabs(x[i * *incx]) > vmax) { vmax = fabs(x[i * *incx]); imax = i; }
  return imax + 1;
}
typedef double (*ddot_fn)(const ptrdiff_t *, const double *, const ptrdiff_t *, const double *, const ptrdiff_t *);
typedef void (*daxpy_fn)(const ptrdiff_t *, const double *, const double *, const ptrdiff_t *, double *, const ptrdiff_t *);
typedef void (*dscal_fn)(const ptrdiff_t *, const double *, double *, const ptrdiff_t *);
typedef void (*dcopy_fn)(const ptrdiff_t *, const double *, const ptrdiff_t *, double *, const ptrdiff_t *);
typedef ptrdiff_t (*idamax_fn)(const ptrdiff_t *, const double *, const ptrdiff_t *);
static ddot_fn p_ddot = naive_ddot;
static daxpy_fn p_daxpy = naive_daxpy;
static dscal_fn p_dscal = naive_dscal;
static dcopy_fn p_dcopy = naive_dcopy;
static idamax_fn p_idamax = naive_idamax;
double ddot(const ptrdiff_t *n, const double *x, const ptrdiff_t *incx, const double *y, const ptrdiff_t *incy) { return p_ddot(n, x, incx, y, incy); }
void daxpy(const ptrdiff_t *n, const double *a, const double *x, const ptrdiff_t *incx, double *y, const ptrdiff_t *incy) { p_daxpy(n, a, x, incx, y, incy); }
void dscal(const ptrdiff_t *n, const double *a, double *x, const ptrdiff_t *incx) { p_dscal(n, a, x, incx); }
void dcopy(const ptrdiff_t *n, const double *x, const ptrdiff_t *incx, double *y, const ptrdiff_t *incy) { p_dcopy(n, x, incx, y, incy); }
ptrdiff_t idamax(const ptrdiff_t *n, const double *x, const ptrdiff_t *incx) { return p_idamax(n, x, incx); }
/* path = NULL: back to the naive loops.  Returns 0 on success, 1 if the library or one of the five symbols is missing
 * (the binding is then left unchanged). */
int shim_use_blas(const char *path, const char *prefix, const char *suffix) {
  void *h; char nm[128]; void *f[5]; int i;
  static const char *names[5] = {"ddot", "daxpy", "dscal", "dcopy", "idamax"};
  if (!path) { p_ddot = naive_ddot; p_daxpy = naive_daxpy; p_dscal = naive_dscal; p_dcopy = naive_dcopy; p_idamax = naive_idamax; return 0; }
  h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!h) return 1;
  for (i = 0; i < 5; i++) {
    snprintf(nm, sizeof nm, "%s%s%s", prefix ? prefix : "", names[i], suffix ? suffix : "");
    f[i] = dlsym(h, nm);
    if (!f[i]) return 1;
  }
  p_ddot = (ddot_fn)f[0]; p_daxpy = (daxpy_fn)f[1]; p_dscal = (dscal_fn)f[2]; p_dcopy = (dcopy_fn)f[3]; p_idamax = (idamax_fn)f[4];
  return 0;
}

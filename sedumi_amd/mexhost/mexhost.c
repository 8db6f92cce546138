/*
 * mexhost.c -- a small host for mexFunction binaries outside MATLAB / Octave.
 *
 * Self-written implementation of the slice of the MEX C API that SeDuMi's hot-path gateways use (mxArray: full, sparse
 * and 1x1 struct arrays; mxCreate / mxGet / mxSet; mexErrMsgTxt as a longjmp back to the caller; the "global" workspace
 * of mexGetVariablePtr / mexPutVariable; mexAtExit).  Built into sedumi_amd/lib/libsdm_mexhost.so.  With it the shims
 * of sedumi_amd/mexshims -- the very mexFunction sources a MATLAB / Octave user compiles with mex / mkoctfile -- run in a
 * container that has neither: bench.py's `mex_inclusive` leg and the boundary tests drive them through
 * sedumi_amd/mexhost.py.  The oracle (oracle/Makefile) compiles the unmodified reference MEX sources against the same
 * header and links them to the same library, so reference and replacement are driven through identical marshalling.
 */
#include "mex.h"
#include <malloc.h>
#include <setjmp.h>
#include <stdio.h>
#include <stdarg.h>

/* Memory policy of this host.  SeDuMi's gateways hand back whole arrays (ADA' three times per iteration, L.L once: megabytes each),
 * so every call allocates and frees blocks far above glibc's mmap threshold: by default each of them is a fresh mmap -- page
 * faults for every 4 KB on first touch -- and a munmap on free.  In the build container that is 2.8 ms per 7 MB sparse array
 * (calloc, fill, free) against 0.58 ms from a heap that keeps its pages; on the GPU boxes of round 4 control07's 3.5 MB arrays measured
 * the same either way (509 against 513 units/s through the shims), MAXCUT-4000's 128 MB arrays 41 ms against 140 ms per unit
 * (profiles/r04n_*, r04u_*).  Like MATLAB's own memory manager this host keeps
 * freed blocks: no mmap for single blocks, no trimming.  (Process-wide, set when the library is loaded; MEXHOST_DEFAULT_MALLOC=1 in the
 * environment leaves glibc's defaults alone.) */
__attribute__((constructor)) static void mexhost_memory_policy(void) {
  const char *off = getenv("MEXHOST_DEFAULT_MALLOC");
  if (off && off[0] == '1') return;
  mallopt(M_MMAP_THRESHOLD, 1 << 30);
  mallopt(M_TRIM_THRESHOLD, 1 << 30);
  mallopt(M_TOP_PAD, 64 << 20);
}

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
  if (a->kind == MEXHOST_SPARSE && a->jc[a->n] == 0) return 0.0;
  return a->pr[0];
}
bool mxIsSparse(const mxArray *a) { return a && a->kind == MEXHOST_SPARSE; }
bool mxIsStruct(const mxArray *a) { return a && a->kind == MEXHOST_STRUCT; }

mxArray *mxGetField(const mxArray *a, mwIndex idx, const char *name) {
  int i;
  (void)idx;
  if (!a || a->kind != MEXHOST_STRUCT) return NULL;
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
  mxArray *a = new_array(MEXHOST_DOUBLE, m, n);
  size_t len = m * n;
  (void)c;
  a->pr = (double *)calloc(len ? len : 1, sizeof(double));
  return a;
}

mxArray *mxCreateSparse(mwSize m, mwSize n, mwSize nzmax, mxComplexity c) {
  mxArray *a = new_array(MEXHOST_SPARSE, m, n);
  (void)c;
  if (nzmax < 1) nzmax = 1;
  a->nzmax = nzmax;
  a->pr = (double *)calloc(nzmax, sizeof(double));
  a->ir = (size_t *)calloc(nzmax, sizeof(size_t));
  a->jc = (size_t *)calloc(n + 1, sizeof(size_t));
  return a;
}

mxArray *mexhost_new_struct(void) { return new_array(MEXHOST_STRUCT, 1, 1); }

mxArray *mxCreateStructMatrix(mwSize m, mwSize n, int nfields, const char **names) {
  mxArray *a = new_array(MEXHOST_STRUCT, m, n);
  int i;
  for (i = 0; i < nfields; i++) mxSetField(a, 0, names[i], NULL);
  return a;
}

void mxSetField(mxArray *a, mwIndex idx, const char *name, mxArray *v) {
  int i;
  (void)idx;
  if (!a || a->kind != MEXHOST_STRUCT) return;
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
  if (s->kind == MEXHOST_DOUBLE) {
    a = mxCreateDoubleMatrix(s->m, s->n, mxREAL);
    memcpy(a->pr, s->pr, s->m * s->n * sizeof(double));
  } else if (s->kind == MEXHOST_SPARSE) {
    a = mxCreateSparse(s->m, s->n, s->nzmax, mxREAL);
    memcpy(a->jc, s->jc, (s->n + 1) * sizeof(size_t));
    memcpy(a->ir, s->ir, s->jc[s->n] * sizeof(size_t));
    memcpy(a->pr, s->pr, s->jc[s->n] * sizeof(double));
  } else {
    a = new_array(MEXHOST_STRUCT, s->m, s->n);
    for (i = 0; i < s->nfields; i++)
      mxSetField(a, 0, s->fnames[i], mxDuplicateArray(s->fvals[i]));
  }
  return a;
}

void mxDestroyArray(mxArray *a) {
  int i;
  if (!a) return;
  if (a->kind == MEXHOST_STRUCT) {
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
  fprintf(stderr, "mexErrMsgTxt outside mexhost_call: %s\n", g_errmsg);
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

int mexhost_call(mexhost_mexfun_t f, int nlhs, mxArray **plhs, int nrhs, const mxArray **prhs) {
  g_errmsg[0] = 0;
  if (setjmp(g_jmp)) { g_jmp_armed = 0; return 1; }
  g_jmp_armed = 1;
  f(nlhs, plhs, nrhs, prhs);
  g_jmp_armed = 0;
  return 0;
}
const char *mexhost_last_error(void) { return g_errmsg; }
int mexhost_kind(const mxArray *a) { return a->kind; }
size_t mexhost_nzmax(const mxArray *a) { return a->nzmax; }
int mexhost_nfields(const mxArray *a) { return a->nfields; }
const char *mexhost_fieldname(const mxArray *a, int i) { return a->fnames[i]; }
mxArray *mexhost_fieldval(const mxArray *a, int i) { return a->fvals[i]; }

/* ------------------------------------------------- "global" workspace */
#define MEXHOST_NGLOB 8
static struct { char name[64]; mxArray *val; } g_glob[MEXHOST_NGLOB];
const mxArray *mexhost_get_global(const char *name) {
  int i;
  for (i = 0; i < MEXHOST_NGLOB; i++) if (g_glob[i].val && !strcmp(g_glob[i].name, name)) return g_glob[i].val;
  return NULL;
}
void mexhost_set_global(const char *name, const mxArray *value) {
  int i, slot = -1;
  for (i = 0; i < MEXHOST_NGLOB; i++) if (g_glob[i].val && !strcmp(g_glob[i].name, name)) slot = i;
  if (slot < 0) for (i = 0; i < MEXHOST_NGLOB && slot < 0; i++) if (!g_glob[i].val) slot = i;
  if (slot < 0) return;
  if (g_glob[slot].val) mxDestroyArray(g_glob[slot].val);
  g_glob[slot].val = value ? mxDuplicateArray(value) : NULL;
  strncpy(g_glob[slot].name, name, sizeof g_glob[slot].name - 1);
}
const mxArray *mexGetVariablePtr(const char *workspace, const char *name) {
  if (strcmp(workspace, "global")) return NULL;
  return mexhost_get_global(name);
}
int mexPutVariable(const char *workspace, const char *name, const mxArray *value) {
  if (strcmp(workspace, "global")) return 1;
  mexhost_set_global(name, value);
  return 0;
}


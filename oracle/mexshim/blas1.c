/*
 * blas1.c -- TEST INFRASTRUCTURE ONLY (part of oracle/).
 *
 * The five BLAS-1 routines the SeDuMi reference hot path calls, with the MATLAB "blas.h" calling
 * convention.  Built into oracle/_ref/libmexshim.so next to the MEX host of the package (sedumi_amd/mexhost, the mx* API);
 * every compiled reference MEX (oracle/_ref/<name>.so) links against both.  Not part of the product.
 */
#include "blas.h"
#include <math.h>
#include <stdio.h>
#include <string.h>
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

/*
 * blas.h -- TEST INFRASTRUCTURE ONLY (part of oracle/).
 * Declarations of the five level-1 BLAS routines the reference hot path calls
 * (sdmauxScalarmul.c:44-66, sdmauxRdot.c:47,57, blkchol2.c:56,69), with the
 * MATLAB "blas.h" calling convention (ptrdiff_t integers, no trailing
 * underscore; blksdp.h:43-62 non-OCTAVE branch).  Implemented in mexshim.c as
 * plain sequential loops with Fortran semantics (idamax is 1-BASED).
 */
#ifndef SDM_ORACLE_BLAS_H
#define SDM_ORACLE_BLAS_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
double ddot(const ptrdiff_t *n, const double *x, const ptrdiff_t *incx, const double *y, const ptrdiff_t *incy);
void daxpy(const ptrdiff_t *n, const double *a, const double *x, const ptrdiff_t *incx, double *y, const ptrdiff_t *incy);
void dscal(const ptrdiff_t *n, const double *a, double *x, const ptrdiff_t *incx);
void dcopy(const ptrdiff_t *n, const double *x, const ptrdiff_t *incx, double *y, const ptrdiff_t *incy);
ptrdiff_t idamax(const ptrdiff_t *n, const double *x, const ptrdiff_t *incx);
#ifdef __cplusplus
}
#endif
#endif

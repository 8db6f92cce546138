"""oracle/restate.py -- TEST INFRASTRUCTURE ONLY: CPU restatement of the hot-path algorithms in numpy.

Each function restates (in dense, readable numpy -- O(m^3), meant for sizes the tests finish in seconds) what
the cited reference routine computes.  It is the checker that travels everywhere (the compiled reference in
oracle/_ref is the stronger pin; tests/test_oracle.py checks this file against it on every config).
PARITY PIN: pinned against oracle/_ref (the unmodified reference C compiled here) -- see tests/test_oracle.py;
the reference itself ships no function-level golden vectors.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp


# ------------------------------------------------------------------ ADA'
def dsqr_vector(K, d):
    """getada1.c:106-118: dsqr = [d.l; -d.det; kron(d.det, ones)] over the LP + Lorentz rows."""
    lpN = int(np.asarray(K["l"]).ravel()[0])
    q = np.asarray(K["q"], dtype=np.int64).ravel()
    det = np.asarray(d["det"], dtype=np.float64).ravel()
    parts = [np.asarray(d["l"], dtype=np.float64).ravel()[:lpN], -det]
    for k, nk in enumerate(q):
        parts.append(np.full(nk - 1, det[k]))
    return np.concatenate(parts)


def getada(At, K, d, DAtq, udsqr):
    """Dense ADA' (m x m, symmetric) and absd as produced by getada1 -> getada2 -> getada3
    (getada1.c:89-152, getada2.c:74-118, getada3.c:253-361, spscale.c:249-305, spmakesym getada3.c:151-180).
    At is the internal N x m matrix (PSD parts folded into the lower triangle)."""
    At = sp.csc_matrix(At)
    N, m = At.shape
    s = np.asarray(K["s"], dtype=np.int64).ravel()
    blk = np.asarray(K["blkstart"], dtype=np.int64).ravel() - 1
    nq = np.asarray(K["q"]).size
    psd0 = blk[1 + nq] if s.size else N
    dsqr = dsqr_vector(K, d)
    Alq = At[:psd0, :].toarray()
    ADA = Alq.T @ (dsqr[:psd0, None] * Alq)                       # getada1: a_i' diag(dsqr) a_j
    if DAtq is not None and DAtq.shape[0] > 0:
        Q = sp.csc_matrix(DAtq).toarray()
        ADA = ADA + Q.T @ Q                                       # getada2: DAt.q' * DAt.q
    base_diag = np.diag(ADA).copy()
    absd = np.zeros(m)
    if s.size == 0:
        return ADA, base_diag.copy()                              # cpspdiag, getada3.c:549-552
    Apsd = At[psd0:, :].toarray()
    Z = np.zeros_like(Apsd)
    off = 0
    ud = np.asarray(udsqr, dtype=np.float64).ravel()
    for n in s:                                                   # real symmetric blocks (sprealdxd)
        D = ud[off:off + n * n].reshape(n, n, order="F")
        for j in range(m):
            x = Apsd[off:off + n * n, j]
            if not x.any():
                continue
            X = x.reshape(n, n, order="F")
            W = D @ X @ D                                         # Z = D sym(X) D = (W + W')/2, spscale.c:283-304
            Z[off:off + n * n, j] = ((W + W.T) / 2).ravel(order="F")
        off += n * n
    ADA = ADA + Apsd.T @ Z                                        # getada3.c:333-351: ada_ij += a_i' daj
    for j in range(m):
        if Apsd[:, j].any():
            absd[j] = base_diag[j] + np.abs(Apsd[:, j] * Z[:, j]).sum()   # getada3.c:341-347
    return (ADA + ADA.T) / 2, absd


# ------------------------------------------------------------ factor / solve
def blkchol_sparse(X, Ljc, Lir, xsuper, perm, pars, absd=None):
    """L D L' = X(perm,perm) with SeDuMi's never-fail pivot rule (blkchol.c:157-231 spchol, blkchol2.c:96-167
    cholonBlk), on the symbolic pattern (Ljc, Lir, xsuper; 0-based).  Returns dense unit-lower L, d, and the
    skip / add lists [(index, value)].  The idamax quirk of maxabs (blkchol2.c:66-70: the 1-based Fortran index
    is used as a C index, i.e. the element AFTER the first maximum is read) is reproduced on the reference's
    packed column storage."""
    X = np.asarray(X.toarray() if sp.issparse(X) else X, dtype=np.float64)
    m = X.shape[0]
    perm = np.asarray(perm, dtype=np.int64)
    P = X[np.ix_(perm, perm)]
    canceltol, maxu, abstol = float(pars["canceltol"]), float(pars["maxu"]), max(float(pars["abstol"]), 0.0)
    orgd = (np.asarray(absd, dtype=np.float64).ravel()[perm] if absd is not None else np.diag(P).copy())
    ub = max(0.0, np.diag(P).max()) / maxu ** 2                                    # blkchol.c:168-175
    lb = np.maximum(canceltol * orgd, abstol)                                      # blkchol.c:180-184
    snode = np.zeros(m, dtype=np.int64)
    for s_ in range(len(xsuper) - 1):
        snode[xsuper[s_]:xsuper[s_ + 1]] = s_
    W = np.tril(P).copy()            # running Schur complement (lower part); column k becomes x_ik = l_ik d_k
    L = np.eye(m)
    dvec = np.zeros(m)
    skip, add = [], []
    for k in range(m):
        rows = np.asarray(Lir[Ljc[k] + 1:Ljc[k + 1]], dtype=np.int64)      # below-diagonal pattern of column k
        xkk = W[k, k]
        if xkk > lb[k]:
            if rows.size > 0 and xkk < ub:
                col = W[rows, k]
                imax = int(np.argmax(np.abs(col)))                         # first maximum (IDAMAX)
                if imax + 1 < col.size:
                    probe = col[imax + 1]
                elif k + 1 < m:                                            # runs into the next column's diagonal
                    probe = W[k + 1, k + 1] if snode[k + 1] == snode[k] else P[k + 1, k + 1]
                else:
                    probe = 0.0
                ubk = abs(probe) / maxu
                if xkk < ubk:
                    add.append((k, ubk - xkk)); xkk = ubk                  # blkchol2.c:125-131
            dvec[k] = xkk
            if rows.size:
                lcol = W[rows, k] / xkk
                L[rows, k] = lcol
                W[np.ix_(rows, rows)] -= np.tril(np.outer(lcol, W[rows, k]))
        else:
            skip.append((k, xkk)); dvec[k] = 0.0                           # blkchol2.c:157-161; L(:,k) = e_k
    return L, dvec, skip, add


def blkchol_dense(X, perm, pars, absd=None):
    """The dense shortcut of symbchol.m:75-77: one supernode, L.L = tril(ones)."""
    m = X.shape[0]
    Ljc = np.concatenate(([0], np.cumsum(np.arange(m, 0, -1))))
    Lir = np.concatenate([np.arange(j, m) for j in range(m)]) if m else np.zeros(0, dtype=np.int64)
    return blkchol_sparse(X, Ljc, Lir, np.array([0, m]), perm, pars, absd)


def fwsolve_dense(L, perm, b):
    """y = L \\ b(perm)   (fwblkslv.c:298-303)"""
    return sla.solve_triangular(L, np.asarray(b, dtype=np.float64)[perm], lower=True, unit_diagonal=True)


def bwsolve_dense(L, perm, b):
    """y(perm) = L' \\ b   (bwblkslv.c:272-278)"""
    y = np.zeros(len(perm))
    y[perm] = sla.solve_triangular(L.T, np.asarray(b, dtype=np.float64), lower=False, unit_diagonal=True)
    return y


def ldlsolve_dense(L, d, perm, b):
    """wrapPcg.m:56-59 without dense columns; skipped pivots act as d=1 (deninfac.m:89-94)."""
    ds = np.where(d > 0, d, 1.0)
    return bwsolve_dense(L, perm, fwsolve_dense(L, perm, b) / ds)

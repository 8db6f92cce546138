"""Host-side mirror of the reference MEX interface for the hot path.

Every function has the name, argument order and argument meaning of the MEX
gateway it replaces (MATLAB conventions: 1-based indices carried as doubles,
structs as dicts, sparse matrices as scipy CSC), converts exactly what the
gateway converts, and calls the C ABI of libsedumi_hip.so (tier 1 of
include/sedumi_hip.h).  So a parity test reads like the MATLAB call site:

    ADA = getada1(ADA, A, Ablkjc[:, 2], Aord["lqperm"], d, K["qblkstart"])   # sedumi.m:450
    ADA = getada2(ADA, DAt, Aord, K)                                          # sedumi.m:451
    ADA, absd = getada3(ADA, A, Ablkjc[:, 2], Aord, udsqr, K)                 # sedumi.m:452
    LL, Ld, Lskip, Ladd = blkchol(L, ADA, pars_chol, absd)                    # sedumi.m:458
    y = bwblkslv(L, fwblkslv(L, b) / Ld)                                      # wrapPcg.m:56-59

Errors raise SdmError (the gateways call mexErrMsgTxt).  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import scipy.sparse as sp

from . import capi
from .capi import SdmError, check, f64, i64, pf, pi


def _csc(X):
    X = sp.csc_matrix(X)
    if not X.has_sorted_indices:
        X = X.copy()
        X.sort_indices()
    return X


def _field(s, name, what):
    if name not in s:
        raise SdmError(f"Missing field {what}.{name}.")
    return s[name]


def _same_pattern(X, pr):
    return sp.csc_matrix((pr, X.indices.copy(), X.indptr.copy()), shape=X.shape)


def _perm0(p, m, what):
    p = i64(np.asarray(p, dtype=np.float64).ravel()) - 1
    if p.size != m:
        raise SdmError(f"Size mismatch {what}.")
    return p


def _Kfields(K):
    l = int(np.asarray(K.get("l", 0)).ravel()[0]) if np.size(K.get("l", 0)) else 0
    q = np.asarray(K.get("q", []), dtype=np.float64).ravel()
    if q.size == 1 and q[0] == 0:
        q = q[:0]
    s = np.asarray(K.get("s", []), dtype=np.float64).ravel()
    if s.size == 1 and s[0] == 0:
        s = s[:0]
    rsdpN = int(np.asarray(K["rsdpN"]).ravel()[0]) if "rsdpN" in K else s.size
    return l, i64(q), i64(s), rsdpN


# --------------------------------------------------------------------- ADA'
def getada1(ADA, A, Ajc2, perm, d, blkstart):
    """ADA = getada1(ADA, A, Ajc2, perm, d, blkstart)   (getada1.c:161-261)"""
    ADA, A = _csc(ADA), _csc(A)
    m = A.shape[1]
    if ADA.shape != (m, m):
        raise SdmError("Size mismatch ADA.")
    Ajc2 = i64(np.asarray(Ajc2, dtype=np.float64))
    if Ajc2.size != m:
        raise SdmError("Size mismatch Ajc2.")
    p0 = _perm0(perm, m, "perm")
    dl = f64(_field(d, "l", "d"))
    ddet = f64(_field(d, "det", "d"))
    qb = i64(np.asarray(blkstart, dtype=np.float64)) - 1      # K.qblkstart, 1-based -> 0-based
    lorN = qb.size - 1
    if lorN != ddet.size:
        raise SdmError("Size d.det mismatch")
    out = np.zeros(ADA.indptr[-1], dtype=np.float64)
    jc, ir = i64(ADA.indptr), i64(ADA.indices)
    Ajc, Air, Apr = i64(A.indptr), i64(A.indices), f64(A.data)
    check(capi.lib().sdm_getada1(C.c_int64(m), pi(jc), pi(ir), C.c_int64(A.shape[0]), pi(Ajc), pi(Air), pf(Apr),
                                 pi(Ajc2), pi(p0), C.c_int64(dl.size), pf(dl), C.c_int64(lorN), pf(ddet), pi(qb), pf(out)))
    return _same_pattern(ADA, out)


def getada2(ADA, DAt, Aord, K):
    """ADA = getada2(ADA, DAt, Aord, K)   (getada2.c:127-214)"""
    ADA = _csc(ADA)
    m = ADA.shape[0]
    out = f64(ADA.data).copy()
    _, q, _, _ = _Kfields(K)
    if q.size == 0:
        return _same_pattern(ADA, out)
    Q = _csc(_field(DAt, "q", "DAt"))
    if Q.shape != (q.size, m):
        raise SdmError("Size mismatch DAt.q.")
    qperm = _perm0(_field(Aord, "qperm", "Aord"), m, "Aord.qperm")
    jc, ir = i64(ADA.indptr), i64(ADA.indices)
    Qjc, Qir, Qpr = i64(Q.indptr), i64(Q.indices), f64(Q.data)
    check(capi.lib().sdm_getada2(C.c_int64(m), pi(jc), pi(ir), pf(out), C.c_int64(q.size), pi(Qjc), pi(Qir), pf(Qpr),
                                 pi(qperm)))
    return _same_pattern(ADA, out)


def getada(ADA, A, K, d, DAt):
    """[ADA, absd] = the effect of `absd = getada(A,K,d,DAt)` on the global ADA_sedumi_ (getada.m:13-40; sedumi.m:446-448
    takes this route when sum(K.s)==0).  Python has no MATLAB globals: the global's pattern comes in as `ADA`, its new
    value goes out first.  Entries that MATLAB's sparse() would drop are kept as explicit zeros."""
    ADA, A = _csc(ADA), _csc(A)
    m = A.shape[1]
    if ADA.shape != (m, m):
        raise SdmError("Size mismatch ADA.")
    lpN, q, s, _ = _Kfields(K)
    if np.sum(s) != 0:
        raise SdmError("getada is the path of problems without PSD blocks (sedumi.m:446).")
    dl, ddet = f64(_field(d, "l", "d")), f64(_field(d, "det", "d"))
    if dl.size != lpN or ddet.size != q.size:
        raise SdmError("Size mismatch d.l / d.det.")
    qb = i64(np.asarray(_field(K, "qblkstart", "K"), dtype=np.float64)) - 1
    jc, ir = i64(ADA.indptr), i64(ADA.indices)
    Ajc, Air, Apr = i64(A.indptr), i64(A.indices), f64(A.data)
    if q.size:
        Q = _csc(_field(DAt, "q", "DAt"))
        if Q.shape != (q.size, m):
            raise SdmError("Size mismatch DAt.q.")
        Qjc, Qir, Qpr = i64(Q.indptr), i64(Q.indices), f64(Q.data)
    else:
        Qjc, Qir, Qpr = np.zeros(m + 1, dtype=np.int64), np.zeros(1, dtype=np.int64), np.zeros(1)
        qb = np.zeros(1, dtype=np.int64)
    out = np.zeros(ADA.nnz, dtype=np.float64)
    absd = np.zeros(m, dtype=np.float64)
    check(capi.lib().sdm_getada(C.c_int64(m), pi(jc), pi(ir), C.c_int64(A.shape[0]), pi(Ajc), pi(Air), pf(Apr), C.c_int64(lpN),
                                pf(dl), C.c_int64(q.size), pf(ddet), pi(qb), pi(Qjc), pi(Qir), pf(Qpr), pf(out), pf(absd)))
    return _same_pattern(ADA, out), absd.reshape(-1, 1)


def getada3(ADA, A, Ajc1, Aord, udsqr, K):
    """[ADA, absd] = getada3(ADA, A, Ajc1, Aord, udsqr, K)   (getada3.c:370-569)"""
    ADA, A = _csc(ADA), _csc(A)
    m = A.shape[1]
    if ADA.shape != (m, m):
        raise SdmError("Size mismatch ADA.")
    lpN, q, s, rsdpN = _Kfields(K)
    Ajc1 = i64(np.asarray(Ajc1, dtype=np.float64))
    if Ajc1.size != m:
        raise SdmError("Ajc1 size mismatch")
    sperm = _perm0(_field(Aord, "sperm", "Aord"), m, "Aord.sperm")
    blkstart = np.asarray(_field(K, "blkstart", "K"), dtype=np.float64).ravel()
    if blkstart.size != 2 + q.size + s.size:
        raise SdmError("Size mismatch K.blkstart.")
    psd_start = i64(blkstart[q.size + 1:]) - 1
    ud = f64(udsqr)
    lenud = int(np.sum(s[:rsdpN] ** 2) + 2 * np.sum(s[rsdpN:] ** 2))
    if ud.size != lenud:
        raise SdmError("udsqr size mismatch.")
    Kc, keep = capi.make_cone(lpN, q, s, rsdpN)
    out = f64(ADA.data).copy()
    absd = np.zeros(m, dtype=np.float64)
    jc, ir = i64(ADA.indptr), i64(ADA.indices)
    Ajc, Air, Apr = i64(A.indptr), i64(A.indices), f64(A.data)
    check(capi.lib().sdm_getada3(C.c_int64(m), pi(jc), pi(ir), pf(out), C.c_int64(A.shape[0]), pi(Ajc), pi(Air), pf(Apr),
                                 pi(Ajc1), pi(sperm), pf(ud), C.byref(Kc), pi(psd_start), pf(absd)))
    del keep
    return _same_pattern(ADA, out), absd.reshape(-1, 1)


# ------------------------------------------------------------- factor / solve
def _Lstruct(L, need_values):
    if not isinstance(L, dict):
        raise SdmError("Parameter `L' should be a structure.")
    LL = _field(L, "L", "L")
    if not sp.issparse(LL):
        raise SdmError("L.L should be sparse.")
    LL = _csc(LL)
    m = LL.shape[0]
    if LL.shape != (m, m):
        raise SdmError("Size L.L mismatch.")
    perm = _perm0(_field(L, "perm", "L"), m, "L.perm")
    xs = i64(np.asarray(_field(L, "xsuper", "L"), dtype=np.float64)) - 1
    if xs.size - 1 > m:
        raise SdmError("Size L.xsuper mismatch.")
    return m, LL, i64(LL.indptr), i64(LL.indices), (f64(LL.data) if need_values else None), perm, xs


def blkchol(L, X, pars=None, absd=None):
    """[L.L, L.d, L.skip, L.add] = blkchol(L, X, pars, absd)   (blkchol.c:239-440)"""
    m, LL, Ljc, Lir, _, perm, xs = _Lstruct(L, False)
    X = _csc(X)
    if X.shape != (m, m):
        raise SdmError("P must be square")
    cp = capi.CholPars(1e-12, 5e2, 1e-20)            # blkchol.c:292-294
    if pars is not None:
        if "canceltol" in pars:
            cp.canceltol = float(np.asarray(pars["canceltol"]).ravel()[0])
        if "maxu" in pars:
            cp.maxu = float(np.asarray(pars["maxu"]).ravel()[0])
        if "abstol" in pars:
            cp.abstol = max(float(np.asarray(pars["abstol"]).ravel()[0]), 0.0)
    ab = None
    if pars is not None and absd is not None:          # absd is only read when pars is given (blkchol.c:312-316)
        ab = f64(absd)
        if ab.size != m:
            raise SdmError("absd size mismatch")
    Xjc, Xir, Xpr = i64(X.indptr), i64(X.indices), f64(X.data)
    Lpr = np.zeros(Ljc[-1], dtype=np.float64)
    d = np.zeros(m, dtype=np.float64)
    nskip, nadd = C.c_int64(0), C.c_int64(0)
    sidx, aidx = np.zeros(m, dtype=np.int64), np.zeros(m, dtype=np.int64)
    sval, aval = np.zeros(m, dtype=np.float64), np.zeros(m, dtype=np.float64)
    check(capi.lib().sdm_blkchol(C.c_int64(m), pi(Ljc), pi(Lir), pi(perm), C.c_int64(xs.size - 1), pi(xs), pi(Xjc), pi(Xir),
                                 pf(Xpr), C.byref(cp), pf(ab), pf(Lpr), pf(d), C.byref(nskip), pi(sidx), pf(sval),
                                 C.byref(nadd), pi(aidx), pf(aval)))
    ns, na = nskip.value, nadd.value
    Lout = sp.csc_matrix((Lpr, LL.indices.copy(), LL.indptr.copy()), shape=(m, m))
    skip = sp.csc_matrix((sval[:ns], sidx[:ns], np.array([0, ns])), shape=(m, 1))
    add = sp.csc_matrix((aval[:na], aidx[:na], np.array([0, na])), shape=(m, 1))
    return Lout, d.reshape(-1, 1), skip, add


def _solve(fw, L, b, ysymb):
    m, LL, Ljc, Lir, Lpr, perm, xs = _Lstruct(L, True)
    lib = capi.lib()
    if sp.issparse(b):
        if ysymb is None:
            raise SdmError("fwblkslv requires more inputs in case of sparse b.")
        B, Y = _csc(b), _csc(ysymb)
        if B.shape[0] != m:
            raise SdmError("Size mismatch b.")
        if Y.shape != B.shape:
            raise SdmError("Size mismatch y.")
        n = B.shape[1]
        Bjc, Bir, Bpr = i64(B.indptr), i64(B.indices), f64(B.data)
        Yjc, Yir = i64(Y.indptr), i64(Y.indices)
        Ypr = np.zeros(max(int(Yjc[-1]), 1), dtype=np.float64)
        if fw:
            check(lib.sdm_fwblkslv_sparse(C.c_int64(m), pi(Ljc), pi(Lir), pf(Lpr), pi(perm), C.c_int64(xs.size - 1), pi(xs),
                                          C.c_int64(n), pi(Bjc), pi(Bir), pf(Bpr), pi(Yjc), pi(Yir), pf(Ypr)))
        else:
            check(lib.sdm_bwblkslv_sparse(C.c_int64(m), pi(Ljc), pi(Lir), pf(Lpr), C.c_int64(xs.size - 1), pi(xs),
                                          C.c_int64(n), pi(Bjc), pi(Bir), pf(Bpr), pi(Yjc), pi(Yir), pf(Ypr)))
        return sp.csc_matrix((Ypr[:Yjc[-1]], Y.indices.copy(), Y.indptr.copy()), shape=Y.shape)
    b = np.asarray(b, dtype=np.float64)
    if b.ndim == 1:
        b = b.reshape(-1, 1)
    if b.shape[0] != m:
        raise SdmError("Size mismatch b.")
    n = b.shape[1]
    bf = f64(b)
    y = np.zeros(m * n, dtype=np.float64)
    fn = lib.sdm_fwblkslv if fw else lib.sdm_bwblkslv
    check(fn(C.c_int64(m), pi(Ljc), pi(Lir), pf(Lpr), pi(perm), C.c_int64(xs.size - 1), pi(xs), C.c_int64(n), pf(bf), pf(y)))
    return y.reshape((m, n), order="F")


def fwblkslv(L, b, ysymb=None):
    """y = fwblkslv(L, b [,ysymb]):  y = L.L \\ b(L.perm,:)   (fwblkslv.c:193-320)"""
    return _solve(True, L, b, ysymb)


def bwblkslv(L, b, ysymb=None):
    """y = bwblkslv(L, b [,ysymb]):  y(L.perm,:) = L.L' \\ b   (bwblkslv.c:182-298)"""
    return _solve(False, L, b, ysymb)


# ------------------------------------------------------------------ symbolic
def ordmmdmex(X):
    """perm = ordmmdmex(X): multiple-minimum-degree ordering of spones(X)   (ordmmdmex.c:75-139)"""
    if not sp.issparse(X):
        raise SdmError("Input matrix must be sparse")
    X = _csc(X)
    m = X.shape[0]
    if X.shape != (m, m):
        raise SdmError("X should be square.")
    jc, ir = i64(X.indptr), i64(X.indices)
    perm = np.zeros(m, dtype=np.int64)
    check(capi.lib().sdm_ordmmd(C.c_int64(m), pi(jc), pi(ir), pi(perm)))
    return (perm + 1).astype(np.float64).reshape(-1, 1)


def symfctmex(X, perm):
    """L = symfctmex(X, perm) -> L.{L, perm, xsuper}   (symfctmex.c:127-272)"""
    X = _csc(X)
    m = X.shape[0]
    if X.shape != (m, m):
        raise SdmError("X must be square")
    p0 = _perm0(perm, m, "perm")
    jc, ir = i64(X.indptr), i64(X.indices)
    nsuper, nnzl = C.c_int64(0), C.c_int64(0)
    lib = capi.lib()
    check(lib.sdm_symfct(C.c_int64(m), pi(jc), pi(ir), pi(p0), None, C.byref(nsuper), None, C.byref(nnzl), None, None))
    pout = np.zeros(m, dtype=np.int64)
    xs = np.zeros(nsuper.value + 1, dtype=np.int64)
    Ljc = np.zeros(m + 1, dtype=np.int64)
    Lir = np.zeros(max(nnzl.value, 1), dtype=np.int64)
    check(lib.sdm_symfct(C.c_int64(m), pi(jc), pi(ir), pi(p0), pi(pout), C.byref(nsuper), pi(xs), C.byref(nnzl), pi(Ljc), pi(Lir)))
    LL = sp.csc_matrix((np.ones(nnzl.value), Lir[:nnzl.value], Ljc), shape=(m, m))
    return {"L": LL, "perm": (pout + 1).astype(np.float64).reshape(-1, 1), "xsuper": (xs + 1).astype(np.float64).reshape(-1, 1)}


def choltmpsiz(L):
    """tmpsiz = choltmpsiz(L)   (choltmpsiz.c:110-173)"""
    LL = _csc(_field(L, "L", "L"))
    m = LL.shape[0]
    xs = i64(np.asarray(_field(L, "xsuper", "L"), dtype=np.float64)) - 1
    jc, ir = i64(LL.indptr), i64(LL.indices)
    out = C.c_int64(0)
    check(capi.lib().sdm_choltmpsiz(C.c_int64(m), pi(jc), pi(ir), C.c_int64(xs.size - 1), pi(xs), C.byref(out)))
    return np.array([[float(out.value)]])


def cholsplit(L, cachsz):
    """split = cholsplit(L, cachsz)   (cholsplit.c:118-184)"""
    LL = _csc(_field(L, "L", "L"))
    m = LL.shape[0]
    xs = i64(np.asarray(_field(L, "xsuper", "L"), dtype=np.float64)) - 1
    jc = i64(LL.indptr)
    split = np.zeros(max(m, 1), dtype=np.int64)
    check(capi.lib().sdm_cholsplit(C.c_int64(m), pi(jc), C.c_int64(xs.size - 1), pi(xs), C.c_double(float(np.asarray(cachsz).ravel()[0])), pi(split)))
    return split[:m].astype(np.float64).reshape(-1, 1)


def incorder(At, Ajc1=None, ifirst=None):
    """[perm, dz] = incorder(At [, Ajc1, ifirst])   (incorder.c:216-330).  perm 1-based m x 1; dz N x m sparse whose
    columns list their row subscripts in the order incorder introduced them (NOT sorted: the order is data for
    finsymbden / getada3 -- the returned csc_matrix keeps it as long as nobody sorts its indices)."""
    if not sp.issparse(At):
        raise SdmError("At must be a sparse matrix.")
    At = sp.csc_matrix(At)
    N, m = At.shape
    jc, ir = i64(At.indptr), i64(At.indices)
    first = 0
    a1 = None
    if Ajc1 is not None:
        a1 = i64(np.asarray(Ajc1, dtype=np.float64))
        if a1.size < m:
            raise SdmError("Ajc1 size mismatch")
        first = int(np.asarray(ifirst).ravel()[0]) - 1
    nin = int(np.sum(jc[1:] - (a1[:m] if a1 is not None else jc[:-1])))
    perm = np.zeros(max(m, 1), dtype=np.int64)
    dzjc = np.zeros(m + 1, dtype=np.int64)
    dzir = np.zeros(max(min(N - first, nin), 1), dtype=np.int64)
    check(capi.lib().sdm_incorder(C.c_int64(N), C.c_int64(m), pi(jc), pi(ir), pi(a1) if a1 is not None else None, C.c_int64(first),
                                  pi(perm), pi(dzjc), pi(dzir)))
    nnz = int(dzjc[m])
    dz = sp.csc_matrix((np.ones(nnz), dzir[:nnz].astype(np.int32), dzjc.astype(np.int32)), shape=(N, m))
    return (perm[:m] + 1).astype(np.float64).reshape(-1, 1), dz


def symbchol(ADA, cachsz=512):
    """symbchol.m:62-83 on top of the entry points above (MATLAB glue, host only)."""
    ADA = _csc(ADA)
    m = ADA.shape[0]
    if ADA.nnz < m * m:
        L = symfctmex(ADA, ordmmdmex(ADA))
    else:
        L = {"perm": np.arange(1, m + 1, dtype=np.float64).reshape(-1, 1), "L": sp.csc_matrix(np.tril(np.ones((m, m)))),
             "xsuper": np.array([[1.0], [m + 1.0]])}
    L["tmpsiz"] = choltmpsiz(L)
    L["split"] = cholsplit(L, cachsz)
    return L


# ------------------------------------------------------------- dense columns
def symbfwblk(L, b):
    """x = symbfwblk(L, b): sparsity pattern (values all 1) of L.L \\ b(L.perm,:)   (symbfwblk.c:270-377)"""
    m, LL, Ljc, Lir, _, perm, xs = _Lstruct(L, False)
    if not sp.issparse(b):
        raise SdmError("B must be sparse")
    B = _csc(b)
    if B.shape[0] != m:
        raise SdmError("L.perm size mismatches B")
    n = B.shape[1]
    Bjc, Bir = i64(B.indptr), i64(B.indices)
    Xjc = np.zeros(n + 1, dtype=np.int64)
    lib = capi.lib()
    args = (C.c_int64(m), pi(Ljc), pi(Lir), pi(perm), C.c_int64(xs.size - 1), pi(xs), C.c_int64(n), pi(Bjc), pi(Bir))
    check(lib.sdm_symbfwblk(*args, pi(Xjc), None))
    Xir = np.zeros(max(int(Xjc[-1]), 1), dtype=np.int64)
    check(lib.sdm_symbfwblk(*args, pi(Xjc), pi(Xir)))
    nnz = int(Xjc[-1])
    return sp.csc_matrix((np.ones(nnz), Xir[:nnz], Xjc), shape=(m, n))


def finsymbden(LAD, perm, dz, firstq):
    """Lden = finsymbden(LAD, perm, dz, firstq) -> dict(LAD, perm, dz, first)   (finsymbden.c:112-214)"""
    LAD, dz = _csc(LAD), sp.csc_matrix(dz)                 # dz: row order inside the columns is data, never sorted
    m, n = LAD.shape
    p = i64(np.asarray(perm, dtype=np.float64)) - 1
    nperm = p.size
    if dz.shape != (m, nperm):
        raise SdmError("dz size mismatch")
    LADjc, LADir, dzjc, dzir = i64(LAD.indptr), i64(LAD.indices), i64(dz.indptr), i64(dz.indices)
    po, jo, fo = np.zeros(max(n, 1), dtype=np.int64), np.zeros(n + 1, dtype=np.int64), np.zeros(max(n, 1), dtype=np.int64)
    check(capi.lib().sdm_finsymbden(C.c_int64(m), C.c_int64(n), pi(LADjc), pi(LADir), C.c_int64(nperm), pi(p), pi(dzjc),
                                    pi(dzir), C.c_int64(int(np.asarray(firstq).ravel()[0]) - 1), pi(po), pi(jo), pi(fo)))
    # the reference keeps dz.ir / dz.pr and only swaps the column pointers (finsymbden.c:196-199)
    nz = int(jo[n])
    dznew = sp.csc_matrix((np.asarray(dz.data[:nz], dtype=np.float64), dzir[:nz].copy(), jo.copy()), shape=(m, n))
    return {"LAD": LAD, "perm": (po[:n] + 1).astype(np.float64).reshape(-1, 1), "dz": dznew,
            "first": (fo[:n] + 1).astype(np.float64).reshape(-1, 1)}


def _dz(dz):
    """Lden.dz / Lsymb.dz: either a scipy CSC matrix or the explicit {'jc','ir'} form produced by finsymbden."""
    if isinstance(dz, dict):
        return i64(dz["jc"]), i64(dz["ir"])
    dz = sp.csc_matrix(dz)
    return i64(dz.indptr), i64(dz.indices)


def dpr1fact(x, d, Lsymb, smult, maxu):
    """[Lden, Ld] = dpr1fact(x, d, Lsymb, smult, maxu)   (dpr1fact.c:630-848)
    Lden = dict(betajc (1-based), beta, p, pivperm (0-based, as the reference), dopiv)."""
    X = _csc(x)
    m, n = X.shape
    lab = f64(d).copy()
    if lab.size != m:
        raise SdmError("Size mismatch d.")
    sm = f64(smult)
    if sm.size != n:
        raise SdmError("Size mismatch smult.")
    dzjc, dzir = _dz(_field(Lsymb, "dz", "Lsymb"))
    colperm = _perm0(_field(Lsymb, "perm", "Lsymb"), n, "Lsymb.perm")
    first = i64(np.asarray(_field(Lsymb, "first", "Lsymb"), dtype=np.float64)) - 1
    if first.size != n:
        raise SdmError("Size mismatch Lsymb.first.")
    pnnz = int(dzjc[1:n + 1].sum())
    betajc = np.zeros(n + 1, dtype=np.int64)
    beta, p = np.zeros(max(pnnz, 1)), np.zeros(max(pnnz, 1))
    pivperm, dopiv = np.zeros(max(pnnz, 1), dtype=np.int64), np.zeros(max(n, 1), dtype=np.int64)
    npp = C.c_int64(0)
    Xjc, Xir, Xpr = i64(X.indptr), i64(X.indices), f64(X.data)
    check(capi.lib().sdm_dpr1fact(C.c_int64(m), C.c_int64(n), pi(Xjc), pi(Xir), pf(Xpr), pf(lab), pi(dzjc), pi(dzir), pi(colperm),
                                  pi(first), pf(sm), C.c_double(float(np.asarray(maxu).ravel()[0])), pi(betajc), pf(beta), pf(p),
                                  pi(pivperm), C.byref(npp), pi(dopiv)))
    Lden = {"betajc": (betajc + 1).astype(np.float64).reshape(-1, 1), "beta": beta[:int(betajc[n])].reshape(-1, 1),
            "p": p[:pnnz].reshape(-1, 1), "pivperm": pivperm[:npp.value].astype(np.float64).reshape(-1, 1),
            "dopiv": dopiv[:n].astype(np.float64).reshape(-1, 1)}
    return Lden, lab.reshape(-1, 1)


def _pr1(fw, Lden, b):
    b = np.asarray(b, dtype=np.float64)
    if sp.issparse(b):
        raise SdmError("b should be full")
    if b.ndim == 1:
        b = b.reshape(-1, 1)
    m, nrhs = b.shape
    betajc = i64(np.asarray(_field(Lden, "betajc", "Lden"), dtype=np.float64)) - 1
    nden = betajc.size - 1
    if nden == 0:
        return b.copy()                                   # fwdpr1.c:132-135: no dense columns
    beta, p = f64(_field(Lden, "beta", "Lden")), f64(_field(Lden, "p", "Lden"))
    dopiv = i64(np.asarray(_field(Lden, "dopiv", "Lden"), dtype=np.float64))
    if dopiv.size != nden:
        raise SdmError("Size mismatch Lden.dopiv.")
    pivperm = i64(np.asarray(_field(Lden, "pivperm", "Lden"), dtype=np.float64))
    dzjc, dzir = _dz(_field(Lden, "dz", "Lden"))
    if beta.size != betajc[-1]:
        raise SdmError("Size mismatch Lden.beta.")
    y = np.zeros(m * nrhs)
    fn = capi.lib().sdm_fwdpr1 if fw else capi.lib().sdm_bwdpr1
    pv = pivperm if pivperm.size else np.zeros(1, dtype=np.int64)
    check(fn(C.c_int64(m), C.c_int64(nrhs), C.c_int64(nden), pi(dzjc), pi(dzir), pi(betajc), pf(beta if beta.size else np.zeros(1)),
             pf(p if p.size else np.zeros(1)), pi(pv), C.c_int64(pivperm.size), pi(dopiv), pf(f64(b)), pf(y)))
    return y.reshape((m, nrhs), order="F")


def fwdpr1(Lden, b):
    """y = fwdpr1(Lden, b): y = PROD_k L(p_k, beta_k) \\ b   (fwdpr1.c:101-202)"""
    return _pr1(True, Lden, b)


def bwdpr1(Lden, b):
    """y = bwdpr1(Lden, b): y = (PROD_k L(p_k, beta_k))' \\ b   (bwdpr1.c:170-275)"""
    return _pr1(False, Lden, b)


def _dense_struct(dense, nlor):
    nl = int(np.asarray(_field(dense, "l", "dense")).ravel()[0])
    q = i64(np.asarray(_field(dense, "q", "dense"), dtype=np.float64)) - 1
    cols = np.asarray(_field(dense, "cols", "dense"), dtype=np.float64).ravel()
    nden = cols.size - nl - q.size
    if nden < 0:
        raise SdmError("dense.q size mismatch.")
    dencols = i64(cols[nl + q.size:]) - 1
    if q.size and (q.min() < 0 or q.max() >= nlor):
        raise SdmError("dense.q out of range")
    return nl, q, dencols, nden


def adendotd(dense, d, sparAd, Ablk, blkstart):
    """Ad = adendotd(dense, d, sparAd, Ablk, blkstart)   (adendotd.c:135-232; getDAtm.m:45)"""
    d1, d2 = f64(_field(d, "q1", "d")), f64(_field(d, "q2", "d"))
    nl, q, dencols, nden = _dense_struct(dense, d1.size)
    A = _csc(_field(dense, "A", "dense"))
    m = A.shape[0]
    if A.shape[1] - nl != q.size + nden:
        raise SdmError("dense.A size mismatch")
    S, B = _csc(sparAd), _csc(Ablk)
    if S.shape[1] != q.size or B.shape[1] != q.size:
        raise SdmError("Size mismatch sparAD")
    bs = np.asarray(blkstart, dtype=np.float64).ravel()
    if bs.size != d1.size + 1:
        raise SdmError("blkstart size mismatch")
    firstQ = int(bs[0]) - 1
    blkend = i64(bs[q + 1]) - 1 if q.size else np.zeros(1, dtype=np.int64)
    out = np.zeros(max(B.nnz, 1), dtype=np.float64)
    adenjc = i64(A.indptr[nl:])
    check(capi.lib().sdm_adendotd(C.c_int64(m), C.c_int64(q.size), C.c_int64(nden), pi(i64(B.indptr)), pi(i64(B.indices)), pf(out),
                                  pi(i64(S.indptr)), pi(i64(S.indices)), pf(f64(S.data)), pi(adenjc), pi(i64(A.indices)), pf(f64(A.data)),
                                  pf(d1), pf(d2), C.c_int64(firstQ), pi(q if q.size else np.zeros(1, dtype=np.int64)),
                                  pi(dencols if nden else np.zeros(1, dtype=np.int64)), pi(blkend)))
    return _same_pattern(B, out[:B.nnz])


def adenscale(dense, d, blkstart):
    """smult = adenscale(dense, d, blkstart)   (adenscale.c:87-160; deninfac.m:61)"""
    detd = f64(_field(d, "det", "d"))
    nl, q, dencols, nden = _dense_struct(dense, detd.size)
    bs = np.asarray(blkstart, dtype=np.float64).ravel()
    if bs.size != detd.size + 1:
        raise SdmError("blkstart size mismatch")
    blkend = i64(bs[q + 1]) - 1 if q.size else np.zeros(1, dtype=np.int64)
    out = np.zeros(max(nden, 1), dtype=np.float64)
    check(capi.lib().sdm_adenscale(C.c_int64(q.size), C.c_int64(nden), pf(detd), pi(q if q.size else np.zeros(1, dtype=np.int64)),
                                   pi(dencols if nden else np.zeros(1, dtype=np.int64)), pi(blkend), pf(out)))
    return out[:nden].reshape(-1, 1)


# ------------------------------------------------------------- next row
def invcholfac(u, K, perm=None):
    """y = invcholfac(u, K, perm): y(perm,perm) = u'*u per PSD block, u upper triangular   (invcholfac.c:59-168).
    u: lenud vector (blocks column-major, Hermitian blocks [Re; Im]); perm: 1-based, concatenated per block (d.perm of
    the scaling), empty / None = no permutation.  Returns the lenud x 1 `udsqr` argument of getada3 (sedumi.m:452)."""
    lpN, q, s, rsdpN = _Kfields(K)
    lenud = int(np.sum(s[:rsdpN] ** 2) + 2 * np.sum(s[rsdpN:] ** 2))
    u = f64(u)
    if u.size != lenud:
        raise SdmError("u size mismatch")
    Kc, keep = capi.make_cone(lpN, q, s, rsdpN)
    y = np.zeros(lenud, dtype=np.float64)
    pp = None
    if perm is not None and np.size(perm) > 0:
        p1 = np.asarray(perm, dtype=np.float64).ravel()
        if p1.size != int(np.sum(s)):
            raise SdmError("perm size mismatch")
        pp = i64(p1 - 1)                       # local to each block (invcholfac.c:129-131), 1-based
    check(capi.lib().sdm_invcholfac(C.byref(Kc), pf(u), pi(pp) if pp is not None else None, pf(y)))
    del keep
    return y.reshape(-1, 1)

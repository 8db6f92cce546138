"""sedumi_amd/driver/conemex.py -- the NON-hot-path MEX functions SeDuMi's interior-point loop calls, on numpy / LAPACK.

`NativeMex.call(name, nlhs, *args)` answers the same calls `sedumi.m` and the `.m` files around it make to the reference's MEX binaries, so that the
loop of `sedumi_amd/driver/loop.py` runs without MATLAB, Octave or the compiled reference.  Hot-path names (getada1/2/3, blkchol, fwblkslv, bwblkslv,
invcholfac, the symbolic ones) are routed to this package's own library (`sedumi_amd.mex`); everything else is stated here:

  Lorentz helpers     ddot (ddot.c:66-160, :170-308), qblkmul (qblkmul.c:60-116)
  PSD helpers         vecsym (vecsym.c:50-125), qrK (qrK.c:76-296), psdframeit (psdframeit.c:63-105), psdinvjmul (psdinvjmul.c:87-160),
                      urotorder (urotorder.c:66-180, :312-490), givensrot (givensrot.c:54-168), sqrtinv (sqrtinv.c:56-90)
  step control        iswnbr (iswnbr.c:66-215), quadadd (quadadd.c:57-83)
  set-up              partitA (partitA.c:73-146), extractA (extractA.c), findblks (findblks.c), sortnnz (sortnnz.c)

Where a reference routine hands an OPAQUE array from one MEX to another, the array here has this module's own content -- chosen for LAPACK, not for
the reference's compact forms:

  * the "frame" of a PSD block (qrK's first output, consumed by psdframeit / psdinvjmul): the unitary factor Qb of  x = Qb R  stored EXPLICITLY (n x n,
    Hermitian blocks as [Re; Im]) -- the reference stores n - 1 Householder vectors and their betas (qrK.c:76-120);
  * the rotations of urotorder (its outputs gjc, g, consumed by givensrot): the unitary G of the re-pivoting stored explicitly (gjc = block offsets
    into g) -- the reference stores Givens pairs (urotorder.c:79-180).  The re-pivoting itself is LAPACK's QR with column pivoting whenever the
    reference's stability test (urotorder.c:110-122: max |u(k, later columns)|^2 > maxu^2 * d_k) fails, the identity otherwise.

What the loop and the hot path see of these -- u upper triangular with a positive diagonal, `perm`, `x = Qb' diag(lab) Qb` -- is the reference's
(tests/test_native_driver.py compares every routine with the compiled reference on the calls of real solves, through those invariants)."""
from __future__ import annotations

import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp


def _v(x):
    return np.asarray(x, dtype=np.float64).ravel()


def _col(x):
    return np.asarray(x, dtype=np.float64).reshape(-1, 1)


class ConeK:
    """conepars (sdmauxCone.c:48-134): the fields of K the routines below read."""

    def __init__(self, K):
        g = lambda k, dflt=0.0: K.get(k, dflt)
        self.l = int(_v(g("l"))[0]) if np.size(g("l")) else 0
        self.q = _v(g("q", [])).astype(np.int64)
        self.s = _v(g("s", [])).astype(np.int64)
        self.rsdpN = int(_v(g("rsdpN", self.s.size))[0]) if np.size(g("rsdpN", self.s.size)) else self.s.size
        self.nr, self.nh = self.s[:self.rsdpN], self.s[self.rsdpN:]
        self.rLen, self.hLen = int(self.nr.sum()), int(self.nh.sum())
        self.rDim, self.hDim = int((self.nr ** 2).sum()), int(2 * (self.nh ** 2).sum())
        self.lenud = self.rDim + self.hDim
        self.lendiag = self.l + 2 * self.q.size + self.rLen + self.hLen

    def blocks(self, x, extra=0):
        """the PSD blocks of a length-lenud vector (+ `extra` trailing entries per Hermitian block... not used) as complex / real matrices"""
        x = _v(x)
        out, o = [], 0
        for k, n in enumerate(self.s):
            if k < self.rsdpN:
                out.append(x[o:o + n * n].reshape(n, n, order="F").copy()); o += n * n
            else:
                re = x[o:o + n * n].reshape(n, n, order="F"); im = x[o + n * n:o + 2 * n * n].reshape(n, n, order="F")
                out.append(re + 1j * im); o += 2 * n * n
        return out

    def pack(self, mats):
        parts = []
        for k, M in enumerate(mats):
            if k < self.rsdpN:
                parts.append(np.real(M).ravel(order="F"))
            else:
                parts.append(np.real(M).ravel(order="F")); parts.append(np.imag(M).ravel(order="F"))
        return np.concatenate(parts) if parts else np.zeros(0)

    def psd_lab(self, lab):
        lab = _v(lab)
        return lab if lab.size == self.rLen + self.hLen else lab[self.l + 2 * self.q.size:]

    def split_lab(self, lab):
        lab = self.psd_lab(lab)
        o, out = 0, []
        for n in self.s:
            out.append(lab[o:o + n]); o += n
        return out


# ------------------------------------------------------------------------------------------------ Lorentz
def _lorentz_window(v, b0, bend, nblk):
    """the norm-bound part of a vector given at full length, as [trace; norm-bound] or bare (ddot.c:197-214, qblkmul.c:84-95)"""
    v = _v(v)
    qdim = bend - b0
    if v.size == qdim:
        return v
    if v.size == nblk + qdim:
        return v[nblk:]
    return v[b0:bend]


def ddot(d, X, blkstart, Xblkjc=None):
    """y(k, j) = d[k]' x_j[k] for every Lorentz block k (ddot.c:66-160)"""
    bs = _v(blkstart).astype(np.int64) - 1
    nblk = bs.size - 1
    dd = _lorentz_window(d, bs[0], bs[-1], nblk)
    if sp.issparse(X):
        # an entry for every (block, column) that holds a stored nonzero of the column -- also when the sum is zero (spddotxj, ddot.c:108-160)
        Xc = sp.csc_matrix(X)
        m = Xc.shape[1]
        rows = sp.coo_matrix(Xc[bs[0]:bs[-1], :])
        blk = np.searchsorted(bs[1:] - bs[0], rows.row, side="right")
        keys = rows.col.astype(np.int64) * max(nblk, 1) + blk
        uniq, inv = np.unique(keys, return_inverse=True)
        vals = np.bincount(inv, weights=dd[rows.row] * rows.data, minlength=uniq.size)
        Y = sp.csc_matrix((vals, (uniq % max(nblk, 1), uniq // max(nblk, 1))), shape=(nblk, m))
        Y.sort_indices()
        return Y
    Xa = np.asarray(X, dtype=np.float64)
    Xa = Xa.reshape(-1, 1) if Xa.ndim < 2 else Xa
    cols = []
    for j in range(Xa.shape[1]):
        xj = _lorentz_window(Xa[:, j], bs[0], bs[-1], nblk)
        cols.append(np.add.reduceat(dd * xj, (bs[:-1] - bs[0])) if nblk else np.zeros(0))
        if nblk:                                               # (empty blocks: reduceat repeats the next entry)
            empty = np.diff(bs) == 0
            cols[-1][empty] = 0.0
    return np.stack(cols, axis=1) if cols else np.zeros((nblk, 0))


def qblkmul(mu, d, blkstart):
    """y[k] = mu(k) d[k] on the Lorentz norm-bound blocks (qblkmul.c:60-116)"""
    bs = _v(blkstart).astype(np.int64) - 1
    nblk = bs.size - 1
    dd = _lorentz_window(d, bs[0], bs[-1], nblk)
    return _col(np.repeat(_v(mu), np.diff(bs)) * dd)


# ------------------------------------------------------------------------------------------------ PSD
def vecsym(x, K):
    """(X + X') / 2 on every PSD block, Hermitian blocks [Re; Im] -> [sym; skew] (vecsym.c:50-125); the rest of x unchanged"""
    cK = ConeK(K)
    x = _v(x).copy()
    off = x.size - cK.lenud
    mats = cK.blocks(x[off:])
    x[off:] = cK.pack([(M + M.conj().T) / 2 for M in mats])
    return _col(x)


def _frames(cK, frms):
    """explicit unitary factors out of this module's frame array"""
    return cK.blocks(_v(frms))


def qrK(x, K, nlhs=1):
    """x = Qb R per PSD block, R upper triangular with a real non-negative diagonal (qrK.c:76-120: Householder; :142-245: Hermitian, where the
    reference also rotates the diagonal real).  Returns (frame, R); the frame is Qb itself (module header)."""
    cK = ConeK(K)
    Qs, Rs = [], []
    for M in cK.blocks(x):
        Q, R = np.linalg.qr(M)
        ph = np.diagonal(R).copy()
        ph = np.where(np.abs(ph) > 0, ph / np.where(np.abs(ph) > 0, np.abs(ph), 1.0), 1.0)
        Q = Q * ph                       # Q diag(ph)
        R = (R.T * np.conj(ph)).T        # diag(conj ph) R
        R = np.triu(R)
        if np.iscomplexobj(R):
            R[np.diag_indices_from(R)] = np.real(np.diagonal(R))
        Qs.append(Q); Rs.append(R)
    q = _col(cK.pack(Qs))
    return (q, _col(cK.pack(Rs))) if nlhs > 1 else q


def psdframeit(lab, frms, K):
    """X = Qb' diag(lab) Qb (psdframeit.c:63-105)"""
    cK = ConeK(K)
    out = [(Q.conj().T * l) @ Q for Q, l in zip(_frames(cK, frms), cK.split_lab(lab))]
    out = [(X + X.conj().T) / 2 for X in out]
    return _col(cK.pack(out))


def psdinvjmul(xlab, xfrm, y, K):
    """z with  x jmul z = y  for  x = Qb' diag(xlab) Qb:  Q z Q' = 2 (Q y Q') ./ (x_i + x_j)  (psdinvjmul.c:87-160)"""
    cK = ConeK(K)
    y = _v(y)
    ymats = cK.blocks(y[y.size - cK.lenud:])
    out = []
    for Q, l, Y in zip(_frames(cK, xfrm), cK.split_lab(xlab), ymats):
        Y = (Y + Y.conj().T) / 2                     # (the reference reads one triangle of y: y is symmetric where it comes from)
        T = Q @ Y @ Q.conj().T
        T = 2.0 * T / (l[:, None] + l[None, :])
        Z = Q.conj().T @ T @ Q
        out.append((Z + Z.conj().T) / 2)
    return _col(cK.pack(out))


def sqrtinv(q, vlab, K):
    """y = (Q / diag(sqrt(vlab)))'  so that  y' y = inv(Q diag(vlab) Q')  (sqrtinv.c:56-90)"""
    cK = ConeK(K)
    out = [(Q / np.sqrt(l)).conj().T for Q, l in zip(cK.blocks(q), cK.split_lab(vlab))]
    return _col(cK.pack(out))


def urotorder(u, K, maxu, permIN=None):
    """Stable re-pivoting of the upper-triangular factor (urotorder.c:66-180): U_out = (G U_in)(:, p) upper triangular, perm_out = perm_in(p).
    Returns (u_out with its lower triangle mirrored as the reference does, perm (1-based doubles), gjc, g): g holds the unitary G of every block
    explicitly, gjc the offsets of the blocks in g (module header)."""
    cK = ConeK(K)
    maxusqr = float(maxu) ** 2
    pin = _v(permIN) if permIN is not None and np.size(permIN) else None
    us, perms, gs, gjc = [], [], [], []
    o = 0
    for k, M in enumerate(cK.blocks(u)):
        n = M.shape[0]
        T = np.triu(M)
        # the reference's test, column by column in the current order: is |u(k, j)|^2 <= maxu^2 * sum_{i >= k} |u(i, k)|^2 for all j > k ?
        a2 = np.abs(T) ** 2
        dk = np.diagonal(a2)                                  # (upper triangular: column k has nothing below its diagonal)
        rowmax = np.array([a2[r, r + 1:].max() if r + 1 < n else 0.0 for r in range(n)])
        if np.all(rowmax <= maxusqr * dk):
            G, R, p = np.eye(n, dtype=T.dtype), T, np.arange(n)
        else:
            Q, R, p = sla.qr(T, pivoting=True)
            ph = np.diagonal(R).copy()
            ph = np.where(np.abs(ph) > 0, ph / np.where(np.abs(ph) > 0, np.abs(ph), 1.0), 1.0)
            Q = Q * ph
            R = np.triu((R.T * np.conj(ph)).T)
            G = Q.conj().T
        if np.iscomplexobj(R):
            R[np.diag_indices_from(R)] = np.real(np.diagonal(R))
        Rm = R + np.triu(R, 1).conj().T                      # triu2sym / triu2herm (urotorder.c:404, :437)
        us.append(Rm)
        perms.append(pin[o:o + n][p] if pin is not None else 1.0 + p)
        gjc.append(float(sum(g.size for g in gs)))
        gs.append(np.real(G).ravel(order="F"))
        if k >= cK.rsdpN:
            gs.append(np.imag(G).ravel(order="F"))
        o += n
    g = np.concatenate(gs) if gs else np.zeros(0)
    return (_col(cK.pack(us)), _col(np.concatenate(perms) if perms else np.zeros(0)), _col(np.asarray(gjc)), _col(g))


def givensrot(gjc, g, x, K):
    """Y = G X per PSD block (givensrot.c:54-88), G as urotorder above left it"""
    cK = ConeK(K)
    gjc = _v(gjc).astype(np.int64)
    g = _v(g)
    out = []
    for k, X in enumerate(cK.blocks(x)):
        n = X.shape[0]
        o = gjc[k]
        G = g[o:o + n * n].reshape(n, n, order="F")
        if k >= cK.rsdpN:
            G = G + 1j * g[o + n * n:o + 2 * n * n].reshape(n, n, order="F")
        out.append(G @ X)
    return _col(cK.pack(out))


# ------------------------------------------------------------------------------------------------ step control
def iswnbr(vSQR, thetaSQR):
    """proximity to the wide region C(theta) of Sturm-Zhang and the projection (1 - alpha) max(h, v) onto it (iswnbr.c:66-215).  The reference
    grows the set T = {j: w_j < h^2} in data order; T is the fixed point of  h^2 = sum_{j not in T} w_j / (r - |T|),  r = n / theta^2,  which is
    found here on the sorted w."""
    w = _v(vSQR)
    n = w.size
    th = float(np.asarray(thetaSQR).ravel()[0])
    gap = float(w.sum())
    r = n / th
    if 1.0 - th <= 1e-8:
        hs = float(w.max()); h = np.sqrt(hs)
        sumdifw = float(np.sum(hs - w)); sumdifv = float(np.sum(h - np.sqrt(w)))
    else:
        ws = np.sort(w)
        pre = np.concatenate(([0.0], np.cumsum(ws)))
        t = 0
        hs = gap / r
        while t < n and ws[t] < hs:
            if ws[t] <= 0.0:
                return 1e100, 0.0, 0.0
            t += 1
            hs = (gap - pre[t]) / (r - t)
        h = np.sqrt(hs)
        sumdifw = float(np.sum(hs - ws[:t])); sumdifv = float(np.sum(h - np.sqrt(ws[:t])))
    alpha = sumdifv / (r * h)
    dsq = alpha * (2.0 - alpha) - (1.0 - alpha) ** 2 * sumdifw / gap
    return float(np.sqrt(r * dsq)) if dsq >= 0 else float("nan"), float(h), float(alpha)


def quadadd(xhi, xlo, y):
    """(zhi, zlo) = (xhi + xlo) + y in doubled precision (quadadd.c:57-83), elementwise"""
    xhi, xlo, y = _v(xhi).copy(), _v(xlo).copy(), _v(y)
    big = np.abs(y) > np.abs(xhi)
    zhi_a = y + xhi
    zlo_a = xlo + (xhi - (zhi_a - y))
    zlo1 = xlo + y
    xlo_b = xlo - (zlo1 - y)
    zhi_b = xhi + zlo1
    zlo_b = xlo_b + (zlo1 - (zhi_b - xhi))
    return _col(np.where(big, zhi_a, zhi_b)), _col(np.where(big, zlo_a, zlo_b))


# ------------------------------------------------------------------------------------------------ set-up
def partitA(A, mainblks):
    """Ablkjc(j, :) = offsets into column j of the first nonzero at or beyond every main-block start (partitA.c:73-146); 0-based, doubles"""
    A = sp.csc_matrix(A); A.sort_indices()
    mb = _v(mainblks).astype(np.int64) - 1
    m = A.shape[1]
    out = np.zeros((m, mb.size))
    for j in range(m):
        rows = A.indices[A.indptr[j]:A.indptr[j + 1]]
        out[j, :] = A.indptr[j] + np.searchsorted(rows, mb, side="left")
    return out


def _block_range(A, Ablkjc, blk0, blk1):
    """(start, end) offsets per column of the main blocks blk0 .. blk1-1 (0 = from the column's start, empty / beyond = to its end)"""
    A = sp.csc_matrix(A)
    m = A.shape[1]
    Ab = np.asarray(Ablkjc, dtype=np.float64).reshape(m, -1).astype(np.int64)
    b0 = int(_v(blk0)[0]) if np.size(blk0) else 0
    b1 = int(_v(blk1)[0]) if np.size(blk1) else Ab.shape[1] + 1
    lo = A.indptr[:-1] if b0 <= 0 else Ab[:, b0 - 1]
    hi = A.indptr[1:] if b1 > Ab.shape[1] else Ab[:, b1 - 1]
    return lo, hi


def extractA(A, Ablkjc, blk0, blk1, blkstart0, blkstart1):
    """the rows blkstart0 .. blkstart1-1 (1-based) of A = its main blocks blk0 .. blk1-1, as a (blkstart1 - blkstart0) x m matrix (extractA.c)"""
    A = sp.csc_matrix(A)
    i0, i1 = int(_v(blkstart0)[0]) - 1, int(_v(blkstart1)[0]) - 1
    E = sp.csc_matrix(A[i0:i1, :]); E.sort_indices()
    return E


def findblks(A, Ablkjc, blk0, blk1, blkstart):
    """pattern (values 1) of which sub-blocks (blkstart: their 1-based first rows, + the end) of the main blocks blk0 .. blk1-1 hold a nonzero of
    every column (findblks.c)"""
    A = sp.csc_matrix(A)
    bs = _v(blkstart).astype(np.int64) - 1
    nblk = max(bs.size - 1, 0)
    m = A.shape[1]
    if nblk == 0:
        return sp.csc_matrix((0, m))
    sub = sp.coo_matrix(A[bs[0]:bs[-1], :])
    blk = np.searchsorted(bs[1:] - bs[0], sub.row, side="right")
    F = sp.csc_matrix((np.ones(sub.nnz), (blk, sub.col)), shape=(nblk, m))
    F.sum_duplicates(); F.data[:] = 1.0; F.sort_indices()
    return F


def sortnnz(At, Ajc1=None, Ajc2=None):
    """constraints ordered by the number of nonzeros between the offsets Ajc1 (default: the column's start) and Ajc2 (default: its end), fewest
    first, ties in their original order (sortnnz.c); 1-based doubles"""
    At = sp.csc_matrix(At)
    lo = _v(Ajc1).astype(np.int64) if Ajc1 is not None and np.size(Ajc1) else At.indptr[:-1]
    hi = _v(Ajc2).astype(np.int64) if Ajc2 is not None and np.size(Ajc2) else At.indptr[1:]
    return _col(np.argsort(hi - lo, kind="stable") + 1.0)


def unwrap_raw(args):
    """arguments for this package's own functions: RawSparse wrappers (sedumi_amd.mexhost: "hand this sparse matrix to a MEX as stored") taken off,
    at the top level and inside structs"""
    from sedumi_amd.mexhost import RawSparse
    one = lambda a: a.X if isinstance(a, RawSparse) else ({k: (v.X if isinstance(v, RawSparse) else v) for k, v in a.items()} if isinstance(a, dict) else a)
    return tuple(one(a) for a in args)


# ------------------------------------------------------------------------------------------------ the host
class NativeMex:
    """`.call(name, nlhs, *args)` like a MEX host: the cone algebra above, the hot path and the symbolic analysis through this package's library."""
    error = RuntimeError

    def __init__(self):
        from sedumi_amd import mex
        self._mex = mex
        self._native = {"ddot": ddot, "qblkmul": qblkmul, "vecsym": vecsym, "psdframeit": psdframeit, "psdinvjmul": psdinvjmul, "sqrtinv": sqrtinv,
                        "givensrot": givensrot, "partitA": partitA, "extractA": extractA, "findblks": findblks, "sortnnz": sortnnz}

    def call(self, name, nlhs, *args):
        args = unwrap_raw(args)
        if name in self._native:
            return self._native[name](*args)
        if name == "qrK":
            return qrK(*args, nlhs=nlhs)
        if name == "urotorder":
            return urotorder(*args)[:max(nlhs, 1)] if nlhs > 1 else urotorder(*args)[0]
        if name == "iswnbr":
            d, h, a = iswnbr(*args)
            return (np.array([[d]]), np.array([[h]]), np.array([[a]]))[:max(nlhs, 1)] if nlhs > 1 else np.array([[d]])
        if name == "quadadd":
            hi, lo = quadadd(*args)
            return (hi, lo) if nlhs > 1 else hi
        m = self._mex
        if name in ("ordmmdmex", "symfctmex", "choltmpsiz", "cholsplit", "incorder", "getada1", "getada2", "getada3", "blkchol", "fwblkslv", "bwblkslv", "invcholfac", "symbfwblk", "finsymbden", "dpr1fact", "fwdpr1", "bwdpr1"):
            return getattr(m, name)(*args)
        raise KeyError("NativeMex: no such MEX function: " + name)

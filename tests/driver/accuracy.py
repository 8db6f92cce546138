"""tests/driver/accuracy.py -- TEST INFRASTRUCTURE (drives the oracle): accuracy of the three stages of the hot path on
the scalings of a REAL run (the reference-hot-path run of sedumi_loop.py), judged against extended precision
(numpy longdouble, 64-bit mantissa) instead of against each other:

  ADA'   : max |triu(ADA - ADA_ld)| / max |ADA_ld|                 reference getada1/2/3 (getada.m) vs library
  factor : max |P ADA P' - L D L'| / max |ADA|  (same ADA' in)      reference blkchol vs library
  solves : |L y - r(perm)| / (|L| |y|),  |L' x - z| / (|L'| |x|)  (same L in)   reference fwblkslv / bwblkslv vs library

Why: late in a run the PSD scaling D = U'U is so ill-conditioned (arch0: cond 1e5 at iteration 10, 1e12 at 30) that the
REFERENCE's own ADA' is only accurate to ~4e-11 of its largest entry.  Two correct double-precision evaluations then
differ by that much, and the 1e-10 tolerance against the reference is a statement about conditioning, not about either
code.  Against extended precision the question "is the library as accurate as the reference" has a clean answer.
"""
import numpy as np
import scipy.sparse as sp

from . import sedumi_loop as sl

LD = np.longdouble


def ada_longdouble(Sx, d, DAt):
    """ADA' = A' * blockdiag(d.l, Lorentz, D (x) D) * A in extended precision (getada1.c / getada2.c / getada3.c semantics)."""
    Kx = Sx["K"]
    A = sp.csc_matrix(Sx["A"]).toarray().astype(LD)
    m = A.shape[1]
    l = int(Kx["l"])
    out = (A[:l, :].T * sl.vec(d["l"]).astype(LD)) @ A[:l, :]
    nq = Kx["q"].size
    if nq:
        mb = Kx["mainblks"].ravel().astype(int) - 1
        qb = Kx["qblkstart"].ravel().astype(int) - 1
        det = sl.vec(d["det"]).astype(LD)
        sv = np.concatenate((-det, np.zeros(mb[2] - mb[1], dtype=LD)))
        for i in range(nq):
            sv[nq + qb[i] - qb[0]:nq + qb[i + 1] - qb[0]] = det[i]
        Aq = A[mb[0]:mb[2], :]
        out = out + (Aq.T * sv) @ Aq
        q1, q2 = sl.vec(d["q1"]).astype(LD), sl.vec(d["q2"]).astype(LD)
        Q = q1[:, None] * A[mb[0]:mb[1], :]
        for i in range(nq):
            Q[i, :] += q2[qb[i] - qb[0]:qb[i + 1] - qb[0]] @ A[qb[i]:qb[i + 1], :]
        out = out + Q.T @ Q
    xi = int(Kx["lq"])
    u = sl.vec(d["u"]).astype(LD)
    ui, pi_ = 0, 0
    for n in Kx["s"].ravel().astype(int):
        U = np.triu(u[ui:ui + n * n].reshape(n, n, order="F")); ui += n * n
        D = U.T @ U
        if np.size(d["perm"]):
            PP = sl.vec(d["perm"])[pi_:pi_ + n].astype(int) - 1; pi_ += n
            Dp = np.zeros_like(D); Dp[np.ix_(PP, PP)] = D; D = Dp
        T = np.zeros((n * n, m), dtype=LD)
        for j in range(m):
            Aj = A[xi:xi + n * n, j].reshape(n, n, order="F")
            Aj = (Aj + Aj.T) / 2
            if np.any(Aj):
                T[:, j] = (D @ Aj @ D).ravel(order="F")
        Asym = A[xi:xi + n * n, :].copy()
        for j in range(m):
            Aj = Asym[:, j].reshape(n, n, order="F"); Asym[:, j] = ((Aj + Aj.T) / 2).ravel(order="F")
        out = out + Asym.T @ T
        xi += n * n
    return out


class Probe(sl.MexShapedHot):
    """Hot path of a run that follows the reference MEX and, at the iterations listed, measures both paths (see module doc)."""
    name = "probe"

    def __init__(self, ref, lib, iters, verbose=False):
        self.ref, self.lib, self.iters, self.verbose = ref, lib, list(iters), verbose
        self.it, self.records = 0, []

    def factor(self, Sx, d, DAt, L, pars):
        self.it += 1
        ref, lib = self.ref, self.lib
        Lr = ref.factor(Sx, d, DAt, L, pars)
        if self.it in self.iters:
            m = Sx["A"].shape[1]
            ADAr = ref.last_ADA.toarray()
            ADAl, absd_l = lib.form(Sx, d, DAt)
            ADAl = ADAl.toarray()
            X = ada_longdouble(Sx, d, DAt)
            sc = np.abs(X).max()
            ea = [float(np.abs(np.triu(M).astype(LD) - np.triu(X)).max() / sc) for M in (ADAr, ADAl)]
            # --- factor: same ADA' (the reference's) into both
            absd = sl.col(np.diag(ADAr)) if not np.sum(Sx["K"]["s"]) else ref.form(Sx, d, DAt)[1]
            perm = sl.vec(L["perm"]).astype(int) - 1
            ef, facs = [], []
            for hot in (ref, lib):
                LL, Ld, Lskip, Ladd = hot.blkchol(L, ref.last_ADA, pars, absd)
                Lm = sp.csc_matrix(LL).toarray().astype(LD)
                Lm = np.tril(Lm, -1) + np.eye(m, dtype=LD)
                R = (Lm * sl.vec(Ld).astype(LD)) @ Lm.T - ADAr[np.ix_(perm, perm)].astype(LD)
                ef.append(float(np.abs(R).max() / np.abs(ADAr).max()))
                facs.append((LL, sl.vec(Ld), int(sp.csc_matrix(Lskip).nnz)))
            # --- solves: same factor (the reference's) into both
            Lx = dict(L); Lx["L"] = facs[0][0]
            Lm = np.tril(sp.csc_matrix(facs[0][0]).toarray(), -1).astype(LD) + np.eye(m, dtype=LD)
            r = np.random.default_rng(self.it).standard_normal(m)
            es = []
            for hot in (ref, lib):
                y = sl.vec(hot.fw(Lx, r)); z = y / facs[0][1]; x = sl.vec(hot.bw(Lx, z))
                e1 = np.abs(Lm @ y.astype(LD) - r[perm].astype(LD)).max() / (np.abs(Lm).sum(axis=1).max() * np.abs(y).max())
                e2 = np.abs(Lm.T @ x[perm].astype(LD) - z.astype(LD)).max() / (np.abs(Lm).sum(axis=0).max() * np.abs(x).max())
                es.append((float(e1), float(e2)))
            rec = {"iter": self.it, "dcond": float(Lr["d"].max() / Lr["d"].min()), "psdcond": dcond(Sx["K"], d), "ada": tuple(ea), "factor": tuple(ef),
                   "skips": (facs[0][2], facs[1][2]), "fw": (es[0][0], es[1][0]), "bw": (es[0][1], es[1][1])}
            self.records.append(rec)
            if self.verbose:
                print("iter %2d  max L.d/min L.d %.1e  cond(D_psd) %s | ADA' err ref %.1e lib %.1e | factor residual ref %.1e lib %.1e (skips %d/%d) | fw residual ref %.1e lib %.1e | bw residual ref %.1e lib %.1e"
                      % (self.it, rec["dcond"], rec["psdcond"], ea[0], ea[1], ef[0], ef[1], facs[0][2], facs[1][2], es[0][0], es[1][0], es[0][1], es[1][1]), flush=True)
        return Lr

    def fw(self, L, r):
        return self.ref.fw(L, r)

    def bw(self, L, r):
        return self.ref.bw(L, r)


def dcond(Kx, d):
    out, ui = [], 0
    u = sl.vec(d["u"])
    for n in Kx["s"].ravel().astype(int):
        U = np.triu(u[ui:ui + n * n].reshape(n, n, order="F")); ui += n * n
        out.append("%.0e" % (np.linalg.cond(U) ** 2))
    return ",".join(out) if out else "-"


def probe(S, iters, lib, verbose=False):
    """Run S (a sedumi_loop.Sedumi) along the reference hot path up to max(iters); returns the records of `iters`."""
    P = Probe(sl.RefHot(S.G), lib, iters, verbose)
    S.hot = P
    S.pars["maxiter"] = max(iters)
    S.solve()
    return P.records

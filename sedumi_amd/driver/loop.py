"""sedumi_amd/driver/loop.py -- SeDuMi's interior-point loop without MATLAB.

A Python restatement of sedumi.m:428-571 and the .m files it calls (file and line cited per function), far enough to solve the
reference's own example problems, on top of

    the hot path      PlanHot()  this package's library through its resident plan: problem, scaling, ADA', factor and solves stay in HBM
                      HipHot()   the same library MEX call by MEX call (sedumi_amd.mex)
    everything else   a MEX host `ref` with `.call(name, nlhs, *args)`: by default sedumi_amd.driver.conemex.NativeMex -- the cone
                      algebra (qrK, psdframeit, psdinvjmul, urotorder, givensrot, sqrtinv, iswnbr, vecsym, ddot, qblkmul, quadadd) on
                      numpy / LAPACK; nothing of the reference is needed at run time.

    from sedumi_amd.driver import solve
    r = solve(At, b, c, K)            # r["x"], r["y"], r["cx"], r["by"], r["iter"], r["feasratio"], r["rows"] (the iteration log)

The tests run the same loop with the compiled reference as the MEX host and / or as the hot path (tests/driver/sedumi_loop.py:
RefHot, ShimHot, ShadowHot) and compare the logs.  Restrictions: real or Hermitian data, no free variables / rotated cones (pretransfo_real), dense LP columns handled
(sedumi.m:356-364, deninfac.m:58-76) but no dense Lorentz blocks, pars = checkpars.m defaults.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import scipy.sparse as sp

from . import glue as gl


def col(x):
    return np.asarray(x, dtype=np.float64).reshape(-1, 1)


def vec(x):
    return np.asarray(x.todense() if sp.issparse(x) else x, dtype=np.float64).ravel()


# ----------------------------------------------------------------------------------------------- pretransfo
def pretransfo(At, b, c, K):
    """pretransfo.m for LP / Lorentz / real PSD (glue.pretransfo_real) and, restated here, for Hermitian PSD blocks
    (K.scomplex) with complex constraints (K.ycomplex): pretransfo.m:113-148 (field checks), :250-259 (complex
    constraints become pairs of real ones), :310-320 (which blocks are Hermitian), :455-481 (Hermitian coefficients folded
    into the lower triangle, [Re; Im] storage), :484-515 (K, the x0 row, At = real(QR At)).  No free variables, rotated
    cones, K.xcomplex or diagonal-block detection for complex data (the example problem quantum.mat needs none)."""
    sc = np.asarray(K.get("scomplex", []), dtype=np.int64).ravel()
    yc = np.asarray(K.get("ycomplex", []), dtype=np.int64).ravel()
    cplx = np.iscomplexobj(At.toarray() if sp.issparse(At) and At.dtype.kind == "c" else At) or np.iscomplexobj(c) or np.iscomplexobj(b)
    if sc.size == 0 and yc.size == 0 and not cplx:
        return gl.pretransfo_real(At, b, c, K)
    Kl = int(np.asarray(K.get("l", 0)).ravel()[0]) if np.size(K.get("l", 0)) else 0
    Kq = np.asarray(K.get("q", []), dtype=np.int64).ravel(); Kq = Kq[Kq > 0]
    Ks = np.asarray(K.get("s", []), dtype=np.int64).ravel(); Ks = Ks[Ks > 0]
    assert Kq.size == 0, "Lorentz cones together with complex data are not restated"
    N = Kl + int((Ks ** 2).sum())
    At = sp.csc_matrix(At, dtype=np.complex128)
    if At.shape[0] != N and At.shape[1] == N:
        At = sp.csc_matrix(At.conj().T)                                # pretransfo.m:173 (At')
    assert At.shape[0] == N
    b = np.asarray(b.todense() if sp.issparse(b) else b, dtype=np.complex128).ravel()
    c = np.asarray(c.todense() if sp.issparse(c) else c, dtype=np.complex128).ravel()
    yc = np.unique(yc)
    if yc.size:                                                        # pretransfo.m:254-256
        b = np.concatenate((b.real, b[yc - 1].imag))
        At = sp.hstack((At, 1j * At[:, yc - 1])).tocsc()
    else:
        b = b.real
    scplx = np.zeros(Ks.size, dtype=bool); scplx[np.unique(sc) - 1] = True
    order = np.concatenate((np.nonzero(~scplx)[0], np.nonzero(scplx)[0]))   # real blocks first, then Hermitian (pretransfo.m:486)
    jstrt_all = Kl + np.concatenate(([0], np.cumsum(Ks[:-1] ** 2)))
    ii, jj, vv = [np.arange(Kl)], [np.arange(Kl)], [np.ones(Kl, dtype=np.complex128)]
    off = Kl
    for k in order:
        n, j0 = int(Ks[k]), int(jstrt_all[k])
        if not scplx[k]:                                               # pretransfo.m:434-452
            idx = np.arange(n * n); cols = idx // n; rows = idx - n * cols
            ii.append(off + np.maximum(rows, cols) + np.minimum(rows, cols) * n); jj.append(j0 + idx); vv.append(np.ones(n * n, dtype=np.complex128))
            off += n * n
        else:                                                          # pretransfo.m:455-481
            bnd = np.arange(2 * n * n); cols = bnd // n; rows = bnd - n * cols
            imgv = cols >= n
            cols = cols - imgv * n
            indxs = np.maximum(rows, cols) + np.minimum(rows, cols) * n + imgv * (n * n) + off
            vals = 1 + imgv * (-1 + 1j * (1 - 2 * (rows > cols)))
            keep = (~imgv) | (rows != cols)
            ii.append(indxs[keep]); jj.append((rows + cols * n + j0)[keep]); vv.append(vals[keep].astype(np.complex128))
            off += 2 * n * n
    KN = off + 1                                                       # + the artificial (x0, z0) row in front
    QR = sp.csc_matrix((np.concatenate(vv), (np.concatenate(ii) + 1, np.concatenate(jj))), shape=(KN, N))
    At2 = sp.csc_matrix((QR @ At).real); At2.eliminate_zeros(); At2.sort_indices()
    c2 = np.asarray(QR @ c).real.ravel()
    from sedumi_amd import problem
    Kint = problem.make_K(Kl + 1, [], Ks[~scplx], Ks[scplx])
    assert int(Kint["N"]) == KN
    return At2, b, c2, Kint


# ----------------------------------------------------------------------------------------------- parameters
def default_pars():
    """checkpars.m:43-193"""
    return {"alg": 2, "beta": 0.5, "theta": 0.25, "stepdif": 2, "w": np.array([1.0, 1.0]), "mu": 1.0, "eps": 1e-8, "bigeps": 1e-3,
            "maxiter": 150, "denq": 0.75, "denf": 10.0,
            "chol": {"skip": 1.0, "abstol": 1e-20, "canceltol": 1e-12, "maxu": 5e5, "maxuden": 5e2},
            "cg": {"qprec": 1, "restol": 5e-3, "stagtol": 5e-14, "maxiter": 49, "refine": 1}}


# ----------------------------------------------------------------------------------------------- hot paths
def _skipfix(Ld, skip, absd, perm, pars):
    """deninfac.m:87-94: a skipped pivot whose (updated) d is still at most its threshold acts as 1"""
    if skip.size:
        perm0 = vec(perm).astype(int) - 1
        dtol = np.maximum(pars["canceltol"] * vec(absd)[perm0[skip]], pars["abstol"])
        Ld[skip[Ld[skip] <= dtol]] = 1.0
    return Ld


class MexShapedHot:
    """sedumi.m:442-463 for a hot path that is called MEX by MEX: ADA', blkchol, deninfac.m:58-94 (dense LP columns: sparfwslv + dpr1fact)."""

    def mexcall(self, name, nlhs, *args):                           # the MEX host of this hot path (subclasses)
        raise NotImplementedError

    def factor(self, S, d, DAt, L, pars, den=None):
        ADA, absd = self.form(S, d, DAt)                            # sedumi.m:446-452
        self.last_ADA = ADA
        LL, Ld, Lskip, Ladd = self.blkchol(L, ADA, pars, absd)      # sedumi.m:458
        L = dict(L)
        L["L"] = LL
        Ld = vec(Ld).copy()
        L["den"] = None
        if den is not None:                                         # deninfac.m:58-76 (dense LP columns)
            sym = den["sym"]
            LAD = self.mexcall("fwblkslv", 1, L, den["A"], sym["LAD"])                     # sparfwslv.m:55-57
            Lden, Ld2 = self.mexcall("dpr1fact", 2, LAD, col(Ld), den["symstruct"], col(vec(d["l"])[den["rows"]]), float(pars["maxuden"]))
            Lden = dict(Lden); Lden["dz"], Lden["first"], Lden["perm"] = den["symstruct"]["dz"], sym["first"], sym["perm"]
            L["den"], Ld = Lden, vec(Ld2).copy()
        skip = sp.csc_matrix(Lskip).nonzero()[0]                    # deninfac.m:87-94
        Ld = _skipfix(Ld, skip, absd, L["perm"], pars)
        L["d"], L["skip"], L["add"] = Ld, Lskip, Ladd
        L["nskip"], L["nadd"] = int(sp.csc_matrix(Lskip).nnz), int(sp.csc_matrix(Ladd).nnz)
        return L

    def fwdpr1(self, L, b):                                         # wrapPcg.m:56
        return b if L.get("den") is None else vec(self.mexcall("fwdpr1", 1, L["den"], col(b)))

    def bwdpr1(self, L, b):                                         # wrapPcg.m:59
        return b if L.get("den") is None else vec(self.mexcall("bwdpr1", 1, L["den"], col(b)))


class HipHot(MexShapedHot):
    """This repository's library behind the same calls (sedumi_amd.mex mirrors the MEX signatures)."""
    name = "sedumi_amd"

    def mexcall(self, name, nlhs, *args):
        from sedumi_amd import mex
        from .conemex import unwrap_raw
        return getattr(mex, name)(*unwrap_raw(args))

    def form(self, S, d, DAt):
        from sedumi_amd import mex
        K = S["K"]
        if np.sum(K["s"]) == 0:
            ADA, absd = mex.getada(S["ADA"], S["A"], K, d, DAt)
            return ADA, absd
        ADA = mex.getada1(S["ADA"], S["A"], S["Ablkjc"][:, 2], S["Aord"]["lqperm"], d, K["qblkstart"])
        ADA = mex.getada2(ADA, DAt, S["Aord"], K)
        ud = mex.invcholfac(d["u"], K, d["perm"] if np.size(d["perm"]) else None)
        return mex.getada3(ADA, S["A"], S["Ablkjc"][:, 2], S["Aord"], ud, K)

    def blkchol(self, L, ADA, pars, absd):
        from sedumi_amd import mex
        return mex.blkchol(L, ADA, pars, absd)

    def fw(self, L, r):
        from sedumi_amd import mex
        return mex.fwblkslv(L, col(r))

    def bw(self, L, r):
        from sedumi_amd import mex
        return mex.bwblkslv(L, col(r))


class PlanHot:
    """The resident tier (sedumi_amd.plan.Plan): the problem, the scaling, ADA', the factor and the solves live in HBM;
    per iteration the loop uploads d.{l,det,q1,q2,u} (+ d.perm), and per solve one right-hand side."""
    name = "sedumi_amd.plan"

    def __init__(self, device=0, device_ops=True):
        self.device, self.plan = device, None
        self._pcg, self._d, self._want_ops = False, None, device_ops

    def factor(self, S, d, DAt, L, pars, den=None):
        from sedumi_amd.plan import Plan
        K = S["K"]
        if self.plan is None:
            self.plan = Plan(self.device)
            self.plan.set_chol(S["L"], S["ADA"])
            if den is not None:                                     # dense LP columns: the resident dense-column unit (sdm_plan_set_dense / _deninfac)
                self.plan.set_dense(den["symstruct_plain"])
                self.plan.upload("ad", np.asarray(sp.csc_matrix(den["A"]).todense()).ravel(order="F"))
            self.plan.set_ada(S["A"], S["Ablkjc"], K, S["DAt"]["q"] if K["q"].size else None)
            self._N = int(S["A"].shape[0])
            ks = K["s"].ravel().astype(int); nr = int(K["rsdpN"])
            self._lenud = int(np.sum(ks[:nr] ** 2) + 2 * np.sum(ks[nr:] ** 2))
            self.plan.pcg_init()                                    # work vectors "xN" / "psd" (no dense columns in this restatement)
            self._pcg = True
        pl = self.plan
        self._d = d
        pl.upload("dl", d["l"]); pl.upload("ddet", d["det"])
        if K["q"].size:
            pl.upload("q1", d["q1"]); pl.upload("q2", d["q2"])
            pl.getdatq()                                            # getDAtm.m:39-44
        if np.sum(K["s"]) > 0:
            pl.upload("u", d["u"])
            pl.invcholfac(d["perm"] if np.size(d["perm"]) else None)   # sedumi.m:452
        pl.getada()                                                 # sedumi.m:446-452
        pl.blkchol(pars, True)                                      # sedumi.m:458
        (si, _), (ai, _) = pl.pivots()
        pat = pl.ADA_pattern
        self.last_ADA = sp.csc_matrix((pl.download("ada"), pat.indices, pat.indptr), shape=pat.shape)
        L = dict(L)
        L["den"] = None
        if den is not None:                                         # deninfac.m:58-76: LAD = L \ Ad and the product-form factors, on the device
            pl.deninfac(vec(d["l"])[den["rows"]], float(pars["maxuden"]))
            Lden, Ld = pl.lden()
            Lden = dict(Lden); Lden["dz"], Lden["first"], Lden["perm"] = den["symstruct_plain"]["dz"], den["sym"]["first"], den["sym"]["perm"]
            L["den"] = Lden
            Ld = _skipfix(Ld, si, pl.download("absd"), L["perm"], pars)
        else:
            Ld = pl.download("d")
            Ld[si] = np.where(Ld[si] <= 0.0, 1.0, Ld[si])           # deninfac.m:87-94 (skipped pivots carry d = 0)
        L["d"], L["nskip"], L["nadd"] = Ld, int(si.size), int(ai.size)
        return L

    def fwdpr1(self, L, b):                                         # wrapPcg.m:56 (the library's product-form kernels)
        from sedumi_amd import mex
        return b if L.get("den") is None else vec(mex.fwdpr1(L["den"], col(b)))

    def bwdpr1(self, L, b):                                         # wrapPcg.m:59
        from sedumi_amd import mex
        return b if L.get("den") is None else vec(mex.bwdpr1(L["den"], col(b)))

    def fw(self, L, r):
        self.plan.upload("rhs", vec(r)); self.plan.fwsolve()
        return self.plan.download("y")

    def bw(self, L, r):
        self.plan.upload("rhs", vec(r)); self.plan.bwsolve()
        return self.plan.download("y")

    # ---- the operators wrapPcg.m:47-66 / loopPcg.m apply around the solves, on the device: the whole-solve tests
    # (reference-held optimal values of examples/test_sedumi.m:22-28, the reference-hot-path log) pin them through every PCG step
    def has_ops(self, d=None):
        return self._want_ops and self.plan is not None and self._pcg and (d is None or d is self._d)
        # (with dense columns the loop keeps Amul on the host: Sedumi._ops)

    def Amul(self, x, transp):
        pl = self.plan
        if not transp:                                             # Amul.m:46  At' * x
            pl.upload("xN", vec(x)); pl.amul(0)
            return pl.download("rhs")
        pl.upload("y", vec(x)); pl.amul(1)                          # Amul.m:48  At * y
        return pl.download("xN", self._N)

    def Amul1_vecsym(self, p):
        """vecsym(Amul(At, dense, p, 1), K)  (wrapPcg.m:65, loopPcg.m) as one device sequence"""
        pl = self.plan
        pl.upload("y", vec(p)); pl.amul(1); pl.vecsym()
        return pl.download("xN", self._N)

    def psdscale(self, d, x, transp):
        """psdscale(d, x, K[, transp]) with the d of the last factor() (its d.u and pivot order are resident)"""
        pl = self.plan
        pl.upload("xN", vec(x)); pl.psdscale(1 if transp else 0, bool(np.size(d["perm"])))
        return pl.download("psd", self._lenud)


class Cone:
    """The .m files of the Jordan-algebra layer, restated; their MEX parts go through the MEX host `ref` (sedumi_amd.driver.conemex.NativeMex by default)."""

    def __init__(self, ref, K):
        self.ref, self.K = ref, K
        self.l = int(K["l"])
        self.q = K["q"].ravel().astype(int)
        self.s = K["s"].ravel().astype(int)
        self.nq = self.q.size
        mb = K["mainblks"].ravel().astype(int)
        self.i1, self.i2, self.i3 = mb[0] - 1, mb[1] - 1, mb[2] - 1     # 0-based starts: Lorentz trace, norm-bound, PSD
        self.lq = int(K["lq"])
        self.N = int(K["N"])
        self.nreal = int(np.asarray(K.get("rsdpN", self.s.size)).ravel()[0])
        self.lenud = int(np.sum(self.s[:self.nreal] ** 2) + 2 * np.sum(self.s[self.nreal:] ** 2))
        self.qb = K["qblkstart"]

    # ---- Lorentz helpers (MEX: ddot.c, qblkmul.c)
    def ddot(self, x2, y):
        if self.nq == 0:
            return np.zeros(0)
        return vec(self.ref.call("ddot", 1, col(x2), col(y), self.qb))

    def qblkmul(self, mu, d):
        if self.nq == 0:
            return np.zeros(0)
        return vec(self.ref.call("qblkmul", 1, col(mu), col(d), self.qb))

    def tdet(self, x):                                             # tdet.m
        if self.nq == 0:
            return np.zeros(0)
        return x[self.i1:self.i2] ** 2 - self.ddot(x[self.i2:self.i3], x)

    def qframeit(self, lab, frmq):                                 # qframeit.m
        n = self.nq
        if lab.size > 2 * n:
            lab = lab[self.l:self.l + 2 * n]
        return np.concatenate(((lab[:n] + lab[n:]) / np.sqrt(2), self.qblkmul(lab[n:] - lab[:n], frmq)))

    def frameit(self, lab, frmq, frms):                            # frameit.m
        ps = vec(self.ref.call("psdframeit", 1, col(lab), col(frms), self.K)) if self.s.size else np.zeros(0)
        return np.concatenate((lab[:self.l], self.qframeit(lab, frmq) if self.nq else np.zeros(0), ps))

    def qjmul(self, x, y):                                         # qjmul.m (full-length arguments)
        if self.nq == 0:
            return np.zeros(0)
        i1, i2, i3 = self.i1, self.i2, self.i3
        z1 = x[i1:i2] * y[i1:i2] + self.ddot(x[i2:i3], y)
        return np.concatenate((z1, self.qblkmul(x[i1:i2], y) + self.qblkmul(y[i1:i2], x))) / np.sqrt(2)

    def qinvjmul(self, labx, frmx, b):                             # qinvjmul.m
        n = self.nq
        if n == 0:
            return np.zeros(0)
        if labx.size > 2 * n:
            labx = labx[self.l:self.l + 2 * n]
        detx = labx[:n] * labx[n:]
        x = self.qframeit(labx, frmx)
        i1, i2 = self.i1, self.i2
        y1 = x[:n] * b[i1:i2] - self.ddot(x[n:], b)
        y1 = y1 / (np.sqrt(2) * detx)
        return np.concatenate((y1, self.qblkmul(np.sqrt(2) / x[:n], b) - self.qblkmul(y1 / x[:n], x[n:])))

    def asmDxq(self, d, x, ddotx=None, want_t=False):              # asmDxq.m
        if self.nq == 0:
            return (np.zeros(0), np.zeros(0)) if want_t else np.zeros(0)
        if x.size >= self.lq:
            t = x[self.i1:self.i2]
        else:
            t = x[:self.nq]
        if ddotx is None:
            ddotx = d["q1"] * t + self.ddot(d["q2"], x)
        x1 = t
        t = (ddotx + t * d["auxdet"]) / d["auxtr"]
        sdet = np.sqrt(d["det"])
        y = np.concatenate((t * d["auxdet"] - sdet * x1, self.qblkmul(sdet, x)))
        if want_t:
            return y, t
        return y + np.concatenate((t * d["q1"], self.qblkmul(t, d["q2"])))

    # ---- PSD helpers (psdscale.m, psdinvscale.m, psdfactor.m, psdeig.m, psdjmul.m, triumtriu.m, minpsdeig.m); Hermitian blocks
    # (the last len(K.s) - K.rsdpN ones) are stored [Re; Im] and handled as complex matrices here, X' = conjugate transpose
    def _blocks(self, x):
        xi = x.size - self.lenud
        for k, n in enumerate(self.s):
            if k < self.nreal:
                yield x[xi:xi + n * n].reshape(n, n, order="F"), n
                xi += n * n
            else:
                yield (x[xi:xi + n * n] + 1j * x[xi + n * n:xi + 2 * n * n]).reshape(n, n, order="F"), n
                xi += 2 * n * n

    def _pack(self, mats, zero_imag_diag=False):
        out = []
        for k, M in enumerate(mats):
            if k < self.nreal:
                out.append(np.real(M).ravel(order="F"))
            else:
                Im = np.imag(M).copy()
                if zero_imag_diag:
                    np.fill_diagonal(Im, 0.0)
                out.append(np.real(M).ravel(order="F")); out.append(Im.ravel(order="F"))
        return np.concatenate(out) if out else np.zeros(0)

    def psdscale(self, ud, x, transp=False):
        if not self.s.size:
            return np.zeros(0)
        if isinstance(ud, dict):
            perm = ud["perm"] if np.size(ud["perm"]) else None
            u = ud["u"]
        else:
            perm, u = None, ud
        out, pi_ = [], 0
        for (TT, n), (XX, _) in zip(self._blocks(u), self._blocks(x)):
            TT = np.triu(TT) if transp else np.tril(TT)
            if perm is not None and not transp:
                PP = perm[pi_:pi_ + n].astype(int) - 1; pi_ += n
                XX = XX[np.ix_(PP, PP)]
            Y = TT.conj().T @ XX @ TT
            if perm is not None and transp:
                PP = perm[pi_:pi_ + n].astype(int) - 1; pi_ += n
                Z = np.zeros_like(Y); Z[np.ix_(PP, PP)] = Y; Y = Z
            out.append(Y)
        return self._pack(out, zero_imag_diag=True)

    def psdinvscale(self, ud, x):
        import scipy.linalg as sl
        out = []
        for (TT, n), (XX, _) in zip(self._blocks(ud), self._blocks(x)):
            TT = np.triu(TT)
            W = sl.solve_triangular(TT.conj(), XX.T, lower=False).T     # XX / TT'
            out.append(sl.solve_triangular(TT, W, lower=False))
        return self._pack(out, zero_imag_diag=True)

    def psdfactor(self, x):
        out = []
        for XX, n in self._blocks(x):
            try:
                Lc = np.linalg.cholesky(XX)                            # chol(XX,'lower')
            except np.linalg.LinAlgError:                              # `return` with the remaining blocks of ux still zero
                ux = np.zeros(self.lenud)
                done = self._pack(out)
                ux[:done.size] = done
                return ux, False
            out.append(Lc + np.tril(Lc, -1).conj().T)
        return self._pack(out), True

    def psdeig(self, x, want_q=False):
        labs, qs = [], []
        for XX, n in self._blocks(x):
            XX = XX + XX.conj().T
            if want_q:
                w, Q = np.linalg.eigh(XX)
                qs.append(Q)
            else:
                w = np.linalg.eigvalsh(XX)
            labs.append(0.5 * w)
        lab = np.concatenate(labs) if labs else np.zeros(0)
        return (lab, self._pack(qs)) if want_q else lab

    def psdjmul(self, x, y):
        out = []
        for (XX, n), (YY, _) in zip(self._blocks(x), self._blocks(y)):
            ZZ = XX @ YY
            out.append(0.5 * (ZZ + ZZ.conj().T))
        return self._pack(out)

    def triumtriu(self, x, y):
        out = []
        for (XX, n), (YY, _) in zip(self._blocks(x), self._blocks(y)):
            ZZ = np.triu(XX) @ np.triu(YY)
            out.append(ZZ + np.triu(ZZ, 1).conj().T)
        return self._pack(out)

    def minpsdeig(self, x):
        return min(np.linalg.eigvalsh(XX + XX.conj().T).min() for XX, n in self._blocks(x)) / 2

    def psdinvjmul(self, lab, frms, b):
        if not self.s.size:
            return np.zeros(0)
        return vec(self.ref.call("psdinvjmul", 1, col(lab), col(frms), col(b), self.K))

    def vecsym(self, x):
        return vec(self.ref.call("vecsym", 1, col(x), self.K))

    def eyeK(self):                                                # eyeK.m (internal K)
        x = np.zeros(self.N)
        x[:self.l] = 1.0
        x[self.l:self.l + self.nq] = np.sqrt(2.0)
        xi = self.lq
        for k, n in enumerate(self.s):
            x[xi:xi + n * n:n + 1] = 1.0
            xi += n * n * (1 if k < self.nreal else 2)
        return x

    def maxeigK(self, x):                                          # maxeigK.m (used by the Farkas test only)
        vals = [x[:self.l].max()] if self.l else []
        if self.nq:
            nrm = np.sqrt(np.maximum(self.ddot(x[self.i2:self.i3], x), 0.0))
            vals.append(((x[self.i1:self.i2] + nrm) / np.sqrt(2)).max())
        for XX, n in self._blocks(x):
            vals.append(np.linalg.eigvalsh(XX + XX.conj().T).max() / 2)
        return max(vals)


# ----------------------------------------------------------------------------------------------- the solver
class Sedumi:
    def __init__(self, At, b, c, K, hot=None, G=None, pars=None, internal=False):
        """(At, b, c, K) as the user passes them to sedumi.m, or -- internal=True -- already through pretransfo.m
        (the golden fixtures store the problems that way)."""
        self.G = G or gl.Glue()                                         # (default: this package's own MEX host, sedumi_amd.driver.conemex)
        self.ref = self.G.ref
        self.pars = pars or default_pars()
        if not internal:
            At, b, c, K = pretransfo(At, b, c, K)                       # sedumi.m:261
        self.A, self.b, self.c, self.K = sp.csc_matrix(At), vec(b), vec(c), K
        self.S = self.G.setup(self.A, K, self.pars["denq"], self.pars["denf"])     # sedumi.m:356-392
        self.den = None
        dense = self.S["dense"]
        if len(dense["cols"]):
            # dense COLUMNS of A = dense variables (rows of At), sedumi.m:356-364: taken out of ADA' and brought back as a product of rank-1
            # factors (symbcholden.m:43-55, deninfac.m:58-76, wrapPcg.m:56-59).  LP variables only; a dense Lorentz block (getdense.m:58-60,
            # adendotd / adenscale) is not restated
            if len(dense["q"]) or int(dense["l"]) != len(dense["cols"]):
                raise NotImplementedError("dense Lorentz blocks are outside this restatement (dense LP columns are handled)")
            sym = self.G.symbcholden(self.S["L"], dense)
            self.den = {"rows": np.asarray(dense["cols"], dtype=np.int64) - 1, "A": sp.csc_matrix(dense["A"]), "sym": sym,
                        "symstruct": {"dz": self.G.raw(sym["dz"]), "perm": sym["perm"], "first": sym["first"]},
                        "symstruct_plain": {"LAD": sym["LAD"], "dz": sym["dz"], "perm": sym["perm"], "first": sym["first"]}}
        self.A = sp.csc_matrix(self.S["A"])
        self.cone = Cone(self.ref, K)
        self.hot = hot or PlanHot()

    # Amul.m:43-56
    def _ops(self, d=None):
        """the hot path's own Amul / vecsym / psdscale (PlanHot: on the device), when it has them for this scaling"""
        h = self.hot
        return h if self.den is None and getattr(h, "has_ops", None) and h.has_ops(d) else None

    def Amul(self, x, transp=0):
        h = self._ops()
        if h is not None:
            return vec(h.Amul(x, transp))
        if not transp:
            y = vec(self.A.T @ x)
            return y if self.den is None else y + vec(self.den["A"] @ vec(x)[self.den["rows"]])      # Amul.m:52
        y = vec(self.A @ x)
        if self.den is not None:
            y[self.den["rows"]] = vec(self.den["A"].T @ x)                                           # Amul.m:54
        return y

    def Amul1_vecsym(self, p):
        """vecsym(Amul(At,dense,p,1), K)   (wrapPcg.m:65)"""
        h = self._ops()
        if h is not None:
            return vec(h.Amul1_vecsym(p))
        return self.cone.vecsym(self.Amul(p, 1))

    def psdscale(self, d, x, transp=False):
        """psdscale(d, x, K[, transp]); x full length or its PSD part only"""
        cn = self.cone
        h = self._ops(d)
        if h is None or not cn.s.size:
            return cn.psdscale(d, x, transp)
        N = self.A.shape[0]
        xf = vec(x) if np.size(x) == N else np.concatenate((np.zeros(N - np.size(x)), vec(x)))
        return vec(h.psdscale(d, xf, transp))

    def Dx(self, d, x, transp):
        """[sqrt(d.l).*x(1:K.l); asmDxq(d,x,K); psdscale(d,x,K[,transp])]"""
        cn = self.cone
        return np.concatenate((np.sqrt(d["l"]) * x[:cn.l], cn.asmDxq(d, x), self.psdscale(d, x, transp)))

    # ---- sdinit.m:40-78
    def sdinit(self):
        cn, K, pars = self.cone, self.K, self.pars
        b, c = self.b, self.c
        n = cn.l + 2 * cn.nq + int(K["rLen"]) + int(K["hLen"])
        R = {"maxb": np.abs(b).max() if b.size else 0.0, "maxc": np.abs(c).max() if c.size else 0.0}
        y = np.zeros(b.size)
        mu = pars["mu"] * np.sqrt((1 + R["maxb"]) * (1 + R["maxc"]))
        ident = cn.eyeK()
        v = mu * ident
        y0 = n * mu
        R["b0"] = mu
        d0 = np.sqrt((1 + R["maxb"]) / (1 + R["maxc"]))
        x0 = pars["mu"]; z0 = mu ** 2 / x0
        cx = d0 * (c @ v)
        R["sd"] = (z0 + cx) / y0
        d = {"l": d0 ** 2 * np.ones(cn.l)}
        d["l"][0] = x0 / z0
        d["det"] = d0 ** 2 * np.ones(cn.nq)
        d["q1"] = (np.sqrt(2) * d0) * np.ones(cn.nq)
        d["q2"] = np.zeros(cn.i3 - cn.i2)
        d["auxdet"] = np.sqrt(2 * d["det"])
        d["auxtr"] = np.sqrt(2) * (d["q1"] + d["auxdet"])
        d["u"] = np.sqrt(d0) * ident[cn.lq:]
        d["perm"] = np.zeros(0)
        vfrm = {"lab": mu * np.ones(n), "q": d["q2"].copy()}
        vfrm["s"] = vec(self.ref.call("qrK", 1, col(d["u"]), K)) if cn.s.size else np.zeros(0)
        Rb = d0 * self.Amul(v, 0)
        R["b"] = (Rb - x0 * b) / y0
        Rc = cn.vecsym(v / d0 - x0 * c) / y0
        Rc[0] = 0.0
        R["c"] = Rc
        R["maxRb"] = max(1e-6, np.abs(R["b"]).max() if b.size else 0.0)
        R["maxRc"] = max(1e-6, np.abs(R["c"]).max())
        R["norm"] = max(R["maxRb"], R["maxRc"], R["sd"])
        R["w"] = 2 * pars["w"] * np.array([R["maxRb"], R["maxRc"]]) / np.array([1 + R["maxb"], 1 + R["maxc"]])
        return d, v, vfrm, y, y0, R

    # ---- wrapPcg.m:40-130 / loopPcg.m
    def precond(self, L, r):
        p = self.hot.fwdpr1(L, vec(self.hot.fw(L, r)))              # wrapPcg.m:56  p = fwdpr1(Lden, sparfwslv(L, r))
        yv = p / L["d"]
        return p, yv

    def bwsolve(self, L, yv):
        return vec(self.hot.bw(L, self.hot.bwdpr1(L, yv)))         # wrapPcg.m:59  sparbwslv(L, bwdpr1(Lden, y))

    def wrapPcg(self, L, d, DAt, rb, rv, cgpars, y0):
        cn = self.cone
        restol = y0 * cgpars["restol"]
        dx = self.Dx(d, rv, True)
        r = self.Amul(dx)
        if rb is not None:
            r = r + rb
        p, yv = self.precond(L, r)
        ssqrNew = p @ yv
        p = self.bwsolve(L, yv)
        x = self.Amul1_vecsym(p)
        dx = self.Dx(d, x, False)
        ssqrdx = dx @ dx
        if ssqrdx <= 0.0:
            return np.zeros(r.size), rv.copy(), 0, r
        k = 1
        alpha = ssqrNew / ssqrdx
        y = alpha * p
        dx = rv - alpha * dx
        x = self.Dx(d, dx, True)
        r = self.Amul(x)
        if rb is not None:
            r = r + rb
        if np.abs(r).max() < restol:
            return y, dx, k, r
        trial = 0
        pcur = p
        while True:
            dy, dk, xx = self.loopPcg(L, d, DAt, r, pcur, ssqrNew, cgpars, restol)
            if dy is None:
                return y, dx, k, r
            k += dk
            y = y + dy
            dx = dx - xx
            x = self.Dx(d, dx, True)
            r = self.Amul(x)
            if rb is not None:
                r = r + rb
            if np.abs(r).max() < restol or trial >= cgpars["refine"]:
                return y, dx, k, r
            pcur = None
            trial += 1

    def PopK(self, d, x):                                          # PopK.m (lpq = 0, 4 outputs)
        cn = self.cone
        i1, i2 = cn.i1, cn.i2
        y = np.concatenate((d["l"] * x[:i1], -d["det"] * x[i1:i2], cn.qblkmul(d["det"], x)))
        ddotx = d["q1"] * x[i1:i2] + cn.ddot(d["q2"], x)
        Dxp = self.psdscale(d, x)
        y = np.concatenate((y, self.psdscale(d, Dxp, True)))
        xTy = x[:cn.lq] @ y[:cn.lq] + np.sum(ddotx ** 2) + np.sum(Dxp ** 2)
        return y, ddotx, Dxp, xTy

    def loopPcg(self, L, d, DAt, b, p, ssqrNew, cgpars, restol):
        cn = self.cone
        k, STOP = 0, 0
        r = b.copy()
        finew = 0.0
        y = None
        normrmin = np.abs(r).max()
        ymin = None
        alpha = 0.0
        Ap = DApq = DAps = None
        while STOP == 0:
            Lr, tmp = self.precond(L, r)
            if p is None:
                ssqrNew = Lr @ tmp
                p = self.bwsolve(L, tmp)
            else:
                ssqrOld = ssqrNew
                ssqrNew = Lr @ tmp
                p = (ssqrNew / ssqrOld) * p
                p = p + self.bwsolve(L, tmp)
            Ap = self.Amul1_vecsym(p)
            DDAp, DApq, DAps, ssqrDAp = self.PopK(d, Ap)
            if ssqrDAp > 0.0:
                k += 1
                alpha = ssqrNew / ssqrDAp
                if y is not None:
                    hi, lo = self.ref.call("quadadd", 2, col(y[0]), col(y[1]), col(alpha * p))
                    y = (vec(hi), vec(lo))
                else:
                    y = (alpha * p, np.zeros(p.size))              # cg.qprec > 0
                tmpv = self.Amul(DDAp) + (vec(sp.csc_matrix(DAt["q"]).T @ DApq) if cn.nq else 0.0)
                r = r - alpha * tmpv
                fiprev = finew
                finew = (b + r) @ y[0] + (b + r) @ y[1]
                normr = np.abs(r).max()
                if normr < normrmin:
                    ymin = y
                    normrmin = normr
                if normr < restol:
                    STOP = 1
                elif finew - fiprev < cgpars["stagtol"] * fiprev:
                    STOP = 2
                elif k >= cgpars["maxiter"]:
                    STOP = 2
            else:
                STOP = 1
        if STOP == 2:
            y = ymin
        if y is None:
            return None, k, None
        if k == 1:
            DAy = alpha * np.concatenate((np.sqrt(d["l"]) * Ap[:cn.l], cn.asmDxq(d, Ap, DApq), DAps))
        else:
            DAy = 0.0
            for part in y:
                Ap2 = self.Amul1_vecsym(part)
                DAy = DAy + self.Dx(d, Ap2, False)
        return y[0], k, DAy

    # ---- sdfactor.m / sddir.m
    def sdfactor(self, L, d, DAt, v, y, R, y0):
        Lsd = {"DRc": self.Dx(d, R["c"], False)}
        yy, xx, kcg, bb = self.wrapPcg(L, d, DAt, y0 * R["b"], y0 * Lsd["DRc"] - 2 * v, self.pars["cg"], min(1, y0) * R["maxRb"])
        Lsd["y"] = yy - y
        Lsd["x"] = xx + v
        Lsd["kcg"], Lsd["b"] = kcg, bb
        Lsd["denom"] = Lsd["x"] @ Lsd["x"] + Lsd["b"] @ Lsd["y"]
        return Lsd

    def sddir(self, L, Lsd, pv, d, v, vfrm, DAt, R, y, y0, pMode):
        cn = self.cone
        if pMode == 1:
            dy0 = (vfrm["lab"] @ pv) / R["b0"]
            pv = cn.frameit(pv, vfrm["q"], vfrm["s"])
        elif pMode == 2:
            dy0 = -y0
            pv = -v
        else:
            dy0 = (v @ pv) / R["b0"]
        dy, dx, kcg, errb = self.wrapPcg(L, d, DAt, dy0 * R["b"], dy0 * Lsd["DRc"] - pv, self.pars["cg"], min(1, y0) * R["maxRb"])
        rdx0 = (y0 * (Lsd["DRc"] @ dx + R["b"] @ dy) - errb @ y) / Lsd["denom"]
        dy = dy - rdx0 * Lsd["y"]
        dx = rdx0 * Lsd["x"] - dx
        err = {"kcg": kcg, "b": rdx0 * Lsd["b"] - errb}
        err["maxb"] = np.abs(err["b"]).max()
        dx[0] = rdx0 * v[0]
        dz = pv - dx
        return dx, dy, dz, dy0, err

    # ---- maxstep.m
    def maxstep(self, dx, x, auxx):
        cn = self.cone
        mindx = np.min(dx[:cn.l] / x[:cn.l])
        if cn.nq:
            reltr = x[cn.i1:cn.i2] * dx[cn.i1:cn.i2] - cn.ddot(x[cn.i2:cn.i3], dx)
            norm2 = reltr ** 2 - cn.tdet(dx) * auxx["tdet"]
            if np.all(norm2 > 0):
                norm2 = np.sqrt(norm2)
            mindx = min(mindx, np.min((reltr - norm2) / auxx["tdet"]))
        if cn.s.size:
            mindx = min(mindx, cn.minpsdeig(cn.psdinvscale(auxx["u"], dx)))
        return 1.0 / max(-mindx, 1e-16)

    # ---- the neighbourhood test shared by widelen.m / trydif.m
    def wstruct(self, x, z):
        cn = self.cone
        w = {"tdetx": cn.tdet(x), "tdetz": cn.tdet(z)}
        detxz = w["tdetx"] * w["tdetz"] / 4
        if cn.nq == 0:
            lab2q = np.zeros(0)
        else:
            halfxz = (x[cn.i1:cn.i2] * z[cn.i1:cn.i2] + cn.ddot(x[cn.i2:cn.i3], z)) / 2
            tmp = halfxz ** 2 - detxz
            lab2q = halfxz + np.sqrt(tmp) if np.all(tmp > 0) else halfxz
        w["ux"], _ = cn.psdfactor(x)
        w["s"] = cn.psdscale(w["ux"], z)
        w["lab"] = np.concatenate((x[:cn.l] * z[:cn.l], detxz / lab2q if cn.nq else np.zeros(0), lab2q, cn.psdeig(w["s"])))
        return w

    def iswnbr(self, lab, thetaSQR):
        delta, h, alpha = self.ref.call("iswnbr", 3, col(lab), float(thetaSQR))
        return float(np.asarray(delta).ravel()[0]), vec(h), float(np.asarray(alpha).ravel()[0])

    def widelen(self, xc, zc, y0, dx, dz, dy0, d2y0, maxt):
        pars = self.pars
        thetaSQR = pars["theta"] ** 2
        if dy0 < -1e-5 * y0:
            fullt = 2 * y0 / (-dy0 + np.sqrt(dy0 ** 2 - 4 * y0 * d2y0)) if d2y0 < 0 else y0 / (-dy0)
            assert fullt > 0
        else:
            fullt = 2 * maxt
        tR = min(maxt, fullt)
        t, ntry = 0.0, 0
        w, wr = None, {}
        while (t < 0.5 * tR) or ((fullt - tR) + (1e-7 * fullt) < (tR - t)) or ntry == 0:
            ntry = 1
            tM = 0.1 * t + 0.9 * tR if tR == maxt else 0.5 * (t + tR)
            wM = self.wstruct(xc + tM * dx, zc + tM * dz)
            deltaM, hM, alphaM = self.iswnbr(wM["lab"], thetaSQR)
            if (deltaM <= pars["beta"]) or ((tM < fullt / 10) and (deltaM < 1)):
                w, t = wM, tM
                wr = {"h": hM, "alpha": alphaM, "delta": deltaM}
            else:
                tR = tM
        if t == 0:
            w, t = wM, tM
            wr = {"h": hM, "alpha": alphaM, "delta": deltaM}
        wr["desc"] = 1
        return t, wr, w

    def trydif(self, t, wrIN, wIN, x, z):
        w = self.wstruct(x, z)
        delta, h, alpha = self.iswnbr(w["lab"], self.pars["theta"] ** 2)
        wr = {"delta": delta, "h": h, "alpha": alpha, "desc": wrIN["desc"]}
        if delta > self.pars["beta"]:
            return 0.0, wrIN, wIN
        return t, wr, w

    # ---- stepdif.m
    def stepdif(self, d, R, y0, x, y, z, dy0, dx, dy, dz, mint, tpmtd):
        b = self.b
        d0 = np.sqrt(d["l"][0])
        rdx0 = dx[0] / x[0]
        rdy0 = dy0 / y0 - rdx0
        rcdx = (b @ dy - rdx0 * (b @ y)) - (dz[0] - rdx0 * z[0]) / d0
        rcdx = rdy0 * R["sd"] + rcdx / y0
        gap = R["b0"] * y0
        if tpmtd > 0:
            del1 = (z @ dx) / gap
            dRg = rdx0 * R["sd"] + rcdx
        else:
            del1 = (x @ dz) / gap
            dRg = (dy0 / y0) * R["sd"] - rcdx
        usegap = (R["sd"] > 0) or (R["sd"] == 0 and dRg > 0)
        if usegap:
            r0 = R["w"][0] + R["w"][1] + R["sd"]
            beta = (rdy0 * R["w"][0] + rcdx) / r0
        else:
            r0 = R["w"][0] + R["w"][1]
            beta = rdy0 * R["w"][0] / r0
        beta = rdx0 + beta if tpmtd > 0 else (dy0 / y0) - beta
        cc = 2 * np.array([beta, rdx0 * del1]) - (rdx0 + del1) * np.array([1.0, beta])

        if cc[0] <= 0:
            t = abs(tpmtd) if cc[1] >= 0 else min(abs(tpmtd), cc[0] / cc[1])
        else:
            t = mint if cc[1] >= 0 else max(mint, cc[0] / cc[1])
        tg = -R["sd"] / dRg if dRg != 0 else t
        if tg <= 0:
            if t > 0:
                tg = t
        else:
            if t < 0:
                tg = t
        if abs(t) > abs(tg):
            if usegap:
                beta = rdy0 * R["w"][0] / r0
                alpha = 1 - R["sd"] / r0
            else:
                beta = (rdy0 * R["w"][0] + rcdx) / r0
                alpha = 1 + R["sd"] / r0
            beta = rdx0 * alpha + beta if tpmtd > 0 else (dy0 / y0) * alpha - beta
            cc = 2 * np.array([beta, rdx0 * del1]) - (rdx0 + del1) * np.array([1.0, beta])
            if t >= 0:
                if cc @ np.array([1.0, -tg]) <= 0:
                    t = abs(tpmtd) if cc[1] >= 0 else min(abs(tpmtd), cc[0] / cc[1])
                else:
                    t = tg
            else:
                if cc @ np.array([1.0, -tg]) >= 0:
                    t = mint if cc[1] >= 0 else max(mint, cc[0] / cc[1])
                else:
                    t = tg
        if y0 + t * dy0 <= 0:
            t = -y0 / dy0
        return t, y0 * rcdx

    # ---- updtransfo.m
    def updtransfo(self, x, z, w, dIN):
        cn, K, ref = self.cone, self.K, self.ref
        w = dict(w)
        if cn.s.size:
            wlab, q = cn.psdeig(w["s"], want_q=True)
            lab = w["lab"].copy()
            lab[cn.l + 2 * cn.nq:] = wlab
            w["lab"] = lab
        else:
            q = np.zeros(0)
        vfrm = {"lab": np.sqrt(w["lab"])}
        d = {"l": dIN["l"] * (x[:cn.l] / z[:cn.l])}
        if cn.nq == 0:
            for k in ("det", "q1", "q2", "auxdet", "auxtr"):
                d[k] = np.zeros(0)
            vfrm["q"] = np.zeros(0)
        else:
            i1, i2, nq = cn.i1, cn.i2, cn.nq
            j3 = i2 + nq
            s = np.sqrt(w["tdetx"] / w["tdetz"])
            d["det"] = dIN["det"] * s
            psi1 = s * z[i1:i2]; psi2 = cn.qblkmul(s, z)
            tmp = vfrm["lab"][i1:i2] + vfrm["lab"][i2:j3]
            chi1 = (x[i1:i2] + psi1) / tmp
            chi2 = cn.qblkmul(1 / tmp, x[i2:cn.lq] - psi2)
            psi1 = x[i1:i2] - psi1
            psi2 = x[i2:cn.lq] + psi2
            dq = cn.asmDxq(dIN, np.concatenate((chi1, chi2)))
            d["q1"] = dq[:nq]; d["q2"] = dq[nq:]
            d["auxdet"] = np.sqrt(2 * d["det"])
            d["auxtr"] = np.sqrt(2) * (d["q1"] + d["auxdet"])
            alpha = (dIN["q1"] * psi1 + cn.ddot(dIN["q2"], psi2)) / d["auxtr"]
            tmp = 2 * np.sqrt(s)
            psi1 = (psi1 - alpha * chi1) / tmp
            psi2 = psi2 - cn.qblkmul(alpha, chi2)
            psi2 = cn.qblkmul(1 / tmp, psi2)
            gamma = (np.sqrt(2) * psi1 + alpha) / dIN["auxtr"]
            tmp = vfrm["lab"][i2:j3] - vfrm["lab"][i1:i2]
            tmp[tmp == 0] = 1
            psi2 = psi2 + cn.qblkmul(gamma, dIN["q2"])
            vfrm["q"] = cn.qblkmul(1 / tmp, psi2)
        if cn.s.size:
            du = cn.triumtriu(w["ux"], dIN["u"])
            args = (col(du), K, 1.1) + ((col(dIN["perm"]),) if np.size(dIN["perm"]) else ())
            du, perm, gjc, g = ref.call("urotorder", 4, *args)
            q = ref.call("givensrot", 1, gjc, g, col(q), K)
            vinv = ref.call("sqrtinv", 1, q, col(vfrm["lab"]), K)
            frs, r = ref.call("qrK", 2, vinv, K)
            vfrm["s"] = vec(frs)
            d["u"] = cn.triumtriu(vec(r), vec(du))
            d["perm"] = vec(perm)
        else:
            vfrm["s"] = np.zeros(0); d["u"] = np.zeros(0); d["perm"] = np.zeros(0)
        return d, vfrm

    # ---- wregion.m
    def wregion(self, L, Lsd, d, v, vfrm, DAt, R, y, y0, wr):
        cn, pars = self.cone, self.pars
        n = vfrm["lab"].size
        STOP = 0
        err = None
        dxmdz = None
        if wr["delta"] > 0.0:
            vTAR = (1 - wr["alpha"]) * np.maximum(wr["h"], vfrm["lab"])
            pv = 2 * (vTAR - vfrm["lab"])
            dx, dy, dz, dy0, errc = self.sddir(L, Lsd, pv, d, v, vfrm, DAt, R, y, y0, 1)
            xc, zc, yc, y0c = v + dx, v + dz, y + dy, y0 + dy0
            uxc = {"tdet": cn.tdet(xc)}; uzc = {"tdet": cn.tdet(zc)}
            uxc["u"], xispos = cn.psdfactor(xc)
            uzc["u"], zispos = cn.psdfactor(zc)
            critval = max(y0, np.sqrt(min(d["l"][0], 1 / d["l"][0])) * v[0])
            critval = max(1e-3, pars["cg"]["restol"]) * critval * R["maxRb"]
            if (not xispos) or (not zispos) or (errc["maxb"] > critval) or (uxc["tdet"].size and uxc["tdet"].min() <= 0.0) \
                    or (uzc["tdet"].size and uzc["tdet"].min() <= 0.0):
                STOP = -1
                err = errc
            pv = -vTAR
            pMode = 1
        else:
            vTAR = vfrm["lab"]
            xc = v
            uxc = {"tdet": 2 * vfrm["lab"][cn.i1:cn.i2] * vfrm["lab"][cn.i2:cn.i2 + cn.nq]}
            uxc["u"], _ = cn.psdfactor(xc)
            zc, uzc, yc, y0c = v, uxc, y, y0
            errc = {"b": np.zeros(y.size), "maxb": 0.0, "db0": 0.0}
            pv = None
            pMode = 2
        if STOP != -1:
            dx, dy, dz, dy0, err = self.sddir(L, Lsd, pv, d, v, vfrm, DAt, R, y, y0, pMode)
            dxmdz = dx - dz
            if pars["alg"] != 0:
                gd1 = np.concatenate((dxmdz[:cn.l] / vTAR[:cn.l], cn.qinvjmul(vTAR, vfrm["q"], dxmdz), cn.psdinvjmul(vTAR, vfrm["s"], dxmdz)))
                maxt1 = min(self.maxstep(dx, xc, uxc), self.maxstep(dz, zc, uzc))
                jm = np.concatenate((gd1[:cn.l] * dxmdz[:cn.l], cn.qjmul(gd1, dxmdz), cn.psdjmul(gd1, dxmdz)))
                if pars["alg"] == 1:
                    tTAR = 1 - (1 - maxt1)
                    pv = tTAR ** 2 * jm
                    pv2 = 2 * tTAR * (1 - tTAR) * ((vTAR.sum() / n) * np.ones(n) - vTAR) - (2 * tTAR) * vTAR
                else:
                    tTAR = 1 - (1 - maxt1) ** 3
                    pv = (tTAR / 4) * jm
                    pv2 = ((1 - tTAR) * tTAR * R["b0"] * y0 / n) / vTAR - (1 + tTAR / 4) * vTAR
                pv = pv + cn.frameit(pv2, vfrm["q"], vfrm["s"])
                dx, dy, dz, dy0, err = self.sddir(L, Lsd, pv, d, v, vfrm, DAt, R, y, y0, 3)
            PHI = 0.5
            if dy0 < 0 and (PHI * dy0 ** 2 * R["maxRb"]) != 0:
                critval = -(PHI * dy0 * R["maxRb"] + err["maxb"]) * y0c / (PHI * dy0 ** 2 * R["maxRb"])
            else:
                critval = 1
            if critval <= 0:
                STOP = -1
            else:
                tp = self.maxstep(dx, xc, uxc)
                td = self.maxstep(dz, zc, uzc)
                if dy0 < 0:
                    tp = min(tp, critval)
                if xc[0] + td * dx[0] < 0:
                    td = xc[0] / (-dx[0])
                maxt = min(tp, td)
                t, wr, w = self.widelen(xc, zc, y0c, dx, dz, dy0, 0, maxt)
                xscl, ynew, zscl, y0new = xc + t * dx, yc + t * dy, zc + t * dz, y0c + t * dy0
                tdif = 0.0
                if pars["stepdif"] == 1:
                    tdif, rcdx = self.stepdif(d, R, y0new, xscl, ynew, zscl, dy0, dx, dy, dz, -t, tp - td)
                    if tdif != 0:
                        rdx0 = dx[0] / xscl[0]
                        mu = 1 + tdif * rdx0
                        if tp > td:
                            newx, newz = xscl + tdif * dx, mu * zscl
                        else:
                            newx, newz = mu * xscl, zscl + tdif * dz
                        tdif, wr, w = self.trydif(tdif, wr, w, newx, newz)
                relt = {}
                if tdif != 0:
                    rdy0 = dy0 - rdx0 * y0new
                    zscl, xscl = newz, newx
                    if tp > td:
                        ynew = mu * ynew; y0new = mu * y0new
                        err["b"] = (tdif * rdy0) * R["b"] + errc["b"] + (t + tdif) * err["b"]
                        err["g"] = tdif * rcdx
                        relt = {"p": (t + tdif) / tp, "d": t / td}
                    else:
                        ynew = ynew + tdif * dy; y0new = y0new + tdif * dy0
                        err["b"] = -(tdif * rdy0) * R["b"] + mu * (errc["b"] + t * err["b"])
                        err["g"] = -tdif * rcdx
                        relt = {"p": t / tp, "d": (t + tdif) / td}
                else:
                    err["b"] = errc["b"] + t * err["b"]
                    err["g"] = 0.0
                    relt = {"p": t / maxt, "d": t / maxt}
                wr["tpmtd"] = tp - td
                err["maxb"] = errc["maxb"] + t * err["maxb"]
                err["db0"] = xscl @ zscl - y0new * R["b0"]
                return xscl, ynew, zscl, y0new, w, relt, dxmdz, err, wr
        relt = {"p": 0.0, "d": 0.0}
        err = err or {}
        err.update({"b": np.zeros(self.b.size), "db0": 0.0, "g": 0.0, "kcg": err.get("kcg", 0)})
        return None, y, None, y0, None, relt, dxmdz, err, wr

    # ---- sedumi.m:396-571 (main loop) and :590-612 (the solution)
    def solve(self, verbose=False):
        """The interior-point loop (sedumi.m:300-640).  Runs under `blas_threads`: the cone algebra is many small LAPACK calls per iteration."""
        with blas_threads(self.K):
            return self._solve(verbose)

    def _solve(self, verbose=False):
        cn, pars, hot, S, K = self.cone, dict(self.pars), self.hot, self.S, self.K
        b = self.b
        d, v, vfrm, y, y0, R = self.sdinit()
        n = vfrm["lab"].size
        merit = (R["w"].sum() + max(R["sd"], 0)) ** 2 * y0 / R["b0"]
        L = dict(S["L"])
        STOP, it = 0, 0
        wr = {"delta": 0.0, "desc": 1}
        feasratio = 0.0
        err = {"kcg": 0}
        Lsd = {"kcg": 0}
        stepdif = pars["stepdif"]
        rows = []
        by = 0.0
        x0 = 1.0
        while STOP == 0:
            it += 1
            if stepdif == 2 and (it > 20 or (it > 1 and (err["kcg"] + Lsd["kcg"] > 3)) or (it > 5 and abs(1 - feasratio) < 0.05)):
                stepdif = 1
            self.pars["stepdif"] = stepdif
            DAt = self.G.getDAtm(S, d)                              # sedumi.m:442
            L = hot.factor(S, d, DAt, L, pars["chol"]) if self.den is None else hot.factor(S, d, DAt, L, pars["chol"], self.den)   # sedumi.m:446-463
            Lsd = self.sdfactor(L, d, DAt, v, y, R, y0)              # sedumi.m:466
            y0Old = y0
            xscl, yNxt, zscl, y0Nxt, w, relt, dxmdz, err, wr = self.wregion(L, Lsd, d, v, vfrm, DAt, R, y, y0, wr)
            if xscl is None:                                        # wregion rejected the step (STOP = -1 inside)
                STOP = -1
                it -= 1
                break
            if y0Nxt > 0:
                R["b"] = R["b"] + err["b"] / y0Nxt
                R["sd"] = R["sd"] + err["g"] / y0Nxt
                R["b0"] = R["b0"] + err["db0"] / y0Nxt
                y0 = y0Nxt
            else:
                R["b"] = (y0Nxt * R["b"] + err["b"]) / y0Old
                R["sd"] = (y0Nxt * R["sd"] + err["g"]) / y0Old
                R["b0"] = (y0Nxt * R["b0"] + err["db0"]) / y0Old
                R["w"][1] = abs(y0Nxt / y0Old) * R["w"][1]
                R["c"] = (y0Nxt / y0Old) * R["c"]
                R["maxRc"] = np.abs(R["c"]).max()
                y0 = y0Old
            R["maxRb"] = np.abs(R["b"]).max()
            R["w"][0] = 2 * pars["w"][0] * R["maxRb"] / (1 + R["maxb"])
            meritOld = merit
            merit = (R["w"].sum() + max(R["sd"], 0)) ** 2 * y0 / R["b0"]
            rate = merit / meritOld
            if rate >= 0.9999 and wr["desc"] == 1:
                STOP = -1
                it -= 1
                break
            feasratio = dxmdz[0] / v[0]
            y = yNxt
            by = float(b @ y)
            d, vfrm = self.updtransfo(xscl, zscl, w, d)
            v = cn.frameit(vfrm["lab"], vfrm["q"], vfrm["s"])
            x0 = np.sqrt(d["l"][0]) * v[0]
            r0 = R["w"].sum()
            cx = by + y0 * R["sd"] - x0 / d["l"][0]
            rgap = max(cx - by, 0) / max(abs(cx), abs(by), 1e-3 * x0)
            precision1 = y0 * r0 / (1 + x0)
            precision2 = (y0 * r0 + rgap) / x0
            row = {"iter": it, "by_x0": by / x0, "gap": merit, "delta": wr["delta"], "rate": rate, "tP": relt["p"], "tD": relt["d"],
                   "feas": feasratio, "kcg1": err["kcg"], "kcg2": Lsd["kcg"], "prec": max(precision1, precision2),
                   "nskip": L["nskip"], "nadd": L["nadd"]}
            rows.append(row)
            if verbose:
                print(" %2d : %10.2E %8.2E %5.3f %6.4f %6.4f %6.4f %6.2f %2d %2d  %1.1E" % (
                    it, row["by_x0"], merit, wr["delta"], rate, relt["p"], relt["d"], feasratio, err["kcg"], Lsd["kcg"], row["prec"]), flush=True)
            if by > 0 and abs(1 + feasratio) < 0.05 and R["b0"] * y0 < 0.5:
                if cn.maxeigK(self.Amul(y, 1)) <= pars["eps"] * by:
                    STOP = 3
                    break
            if precision1 < pars["eps"]:
                if precision2 < pars["eps"]:
                    STOP = 1
                    break
                elif y0 * R["maxRb"] + x0 * R["maxb"] < -pars["eps"] * cx:
                    STOP = 1
                    break
                elif y0 * R["maxRc"] + x0 * R["maxc"] < pars["eps"] * by:
                    STOP = 1
                    break
            if it >= pars["maxiter"]:
                STOP = -1
        self.pars["stepdif"] = pars["stepdif"]
        # ---- the solution in the scaled-back variables (sedumi.m:590-612): x = D(d) v, objective values c'x/x0, b'y/x0
        x = self.Dx(d, v, True)
        x0 = x[0]
        cx = float(self.c @ x)
        by = float(b @ y)
        return {"iter": it, "STOP": STOP, "cx": cx / x0 if x0 > 0 else cx, "by": by / x0 if x0 > 0 else by, "x0": x0, "rows": rows,
                "feasratio": feasratio, "hot": hot.name}


def blas_threads(K):
    """Context manager: the BLAS / LAPACK thread pools numpy and scipy use, sized for the cone's blocks.  With a pool as wide as the host (256
    threads on the MI355X box) every small call -- a 70 x 70 triangular solve, eigh, qr, dozens per iteration -- pays the pool's wake-up:
    control07 took 1.76 s in the default pool and 0.58 s with one thread (profiles/r08w_driver_profile_control07.txt).  One thread while no PSD /
    Lorentz block exceeds order 256, up to 16 beyond (dense eigh / chol of a MAXCUT block).  Without threadpoolctl: nothing is changed."""
    import contextlib
    try:
        from threadpoolctl import threadpool_limits
    except ImportError:                                                 # pragma: no cover
        return contextlib.nullcontext()
    nmax = max([0] + [int(n) for key in ("s", "q") for n in np.atleast_1d(K.get(key, [])).ravel()])
    ncpu = os.cpu_count() or 1
    env = os.environ.get("SEDUMI_DRIVER_BLAS_THREADS")
    return threadpool_limits(limits=int(env) if env else (1 if nmax <= 256 else min(16, ncpu)))


def load_mat(path):
    """(At, b, c, K) of a SeDuMi problem file (.mat with At | A, b, c, K -- e.g. the reference's examples/*.mat)."""
    import scipy.io as sio
    d = sio.loadmat(path)
    K = {k: d["K"][k][0, 0].astype(float).ravel() for k in d["K"].dtype.names}
    At = d["At"] if "At" in d else d["A"].T
    return At.astype(np.complex128) if np.iscomplexobj(d["c"]) else At, d["b"], d["c"], K


def solve(At, b, c, K, pars=None, hot=None, verbose=False):
    """sedumi(At, b, c, K) without MATLAB: the loop above on this package's library (resident plan) and its own cone algebra.
    Returns the dictionary of Sedumi.solve (x, y, cx, by, iter, feasratio, rows = the iteration log, ...)."""
    return Sedumi(At, b, c, K, hot=hot, pars=pars).solve(verbose=verbose)

"""sedumi_amd/driver/glue.py -- the MATLAB glue between the user-level problem (At, b, c, K) and the MEX calls of SeDuMi's loop, in Python.

Restated `.m` lines (file:line into /root/reference), every MEX call going through `self.ref.call(name, nlhs, *args)`:

  pretransfo.m:64-542   (real K.l / K.q / K.s case only)   -> pretransfo_real
  sedumi.m:356-392      (dense split, Aord, ADA pattern)    -> setup
  getdense.m:38-75                                           -> getdense
  getsymbada.m:41-60                                         -> getsymbada
  symbchol.m:62-83                                           -> symbchol
  getDAtm.m:39-47                                            -> getDAtm
  sdinit.m:63-78        (iteration-1 scaling)                -> sdinit_scaling
  sedumi.m:450-458      (getada1/2/3 + blkchol)              -> iteration_ref
  deninfac.m:58-94, wrapPcg.m:56-59                          -> solve_ref

Part of the MATLAB-free driver (sedumi_amd/driver/); the test oracle drives the compiled reference through the
same class (oracle/glue.py passes oracle.refmex.RefMex as the host).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp



def _col(x):
    return np.asarray(x, dtype=np.float64).reshape(-1, 1)


def spars(X):
    m, n = X.shape
    return X.nnz / float(m * n) if m * n else 0.0


# --------------------------------------------------------------------------
def pretransfo_real(At, b, c, K):
    """pretransfo.m restricted to real LP / Lorentz / PSD cones without free
    variables, rotated cones or complex data (pretransfo.m:64-542).
    ``K`` is a dict with optional keys l, q, s.  Returns (At, b, c, Kint)."""
    At = sp.csc_matrix(At, dtype=np.float64)
    Kl = int(np.asarray(K.get("l", 0)).ravel()[0]) if np.size(K.get("l", 0)) else 0
    Kq = np.asarray(K.get("q", []), dtype=np.int64).ravel()
    Kq = Kq[Kq > 0]
    Ks = np.asarray(K.get("s", []), dtype=np.int64).ravel()
    Ks = Ks[Ks > 0]
    N_flqr = Kl + int(Kq.sum())
    N = N_flqr + int((Ks ** 2).sum())
    if At.shape[0] != N and At.shape[1] == N:
        At = sp.csc_matrix(At.T)
    assert At.shape[0] == N, "(At,K) size mismatch"
    b = np.asarray(b.todense() if sp.issparse(b) else b, dtype=np.float64).ravel()
    c = np.asarray(c.todense() if sp.issparse(c) else c, dtype=np.float64).ravel()
    L_s = len(Ks)
    # --- diagonal PSD blocks -> LP (pretransfo.m:241-256)
    sdiag = np.ones(L_s, dtype=bool)
    if L_s:
        strt = np.concatenate(([0], np.cumsum(Ks[:-1] ** 2)))  # 0-based starts rel. to N_flqr
        rows_any = np.zeros(N - N_flqr, dtype=bool)
        Acsr = sp.csr_matrix(At[N_flqr:, :])
        rows_any[np.diff(Acsr.indptr) > 0] = True
        rows_any |= c[N_flqr:] != 0
        spattern = np.nonzero(rows_any)[0]
        blk = np.searchsorted(strt, spattern, side="right") - 1
        offd = (spattern - strt[blk]) % (Ks[blk] + 1) != 0
        sdiag[np.unique(blk[offd])] = False
    sreal = ~sdiag
    ii, jj = [], []
    newL = 0
    if Kl:
        ii.append(np.arange(newL, newL + Kl)); jj.append(np.arange(0, Kl)); newL += Kl
    if sdiag.any():
        dsize = Ks[sdiag]
        jstrt = N_flqr + np.concatenate(([0], np.cumsum(Ks[:-1] ** 2)))
        jstrt = jstrt[sdiag]
        for n_k, j0 in zip(dsize, jstrt):
            ii.append(np.arange(newL, newL + n_k))
            jj.append(j0 + (n_k + 1) * np.arange(n_k))
            newL += n_k
    tr_off = newL
    nb_off = newL + len(Kq)
    if len(Kq):
        ndxs = np.concatenate(([0], np.cumsum(Kq[:-1])))
        N_q = int(Kq.sum())
        it = np.full(N_q, -1, dtype=np.int64)
        it[ndxs] = tr_off + np.arange(len(Kq))
        it[it < 0] = nb_off + np.arange(N_q - len(Kq))
        ii.append(it); jj.append(Kl + np.arange(N_q))
        nb_off += N_q - len(Kq)
    if sreal.any():
        jstrt_all = N_flqr + np.concatenate(([0], np.cumsum(Ks[:-1] ** 2)))
        for n_k, j0 in zip(Ks[sreal], jstrt_all[sreal]):
            idx = np.arange(n_k * n_k)
            cols = idx // n_k
            rows = idx - n_k * cols
            ii.append(nb_off + np.maximum(rows, cols) + np.minimum(rows, cols) * n_k)
            jj.append(j0 + idx)
            nb_off += n_k * n_k
    Kint_l = newL + 1
    Ks_new = Ks[sreal]
    KN = Kint_l + int(Kq.sum()) + int((Ks_new ** 2).sum())
    ii = np.concatenate(ii) + 1 if ii else np.zeros(0, dtype=np.int64)
    jj = np.concatenate(jj) if jj else np.zeros(0, dtype=np.int64)
    QR = sp.csc_matrix((np.ones(len(ii)), (ii, jj)), shape=(KN, N))
    At2 = sp.csc_matrix(QR @ At)
    At2.sum_duplicates(); At2.sort_indices()
    c2 = np.asarray(QR @ c).ravel()
    Kint = {
        "f": 0.0, "l": float(Kint_l), "q": Kq.astype(np.float64).reshape(1, -1),
        "r": np.zeros((0, 1)), "s": Ks_new.astype(np.float64).reshape(1, -1),
        "rsdpN": float(len(Ks_new)), "N": float(KN), "m": float(len(b)),
    }
    blkstart = np.cumsum(np.concatenate(([Kint_l + 1, len(Kq)], Kq - 1, Ks_new ** 2))).astype(np.float64)
    Kint["blkstart"] = blkstart.reshape(1, -1)
    Kint["rLen"] = float(Ks_new.sum()); Kint["hLen"] = 0.0
    Kint["qMaxn"] = float(max([0] + list(Kq))); Kint["rMaxn"] = float(max([0] + list(Ks_new)))
    Kint["hMaxn"] = 0.0
    mb = blkstart[np.cumsum([0, 1, len(Kq)])]
    Kint["mainblks"] = mb.reshape(1, -1)
    Kint["qblkstart"] = blkstart[1:2 + len(Kq)].reshape(1, -1)
    Kint["sblkstart"] = blkstart[1 + len(Kq):].reshape(1, -1)
    Kint["lq"] = float(mb[-1] - 1)
    return At2, b, c2, Kint


# --------------------------------------------------------------------------
class Glue:
    def __init__(self, ref=None):
        """ref: the MEX host whose `.call(name, nlhs, *args)` runs the MEX functions -- by default this package's own
        (sedumi_amd.driver.conemex.NativeMex: numpy / LAPACK for the cone algebra, the library for the hot path); the tests pass
        the compiled reference (oracle.refmex.RefMex) through the subclass in oracle/glue.py."""
        if ref is None:
            from .conemex import NativeMex
            ref = NativeMex()
        self.ref = ref

    # getdense.m:38-75
    def getdense(self, A, Ablkjc, K, denq=0.75, denf=10.0):
        ref = self.ref
        N, m = A.shape
        NORMDEN = 5
        E = ref.call("extractA", 1, A, Ablkjc, 0.0, 3.0, 1.0, K["lq"] + 1)
        colnz = np.asarray((E != 0).sum(axis=1)).ravel().astype(np.float64)
        nblk_s = K["sblkstart"].size - 1
        if nblk_s > 0:
            F = ref.call("findblks", 1, A, Ablkjc, 3.0, np.zeros((0, 0)), K["sblkstart"])
            h = max(NORMDEN, float(np.asarray(F.sum(axis=1)).max()) if F.shape[0] else 0)
        else:
            h = NORMDEN
        i1 = int(K["mainblks"].ravel()[0]); i2 = int(K["mainblks"].ravel()[1])
        Ablkq = None
        if i1 < i2:
            Ablkq = ref.call("extractA", 1, A, Ablkjc, 1.0, 2.0, float(i1), float(i2))
            Ablkq2 = ref.call("findblks", 1, A, Ablkjc, 2.0, 3.0, K["qblkstart"])
            Ablkq = sp.csc_matrix(((Ablkq != 0).astype(float) + Ablkq2) != 0).astype(float)
            colnz[i1 - 1:i2 - 1] = np.asarray(Ablkq.sum(axis=1)).ravel()
        big = colnz[colnz > h]
        denqN = int(np.ceil(denq * len(colnz))) - (len(colnz) - len(big))
        if denqN < 1:
            spquant = h
        else:
            spquant = np.sort(big)[denqN - 1]
        cols = np.nonzero(colnz > denf * spquant)[0] + 1          # 1-based
        dq = np.nonzero(colnz[i1 - 1:i2 - 1] > denf * spquant)[0] + 1
        dl = int((cols < i1).sum())
        if len(cols) > m / 2:
            dl = 0; cols = np.zeros(0, dtype=np.int64); dq = np.zeros(0, dtype=np.int64)
        if len(dq) == 0:
            Adotdden = sp.csc_matrix((m, 0))
        else:
            Adotdden = sp.csc_matrix(Ablkq[dq - 1, :].T)
        return {"cols": cols, "q": dq, "l": dl}, Adotdden

    # getsymbada.m:41-60
    def getsymbada(self, A, Ablkjc, DAtq, psdblkstart):
        ref = self.ref
        m = A.shape[1]
        Alpq = ref.call("extractA", 1, A, Ablkjc, 0.0, 3.0, 1.0, float(psdblkstart.ravel()[0]))
        Alpq = sp.csc_matrix(Alpq != 0).astype(float)
        Ablks = ref.call("findblks", 1, A, Ablkjc, 3.0, np.zeros((0, 0)), psdblkstart)
        full = lambda: sp.csc_matrix(np.ones((m, m)))
        hasq = DAtq is not None and DAtq.shape[0] > 0
        if spars(Ablks) == 1 or spars(Alpq) == 1 or (hasq and spars(DAtq) == 1):
            return full()
        S = sp.csc_matrix((DAtq.T @ DAtq)) if hasq else sp.csc_matrix((m, m))
        if spars(S) > 0.9:
            return full()
        S = sp.csc_matrix(S + Alpq.T @ Alpq)
        if spars(S) > 0.9:
            return full()
        S = sp.csc_matrix(S + Ablks.T @ Ablks)
        S.sort_indices()
        return S

    # symbchol.m:62-83
    def symbchol(self, ADA, cachsz=512.0):
        ref = self.ref
        m = ADA.shape[0]
        if spars(ADA) < 1:
            perm = ref.call("ordmmdmex", 1, ADA)
            L = ref.call("symfctmex", 1, ADA, perm)
        else:
            L = {"perm": np.arange(1, m + 1, dtype=np.float64).reshape(-1, 1),
                 "L": sp.csc_matrix(np.tril(np.ones((m, m)))),
                 "xsuper": np.array([[1.0], [m + 1.0]])}
        L["tmpsiz"] = ref.call("choltmpsiz", 1, L)
        L["split"] = ref.call("cholsplit", 1, L, cachsz)
        return L

    @staticmethod
    def raw(X):
        """a sparse matrix to be handed to a MEX exactly as stored (incorder's dz: the row order inside its columns is data)"""
        from sedumi_amd.mexhost import RawSparse
        return RawSparse(X)

    # symbcholden.m:43-55, dense LP columns only (LAD = [symbfwblk(L, dense.A(:, 1:dense.l))]; no Lorentz blocks / trace columns)
    def symbcholden(self, L, dense):
        ref = self.ref
        LAD = ref.call("symbfwblk", 1, L, sp.csc_matrix(dense["A"]))
        perm, dz = ref.call("incorder", 2, LAD)
        return ref.call("finsymbden", 1, LAD, perm, self.raw(dz), float(int(dense["l"]) + 1))

    # sedumi.m:356-392
    def setup(self, A, K, denq=0.75, denf=10.0):
        ref = self.ref
        m = A.shape[1]
        A = sp.csc_matrix(A)
        Ablkjc = ref.call("partitA", 1, A, K["mainblks"])
        dense, denq_pat = self.getdense(A, Ablkjc, K, denq, denf)
        if len(dense["cols"]):
            Acsr = sp.csr_matrix(A)
            dense["A"] = sp.csc_matrix(Acsr[dense["cols"] - 1, :].T)
            keep = np.ones(A.shape[0]); keep[dense["cols"] - 1] = 0.0
            A = sp.csc_matrix(sp.diags(keep) @ A)
            A.eliminate_zeros(); A.sort_indices()
            Ablkjc = ref.call("partitA", 1, A, K["mainblks"])
        else:
            dense["A"] = sp.csc_matrix((m, 0))
        Aord = {}
        Aord["lqperm"] = ref.call("sortnnz", 1, A, np.zeros((0, 0)), Ablkjc[:, 2])
        DAt = {"denq": denq_pat}
        nq = K["q"].size
        if nq > 0:
            q = ref.call("findblks", 1, A, Ablkjc, 2.0, 3.0, K["qblkstart"])
            q = sp.lil_matrix(q)
            if len(dense["q"]):
                q[dense["q"] - 1, :] = 0.0
            q = sp.csc_matrix(q); q.eliminate_zeros()
            E = ref.call("extractA", 1, A, Ablkjc, 1.0, 2.0,
                         float(K["mainblks"].ravel()[0]), float(K["mainblks"].ravel()[1]))
            q = sp.csc_matrix(q + (E != 0).astype(float))
            q.sort_indices()
            DAt["q"] = q
            Aord["qperm"] = ref.call("sortnnz", 1, q, np.zeros((0, 0)), np.zeros((0, 0)))
        else:
            DAt["q"] = sp.csc_matrix((0, m))
            Aord["qperm"] = np.arange(1, m + 1, dtype=np.float64).reshape(-1, 1)
        sperm, dz = ref.call("incorder", 2, A, Ablkjc[:, 2], float(K["mainblks"].ravel()[2]))
        Aord["sperm"] = sperm; Aord["dz"] = dz
        ADA = self.getsymbada(A, Ablkjc, DAt["q"], K["sblkstart"])
        L = self.symbchol(ADA)
        return {"A": A, "Ablkjc": Ablkjc, "dense": dense, "Aord": Aord, "DAt": DAt,
                "ADA": ADA, "L": L, "K": K}

    # getDAtm.m:39-47 (no dense Lorentz blocks handled: denq empty)
    def getDAtm(self, S, d):
        ref = self.ref
        A, K, Ablkjc = S["A"], S["K"], S["Ablkjc"]
        nq = K["q"].size
        m = A.shape[1]
        if nq == 0:
            return {"q": sp.csc_matrix((0, m)), "denq": sp.csc_matrix((m, 0))}
        q = ref.call("extractA", 1, A, Ablkjc, 1.0, 2.0,
                     float(K["mainblks"].ravel()[0]), float(K["mainblks"].ravel()[1]))
        q = sp.csc_matrix(sp.diags(np.asarray(d["q1"]).ravel()) @ q)
        q = sp.csc_matrix(q + ref.call("ddot", 1, _col(d["q2"]), A, K["qblkstart"], Ablkjc))
        q.sort_indices()
        return {"q": q, "denq": sp.csc_matrix((m, 0))}

    # sedumi.m:450-458
    def iteration_ref(self, S, d, udsqr, pars_chol=None):
        """Run getada1 -> getada2 -> getada3 -> blkchol with the reference MEX."""
        ref = self.ref
        K = S["K"]
        pars_chol = pars_chol or default_pars_chol()
        DAt = self.getDAtm(S, d)
        dstruct = {"l": _col(d["l"]), "det": _col(d["det"])}
        ADA1 = ref.call("getada1", 1, S["ADA"], S["A"], S["Ablkjc"][:, 2], S["Aord"]["lqperm"],
                        dstruct, K["qblkstart"])
        ADA2 = ref.call("getada2", 1, ADA1, DAt, S["Aord"], K)
        ADA3, absd = ref.call("getada3", 2, ADA2, S["A"], S["Ablkjc"][:, 2], S["Aord"], _col(udsqr), K)
        LL, Ld, Lskip, Ladd = ref.call("blkchol", 4, S["L"], ADA3, pars_chol, absd)
        return {"DAt": DAt, "ADA1": ADA1, "ADA2": ADA2, "ADA": ADA3, "absd": absd,
                "LL": LL, "Ld": Ld, "Lskip": Lskip, "Ladd": Ladd}

    # wrapPcg.m:56-59 without dense columns
    def solve_ref(self, S, fac, r):
        ref = self.ref
        L = dict(S["L"]); L["L"] = fac["LL"]
        p = ref.call("fwblkslv", 1, L, _col(r))
        y = p / fac["Ld"]
        return ref.call("bwblkslv", 1, L, y)


def default_pars_chol():
    """checkpars.m:144-168"""
    return {"skip": 1.0, "canceltol": 1e-12, "maxu": 5e5, "abstol": 1e-20, "maxuden": 5e2}


def sdinit_scaling(K, b, c, mu_par=1.0):
    """sdinit.m:63-78: iteration-1 (identity) scaling. Returns (d, udsqr)."""
    maxb = np.abs(b).max() if len(b) else 0.0
    maxc = np.abs(c).max() if len(c) else 0.0
    mu = mu_par * np.sqrt((1 + maxb) * (1 + maxc))
    d0 = np.sqrt((1 + maxb) / (1 + maxc))
    x0 = mu_par; z0 = mu ** 2 / x0
    Kl = int(K["l"]); nq = K["q"].size
    d = {"l": d0 ** 2 * np.ones(Kl), "det": d0 ** 2 * np.ones(nq),
         "q1": np.sqrt(2) * d0 * np.ones(nq),
         "q2": np.zeros(int(K["mainblks"].ravel()[2] - K["mainblks"].ravel()[1]))}
    d["l"][0] = x0 / z0
    ud = []
    for n in K["s"].ravel().astype(int):
        ud.append((d0 * np.eye(n)).ravel(order="F"))   # U = sqrt(d0) I  =>  U'U = d0 I
    udsqr = np.concatenate(ud) if ud else np.zeros(0)
    return d, udsqr


def random_scaling(K, seed=0, cond=1e4):
    """Synthetic 'late iteration' scaling: SPD D_k = Q diag(logspace) Q' per PSD
    block, positive d.l / d.det, random Lorentz q1/q2."""
    rng = np.random.default_rng(seed)
    Kl = int(K["l"]); nq = K["q"].size
    e = np.log10(cond) / 2
    d = {"l": 10.0 ** rng.uniform(-e, e, Kl), "det": 10.0 ** rng.uniform(-e / 2, e / 2, nq),
         "q1": 1.0 + rng.random(nq),
         "q2": 0.3 * rng.standard_normal(int(K["mainblks"].ravel()[2] - K["mainblks"].ravel()[1]))}
    ud = []
    for n in K["s"].ravel().astype(int):
        Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
        w = np.logspace(-e, e, n)
        D = (Q * w) @ Q.T
        D = (D + D.T) / 2
        ud.append(D.ravel(order="F"))
    udsqr = np.concatenate(ud) if ud else np.zeros(0)
    return d, udsqr

"""tests/driver/sedumi_loop.py -- TEST INFRASTRUCTURE ONLY (it drives the oracle).

The interior-point loop of the product (sedumi_amd/driver/loop.py: sedumi.m:428-571 restated) with the parts the tests switch:

    hot = RefHot(glue)   the reference MEX (oracle/_ref):   getada1/2/3 | getada.m, blkchol, fwblkslv, bwblkslv, invcholfac
    hot = HipHot()       this repository's library through sedumi_amd.mex (HIP on a GPU box, the fiber emulator on CPU)
    hot = PlanHot()      the same library through its resident plan: problem, scaling, ADA', factor and solves stay in HBM
    hot = ShimHot(host)  the same library through its built mexFunction shims on a MEX host (what an unmodified sedumi.m calls)

and, HERE, the compiled reference as the MEX host of everything outside the hot path (the cone algebra qrK, psdframeit, psdinvjmul,
urotorder, givensrot, sqrtinv, iswnbr, vecsym, ddot, qblkmul, quadadd and the set-up MEX): `Sedumi` below defaults to oracle.glue.Glue and
RefHot, so that everything but the hot path is IDENTICAL in the runs being compared and the two iteration logs answer the north_star's
acceptance question: same iteration count, same residual columns, same objective values (examples/test_sedumi.m:22-28).  The product's own
defaults (native cone algebra, resident hot path) are tested in tests/test_native_driver.py.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from sedumi_amd.driver import loop as _loop  # noqa: E402
from sedumi_amd.driver.loop import *  # noqa: E402,F401,F403
from sedumi_amd.driver.loop import MexShapedHot, col, vec  # noqa: E402
from oracle import glue as gl  # noqa: E402   (AFTER the star import: the product's loop has a `gl` of its own -- its glue with the native MEX host)


class RefHot(MexShapedHot):
    """The reference's own MEX for the normal-equations path (sedumi.m:446-458, wrapPcg.m:56-59)."""
    name = "reference"

    def __init__(self, G):
        self.G, self.ref = G, G.ref

    def mexcall(self, name, nlhs, *args):
        return self.ref.call(name, nlhs, *args)

    def form(self, S, d, DAt):
        K, ref = S["K"], self.ref
        if np.sum(K["s"]) == 0:                                    # getada.m:13-40
            A = sp.csc_matrix(S["A"])
            nlq = int(K["mainblks"].ravel()[2]) - 1
            sv = np.concatenate((vec(d["l"]), -vec(d["det"]), np.zeros(nlq - int(K["l"]) - K["q"].size)))
            qb = K["qblkstart"].ravel().astype(int) - 1
            for i in range(K["q"].size):
                sv[qb[i]:qb[i + 1]] = vec(d["det"])[i]
            Alq = A[:nlq, :]
            Q = sp.csc_matrix(DAt["q"])
            ADA = sp.csc_matrix(Q.T @ Q + Alq.T @ sp.diags(sv) @ Alq)
            ADA.sort_indices()
            return ADA, col(ADA.diagonal())
        dstruct = {"l": col(d["l"]), "det": col(d["det"])}
        ADA = ref.call("getada1", 1, S["ADA"], S["A"], S["Ablkjc"][:, 2], S["Aord"]["lqperm"], dstruct, K["qblkstart"])
        ADA = ref.call("getada2", 1, ADA, DAt, S["Aord"], K)
        ud = ref.call("invcholfac", 1, col(d["u"]), K, col(d["perm"])) if np.size(d["perm"]) else ref.call("invcholfac", 1, col(d["u"]), K)
        return tuple(ref.call("getada3", 2, ADA, S["A"], S["Ablkjc"][:, 2], S["Aord"], ud, K))

    def blkchol(self, L, ADA, pars, absd):
        return self.ref.call("blkchol", 4, L, ADA, pars, absd)

    def fw(self, L, r):
        return self.ref.call("fwblkslv", 1, L, col(r))

    def bw(self, L, r):
        return self.ref.call("bwblkslv", 1, L, col(r))


class ShimHot(MexShapedHot):
    """The drop-in tier as an unmodified sedumi.m drives it: the built mexFunction shims (sedumi_amd/mexshims) on a MEX host, the
    global ADA_sedumi_ handed from iteration to iteration as sedumi.m:450-452 does (`ADA_sedumi_ = getada1(ADA_sedumi_, ...)`: with
    lazy intermediates the array that comes back is a token, and it goes in again next time), L.L / L.d through blkchol.mex."""
    name = "sedumi_amd.mexshims"

    def mexcall(self, name, nlhs, *args):
        return self.host.call(name, nlhs, *args)

    def __init__(self, host):
        self.host, self.ADA = host, None                          # host.call(name, nlhs, *args): oracle.refmex.RefMex(mex_dir=<shims>) or sedumi_amd.mexhost.MexHost

    def form(self, S, d, DAt):
        K, h = S["K"], self.host
        if self.ADA is None:
            self.ADA = S["ADA"]                                    # sedumi.m:382 getsymbada
        if np.sum(K["s"]) == 0:                                    # sedumi.m:446-448: getada.m's route, the global updated by the gateway
            h.set_global("ADA_sedumi_", self.ADA)
            absd = h.call("getada", 1, S["A"], K, {"l": col(d["l"]), "det": col(d["det"])}, DAt)
            self.ADA = h.get_global("ADA_sedumi_")
            return self.ADA, absd
        dstruct = {"l": col(d["l"]), "det": col(d["det"])}
        ADA = h.call("getada1", 1, self.ADA, S["A"], S["Ablkjc"][:, 2], S["Aord"]["lqperm"], dstruct, K["qblkstart"])
        ADA = h.call("getada2", 1, ADA, DAt, S["Aord"], K)
        ud = h.call("invcholfac", 1, col(d["u"]), K, col(d["perm"])) if np.size(d["perm"]) else h.call("invcholfac", 1, col(d["u"]), K)
        self.ADA, absd = h.call("getada3", 2, ADA, S["A"], S["Ablkjc"][:, 2], S["Aord"], ud, K)
        return self.ADA, absd

    def blkchol(self, L, ADA, pars, absd):
        return self.host.call("blkchol", 4, L, ADA, pars, absd)

    def fw(self, L, r):
        return self.host.call("fwblkslv", 1, L, col(r))

    def bw(self, L, r):
        return self.host.call("bwblkslv", 1, L, col(r))


class ShadowHot:
    """Runs `shadow` next to `primary` on identical inputs -- every factorisation and every solve of a whole run, i.e. the
    scalings a real solve produces, ill-conditioned tail included -- records how far the two are apart, and continues
    with the primary's results."""

    def __init__(self, primary, shadow):
        self.primary, self.shadow, self.name = primary, shadow, primary.name + "+shadow"
        self.records, self.it = [], 0

    @staticmethod
    def _rel(a, b):
        return float(np.linalg.norm(vec(a) - vec(b)) / max(np.linalg.norm(vec(b)), 1e-300))

    def factor(self, S, d, DAt, L, pars):
        Lp = self.primary.factor(S, d, DAt, L, pars)
        self.Ls = self.shadow.factor(S, d, DAt, L, pars)
        self.it += 1
        Ap, As = getattr(self.primary, "last_ADA", None), getattr(self.shadow, "last_ADA", None)
        self.cur = {"iter": self.it, "d": self._rel(self.Ls["d"], Lp["d"]),
                    "ada": float(abs(As - Ap).max() / abs(Ap).max()) if Ap is not None and As is not None else np.nan,
                    "ada_asym": float(abs(As - As.T).max() / abs(As).max()) if As is not None else np.nan, "nskip": (self.Ls["nskip"], Lp["nskip"]),
                    "nadd": (self.Ls["nadd"], Lp["nadd"]), "fw": 0.0, "bw": 0.0, "nsolves": 0,
                    "dcond": float(np.max(Lp["d"]) / max(np.min(Lp["d"]), 1e-300))}
        self.records.append(self.cur)
        return Lp

    def fwdpr1(self, L, b):                                        # (no dense columns in the shadow runs: the identity)
        return self.primary.fwdpr1(L, b)

    def bwdpr1(self, L, b):
        return self.primary.bwdpr1(L, b)

    def fw(self, L, r):
        a, b = self.primary.fw(L, r), self.shadow.fw(self.Ls, r)
        self.cur["fw"] = max(self.cur["fw"], self._rel(b, a)); self.cur["nsolves"] += 1
        self.rhs, self.fw_shadow = vec(r), vec(b)
        return a

    def bw(self, L, r):
        """Always the second half of x = ADA \\ rhs (wrapPcg.m:56-59): next to the distance between the two results, the
        normwise backward error |ADA x - rhs| / (|ADA| |x| + |rhs|) of each path is kept -- the two factors differ at the
        1e-11 level late in a run, so on an ill-conditioned ADA' the solutions legitimately differ by cond x that."""
        a = self.primary.bw(L, r)
        b = self.shadow.bw(self.Ls, self.fw_shadow / vec(self.Ls["d"]))              # the shadow's own (fw ./ d)
        self.cur["bw"] = max(self.cur["bw"], self._rel(b, a))
        if self.cur["nskip"] == (0, 0):
            for key, hot, x in (("berr_ref", self.primary, vec(a)), ("berr_lib", self.shadow, vec(b))):
                ADA = getattr(hot, "last_ADA", None)                # each path against the ADA' it factored itself
                if ADA is not None:
                    nA = abs(ADA).sum(axis=0).max()
                    e = np.abs(ADA @ x - self.rhs).max() / (nA * np.abs(x).max() + np.abs(self.rhs).max())
                    self.cur[key] = max(self.cur.get(key, 0.0), float(e))
        return a


# ----------------------------------------------------------------------------------------------- cone algebra

class Sedumi(_loop.Sedumi):
    """the product's loop with the oracle as the MEX host of everything outside the hot path, and the reference hot path by default"""

    def __init__(self, At, b, c, K, hot=None, G=None, pars=None, internal=False):
        G = G or gl.Glue()
        super().__init__(At, b, c, K, hot=hot or RefHot(G), G=G, pars=pars, internal=internal)


def load_example(name):
    """At, b, c, K of an example problem of the reference (examples/*.mat).  Build container only."""
    return _loop.load_mat(f"/root/reference/examples/{name}.mat")

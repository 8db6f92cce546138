"""Resident plan: problem data, ADA', the factor and the work vectors stay in HBM across the calls of an
IPM iteration (tier 2 of include/sedumi_hip.h).

    plan = Plan(device=0)
    plan.set_chol(L, ADA_pattern)                  # once per solve  (sedumi.m:382-387)
    plan.set_ada(At, Ablkjc, K, DAtq_pattern)      # once per solve  (sedumi.m:356-378)
    # every iteration (sedumi.m:442-473):
    plan.upload("dl", d.l); plan.upload("ddet", d.det); plan.upload("qpr", DAt.q values); plan.upload("udsqr", udsqr)
    plan.getada(); plan.blkchol(pars); plan.upload("rhs", r); plan.ldlsolve(); y = plan.download("y")
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import scipy.sparse as sp

from . import capi
from .capi import SdmError, check, f64, i64, pf, pi


class Plan:
    def __init__(self, device=0, stream=None):
        self._lib = capi.lib()
        self._p = self._lib.sdm_plan_create(int(device), C.c_void_p(stream) if stream else None)
        if not self._p:
            raise SdmError(self._lib.sdm_last_error().decode())
        self.m = 0
        self.nnzL = 0
        self.nnzADA = 0

    def close(self):
        if self._p:
            self._lib.sdm_plan_destroy(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------ symbolic
    def set_chol(self, L, ADA):
        LL = sp.csc_matrix(L["L"])
        LL.sort_indices()
        ADA = sp.csc_matrix(ADA)
        ADA.sort_indices()
        m = LL.shape[0]
        Ljc, Lir = i64(LL.indptr), i64(LL.indices)
        perm = i64(np.asarray(L["perm"], dtype=np.float64)) - 1
        xs = i64(np.asarray(L["xsuper"], dtype=np.float64)) - 1
        jc, ir = i64(ADA.indptr), i64(ADA.indices)
        check(self._lib.sdm_plan_set_chol(C.c_void_p(self._p), C.c_int64(m), pi(Ljc), pi(Lir), pi(perm),
                                          C.c_int64(xs.size - 1), pi(xs), pi(jc), pi(ir)))
        self.m, self.nnzL, self.nnzADA = m, int(Ljc[-1]), int(jc[-1])
        self.L_pattern, self.ADA_pattern = LL, ADA

    def set_ada(self, At, Ablkjc, K, Qpattern=None):
        At = sp.csc_matrix(At)
        At.sort_indices()
        m = At.shape[1]
        lpN = int(np.asarray(K["l"]).ravel()[0])
        q = i64(np.asarray(K["q"], dtype=np.float64))
        s = i64(np.asarray(K["s"], dtype=np.float64))
        rsdpN = int(np.asarray(K.get("rsdpN", s.size)).ravel()[0])
        blkstart = np.asarray(K["blkstart"], dtype=np.float64).ravel()
        qb = i64(blkstart[1:2 + q.size]) - 1
        psd = i64(blkstart[1 + q.size:]) - 1
        Ajc, Air, Apr = i64(At.indptr), i64(At.indices), f64(At.data)
        Ajc_psd = i64(np.asarray(Ablkjc, dtype=np.float64)[:, 2])
        if Qpattern is None or q.size == 0:
            Qjc, Qir = np.zeros(m + 1, dtype=np.int64), np.zeros(1, dtype=np.int64)
            self.nnzQ = 0
        else:
            Q = sp.csc_matrix(Qpattern)
            Q.sort_indices()
            Qjc, Qir = i64(Q.indptr), i64(Q.indices)
            self.nnzQ = int(Qjc[-1])
        Kc, keep = capi.make_cone(lpN, q, s, rsdpN)
        check(self._lib.sdm_plan_set_ada(C.c_void_p(self._p), C.c_int64(At.shape[0]), C.c_int64(m), pi(Ajc), pi(Air), pf(Apr),
                                         pi(Ajc_psd), C.byref(Kc), pi(qb), pi(psd), pi(Qjc), pi(Qir)))
        del keep

    # ------------------------------------------------------------- buffers
    def upload(self, name, arr):
        a = f64(arr)
        check(self._lib.sdm_plan_upload(C.c_void_p(self._p), name.encode(), pf(a), C.c_int64(a.size)))

    def download(self, name, n=None):
        if n is None:
            n = {"ada": self.nnzADA, "lpr": self.nnzL}.get(name, self.m)
        out = np.zeros(int(n), dtype=np.float64)
        check(self._lib.sdm_plan_download(C.c_void_p(self._p), name.encode(), pf(out), C.c_int64(out.size)))
        return out

    def devptr(self, name):
        n = C.c_int64(0)
        p = self._lib.sdm_plan_devptr(C.c_void_p(self._p), name.encode(), C.byref(n))
        if not p:
            raise SdmError(self._lib.sdm_last_error().decode())
        return p, n.value

    # ------------------------------------------------------------- numeric
    def getada(self):
        check(self._lib.sdm_plan_getada(C.c_void_p(self._p)))

    def invcholfac(self, perm=None):
        """udsqr = invcholfac(u, K, perm) on the device (sedumi.m:452): reads plan buffer "u" (upload it first), leaves
        the result in "udsqr" for getada().  perm: d.perm (1-based doubles, concatenated per block) or None."""
        if perm is None:
            check(self._lib.sdm_plan_invcholfac(C.c_void_p(self._p), None))
        else:
            p0 = np.ascontiguousarray(np.asarray(perm, dtype=np.float64).ravel() - 1, dtype=np.int64)
            check(self._lib.sdm_plan_invcholfac(C.c_void_p(self._p), p0.ctypes.data_as(C.POINTER(C.c_int64))))

    # ------------------------------------------- operators around the solves (wrapPcg.m / loopPcg.m)
    def pcg_init(self, dense_cols=None, denseA=None):
        """Work vectors "xN" / "psd" and, optionally, the dense columns of Amul.m:50-56 (dense.cols 1-based, dense.A m x nden)."""
        if dense_cols is None or np.size(dense_cols) == 0:
            check(self._lib.sdm_plan_pcg_init(C.c_void_p(self._p), C.c_int64(0), None, None))
        else:
            c0 = i64(np.asarray(dense_cols, dtype=np.float64)) - 1
            Ad = f64(np.asarray(denseA.todense() if sp.issparse(denseA) else denseA).ravel(order="F"))
            check(self._lib.sdm_plan_pcg_init(C.c_void_p(self._p), C.c_int64(c0.size), pi(c0), pf(Ad)))

    def amul(self, transp=0):
        """transp = 0: "rhs" = At' * "xN" (+ dense part); transp = 1: "xN" = At * "y"   (Amul.m:43-56)."""
        check(self._lib.sdm_plan_amul(C.c_void_p(self._p), int(transp)))

    def vecsym(self):
        check(self._lib.sdm_plan_vecsym(C.c_void_p(self._p)))

    def psdscale(self, transp=0, use_perm=False):
        check(self._lib.sdm_plan_psdscale(C.c_void_p(self._p), int(transp), 1 if use_perm else 0))

    def load_factor(self, LL, Ld=None):
        """Make an externally computed factor resident: L.L values on the plan's pattern and L.d."""
        LL = sp.csc_matrix(LL); LL.sort_indices()
        if not (np.array_equal(LL.indptr, self.L_pattern.indptr) and np.array_equal(LL.indices, self.L_pattern.indices)):
            raise SdmError("load_factor: L.L does not have the plan's pattern")
        v = f64(LL.data)
        dd = f64(Ld) if Ld is not None else None
        check(self._lib.sdm_plan_load_factor(C.c_void_p(self._p), pf(v), pf(dd) if dd is not None else None))

    def set_dense(self, symLden):
        """Symbolic data of the dense columns (symbcholden.m:43-55): symLden = {LAD, dz, perm, first} as finsymbden
        returns it (1-based perm / first).  Call after set_chol."""
        LAD = sp.csc_matrix(symLden["LAD"]); LAD.sort_indices()
        dz = symLden["dz"]
        dz = dz.X if hasattr(dz, "X") else sp.csc_matrix(dz)          # rows in the order incorder introduced them
        nden = LAD.shape[1]
        jc, ir = i64(LAD.indptr), i64(LAD.indices)
        dzjc, dzir = i64(dz.indptr[:nden + 1]), i64(dz.indices)
        cp = i64(np.asarray(symLden["perm"], dtype=np.float64)) - 1
        fi = i64(np.asarray(symLden["first"], dtype=np.float64)) - 1
        check(self._lib.sdm_plan_set_dense(C.c_void_p(self._p), C.c_int64(nden), pi(jc), pi(ir), pi(dzjc), pi(dzir), pi(cp), pi(fi)))
        self.nden, self._dz_pnnz = nden, int(np.sum(dzjc[1:]))

    def deninfac(self, smult, maxuden=500.0):
        """LAD = L \\ Ad(perm,:), Lden = dpr1fact(LAD, L.d, symLden, smult, maxuden) on the device (upload "ad" first).
        Returns False (earlier versions: True when a host algorithm had to take over; the whole of dpr1fact runs on the device now)."""
        sm = f64(smult)
        fb = C.c_int(0)
        check(self._lib.sdm_plan_deninfac(C.c_void_p(self._p), pf(sm), C.c_double(float(maxuden)), C.byref(fb)))
        return bool(fb.value)

    def lden(self):
        """The resident product-form factors as dpr1fact returns them: ({betajc (1-based), beta, p, pivperm, dopiv}, Ld)."""
        n, pn = self.nden, max(self._dz_pnnz, 1)
        betajc, dopiv, pivperm = np.zeros(n + 1, dtype=np.int64), np.zeros(n, dtype=np.int64), np.zeros(pn, dtype=np.int64)
        beta, p, Ld = np.zeros(pn), np.zeros(pn), np.zeros(self.m)
        npp = C.c_int64(0)
        check(self._lib.sdm_plan_lden(C.c_void_p(self._p), pi(betajc), pf(beta), pf(p), pi(pivperm), C.byref(npp), pi(dopiv), pf(Ld)))
        return ({"betajc": (betajc + 1).astype(np.float64), "beta": beta[:betajc[-1]], "p": p[:self._dz_pnnz],
                 "pivperm": pivperm[:npp.value].astype(np.float64), "dopiv": dopiv.astype(np.float64)}, Ld)

    def set_growth_max(self, growth_max):
        """Growth bound above which a diagonal super-block of L is solved by substitution instead of its explicit
        inverse (0 = substitution everywhere); effective from the next blkchol."""
        check(self._lib.sdm_plan_set_growth_max(C.c_void_p(self._p), C.c_double(float(growth_max))))

    def set_refinement(self, mode, refine_max=1e10):
        """Blocks beyond growth_max but within refine_max: 1 = inverse + iterative refinement once a solve has met one (default),
        0 = always substitution, 2 = refinement launches in every solve."""
        check(self._lib.sdm_plan_set_refinement(C.c_void_p(self._p), C.c_int(int(mode)), C.c_double(float(refine_max))))

    def set_one_launch_fronts(self, on):
        """False: the NEXT set_chol plans every front on the launch-per-panel path (the comparison switch of tests and tools)."""
        check(self._lib.sdm_plan_set_one_launch_fronts(C.c_void_p(self._p), C.c_int(1 if on else 0)))

    def set_tile_workgroups(self, n):
        """Workgroups the update tiles of a big front's panel launch are dealt to, for the NEXT set_chol (0 = the device's compute units)."""
        check(self._lib.sdm_plan_set_tile_workgroups(C.c_void_p(self._p), C.c_int(int(n))))

    def set_one_launch_inverse(self, on):
        """False: the super-block inverses of small problems by a launch per stage too (k_sinv128 + k_stile instead of k_sprep)."""
        check(self._lib.sdm_plan_set_one_launch_inverse(C.c_void_p(self._p), C.c_int(1 if on else 0)))

    def set_solve_width(self, width):
        """Super-block width of the solves for the NEXT set_chol (0 = automatic, or a power of two in 256 .. 2048)."""
        check(self._lib.sdm_plan_set_solve_width(C.c_void_p(self._p), C.c_int64(int(width))))

    def solve_width(self):
        """Super-block width of the solves in force (chosen at set_chol)."""
        w = C.c_int64(0)
        check(self._lib.sdm_plan_get_solve_width(C.c_void_p(self._p), C.byref(w)))
        return w.value

    def solve_stats(self):
        """(super-blocks, blocks on the substitution fallback, largest growth) of the last factorisation."""
        nb, bad, g = C.c_int64(0), C.c_int64(0), C.c_double(0.0)
        check(self._lib.sdm_plan_solve_stats(C.c_void_p(self._p), C.byref(nb), C.byref(bad), C.byref(g)))
        return nb.value, bad.value, g.value

    def getdatq(self):
        """qpr = values of DAt.q (getDAtm.m:39-44) from the resident "q1", "q2" (upload them first)."""
        check(self._lib.sdm_plan_getdatq(C.c_void_p(self._p)))

    def getada_cols(self, j0, j1):
        """Columns j0 <= j < j1 of ADA' (and absd[j0:j1]) only; the rest of "ada" is left untouched."""
        check(self._lib.sdm_plan_getada_cols(C.c_void_p(self._p), C.c_int64(int(j0)), C.c_int64(int(j1))))

    def copy(self, name, tensor, offset, nelem, to_plan):
        """Device-to-device copy between plan buffer `name`[offset:offset+nelem] and a contiguous float64 torch
        tensor living on the plan's device (RCCL send / receive buffers of sedumi_amd.dist)."""
        if tensor.dtype.itemsize != 8 or not tensor.is_contiguous() or tensor.numel() < nelem:
            raise SdmError("copy: need a contiguous float64 tensor with at least nelem elements")
        if not tensor.is_cuda and capi.backend() != "emu":
            # host tensor next to a real GPU plan (gloo smoke runs): staged through the host copies of the buffer
            import torch
            n = self.devptr(name)[1]
            full = self.download(name, n)
            if to_plan:
                full[offset:offset + nelem] = tensor[:nelem].numpy()
                self.upload(name, full)
            else:
                tensor[:nelem] = torch.from_numpy(full[offset:offset + nelem].copy())
            return
        check(self._lib.sdm_plan_copy(C.c_void_p(self._p), name.encode(), C.c_void_p(tensor.data_ptr()), C.c_int64(int(offset)),
                                      C.c_int64(int(nelem)), 1 if to_plan else 0))

    def blkchol(self, pars=None, use_absd=True):
        cp = capi.CholPars(1e-12, 5e5, 1e-20)          # checkpars.m:144-168
        if pars:
            cp.canceltol = float(pars.get("canceltol", cp.canceltol))
            cp.maxu = float(pars.get("maxu", cp.maxu))
            cp.abstol = max(float(pars.get("abstol", cp.abstol)), 0.0)
        check(self._lib.sdm_plan_blkchol(C.c_void_p(self._p), C.byref(cp), 1 if use_absd else 0))

    def blkchol_wait(self, pars=None, use_absd=False):
        """blkchol, waited for, with the one recovery of sdm_plan_blkchol_wait (what blkchol.mex calls)."""
        cp = capi.CholPars(1e-12, 5e5, 1e-20)
        if pars:
            cp.canceltol = float(pars.get("canceltol", cp.canceltol))
            cp.maxu = float(pars.get("maxu", cp.maxu))
            cp.abstol = max(float(pars.get("abstol", cp.abstol)), 0.0)
        check(self._lib.sdm_plan_blkchol_wait(C.c_void_p(self._p), C.byref(cp), 1 if use_absd else 0))

    # ---- the factorisation and the solve level by level (sedumi_amd.dist.SeparatorShardedSolver)
    def set_active_supernodes(self, active):
        """Before set_chol: the supernodes (0/1 per supernode of the symbolic factor) this plan factors and solves."""
        a = np.ascontiguousarray(active, dtype=np.int32)
        check(self._lib.sdm_plan_set_active_supernodes(C.c_void_p(self._p), a.ctypes.data_as(C.POINTER(C.c_int)), C.c_int64(a.size)))

    def front_layout(self, nsuper):
        """dict of int64 arrays per supernode: etree level, slice of "fronts" (foff, fsize), slice of "wvec" (woff, ms), columns (first, ns);
        plus nlevels."""
        out = {k: np.zeros(nsuper, dtype=np.int64) for k in ("level", "foff", "fsize", "woff", "ms", "first", "ns")}
        nl = C.c_int64(0)
        check(self._lib.sdm_plan_front_layout(C.c_void_p(self._p), C.byref(nl), *[pi(out[k]) for k in ("level", "foff", "fsize", "woff", "ms", "first", "ns")]))
        out["nlevels"] = nl.value
        return out

    def blkchol_begin(self, pars=None, use_absd=True):
        cp = capi.CholPars(1e-12, 5e5, 1e-20)
        if pars:
            cp.canceltol = float(pars.get("canceltol", cp.canceltol))
            cp.maxu = float(pars.get("maxu", cp.maxu))
            cp.abstol = max(float(pars.get("abstol", cp.abstol)), 0.0)
        check(self._lib.sdm_plan_blkchol_begin(C.c_void_p(self._p), C.byref(cp), 1 if use_absd else 0))

    def blkchol_levels(self, l0, l1, extend_only=False):
        check(self._lib.sdm_plan_blkchol_levels(C.c_void_p(self._p), C.c_int64(int(l0)), C.c_int64(int(l1)), 1 if extend_only else 0))

    # ---- one dense front block-column-cyclically over several ranks (sedumi_amd.dist.BlockCyclicFactor)
    def set_column_owner(self, world, rank, blk=1):
        """Tile column c (64 columns) of the front belongs to rank (c // blk) % world; before blkchol_begin."""
        check(self._lib.sdm_plan_set_column_owner(C.c_void_p(self._p), C.c_int(int(world)), C.c_int(int(rank)), C.c_int(int(blk))))

    def blkchol_panels(self, l0, l1, pan0, pan1):
        """The panel launches pan0 .. pan1-1 of the levels l0 .. l1-1 (launch-per-panel path)."""
        check(self._lib.sdm_plan_blkchol_panels(C.c_void_p(self._p), C.c_int64(int(l0)), C.c_int64(int(l1)), C.c_int64(int(pan0)), C.c_int64(int(pan1))))

    def panel_slice(self, panel):
        """(offset, nelem) of the slice of "fronts" that holds the columns of panel `panel` of the (one) front."""
        off, n = C.c_int64(0), C.c_int64(0)
        check(self._lib.sdm_plan_panel_record(C.c_void_p(self._p), C.c_int64(int(panel)), C.c_int(-1), C.byref(off), C.byref(n)))
        return off.value, n.value

    def panel_record(self, panel, unpack):
        """Pack (owner) / unpack (receiver) the record of a finished panel in the plan buffer "panelrec"; returns the slice
        (offset, nelem) of "fronts" that holds the panel's columns."""
        off, n = C.c_int64(0), C.c_int64(0)
        check(self._lib.sdm_plan_panel_record(C.c_void_p(self._p), C.c_int64(int(panel)), C.c_int(1 if unpack else 0), C.byref(off), C.byref(n)))
        return off.value, n.value

    def blkchol_end(self):
        check(self._lib.sdm_plan_blkchol_end(C.c_void_p(self._p)))

    def solve_levels(self, what, l0, l1):
        """what: 1 assembly of the forward sweep of levels l0 .. l1-1, 2 their forward sweep without it, 3 both, 4 backward sweep."""
        check(self._lib.sdm_plan_solve_levels(C.c_void_p(self._p), C.c_int(int(what)), C.c_int64(int(l0)), C.c_int64(int(l1))))

    def pivots(self):
        m = self.m
        ns, na = C.c_int64(0), C.c_int64(0)
        si, ai = np.zeros(m, dtype=np.int64), np.zeros(m, dtype=np.int64)
        sv, av = np.zeros(m), np.zeros(m)
        self._lib.sdm_plan_pivots.argtypes = None
        check(self._lib.sdm_plan_pivots(C.c_void_p(self._p), C.byref(ns), pi(si), pf(sv), C.byref(na), pi(ai), pf(av)))
        return (si[:ns.value], sv[:ns.value]), (ai[:na.value], av[:na.value])

    def fwsolve(self):
        check(self._lib.sdm_plan_fwsolve(C.c_void_p(self._p)))

    def bwsolve(self):
        check(self._lib.sdm_plan_bwsolve(C.c_void_p(self._p)))

    def ldlsolve(self):
        check(self._lib.sdm_plan_ldlsolve(C.c_void_p(self._p)))

    # ---------------------------------------------------------------- graphs
    def graph_capture(self, fn):
        """Record the asynchronous plan calls made by fn() into one hipGraph; returns the id for graph_launch."""
        check(self._lib.sdm_plan_graph_begin(C.c_void_p(self._p)))
        try:
            fn()
        finally:
            gid = C.c_int(-1)
            rc = self._lib.sdm_plan_graph_end(C.c_void_p(self._p), C.byref(gid))
        check(rc)
        return gid.value

    def graph_launch(self, gid):
        check(self._lib.sdm_plan_graph_launch(C.c_void_p(self._p), C.c_int(gid)))

    def sync(self):
        check(self._lib.sdm_plan_sync(C.c_void_p(self._p)))

    # ------------------------------------------------------ per-kernel timing
    def kprof(self, on):
        check(self._lib.sdm_plan_kprof_enable(C.c_void_p(self._p), 1 if on else 0))

    def kprof_summary(self):
        buf = C.create_string_buffer(8192)
        check(self._lib.sdm_plan_kprof_summary(C.c_void_p(self._p), buf, C.c_int64(8192)))
        out = {}
        for rec in buf.value.decode().split(";"):
            if rec:
                name, calls, ms = rec.rsplit(":", 2)
                c0, m0 = out.get(name, (0, 0.0))
                out[name] = (c0 + int(calls), m0 + float(ms))
        return out

    # --------------------------------------------------------------- timing
    def timer_begin(self, slot):
        check(self._lib.sdm_plan_timer_begin(C.c_void_p(self._p), slot))

    def timer_end(self, slot):
        check(self._lib.sdm_plan_timer_end(C.c_void_p(self._p), slot))

    def timer_ms(self, slot):
        ms = C.c_float(0)
        check(self._lib.sdm_plan_timer_ms(C.c_void_p(self._p), slot, C.byref(ms)))
        return float(ms.value)

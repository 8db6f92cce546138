"""sedumi_amd/mexhost.py -- run mexFunction binaries without MATLAB / Octave.

ctypes driver for ``sedumi_amd/lib/libsdm_mexhost.so`` (sedumi_amd/mexhost: a self-written implementation of the slice
of the MEX C API that SeDuMi's hot-path gateways use).  It marshals numpy / scipy.sparse / dict values to ``mxArray``
and back and calls a ``mexFunction`` exactly as the interpreter would, so e.g.

    host = MexHost(mex_dir)            # directory of <name>.so files that export mexFunction
    LL, Ld, Lskip, Ladd = host.call("blkchol", 4, L, ADA, pars, absd)

is the MATLAB call ``[L.L,L.d,L.skip,L.add] = blkchol(L,ADA,pars,absd)`` (sedumi.m:458).  ``mex_dir`` defaults to
``sedumi_amd/lib/mex``: the shims of sedumi_amd/mexshims built against this host by ``sedumi_amd.build`` (what
bench.py's ``mex_inclusive`` leg times).  ``call_mx`` keeps inputs and outputs as ``mxArray`` handles, so that an array
one gateway returned is handed to the next one BY REFERENCE, as MATLAB does.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")
HOST_LIB = os.path.join(LIB_DIR, "libsdm_mexhost.so")
MEX_DIR = os.path.join(LIB_DIR, "mex")

MX_DOUBLE, MX_SPARSE, MX_STRUCT = 0, 1, 2


class _MxArray(C.Structure):
    pass


_MxArray._fields_ = [
    ("kind", C.c_int),
    ("m", C.c_size_t),
    ("n", C.c_size_t),
    ("pr", C.POINTER(C.c_double)),
    ("ir", C.POINTER(C.c_size_t)),
    ("jc", C.POINTER(C.c_size_t)),
    ("nzmax", C.c_size_t),
    ("nfields", C.c_int),
    ("fnames", C.POINTER(C.c_char_p)),
    ("fvals", C.POINTER(C.POINTER(_MxArray))),
]
_MxP = C.POINTER(_MxArray)


class RawSparse:
    """A sparse matrix to be handed to a MEX exactly as stored (no index sorting): incorder's `dz`
    lists the rows of every column in the order in which they were introduced (incorder.c:171-208)."""

    def __init__(self, X):
        self.X = sp.csc_matrix(X)


class MexError(RuntimeError):
    """What mexErrMsgTxt raised."""


class MexHost:
    """Loads the host library and (lazily) the per-MEX shared objects of `mex_dir`."""

    def __init__(self, mex_dir: str | None = None, host_lib: str = HOST_LIB):
        self.dir = mex_dir or MEX_DIR          # where <name>.so with a mexFunction are looked up
        if not os.path.exists(host_lib):
            raise MexError(f"{host_lib} missing: run `python -m sedumi_amd.build`")
        self.shim = C.CDLL(host_lib, mode=C.RTLD_GLOBAL)
        s = self.shim
        s.mxCreateDoubleMatrix.restype = _MxP
        s.mxCreateDoubleMatrix.argtypes = [C.c_size_t, C.c_size_t, C.c_int]
        s.mxCreateSparse.restype = _MxP
        s.mxCreateSparse.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t, C.c_int]
        s.mexhost_new_struct.restype = _MxP
        s.mxSetField.argtypes = [_MxP, C.c_size_t, C.c_char_p, _MxP]
        s.mxGetField.restype = _MxP
        s.mxGetField.argtypes = [_MxP, C.c_size_t, C.c_char_p]
        s.mxDestroyArray.argtypes = [_MxP]
        s.mexhost_call.restype = C.c_int
        s.mexhost_call.argtypes = [C.c_void_p, C.c_int, C.POINTER(_MxP), C.c_int, C.POINTER(_MxP)]
        s.mexhost_last_error.restype = C.c_char_p
        self._mex = {}
        self.error = MexError

    # ------------------------------------------------------------ python -> mx
    def to_mx(self, v):
        s = self.shim
        if isinstance(v, dict):
            a = s.mexhost_new_struct()
            for k, val in v.items():
                s.mxSetField(a, 0, k.encode(), self.to_mx(val))
            return a
        if isinstance(v, RawSparse):
            v = v.X                                        # row order inside the columns is part of the data
        elif sp.issparse(v):
            v = sp.csc_matrix(v)
            if not v.has_sorted_indices:
                v = v.copy()
                v.sort_indices()
        if sp.issparse(v):
            m, n = v.shape
            nnz = int(v.indptr[-1])
            a = s.mxCreateSparse(m, n, max(nnz, 1), 0)
            jc = np.ascontiguousarray(v.indptr, dtype=np.uint64)
            C.memmove(a.contents.jc, jc.ctypes.data, jc.nbytes)
            if nnz:
                ir = np.ascontiguousarray(v.indices[:nnz], dtype=np.uint64)
                pr = np.ascontiguousarray(v.data[:nnz], dtype=np.float64)
                C.memmove(a.contents.ir, ir.ctypes.data, ir.nbytes)
                C.memmove(a.contents.pr, pr.ctypes.data, pr.nbytes)
            return a
        arr = np.asarray(v, dtype=np.float64)
        if arr.ndim == 0:
            arr = arr.reshape(1, 1)
        elif arr.ndim == 1:
            arr = arr.reshape(-1, 1)
        m, n = arr.shape
        a = s.mxCreateDoubleMatrix(m, n, 0)
        if arr.size:
            f = np.asfortranarray(arr)
            C.memmove(a.contents.pr, f.ctypes.data, f.nbytes)
        return a

    # ------------------------------------------------------------ mx -> python
    def from_mx(self, a):
        if not a:
            return None
        c = a.contents
        if c.kind == MX_DOUBLE:
            size = c.m * c.n
            out = np.empty(size, dtype=np.float64)
            if size:
                C.memmove(out.ctypes.data, c.pr, size * 8)
            return out.reshape((c.m, c.n), order="F")
        if c.kind == MX_SPARSE:
            jc = np.empty(c.n + 1, dtype=np.uint64)
            C.memmove(jc.ctypes.data, c.jc, jc.nbytes)
            nnz = int(jc[-1])
            ir = np.empty(nnz, dtype=np.uint64)
            pr = np.empty(nnz, dtype=np.float64)
            if nnz:
                C.memmove(ir.ctypes.data, c.ir, nnz * 8)
                C.memmove(pr.ctypes.data, c.pr, nnz * 8)
            return sp.csc_matrix((pr, ir.astype(np.int64), jc.astype(np.int64)), shape=(c.m, c.n))
        out = {}
        for i in range(c.nfields):
            out[c.fnames[i].decode()] = self.from_mx(c.fvals[i])
        return out

    # ------------------------------------------------------------------- call
    def _fn(self, name):
        if name not in self._mex:
            lib = C.CDLL(os.path.join(self.dir, name + ".so"))
            self._mex[name] = C.cast(lib.mexFunction, C.c_void_p)
        return self._mex[name]

    def call(self, name, nlhs, *args):
        """Run MEX ``name`` with ``nlhs`` outputs; returns a tuple (or the single value when nlhs<=1)."""
        fn = self._fn(name)
        nrhs = len(args)
        prhs = (_MxP * max(nrhs, 1))()
        for i, v in enumerate(args):
            prhs[i] = self.to_mx(v)
        nout = max(nlhs, 1)
        plhs = (_MxP * nout)()
        rc = self.shim.mexhost_call(fn, nlhs, plhs, nrhs, prhs)
        try:
            if rc:
                raise self.error(f"{name}: {self.shim.mexhost_last_error().decode()}")
            outs = tuple(self.from_mx(plhs[i]) for i in range(nout))
        finally:
            for i in range(nrhs):
                self.shim.mxDestroyArray(prhs[i])
            if not rc:
                for i in range(nout):
                    if plhs[i]:
                        self.shim.mxDestroyArray(plhs[i])
        return outs[0] if nlhs <= 1 else outs

    def set_global(self, name, value):
        """MATLAB `global name; name = value` for the MEX files that use mexGetVariablePtr / mexPutVariable."""
        self.shim.mexhost_set_global.argtypes = [C.c_char_p, _MxP]
        if value is None:                       # `clear global name`
            self.shim.mexhost_set_global(name.encode(), None)
            return
        mx = self.to_mx(value)
        try:
            self.shim.mexhost_set_global(name.encode(), mx)
        finally:
            self.shim.mxDestroyArray(mx)

    def get_global(self, name):
        self.shim.mexhost_get_global.restype = _MxP
        self.shim.mexhost_get_global.argtypes = [C.c_char_p]
        mx = self.shim.mexhost_get_global(name.encode())
        return self.from_mx(mx) if mx else None

    def timed_call(self, name, nlhs, args, reps=1):
        """Like call() but marshals once and times only mexFunction itself.
        Returns (seconds_per_call_list, outputs_of_last_call)."""
        import time
        fn = self._fn(name)
        nrhs = len(args)
        prhs = (_MxP * max(nrhs, 1))()
        for i, v in enumerate(args):
            prhs[i] = self.to_mx(v)
        nout = max(nlhs, 1)
        times, outs = [], None
        try:
            for _ in range(reps):
                plhs = (_MxP * nout)()
                t0 = time.perf_counter()
                rc = self.shim.mexhost_call(fn, nlhs, plhs, nrhs, prhs)
                times.append(time.perf_counter() - t0)
                if rc:
                    raise self.error(f"{name}: {self.shim.mexhost_last_error().decode()}")
                outs = tuple(self.from_mx(plhs[i]) for i in range(nout))
                for i in range(nout):
                    if plhs[i]:
                        self.shim.mxDestroyArray(plhs[i])
        finally:
            for i in range(nrhs):
                self.shim.mxDestroyArray(prhs[i])
        return times, (outs[0] if nlhs <= 1 else outs)

    # ------------------------------------------------- by reference: mxArray handles in, mxArray handles out
    def call_mx(self, name, nlhs, *mx_args):
        """mexFunction ``name`` on mxArray handles (from to_mx or an earlier call_mx).  Nothing is copied or freed: the caller
        keeps the inputs and owns the returned handles (free()).  Returns (seconds inside mexFunction, tuple of handles)."""
        import time
        fn = self._fn(name)
        nrhs = len(mx_args)
        prhs = (_MxP * max(nrhs, 1))(*mx_args)
        nout = max(nlhs, 1)
        plhs = (_MxP * nout)()
        t0 = time.perf_counter()
        rc = self.shim.mexhost_call(fn, nlhs, plhs, nrhs, prhs)
        dt = time.perf_counter() - t0
        if rc:
            raise self.error(f"{name}: {self.shim.mexhost_last_error().decode()}")
        return dt, tuple(plhs[i] for i in range(nout))

    def free(self, *handles):
        for h in handles:
            if h:
                self.shim.mxDestroyArray(h)

    def field(self, mx_struct, name):
        """handle of a struct field (owned by the struct)"""
        return self.shim.mxGetField(mx_struct, 0, name.encode())

    def set_field(self, mx_struct, name, mx_value):
        """struct.name = value BY REFERENCE (the struct owns the value afterwards; a previous value is destroyed), like
        MATLAB's `L.L = LL` of a returned array"""
        old = self.shim.mxGetField(mx_struct, 0, name.encode())
        if old and C.addressof(old.contents) != C.addressof(mx_value.contents):
            self.shim.mxDestroyArray(old)
        self.shim.mxSetField(mx_struct, 0, name.encode(), mx_value)

    def values(self, mx):
        """numpy view (no copy) of the values of a full or sparse mxArray"""
        c = mx.contents
        n = c.m * c.n if c.kind == MX_DOUBLE else int(c.jc[c.n])
        return np.ctypeslib.as_array(c.pr, shape=(max(n, 1),))[:n]


def iteration_units(host, A, Ajc3, Aord, K, dstruct, DAt, udsqr, L, ADA0, pars_chol, rhs, nunits, nsolve=4, check=None):
    """`nunits` IPM iteration units as an unmodified sedumi.m issues them (sedumi.m:450-458, wrapPcg.m:56-59) through the mexFunction
    binaries of `host`, with every array handed from one gateway to the next BY REFERENCE (call_mx), the way MATLAB passes them:

        ADA = getada1(ADA, A, Ablkjc(:,3), Aord.lqperm, d, K.qblkstart); ADA = getada2(ADA, DAt, Aord, K);
        [ADA, absd] = getada3(ADA, A, Ablkjc(:,3), Aord, udsqr, K);     [L.L, L.d, L.skip, L.add] = blkchol(L, ADA, pars.chol, absd);
        nsolve x:  p = fwblkslv(L, r);  y = p ./ L.d;  p = bwblkslv(L, y)      (the ./ on the host, as wrapPcg.m:57)

    The inputs are marshalled once (they are the same arrays every iteration in MATLAB too); only the time inside mexFunction is
    measured.  Returns (per-unit dicts of seconds by stage, last solution).  check(ADA, absd, LL, Ld) is called with numpy copies."""
    to = host.to_mx
    mxA, mxAjc, lqperm, qblk = to(A), to(np.asarray(Ajc3, dtype=np.float64)), to(Aord["lqperm"]), to(K["qblkstart"])
    mxAord, mxK, mxD, mxDAt, mxud, mxpars = to(Aord), to(K), to(dstruct), to(DAt), to(np.asarray(udsqr, dtype=np.float64).reshape(-1, 1)), to(dict(pars_chol))
    mxADA0, mxL, b = to(ADA0), to(L), to(np.asarray(rhs, dtype=np.float64).reshape(-1, 1))
    times, y = [], None
    try:
        for _ in range(nunits):
            t1, (A1,) = host.call_mx("getada1", 1, mxADA0, mxA, mxAjc, lqperm, mxD, qblk)
            t2, (A2,) = host.call_mx("getada2", 1, A1, mxDAt, mxAord, mxK)
            t3, (A3, absd) = host.call_mx("getada3", 2, A2, mxA, mxAjc, mxAord, mxud, mxK)
            t4, (LL, Ld, Lskip, Ladd) = host.call_mx("blkchol", 4, mxL, A3, mxpars, absd)
            if check is not None:
                check(host.from_mx(A3), host.from_mx(absd), host.from_mx(LL), host.from_mx(Ld))
            host.set_field(mxL, "L", LL)                            # L.L = LL: by reference
            dv = host.values(Ld)
            dsafe = np.where(dv > 0, dv, 1.0)                       # skipped pivots act as 1 (deninfac.m:89-94)
            ts = 0.0
            for _s in range(nsolve):
                tf, (p,) = host.call_mx("fwblkslv", 1, mxL, b)
                host.values(p)[:] /= dsafe
                tb, (yy,) = host.call_mx("bwblkslv", 1, mxL, p)
                y = host.values(yy).copy()
                host.free(p, yy)
                ts += tf + tb
            times.append({"getada1": t1, "getada2": t2, "getada3": t3, "blkchol": t4, "solves": ts})
            host.free(A1, A2, A3, absd, Ld, Lskip, Ladd)
    finally:
        host.free(mxA, mxAjc, lqperm, qblk, mxAord, mxK, mxD, mxDAt, mxud, mxpars, mxADA0, mxL, b)
    return times, y

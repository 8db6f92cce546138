"""oracle/refmex.py -- TEST INFRASTRUCTURE ONLY.

ctypes driver for the reference SeDuMi MEX files compiled (unmodified) into
``oracle/_ref/<name>.so`` by ``oracle/Makefile`` against our MEX-API shim
(``oracle/mexshim``).  It marshals numpy / scipy.sparse / dict values to the
shim's ``mxArray`` and back, so a test can write e.g.

    ref = RefMex()
    LL, Ld, Lskip, Ladd = ref.call("blkchol", 4, L, ADA, pars, absd)

exactly like the MATLAB call ``[L.L,L.d,L.skip,L.add] = blkchol(L,ADA,pars,absd)``
(sedumi.m:458).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product never does.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(_HERE, "_ref")

SHIM_DOUBLE, SHIM_SPARSE, SHIM_STRUCT = 0, 1, 2


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


def available() -> bool:
    return os.path.exists(os.path.join(REF_DIR, "libmexshim.so"))


def find_openblas():
    """The ILP64 OpenBLAS inside numpy's wheel (Fortran entry points scipy_<name>_64_): (path, prefix, suffix) or None."""
    import glob
    for base in (os.path.join(os.path.dirname(np.__file__) + ".libs"), os.path.join(os.path.dirname(os.path.dirname(np.__file__)), "scipy.libs")):
        for f in sorted(glob.glob(os.path.join(base, "libscipy_openblas64_*.so"))):
            return f, "scipy_", "_64_"
    return None


class RawSparse:
    """A sparse matrix to be handed to a reference MEX exactly as stored (no index sorting): incorder's `dz`
    lists the rows of every column in the order in which they were introduced (incorder.c:171-208)."""

    def __init__(self, X):
        self.X = sp.csc_matrix(X)


class RefMexError(RuntimeError):
    pass


class RefMex:
    """Loads the shim and (lazily) the per-MEX shared objects."""

    def __init__(self, ref_dir: str = REF_DIR, mex_dir: str | None = None):
        self.dir = mex_dir or ref_dir          # where <name>.so with a mexFunction are looked up
        path = os.path.join(ref_dir, "libmexshim.so")
        if not os.path.exists(path):
            raise RefMexError(f"{path} missing: run `make -C oracle ref` (needs /root/reference)")
        self.shim = C.CDLL(path, mode=C.RTLD_GLOBAL)
        s = self.shim
        s.mxCreateDoubleMatrix.restype = _MxP
        s.mxCreateDoubleMatrix.argtypes = [C.c_size_t, C.c_size_t, C.c_int]
        s.mxCreateSparse.restype = _MxP
        s.mxCreateSparse.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t, C.c_int]
        s.shim_new_struct.restype = _MxP
        s.mxSetField.argtypes = [_MxP, C.c_size_t, C.c_char_p, _MxP]
        s.mxDestroyArray.argtypes = [_MxP]
        s.shim_call.restype = C.c_int
        s.shim_call.argtypes = [C.c_void_p, C.c_int, C.POINTER(_MxP), C.c_int, C.POINTER(_MxP)]
        s.shim_last_error.restype = C.c_char_p
        self._mex = {}

    # ------------------------------------------------------------ python -> mx
    def to_mx(self, v):
        s = self.shim
        if isinstance(v, dict):
            a = s.shim_new_struct()
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
        if c.kind == SHIM_DOUBLE:
            size = c.m * c.n
            out = np.empty(size, dtype=np.float64)
            if size:
                C.memmove(out.ctypes.data, c.pr, size * 8)
            return out.reshape((c.m, c.n), order="F")
        if c.kind == SHIM_SPARSE:
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
        """Run reference MEX ``name`` with ``nlhs`` outputs; returns a tuple
        (or the single value when nlhs<=1)."""
        fn = self._fn(name)
        nrhs = len(args)
        prhs = (_MxP * max(nrhs, 1))()
        for i, v in enumerate(args):
            prhs[i] = self.to_mx(v)
        nout = max(nlhs, 1)
        plhs = (_MxP * nout)()
        rc = self.shim.shim_call(fn, nlhs, plhs, nrhs, prhs)
        try:
            if rc:
                raise RefMexError(f"{name}: {self.shim.shim_last_error().decode()}")
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
        self.shim.shim_set_global.argtypes = [C.c_char_p, _MxP]
        if value is None:                       # `clear global name`
            self.shim.shim_set_global(name.encode(), None)
            return
        mx = self.to_mx(value)
        try:
            self.shim.shim_set_global(name.encode(), mx)
        finally:
            self.shim.mxDestroyArray(mx)

    def get_global(self, name):
        self.shim.shim_get_global.restype = _MxP
        self.shim.shim_get_global.argtypes = [C.c_char_p]
        mx = self.shim.shim_get_global(name.encode())
        return self.from_mx(mx) if mx else None

    def use_blas(self, lib=None):
        """Bind the BLAS-1 calls of the reference (ddot, daxpy, dscal, dcopy, idamax) to an optimised host BLAS
        (lib = (path, prefix, suffix), see find_openblas) or back to the shim's naive loops (lib = None)."""
        self.shim.shim_use_blas.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p]
        if lib is None:
            return self.shim.shim_use_blas(None, None, None) == 0
        os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
        return self.shim.shim_use_blas(lib[0].encode(), lib[1].encode(), lib[2].encode()) == 0

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
                rc = self.shim.shim_call(fn, nlhs, plhs, nrhs, prhs)
                times.append(time.perf_counter() - t0)
                if rc:
                    raise RefMexError(f"{name}: {self.shim.shim_last_error().decode()}")
                outs = tuple(self.from_mx(plhs[i]) for i in range(nout))
                for i in range(nout):
                    if plhs[i]:
                        self.shim.mxDestroyArray(plhs[i])
        finally:
            for i in range(nrhs):
                self.shim.mxDestroyArray(prhs[i])
        return times, (outs[0] if nlhs <= 1 else outs)

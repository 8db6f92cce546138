"""oracle/refmex.py -- TEST INFRASTRUCTURE ONLY.

Driver for the reference SeDuMi MEX files compiled (unmodified) into
``oracle/_ref/<name>.so`` by ``oracle/Makefile`` against the MEX host of the
package (``sedumi_amd/mexhost``: the mx* API; ``sedumi_amd.mexhost.MexHost``
marshals numpy / scipy.sparse / dict values to ``mxArray`` and back) plus the
BLAS-1 of ``oracle/mexshim``, so a test can write e.g.

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

from sedumi_amd.mexhost import HOST_LIB, MexError, MexHost, RawSparse  # noqa: F401  (the marshalling is the package's)

RefMexError = MexError


def available() -> bool:
    return os.path.exists(os.path.join(REF_DIR, "libmexshim.so")) and os.path.exists(HOST_LIB)


def find_openblas():
    """The ILP64 OpenBLAS inside numpy's wheel (Fortran entry points scipy_<name>_64_): (path, prefix, suffix) or None."""
    import glob
    for base in (os.path.join(os.path.dirname(np.__file__) + ".libs"), os.path.join(os.path.dirname(os.path.dirname(np.__file__)), "scipy.libs")):
        for f in sorted(glob.glob(os.path.join(base, "libscipy_openblas64_*.so"))):
            return f, "scipy_", "_64_"
    return None


class RefMex(MexHost):
    """The compiled reference MEX of ref_dir (or, with mex_dir, any other directory of mexFunction binaries driven through
    the same host -- the tests run the package's shims that way)."""

    def __init__(self, ref_dir: str = REF_DIR, mex_dir: str | None = None):
        path = os.path.join(ref_dir, "libmexshim.so")
        if not os.path.exists(path):
            raise RefMexError(f"{path} missing: run `make -C oracle ref` (needs /root/reference)")
        super().__init__(mex_dir or ref_dir)
        self.blas = C.CDLL(path, mode=C.RTLD_GLOBAL)      # BLAS-1 the reference objects link against

    def use_blas(self, lib=None):
        """Bind the BLAS-1 calls of the reference (ddot, daxpy, dscal, dcopy, idamax) to an optimised host BLAS
        (lib = (path, prefix, suffix), see find_openblas) or back to the shim's naive loops (lib = None)."""
        self.blas.shim_use_blas.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p]
        if lib is None:
            return self.blas.shim_use_blas(None, None, None) == 0
        os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
        return self.blas.shim_use_blas(lib[0].encode(), lib[1].encode(), lib[2].encode()) == 0

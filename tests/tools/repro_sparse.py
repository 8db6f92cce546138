"""Debug helper: sparse random factor on the GPU against the reference, reporting the first supernodes that differ."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import spd_pattern, relerr
from oracle import glue as gl
from oracle.refmex import RefMex, REF_DIR
from sedumi_amd import mex
refmex = RefMex(REF_DIR)
m = 2000; rng = np.random.default_rng(m)
X = spd_pattern("rand", m, rng, 0.004)
L = mex.symbchol(X)
xs = L["xsuper"].ravel().astype(int) - 1
pars = gl.default_pars_chol()
r = refmex.call("blkchol", 4, L, X, pars)
for rep in range(3):
    o = mex.blkchol(L, X, pars)
    dd = np.abs(o[1].ravel() - r[1].ravel()) / np.abs(r[1].ravel())
    bad = np.nonzero(dd > 1e-9)[0]
    Lo, Lr = o[0].tocsc(), r[0].tocsc()
    colerr = np.array([np.abs(Lo.data[Lo.indptr[j]:Lo.indptr[j+1]] - Lr.data[Lr.indptr[j]:Lr.indptr[j+1]]).max() for j in range(m)])
    badc = np.nonzero(colerr > 1e-9)[0]
    sn = np.searchsorted(xs, badc, side="right") - 1
    print("rep", rep, "relerr L", relerr(o[0], r[0]), "bad d", bad[:8], bad.size, "bad L cols", badc[:8], badc.size)
    for s in np.unique(sn)[:6]:
        j = xs[s]; print("   supernode", s, "first col", j, "ns", xs[s+1]-xs[s], "ms", Lr.indptr[j+1]-Lr.indptr[j], "col err", colerr[j])

"""Debug helper: bordered-block factors on the GPU against the reference for a range of sizes."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import relerr
from oracle import glue as gl
from oracle.refmex import RefMex, REF_DIR
from sedumi_amd import mex
from test_emu_parity import _bordered_blocks
refmex = RefMex(REF_DIR)
G = gl.Glue()
for (n1, n2, nc) in [(100, 130, 900), (100, 130, 1500), (100, 130, 2200), (64, 64, 3100), (100, 130, 3100)]:
    rng = np.random.default_rng(9)
    X = _bordered_blocks(n1, n2, nc, rng)
    L = G.symbchol(X)
    xs = L["xsuper"].ravel().astype(int) - 1
    pars = gl.default_pars_chol()
    r = refmex.call("blkchol", 4, L, X, pars)
    o = mex.blkchol(L, X, pars)
    Lo, Lr = o[0].tocsc(), r[0].tocsc()
    m = X.shape[0]
    colerr = np.array([np.abs(Lo.data[Lo.indptr[j]:Lo.indptr[j+1]] - Lr.data[Lr.indptr[j]:Lr.indptr[j+1]]).max() for j in range(m)])
    badc = np.nonzero(colerr > 1e-9)[0]
    print((n1, n2, nc), "xsuper", xs, "relerr", relerr(o[0], r[0]), "first bad cols", badc[:6], badc.size)
    if badc.size:
        j = badc[0]
        rows = Lr.indices[Lr.indptr[j]:Lr.indptr[j+1]]
        e = np.abs(Lo.data[Lo.indptr[j]:Lo.indptr[j+1]] - Lr.data[Lr.indptr[j]:Lr.indptr[j+1]])
        bad_rows = rows[e > 1e-9]
        print("   col", j, "bad rows", bad_rows[:8], "...", bad_rows[-4:], bad_rows.size)

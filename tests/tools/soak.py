"""Randomised soak of the factor / solve path on the GPU against the compiled reference: dense fronts of random order
(well conditioned and rank deficient, several maxu), sparse patterns, bordered blocks.  Time bounded.
    python tests/tools/soak.py [seconds]"""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import scipy.sparse as sp
from helpers import relerr, spd_pattern
from oracle import glue as gl
from oracle.refmex import RefMex, REF_DIR
from sedumi_amd import mex, problem
from helpers import bordered_blocks as _bordered_blocks

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
ref = RefMex(REF_DIR)
G = gl.Glue()
t_end = time.time() + budget
rng = np.random.default_rng(12345)
n_ok = n_bad = 0
case = 0
while time.time() < t_end:
    case += 1
    kind = rng.choice(["dense", "dense_def", "sparse", "border"], p=[0.35, 0.35, 0.2, 0.1])
    pars = dict(gl.default_pars_chol())
    absd = None
    if kind == "dense":
        m = int(rng.integers(65, 900))
        B = rng.standard_normal((m, m)) / np.sqrt(m)
        X = sp.csc_matrix(B @ B.T + 0.5 * np.eye(m)); L = problem.dense_symbolic(m)
    elif kind == "dense_def":
        m = int(rng.integers(65, 700)); r = int(rng.integers(m // 3, m))
        B = rng.standard_normal((m, r))
        X = B @ B.T
        X = sp.csc_matrix(X + np.diag(10.0 ** rng.uniform(-14, -2, m)) * np.abs(X).max()); L = problem.dense_symbolic(m)
        pars["maxu"] = float(rng.choice([5e5, 30.0, 2.0]))
        if rng.random() < 0.5:
            absd = (np.abs(X.diagonal()) * rng.choice([1.0, 1e3, 1e8], m)).reshape(-1, 1)
    elif kind == "sparse":
        m = int(rng.integers(200, 2500)); dens = float(rng.choice([0.002, 0.005, 0.02]))
        X = spd_pattern("rand", m, rng, dens); L = mex.symbchol(X)
    else:
        n1, n2, nc = int(rng.integers(40, 200)), int(rng.integers(40, 200)), int(rng.integers(200, 1500))
        X = _bordered_blocks(n1, n2, nc, rng); L = G.symbchol(X)
    args = (L, X, pars) + ((absd,) if absd is not None else ())
    r = ref.call("blkchol", 4, *args)
    o = mex.blkchol(*args)
    ok = np.array_equal(o[2].indices, r[2].indices) and np.array_equal(o[3].indices, r[3].indices)
    ok = ok and relerr(o[1], r[1]) < 1e-8
    if kind in ("dense", "sparse", "border"):
        ok = ok and relerr(o[0], r[0]) < 1e-10
        L2 = dict(L); L2["L"] = r[0]
        rhs = rng.standard_normal((X.shape[0], 1))
        ok = ok and relerr(mex.fwblkslv(L2, rhs), ref.call("fwblkslv", 1, L2, rhs)) < 1e-10
        ok = ok and relerr(mex.bwblkslv(L2, rhs), ref.call("bwblkslv", 1, L2, rhs)) < 1e-10
    if ok:
        n_ok += 1
    else:
        n_bad += 1
        print("MISMATCH case", case, kind, "m", X.shape[0], "maxu", pars["maxu"], "absd", absd is not None,
              "d err", relerr(o[1], r[1]), "skip", o[2].nnz, r[2].nnz, "add", o[3].nnz, r[3].nnz, flush=True)
print("soak:", n_ok, "ok,", n_bad, "mismatches in", case, "cases")

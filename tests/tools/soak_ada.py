"""Randomised soak of the whole iteration unit (getada1/2/3, blkchol, solves) on the GPU against the compiled
reference on small mixed-cone problems (LP, Lorentz, real and Hermitian PSD blocks, sparse and dense columns).
    python tests/tools/soak_ada.py [seconds]"""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import check_iteration
from oracle import glue as gl
from sedumi_amd import problem

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
G = gl.Glue()
rng = np.random.default_rng(777)
t_end = time.time() + budget
n_ok = n_bad = 0
while time.time() < t_end:
    m = int(rng.integers(8, 140))
    nq = int(rng.integers(0, 30)); ns = int(rng.integers(0, 4)); nh = int(rng.integers(0, 2))
    kw = dict(m=m, lp=int(rng.integers(0, 12)), q=tuple(int(v) for v in rng.integers(2, 7, nq)),
              s=tuple(int(v) for v in rng.integers(2, 40, ns)), hs=tuple(int(v) for v in rng.integers(2, 12, nh)),
              dens=float(rng.choice([0.05, 0.2, 0.6])), block_local=bool(rng.random() < 0.3), seed=int(rng.integers(1 << 30)))
    nrows = kw["lp"] + sum(kw["q"]) + sum(n * (n + 1) // 2 for n in kw["s"]) + sum(n * n for n in kw["hs"])
    if nrows < 1.5 * m:
        continue                                   # A would not have full row rank: ADA' singular, pivot decisions are rounding luck
    try:
        P = problem.random_sdp(**kw)
    except ValueError:
        continue                                   # generator limits (block_local capacity)
    try:
        check_iteration(G, P, seed=int(rng.integers(1000)))
        n_ok += 1
    except AssertionError as e:
        n_bad += 1
        print("MISMATCH", kw, repr(e)[:400], flush=True)
print("soak_ada:", n_ok, "ok,", n_bad, "mismatches")

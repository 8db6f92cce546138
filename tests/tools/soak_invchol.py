"""Randomised soak of invcholfac on the GPU against the compiled reference (random block lists, real + Hermitian)."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import relerr
from oracle.refmex import RefMex, REF_DIR
from sedumi_amd import mex, problem
from test_invcholfac import scaling_factor_case
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
ref = RefMex(REF_DIR); rng = np.random.default_rng(5); t_end = time.time() + budget; ok = bad = 0
while time.time() < t_end:
    s = [int(v) for v in rng.integers(1, 260, int(rng.integers(0, 4)))]
    hs = [int(v) for v in rng.integers(1, 140, int(rng.integers(0, 3)))]
    if not s and not hs:
        continue
    K = problem.make_K(1, [], s, hs=hs)
    u, perm = scaling_factor_case(K, seed=int(rng.integers(1 << 30)))
    pm = perm if rng.random() < 0.7 else None
    args = (u.reshape(-1, 1), K) + ((pm.reshape(-1, 1),) if pm is not None else ())
    e = relerr(mex.invcholfac(u, K, pm), ref.call("invcholfac", 1, *args))
    if e < 1e-10: ok += 1
    else: bad += 1; print("MISMATCH", s, hs, pm is not None, e, flush=True)
print("soak_invchol:", ok, "ok,", bad, "mismatches")

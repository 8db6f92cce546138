"""SURVEY 8f N1 timing: udsqr = invcholfac(u, K, perm) resident on the device (plan buffers "u" -> "udsqr") next to the
compiled reference gateway on one host core.  Run on the GPU box: python tests/tools/time_invcholfac.py"""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from sedumi_amd import problem
from sedumi_amd.plan import Plan
from oracle.refmex import RefMex, REF_DIR
from test_invcholfac import scaling_factor_case

ref = RefMex(REF_DIR)
for name, P in (("control07-shaped (70, 35)", problem.control_like(seed=0)), ("64 x 200", problem.blockdiag_sdp()),
                ("MAXCUT-2000", problem.maxcut(2000)), ("MAXCUT-4000", problem.maxcut(4000))):
    u, perm = scaling_factor_case(P.K, seed=1, garbage_lower=False)
    plan = Plan(0)
    plan.set_chol(problem.dense_symbolic(8), problem.dense_pattern(8)) if False else None
    ADApat = problem.symb_ada(P) if "64" in name else problem.dense_pattern(P.m)
    L = problem.dense_symbolic(P.m) if "64" not in name else None
    if L is None:
        from sedumi_amd import mex
        L = mex.symbchol(ADApat)
    plan.set_chol(L, ADApat); plan.set_ada(P.At, P.Ablkjc, P.K, problem.lorentz_pattern(P))
    plan.upload("u", u)
    for _ in range(3): plan.invcholfac(perm)
    plan.sync(); t0 = time.perf_counter(); reps = 20
    for _ in range(reps): plan.invcholfac(perm)
    plan.sync(); tg = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter(); nrep = 0
    while time.perf_counter() - t0 < 2.0:
        ref.call("invcholfac", 1, u.reshape(-1, 1), P.K, perm.reshape(-1, 1)); nrep += 1
    tc = (time.perf_counter() - t0) / nrep
    n = P.K["s"].ravel()
    flops = float(np.sum(n ** 3) / 3)
    print(f"{name:28s} lenud {u.size:9d}  device {tg*1e6:9.1f} us ({flops/tg/1e9:8.1f} GF/s useful)   reference 1 core {tc*1e3:9.2f} ms   ratio {tc/tg:7.0f}x")
    plan.close()

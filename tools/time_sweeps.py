import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from sedumi_amd import problem
from sedumi_amd.plan import Plan
P = problem.control_like(seed=0)
L, ADA, Q = problem.dense_symbolic(P.m), problem.dense_pattern(P.m), problem.lorentz_pattern(P)
d, ud = problem.spd_scaling(P.K, seed=5)
plan = Plan(0); plan.set_chol(L, ADA); plan.set_ada(P.At, P.Ablkjc, P.K, Q)
plan.upload("dl", d["l"]); plan.upload("ddet", d["det"]); plan.upload("udsqr", ud); plan.upload("rhs", np.ones(P.m))
plan.getada(); plan.blkchol(None, True); plan.sync()
for name, fn in (("fw", plan.fwsolve), ("bw", plan.bwsolve), ("ldl", plan.ldlsolve)):
    for _ in range(5): fn()
    plan.sync(); t0 = time.perf_counter()
    for _ in range(200): fn()
    plan.sync(); print(name, "us/launch (wall, back to back)", (time.perf_counter() - t0) / 200 * 1e6)

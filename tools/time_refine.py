"""tools/time_refine.py [workload] -- the solves of one bench workload with EVERY super-block beyond the growth bound (growth_max = 0):
microseconds per fw + ./d + bw solve by substitution (sdm_plan_set_refinement mode 0) and by inverse + iterative refinement (mode 2),
next to the solve within the bound; relative residual of each against the factored matrix.  One JSON line."""
import json
import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from sedumi_amd.plan import Plan  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "control07"
P, L, ADA, Q, d, ud, rhs, qpr, note = bench.build_workload(name, 0)
out = {"workload": P.name}
for tag, mode, bound in (("within_bound", 1, None), ("substitution", 0, 0.0), ("refined", 2, 0.0)):
    plan = Plan(0)
    plan.set_refinement(mode)
    if bound is not None:
        plan.set_growth_max(bound)
    plan.set_chol(L, ADA)
    plan.set_ada(P.At, P.Ablkjc, P.K, Q)
    plan.upload("dl", d["l"]); plan.upload("ddet", d["det"]); plan.upload("udsqr", ud); plan.upload("rhs", rhs)
    if qpr is not None:
        plan.upload("qpr", qpr)
    plan.getada(); plan.blkchol(bench.PARS, True)
    for _ in range(8):
        plan.ldlsolve()
    plan.sync()
    reps = 40
    plan.timer_begin(2)
    for _ in range(reps):
        plan.ldlsolve()
    plan.timer_end(2)
    us = 1e3 * plan.timer_ms(2) / reps
    plan.kprof(True)
    for _ in range(5):
        plan.ldlsolve()
    prof = plan.kprof_summary(); plan.kprof(False)
    yv = plan.download("y", plan.m)
    A = sp.csc_matrix((plan.download("ada", ADA.nnz), ADA.indices, ADA.indptr), shape=ADA.shape)
    if abs(A - A.T).sum() > 0:
        A = sp.tril(A) + sp.tril(A, -1).T
    nb, nbad, growth = plan.solve_stats()
    out[tag] = {"us_per_solve": round(us, 2), "launches_per_solve": sum(v[0] for k, v in prof.items() if k.startswith("k_sfw") or k.startswith("k_sbw")) / 5.0,
                "relres": float(np.linalg.norm(A @ yv - rhs) / np.linalg.norm(rhs)), "blocks_beyond_bound": nbad, "growth": growth,
                "kernel_us_with_events": {k: round(1e3 * v[1] / v[0], 2) for k, v in prof.items() if k.startswith("k_s")}}
    plan.close()
print(json.dumps(out), flush=True)

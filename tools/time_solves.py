"""tools/time_solves.py <workload> [widths...] -- the solves of one bench workload with every super-block width asked for
(sdm_plan_set_solve_width; 0 = the automatic choice): microseconds per fw + ./d + bw solve (HIP events around 20 x 4
solves), launches per solve, GB/s against the algorithmic bytes of SURVEY.md 8(d), and what the inversion after every
factorisation costs at that width (per-kernel HIP events of one factorisation).  One JSON line per width."""
import json
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
if os.environ.get("SDM_LIB"):                      # a measurement variant (python -m sedumi_amd.build --variant <tag> <flags>)
    from sedumi_amd import capi
    capi.use_library(os.path.join(ROOT, "sedumi_amd", "lib", os.environ["SDM_LIB"]))
import bench  # noqa: E402
from sedumi_amd.plan import Plan  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "control07"
widths = [int(w) for w in sys.argv[2:]] or [0]
P, L, ADA, Q, d, ud, rhs, qpr, note = bench.build_workload(name, 0)
for width in widths:
    plan = Plan(0)
    plan.set_solve_width(width)
    plan.set_chol(L, ADA)
    plan.set_ada(P.At, P.Ablkjc, P.K, Q)
    plan.upload("dl", d["l"]); plan.upload("ddet", d["det"]); plan.upload("udsqr", ud); plan.upload("rhs", rhs)
    if qpr is not None:
        plan.upload("qpr", qpr)
    plan.getada()
    for _ in range(3):
        plan.blkchol(bench.PARS, True)
        for _ in range(4):
            plan.ldlsolve()
    plan.sync()
    reps = 20
    fac = sol = 0.0
    for _ in range(reps):
        plan.timer_begin(1); plan.blkchol(bench.PARS, True); plan.timer_end(1)
        plan.timer_begin(2)
        for _ in range(4):
            plan.ldlsolve()
        plan.timer_end(2)
        fac += plan.timer_ms(1); sol += plan.timer_ms(2)
    plan.kprof(True)
    for _ in range(5):
        plan.blkchol(bench.PARS, True)
        for _ in range(4):
            plan.ldlsolve()
    prof = plan.kprof_summary()
    plan.kprof(False)
    lind = int(np.sum(np.diff(plan.L_pattern.indptr)[(np.asarray(bench.plan_xsuper(plan)) - 1)[:-1]]))
    bytes_solve = 2.0 * (8.0 * plan.nnzL + 8.0 * lind + 16.0 * plan.m)
    us = 1e3 * sol / reps / 4
    nl = sum(v[0] for k, v in prof.items() if k.startswith("k_sfw") or k.startswith("k_sbw")) / 20.0
    prep = {k: round(1e3 * v[1] / 5, 2) for k, v in prof.items() if k in ("k_sprep", "k_sinv128", "k_stile")}
    nb, nbad, growth = plan.solve_stats()
    yv = plan.download("y", plan.m)                                   # (residual of the last solve against the matrix that was factored)
    import scipy.sparse as sp
    A = sp.csc_matrix((plan.download("ada", ADA.nnz), ADA.indices, ADA.indptr), shape=ADA.shape)
    if abs(A - A.T).sum() > 0:
        A = sp.tril(A) + sp.tril(A, -1).T
    resid = float(np.linalg.norm(A @ yv - rhs) / np.linalg.norm(rhs))
    print(json.dumps({"workload": P.name, "m": plan.m, "width": plan.solve_width(), "us_per_solve": round(us, 2), "launches_per_solve": nl,
                      "GBs": round(bytes_solve / us / 1e3, 1), "frac_of_hbm_peak": round(bytes_solve / us / 1e3 / 8000.0, 4),
                      "factor_incl_inversion_ms": round(fac / reps, 4), "inversion_us_by_kernel_with_events": prep,
                      "kernel_us_with_events": {k: round(1e3 * v[1] / v[0], 2) for k, v in prof.items() if k.startswith("k_s") or k.startswith("k_ldl")},
                      "super_blocks": nb, "bad": nbad, "growth": growth, "relres": resid,
                      "lib": os.environ.get("SDM_LIB", "")}), flush=True)
    plan.close()

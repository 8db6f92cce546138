"""tools/launch_cost.py -- host time per enqueued solve (control07): the lean form (2 launches) and the refinement form (10 launches)."""
import sys, time
import os; sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..')))
import bench
from sedumi_amd.plan import Plan
P, L, ADA, Q, d, ud, rhs, qpr, note = bench.build_workload("control07", 0)
for mode, bound in ((1, None), (2, 0.0)):
    plan = Plan(0)
    plan.set_refinement(mode)
    if bound is not None: plan.set_growth_max(bound)
    plan.set_chol(L, ADA); plan.set_ada(P.At, P.Ablkjc, P.K, Q)
    plan.upload("dl", d["l"]); plan.upload("ddet", d["det"]); plan.upload("udsqr", ud); plan.upload("rhs", rhs)
    plan.getada(); plan.blkchol(bench.PARS, True)
    for _ in range(5): plan.ldlsolve()
    plan.sync()
    n = 100
    t0 = time.perf_counter()
    for _ in range(n): plan.ldlsolve()
    t1 = time.perf_counter()
    plan.sync()
    t2 = time.perf_counter()
    print("mode", mode, "enqueue us per solve %.1f" % (1e6 * (t1 - t0) / n), "incl. drain %.1f" % (1e6 * (t2 - t0) / n), flush=True)
    plan.close()

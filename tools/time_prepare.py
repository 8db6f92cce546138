"""tools/time_prepare.py [workload] -- device time of back-to-back factorisations (sdm_plan_blkchol: assembly, pivot bounds, the
LDL', the solve preparation) and the per-kernel times of everything in it but the LDL' launches."""
import os
import sys


ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "control07"
reps = 200
P, L, ADA, Q, d, ud, rhs, qpr, note = bench.build_workload(name, 1)
plan = bench.make_plan(0, P, L, ADA, Q, d, ud, rhs, qpr)
plan.getada()
for _ in range(20):
    plan.blkchol(bench.PARS, True)
plan.sync()
plan.timer_begin(0)
for _ in range(reps):
    plan.blkchol(bench.PARS, True)
plan.timer_end(0)
plan.kprof(True)
for _ in range(20):
    plan.blkchol(bench.PARS, True)
prof = plan.kprof_summary()
plan.kprof(False)
print(name, "blkchol ms:", plan.timer_ms(0) / reps,
      {k: round(v[1] / 20, 4) for k, v in prof.items() if not k.startswith("k_ldl")})

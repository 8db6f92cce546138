"""tools/time_prepare.py [workload] -- time of the solve preparation (inverses of the diagonal super-blocks + premultiplication,
sdm_solve.hip::solve_prepare) alone: sdm_plan_load_factor re-runs it on the resident factor (upload of L excluded by
differencing against an upload-only timing is not attempted: the figure printed is the device time between two events
around `reps` back-to-back blkchol calls minus the same with SDM_SPREP_OFF -- run the tool twice).  Prints ms per blkchol."""
import os
import sys

import numpy as np

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
print(name, "SDM_SPREP_OFF" if os.environ.get("SDM_SPREP_OFF") else "fused", "blkchol ms:", plan.timer_ms(0) / reps,
      {k: round(v[1] / 20, 4) for k, v in prof.items() if not k.startswith("k_ldl")})

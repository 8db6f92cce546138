"""tools/time_factor.py <workload> [library] -- the factorisation alone (sdm_plan_blkchol incl. the inverses for the solves), HIP-event timed, and the
panel launches' share of it (kernel profile with events); `library`: a measurement build (python -m sedumi_amd.build --variant <tag> <flags>)."""
import json
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from sedumi_amd import capi  # noqa: E402
if len(sys.argv) > 2:
    capi.use_library(os.path.abspath(sys.argv[2]))
import bench  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "maxcut4000"
P, L, ADA, Q, d, ud, rhs, qpr, note = bench.build_workload(name, 0)
plan = bench.make_plan(0, P, L, ADA, Q, d, ud, rhs, qpr)
plan.getada()
for _ in range(3):
    plan.blkchol(bench.PARS, True)
plan.sync()
tot, n = 0.0, 10
for _ in range(n):
    plan.timer_begin(1); plan.blkchol(bench.PARS, True); plan.timer_end(1)
    tot += plan.timer_ms(1)
plan.kprof(True)
for _ in range(5):
    plan.blkchol(bench.PARS, True)
prof = plan.kprof_summary()
plan.kprof(False)
plan.ldlsolve()
y = plan.download("y")
print(json.dumps({"workload": name, "lib": os.path.basename(sys.argv[2]) if len(sys.argv) > 2 else "", "factor_ms": tot / n,
                  "kernel_ms_per_factor_with_events": {k: round(v[1] / 5, 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])},
                  "y_norm": float((y * y).sum() ** 0.5)}))

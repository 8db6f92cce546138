"""tools/trace_units.py [workload] -- 60 bench units of a workload (default control07) and nothing else (for rocprofv3 --kernel-trace: tools/unit_timeline.py)."""
import os
import sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import bench  # noqa: E402
P, L, ADA, Q, d, ud, rhs, qpr, note = bench.build_workload(sys.argv[1] if len(sys.argv) > 1 else "control07", 0)
plan = bench.make_plan(0, P, L, ADA, Q, d, ud, rhs, qpr)
step = bench.unit_fn(plan)
for _ in range(60):
    step()
plan.sync()

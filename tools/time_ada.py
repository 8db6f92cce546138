"""tools/time_ada.py [workload] -- ADA' of one bench workload: per-kernel microseconds (HIP events around every launch, 20 calls) and the
phase time.  SDM_LIB=<file in sedumi_amd/lib> selects a measurement build (python -m sedumi_amd.build --variant <tag> <flags>)."""
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
if os.environ.get("SDM_LIB"):
    from sedumi_amd import capi
    capi.use_library(os.path.join(ROOT, "sedumi_amd", "lib", os.environ["SDM_LIB"]))
import bench  # noqa: E402
from sedumi_amd.plan import Plan  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "control07"
P, L, ADA, Q, d, ud, rhs, qpr, note = bench.build_workload(name, 0)
plan = bench.make_plan(0, P, L, ADA, Q, d, ud, rhs, qpr)
for _ in range(5):
    plan.getada()
plan.sync()
tot = 0.0
for _ in range(20):
    plan.timer_begin(0); plan.getada(); plan.timer_end(0)
    tot += plan.timer_ms(0)
plan.kprof(True)
for _ in range(20):
    plan.getada()
prof = plan.kprof_summary()
plan.kprof(False)
print(json.dumps({"workload": name, "lib": os.environ.get("SDM_LIB", ""), "ada_us": round(1e3 * tot / 20, 2),
                  "kernel_us_with_events": {k: round(1e3 * v[1] / v[0], 2) for k, v in prof.items()}}), flush=True)

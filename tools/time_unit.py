"""tools/time_unit.py <workload> [steps] -- ms per unit, its three phases and the per-kernel HIP-event times of one bench workload (the
roofline leg of bench.py without the rest of the line); SDM_LIB=<variant .so> measures a variant build.  One JSON line."""
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
if os.environ.get("SDM_LIB"):
    from sedumi_amd import capi
    capi.use_library(os.path.join(ROOT, "sedumi_amd", "lib", os.environ["SDM_LIB"]))
import bench  # noqa: E402
import numpy as np  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "control07"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
P, L, ADA, Q, d, ud, rhs, qpr, note = bench.build_workload(name, 0)
plan = bench.make_plan(0, P, L, ADA, Q, d, ud, rhs, qpr)
plan._xsuper = np.asarray(L["xsuper"]).ravel().astype(np.int64)
el = bench.time_steps(plan, bench.unit_fn(plan), steps, 5)
roof, phases = bench.profile_unit(plan, P, ud, min(steps, 20))
print(json.dumps({"workload": name, "lib": os.environ.get("SDM_LIB", "default"), "ms_per_step": 1e3 * el / steps,
                  "phases_ms": {k: phases[k] for k in ("ada_ms", "factor_ms", "solves_ms")},
                  "kernels_ms_per_step": roof and {k: round(v, 5) for k, v in roof["stage_ms_per_step"].items()}}))

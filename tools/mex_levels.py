"""tools/mex_levels.py <workload> [units] -- bench.py's mex_inclusive leg (the iteration unit through the built mexFunction shims) at every
level of lazy intermediates (sdm_mexcache_set_lazy 0 / 1 / 2): ms per unit, stage times, host words checksummed.  One JSON line per level."""
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "control07"
units = int(sys.argv[2]) if len(sys.argv) > 2 else 20
P, L, ADA, Q, d, ud, rhs, qpr, note = bench.build_workload(name, 0)
for lazy in (0, 1, 2):
    r = bench.mex_inclusive(P, L, ADA, Q, d, ud, rhs, qpr, units, None, lazy=lazy)
    print(json.dumps({"workload": name, "lazy": lazy, **{k: r.get(k) for k in ("ms_per_step", "first_unit_ms", "stage_ms_per_unit", "content_checks", "cache_counters", "error")}}), flush=True)

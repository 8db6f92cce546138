"""tools/factor_hash.py [workload...] -- sha256 of the factor (L values, d) and of one solve of bench workloads on the resident plan: a change that
claims "the same operations in the same order" must leave these unchanged."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

for name in sys.argv[1:] or ["control07", "arch0", "nb", "maxcut2000"]:
    P, L, ADA, Q, d, ud, rhs, qpr, note = bench.build_workload(name, 0)
    plan = bench.make_plan(0, P, L, ADA, Q, d, ud, rhs, qpr)
    plan.getada(); plan.blkchol(bench.PARS, True); plan.ldlsolve()
    h = {k: hashlib.sha256(np.ascontiguousarray(plan.download(k)).tobytes()).hexdigest()[:16] for k in ("ada", "lpr", "d", "y")}
    print(json.dumps({"workload": name, **h}), flush=True)
    plan.close()

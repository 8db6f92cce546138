"""tools/merge_probe.py <m> [width] -- the merged sweep launches (SEDUMI_HIP_SWEEP_MERGE = 0 separate, 1 the default, 2 every row / step launch of a
one-front level) on one dense front of order m: microseconds per solve by merge level and whether the results match the separate launches bit for
bit over `reps` solves each."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
if os.environ.get("SDM_LIB"):                                     # a variant build (python -m sedumi_amd.build --variant <tag> ...)
    from sedumi_amd import capi
    capi.use_library(os.path.join(ROOT, "sedumi_amd", "lib", os.environ["SDM_LIB"]))
from sedumi_amd import problem  # noqa: E402
from sedumi_amd.plan import Plan  # noqa: E402

m = int(sys.argv[1])
width = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(m)
X = rng.standard_normal((m, m)); X = 0.5 * (X + X.T) / np.sqrt(m); X[np.diag_indices(m)] = 4.0 + rng.random(m)
plan = Plan(0)
plan.set_solve_width(width)
plan.set_chol(problem.dense_symbolic(m), problem.dense_pattern(m))
plan.upload("ada", X.ravel(order="F")); del X
plan.upload("rhs", rng.standard_normal(m))
plan.blkchol(None, False); plan.sync()


rhss = [rng.standard_normal(m) for _ in range(3)]       # in turn: what a solve leaves in the work vectors is not the next one's data


def run(level, reps=21):
    os.environ["SEDUMI_HIP_SWEEP_MERGE"] = str(level)
    ys = []
    for it in range(reps):
        plan.upload("rhs", rhss[it % 3]); plan.upload("y", np.zeros(m)); plan.ldlsolve(); ys.append(plan.download("y"))
    plan.sync()
    t0 = time.perf_counter()
    for _ in range(50):
        plan.ldlsolve()
    plan.sync()
    return 1e6 * (time.perf_counter() - t0) / 50, ys


t0, ref = run(0)
print(json.dumps({"m": m, "width": width or "auto", "level": 0, "us": round(t0, 1)}), flush=True)
for level in (1, 2):
    t, ys = run(level)
    bad = sum(not np.array_equal(y, ref[it % 3]) for it, y in enumerate(ys))
    err = max(float(np.max(np.abs(y - ref[it % 3]))) for it, y in enumerate(ys)) / float(np.max(np.abs(ref[0])))
    print(json.dumps({"m": m, "level": level, "us": round(t, 1), "solves_differing": bad, "of": len(ys), "max_rel_diff": err}), flush=True)
plan.close()

"""tools/merge_soak.py [seconds=120] [seed=0] -- soak of the merged sweep launches: dense fronts of random order (600 ... 9000) and super-block width
(512 / 1024 / 2048), blocks beyond the growth bound in some (growth_max 0 / 1e-3: the substitution fallback inside the merged launch), three right-hand
sides in turn, every solve at merge level 1 and 2 compared with the separate launches (bit for bit where every block is within the bound, to 1e-9
otherwise) and with X y = b.  One JSON line at the end."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from sedumi_amd import problem  # noqa: E402
from sedumi_amd.plan import Plan  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end = time.time() + budget
ncase = nsolve = nmerged = 0
worst = 0.0
while time.time() < t_end:
    width = int(rng.choice([512, 1024, 2048]))
    m = int(rng.integers(width + 40, min(9000, 6 * width)))
    thr = rng.choice([None, None, 0.0, 1e-3])
    X = rng.standard_normal((m, m)); X = 0.5 * (X + X.T) / np.sqrt(m); X[np.diag_indices(m)] = 4.0 + rng.random(m)
    plan = Plan(0)
    plan.set_solve_width(width)
    plan.set_refinement(0)                                             # (substitute blocks beyond the bound: the sweeps keep merging)
    plan.set_chol(problem.dense_symbolic(m), problem.dense_pattern(m))
    if thr is not None:
        plan.set_growth_max(float(thr))
    plan.upload("ada", X.ravel(order="F"))
    plan.blkchol(None, False)
    _, nbad, _ = plan.solve_stats()
    rhss = [rng.standard_normal(m) for _ in range(3)]
    ref = []
    os.environ["SEDUMI_HIP_SWEEP_MERGE"] = "0"
    for r in rhss:
        plan.upload("rhs", r); plan.ldlsolve(); ref.append(plan.download("y"))
        res = float(np.max(np.abs(X @ ref[-1] - r)) / np.max(np.abs(r)))
        assert res < 1e-9, ("separate launches", m, width, thr, res)
    for level in (1, 2):
        os.environ["SEDUMI_HIP_SWEEP_MERGE"] = str(level)
        plan.kprof(True)
        for it in range(9):
            plan.upload("rhs", rhss[it % 3]); plan.upload("y", np.zeros(m)); plan.ldlsolve()
            y = plan.download("y")
            nsolve += 1
            if nbad == 0:
                assert np.array_equal(y, ref[it % 3]), ("bits", m, width, thr, level, it)
            else:
                e = float(np.max(np.abs(y - ref[it % 3])) / np.max(np.abs(ref[it % 3])))
                worst = max(worst, e)
                assert e < 1e-9, ("fallback", m, width, thr, level, it, e)
        prof = plan.kprof_summary(); plan.kprof(False)
        nmerged += sum(v[0] for k, v in prof.items() if "rows_diag" in k or "step_diag" in k)
    plan.close()
    ncase += 1
os.environ.pop("SEDUMI_HIP_SWEEP_MERGE", None)
print(json.dumps({"cases": ncase, "merged_solves_checked": nsolve, "merged_launches": nmerged, "worst_rel_diff_with_blocks_beyond_the_bound": worst, "seconds": budget}))

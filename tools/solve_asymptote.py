"""tools/solve_asymptote.py <m>... -- the fw + ./d + bw solve of ONE dense front of order m (a diagonally dominant random matrix, factored by the
library): microseconds per solve and the fraction of the 8 TB/s HBM peak on the algorithmic bytes of SURVEY.md 8(d), for orders beyond what the
ADA' stage admits (one PSD block of order > 8700 does not fit its LDS staging): where the sweeps go as the launches grow."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from sedumi_amd import problem  # noqa: E402
from sedumi_amd.plan import Plan  # noqa: E402

for m in (int(a) for a in sys.argv[1:]):
    rng = np.random.default_rng(m)
    X = rng.standard_normal((m, m)).astype(np.float64)
    X = 0.5 * (X + X.T) / np.sqrt(m)
    X[np.diag_indices(m)] = 4.0 + rng.random(m)
    plan = Plan(0)
    plan.set_chol(problem.dense_symbolic(m), problem.dense_pattern(m))
    plan.upload("ada", X.ravel(order="F")); del X
    plan.upload("rhs", rng.standard_normal(m))
    t0 = time.perf_counter()
    plan.blkchol(None, False); plan.sync()
    t_fac = time.perf_counter() - t0
    res = {}
    for merge in (0, 1, 2):                                            # SEDUMI_HIP_SWEEP_MERGE: separate launches / merged where rows stream on / every row launch
        os.environ["SEDUMI_HIP_SWEEP_MERGE"] = str(merge)
        for _ in range(5):
            plan.ldlsolve()
        plan.sync()
        n = 50
        t0 = time.perf_counter()
        for _ in range(n):
            plan.ldlsolve()
        plan.sync()
        t = (time.perf_counter() - t0) / n
        plan.kprof(True); plan.ldlsolve(); prof = plan.kprof_summary(); plan.kprof(False)
        res[merge] = (t, sum(v[0] for v in prof.values()), plan.download("y"))
    same = all(np.array_equal(res[0][2], res[k][2]) for k in (1, 2))
    t, nl = res[1][0], res[1][1]
    nnzL = m * (m + 1) // 2
    nbytes = 2.0 * (8.0 * nnzL + 8.0 * m + 16.0 * m)
    nb, nbad, growth = plan.solve_stats()
    print(json.dumps({"m": m, "us_per_solve": 1e6 * t, "algorithmic_bytes_per_solve": nbytes, "achieved_GBs": nbytes / t / 1e9, "frac_of_hbm_peak": nbytes / t / 8e12,
                      "launches_per_solve": nl, "us_by_merge_level": {k: round(1e6 * v[0], 1) for k, v in res.items()},
                      "launches_by_merge_level": {k: v[1] for k, v in res.items()}, "merged_results_bit_identical": bool(same), "super_blocks": nb, "blocks_beyond_growth_bound": nbad, "first_factorisation_s": t_fac}), flush=True)
    plan.close()

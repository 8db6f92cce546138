"""tools/l_error_probe.py [seconds] [mmin mmax] -- rank-deficient dense fronts (helpers.rank_deficient_front_case): relative Frobenius error of
L.L and L.d against the compiled reference, next to the size of the largest multiplier: how far the blocked matrix-core row solve (a
different summation order than blkchol2.c's) departs from the reference when pivots are skipped and multipliers are large."""
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import rank_deficient_front_case, relerr  # noqa: E402
from oracle.refmex import RefMex, REF_DIR  # noqa: E402
from sedumi_amd import mex  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
mmin, mmax = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (100, 340)
ref = RefMex(REF_DIR)
rng = np.random.default_rng(99)
t_end = time.time() + budget
errs = []
while time.time() < t_end:
    args = rank_deficient_front_case(rng, mmin, mmax)
    rr = ref.call("blkchol", 4, *args)
    o = mex.blkchol(*args)
    same = np.array_equal(o[2].indices, rr[2].indices) and np.array_equal(o[3].indices, rr[3].indices)
    errs.append((relerr(sp.csc_matrix(o[0]).data, sp.csc_matrix(rr[0]).data), relerr(o[1], rr[1]), float(np.abs(sp.csc_matrix(rr[0]).data).max()), rr[2].nnz, same))
e = np.array([(a, b, c, d) for a, b, c, d, _ in errs])
print(json.dumps({"fronts": len(errs), "rows": [mmin, mmax], "decisions_equal": int(sum(x[4] for x in errs)),
                  "L_err_max": float(e[:, 0].max()), "L_err_median": float(np.median(e[:, 0])), "L_err_above_1e-10": int((e[:, 0] > 1e-10).sum()),
                  "d_err_max": float(e[:, 1].max()), "largest_multiplier_max": float(e[:, 2].max()),
                  "L_err_above_1e-10_cases": [{"L_err": float(a), "d_err": float(b), "max_abs_L": float(c), "skipped": int(d)} for a, b, c, d in e if a > 1e-10][:8]}), flush=True)

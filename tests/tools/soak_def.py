"""Rank-deficient dense fronts only (the never-fail rule's probe / added-diagonal path), against the compiled reference.
    python tests/tools/soak_def.py [seconds] [mmin mmax] [panel]        panel: the launch-per-panel path instead of k_ldl_front"""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import scipy.sparse as sp
from helpers import relerr, rank_deficient_front_case
from oracle import glue as gl
from oracle.refmex import RefMex, REF_DIR
from sedumi_amd import mex, problem
from sedumi_amd.plan import Plan

panel_path = "panel" in sys.argv
if panel_path:
    sys.argv.remove("panel")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
mmin, mmax = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (320, 700)
ref = RefMex(REF_DIR)
t_end = time.time() + budget
rng = np.random.default_rng(777)
n_ok = n_bad = case = 0
while time.time() < t_end:
    case += 1
    args = rank_deficient_front_case(rng, mmin, mmax)
    m, pars, absd = args[1].shape[0], args[2], (args[3] if len(args) > 3 else None)
    rr = ref.call("blkchol", 4, *args)
    if panel_path:                                                    # the launch-per-panel path: through the resident plan
        plan = Plan(0)
        plan.set_one_launch_fronts(False)
        plan.set_chol(args[0], args[1])
        plan.upload("ada", sp.csc_matrix(args[1]).data)
        if absd is not None:
            plan.upload("absd", absd.ravel())
        plan.blkchol(pars, absd is not None)
        (si, _), (ai, _) = plan.pivots()
        o = (None, plan.download("d").reshape(-1, 1), sp.csc_matrix((np.ones(si.size), si, [0, si.size]), shape=(m, 1)), sp.csc_matrix((np.ones(ai.size), ai, [0, ai.size]), shape=(m, 1)))
        plan.close()
    else:
        o = mex.blkchol(*args)
    ok = np.array_equal(o[2].indices, rr[2].indices) and np.array_equal(o[3].indices, rr[3].indices) and relerr(o[1], rr[1]) < 1e-8
    if ok:
        n_ok += 1
    else:
        n_bad += 1
        first = next((int(a) for a in np.sort(np.setxor1d(o[3].indices, rr[3].indices))), -1)
        print("MISMATCH case", case, "m", m, "maxu", pars["maxu"], "absd", absd is not None, "d err", relerr(o[1], rr[1]), "skip", o[2].nnz, rr[2].nnz,
              "add", o[3].nnz, rr[3].nnz, "first differing add index", first, flush=True)
print("soak_def:", n_ok, "ok,", n_bad, "mismatches in", case, "cases", "(launch-per-panel path)" if panel_path else "")

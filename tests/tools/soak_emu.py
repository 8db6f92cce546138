"""The soak of rank-deficient fronts (tests/tools/soak_def.py) on the CPU: the emulator with one process per workgroup
(tests/hipemu: emu_launch_concurrent), against the compiled reference.  Not part of the suites.
    python tests/tools/soak_emu.py [seconds] [mmin mmax] [panel]"""
import ctypes, os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import scipy.sparse as sp
import helpers
from helpers import relerr, rank_deficient_front_case
helpers.use_emu()
from hipemu import build_emu
lib = ctypes.CDLL(build_emu.build())
lib._Z18emu_set_concurrenti(1)
from oracle.refmex import RefMex, REF_DIR
from sedumi_amd.plan import Plan

panel_path = "panel" in sys.argv
if panel_path:
    sys.argv.remove("panel")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
mmin, mmax = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (320, 700)
ref = RefMex(REF_DIR)
t_end = time.time() + budget
rng = np.random.default_rng(4242)
n_ok = n_bad = case = 0
while time.time() < t_end:
    case += 1
    args = rank_deficient_front_case(rng, mmin, mmax)
    m, pars, absd = args[1].shape[0], args[2], (args[3] if len(args) > 3 else None)
    rr = ref.call("blkchol", 4, *args)
    plan = Plan(0)
    plan.set_one_launch_fronts(not panel_path)
    plan.set_chol(args[0], args[1])
    plan.upload("ada", sp.csc_matrix(args[1]).data); plan.upload("rhs", np.ones(m))
    if absd is not None:
        plan.upload("absd", absd.ravel())
    plan.blkchol(pars, absd is not None)
    plan.ldlsolve()                                                   # (the follower's inverse / k_sprep are part of what runs concurrently)
    (si, _), (ai, _) = plan.pivots()
    d = plan.download("d")
    ok = np.array_equal(si, rr[2].indices) and np.array_equal(ai, rr[3].indices) and relerr(d, rr[1].ravel()) < 1e-8
    n_ok += ok; n_bad += not ok
    if not ok:
        print("MISMATCH case", case, "m", m, "maxu", pars["maxu"], "absd", absd is not None, "d err", relerr(d, rr[1].ravel()), "skip", si.size, rr[2].nnz,
              "add", ai.size, rr[3].nnz, flush=True)
print("soak_emu:", n_ok, "ok,", n_bad, "mismatches in", case, "cases", "(launch-per-panel path)" if panel_path else "(one-launch fronts)")

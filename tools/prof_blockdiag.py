"""Per-kernel time of one iteration unit on the block-diagonal config (BASELINE.json configs[4]) on one GPU."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from sedumi_amd import mex, problem
from sedumi_amd.plan import Plan
nblk, n, mper = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (64, 200, 150)))
P = problem.blockdiag_sdp(nblk=nblk, n=n, mper=mper, nnz=20, seed=4)
d, ud = problem.spd_scaling(P.K, seed=5)
ADApat = problem.symb_ada(P); L = mex.symbchol(ADApat)
xs = np.asarray(L["xsuper"]).ravel(); print("m", P.m, "nsuper", xs.size - 1, "max ns", int(np.diff(xs).max()), "nnzL", L["L"].nnz)
pl = Plan(0); pl.set_chol(L, ADApat); pl.set_ada(P.At, P.Ablkjc, P.K, problem.lorentz_pattern(P))
pl.upload("dl", d["l"]); pl.upload("ddet", d["det"]); pl.upload("udsqr", ud); pl.upload("rhs", np.ones(P.m))
def step():
    pl.getada(); pl.blkchol(None, True)
    for _ in range(4): pl.ldlsolve()
for _ in range(3): step()
pl.sync(); t0 = time.perf_counter()
for _ in range(10): step()
pl.sync(); print("ms/unit", (time.perf_counter() - t0) * 100)
pl.kprof(True)
for _ in range(5): step()
prof = pl.kprof_summary(); pl.kprof(False)
for k, (c, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1]): print(f"  {k:24s} calls/unit {c/5:7.1f}  ms/unit {ms/5:.3f}  us/launch {1e3*ms/c:.1f}")

"""First-light / stage-timing script for the GPU box (also runs against the emulator with --emu).
Compares every stage of one IPM iteration unit with the compiled reference and prints timings."""
import argparse
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from sedumi_amd import capi, mex, problem  # noqa: E402
from sedumi_amd.plan import Plan  # noqa: E402


def relerr(a, b):
    a = np.asarray(a.todense() if sp.issparse(a) else a)
    b = np.asarray(b.todense() if sp.issparse(b) else b)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--emu", action="store_true")
    ap.add_argument("--workload", default="control")
    ap.add_argument("--n", type=int, default=0)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--noref", action="store_true")
    args = ap.parse_args()
    if args.emu:
        capi.use_library(os.path.join(ROOT, "tests", "hipemu", "libsedumi_hipemu.so"))
    print("backend", capi.backend(), "devices", capi.device_count(), flush=True)
    if args.workload == "control":
        P = problem.control_like()
    elif args.workload == "maxcut":
        P = problem.maxcut(args.n or 1000)
    elif args.workload == "blockdiag":
        P = problem.blockdiag_sdp(nblk=args.n or 8, n=40, mper=30, nnz=8)
    else:
        P = problem.random_sdp(m=40, seed=3)
    print(P.name, "At", P.At.shape, P.At.nnz, flush=True)
    d, ud = problem.spd_scaling(P.K, seed=5)
    m = P.m
    rhs = np.random.default_rng(0).standard_normal(m)
    pars = {"canceltol": 1e-12, "maxu": 5e5, "abstol": 1e-20}
    have_ref = False
    if not args.noref:
        from oracle import glue as gl, refmex
        if refmex.available():
            have_ref = True
            G = gl.Glue()
            t = time.time(); S = G.setup(P.At, P.K); print("ref setup %.2fs  nsuper %d nnzL %d" % (time.time() - t, S["L"]["xsuper"].size - 1, S["L"]["L"].nnz))
            assert np.array_equal(S["Ablkjc"], P.Ablkjc), "partitA mismatch"
            dd = {"l": d["l"], "det": d["det"], "q1": np.ones(P.K["q"].size), "q2": np.zeros(int(P.K["mainblks"].ravel()[2] - P.K["mainblks"].ravel()[1]))}
            t = time.time(); it = G.iteration_ref(S, dd, ud, dict(pars)); tref_it = time.time() - t
            t = time.time(); yref = G.solve_ref(S, it, rhs); tref_solve = time.time() - t
            print("ref getada+blkchol %.4fs, solve %.5fs" % (tref_it, tref_solve), flush=True)
            L, ADApat, Qpat = S["L"], S["ADA"], S["DAt"]["q"]
    if not have_ref:
        L, ADApat, Qpat = problem.dense_symbolic(m), problem.dense_pattern(m), problem.lorentz_pattern(P)
    # ---- tier-1 parity
    if have_ref:
        t = time.time()
        A3, absd = mex.getada3(it["ADA2"], S["A"], S["Ablkjc"][:, 2], S["Aord"], ud, P.K)
        print("tier1 getada3: ADA %.2e absd %.2e  (%.3fs)" % (relerr(A3, it["ADA"]), relerr(absd, it["absd"]), time.time() - t))
        t = time.time()
        LL, Ld, Lskip, Ladd = mex.blkchol(L, it["ADA"], pars, it["absd"])
        print("tier1 blkchol: L %.2e d %.2e skip %d/%d add %d/%d (%.3fs)" % (relerr(LL, it["LL"]), relerr(Ld, it["Ld"]), Lskip.nnz, it["Lskip"].nnz, Ladd.nnz, it["Ladd"].nnz, time.time() - t))
        L2 = dict(L); L2["L"] = it["LL"]
        yf = mex.fwblkslv(L2, rhs); yfr = G.ref.call("fwblkslv", 1, L2, rhs.reshape(-1, 1))
        yb = mex.bwblkslv(L2, rhs); ybr = G.ref.call("bwblkslv", 1, L2, rhs.reshape(-1, 1))
        print("tier1 fw %.2e bw %.2e" % (relerr(yf, yfr), relerr(yb, ybr)), flush=True)
    # ---- resident plan
    plan = Plan(0)
    t = time.time(); plan.set_chol(L, ADApat); plan.set_ada(P.At, P.Ablkjc, P.K, Qpat); print("plan setup %.3fs" % (time.time() - t))
    plan.upload("dl", d["l"]); plan.upload("ddet", d["det"]); plan.upload("udsqr", ud); plan.upload("rhs", rhs)
    if have_ref and Qpat.nnz:
        Qc = sp.csc_matrix(Qpat); Qn = sp.csc_matrix(it["DAt"]["q"])
        rows = Qc.indices; cols = np.repeat(np.arange(m), np.diff(Qc.indptr))
        plan.upload("qpr", np.asarray(Qn[rows, cols]).ravel())
    names = ["getada", "blkchol", "ldlsolve x4", "total"]
    best = [1e30] * 4
    for rep in range(args.reps):
        plan.timer_begin(3)
        plan.timer_begin(0); plan.getada(); plan.timer_end(0)
        plan.timer_begin(1); plan.blkchol(pars, True); plan.timer_end(1)
        plan.timer_begin(2)
        for _ in range(4):
            plan.ldlsolve()
        plan.timer_end(2); plan.timer_end(3)
        plan.sync()
        for i in range(4):
            best[i] = min(best[i], plan.timer_ms(i))
    for i in range(4):
        print("  %-12s %9.3f ms" % (names[i], best[i]))
    print("  iters/s %.1f" % (1000.0 / best[3]), flush=True)
    if have_ref:
        ada = plan.download("ada"); absd = plan.download("absd"); dfac = plan.download("d"); y = plan.download("y")
        lpr = plan.download("lpr")
        print("plan: ADA %.2e absd %.2e d %.2e L %.2e y %.2e" % (relerr(ada, it["ADA"].data), relerr(absd, it["absd"].ravel()),
              relerr(dfac, it["Ld"].ravel()), relerr(lpr, it["LL"].data), relerr(y, yref.ravel())))
        print("speedup vs ref (getada+blkchol+4 solves): %.1fx" % ((tref_it + 4 * tref_solve) * 1000.0 / best[3]))


if __name__ == "__main__":
    main()

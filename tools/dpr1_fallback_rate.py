"""tools/dpr1_fallback_rate.py [m n ndense] -- how often does dpr1fact leave the device?  (Since round 4: never -- k_dpr1_general is the
whole of dodpr1fact and *host_fallback is always 0; the tool now counts how many iterations of a real run NEED the general case, i.e.
postponed pivots or dependent rows, by the factors' dopiv flags.)

Until round 3 sdm_plan_deninfac factored the dense columns with scan kernels when every pivot was accepted in the first round
and handed the case to a host algorithm otherwise (postponed pivots, dependent rows: dpr1fact.c:168-202, 224-240, 371-476).  This tool
drives a whole interior-point solve of a sparse LP with dense columns -- a plain infeasible primal-dual path-following method in numpy
(Mehrotra predictor-corrector, normal equations; the iterates only serve as REAL scalings d = x ./ z for the hot path) -- and per
iteration runs getada / blkchol / deninfac on the resident plan exactly as sedumi.m:449-462 does (dense rows removed from At, smult =
d of the dense variables), recording whether the host algorithm was needed.  One JSON line."""
import json
import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from sedumi_amd import mex, problem  # noqa: E402

m, n, nd = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (400, 4000, 6)
rng = np.random.default_rng(7)
P0 = problem.lp_dense_cols(m=m, n=n, dens=0.01, ndense=nd, seed=3)
At = sp.csc_matrix(P0.At)                       # (n+1) x m, row 0 empty
A = sp.csr_matrix(At.T)                         # m x (n+1)
N = A.shape[1]
x0 = rng.uniform(0.5, 2.0, N); z0 = rng.uniform(0.5, 2.0, N); y0 = np.zeros(m)
b = A @ x0
c = A.T @ rng.standard_normal(m) + z0           # a feasible dual exists: bounded problem
rows_dense = 1 + np.arange(nd)
denseA = sp.csc_matrix(At[rows_dense, :].T)
keep = np.ones(N); keep[rows_dense] = 0.0
Ps = problem.Problem(sp.csc_matrix(sp.diags(keep) @ At), P0.K, "lp"); Ps.At.eliminate_zeros()
Ps = problem.Problem(Ps.At, P0.K, "lp")
ADA = sp.csc_matrix(Ps.At.T @ Ps.At); ADA.data[:] = 1.0; ADA.sort_indices()
L = mex.symbchol(ADA)
LADsym = mex.symbfwblk(L, denseA)
perm, dz = mex.incorder(LADsym)
sym = mex.finsymbden(LADsym, perm, dz, float(nd + 1))
plan = bench.make_plan(0, Ps, L, ADA, problem.lorentz_pattern(Ps), {"l": np.ones(N), "det": np.zeros(0)}, np.zeros(0), np.zeros(m), None)
plan.set_dense(sym)
plan.upload("ad", np.asarray(denseA.todense()).ravel(order="F"))

x, z, y = np.ones(N), np.ones(N), np.zeros(m)
Ad = A.toarray()
log = []
for it in range(60):
    rp, rd, mu = b - A @ x, c - A.T @ y - z, x @ z / N
    gap = abs(c @ x - b @ y) / (1 + abs(c @ x))
    d = x / z
    # the hot path on these scalings (results unused: the numpy solve below keeps the iterates independent of the library)
    plan.upload("dl", d)
    plan.getada(); plan.blkchol(bench.PARS, True)
    fb = plan.deninfac(d[rows_dense], 500.0)
    general = bool(np.asarray(plan.lden()[0]["dopiv"]).any())      # a column whose rows were reordered: postponed pivots / dependent rows
    log.append({"it": it, "mu": float(mu), "cond_d": float(d.max() / d.min()), "host": bool(fb), "general": general})
    if max(np.linalg.norm(rp) / (1 + np.linalg.norm(b)), np.linalg.norm(rd) / (1 + np.linalg.norm(c)), gap) < 1e-9:
        break
    M = (Ad * d) @ Ad.T
    Lc = np.linalg.cholesky(M + 1e-14 * np.trace(M) / m * np.eye(m))

    def solve(rc):
        rhs = rp + Ad @ (d * rd - rc / z)
        dy = np.linalg.solve(Lc.T, np.linalg.solve(Lc, rhs))
        dz_ = rd - A.T @ dy
        dx = (rc - x * dz_) / z
        return dx, dy, dz_

    def step(v, dv):
        neg = dv < 0
        return min(1.0, 0.995 * float(np.min(-v[neg] / dv[neg]))) if neg.any() else 1.0
    dx, dy, dzz = solve(-x * z)
    ap, ad_ = step(x, dx), step(z, dzz)
    sig = (((x + ap * dx) @ (z + ad_ * dzz)) / N / mu) ** 3
    dx, dy, dzz = solve(sig * mu - x * z - dx * dzz)
    ap, ad_ = step(x, dx), step(z, dzz)
    x, y, z = x + ap * dx, y + ad_ * dy, z + ad_ * dzz
plan.close()
hits = [e["it"] for e in log if e["host"]]
gen = [e["it"] for e in log if e["general"]]
print(json.dumps({"problem": f"LP m={m} n={n} dense={nd}", "iterations": len(log), "final_mu": log[-1]["mu"], "final_cond_d": log[-1]["cond_d"],
                  "deninfac_calls": len(log), "host_algorithm_needed": len(hits), "at_iterations": hits,
                  "general_case_on_the_device": len(gen), "general_case_at_iterations": gen,
                  "cond_d_at_first_hit": (log[hits[0]]["cond_d"] if hits else None)}), flush=True)

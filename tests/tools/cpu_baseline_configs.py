"""TEST / MEASUREMENT INFRASTRUCTURE (uses oracle/).  CPU reference (oracle/_ref: the unmodified reference C, 1 core) timed on ONE iteration unit of the larger configs,
for the DESIGN.md table.  Usage (from the repo root): python tests/tools/cpu_baseline_configs.py maxcut2000|blockdiag"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from oracle import glue as gl
from sedumi_amd import problem
which = sys.argv[1] if len(sys.argv) > 1 else "maxcut2000"
P = problem.maxcut(int(which[6:])) if which.startswith("maxcut") else problem.blockdiag_sdp(nblk=64, n=200, mper=150, nnz=20, seed=4)
G = gl.Glue(); ref = G.ref
t0 = time.perf_counter(); S = G.setup(P.At, P.K); print("setup (ordering, symbolic, incorder) %.1f s" % (time.perf_counter() - t0), flush=True)
d, ud = problem.spd_scaling(P.K, seed=5)
K = P.K
dd = {"l": d["l"], "det": d["det"], "q1": np.ones(K["q"].size), "q2": np.zeros(0)}
DAt = G.getDAtm(S, dd)
dstruct = {"l": dd["l"].reshape(-1, 1), "det": dd["det"].reshape(-1, 1)}
pars = gl.default_pars_chol()
rhs = np.ones((P.m, 1))
t = {}
t["getada1"], A1 = ref.timed_call("getada1", 1, (S["ADA"], S["A"], S["Ablkjc"][:, 2], S["Aord"]["lqperm"], dstruct, K["qblkstart"]))
t["getada2"], A2 = ref.timed_call("getada2", 1, (A1, DAt, S["Aord"], K))
t["getada3"], (A3, absd) = ref.timed_call("getada3", 2, (A2, S["A"], S["Ablkjc"][:, 2], S["Aord"], ud.reshape(-1, 1), K))
t["blkchol"], (LL, Ld, _, _) = ref.timed_call("blkchol", 4, (S["L"], A3, dict(pars), absd))
L = dict(S["L"]); L["L"] = LL
ts = 0.0
for _ in range(4):
    tf, p = ref.timed_call("fwblkslv", 1, (L, rhs)); tb, _y = ref.timed_call("bwblkslv", 1, (L, p / Ld)); ts += tf[0] + tb[0]
tot = sum(v[0] for v in t.values()) + ts
print(which, {k: round(v[0], 4) for k, v in t.items()}, "solves(4x fw+bw) %.4f" % ts, "unit %.3f s -> %.3f units/s" % (tot, 1 / tot))

"""tools/mex_counters.py <workload> [units] -- the iteration unit through the built mexFunction shims (bench.py's mex_inclusive leg), with
the cache counters of sdm_mexcache printed after every unit and the stage times of every unit: shows WHEN something is rebuilt or
re-uploaded.  MEXHOST_DEFAULT_MALLOC=1 runs it with glibc's default allocation policy."""
import ctypes
import json
import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from sedumi_amd import capi, mex, mexhost  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "control07"
units = int(sys.argv[2]) if len(sys.argv) > 2 else 4
P, L, ADA, Q, d, ud, rhs, qpr, note = bench.build_workload(name, 0)
lib = capi.lib()
st = (ctypes.c_int64 * 16)()
K, m, At = P.K, P.m, sp.csc_matrix(P.At)
nlq = np.asarray(P.Ablkjc)[:, 2] - At.indptr[:-1]
Aord = {"lqperm": (np.argsort(nlq, kind="stable") + 1.0).reshape(-1, 1), "qperm": np.arange(1, m + 1, dtype=np.float64).reshape(-1, 1)}
Qm = sp.csc_matrix(Q)
if Qm.nnz:
    Qm = sp.csc_matrix((np.asarray(qpr, dtype=np.float64), Qm.indices, Qm.indptr), shape=Qm.shape)
    Aord["qperm"] = (np.argsort(np.diff(Qm.indptr), kind="stable") + 1.0).reshape(-1, 1)
sperm, _dz = mex.incorder(At, np.asarray(P.Ablkjc)[:, 2], float(np.asarray(K["mainblks"]).ravel()[2]))
Aord["sperm"] = np.asarray(sperm, dtype=np.float64).reshape(-1, 1)
dstruct = {"l": np.asarray(d["l"], dtype=np.float64).reshape(-1, 1), "det": np.asarray(d["det"], dtype=np.float64).reshape(-1, 1)}
lib.sdm_mexcache_clear()
host = mexhost.MexHost(None)
names = ["ada_build", "ada_reuse", "ada_upload", "ada_resident", "chol_build", "chol_reuse", "x_upload", "x_resident", "solve_resident", "solve_stateless", "at_upload", "host_words_checksummed", "epoch", "checksum_ns", "checksum_calls"]


def after_unit(*_):
    lib.sdm_mexcache_stats(st, ctypes.c_int64(16))
    print(json.dumps(dict(zip(names, list(st)[:15]))), flush=True)


times, y = mexhost.iteration_units(host, At, np.asarray(P.Ablkjc)[:, 2], Aord, K, dstruct, {"q": Qm}, ud, L, ADA, bench.PARS, rhs, units, 4, check=after_unit)
for t in times:
    print({k: round(1e3 * v, 2) for k, v in t.items()}, flush=True)
lib.sdm_mexcache_clear()

"""tools/phase_front.py [m] -- in-kernel phase clocks of k_ldl_front on one dense front (needs `python -m sedumi_amd.build --phases`)."""
import ctypes as C
import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sedumi_amd import capi, problem  # noqa: E402
capi.use_library(os.path.join(ROOT, "sedumi_amd", "lib", "libsedumi_hip_phases.so"))
from sedumi_amd.plan import Plan  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 666
lib = capi.lib()
rng = np.random.default_rng(1)
B = rng.standard_normal((m, m))
X = sp.csc_matrix(B @ B.T + m * np.eye(m)); X.sort_indices()
plan = Plan(0)
plan.set_chol(problem.dense_symbolic(m), X)
plan.upload("ada", X.data)
for _ in range(3):
    plan.blkchol(None, False)
plan.sync()
buf = (C.c_longlong * 32)()
lib.sdm_debug_phases_chol(buf, 1)
plan.blkchol(None, False); plan.sync()
lib.sdm_debug_phases_chol(buf, 0)
v = np.array(list(buf), dtype=np.float64) / 100.0
npan = (m + 63) // 64
print("m=%d, %d panels; us summed over the panels:" % (m, npan))
print("  D (work-item 0 of the block's workgroup): load %.0f | sweeps %.0f | trailing %.0f | barrier %.0f | copy %.0f | barrier %.0f | write-back + publish %.0f" % tuple(v[16:23]))
print("  inside the sweeps: look-ahead (LDS + products) %.0f | pivots %.0f | write-back + bookkeeping %.0f" % tuple(v[6:9]))
print("  next block's workgroup: until the last 16 columns arrive %.0f | stage + triangle %.0f | rows stored + acknowledged + counted %.0f | fence + counters %.0f | diagonal tile %.0f | stores + count %.0f"
      % tuple(v[0:6]))
print("  diagonal tile inside update_tile: loads+fill %.0f mfma %.0f to S + HBM %.0f" % (v[14], v[15], v[31]))

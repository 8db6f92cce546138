"""tools/phase_chain.py [m] -- in-kernel phase clocks of k_ldl_front's chain workgroup on one dense front (needs
`python -m sedumi_amd.build --phases`): where a step's time goes -- the sweeps, the wait for the other casts, what the row
wavefronts had left after the last sweep."""
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
plan.timer_begin(0); plan.blkchol(None, False); plan.timer_end(0); plan.sync()
lib.sdm_debug_phases_chol(buf, 0)
raw = np.array(list(buf), dtype=np.float64)
v = raw / 100.0
npan = (m + 63) // 64
print("m=%d, %d panels, blkchol %.1f us; us summed over the steps:" % (m, npan, 1e3 * plan.timer_ms(0)))
print("  wavefront 0: sweeps %.1f | waiting at X1 %.1f | after X1 %.1f" % tuple(v[0:3]))
print("  row wavefront 0: behind the sweeps %.1f | left after the last sweep %.1f | X1 %.1f | epilogue %.1f" % tuple(v[3:7]))
print("  row blocks done when the sweeps ended (sum) %d, staged %d of %d steps" % (raw[8], raw[9], raw[10]))

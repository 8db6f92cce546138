"""tools/trace_front.py [m] -- time stamps along the chain of k_ldl_front on one dense front (needs `python -m sedumi_amd.build
--phases`): per panel q when D(q)'s sweeps start and end, when the four 16-column groups of block q arrive at the workgroup of
tile row q+1, when that workgroup has its diagonal tile ready and when its rows are acknowledged."""
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
plan.timer_begin(0); plan.blkchol(None, False); plan.timer_end(0); plan.sync()
buf = (C.c_longlong * 2048)()
lib.sdm_debug_trace_chol(buf)
t = np.array(list(buf), dtype=np.float64).reshape(-1, 16) / 100.0
npan = (m + 63) // 64
t0 = t[0, 0]
print("m=%d, blkchol %.1f us.  us since D(0) started; per panel: D start | sweeps end | groups 0..3 arrive at the next workgroup | its tile ready | rows acknowledged" % (m, 1e3 * plan.timer_ms(0)))
for q in range(npan):
    r = t[q] - t0
    print("  q=%2d  %7.2f  %7.2f | %7.2f %7.2f %7.2f %7.2f | %7.2f  %7.2f   sweeps %.2f, end->last group %.2f, ->tile %.2f, ->acks %.2f, ->next D %.2f" % (
        q, r[0], r[1], r[2], r[3], r[4], r[5], r[7], r[8], r[1] - r[0], r[5] - r[1], r[7] - r[5], r[8] - r[7], (t[q + 1, 0] - t[q, 8]) if q + 1 < npan else 0))

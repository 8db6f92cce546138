"""tools/phase_prepare.py [workload] -- in-kernel phase clocks of the solve preparation (needs `python -m sedumi_amd.build --phases`;
run with SDM_SPREP_OFF=1 for the four-launch form).  Work-item 0 of every workgroup adds wall_clock64 ticks (100 MHz)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sedumi_amd import capi  # noqa: E402
capi.use_library(os.path.join(ROOT, "sedumi_amd", "lib", "libsedumi_hip_phases.so"))
import bench  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "control07"
lib = capi.lib()
P, L, ADA, Q, d, ud, rhs, qpr, note = bench.build_workload(name, 1)
plan = bench.make_plan(0, P, L, ADA, Q, d, ud, rhs, qpr)
plan.getada()
for _ in range(3):
    plan.blkchol(bench.PARS, True)
plan.sync()
buf = (C.c_longlong * 32)()
lib.sdm_debug_phases_solve(buf, 1)
plan.blkchol(bench.PARS, True); plan.sync()
lib.sdm_debug_phases_solve(buf, 0)
v = np.array(list(buf), dtype=np.float64) / 100.0
print(name, "k_sinv128 (us summed over workgroups): loads+stage %.1f | 32x32 rows %.1f | bar %.1f | 64-level %.1f | bar+S stores %.1f | 128-level mma %.1f | store %.1f"
      % tuple(v[0:7]))
for mode in range(3):
    n = max(buf[10 + 4 * mode], 1)
    print("  k_stile mode %d: %d tiles, per tile: loads + K loop %.2f us | epilogue %.2f us" % (mode, buf[10 + 4 * mode], v[8 + 4 * mode] / n, v[9 + 4 * mode] / n))

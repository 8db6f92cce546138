"""tools/phase_tiles.py [workload] -- in-kernel phase clocks of the update-tile pipeline of k_ldl_panel (panel_role_tiles_stream; needs
`python -m sedumi_amd.build --phases`): per step of a tile workgroup, what work-item 0 (a wavefront that prepares first) and work-item 256
(one that issues its MFMAs first) spend where.  wall_clock64 ticks (100 MHz)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sedumi_amd import capi  # noqa: E402
capi.use_library(os.path.join(ROOT, "sedumi_amd", "lib", "libsedumi_hip_phases.so"))
lib = capi.lib()
import bench  # noqa: E402
P, L, ADA, Q, d, ud, rhs0, qpr, note = bench.build_workload(sys.argv[1] if len(sys.argv) > 1 else "maxcut4000", 0)
plan = bench.make_plan(0, P, L, ADA, Q, d, ud, rhs0, qpr)
plan.getada()
plan.blkchol(bench.PARS, True); plan.sync()
buf = (C.c_longlong * 32)()
lib.sdm_debug_phases_chol(buf, 1)
plan.blkchol(bench.PARS, True); plan.sync()
lib.sdm_debug_phases_chol(buf, 0)
v = np.array(list(buf), dtype=np.float64)
n = max(v[13], 1.0)
us = v / 100.0 / n
print("work-item 0 (a wavefront that prepares first), %d steps, us per step: finished tile stored %.2f | wait for operands + LDS writes %.2f | next tile's values loaded %.2f | "
      "next tile located %.2f | loads of the step after issued %.2f | MFMA loop + c update %.2f | barrier %.2f"
      % (int(n), us[0], us[9], us[24], us[25], us[10], us[11], us[12]))

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
capi.use_library(os.path.join(ROOT, "sedumi_amd", "lib", sys.argv[2] if len(sys.argv) > 2 else "libsedumi_hip_phases.so"))
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
print("work-item 0, %d steps (half a chunk of a macro tile each), us per step: new macro tile's values loaded %.2f | wait for the next step's operands + LDS writes %.2f | "
      "next tile located + loads of the step after issued %.2f | product (64 MFMAs per wavefront) %.2f | finished macro tile stored %.2f | barrier %.2f | sum %.2f"
      % (int(n), us[24], us[12], us[0], us[9], us[10], us[11], us[[24, 12, 0, 9, 10, 11]].sum()))

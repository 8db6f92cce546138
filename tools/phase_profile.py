"""In-kernel phase clocks of the ADA' kernels on the bench workload (needs `python -m sedumi_amd.build --phases`).
Work-item 0 of every workgroup adds wall_clock64 ticks (100 MHz) between marks; sums over workgroups are printed."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sedumi_amd import capi, problem  # noqa: E402
from sedumi_amd.plan import Plan  # noqa: E402

capi.use_library(os.path.join(ROOT, "sedumi_amd", "lib", "libsedumi_hip_phases.so"))
lib = capi.lib()
import bench  # noqa: E402
P, L, ADA, Q, d, ud, rhs0, qpr, note = bench.build_workload(sys.argv[1] if len(sys.argv) > 1 else "control07", 0)
plan = Plan(0)
plan.set_chol(L, ADA); plan.set_ada(P.At, P.Ablkjc, P.K, Q)
plan.upload("dl", d["l"]); plan.upload("ddet", d["det"]); plan.upload("udsqr", ud)
if qpr is not None:
    plan.upload("qpr", qpr)
for _ in range(3):
    plan.getada()
plan.sync()
buf = (C.c_longlong * 32)()
lib.sdm_debug_phases_ada(buf, 1)
plan.getada(); plan.sync()
lib.sdm_debug_phases_ada(buf, 0)
v = np.array(list(buf), dtype=np.float64) / 100.0
print("stage1 (us summed over %d tasks, %d with n>40): stage nz %.0f | Y %.0f | bar %.0f | mfma %.0f | bar %.0f | Z write+bar %.0f | targets %.0f"
      % (buf[30], buf[31], v[0], v[1], v[2], v[3], v[4], v[5], v[6]))

plan.blkchol(None, True); plan.sync()
lib.sdm_debug_phases_chol(buf, 1)
plan.blkchol(None, True); plan.sync()
lib.sdm_debug_phases_chol(buf, 0)
v = np.array(list(buf), dtype=np.float64) / 100.0
print("factor (k_ldl_panel), workgroup 0 of every panel launch, us summed: load %.0f | sweeps (wavefront 0) %.0f | trailing (helpers) %.0f | "
      "barrier %.0f | copy %.0f | barrier %.0f | write-back + publish %.0f | rows solved in the workgroup %.0f" % tuple(v[16:24]))
print("  diagonal tile of the previous update in workgroup 0: loads+fill %.0f mfma %.0f to S + HBM %.0f" % (v[14], v[15], v[31]))
print("  in-workgroup blocked substitution %.0f   ride-along update tiles (work-item 0 of each tile group): loads+fill %.0f mfma %.0f rmw %.0f"
      % (v[26], v[28], v[29], v[30]))

rhs = np.random.default_rng(0).standard_normal(P.m)
plan.upload("rhs", rhs)
plan.ldlsolve(); plan.sync()
lib.sdm_debug_phases_chol(buf, 1)
plan.ldlsolve(); plan.sync()
lib.sdm_debug_phases_chol(buf, 0)
v = np.array(list(buf), dtype=np.float64) / 100.0
print("solve (one launch, work-item 0, us): fw: stage %.1f | in-block solves %.1f | wait for stream %.1f | wait for next-block rows %.1f | far-wave stream (n/a) %.1f | tail %.1f"
      % tuple(v[0:6]))
print("  bw: prologue %.1f | next-block dots + barrier %.1f | in-block solves %.1f | wait for stream %.1f" % (v[8], v[9], v[10], v[11]))

"""tools/driver_shadow.py <name> [mex|plan] -- run the loop restatement along the REFERENCE hot path and, next to it on the
same inputs, the library; print per iteration how far L.d and the forward / backward solves are apart."""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa: E402
from driver import sedumi_loop as sl  # noqa: E402

name = sys.argv[1]
tier = sys.argv[2] if len(sys.argv) > 2 else "plan"
if os.environ.get("SDM_DRIVER_EMU"):
    helpers.use_emu()
_, At, K = helpers.load_golden(name)
g = np.load(os.path.join(ROOT, "tests", "golden", f"driver_{name}.npz"))
S = sl.Sedumi(At, g["b"], g["c"], K, internal=True)
S.hot = sl.ShadowHot(sl.RefHot(S.G), sl.HipHot() if tier == "mex" else sl.PlanHot())
r = S.solve()
print(" it | max d / min d | maxerr ADA | relerr L.d | relerr fw | relerr bw | backward error ref / lib | solves | skip lib/ref")
for c in S.hot.records:
    print("%3d | %13.3e | %10.2e | %10.2e | %9.2e | %9.2e | %10.2e %10.2e   | %6d | %d/%d" % (
        c["iter"], c["dcond"], c["ada"], c["d"], c["fw"], c["bw"], c.get("berr_ref", np.nan), c.get("berr_lib", np.nan), c["nsolves"], *c["nskip"]))
print("iter", r["iter"], "STOP", r["STOP"], "cx", r["cx"], "by", r["by"])

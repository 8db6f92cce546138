"""tools/driver_log.py <name> [mex|plan] -- run the loop restatement (tests/driver/sedumi_loop.py; test infrastructure) on the
GPU library and print its iteration log next to the committed reference-hot-path log (tests/golden/driver_<name>.npz)."""
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
S = sl.Sedumi(At, g["b"], g["c"], K, hot=sl.HipHot() if tier == "mex" else sl.PlanHot(), internal=True)
r = S.solve()
cols = [str(c) for c in g["cols"]]
ref = g["rows"]
print(" it |        b*y/x0 (ref)        b*y/x0 (lib) |  gap(ref)  gap(lib) | delta r/l   | rate r/l        | tP r/l          | tD r/l          | cg r/l    | skip r/l")
for i in range(max(len(r["rows"]), ref.shape[0])):
    a = {k: ref[i, cols.index(k)] for k in cols} if i < ref.shape[0] else None
    b = r["rows"][i] if i < len(r["rows"]) else None
    f = lambda d, k, fmt: (fmt % d[k]) if d is not None else "-"
    print("%3d | %s %s | %s %s | %s %s | %s %s | %s %s | %s %s | %s,%s %s,%s | %s %s" % (
        i + 1, f(a, "by_x0", "%19.12e"), f(b, "by_x0", "%19.12e"), f(a, "gap", "%9.3e"), f(b, "gap", "%9.3e"), f(a, "delta", "%5.3f"), f(b, "delta", "%5.3f"),
        f(a, "rate", "%6.4f"), f(b, "rate", "%6.4f"), f(a, "tP", "%6.4f"), f(b, "tP", "%6.4f"), f(a, "tD", "%6.4f"), f(b, "tD", "%6.4f"),
        f(a, "kcg1", "%d"), f(a, "kcg2", "%d"), f(b, "kcg1", "%d"), f(b, "kcg2", "%d"), f(a, "nskip", "%d"), f(b, "nskip", "%d")))
print("reference hot path: iter %d STOP %d cx %.12e by %.12e" % (int(g["iter"]), int(g["STOP"]), float(g["cx"]), float(g["by"])))
print("%-18s: iter %d STOP %d cx %.12e by %.12e" % (r["hot"], r["iter"], r["STOP"], r["cx"], r["by"]))

"""tools/arch0_iterations.py -- arch0.mat solved by the loop restatement (tests/driver, test infrastructure) with the reference hot path and with\nthe library (MEX-shaped and resident tier) on the same box: iteration counts, stop codes, optimal values."""
import sys, os
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_driver as td
import helpers
from driver import sedumi_loop as sl
r = td.reference_run("arch0")
print("reference hot path on this box: iter", r["iter"], "STOP", r["STOP"], "cx %.12e by %.12e" % (r["cx"], r["by"]))
helpers.use_hip()
for tier, hot in (("mex", sl.HipHot()), ("plan", sl.PlanHot())):
    q = td.run("arch0", hot)
    print("library (%s) on this box: iter" % tier, q["iter"], "STOP", q["STOP"], "cx %.12e by %.12e" % (q["cx"], q["by"]))

"""tools/driver_profile.py <name> -- where a whole solve by the product driver (sedumi_amd.driver) spends its host time: cProfile of one solve (after a
first, untimed one that pays the builds and the symbolic phase's caches), top functions by own time."""
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa: E402
helpers.use_hip()
import test_driver as td  # noqa: E402
from sedumi_amd.driver import loop as lp  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "control07"
At, K, g = td.problem(name)
lp.Sedumi(At, g["b"], g["c"], K, internal=True).solve()
t0 = time.time()
pr = cProfile.Profile(); pr.enable()
r = lp.Sedumi(At, g["b"], g["c"], K, internal=True).solve()
pr.disable()
print("%s: %d iterations in %.3f s" % (name, r["iter"], time.time() - t0))
for key in ("tottime", "cumtime"):
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats(key).print_stats(28); print(s.getvalue()[:6000])

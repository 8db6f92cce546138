"""tools/accuracy_probe.py <name> <iter,iter,...> [mex|plan] -- print tests/driver/accuracy.py's measurements (accuracy of
ADA', factor and solves of the reference and of the library against extended precision, on the scalings of a real run).
Test infrastructure (drives the oracle).  SDM_DRIVER_EMU=1 runs the library through the fiber emulator."""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa: E402
from driver import accuracy, sedumi_loop as sl  # noqa: E402

name = sys.argv[1]
iters = [int(x) for x in sys.argv[2].split(",")]
if os.environ.get("SDM_DRIVER_EMU"):
    helpers.use_emu()
_, At, K = helpers.load_golden(name)
g = np.load(os.path.join(ROOT, "tests", "golden", f"driver_{name}.npz"))
S = sl.Sedumi(At, g["b"], g["c"], K, internal=True)
accuracy.probe(S, iters, sl.HipHot(), verbose=True)

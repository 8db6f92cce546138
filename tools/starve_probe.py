"""tools/starve_probe.py [busy_cus] [ms] -- control07's shape factored (sdm_plan_blkchol_wait) while another process holds busy_cus
compute units for ms milliseconds (tests/gpuhog): how long the call takes and which path the plan is on afterwards.
SDM_LIB=<file in sedumi_amd/lib> selects a measurement build."""
import json
import os
import subprocess
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "gpuhog"))
if os.environ.get("SDM_LIB"):
    from sedumi_amd import capi
    capi.use_library(os.path.join(ROOT, "sedumi_amd", "lib", os.environ["SDM_LIB"]))
import build_hog  # noqa: E402
from sedumi_amd import problem  # noqa: E402
from sedumi_amd.plan import Plan  # noqa: E402

busy = int(sys.argv[1]) if len(sys.argv) > 1 else 232
ms = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
hog = build_hog.build()
m = 666
rng = np.random.default_rng(1)
B = rng.standard_normal((m, m))
X = sp.csc_matrix(B @ B.T + m * np.eye(m)); X.sort_indices()
plan = Plan(0)
plan.set_chol(problem.dense_symbolic(m), X)
plan.upload("ada", X.data)
plan.blkchol_wait(None, False)
d0 = plan.download("d")
code = "import ctypes, sys; sys.exit(ctypes.CDLL(%r).hog_run(0, %d, 150 * 1024, %d))" % (hog, busy, ms)
other = subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, text=True)
assert other.stdout.readline().strip() == "started"
time.sleep(0.05)
t0 = time.time()
err = None
try:
    plan.blkchol_wait(None, False)
except Exception as e:                     # noqa: BLE001
    err = str(e)
dt = time.time() - t0
other.wait(timeout=60)
plan.kprof(True); plan.blkchol(None, False); plan.sync(); prof = plan.kprof_summary(); plan.kprof(False)
print(json.dumps({"lib": os.environ.get("SDM_LIB", ""), "busy_cus": busy, "other_process_ms": ms, "blkchol_wait_s": round(dt, 4), "error": err,
                  "on_panel_path_afterwards": "k_ldl_panel" in prof, "d_equal": bool(err is None and np.array_equal(plan.download("d"), d0))}), flush=True)

"""tools/variant_check.py -- a measurement build (SDM_LIB=<file in sedumi_amd/lib>) against the committed golden fixtures of the reference's
own examples and a seeded set of rank-deficient fronts of 100 .. 330 rows (decisions index by index, pivots to 1e-8), then the factor
times of arch0, nb and control07.  One line per check."""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from sedumi_amd import capi, mex  # noqa: E402
if os.environ.get("SDM_LIB"):
    capi.use_library(os.path.join(ROOT, "sedumi_amd", "lib", os.environ["SDM_LIB"]))
import helpers  # noqa: E402
from oracle.refmex import RefMex, REF_DIR  # noqa: E402

for name in ("arch0", "nb", "control07"):
    for tag in ("rand", "init"):
        try:
            helpers.check_golden(name, tag)
            print("golden", name, tag, "ok", flush=True)
        except Exception as e:  # noqa: BLE001
            print("golden", name, tag, "FAILED", repr(e)[:200], flush=True)
ref = RefMex(REF_DIR)
rng = np.random.default_rng(4242)
bad = 0
ncase = 150
for case in range(ncase):
    args = helpers.rank_deficient_front_case(rng, 100, 330)
    rr = ref.call("blkchol", 4, *args)
    o = mex.blkchol(*args)
    ok = np.array_equal(o[2].indices, rr[2].indices) and np.array_equal(o[3].indices, rr[3].indices) and helpers.relerr(o[1], rr[1]) < 1e-8
    bad += 0 if ok else 1
print("rank-deficient fronts of 100 .. 330 rows:", ncase - bad, "ok,", bad, "mismatches", flush=True)

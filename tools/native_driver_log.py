"""tools/native_driver_log.py <name>... -- the product driver (sedumi_amd.driver: native cone algebra, resident hot path) on the reference's examples
next to the same-host run with the compiled reference everywhere (tests/driver, test infrastructure): iterations, objectives, the last log rows."""
import json
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa: E402
helpers.use_hip()
import test_driver as td  # noqa: E402
from sedumi_amd.driver import loop as lp  # noqa: E402

for name in sys.argv[1:] or ["control07"]:
    At, K, g = td.problem(name)
    t0 = time.time()
    r = lp.Sedumi(At, g["b"], g["c"], K, internal=True).solve()
    t1 = time.time()
    if name in ("trto3", "OH_2Pi"):                     # (six CPU minutes each with the reference MEX: their reference-hot-path run is a committed fixture)
        ref = {"iter": int(g["iter"]), "cx": float(g["cx"]), "by": float(g["by"]), "STOP": None, "rows": []}
    else:
        ref = td.reference_run(name)
    t2 = time.time()
    print(json.dumps({"problem": name, "native_iter": r["iter"], "ref_iter": ref["iter"], "native_cx": r["cx"], "ref_cx": ref["cx"], "native_by": r["by"], "ref_by": ref["by"],
                      "opt": td.OPT.get(name), "native_STOP": r["STOP"], "ref_STOP": ref["STOP"], "native_feasratio": r["feasratio"], "native_s": t1 - t0, "reference_s": t2 - t1,
                      "last_rows_native": [{k: row[k] for k in ("by_x0", "gap", "prec", "delta", "rate", "kcg1", "kcg2")} for row in r["rows"][-3:]],
                      "last_rows_ref": [{k: row[k] for k in ("by_x0", "gap", "prec", "delta", "rate", "kcg1", "kcg2")} for row in ref["rows"][-3:]]}), flush=True)

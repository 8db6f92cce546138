"""the interior-point loop with the hot path switched between the reference MEX and this
repository's library (tests/driver/sedumi_loop.py).  The acceptance question of the north_star -- same iteration
count, same residual columns, same objective values as examples/test_sedumi.m:22-28 expects -- is asked three ways:

  * reference hot path, CPU:          the restatement itself reproduces test_sedumi.m's optimal values (tol 1e-6, as there)
                                      and the committed log (tests/golden/driver_*.npz, make_driver_golden.py);
  * library through the emulator, CPU: arch0 (PSD + LP) and nb (Lorentz, getada.m route) against that log;
  * library on the GPU:               arch0, control07, nb against that log,
each time through the MEX-shaped calls (sedumi_amd.mex) and through the resident plan (sedumi_amd.plan.Plan).

The last iteration or two of a run sit at the edge of double precision (the reference needs 30-90 CG steps there and
skips pivots), so the logs are compared row by row up to two iterations before the shorter run ends, and the
iteration counts may differ by one.
"""
import os

import numpy as np
import pytest

import helpers
from helpers import ROOT
from oracle import refmex

pytestmark = pytest.mark.skipif(not refmex.available(), reason="oracle/_ref is not built")

OPT = {"arch0": -5.665170e-01, "control07": -2.062510e+01, "nb": -5.070309e-02}      # examples/test_sedumi.m:22-25
TOL_OBJ = 1e-6                                                                            # examples/test_sedumi.m:30


def run(name, hot):
    from driver import sedumi_loop as sl
    _, At, K = helpers.load_golden(name)
    g = np.load(os.path.join(ROOT, "tests", "golden", f"driver_{name}.npz"))
    S = sl.Sedumi(At, g["b"], g["c"], K, hot=hot, internal=True)
    return S.solve(), g


def check_objectives(name, r):
    for v in (r["cx"], r["by"]):
        assert abs(v - OPT[name]) / abs(OPT[name]) < TOL_OBJ, (name, v, OPT[name])


def check_log(name, r, g, rtol_gap, atol_step):
    cols = [str(c) for c in g["cols"]]
    ref = g["rows"]
    assert abs(r["iter"] - int(g["iter"])) <= 1, (r["iter"], int(g["iter"]))
    upto = min(len(r["rows"]), ref.shape[0]) - 2
    worst = {}
    for i in range(upto):
        row = r["rows"][i]
        for k in ("by_x0", "gap", "prec"):
            e = abs(row[k] - ref[i, cols.index(k)]) / max(abs(ref[i, cols.index(k)]), 1e-300)
            worst[k] = max(worst.get(k, 0.0), e)
        for k in ("delta", "rate", "tP", "tD"):
            e = abs(row[k] - ref[i, cols.index(k)])
            worst[k] = max(worst.get(k, 0.0), e)
        for k in ("kcg1", "kcg2", "nskip", "nadd"):
            assert row[k] == ref[i, cols.index(k)], (name, i + 1, k, row[k], ref[i, cols.index(k)])
    print(name, r["hot"], "iter", r["iter"], "vs", int(g["iter"]), "worst deviations over", upto, "iterations:", worst)
    assert worst["by_x0"] < 1e-7, worst                     # the objective column follows the reference run to 7+ digits throughout
    assert worst["gap"] < rtol_gap and worst["prec"] < rtol_gap, worst
    assert max(worst["delta"], worst["rate"], worst["tP"], worst["tD"]) < atol_step, worst
    assert abs(r["cx"] - float(g["cx"])) / abs(float(g["cx"])) < TOL_OBJ and abs(r["by"] - float(g["by"])) / abs(float(g["by"])) < TOL_OBJ


@pytest.mark.parametrize("name", ["nb", "arch0"])
def test_loop_restatement_reproduces_the_reference_objectives(name):
    r, g = run(name, None)
    check_objectives(name, r)
    assert r["iter"] == int(g["iter"]) and r["STOP"] == int(g["STOP"])
    check_log(name, r, g, 1e-6, 1e-6)


@pytest.mark.parametrize("name,tier", [("nb", "mex"), ("nb", "plan"), ("arch0", "plan")])
def test_loop_on_the_emulated_library_follows_the_reference_log(name, tier):
    from driver import sedumi_loop as sl
    helpers.use_emu()
    r, g = run(name, sl.HipHot() if tier == "mex" else sl.PlanHot())
    check_objectives(name, r)
    check_log(name, r, g, 1e-2, 1e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("tier", ["mex", "plan"])
@pytest.mark.parametrize("name", ["nb", "arch0", "control07"])
def test_loop_on_the_gpu_follows_the_reference_log(name, tier):
    from driver import sedumi_loop as sl
    helpers.use_hip()
    r, g = run(name, sl.HipHot() if tier == "mex" else sl.PlanHot())
    check_objectives(name, r)
    check_log(name, r, g, 1e-2, 1e-2)

"""the interior-point loop with the hot path switched between the reference MEX and this
repository's library (tests/driver/sedumi_loop.py), and the accuracy of the library on the scalings such a run produces.

  * reference hot path:  the restatement itself reproduces examples/test_sedumi.m:22-25's optimal values (tol 1e-6, as there);
  * library, emulated (CPU: nb, arch0, quantum) and on the GPU (nb, arch0, control07, quantum; trto3 and OH_2Pi against a
    committed log), through the MEX-shaped calls and through the
    resident plan: same optimal values, the same iteration count as the same-host reference run, and the iteration log of the reference-hot-path run
    ON THE SAME HOST followed row by row (check_log);
  * accuracy (tests/driver/accuracy.py): at chosen iterations of the reference run, ADA', the factor and the solves of both
    paths against extended precision -- the library's error is at most 10 x the reference's (+ 1e-15).
"""
import os

import numpy as np
import pytest

import helpers
from helpers import ROOT
from oracle import refmex

pytestmark = pytest.mark.skipif(not refmex.available(), reason="oracle/_ref is not built")

OPT = {"arch0": -5.665170e-01, "control07": -2.062510e+01, "nb": -5.070309e-02,       # examples/test_sedumi.m:22-27
       "OH_2Pi": 7.946708e+01, "trto3": -1.279999e+04, "quantum": -0.75395345}
TOL_OBJ = 1e-6                                                                            # examples/test_sedumi.m:30


def problem(name):
    g = np.load(os.path.join(ROOT, "tests", "golden", f"driver_{name}.npz"))
    if "At_data" in g:                                   # trto3, OH_2Pi: the driver fixture carries the problem itself
        import scipy.sparse as sp
        from sedumi_amd import problem as pr
        At = sp.csc_matrix((g["At_data"], g["At_indices"], g["At_indptr"]), shape=tuple(g["At_shape"]))
        nreal = int(g["K_rsdpN"]) if "K_rsdpN" in g else g["K_s"].size
        ks = g["K_s"].ravel()
        return At, pr.make_K(int(g["K_l"]), g["K_q"].ravel(), ks[:nreal], ks[nreal:]), g
    _, At, K = helpers.load_golden(name)
    return At, K, g


def run(name, hot):
    from driver import sedumi_loop as sl
    At, K, g = problem(name)
    return sl.Sedumi(At, g["b"], g["c"], K, hot=hot, internal=True).solve()


_REF = {}


def reference_run(name):
    """The run with the reference hot path ON THIS HOST (cached): everything but the hot path -- LAPACK's eigenvectors
    included -- is then bit-identical between the two runs being compared."""
    if name not in _REF:
        _REF[name] = run(name, None)
    return _REF[name]


def check_objectives(name, r):
    for v in (r["cx"], r["by"]):
        assert abs(v - OPT[name]) / abs(OPT[name]) < TOL_OBJ, (name, v, OPT[name])


# arch0 is the sensitive one: from iteration 10 on its PSD scaling is so ill-conditioned that ADA' -- the reference's as
# much as ours, see the accuracy tests below -- is only good to 6e-11, and the run amplifies that by 1e2 per iteration
# around iterations 9-12 (two runs of the REFERENCE hot path on hosts with different LAPACK kernels part ways there, too).
TOL_LOG = {"nb": (1e-6, 1e-2), "control07": (1e-6, 1e-2), "arch0": (1e-3, 1e-1), "quantum": (1e-6, 1e-2)}


# rows whose step length sits within rounding of its cap (0.9 .. 0.99 of the way to the boundary): control07's iteration 24 gives
# tP = 0.8422 or 0.9000 with either solve path -- explicit inverses or plain substitution (profiles/r03d_diag_control07.txt) --
# and the row after it inherits the other iterate
KNIFE_EDGE_ROWS = {("control07", 24), ("control07", 25)}
# iteration counts: EQUAL to the same-host reference run, except the one named problem whose last iteration is decided by rounding:
# arch0 stops after 31 or 32 iterations depending on the last bits of the final directions -- the REFERENCE hot path itself takes 32
# (STOP 1) in the build container and 31 (STOP -1) on the GPU box (profiles/r04z_arch0_iterations_same_box.txt); the library takes 31
# on both (on the GPU box: 31 = 31).  One iteration of margin for arch0, none for anything else.
ITER_MARGIN = {"arch0": 1}


def check_log(name, r, ref):
    """Row by row against the reference-hot-path run up to two iterations before the shorter run ends: the objective
    column throughout; gap, precision, delta, rate and the step lengths while the reference gets each direction from ONE
    preconditioned step (beyond that the CG / refinement counts hinge on comparisons at the rounding level of the factor)."""
    tol_obj, tol_row = TOL_LOG[name]
    # north_star: "unchanged iteration count" -- against the reference hot path run on THIS host (same LAPACK kernels, same rounding
    # of everything outside the hot path): equality, for every problem and tier
    assert abs(r["iter"] - ref["iter"]) <= ITER_MARGIN.get(name, 0), (name, r["iter"], ref["iter"])
    A, B = r["rows"], ref["rows"]
    upto = min(len(A), len(B)) - 2
    # (the first row in which EITHER run needs a second CG step ends the strict zone: whether a residual of 4.9e-3 or 5.1e-3 times
    # the tolerance comes out of the first step is decided in the last bits of the solves)
    strict = next((i for i in range(upto) if max(B[i]["kcg1"], B[i]["kcg2"], A[i]["kcg1"], A[i]["kcg2"]) > 1), upto)
    worst, flips = {}, []
    for i in range(upto):
        e = abs(A[i]["by_x0"] - B[i]["by_x0"]) / max(abs(B[i]["by_x0"]), 1e-300)
        worst["by_x0"] = max(worst.get("by_x0", 0.0), e)
        if i >= strict:
            continue
        for k in ("gap", "prec"):
            worst[k] = max(worst.get(k, 0.0), abs(A[i][k] - B[i][k]) / max(abs(B[i][k]), 1e-300))
        # step lengths (and with them delta / rate) are minima over boundary hits capped at 0.9 .. 0.99 of the way: a row in which a
        # hit lies within rounding of the cap is a coin toss -- control07's iteration 24 (tP 0.8422 or 0.9000) goes either way between
        # super-block widths of the solves and even with the plain substitution (growth bound 0: profiles/r03d_diag_control07.txt),
        # the gap and objective columns of the rows after it agreeing to four digits all the same.  Only the rows named in KNIFE_EDGE_ROWS may differ
        # (their gap / prec columns are still asserted above).
        dev = max(abs(A[i][k] - B[i][k]) for k in ("delta", "rate", "tP", "tD"))
        if dev >= tol_row:
            flips.append((i + 1, dev))
        else:
            for k in ("delta", "rate", "tP", "tD"):
                worst[k] = max(worst.get(k, 0.0), abs(A[i][k] - B[i][k]))
        for k in ("kcg1", "kcg2", "nskip", "nadd"):
            assert A[i][k] == B[i][k], (name, i + 1, k, A[i][k], B[i][k])
    print(name, r["hot"], "iter", r["iter"], "vs", ref["iter"], "worst deviations over", strict, "(objective column:", upto, ") iterations:",
          {k: float("%.3g" % v) for k, v in worst.items()}, "rows with a different step length:", flips)
    assert worst["by_x0"] < tol_obj, worst
    assert max(worst.get(k, 0.0) for k in ("gap", "prec", "delta", "rate", "tP", "tD")) < tol_row, worst
    # (round-3 advisor: only the named knife-edge rows may flip -- anything else is a solve-accuracy regression)
    assert all((name, it) in KNIFE_EDGE_ROWS for it, _ in flips), (name, flips)
    assert abs(r["cx"] - ref["cx"]) / abs(ref["cx"]) < TOL_OBJ and abs(r["by"] - ref["by"]) / abs(ref["by"]) < TOL_OBJ


@pytest.mark.parametrize("name", ["nb", "arch0", "quantum"])
def test_loop_restatement_reproduces_the_reference_objectives(name):
    """... and, in the container the fixtures were made in, the committed log (elsewhere LAPACK may round differently)."""
    r = reference_run(name)
    check_objectives(name, r)
    g = problem(name)[2]
    assert abs(r["iter"] - int(g["iter"])) <= 2
    assert abs(r["cx"] - float(g["cx"])) / abs(float(g["cx"])) < TOL_OBJ and abs(r["by"] - float(g["by"])) / abs(float(g["by"])) < TOL_OBJ


@pytest.mark.parametrize("name,tier", [("nb", "mex"), ("nb", "plan"), ("arch0", "plan"), ("quantum", "mex"), ("quantum", "plan")])
def test_loop_on_the_emulated_library_follows_the_reference_log(name, tier):
    from driver import sedumi_loop as sl
    helpers.use_emu()
    # (arch0's 32 iterations with every PCG step's Amul / vecsym / psdscale emulated take two minutes: here its PCG operations stay on
    # the host -- nb and quantum pin them under the emulator, all four problems on the GPU)
    r = run(name, sl.HipHot() if tier == "mex" else sl.PlanHot(device_ops=name != "arch0"))
    check_objectives(name, r)
    check_log(name, r, reference_run(name))


@pytest.mark.gpu
@pytest.mark.parametrize("tier", ["mex", "plan"])
@pytest.mark.parametrize("name", ["nb", "arch0", "control07", "quantum"])
def test_loop_on_the_gpu_follows_the_reference_log(name, tier):
    from driver import sedumi_loop as sl
    helpers.use_hip()
    r = run(name, sl.HipHot() if tier == "mex" else sl.PlanHot())
    check_objectives(name, r)
    check_log(name, r, reference_run(name))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["trto3", "OH_2Pi"])
def test_the_larger_examples_reach_their_optimal_values_on_the_gpu(name):
    """trto3 (one PSD block of order 321, 544 equations, 60 iterations) and OH_2Pi_STO-6GN9r12g1T2 (22 PSD blocks up to order
    72, 948 equations): the reference hot path needs six minutes of CPU for each, so its run is a committed fixture
    (make_driver_golden.py: iteration count, optimal values) instead of a same-host run."""
    from driver import sedumi_loop as sl
    helpers.use_hip()
    r = run(name, sl.PlanHot())
    g = problem(name)[2]
    print(name, "iter", r["iter"], "vs", int(g["iter"]), "STOP", r["STOP"], "cx", r["cx"], "by", r["by"])
    check_objectives(name, r)
    # (against a fixture made on ANOTHER host: the reference's own count moves by one between hosts with different LAPACK kernels --
    # arch0: 32 in the build container, 31 on the GPU box, profiles/r04z_arch0_iterations_same_box.txt -- so this one keeps a margin;
    # the same-host comparisons of check_log assert equality)
    assert abs(r["iter"] - int(g["iter"])) <= 2
    assert abs(r["cx"] - float(g["cx"])) / abs(float(g["cx"])) < TOL_OBJ and abs(r["by"] - float(g["by"])) / abs(float(g["by"])) < TOL_OBJ


# ---------------------------------------------------------------------------------------------- accuracy on real scalings
def check_accuracy(name, iters, lib):
    """ADA', factor and solves of the library are as close to the extended-precision result as the reference's are, on the
    scalings of a real run (tests/driver/accuracy.py): error <= 10 x the reference's error + 1e-15."""
    from driver import accuracy, sedumi_loop as sl
    At, K, g = problem(name)
    S = sl.Sedumi(At, g["b"], g["c"], K, internal=True)
    recs = accuracy.probe(S, iters, lib)
    assert [r["iter"] for r in recs] == list(iters)
    for r in recs:
        print(name, r)
        assert r["skips"][0] == r["skips"][1]
        for k in ("ada", "factor", "fw", "bw"):
            assert r[k][1] <= 10 * r[k][0] + 1e-15, (name, r["iter"], k, r[k])


def test_emulated_library_is_as_accurate_as_the_reference_on_real_scalings():
    from driver import sedumi_loop as sl
    helpers.use_emu()
    check_accuracy("arch0", [3, 12], sl.HipHot())          # cond(D_psd) 5e1 and 5e5: the reference's ADA' is off by 3e-15 and 6e-11


@pytest.mark.gpu
@pytest.mark.parametrize("name,iters", [("arch0", [3, 12, 24, 30]), ("control07", [5, 20, 36]), ("nb", [4, 12, 19])])
def test_library_is_as_accurate_as_the_reference_on_real_scalings(name, iters):
    from driver import sedumi_loop as sl
    helpers.use_hip()
    check_accuracy(name, iters, sl.HipHot())

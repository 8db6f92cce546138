"""k_ldl_front exactly as the GPU runs it -- ONE launch, every workgroup waiting for the others' progress, the chain
workgroup's fused row solve, the data-tagged hand-over of the diagonal blocks -- on the CPU: the emulator forks one
process per workgroup over shared "device" memory (tests/hipemu: emu_launch_concurrent) instead of stepping the launch
through its phases.  Same sources, same arithmetic; what differs from the device is only the timing, which is the point:
an ordering the kernel relies on without enforcing it shows up here (the reintroduced round-3 bug -- rows stored under the
column probe of a later group -- fails 20 of 80 of the cases below)."""
import ctypes

import numpy as np
import pytest
import scipy.sparse as sp

import helpers
from helpers import TOL, relerr


@pytest.fixture()
def concurrent_emu():
    helpers.use_emu()
    from hipemu import build_emu
    lib = ctypes.CDLL(build_emu.build())
    lib._Z18emu_set_concurrenti(1)
    yield lib
    lib._Z18emu_set_concurrenti(0)


def _dense_front(m, seed):
    rng = np.random.default_rng(seed)
    B = rng.standard_normal((m, m))
    X = sp.csc_matrix(B @ B.T + m * np.eye(m)); X.sort_indices()
    return X, rng.standard_normal(m)


def _factor_solve(m, seed, lib, concurrent, one_launch=True):
    from sedumi_amd import problem
    from sedumi_amd.plan import Plan
    X, rhs = _dense_front(m, seed)
    lib._Z18emu_set_concurrenti(int(concurrent))
    plan = Plan(0)
    plan.set_one_launch_fronts(one_launch)
    plan.set_chol(problem.dense_symbolic(m), X)
    plan.upload("ada", X.data); plan.upload("rhs", rhs)
    plan.kprof(True)
    plan.blkchol(None, False); plan.ldlsolve()
    names = set(plan.kprof_summary().keys())
    plan.kprof(False)
    return plan.download("lpr"), plan.download("d"), plan.download("y"), names


@pytest.mark.parametrize("m", [330, 400, 666])
def test_concurrent_workgroups_give_the_bits_of_the_phased_launch(refmex, concurrent_emu, m):
    """m = 330: 6 row + 10 tile workgroups; 400: a partial last tile row; 666 (control07's shape): 11 + 45 processes."""
    from oracle import glue as gl
    from sedumi_amd import problem
    l1, d1, y1, k1 = _factor_solve(m, m, concurrent_emu, True)
    l0, d0, y0, k0 = _factor_solve(m, m, concurrent_emu, False)
    assert "k_ldl_front" in k0 and "k_ldl_panel" not in k0 and "k_ldl_front" in k1 and "k_ldl_panel" not in k1
    assert "k_sinv_follow" in k1          # beside the factorisation, polling its counters (a second group of processes)
    assert np.array_equal(l1, l0) and np.array_equal(d1, d0) and np.array_equal(y1, y0)
    X, _ = _dense_front(m, m)
    r = refmex.call("blkchol", 4, problem.dense_symbolic(m), X, gl.default_pars_chol())
    assert relerr(d1, r[1].ravel()) < TOL and relerr(l1, sp.csc_matrix(r[0]).data) < TOL


@pytest.mark.parametrize("reverse", [0, 1])
def test_rank_deficient_fronts_as_concurrent_workgroups(refmex, concurrent_emu, reverse):
    """The cases of the GPU soak (helpers.rank_deficient_front_case, 320 .. 450 rows here): skip / add decisions index by
    index and the pivots of the reference, with the column probe anywhere in a block.  reverse: the work-items of every
    workgroup scheduled in descending order (a missing barrier shows as a different result)."""
    from sedumi_amd import mex
    concurrent_emu._Z15emu_set_reversei(reverse)
    try:
        _rank_deficient_cases(refmex, mex, 777 + reverse, 18)
    finally:
        concurrent_emu._Z15emu_set_reversei(0)


def _rank_deficient_cases(refmex, mex, seed, ncases):
    rng = np.random.default_rng(seed)
    nadd = 0
    for case in range(ncases):
        args = helpers.rank_deficient_front_case(rng, 320, 450)
        rr = refmex.call("blkchol", 4, *args)
        o = mex.blkchol(*args)
        assert np.array_equal(o[2].indices, rr[2].indices) and np.array_equal(o[3].indices, rr[3].indices), (case, o[3].nnz, rr[3].nnz)
        assert relerr(o[1], rr[1]) < 1e-8, case
        nadd += rr[3].nnz
    assert nadd > 0


@pytest.mark.parametrize("m,maxu", [(400, 30.0), (400, 2.0)])
def test_pivot_rule_as_concurrent_workgroups(refmex, concurrent_emu, m, maxu):
    helpers.check_one_launch_pivot_rule(refmex, m, maxu)


@pytest.mark.parametrize("two_leaves", [False, True])
def test_levels_with_rows_below_as_concurrent_workgroups(refmex, glue, concurrent_emu, two_leaves):
    """Fronts with rows below their columns (update matrix for the parent), two fronts in one launch (grid.y = 2)."""
    helpers.check_one_launch_levels(refmex, glue, two_leaves)


def test_control07_unit_as_concurrent_workgroups(concurrent_emu):
    """The bench workload's iteration unit against its golden reference outputs, the factorisation as 56 concurrent processes."""
    errs = helpers.check_golden("control07", "rand")
    assert max(errs.values()) < TOL, errs


@pytest.mark.parametrize("m", [200, 700])
def test_panel_launches_as_concurrent_workgroups(refmex, concurrent_emu, m):
    """The launch-per-panel path (k_ldl_panel: diagonal block, row solves and the previous update's tiles in ONE launch whose
    roles wait for each other; k_sprep: the inverses for the solves, its workgroups chained by counters) with a process
    per workgroup, against the phased run and the reference.  200 rows: no row-solve workgroups (the roles are remapped)."""
    from oracle import glue as gl
    from sedumi_amd import problem
    l1, d1, y1, k1 = _factor_solve(m, m, concurrent_emu, True, one_launch=False)
    l0, d0, y0, k0 = _factor_solve(m, m, concurrent_emu, False, one_launch=False)
    assert "k_ldl_panel" in k1 and "k_ldl_front" not in k1 and "k_sprep" in k1
    assert np.array_equal(l1, l0) and np.array_equal(d1, d0) and np.array_equal(y1, y0)
    X, _ = _dense_front(m, m)
    r = refmex.call("blkchol", 4, problem.dense_symbolic(m), X, gl.default_pars_chol())
    assert relerr(d1, r[1].ravel()) < TOL and relerr(l1, sp.csc_matrix(r[0]).data) < TOL


# ---- the sparse / multi-front cases of tests/test_emu_parity.py once more, their factor launches as concurrent workgroups
# (levels of many small fronts: grid.y > 1; launches of more than 200 workgroups keep the phased form)
import test_emu_parity as _tp  # noqa: E402


@pytest.mark.parametrize("kind,m", [("rand", 120), ("arrow", 70), ("blockdiag", 100), ("grid", 100)])
def test_sparse_factor_and_solves_as_concurrent_workgroups(refmex, glue, concurrent_emu, kind, m):
    _tp.test_sparse_factor_and_solves(refmex, glue, kind, m)


@pytest.mark.parametrize("case", range(6))
def test_pivot_decisions_as_concurrent_workgroups(refmex, glue, concurrent_emu, case):
    _tp.test_pivot_decisions_skip_and_add(refmex, glue, case)


@pytest.mark.parametrize("kind,m,thr", [("rand", 200, 0.0), ("grid", 144, 0.0), ("bordered", 0, 1e4)])
def test_multifront_solves_as_concurrent_workgroups(refmex, glue, concurrent_emu, kind, m, thr):
    _tp.test_multifront_solves_with_and_without_fallback(refmex, glue, kind, m, thr)


def test_iteration_units_as_concurrent_workgroups(glue, concurrent_emu):
    _tp.test_iteration_unit_block_diagonal_multi_supernode(glue)
    _tp.test_iteration_unit_maxcut_small(glue)


@pytest.mark.parametrize("m,reverse", [(600, 0), (600, 1), (700, 0)])
@pytest.mark.parametrize("last_first", [0, 1])
def test_merged_sweep_launches_as_concurrent_workgroups(concurrent_emu, m, reverse, last_first):
    """k_sfw_rows_diag / k_sbw_step_diag (a row / step launch and the next diagonal block's launch as one) with one process
    per workgroup: the diagonal role's workgroups really wait for the urgent rows' chunk counters and read what those stored.  Two super-blocks of
    512, every row launch merged (SEDUMI_HIP_SWEEP_MERGE=2): 44 processes forward, 256 backward.  The bits of the separate launches; reverse: the
    work-items of every workgroup scheduled in descending order; last_first: the processes forked from the last workgroup down, so that the waiting
    role is up before the role it waits for (with the wait removed from the kernel this case fails; forked in grid order it does not)."""
    import os
    from sedumi_amd import problem
    from sedumi_amd.plan import Plan
    X, rhs = _dense_front(m, m)
    concurrent_emu._Z15emu_set_reversei(reverse)
    concurrent_emu._Z22emu_set_reverse_blocksi(last_first)
    try:
        plan = Plan(0)
        plan.set_solve_width(512)
        plan.set_chol(problem.dense_symbolic(m), X)
        plan.upload("ada", X.data); plan.upload("rhs", rhs)
        concurrent_emu._Z18emu_set_concurrenti(0)
        plan.blkchol(None, False)
        os.environ["SEDUMI_HIP_SWEEP_MERGE"] = "0"
        rhss = [rhs, np.cos(np.arange(m) * 0.37) + 2.0]               # (two right-hand sides in turn: what a solve leaves behind is NOT the next one's data)
        wants = []
        for r in rhss:
            plan.upload("rhs", r); plan.ldlsolve(); wants.append(plan.download("y"))
            assert relerr(X @ wants[-1], r) < 1e-10
        os.environ["SEDUMI_HIP_SWEEP_MERGE"] = "2"
        concurrent_emu._Z18emu_set_concurrenti(1)
        plan.kprof(True)
        for it in range(2):                                             # (the counter sets are re-armed between sweeps)
            plan.upload("rhs", rhss[it]); plan.upload("y", np.zeros(m)); plan.ldlsolve()
            assert np.array_equal(plan.download("y"), wants[it]), it
        prof = plan.kprof_summary(); plan.kprof(False)
        assert prof["k_sfw_rows_diag"][0] == 2 and prof["k_sbw_step_diag"][0] == 2, prof
        plan.close()
    finally:
        os.environ.pop("SEDUMI_HIP_SWEEP_MERGE", None)
        concurrent_emu._Z15emu_set_reversei(0)
        concurrent_emu._Z22emu_set_reverse_blocksi(0)

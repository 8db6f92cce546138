"""Kernel-logic parity on CPU: the HIP kernel sources of sedumi_amd/csrc compiled against the fiber emulator
(tests/hipemu) and run through the same C ABI / MEX-mirror calls as on the GPU, against the compiled reference.
These tests exist because the build container has no GPU; the GPU parity tests proper are test_gpu_parity.py."""
import ctypes

import numpy as np
import pytest
import scipy.sparse as sp

import helpers
from helpers import TOL, bordered_blocks, check_golden, check_iteration, load_golden, ref_scaling, relerr, spd_pattern, use_emu


@pytest.fixture(scope="module", autouse=True)
def _emu():
    use_emu()


def _set_reverse(rev):
    from sedumi_amd import capi
    lib = ctypes.CDLL(capi._lib_path)
    lib._Z15emu_set_reversei(int(rev))


@pytest.mark.parametrize("seed,kw", [(0, {}), (1, dict(m=20, lp=0, q=(), s=(6, 3))), (2, dict(m=35, lp=8, q=(4, 3, 5), s=())),
                                     (3, dict(m=40, block_local=True)), (4, dict(m=16, lp=3, q=(3,), s=(9,), dens=0.9)),
                                     (11, dict(m=24, lp=3, q=(3,), s=(4,), hs=(5, 3))),          # Hermitian PSD blocks (spcpxdxd)
                                     (12, dict(m=30, lp=0, q=(), s=(), hs=(6,), dens=0.5)),
                                     # every constraint in 12 / 18 PSD blocks: k_psd_stage2_ell stages z_j from that many segments (one batch / the task-by-task path)
                                     (13, dict(m=26, lp=2, q=(), s=(3,) * 12, dens=0.9)), (14, dict(m=26, lp=2, q=(), s=(3,) * 18, dens=0.9))])
def test_iteration_unit_small_mixed_cones(glue, seed, kw):
    from sedumi_amd import problem
    P = problem.random_sdp(seed=seed, **kw)
    for rev in (0, 1):                       # both work-item schedules: a missing barrier shows up in one of them
        _set_reverse(rev)
        check_iteration(glue, P, seed=seed)
    _set_reverse(0)
    check_iteration(glue, P, seed=seed, identity=True)


def test_iteration_unit_block_diagonal_multi_supernode(glue):
    from sedumi_amd import problem
    P = problem.blockdiag_sdp(nblk=5, n=12, mper=9, nnz=5, seed=3)
    errs, S, _ = check_iteration(glue, P, seed=1)
    assert S["L"]["xsuper"].size - 1 > 1


@pytest.mark.parametrize("kw", [dict(nblk=3, n=100, mper=10, nnz=12, seed=8), dict(nblk=2, n=130, mper=14, nnz=30, seed=9)])
def test_iteration_unit_blocks_above_96(glue, kw):
    """PSD blocks of order 100 and 130: the two-dot stage 1 of ADA' with the targets of a block regrouped for conflict-free
    LDS gathers (the order of U_k is the library's own business), stage by stage against the reference."""
    from sedumi_amd import problem
    P = problem.blockdiag_sdp(**kw)
    errs, S, _ = check_iteration(glue, P, seed=kw["seed"])
    assert max(errs.values()) < TOL, errs


def _few_nonzero_sdp(seed, m, blocks, lp=0):
    """Every constraint has at most two nonzeros per PSD block (diagonal entries, and off-diagonal ones stored in ONE triangle with the
    doubled value, as blockdiag_sdp does): the shape that takes the pairwise form of the PSD part (k_psd_direct)."""
    from sedumi_amd import problem
    rng = np.random.default_rng(seed)
    K = problem.make_K(lp + 1, [], list(blocks))
    start, _ = problem._psd_rows(K)
    rows, cols, vals = [], [], []
    for j in range(m):
        for k, n in enumerate(blocks):
            kind = rng.integers(0, 4)                      # 0: nothing in this block, 1: one diagonal, 2: one off-diagonal, 3: two entries
            if kind == 0 and not (k == j % len(blocks)):
                continue
            used = set()
            for _ in range(2 if kind == 3 else 1):
                r, c = sorted(rng.integers(0, n, size=2))
                if kind == 1:
                    c = r
                if (r, c) in used:
                    continue
                used.add((r, c))
                rows.append(start[k] + c + r * n); cols.append(j); vals.append(rng.standard_normal() * (1.0 if r == c else 2.0))
        for r in rng.integers(1, lp + 1, size=min(lp, 2)) if lp else []:
            rows.append(int(r)); cols.append(j); vals.append(rng.standard_normal())
    At = sp.csc_matrix((vals, (rows, cols)), shape=(int(K["N"]), m))
    At.sum_duplicates()
    return problem.Problem(At, K, f"few_nonzero_sdp(seed={seed})")


@pytest.mark.parametrize("seed,m,blocks,lp", [(1, 30, (40,), 0), (2, 24, (9, 14, 6), 0), (3, 40, (25, 12), 5)])
def test_iteration_unit_constraints_of_one_or_two_nonzeros_per_block(glue, seed, m, blocks, lp):
    """MAXCUT-type constraints: the PSD part of ADA' in its pairwise form (k_psd_direct, no z_j), stage by stage against the reference."""
    P = _few_nonzero_sdp(seed, m, blocks, lp)
    for rev in (0, 1):
        _set_reverse(rev)
        errs, S, _ = check_iteration(glue, P, seed=seed)
        assert max(errs.values()) < TOL, errs
    _set_reverse(0)


def test_iteration_unit_many_tiny_blocks_keep_the_two_stage_path(glue):
    """1500 PSD blocks of order 2 with at most two nonzeros per constraint and block: k_psd_direct's per-block LDS tables (56 bytes
    per block) would pass the dynamic-LDS limit, so ada_psd must fall back to stage 1 + stage 2 (round-3 advisor)."""
    P = _few_nonzero_sdp(7, 12, (2,) * 1500, 0)
    errs, S, _ = check_iteration(glue, P, seed=7)
    assert max(errs.values()) < TOL, errs


@pytest.mark.parametrize("n", [90, 133])
def test_direct_kernel_for_full_columns_matches_the_generic_one(n):
    helpers.check_direct_columns_kernel(n)


def test_iteration_unit_maxcut_small(glue):
    from sedumi_amd import problem
    errs, S, _ = check_iteration(glue, problem.maxcut(90), seed=5)
    assert max(errs.values()) < TOL, errs


@pytest.mark.parametrize("name", ["arch0", "nb"])
def test_golden_small_examples(name):
    for tag in ("init", "rand"):
        errs = check_golden(name, tag)
        assert max(errs.values()) < TOL, errs


@pytest.mark.parametrize("kind,m", [("rand", 120), ("band", 90), ("arrow", 70), ("blockdiag", 100), ("grid", 100), ("diag", 9)])
def test_sparse_factor_and_solves(refmex, glue, kind, m):
    from oracle import glue as gl
    from sedumi_amd import mex
    rng = np.random.default_rng(m)
    X = spd_pattern(kind, m, rng, 0.04)
    L = mex.symbchol(X)                       # own ordering + symbolic (bit-exact with the reference, test_oracle.py)
    pars = gl.default_pars_chol()
    r = refmex.call("blkchol", 4, L, X, pars)
    o = mex.blkchol(L, X, pars)
    assert relerr(o[0], r[0]) < TOL and relerr(o[1], r[1]) < TOL
    L2 = dict(L); L2["L"] = r[0]
    rhs = rng.standard_normal((X.shape[0], 3))
    assert relerr(mex.fwblkslv(L2, rhs), refmex.call("fwblkslv", 1, L2, rhs)) < TOL
    assert relerr(mex.bwblkslv(L2, rhs), refmex.call("bwblkslv", 1, L2, rhs)) < TOL


@pytest.mark.parametrize("m", [130, 200, 330, 1216])
def test_dense_front_with_several_row_batches(refmex, m):
    """One dense supernode whose rows below the first 64-column panel span several workgroups of the panel
    kernel (matrix-core row solve, deferred in-place copy of the diagonal block).  The emulator runs the
    workgroups of a launch one after the other, which turns any in-place update another workgroup still has
    to read into a deterministic failure.  m = 1216: 19 panels, up to 171 update tiles riding along with a
    diagonal-block launch, two per workgroup."""
    from oracle import glue as gl
    from sedumi_amd import mex, problem
    rng = np.random.default_rng(m)
    B = rng.standard_normal((m, m))
    X = sp.csc_matrix(B @ B.T + m * np.eye(m))
    L = problem.dense_symbolic(m)
    pars = gl.default_pars_chol()
    r = refmex.call("blkchol", 4, L, X, pars)
    o = mex.blkchol(L, X, pars)
    assert relerr(o[0], r[0]) < TOL and relerr(o[1], r[1]) < TOL and o[2].nnz == 0 and o[3].nnz == 0
    L2 = dict(L); L2["L"] = r[0]
    rhs = rng.standard_normal((m, 2))
    assert relerr(mex.fwblkslv(L2, rhs), refmex.call("fwblkslv", 1, L2, rhs)) < TOL
    assert relerr(mex.bwblkslv(L2, rhs), refmex.call("bwblkslv", 1, L2, rhs)) < TOL


@pytest.mark.parametrize("case", range(6))
def test_pivot_decisions_skip_and_add(refmex, glue, case):
    """Never-fail pivot rule (blkchol2.c:114-161): same skipped / diag-added columns as the reference."""
    from oracle import glue as gl
    from sedumi_amd import mex
    rng = np.random.default_rng(50 + case)
    m = [40, 90, 70, 100, 64, 33][case]
    if case % 2 == 0:
        B = rng.standard_normal((m, m // 2))
        X = B @ B.T
        X = sp.csc_matrix(X + np.diag(10.0 ** rng.uniform(-14, -2, m)) * np.abs(X).max())
    else:
        S = sp.random(m, m, density=0.05, random_state=rng, format="csc"); S = S + S.T
        sc = 10.0 ** rng.uniform(-7, 3, m)
        X = sp.diags(sc) @ (S + sp.diags(np.asarray(abs(S).sum(axis=1)).ravel() * rng.choice([1.0, 1.0, 0.5], m) + 1e-3)) @ sp.diags(sc)
        X = sp.csc_matrix(X); X.sort_indices()
    L = glue.symbchol(X)
    for maxu in (5e5, 30.0, 2.0):
        pars = dict(gl.default_pars_chol()); pars["maxu"] = maxu
        absd = np.abs(X.diagonal()) * rng.choice([1.0, 1e3, 1e8], m) if case > 2 else None
        args = (L, X, pars) + ((absd,) if absd is not None else ())
        r = refmex.call("blkchol", 4, *args)
        o = mex.blkchol(*args)
        assert np.array_equal(o[2].indices, r[2].indices) and np.array_equal(o[3].indices, r[3].indices)
        assert relerr(o[1], r[1]) < 1e-8
        if r[2].nnz:
            assert relerr(o[2], r[2]) < 1e-6
        if r[3].nnz:
            assert relerr(o[3], r[3]) < 1e-6


def test_sparse_rhs_solves(refmex, glue):
    """fwblkslv / bwblkslv with a sparse right-hand side and the symbfwblk pattern (deninfac.m:67)."""
    from oracle import glue as gl
    from sedumi_amd import mex
    rng = np.random.default_rng(3)
    X = spd_pattern("rand", 80, rng, 0.05)
    L = glue.symbchol(X)
    r = refmex.call("blkchol", 4, L, X, gl.default_pars_chol())
    L2 = dict(L); L2["L"] = r[0]
    B = sp.random(80, 4, density=0.05, random_state=rng, format="csc")
    Ys = refmex.call("symbfwblk", 1, L2, B)
    yr = refmex.call("fwblkslv", 1, L2, B, Ys)
    yo = mex.fwblkslv(L2, B, Ys)
    assert np.array_equal(yo.indices, yr.indices) and relerr(yo, yr) < TOL


def test_resident_plan_matches_reference(glue):
    from sedumi_amd import problem
    from sedumi_amd.plan import Plan
    from helpers import ref_scaling
    P = problem.random_sdp(m=30, seed=9)
    S = glue.setup(P.At, P.K)
    d, ud = ref_scaling(P, 4)
    it = glue.iteration_ref(S, d, ud)
    rhs = np.random.default_rng(1).standard_normal(P.m)
    plan = Plan(0)
    plan.set_chol(S["L"], S["ADA"])
    Qpat = sp.csc_matrix(S["DAt"]["q"])
    plan.set_ada(P.At, P.Ablkjc, P.K, Qpat)
    Qn = sp.csc_matrix(it["DAt"]["q"])
    cols = np.repeat(np.arange(P.m), np.diff(Qpat.indptr))
    plan.upload("dl", d["l"]); plan.upload("ddet", d["det"]); plan.upload("udsqr", ud); plan.upload("rhs", rhs)
    if Qpat.nnz:
        plan.upload("qpr", np.asarray(Qn[Qpat.indices, cols]).ravel())
    plan.getada(); plan.blkchol(None, True); plan.ldlsolve()
    assert relerr(plan.download("ada"), it["ADA"].data) < TOL
    assert relerr(plan.download("absd"), it["absd"].ravel()) < TOL
    assert relerr(plan.download("d"), it["Ld"].ravel()) < TOL
    assert relerr(plan.download("lpr"), it["LL"].data) < TOL
    assert relerr(plan.download("y"), glue.solve_ref(S, it, rhs).ravel()) < TOL
    (si, sv), (ai, av) = plan.pivots()
    assert len(si) == it["Lskip"].nnz and len(ai) == it["Ladd"].nnz
    prof_before = plan.kprof_summary()
    plan.kprof(True); plan.ldlsolve(); prof = plan.kprof_summary(); plan.kprof(False)
    assert not prof_before and "k_sfw_diag" in prof and "k_sbw_diag" in prof
    plan.close()


def test_bad_inputs_raise_like_mexErrMsgTxt():
    from sedumi_amd import mex
    from sedumi_amd.capi import SdmError
    L = {"L": sp.csc_matrix(np.tril(np.ones((4, 4)))), "perm": np.arange(1, 5.0)}
    with pytest.raises(SdmError, match="Missing field L.xsuper"):
        mex.fwblkslv(L, np.ones(4))
    L["xsuper"] = np.array([1.0, 5.0])
    with pytest.raises(SdmError, match="Size mismatch b"):
        mex.fwblkslv(L, np.ones(5))
    with pytest.raises(SdmError):
        mex.fwblkslv(L, sp.csc_matrix(np.ones((4, 1))))       # sparse b needs ysymb (fwblkslv.c:243-244)


def test_big_single_front_solves(refmex):
    """Single dense front of 1100 rows = five 256-column super-blocks: diagonal inverses at all three levels (64 / 128 / 256), premultiplied block rows, four steps per sweep."""
    from sedumi_amd import mex, problem
    m = 1100
    rng = np.random.default_rng(7)
    Lv = np.tril(rng.standard_normal((m, m)) * (0.3 / np.sqrt(m)), -1) + np.eye(m)
    L = problem.dense_symbolic(m)
    L["L"] = sp.csc_matrix(Lv)
    rhs = rng.standard_normal((m, 1))
    assert relerr(mex.fwblkslv(L, rhs), refmex.call("fwblkslv", 1, L, rhs)) < TOL
    assert relerr(mex.bwblkslv(L, rhs), refmex.call("bwblkslv", 1, L, rhs)) < TOL


@pytest.mark.parametrize("n1,n2,nc", [(100, 130, 900), (128, 65, 900), (64, 192, 900), (128, 200, 841)])
def test_pipelined_sweeps_full_workgroup_fronts(refmex, glue, n1, n2, nc):
    """Fronts of >= 961 rows run the sweeps with 16 wavefronts and the look-ahead schedule (front_fw_pipe /
    front_bw_pipe): rows right below a panel from prefetched registers, rows beyond overlapped with the next in-block
    solve, partial last panels, odd row counts, rows below the supernode's own columns."""
    from oracle import glue as gl
    from sedumi_amd import mex
    rng = np.random.default_rng(n1 + nc)
    X = bordered_blocks(n1, n2, nc, rng)
    L = glue.symbchol(X)
    xs = L["xsuper"].ravel().astype(int)
    assert xs.size - 1 >= 2 and np.diff(L["L"].indptr)[xs[0] - 1] >= 961 > xs[1] - xs[0]
    r = refmex.call("blkchol", 4, L, X, gl.default_pars_chol())
    L2 = dict(L); L2["L"] = r[0]
    rhs = rng.standard_normal((X.shape[0], 2))
    for rev in (0, 1):
        _set_reverse(rev)
        assert relerr(mex.fwblkslv(L2, rhs), refmex.call("fwblkslv", 1, L2, rhs)) < TOL
        assert relerr(mex.bwblkslv(L2, rhs), refmex.call("bwblkslv", 1, L2, rhs)) < TOL
    _set_reverse(0)


@pytest.mark.parametrize("m", [961, 1000, 1023])
def test_pipelined_sweeps_single_front(refmex, m):
    """Single dense front of about 1000 rows (four super-blocks, the last one partial) on the look-ahead
    schedule; m = 1023 has an odd row count, m = 1000 a partial last panel."""
    from sedumi_amd import mex, problem
    rng = np.random.default_rng(m)
    Lv = np.tril(rng.standard_normal((m, m)) * (0.3 / np.sqrt(m)), -1) + np.eye(m)
    L = problem.dense_symbolic(m)
    L["L"] = sp.csc_matrix(Lv)
    rhs = rng.standard_normal((m, 1))
    assert relerr(mex.fwblkslv(L, rhs), refmex.call("fwblkslv", 1, L, rhs)) < TOL
    assert relerr(mex.bwblkslv(L, rhs), refmex.call("bwblkslv", 1, L, rhs)) < TOL


@pytest.mark.parametrize("m,thr", [(700, 0.0), (700, 1e-3), (300, 0.0), (90, 0.0)])
def test_solves_with_substitution_fallback_blocks(refmex, m, thr):
    """The solves apply the diagonal super-blocks of L (256 columns here: set_solve_width) as explicit inverses unless a
    block's growth max|inv| * max|L| exceeds the plan's bound; such a block is solved by substitution against the factor
    by one workgroup of the launch.  Bound 0: every block on the fallback; bound 1e-3 on a factor whose middle super-block
    has tiny multipliers: good and bad blocks mixed in one front."""
    from sedumi_amd import problem
    from sedumi_amd.plan import Plan
    rng = np.random.default_rng(m)
    Lv = np.tril(rng.standard_normal((m, m)) * (0.5 / np.sqrt(m)), -1) + np.eye(m)
    if thr > 0:
        Lv[256:512, 256:512] = np.tril(Lv[256:512, 256:512], -1) * 1e-3 + np.eye(256)      # growth ~ 1e-4: stays on the inverse
    d = 0.5 + rng.random(m)
    X = Lv @ np.diag(d) @ Lv.T
    L = problem.dense_symbolic(m)
    plan = Plan(0); plan.set_solve_width(256); plan.set_chol(L, problem.dense_pattern(m))
    plan.set_growth_max(thr)
    plan.upload("ada", X.ravel(order="F"))
    plan.blkchol(None, False)
    nb, bad, _ = plan.solve_stats()
    assert nb == (m + 255) // 256 and bad == (nb if thr == 0 else nb - 1)
    Ll = dict(L); Ll["L"] = sp.csc_matrix(np.tril(plan_dense_L(plan, m)))
    rhs = rng.standard_normal((m, 1))
    plan.upload("rhs", rhs.ravel()); plan.fwsolve()
    assert relerr(plan.download("y"), refmex.call("fwblkslv", 1, Ll, rhs).ravel()) < TOL
    plan.bwsolve()
    assert relerr(plan.download("y"), refmex.call("bwblkslv", 1, Ll, rhs).ravel()) < TOL
    plan.ldlsolve()
    dd = plan.download("d")
    want = refmex.call("bwblkslv", 1, Ll, refmex.call("fwblkslv", 1, Ll, rhs) / dd.reshape(-1, 1)).ravel()
    assert relerr(plan.download("y"), want) < TOL
    plan.close()


def plan_dense_L(plan, m):
    Lp = plan.download("lpr")
    out = np.zeros((m, m))
    out[np.tril_indices(m)[::-1]] = 0
    M = sp.csc_matrix((Lp, plan.L_pattern.indices, plan.L_pattern.indptr), shape=(m, m))
    return M.toarray()


@pytest.mark.parametrize("kind,m,thr", [("rand", 200, 0.0), ("grid", 144, 0.0), ("bordered", 0, 0.0), ("bordered", 0, 1e4), ("arrow", 80, 0.0)])
def test_multifront_solves_with_and_without_fallback(refmex, glue, kind, m, thr):
    """Multi-front factors (children's update vectors, rows below the supernodes, several etree levels) through the
    resident solves with every super-block on the substitution fallback (bound 0) and on the inverse path."""
    from oracle import glue as gl
    from sedumi_amd.plan import Plan
    rng = np.random.default_rng(17)
    X = bordered_blocks(300, 70, 40, rng) if kind == "bordered" else spd_pattern(kind, m, rng, 0.04)
    L = glue.symbchol(X)
    n = X.shape[0]
    r = refmex.call("blkchol", 4, L, X, gl.default_pars_chol())
    plan = Plan(0); plan.set_chol(L, X)
    plan.set_growth_max(thr)
    plan.upload("ada", sp.csc_matrix(X).data)
    plan.blkchol(gl.default_pars_chol(), False)
    nb, bad, _ = plan.solve_stats()
    assert (0 < bad <= nb) if thr == 0 else bad == 0          # 1 x 1 blocks have growth 0: never "bad"
    assert relerr(plan.download("lpr"), r[0].data) < TOL
    Lr = dict(L); Lr["L"] = r[0]
    rhs = rng.standard_normal((n, 1))
    plan.upload("rhs", rhs.ravel())
    plan.fwsolve(); assert relerr(plan.download("y"), refmex.call("fwblkslv", 1, Lr, rhs).ravel()) < TOL
    plan.bwsolve(); assert relerr(plan.download("y"), refmex.call("bwblkslv", 1, Lr, rhs).ravel()) < TOL
    plan.ldlsolve()
    want = refmex.call("bwblkslv", 1, Lr, refmex.call("fwblkslv", 1, Lr, rhs) / r[1].reshape(-1, 1)).ravel()
    assert relerr(plan.download("y"), want) < TOL
    plan.close()


@pytest.mark.parametrize("kw", [dict(m=35, lp=8, q=(4, 3, 5), s=()), dict(m=20, lp=12, q=(), s=()), dict(m=40, lp=0, q=(6, 9, 3), s=(), dens=0.5)])
def test_getada_gateway_no_psd(glue, kw):
    """absd = getada(A,K,d,DAt) (getada.m:13-40, the route of sedumi.m:446-448 for sum(K.s)==0): the whole ADA' of an
    LP / SOCP problem in one call, against the reference's getada1 -> getada2 -> getada3 chain (whose result for a
    problem without PSD blocks is the same symmetric matrix) and absd = diag(ADA')."""
    from sedumi_amd import mex, problem
    P = problem.random_sdp(seed=23, **kw)
    S = glue.setup(P.At, P.K)
    d, ud = ref_scaling(P, 5)
    it = glue.iteration_ref(S, d, ud)
    ADA, absd = mex.getada(S["ADA"], S["A"], P.K, d, it["DAt"])
    assert relerr(ADA, it["ADA"]) < TOL
    assert relerr(absd.ravel(), it["ADA"].diagonal()) < TOL
    # the MATLAB formula itself
    A = sp.csc_matrix(S["A"]).toarray()
    nlq = int(P.K["mainblks"].ravel()[2]) - 1
    sv = np.concatenate((d["l"], -d["det"], np.zeros(nlq - int(P.K["l"]) - P.K["q"].size)))
    qb = P.K["qblkstart"].ravel().astype(int) - 1
    for i in range(P.K["q"].size):
        sv[qb[i]:qb[i + 1]] = d["det"][i]
    Q = sp.csc_matrix(it["DAt"]["q"]).toarray() if P.K["q"].size else np.zeros((0, P.m))
    want = Q.T @ Q + A[:nlq].T @ np.diag(sv) @ A[:nlq]
    assert relerr(ADA.toarray(), want) < 1e-12


def test_getada_gateway_on_the_nb_example():
    """examples/nb.mat (BASELINE.json configs[2]; sum(K.s)==0, so sedumi.m forms ADA' through getada): the committed
    reference ADA' / absd of the golden fixture through the getada gateway."""
    from sedumi_amd import mex, problem
    z, At, K = load_golden("nb")
    m = At.shape[1]
    for tag in ("init", "rand"):
        d = {"l": z[f"{tag}_dl"], "det": z[f"{tag}_ddet"]}
        Q = sp.csc_matrix((z[f"{tag}_DAtq_data"], z[f"{tag}_DAtq_indices"], z[f"{tag}_DAtq_indptr"]), shape=tuple(z[f"{tag}_DAtq_shape"]))
        ADA, absd = mex.getada(problem.dense_pattern(m), At, K, d, {"q": Q})
        A = ADA.toarray()
        assert relerr(A[np.triu_indices(m)], z[f"{tag}_ADA_triu"]) < TOL and relerr(A, A.T) < 1e-14
        assert relerr(absd.ravel(), z[f"{tag}_absd"]) < TOL


@pytest.mark.parametrize("m", [112, 123, 174, 330, 512, 666, 900, 1000])
def test_one_launch_front_matches_the_panel_launches_bit_for_bit(refmex, m):
    helpers.check_one_launch_front(refmex, m)


@pytest.mark.parametrize("m", [330, 900, 1000])
def test_streamed_update_tiles_give_the_same_bits_whatever_the_number_of_workgroups(refmex, m):
    """(m = 900, 15 tile rows: one deferred group whose far triangle goes in one launch; m = 1000, 16: two groups, the first one's far
    triangle dealt over four launches -- tile_sched in sdm_chol.hip)"""
    helpers.check_streamed_update_tiles(refmex, m)


@pytest.mark.parametrize("two_leaves", [False, True])
def test_one_launch_front_levels_with_rows_below(refmex, glue, two_leaves):
    helpers.check_one_launch_levels(refmex, glue, two_leaves)






@pytest.mark.parametrize("m,maxu", [(400, 5e5), (400, 30.0), (400, 2.0), (666, 30.0)])
def test_one_launch_front_pivot_rule(refmex, m, maxu):
    helpers.check_one_launch_pivot_rule(refmex, m, maxu)


@pytest.mark.parametrize("m,thr", [(90, None), (300, None), (666, None), (700, 0.0), (700, 1e-3), (256, 0.0), (530, None), (1100, None)])   # (1100: rows split over two wavefronts)
def test_solve_widths(m, thr):
    helpers.check_solve_widths(m, thr)


@pytest.mark.parametrize("m", [300, 530, 700])
def test_inverse_by_one_launch_and_by_a_launch_per_stage(m):
    helpers.check_inverse_launch_paths(m)


@pytest.mark.parametrize("m,seed,glo,ghi,cancelling", [(320, 3, 1e5, 1e7, False), (400, 7, 3e6, 5e7, True), (400, 11, 3e8, 8e9, False)])
def test_refined_solves_are_as_accurate_as_substitution(m, seed, glo, ghi, cancelling):
    """Super-blocks beyond the growth bound: explicit inverse + two refinement steps against the factor (k_sfw_resid / k_sbw_resid)
    give the accuracy of the substitution they replace -- measured against an extended-precision solve on ill-conditioned factors."""
    helpers.check_refined_solve_accuracy(m, seed, glo, ghi, cancelling)


def test_refinement_launches_follow_the_conditioning_of_the_factors():
    helpers.check_refinement_prediction()


def test_blocking_blkchol_recovers_from_a_starved_one_launch_level(refmex):
    """sdm_plan_blkchol_wait (what blkchol.mex and sdm_blkchol call): a one-launch level whose workgroups could not all become
    resident raises the plan's time-out flag; the factorisation is repeated once on the launch-per-panel path and the plan
    stays there.  The time-out is injected (tests/hipemu: emu_inject_timeouts)."""
    import ctypes
    from hipemu import build_emu
    from oracle import glue as gl
    from sedumi_amd import mex, problem
    lib = ctypes.CDLL(build_emu.build())
    m = 330
    rng = np.random.default_rng(3)
    B = rng.standard_normal((m, m))
    X = sp.csc_matrix(B @ B.T + m * np.eye(m)); X.sort_indices()
    L = problem.dense_symbolic(m)
    r = refmex.call("blkchol", 4, L, X, gl.default_pars_chol())
    lib._Z19emu_inject_timeoutsi(1)
    o = mex.blkchol(L, X, gl.default_pars_chol())
    assert relerr(o[1].ravel(), r[1].ravel()) < TOL and relerr(sp.csc_matrix(o[0]).data, sp.csc_matrix(r[0]).data) < TOL
    assert lib._Z25emu_take_injected_timeoutv() == 0                       # it was consumed by the first attempt
    # a second factorisation of the same pattern by the same (cached) plan: it stays on the launch-per-panel path, where nothing
    # waits for workgroups that are not resident yet -- an injected time-out is not even looked at
    lib._Z19emu_inject_timeoutsi(1)
    try:
        o = mex.blkchol(L, X, gl.default_pars_chol())
        assert relerr(o[1].ravel(), r[1].ravel()) < TOL
    finally:
        lib._Z19emu_inject_timeoutsi(0)


def test_set_chol_after_a_timeout_keeps_the_plan_on_the_panel_path(refmex):
    """A plan whose one-launch level timed out (front_disabled) and that is then given a symbolic factor again must not plan the
    inverse-behind-the-factor path: that once left the inverse arenas at zero and the solves returned 0 (round-3 advisor)."""
    from hipemu import build_emu
    from oracle import glue as gl
    from sedumi_amd import plan as pl, problem
    lib = ctypes.CDLL(build_emu.build())
    m = 330
    rng = np.random.default_rng(5)
    B = rng.standard_normal((m, m))
    X = sp.csc_matrix(B @ B.T + m * np.eye(m)); X.sort_indices()
    L = problem.dense_symbolic(m)
    pars = gl.default_pars_chol()
    LL, Ld, _, _ = refmex.call("blkchol", 4, L, X, pars)
    Lf = dict(L); Lf["L"] = LL
    b = rng.standard_normal((m, 1))
    yref = refmex.call("bwblkslv", 1, Lf, refmex.call("fwblkslv", 1, Lf, b) / Ld)
    P = pl.Plan()
    try:
        P.set_chol(L, X)
        P.upload("ada", X.data)
        lib._Z19emu_inject_timeoutsi(1)
        P.blkchol_wait(pars)                          # first attempt times out, the repeat runs on the launch-per-panel path
        assert lib._Z25emu_take_injected_timeoutv() == 0
        for again in range(2):
            if again:
                P.set_chol(L, X)                      # a new solve on the same plan
                P.upload("ada", X.data)
                P.blkchol_wait(pars)
            P.upload("rhs", b)
            P.ldlsolve()
            assert relerr(P.download("y"), yref.ravel()) < TOL
            assert relerr(P.download("d"), Ld.ravel()) < TOL
    finally:
        lib._Z19emu_inject_timeoutsi(0)
        P.close()


@pytest.mark.parametrize("ns,ms", [(4000, 4000), (1000, 1000), (900, 900), (960, 960), (1100, 1100), (1216, 1216), (700, 1500), (1300, 2100), (1024, 1030),
                                   (2050, 2050), (333, 2000), (64 * 9, 64 * 9 + 129), (64 * 9, 64 * 9 + 128), (2900, 3333)])
def test_update_schedule_gives_every_tile_every_panel_once_and_in_order(ns, ms):
    """tile_sched (sdm_chol.hip): the trailing updates of a big front on the launch-per-panel path -- eager tiles of the panel before, deferred
    macro tiles of whole groups of panels -- enumerated on the host for every panel launch (sdm_debug_tile_items) and replayed: every tile
    (I, J) of the front's lower triangle receives the panels 0 .. min(J, NP) - 1, each exactly once, in ascending order, never twice within a
    launch (two workgroups on one tile), and column q is complete but for panel q - 1 when launch q starts.  Fronts with rows beyond their
    columns (ms > ns) and partial last panels included."""
    import ctypes
    import os
    import sys
    sys.path.insert(0, os.path.join(helpers.ROOT, "tests", "hipemu"))
    import build_emu
    lib = ctypes.CDLL(build_emu.build())
    NB = 64
    NP, T = -(-ns // NB), -(-ms // NB)
    got = {(I, J): [] for J in range(T) for I in range(J, T)}
    buf = (ctypes.c_int * (5 * 8192))()
    deferred_seen = False
    for q in range(1, NP):
        # what launch q needs of its own column: everything but panel q - 1 (the row-solve workgroups and the diagonal block apply that one)
        for I in range(q, T):
            assert got[(I, q)] == list(range(q - 1)), (q, I, got[(I, q)])
        for I in range(q, T):
            got[(I, q)].append(q - 1)
        n = lib.sdm_debug_tile_items(ns, ms, q, buf, 8192)
        touched = set()
        if n < 0:                                                  # no row-solve workgroups: the eager tiles of the round-4 role, every tile but column q
            for J in range(q + 1, T):
                for I in range(J, T):
                    got[(I, J)].append(q - 1)
            continue
        assert n <= 8192
        for u in range(n):
            I0, J0, act, p0, npan = buf[5 * u:5 * u + 5]
            assert act != 0 and npan >= 1
            deferred_seen |= npan > 1
            for a in range(2):
                for b in range(2):
                    if act >> (2 * a + b) & 1:
                        I, J = q + I0 + a, q + J0 + b
                        assert q < J <= I < T and (I, J) not in touched, (q, u, I, J)
                        touched.add((I, J))
                        assert p0 + npan <= q                     # only final panels
                        got[(I, J)].extend(range(p0, p0 + npan))
    last_has_rows = ms > ns                                        # k_ldl_update: the last panel's update of the rows beyond the supernode
    for (I, J), panels in got.items():
        want = list(range(min(J, NP)))
        if J >= NP and last_has_rows:
            want = want[:-1]                                       # (applied by the stand-alone k_ldl_update, not by a panel launch)
        assert panels == want, ((I, J), panels, want)
    if T >= 12 and NP >= 6:
        assert deferred_seen

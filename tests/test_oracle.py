"""Pins the numpy restatement (oracle/restate.py) against the compiled reference MEX (oracle/_ref) and the
committed golden fixtures, and the host-only symbolic code (ordering / symbolic factorisation, integer,
bit-exact) against the reference.  No GPU needed."""
import numpy as np
import pytest
import scipy.sparse as sp

from helpers import TOL, relerr, spd_pattern, use_emu, load_golden, ref_scaling


@pytest.fixture(scope="module", autouse=True)
def _emu():
    use_emu()          # the symbolic entry points are plain host C++; the emulated build exports them too


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_restate_getada_and_blkchol_match_reference(glue, seed):
    from oracle import glue as gl, restate
    from sedumi_amd import problem
    P = problem.random_sdp(m=25 + 5 * seed, lp=4 + seed, q=(3, 5)[:1 + seed % 2], s=(4, 6, 3), seed=seed)
    S = glue.setup(P.At, P.K)
    d, ud = ref_scaling(P, seed)
    it = glue.iteration_ref(S, d, ud)
    ADA, absd = restate.getada(S["A"], P.K, d, it["DAt"]["q"], ud)
    assert relerr(ADA, it["ADA"]) < 1e-13 and relerr(absd, it["absd"].ravel()) < 1e-13
    L = S["L"]
    LL = sp.csc_matrix(L["L"])
    Lr, dr, skip, add = restate.blkchol_sparse(it["ADA"], LL.indptr, LL.indices, L["xsuper"].ravel().astype(int) - 1,
                                               L["perm"].ravel().astype(int) - 1, gl.default_pars_chol(), it["absd"])
    assert relerr(Lr, it["LL"]) < 1e-12 and relerr(dr, it["Ld"].ravel()) < 1e-12
    assert [k for k, _ in skip] == list(it["Lskip"].indices) and [k for k, _ in add] == list(it["Ladd"].indices)
    rhs = np.random.default_rng(seed).standard_normal(P.m)
    perm0 = L["perm"].ravel().astype(int) - 1
    y = restate.ldlsolve_dense(it["LL"].toarray(), it["Ld"].ravel(), perm0, rhs)
    assert relerr(y, glue.solve_ref(S, it, rhs).ravel()) < 1e-12


@pytest.mark.parametrize("kind,m", [("rand", 80), ("band", 60), ("arrow", 50), ("grid", 64)])
def test_restate_pivot_rule_matches_reference_decisions(refmex, glue, kind, m):
    """skip / add decisions (incl. the idamax quirk of maxabs, blkchol2.c:66-70) on badly scaled matrices."""
    from oracle import glue as gl, restate
    rng = np.random.default_rng(m)
    X0 = spd_pattern(kind, m, rng, 0.06)
    sc = 10.0 ** rng.uniform(-7, 3, X0.shape[0])
    X = sp.csc_matrix(sp.diags(sc) @ X0 @ sp.diags(sc)); X.sort_indices()
    L = glue.symbchol(X)
    LL = sp.csc_matrix(L["L"])
    for maxu in (5e5, 30.0, 2.0):
        pars = dict(gl.default_pars_chol()); pars["maxu"] = maxu
        r = refmex.call("blkchol", 4, L, X, pars)
        Lr, dr, skip, add = restate.blkchol_sparse(X, LL.indptr, LL.indices, L["xsuper"].ravel().astype(int) - 1,
                                                   L["perm"].ravel().astype(int) - 1, pars)
        assert [k for k, _ in skip] == list(r[2].indices) and [k for k, _ in add] == list(r[3].indices)
        assert relerr(dr, r[1].ravel()) < 1e-9


@pytest.mark.parametrize("name", ["arch0", "control07", "nb"])
def test_restate_against_golden_fixture(name):
    """The committed fixtures (reference MEX outputs on the reference's own example problems)."""
    from oracle import restate
    z, At, K = load_golden(name)
    m = At.shape[1]
    if m > 300:
        pytest.skip("dense numpy restatement is O(m^3): fixture checked through the HIP path instead")
    for tag in ("init", "rand"):
        d = {"l": z[f"{tag}_dl"], "det": z[f"{tag}_ddet"]}
        Q = sp.csc_matrix((z[f"{tag}_DAtq_data"], z[f"{tag}_DAtq_indices"], z[f"{tag}_DAtq_indptr"]), shape=tuple(z[f"{tag}_DAtq_shape"]))
        ADA, absd = restate.getada(At, K, d, Q, z[f"{tag}_udsqr"])
        assert relerr(absd, z[f"{tag}_absd"]) < 1e-12
        assert relerr(np.diag(ADA), z[f"{tag}_ADA_diag"]) < 1e-12
        assert relerr(ADA[z[f"{tag}_si"], z[f"{tag}_sj"]], z[f"{tag}_ADA_s"]) < 1e-12
        Lr, dr, skip, add = restate.blkchol_dense(ADA, np.arange(m), {"canceltol": 1e-12, "maxu": 5e5, "abstol": 1e-20}, absd)
        assert relerr(dr, z[f"{tag}_Ld"]) < 1e-9
        y = restate.ldlsolve_dense(Lr, dr, np.arange(m), z["rhs"])
        assert relerr(y, z[f"{tag}_y"]) < 1e-8


KINDS = [("rand", 0.02), ("rand", 0.1), ("rand", 0.5), ("band", 0), ("arrow", 0), ("blockdiag", 0), ("grid", 0), ("diag", 0)]


@pytest.mark.parametrize("kind,dens", KINDS)
@pytest.mark.parametrize("m", [1, 2, 3, 5, 17, 60, 150, 400])
def test_ordering_and_symbolic_are_bit_exact(refmex, kind, dens, m):
    """ordmmdmex / symfctmex / choltmpsiz / cholsplit: integer outputs identical to the reference."""
    from sedumi_amd import mex
    rng = np.random.default_rng(1000 * m + int(100 * dens))
    X = spd_pattern(kind, m, rng, dens)
    pr = refmex.call("ordmmdmex", 1, X)
    assert np.array_equal(mex.ordmmdmex(X), pr)
    Lr = refmex.call("symfctmex", 1, X, pr)
    Lo = mex.symfctmex(X, pr)
    assert np.array_equal(Lo["perm"], Lr["perm"]) and np.array_equal(Lo["xsuper"], Lr["xsuper"])
    assert np.array_equal(Lo["L"].indptr, Lr["L"].indptr) and np.array_equal(Lo["L"].indices, Lr["L"].indices)
    # a non-MMD input permutation as well
    p2 = rng.permutation(X.shape[0]).astype(np.float64) + 1
    L2r, L2o = refmex.call("symfctmex", 1, X, p2), mex.symfctmex(X, p2)
    assert np.array_equal(L2o["perm"], L2r["perm"]) and np.array_equal(L2o["xsuper"], L2r["xsuper"])
    assert np.array_equal(L2o["L"].indices, L2r["L"].indices)
    assert np.array_equal(mex.choltmpsiz(Lr), refmex.call("choltmpsiz", 1, Lr))
    for cachsz in (512.0, 1.0):
        assert np.array_equal(mex.cholsplit(Lr, cachsz), refmex.call("cholsplit", 1, Lr, cachsz))


def test_ordering_marker_wraparound_large(refmex):
    """n large enough for the maxint=32767 marker reset of GENMMD (ordmmd.c:87) to fire."""
    from sedumi_amd import mex
    rng = np.random.default_rng(7)
    X = spd_pattern("rand", 12000, rng, 0.0003)
    pr = refmex.call("ordmmdmex", 1, X)
    assert np.array_equal(mex.ordmmdmex(X), pr)
    Lr, Lo = refmex.call("symfctmex", 1, X, pr), mex.symfctmex(X, pr)
    assert np.array_equal(Lo["perm"], Lr["perm"]) and np.array_equal(Lo["L"].indices, Lr["L"].indices)


@pytest.mark.parametrize("N,m,dens,first,seed", [(60, 25, 0.15, 0, 1), (300, 120, 0.05, 40, 2), (90, 90, 0.3, 10, 3), (50, 1, 0.5, 0, 4),
                                                 (400, 333, 0.01, 100, 5), (30, 40, 0.0, 0, 6), (2000, 900, 0.004, 500, 7)])
def test_incorder_is_bit_exact(refmex, N, m, dens, first, seed):
    """[perm, dz] = incorder(At [, Ajc1, ifirst]) (incorder.c:140-209): the greedy order with its position-dependent
    tie-breaking and the per-column subscript order of dz, from the ordered-set formulation, against the reference's
    O(m^2) scan -- whole columns (symbcholden.m:50) and the rows from ifirst on (sedumi.m:378)."""
    from oracle.refmex import RawSparse
    from sedumi_amd import mex
    rng = np.random.default_rng(seed)
    At = sp.random(N, m, density=dens, random_state=rng, format="csc")
    At.data[:] = 1.0
    At.sort_indices()
    if dens > 0:                                   # many equal lengths: the tie-breaking decides
        At = sp.csc_matrix(sp.hstack([At, At[:, : m // 3]]))[:, :m] if m > 3 else At
        At.sort_indices()
    for use_first in (False, True):
        if use_first:
            Ajc1 = np.array([At.indptr[j] + np.searchsorted(At.indices[At.indptr[j]:At.indptr[j + 1]], first) for j in range(m)], dtype=np.float64)
            pr, dzr = refmex.call("incorder", 2, At, Ajc1.reshape(-1, 1), float(first + 1))
            po, dzo = mex.incorder(At, Ajc1, float(first + 1))
        else:
            pr, dzr = refmex.call("incorder", 2, At)
            po, dzo = mex.incorder(At)
        assert np.array_equal(po.ravel(), np.asarray(pr).ravel())
        assert np.array_equal(dzo.indptr, dzr.indptr) and np.array_equal(dzo.indices, dzr.indices)

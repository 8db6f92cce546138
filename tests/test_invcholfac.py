"""y = invcholfac(u, K, perm) -- the `udsqr` argument of getada3 (sedumi.m:452).  Kernel logic on the CPU
emulator against the compiled reference (oracle/_ref/invcholfac.so) and against a numpy restatement of
invcholfac.c:59-168; the GPU run of the same checks is in test_gpu_parity.py."""
import numpy as np
import pytest

from helpers import TOL, relerr, use_emu


def scaling_factor_case(K, seed=0, garbage_lower=True):
    """u = per-block upper-triangular factors as the scaling update leaves them (the strict lower triangle holds
    whatever was there: the gateway must ignore it), perm = a pivoting order per block (1-based, local)."""
    rng = np.random.default_rng(seed)
    s = K["s"].ravel().astype(int)
    r = int(K["rsdpN"])
    us, perms = [], []
    for k, n in enumerate(s):
        planes = 1 if k < r else 2
        U = rng.standard_normal((planes, n, n))
        if not garbage_lower:
            U = np.stack([np.triu(p) for p in U])
        U[0][np.arange(n), np.arange(n)] = 1.0 + rng.random(n)
        if planes == 2:
            U[1][np.arange(n), np.arange(n)] = 0.0          # prpiutmulx assumes Im diag(U) == 0
        us.append(np.concatenate([p.ravel(order="F") for p in U]))
        perms.append(rng.permutation(n) + 1.0)
    return np.concatenate(us), np.concatenate(perms)


def restate(u, K, perm):
    """numpy restatement of invcholfac.c:59-168 (utmulx / prpiutmulx, triu2sym / triu2herm, invmatperm)."""
    s = K["s"].ravel().astype(int)
    r = int(K["rsdpN"])
    out, o, po = [], 0, 0
    for k, n in enumerate(s):
        Ur = np.triu(u[o:o + n * n].reshape(n, n, order="F")); o += n * n
        if k < r:
            Z = Ur.T @ Ur
            planes = [Z]
        else:
            Ui = np.triu(u[o:o + n * n].reshape(n, n, order="F"), 1); o += n * n
            U = Ur + 1j * Ui
            Z = U.conj().T @ U
            planes = [Z.real, Z.imag - np.diag(np.diag(Z.imag))]
        p = (perm[po:po + n].astype(int) - 1) if perm is not None else np.arange(n)
        po += n
        for Zp in planes:
            Y = np.zeros((n, n)); Y[np.ix_(p, p)] = Zp
            out.append(Y.ravel(order="F"))
    return np.concatenate(out)


CASES = [dict(s=[5]), dict(s=[70, 35]), dict(s=[1, 2, 64, 65]), dict(s=[130]), dict(s=[6], hs=[4, 9]), dict(s=[], hs=[70]),
         dict(s=[200, 3], hs=[66])]


@pytest.fixture(scope="module", autouse=True)
def _emu():
    use_emu()


@pytest.mark.parametrize("case", range(len(CASES)))
def test_invcholfac_matches_reference_and_restatement(refmex, case):
    from sedumi_amd import mex, problem
    kw = CASES[case]
    K = problem.make_K(1, [], kw.get("s", []), hs=kw.get("hs", ()))
    u, perm = scaling_factor_case(K, seed=case)
    for pm in (perm, None):
        args = (u.reshape(-1, 1), K) + ((pm.reshape(-1, 1),) if pm is not None else ())
        yr = refmex.call("invcholfac", 1, *args)
        yo = mex.invcholfac(u, K, pm)
        assert relerr(yo, yr) < TOL
        assert relerr(restate(u, K, pm), yr.ravel()) < 1e-12                 # the restatement is pinned by the reference too


def test_invcholfac_feeds_getada3(refmex, glue):
    """invcholfac -> getada3 chained on the device (plan buffers "u" -> "udsqr" -> "ada") against the reference chain."""
    from sedumi_amd import problem
    from sedumi_amd.plan import Plan
    P = problem.random_sdp(m=24, lp=3, q=(), s=(7, 4), hs=(5,), seed=5)
    S = glue.setup(P.At, P.K)
    u, perm = scaling_factor_case(P.K, seed=8)
    ud = refmex.call("invcholfac", 1, u.reshape(-1, 1), P.K, perm.reshape(-1, 1))
    d = {"l": np.ones(int(P.K["l"])), "det": np.ones(0)}
    it = glue.iteration_ref(S, dict(d, q1=np.ones(0), q2=np.zeros(0)), ud.ravel())
    plan = Plan(0)
    plan.set_chol(S["L"], S["ADA"]); plan.set_ada(P.At, P.Ablkjc, P.K, problem.lorentz_pattern(P))
    plan.upload("dl", d["l"]); plan.upload("ddet", d["det"]); plan.upload("u", u)
    plan.invcholfac(perm); plan.getada()
    assert relerr(plan.download("udsqr", ud.size), ud.ravel()) < TOL
    assert relerr(plan.download("ada"), it["ADA"].data) < TOL
    plan.close()


def test_invcholfac_bad_inputs():
    from sedumi_amd import mex, problem
    from sedumi_amd.capi import SdmError
    K = problem.make_K(1, [], [4])
    with pytest.raises(SdmError, match="u size mismatch"):
        mex.invcholfac(np.ones(15), K)
    with pytest.raises(SdmError):
        mex.invcholfac(np.ones(16), K, np.array([1.0, 1.0, 2.0, 3.0]))     # not a permutation


@pytest.mark.parametrize("kw", [dict(m=35, lp=8, q=(4, 3, 5), s=()), dict(m=24, lp=3, q=(3,) * 40, s=(4,), dens=0.6),
                                dict(m=30, lp=0, q=(6, 2), s=(5,), dens=0.2)])
def test_datq_on_device_matches_getDAtm(glue, kw):
    """SURVEY 8f N3 (the Lorentz half of getDAtm.m:39-44): DAt.q = diag(d.q1) * A(trace rows,:) + ddot(d.q2, A, ...) formed
    on the device from the resident d.q1 / d.q2, against the reference's extractA + ddot MEX chain, then through
    getada2's resident counterpart."""
    import scipy.sparse as sp
    from sedumi_amd import problem
    from sedumi_amd.plan import Plan
    from helpers import ref_scaling
    P = problem.random_sdp(seed=17, **kw)
    S = glue.setup(P.At, P.K)
    d, ud = ref_scaling(P, 3)
    DAt = glue.getDAtm(S, d)
    Qpat = sp.csc_matrix(problem.lorentz_pattern(P))
    plan = Plan(0)
    plan.set_chol(S["L"], S["ADA"]); plan.set_ada(P.At, P.Ablkjc, P.K, Qpat)
    plan.upload("dl", d["l"]); plan.upload("ddet", d["det"]); plan.upload("udsqr", ud)
    plan.upload("q1", d["q1"]); plan.upload("q2", d["q2"])
    plan.getdatq()
    cols = np.repeat(np.arange(P.m), np.diff(Qpat.indptr))
    want = np.asarray(sp.csc_matrix(DAt["q"])[Qpat.indices, cols]).ravel()
    assert relerr(plan.download("qpr", Qpat.nnz), want) < TOL
    plan.getada()
    it = glue.iteration_ref(S, d, ud)
    assert relerr(plan.download("ada"), it["ADA"].data) < TOL
    plan.close()


def test_next_row_calls_are_noops_without_their_cones(glue):
    """invcholfac without PSD blocks and getdatq without Lorentz cones leave the plan untouched (the MATLAB calls return
    empty arrays there: invcholfac.c:95-96 with lenud = 0, getDAtm.m:41 with nq = 0)."""
    from sedumi_amd import mex, problem
    from sedumi_amd.plan import Plan
    P = problem.random_sdp(m=12, lp=9, q=(), s=(), seed=2)             # LP only
    S = glue.setup(P.At, P.K)
    plan = Plan(0)
    plan.set_chol(S["L"], S["ADA"]); plan.set_ada(P.At, P.Ablkjc, P.K, problem.lorentz_pattern(P))
    plan.upload("dl", np.ones(int(P.K["l"]))); plan.upload("ddet", np.zeros(0))
    plan.invcholfac(None); plan.getdatq(); plan.getada()
    it = glue.iteration_ref(S, {"l": np.ones(int(P.K["l"])), "det": np.zeros(0), "q1": np.zeros(0), "q2": np.zeros(0)}, np.zeros(0))
    assert relerr(plan.download("ada"), it["ADA"].data) < TOL
    plan.close()
    assert mex.invcholfac(np.zeros(0), P.K).size == 0


def test_bench_workloads_build():
    """bench.py's workload table (BASELINE.json configs[1..3] shapes) builds its inputs on the host."""
    import importlib.util, os
    from helpers import ROOT
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    for name, m in (("control07", 666), ("control07_like", 666), ("nb", 123), ("maxcut300", 300), ("blockdiag:4:10:6", 24)):
        P, L, ADA, Q, d, ud, rhs, qpr, note = bench.build_workload(name, seed=0)
        assert P.m == m and ADA.shape == (m, m) and rhs.size == m
        assert ud.size == int(np.sum(P.K["s"].ravel() ** 2)) and np.asarray(d["l"]).size == int(P.K["l"])
        assert (qpr is None) == (Q.nnz == 0)
    assert "control07.mat" in bench.build_workload("control07", 0)[8]

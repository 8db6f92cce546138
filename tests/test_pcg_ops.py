"""Amul / vecsym / psdscale on the resident plan against the reference (vecsym.c compiled; Amul.m and
psdscale.m restated line by line in numpy below -- they are MATLAB files).  CPU run = fiber emulator; the GPU run of
the same checks is in test_gpu_parity.py."""
import numpy as np
import pytest
import scipy.sparse as sp

from helpers import TOL, relerr, use_emu


@pytest.fixture(scope="module", autouse=True)
def _emu():
    use_emu()


def psdscale_restated(u, perm, x, K, transp):
    """psdscale.m:46-119 (ud = struct with u and perm; perm None = empty)."""
    Ks = K["s"].ravel().astype(int)
    nr = int(K["rsdpN"])
    out, ui, xi, pi_ = [], 0, 0, 0
    xpsd = x[x.size - int(np.sum(Ks ** 2) + np.sum(Ks[nr:] ** 2)):]
    for i, ki in enumerate(Ks):
        qi = ki * ki
        TT = u[ui:ui + qi].astype(complex); ui += qi
        if i >= nr:
            TT = TT + 1j * u[ui:ui + qi]; ui += qi
        TT = TT.reshape(ki, ki, order="F")
        TT = np.triu(TT) if transp else np.tril(TT)                 # psdscale.m:85-89
        XX = xpsd[xi:xi + qi].astype(complex); xi += qi
        if i >= nr:
            XX = XX + 1j * xpsd[xi:xi + qi]; xi += qi
        XX = XX.reshape(ki, ki, order="F")
        if perm is not None and not transp:                         # prep   (psdscale.m:96-101)
            PP = perm[pi_:pi_ + ki].astype(int) - 1; pi_ += ki
            XX = XX[np.ix_(PP, PP)]
        XX = TT.conj().T @ XX @ TT                                  # psdscale.m:105
        if perm is not None and transp:                             # postp  (psdscale.m:106-111)
            PP = perm[pi_:pi_ + ki].astype(int) - 1; pi_ += ki
            Y = np.zeros_like(XX); Y[np.ix_(PP, PP)] = XX; XX = Y
        out.append(XX.real.ravel(order="F"))
        if i >= nr:
            Z = XX.imag.copy(); Z[np.arange(ki), np.arange(ki)] = 0.0   # psdscale.m:116
            out.append(Z.ravel(order="F"))
    return np.concatenate(out) if out else np.zeros(0)


CASES = [dict(m=20, lp=4, q=(3,), s=(5, 7)), dict(m=30, lp=0, q=(), s=(70, 9), hs=(6,)), dict(m=12, lp=3, q=(4, 3), s=(), hs=(66, 3)),
         dict(m=15, lp=2, q=(), s=(130,))]


def check_pcg_ops(refmex, kw, seed=0):
    from sedumi_amd import problem
    from sedumi_amd.plan import Plan
    from test_invcholfac import scaling_factor_case
    rng = np.random.default_rng(seed)
    P = problem.random_sdp(seed=seed + 40, dens=0.3, **kw)
    N, m = P.At.shape
    plan = Plan(0)
    plan.set_chol(problem.dense_symbolic(m), problem.dense_pattern(m))
    plan.set_ada(P.At, P.Ablkjc, P.K, problem.lorentz_pattern(P))
    # dense columns of Amul.m:50-56: two cone variables taken out of At
    cols = np.array([2.0, 5.0]) if N > 6 else np.zeros(0)
    At = sp.csc_matrix(P.At)
    denseA = sp.csc_matrix(At[(cols - 1).astype(int), :].T) if cols.size else None
    plan.pcg_init(cols, denseA)
    x = rng.standard_normal(N); y = rng.standard_normal(m)
    plan.upload("xN", x); plan.amul(0)
    want = np.asarray(x @ At).ravel() + (np.asarray(denseA @ x[(cols - 1).astype(int)]).ravel() if cols.size else 0.0)   # Amul.m:46,52
    assert relerr(plan.download("rhs"), want) < TOL
    plan.upload("y", y); plan.amul(1)
    want = np.asarray(At @ y).ravel()                                # Amul.m:48
    if cols.size:
        want[(cols - 1).astype(int)] = np.asarray(denseA.T @ y).ravel()   # Amul.m:54
    assert relerr(plan.download("xN", N), want) < TOL
    # vecsym against the compiled reference
    xs = rng.standard_normal(N)
    plan.upload("xN", xs); plan.vecsym()
    assert relerr(plan.download("xN", N), refmex.call("vecsym", 1, xs.reshape(-1, 1), P.K).ravel()) < 1e-15
    # psdscale against the restatement of psdscale.m, all four (transp, perm) combinations
    lenud = int(np.sum(P.K["s"].ravel()[:int(P.K["rsdpN"])] ** 2) + 2 * np.sum(P.K["s"].ravel()[int(P.K["rsdpN"]):] ** 2))
    if lenud:
        u, perm = scaling_factor_case(P.K, seed=seed + 3)
        plan.upload("u", u); plan.invcholfac(perm)                   # leaves the pivot order resident
        xv = plan.download("xN", N)
        for transp in (0, 1):
            for use_perm in (False, True):
                plan.psdscale(transp, use_perm)
                want = psdscale_restated(u, perm if use_perm else None, xv, P.K, transp)
                assert relerr(plan.download("psd", lenud), want) < TOL, (transp, use_perm)
    plan.close()


@pytest.mark.parametrize("case", range(len(CASES)))
def test_pcg_operators(refmex, case):
    check_pcg_ops(refmex, CASES[case], seed=case)

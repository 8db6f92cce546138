"""SURVEY.md section 8f row N4 as a PRODUCT: sedumi_amd.driver -- SeDuMi's loop without MATLAB, the cone algebra outside the hot path on numpy /
LAPACK (sedumi_amd/driver/conemex.py) instead of the reference's MEX.

  * every routine of conemex.py against the COMPILED REFERENCE on the calls that whole solves of the reference's examples make (recorded here by
    running the loop with the oracle as its MEX host): values where the routine's output is defined by its input (ddot, qblkmul, vecsym, quadadd,
    iswnbr, sqrtinv, partitA, extractA, findblks), invariants where the reference hands an opaque compact form from one MEX to another (qrK's
    frame, urotorder's rotations: this module stores the unitary matrices explicitly; psdframeit / psdinvjmul are then compared with the
    reference's results on the native frame brought to the reference's sign convention);
  * whole solves with the product's defaults (native cone algebra; hot path = this library): the optimal values of examples/test_sedumi.m:22-28
    to its 1e-6 and the iteration count of the same-host run with the reference everywhere (emulator: nb, quantum; GPU: + arch0, control07 on the
    resident plan);
  * nothing under sedumi_amd/ imports the oracle."""
import collections
import os
import re

import numpy as np
import pytest
import scipy.sparse as sp

import helpers
from helpers import ROOT
from oracle import refmex

pytestmark = pytest.mark.skipif(not refmex.available(), reason="oracle/_ref is not built")

CONE = {"ddot", "qblkmul", "psdframeit", "psdinvjmul", "vecsym", "qrK", "quadadd", "iswnbr", "urotorder", "givensrot", "sqrtinv", "extractA",
        "findblks", "sortnnz", "partitA"}


class _Recorder:
    """a MEX host that passes every call on to the compiled reference and keeps (nlhs, args, outputs) of the cone-algebra calls"""

    def __init__(self, ref, cap=40):
        self.ref, self.log, self.cap = ref, collections.defaultdict(list), cap

    def __getattr__(self, k):
        return getattr(self.ref, k)

    def call(self, name, nlhs, *args):
        out = self.ref.call(name, nlhs, *args)
        if name in CONE and len(self.log[name]) < self.cap:
            self.log[name].append((nlhs, args, out))
        return out


_LOGS = {}


def recorded(name):
    if name not in _LOGS:
        import test_driver as td
        from driver import sedumi_loop as sl
        from oracle import glue as gl
        G = gl.Glue()
        rec = _Recorder(G.ref)
        G.ref = rec
        At, K, g = td.problem(name)
        sl.Sedumi(At, g["b"], g["c"], K, G=G, internal=True).solve()
        _LOGS[name] = rec.log
    return _LOGS[name]


def _rel(a, b):
    a = np.asarray(a.todense() if sp.issparse(a) else a, dtype=float)
    b = np.asarray(b.todense() if sp.issparse(b) else b, dtype=float)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)) if a.size else 0.0


@pytest.mark.parametrize("name", ["nb", "quantum"])          # Lorentz cones; two Hermitian PSD blocks
def test_native_cone_routines_against_the_reference_on_the_calls_of_a_solve(name):
    from sedumi_amd.driver import conemex as cm
    log = recorded(name)
    seen = set()
    for fn in ("ddot", "qblkmul", "vecsym", "partitA", "extractA", "findblks", "sqrtinv"):
        for nlhs, args, out in log.get(fn, []):
            got = getattr(cm, fn)(*args)
            assert _rel(got, out) < 1e-13, fn
            if sp.issparse(out):                                # the pattern too (explicit zeros included)
                g, o = sp.csc_matrix(got), sp.csc_matrix(out)
                g.sort_indices(); o.sort_indices()
                assert np.array_equal(g.indptr, o.indptr) and np.array_equal(g.indices, o.indices), fn
            seen.add(fn)
    for nlhs, args, out in log.get("quadadd", []):
        hi, lo = cm.quadadd(*args)
        assert np.array_equal(hi, out[0]) and np.array_equal(lo, out[1])
        seen.add("quadadd")
    for nlhs, args, out in log.get("iswnbr", []):
        d, h, a = cm.iswnbr(*args)
        ref = [float(np.asarray(o).ravel()[0]) for o in out]
        assert abs(h - ref[1]) <= 1e-13 * abs(ref[1]) and abs(a - ref[2]) <= 1e-12 * max(abs(ref[2]), 1e-3)
        assert ref[0] > 1e99 or abs(d - ref[0]) <= 1e-9 * max(abs(ref[0]), 1e-3)      # (delta = sqrt of a difference of nearly equal numbers)
        seen.add("iswnbr")
    for nlhs, args, out in log.get("sortnnz", []):
        # (the reference's comparator returns `char` through a cast function pointer: its order is not even sorted -- sortnnz.c:63-70,
        # sdmauxCmp.c:54; the native one is the stable sort the routine's header describes)
        A = sp.csc_matrix(args[0])
        lo = np.asarray(args[1]).ravel() if np.size(args[1]) else A.indptr[:-1]
        hi = np.asarray(args[2]).ravel() if np.size(args[2]) else A.indptr[1:]
        p = cm.sortnnz(*args).ravel().astype(int) - 1
        assert np.array_equal(np.sort(p), np.arange(A.shape[1])) and np.all(np.diff((hi - lo)[p]) >= 0)
        seen.add("sortnnz")
    # ---- the PSD chain: frames and rotations are this module's own explicit matrices
    frames = {}
    for nlhs, args, out in log.get("qrK", []):
        x, K = args
        cK = cm.ConeK(K)
        q, r = cm.qrK(x, K, nlhs=2)
        for X, Q, R in zip(cK.blocks(x), cK.blocks(q), cK.blocks(r)):
            assert np.abs(Q @ R - X).max() <= 1e-13 * max(np.abs(X).max(), 1e-300) and np.abs(Q.conj().T @ Q - np.eye(Q.shape[0])).max() < 1e-13
            assert np.abs(np.tril(R, -1)).max() == 0.0 and np.real(np.diagonal(R)).min() >= 0.0
        qref = out[0] if nlhs > 1 else out
        if nlhs > 1:
            Qs = []
            for Q, R, Rr in zip(cK.blocks(q), cK.blocks(r), cK.blocks(out[1])):
                assert _rel(np.abs(np.triu(R)), np.abs(np.triu(Rr))) < 1e-12                       # R is unique up to the phases of its rows
                S = np.sign(np.real(np.diagonal(Rr))); S[S == 0] = 1
                Qs.append(Q * S)                                                                    # the reference's sign convention, for the comparisons below
            frames[np.asarray(qref).tobytes()] = cK.pack(Qs)
        else:
            frames[np.asarray(qref).tobytes()] = q
        seen.add("qrK")
    for fn in ("psdframeit", "psdinvjmul"):
        for nlhs, args, out in log.get(fn, []):
            key = np.asarray(args[1]).tobytes()
            if key in frames:
                a = list(args); a[1] = frames[key]
                assert _rel(getattr(cm, fn)(*a), out) < 1e-9, fn
                seen.add(fn)
    for nlhs, args, out in log.get("urotorder", []):
        u, K, maxu = args[0], args[1], float(np.asarray(args[2]).ravel()[0])
        pin = np.asarray(args[3]).ravel() if len(args) > 3 and np.size(args[3]) else None
        uo, perm, gjc, g = cm.urotorder(*args)
        cK = cm.ConeK(K)
        y = cm.givensrot(gjc, g, cK.pack([np.triu(M) for M in cK.blocks(u)]), K)
        o = 0
        for M, U, Y in zip(cK.blocks(u), cK.blocks(uo), cK.blocks(y)):
            n = M.shape[0]
            p = perm.ravel()[o:o + n]
            pk = pin[o:o + n] if pin is not None else np.arange(1, n + 1.0)
            pp = np.array([int(np.flatnonzero(pk == v)[0]) for v in p])                             # perm_out = perm_in(pp)
            assert np.array_equal(np.sort(pp), np.arange(n))
            scale = max(np.abs(M).max(), 1e-300)
            assert np.abs(Y[:, pp] - np.triu(U)).max() <= 1e-13 * scale                              # (G triu(u))(:, pp) is the new upper-triangular factor
            assert np.abs(np.tril(U, -1) - np.triu(U, 1).conj().T).max() == 0.0                     # lower triangle mirrored, as the reference stores it
            a2 = np.abs(np.triu(U)) ** 2
            assert all(a2[r, r + 1:].max() <= maxu ** 2 * a2[r, r] * (1 + 1e-12) + 1e-300 for r in range(n - 1))   # stable: urotorder.c:110-122
            o += n
        seen.add("urotorder"); seen.add("givensrot")
    want = {"nb": {"ddot", "qblkmul", "vecsym", "partitA", "extractA", "findblks", "sortnnz", "iswnbr"},
            "quantum": {"vecsym", "partitA", "extractA", "findblks", "sortnnz", "iswnbr", "qrK", "psdframeit", "psdinvjmul", "sqrtinv", "urotorder", "givensrot"}}[name]
    assert want <= seen, want - seen


def _check_against_reference_run(name, r, margin=0):
    import test_driver as td
    td.check_objectives(name, r)
    ref = td.reference_run(name)
    assert abs(r["iter"] - ref["iter"]) <= margin, (r["iter"], ref["iter"])
    assert abs(r["cx"] - ref["cx"]) <= 1e-6 * abs(ref["cx"]) and abs(r["by"] - ref["by"]) <= 1e-6 * abs(ref["by"])


@pytest.mark.parametrize("name", ["nb", "quantum"])
def test_product_driver_solves_the_examples_on_the_emulated_library(name):
    """sedumi_amd.driver with its own cone algebra (no oracle anywhere in the run) and the library MEX call by MEX call."""
    import test_driver as td
    from sedumi_amd.driver import loop as lp
    helpers.use_emu()
    At, K, g = td.problem(name)
    S = lp.Sedumi(At, g["b"], g["c"], K, hot=lp.HipHot(), internal=True)
    assert type(S.ref).__name__ == "NativeMex" and type(S.G).__module__ == "sedumi_amd.driver.glue"
    _check_against_reference_run(name, S.solve())


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["nb", "quantum", "arch0", "control07"])
def test_product_driver_solves_the_examples_on_the_gpu(name):
    """The product's defaults: native cone algebra, the hot path on the resident plan (ADA', factor, solves, invcholfac, Amul / vecsym / psdscale
    in HBM).  One iteration of margin for arch0 (its last iteration is decided by rounding, tests/test_driver.py: ITER_MARGIN) and for control07 (both runs
    end with STOP -1 after 50 - 99 CG steps in their last iterations, at the accuracy limit of the problem: 41 iterations native against 40 with the
    reference's cone MEX, objectives -20.6250877 / -20.6250884: profiles/r08p_native_driver.jsonl)."""
    import test_driver as td
    from sedumi_amd.driver import loop as lp
    helpers.use_hip()
    At, K, g = td.problem(name)
    S = lp.Sedumi(At, g["b"], g["c"], K, internal=True)
    assert type(S.ref).__name__ == "NativeMex" and type(S.hot).__name__ == "PlanHot"
    _check_against_reference_run(name, S.solve(), margin=1 if name in ("arch0", "control07") else 0)


def test_user_level_entry_point_on_a_small_sdp():
    """solve(At, b, c, K) from user-level data (pretransfo included): a random feasible SDP + LP + Lorentz problem, checked by its own optimality
    conditions (primal / dual feasibility and a closed gap), on the emulated library."""
    from sedumi_amd.driver import solve
    from sedumi_amd.driver import loop as lp
    helpers.use_emu()
    rng = np.random.default_rng(5)
    Kl, q, s, m = 4, 5, 6, 7
    N = Kl + q + s * s
    X0 = np.concatenate((1 + rng.random(Kl), [3.0], 0.3 * rng.standard_normal(q - 1), (lambda B: (B @ B.T + s * np.eye(s)).ravel())(rng.standard_normal((s, s)))))
    At = rng.standard_normal((N, m))
    for j in range(m):                                          # symmetric PSD parts
        B = At[Kl + q:, j].reshape(s, s); At[Kl + q:, j] = ((B + B.T) / 2).ravel()
    b = At.T @ X0
    Z0 = np.concatenate((1 + rng.random(Kl), [3.0], 0.3 * rng.standard_normal(q - 1), (lambda B: (B @ B.T + s * np.eye(s)).ravel())(rng.standard_normal((s, s)))))
    c = Z0 + At @ rng.standard_normal(m)
    r = solve(sp.csc_matrix(At), b, c, {"l": Kl, "q": [q], "s": [s]}, hot=lp.HipHot())
    assert r["STOP"] in (1, -1) or r["iter"] > 3
    assert abs(r["cx"] - r["by"]) <= 1e-6 * (1 + abs(r["cx"]))


def _dense_lp(m, n, nd, seed):
    """a feasible, bounded LP whose first nd variables appear in every constraint (dense columns of A: getdense.m:38-75 takes them out of ADA')"""
    rng = np.random.default_rng(seed)
    A = sp.random(m, n, density=0.03, random_state=seed, format="csr").toarray()
    A[:, :nd] = rng.standard_normal((m, nd))
    A[np.arange(m), nd + np.arange(m)] += 1.0                                   # full row rank
    x0 = 1 + rng.random(n)
    b = A @ x0
    c = (1 + rng.random(n)) + A.T @ rng.standard_normal(m)
    return A, b, c


def test_dense_columns_through_the_product_form_on_the_emulated_library():
    """sedumi.m:356-364, symbcholden.m:43-55, deninfac.m:58-76, wrapPcg.m:56-59 in the product driver: an LP with three dense columns is solved
    through symbfwblk / incorder / finsymbden, the sparse forward solve of the dense columns, dpr1fact and fwdpr1 / bwdpr1 around every solve
    (SURVEY.md 8a rows a18, a20 - a22 in a whole solve) -- against HiGHS, and against the same loop with the compiled reference everywhere."""
    from scipy.optimize import linprog
    from driver import sedumi_loop as sl
    from sedumi_amd.driver import loop as lp
    helpers.use_emu()
    A, b, c = _dense_lp(70, 400, 3, 3)
    want = linprog(c, A_eq=A, b_eq=b, bounds=(0, None), method="highs")
    assert want.status == 0
    At, K = sp.csc_matrix(A.T), {"l": A.shape[1]}
    S = lp.Sedumi(At, b, c, K, hot=lp.HipHot())
    assert S.den is not None and np.array_equal(S.den["rows"], [1, 2, 3])         # (row 0 of the internal At is x0, pretransfo.m)
    r = S.solve()
    assert abs(r["cx"] - want.fun) <= 1e-6 * abs(want.fun) and abs(r["by"] - want.fun) <= 1e-6 * abs(want.fun)
    ref = sl.Sedumi(At, b, c, K).solve()                                          # oracle MEX host, reference hot path incl. its dpr1fact / fwdpr1 / bwdpr1
    assert ref["hot"] == "reference" and abs(ref["cx"] - want.fun) <= 1e-6 * abs(want.fun)
    assert abs(r["iter"] - ref["iter"]) <= 1


@pytest.mark.gpu
def test_dense_columns_on_the_resident_plan_on_the_gpu():
    """The same with the product's default hot path: the resident dense-column unit (sdm_plan_set_dense / sdm_plan_deninfac: the forward solve of the dense columns and
    dpr1fact on the device) and the product-form kernels around every solve."""
    from scipy.optimize import linprog
    from sedumi_amd.driver import loop as lp
    helpers.use_hip()
    A, b, c = _dense_lp(150, 900, 5, 7)
    want = linprog(c, A_eq=A, b_eq=b, bounds=(0, None), method="highs")
    assert want.status == 0
    S = lp.Sedumi(sp.csc_matrix(A.T), b, c, {"l": A.shape[1]})
    assert S.den is not None and S.den["rows"].size == 5 and type(S.hot).__name__ == "PlanHot"
    r = S.solve()
    assert abs(r["cx"] - want.fun) <= 1e-6 * abs(want.fun) and abs(r["by"] - want.fun) <= 1e-6 * abs(want.fun)


def test_the_product_does_not_import_the_oracle():
    pat = re.compile(r"^\s*(from\s+oracle\b|import\s+oracle\b|from\s+\.\.?oracle\b)", re.M)
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "sedumi_amd")):
        for f in files:
            if f.endswith(".py") and pat.search(open(os.path.join(dirpath, f)).read()):
                bad.append(os.path.join(dirpath, f))
    assert not bad, bad


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["trto3", "OH_2Pi"])
def test_product_driver_solves_the_larger_examples_on_the_gpu(name):
    """trto3 (one PSD block of order 321, 544 equations) and OH_2Pi_STO-6GN9r12g1T2 (22 PSD blocks, 948 equations): the reference hot path needs six
    CPU minutes for each, so the comparison is with the committed fixture of that run (tests/golden/driver_*.npz: iteration count, optimal values):
    the optimal values to examples/test_sedumi.m's 1e-6, the iteration count within two (trto3: 62 against 60 -- its last iterations are at the
    accuracy limit; OH_2Pi: 20 = 20).  26 s and 15 s on the MI355X."""
    import test_driver as td
    from sedumi_amd.driver import loop as lp
    helpers.use_hip()
    At, K, g = td.problem(name)
    r = lp.Sedumi(At, g["b"], g["c"], K, internal=True).solve()
    td.check_objectives(name, r)
    assert abs(r["iter"] - int(g["iter"])) <= 2
    assert abs(r["cx"] - float(g["cx"])) <= 1e-6 * abs(float(g["cx"])) and abs(r["by"] - float(g["by"])) <= 1e-6 * abs(float(g["by"]))

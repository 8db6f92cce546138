"""Shared helpers of the parity tests (CPU-emulated and GPU runs use the same checks)."""
import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

TOL = 1e-10          # north_star: factor / solves within 1e-10 relative Frobenius of the reference MEX path


def relerr(a, b):
    a = np.asarray(a.todense() if sp.issparse(a) else a, dtype=np.float64)
    b = np.asarray(b.todense() if sp.issparse(b) else b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def use_emu():
    """Bind sedumi_amd to the fiber-emulated build of the kernel sources (tests only)."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))
    import build_emu
    from sedumi_amd import capi
    capi.use_library(build_emu.build())
    assert capi.backend() == "emu"


def use_hip():
    from sedumi_amd import capi
    capi.use_library(None)
    assert capi.backend() == "hip-gfx950", "GPU tests must run on the hipcc-built library"
    assert capi.device_count() >= 1, "no HIP device visible"


def ref_scaling(P, seed, identity=False):
    """d (with the Lorentz fields getDAtm needs) and udsqr for problem P."""
    from sedumi_amd import problem
    d, ud = problem.spd_scaling(P.K, seed=seed, identity=identity)
    rng = np.random.default_rng(seed + 100)
    nq = P.K["q"].size
    d["q1"] = 1.0 + rng.random(nq)
    d["q2"] = 0.3 * rng.standard_normal(int(P.K["mainblks"].ravel()[2] - P.K["mainblks"].ravel()[1]))
    return d, ud


def check_iteration(G, P, seed=0, identity=False, tol=TOL, pars=None):
    """One iteration unit through the MEX-equivalent calls (sedumi.m:450-458, wrapPcg.m:56-59), stage by stage
    against the compiled reference.  Returns the dict of relative errors."""
    from oracle import glue as gl
    from sedumi_amd import mex
    S = G.setup(P.At, P.K)
    assert np.array_equal(S["Ablkjc"], P.Ablkjc)
    d, ud = ref_scaling(P, seed, identity)
    pars = pars or gl.default_pars_chol()
    it = G.iteration_ref(S, d, ud, dict(pars))
    K = P.K
    errs = {}
    A1 = mex.getada1(S["ADA"], S["A"], S["Ablkjc"][:, 2], S["Aord"]["lqperm"], d, K["qblkstart"])
    errs["getada1"] = relerr(A1, it["ADA1"])
    A2 = mex.getada2(it["ADA1"], it["DAt"], S["Aord"], K)
    errs["getada2"] = relerr(A2, it["ADA2"])
    A3, absd = mex.getada3(it["ADA2"], S["A"], S["Ablkjc"][:, 2], S["Aord"], ud, K)
    errs["getada3"] = relerr(A3, it["ADA"])
    if K["s"].size:
        errs["absd"] = relerr(absd, it["absd"])
    else:
        # no PSD blocks: getada3.c:549-552 documents absd = diag(ADA) (cpspdiag).  The compiled reference returns
        # garbage there because cpspdiag bsearches with a comparator that returns `char` through an `int (*)()`
        # pointer (sdmauxCmp.c:48, blksdp.h:121) -- undefined behaviour; sedumi.m never takes this branch
        # (sum(K.s)==0 goes through getada.m, sedumi.m:446-448).  We pin the documented semantics.
        errs["absd"] = relerr(absd.ravel(), it["ADA"].diagonal())
    if not K["s"].size:
        # feed both factorizations the documented absd (see above); with the reference's garbage absd (zeros) the
        # noise-level pivots of a rank-deficient ADA' would be accepted or skipped by rounding luck
        absd_in = it["ADA"].diagonal().reshape(-1, 1)
        it["LL"], it["Ld"], it["Lskip"], it["Ladd"] = G.ref.call("blkchol", 4, S["L"], it["ADA"], pars, absd_in)
    else:
        absd_in = it["absd"]
    LL, Ld, Lskip, Ladd = mex.blkchol(S["L"], it["ADA"], pars, absd_in)
    errs["L"] = relerr(LL, it["LL"])
    errs["d"] = relerr(Ld, it["Ld"])
    assert np.array_equal(Lskip.indices, it["Lskip"].indices), "skip decisions differ"
    assert np.array_equal(Ladd.indices, it["Ladd"].indices), "add decisions differ"
    L = dict(S["L"]); L["L"] = it["LL"]
    rhs = np.random.default_rng(seed).standard_normal((P.m, 2))
    errs["fw"] = relerr(mex.fwblkslv(L, rhs), G.ref.call("fwblkslv", 1, L, rhs))
    errs["bw"] = relerr(mex.bwblkslv(L, rhs), G.ref.call("bwblkslv", 1, L, rhs))
    bad = {k: v for k, v in errs.items() if not v < tol}
    assert not bad, f"{P.name}: {bad} (all: {errs})"
    return errs, S, it


def spd_pattern(kind, m, rng, dens=0.03):
    """Sparse symmetric diagonally dominant test matrices (the shipped examples never reach
    the sparse machinery, so the patterns are generated)."""
    if kind == "rand":
        B = sp.random(m, m, density=dens, random_state=rng, format="csc")
        X = B + B.T
    elif kind == "band":
        ks = [k for k in (1, 2, 5) if k < m]
        X = sp.diags([rng.standard_normal(m - k) for k in ks], ks, shape=(m, m), format="csc") if ks else sp.csc_matrix((m, m))
        X = X + X.T
    elif kind == "arrow":
        X = sp.lil_matrix((m, m))
        X[max(m - 3, 0):, :] = rng.standard_normal((min(3, m), m))
        X = sp.csc_matrix(X)
        X = X + X.T
    elif kind == "blockdiag":
        X = sp.block_diag([sp.csc_matrix(rng.standard_normal((b, b))) for b in rng.integers(1, 12, size=max(1, m // 6))], format="csc")
        X = X + X.T
    elif kind == "grid":
        n = int(np.sqrt(m))
        T = sp.diags([-1, -1], [1, -1], shape=(n, n))
        X = sp.csc_matrix(sp.kron(sp.eye(n), T) + sp.kron(T, sp.eye(n)))
    elif kind == "diag":
        X = sp.csc_matrix((m, m))
    else:
        raise ValueError(kind)
    X = sp.csc_matrix(X)
    dg = np.asarray(abs(X).sum(axis=1)).ravel() + 1.0
    X = sp.csc_matrix(X + sp.diags(dg))
    X.sort_indices()
    return X


def load_golden(name):
    z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    from sedumi_amd import problem
    At = sp.csc_matrix((z["At_data"], z["At_indices"], z["At_indptr"]), shape=tuple(z["At_shape"]))
    K = problem.make_K(int(z["K_l"]), z["K_q"].ravel(), z["K_s"].ravel())
    assert np.array_equal(K["blkstart"].ravel(), z["K_blkstart"].ravel())
    return z, At, K


def check_golden(name, tag):
    """Hot path through the MEX-equivalent calls on the inputs of a committed fixture (a reference example
    problem) against the outputs the unmodified reference MEX produced for it (tests/golden/make_golden.py)."""
    from sedumi_amd import mex, problem
    z, At, K = load_golden(name)
    m = At.shape[1]
    ADApat, L = problem.dense_pattern(m), problem.dense_symbolic(m)
    d = {"l": z[f"{tag}_dl"], "det": z[f"{tag}_ddet"]}
    Aord = {"lqperm": z["lqperm"], "qperm": z["qperm"], "sperm": z["sperm"]}
    A1 = mex.getada1(ADApat, At, z["Ablkjc"][:, 2], Aord["lqperm"], d, K["qblkstart"])
    assert abs(np.linalg.norm(A1.toarray()) - z[f"{tag}_ADA1_fro"]) <= 1e-12 * max(1.0, z[f"{tag}_ADA1_fro"])
    if f"{tag}_DAtq_data" in z.files:
        Q = sp.csc_matrix((z[f"{tag}_DAtq_data"], z[f"{tag}_DAtq_indices"], z[f"{tag}_DAtq_indptr"]), shape=tuple(z[f"{tag}_DAtq_shape"]))
    else:
        Q = sp.csc_matrix((0, m))
    A2 = mex.getada2(A1, {"q": Q}, Aord, K)
    if f"{tag}_ADA2_fro" in z.files:
        assert abs(np.linalg.norm(A2.toarray()) - z[f"{tag}_ADA2_fro"]) <= 1e-12 * max(1.0, z[f"{tag}_ADA2_fro"])
    A3, absd = mex.getada3(A2, At, z["Ablkjc"][:, 2], Aord, z[f"{tag}_udsqr"], K)
    ADA = A3.toarray()
    si, sj = z[f"{tag}_si"], z[f"{tag}_sj"]
    errs = {"absd": relerr(absd.ravel(), z[f"{tag}_absd"]), "ADA_s": relerr(ADA[si, sj], z[f"{tag}_ADA_s"]),
            "ADA_diag": relerr(np.diag(ADA), z[f"{tag}_ADA_diag"]),
            "ADA_fro": abs(np.linalg.norm(ADA) - z[f"{tag}_ADA_fro"]) / z[f"{tag}_ADA_fro"]}
    pars = {"canceltol": 1e-12, "maxu": 5e5, "abstol": 1e-20}
    LL, Ld, Lskip, Ladd = mex.blkchol(L, A3, pars, absd)
    Lf = LL.toarray()
    errs["Ld"] = relerr(Ld.ravel(), z[f"{tag}_Ld"])
    errs["L_s"] = relerr(Lf[np.maximum(si, sj), np.minimum(si, sj)], z[f"{tag}_L_s"])
    errs["L_fro"] = abs(np.linalg.norm(Lf) - z[f"{tag}_L_fro"]) / z[f"{tag}_L_fro"]
    assert Lskip.nnz == int(z[f"{tag}_nskip"]) and Ladd.nnz == int(z[f"{tag}_nadd"])
    if f"{tag}_L_tril" in z.files:                    # the full arrays, entry by entry
        errs["ADA_full"] = relerr(ADA[np.triu_indices(m)], z[f"{tag}_ADA_triu"])
        errs["ADA_sym"] = relerr(ADA, ADA.T)
        errs["L_full"] = relerr(Lf[np.tril_indices(m)], z[f"{tag}_L_tril"])
        errs["L_maxabs"] = float(np.abs(Lf[np.tril_indices(m)] - z[f"{tag}_L_tril"]).max() / np.abs(z[f"{tag}_L_tril"]).max())
    Ls = dict(L); Ls["L"] = LL
    yfw = mex.fwblkslv(Ls, z["rhs"])
    errs["yfw"] = relerr(yfw.ravel(), z[f"{tag}_yfw"])
    y = mex.bwblkslv(Ls, yfw / Ld)
    errs["y"] = relerr(y.ravel(), z[f"{tag}_y"])
    return errs


# ------------------------------------------------------------------ k_ldl_front against the launch-per-panel path
def bordered_blocks(n1, n2, nc, rng):
    """Two dense diagonal blocks, each coupled to a dense trailing block: the leaf fronts have many rows below
    their own columns (ms > ns)."""
    m = n1 + n2 + nc
    X = np.zeros((m, m))
    X[:n1, :n1] = rng.standard_normal((n1, n1)); X[n1:n1 + n2, n1:n1 + n2] = rng.standard_normal((n2, n2))
    X[n1 + n2:, :] = rng.standard_normal((nc, m))
    X = 0.1 * (X + X.T) / np.sqrt(m)
    X = X + np.diag(np.abs(X).sum(axis=1) + 1.0)
    X = sp.csc_matrix(X); X.sort_indices()
    return X


def _factor_both_ways(X, L, pars=None, rhs=None):
    """(lpr, d, pivots, y, kernels) of the resident plan with the one-launch front kernel and with the launch-per-panel path
    (Plan.set_one_launch_fronts(False) before set_chol: the comparison switch)."""
    from sedumi_amd.plan import Plan
    out = []
    for off in (False, True):
        plan = Plan(0)
        plan.set_one_launch_fronts(not off)
        plan.set_chol(L, X)
        plan.upload("ada", sp.csc_matrix(X).data)
        plan.upload("rhs", rhs if rhs is not None else np.ones(X.shape[0]))
        plan.kprof(True)
        plan.blkchol(pars, False)
        plan.ldlsolve()
        names = set(plan.kprof_summary().keys())
        plan.kprof(False)
        out.append((plan.download("lpr"), plan.download("d"), plan.pivots(), plan.download("y"), names))
    return out


def check_streamed_update_tiles(refmex, m, caps=(3, 16, 0)):
    """Launch-per-panel path of one dense front with the trailing-update tiles of every launch dealt to `cap` workgroups, each working
    through its tile pairs as a pipeline (panel_role_tiles_stream; Plan.set_tile_workgroups): whatever the number of workgroups -- 3:
    every one loops many times; 0: as many as the device has compute units -- every tile gets the same operations in the same order:
    the same bits, and within tolerance of the reference."""
    from oracle import glue as gl
    from sedumi_amd import problem
    from sedumi_amd.plan import Plan
    rng = np.random.default_rng(m)
    B = rng.standard_normal((m, m))
    X = sp.csc_matrix(B @ B.T + m * np.eye(m)); X.sort_indices()
    L = problem.dense_symbolic(m)
    res = []
    for cap in caps:
        plan = Plan(0)
        plan.set_one_launch_fronts(False)
        plan.set_tile_workgroups(cap)
        plan.set_chol(L, X)
        plan.upload("ada", sp.csc_matrix(X).data)
        plan.blkchol(None, False)
        res.append((plan.download("lpr"), plan.download("d")))
        plan.close()
    for lpr, d in res[1:]:
        assert np.array_equal(lpr, res[0][0]) and np.array_equal(d, res[0][1])
    r = refmex.call("blkchol", 4, L, X, gl.default_pars_chol())
    assert relerr(res[0][1], r[1].ravel()) < TOL and relerr(res[0][0], sp.csc_matrix(r[0]).data) < TOL


def check_one_launch_front(refmex, m):
    """k_ldl_front (one workgroup per tile row, the whole front in one launch) against the launch-per-panel path on single
    dense fronts: same device functions in the same order per entry, so the same bits -- and both within tolerance of
    the reference.  m = 666 is control07's shape (partial last panel); up to 21 tile rows every tile has a workgroup of its
    own on a whole device."""
    from oracle import glue as gl
    from sedumi_amd import problem
    rng = np.random.default_rng(m)
    B = rng.standard_normal((m, m))
    X = sp.csc_matrix(B @ B.T + m * np.eye(m)); X.sort_indices()
    L = problem.dense_symbolic(m)
    (l1, d1, p1, y1, k1), (l2, d2, p2, y2, k2) = _factor_both_ways(X, L, rhs=rng.standard_normal(m))
    assert "k_ldl_front" in k1 and "k_ldl_panel" not in k1 and "k_ldl_front" not in k2 and "k_ldl_panel" in k2
    # up to 14 tile rows (on a whole MI355X; always in the emulator) the inverse for the solves is built BEHIND the factor: by the last
    # workgroups of the k_ldl_front launch itself on the device, by a launch of its own (k_sinv_follow) in the emulator
    from sedumi_amd import capi
    behind = m <= 896 or capi.backend() == "emu"
    assert ("k_sinv_follow" in k1) == (capi.backend() == "emu")
    assert "k_sinv_follow" not in k2
    assert behind != ("k_sprep" in k1 or "k_sinv128" in k1)
    assert np.array_equal(l1, l2) and np.array_equal(d1, d2) and relerr(y1, y2) < 1e-12   # (the inverses for the solves are built differently: behind k_ldl_front / after the panel launches)
    r = refmex.call("blkchol", 4, L, X, gl.default_pars_chol())
    assert relerr(d1, r[1].ravel()) < TOL and relerr(l1, sp.csc_matrix(r[0]).data) < TOL


def check_one_launch_levels(refmex, glue, two_leaves=False):
    """A leaf front of 64 columns with 800 rows below them (the update matrix for the parent) and the dense root of 928
    columns: both levels take k_ldl_front, extend-add in between.  two_leaves: two leaf fronts of 64 columns (764 rows each)
    in ONE k_ldl_front launch (grid.y = 2) under a root of 764 columns."""
    from oracle import glue as gl
    rng = np.random.default_rng(5)
    if two_leaves:
        sizes, nc = [64, 64, 64], 700
        m = sum(sizes) + nc
        X = np.zeros((m, m)); o = 0
        for n in sizes:
            X[o:o + n, o:o + n] = rng.standard_normal((n, n)); o += n
        X[o:, :] = rng.standard_normal((nc, m))
        X = 0.1 * (X + X.T) / np.sqrt(m)
        X = sp.csc_matrix(X + np.diag(np.abs(X).sum(axis=1) + 1.0)); X.sort_indices()
    else:
        X = bordered_blocks(64, 128, 800, rng)
    L = glue.symbchol(X)
    xs = L["xsuper"].ravel().astype(int)
    assert xs.size - 1 == (3 if two_leaves else 2) and xs[1] - xs[0] == 64
    (l1, d1, p1, y1, k1), (l2, d2, p2, y2, k2) = _factor_both_ways(X, L, rhs=rng.standard_normal(X.shape[0]))
    assert "k_ldl_front" in k1 and "k_ldl_panel" not in k1 and "k_ldl_front" not in k2
    assert np.array_equal(l1, l2) and np.array_equal(d1, d2) and relerr(y1, y2) < 1e-12   # (the inverses for the solves are built differently: behind k_ldl_front / after the panel launches)
    r = refmex.call("blkchol", 4, L, X, gl.default_pars_chol())
    assert relerr(d1, r[1].ravel()) < TOL and relerr(l1, sp.csc_matrix(r[0]).data) < TOL


def check_one_launch_pivot_rule(refmex, m, maxu):
    """Rank-deficient dense front of k_ldl_front's size: skipped pivots, the never-fail rule's column probe (the rare path
    of the diagonal-block code: it waits for the update steps of the rows below) and added diagonals -- same decisions
    as the reference, same bits as the launch-per-panel path."""
    from oracle import glue as gl
    from sedumi_amd import problem
    rng = np.random.default_rng(m + int(maxu))
    h = m // 2
    B = rng.standard_normal((h, h))
    A = B @ B.T + h * np.eye(h)
    # second half: the Schur complement is E = diag(s) (C C' / h + I) diag(s), s_j^2 between 1e-10 h and 1e-5 h -- pivots and
    # off-diagonals far above the rounding noise of the elimination (1e-13 h), so that every decision is a clean one
    C = rng.standard_normal((m - h, m - h))
    sj = np.sqrt(10.0 ** rng.uniform(-10, -5, m - h) * h)
    E = (C @ C.T / (m - h) + np.eye(m - h)) * np.outer(sj, sj)
    X = np.block([[A, A], [A, A + E]])
    X = np.pad(X, ((0, m - 2 * h), (0, m - 2 * h))); X[2 * h:, 2 * h:] = np.eye(m - 2 * h)
    X = sp.csc_matrix(X + 0.0); X = sp.csc_matrix((X.toarray().ravel(order="F"), np.tile(np.arange(m), m), np.arange(0, m * m + 1, m)), shape=(m, m))
    L = problem.dense_symbolic(m)
    pars = dict(gl.default_pars_chol()); pars["maxu"] = maxu; pars["canceltol"] = 1e-8          # about half of those pivots are skipped
    (l1, d1, p1, y1, k1), (l2, d2, p2, y2, k2) = _factor_both_ways(X, L, pars, rng.standard_normal(m))
    assert "k_ldl_front" in k1 and "k_ldl_front" not in k2
    r = refmex.call("blkchol", 4, L, X, pars)
    (si, sv), (ai, av) = p1
    assert np.array_equal(si, sp.csc_matrix(r[2]).indices) and np.array_equal(ai, sp.csc_matrix(r[3]).indices)
    assert si.size + ai.size > 0
    assert np.array_equal(l1, l2) and np.array_equal(d1, d2) and np.array_equal(p1[0][0], p2[0][0]) and np.array_equal(p1[1][1], p2[1][1])
    assert relerr(d1, r[1].ravel()) < 1e-8


def rank_deficient_front_case(rng, mmin=320, mmax=700):
    """(L, X, pars[, absd]) of one dense front X = B B' (rank r between m/3 and m) + a diagonal of 1e-14 .. 1e-2 of its largest
    entry, maxu drawn from {5e5, 30, 2}, with or without a scaled |diagonal| as absd: skipped pivots, column probes and added
    diagonals ANYWHERE in a block -- also after the block's first 16-column groups have been handed on (tests/tools/soak_def.py
    draws the same cases without end)."""
    from oracle import glue as gl
    from sedumi_amd import problem
    pars = dict(gl.default_pars_chol())
    m = int(rng.integers(mmin, mmax)); r = int(rng.integers(m // 3, m))
    B = rng.standard_normal((m, r))
    X = B @ B.T
    X = sp.csc_matrix(X + np.diag(10.0 ** rng.uniform(-14, -2, m)) * np.abs(X).max()); L = problem.dense_symbolic(m)
    pars["maxu"] = float(rng.choice([5e5, 30.0, 2.0]))
    absd = (np.abs(X.diagonal()) * rng.choice([1.0, 1e3, 1e8], m)).reshape(-1, 1) if rng.random() < 0.5 else None
    return (L, X, pars) + ((absd,) if absd is not None else ())


def check_rank_deficient_fronts(refmex, ncases, seed=777):
    """k_ldl_front on the cases of rank_deficient_front_case: the reference's skip / add decisions, index by index, and its
    pivots.  (Round 3 found 175 of 1475 of these wrong while the chain workgroup stored its solved rows group by group --
    under the column probe of a later group of the same block, which reads those rows unsolved: profiles/r03ap_soak_def.txt.)"""
    from sedumi_amd import mex
    rng = np.random.default_rng(seed)
    nprobe = 0
    for case in range(ncases):
        args = rank_deficient_front_case(rng)
        rr = refmex.call("blkchol", 4, *args)
        o = mex.blkchol(*args)
        assert np.array_equal(o[2].indices, rr[2].indices) and np.array_equal(o[3].indices, rr[3].indices), (case, o[3].nnz, rr[3].nnz)
        assert relerr(o[1], rr[1]) < 1e-8, case
        nprobe += rr[3].nnz
    assert nprobe > 0


def check_inverse_launch_paths(m, seed=0):
    """The inverses of the diagonal super-blocks by ONE launch with completion counters (k_sprep: problems whose items fit the device)
    and by a launch per stage (k_sinv128 + k_stile, items sorted longest first; Plan.set_one_launch_inverse(False)): the same
    solutions bit for bit, and numpy's to rounding."""
    from sedumi_amd import problem
    from sedumi_amd.plan import Plan
    rng = np.random.default_rng(m + seed)
    Lv = np.tril(rng.standard_normal((m, m)) * (0.5 / np.sqrt(m)), -1) + np.eye(m)
    d = 0.5 + rng.random(m)
    X = Lv @ np.diag(d) @ Lv.T
    L = problem.dense_symbolic(m)
    rhs = rng.standard_normal(m)
    want = np.linalg.solve(X, rhs)
    got = {}
    for one in (True, False):
        plan = Plan(0)
        plan.set_one_launch_inverse(one)
        plan.set_one_launch_fronts(False)          # (the inverse behind a one-launch front is built by the follower, by neither of the two)
        plan.set_chol(L, problem.dense_pattern(m))
        plan.upload("ada", X.ravel(order="F"))
        plan.kprof(True)
        plan.blkchol(None, False)
        prof = plan.kprof_summary()
        plan.kprof(False)
        assert ("k_sprep" in prof) == one and ("k_stile" in prof) == (not one), sorted(prof)
        plan.upload("rhs", rhs); plan.ldlsolve(); got[one] = plan.download("y")
        nb, bad, _ = plan.solve_stats()
        assert bad == 0
        plan.close()
    assert np.array_equal(got[True], got[False])
    assert np.max(np.abs(got[True] - want)) / np.max(np.abs(want)) < 1e-10


def check_solve_widths(m, thr, seed=0):
    """The solves of a one-front factor with every super-block width the front admits (sdm_plan_set_solve_width: 256, 512,
    ... up to the automatic choice = one block when m <= 2048): every width against numpy's solve -- on the inverse path,
    with every super-block beyond the growth bound (bound 0) and with good and bad blocks mixed -- several solves in
    a row, and the launch count 2 (2 nsb - 1) per fw + bw.  Blocks beyond the bound three ways (sdm_plan_set_refinement): always
    substituted (mode 0), inverse + iterative refinement in every solve (mode 2: 4 more launches per diagonal stage), and the default
    (mode 1): the first solve substitutes and leaves a note, the later ones refine."""
    from sedumi_amd import problem
    from sedumi_amd.plan import Plan
    rng = np.random.default_rng(m + seed)
    Lv = np.tril(rng.standard_normal((m, m)) * (0.5 / np.sqrt(m)), -1) + np.eye(m)
    if thr is not None and thr > 0 and m >= 512:
        Lv[256:512, 256:512] = np.tril(Lv[256:512, 256:512], -1) * 1e-3 + np.eye(256)
    d = 0.5 + rng.random(m)
    X = Lv @ np.diag(d) @ Lv.T
    L = problem.dense_symbolic(m)
    rhss = [rng.standard_normal(m) for _ in range(3)]
    wants = [np.linalg.solve(X, r) for r in rhss]
    wauto = 256
    while wauto < m and wauto < 2048:
        wauto *= 2
    widths = [0] + [w for w in (256, 512, 1024) if w < wauto]
    # (all three refinement modes at the automatic width, substitution and the default mode at the other widths whose sweeps can merge)
    for width, refine in [(w, r) for w in widths for r in ((1,) if thr is None else (0, 1, 2) if w == 0 else (0, 1) if w > 256 else (1,))]:
        plan = Plan(0)
        plan.set_solve_width(width)
        plan.set_refinement(refine)
        plan.set_chol(L, problem.dense_pattern(m))
        if thr is not None:
            plan.set_growth_max(thr)
        plan.upload("ada", X.ravel(order="F"))
        plan.blkchol(None, False)
        W = width or wauto
        nsb = (m + W - 1) // W
        nb, bad, _ = plan.solve_stats()
        assert nb == nsb
        if thr == 0.0:
            assert bad == nsb
        elif thr is None:
            assert bad == 0
        # the sweeps' launches: separate (SEDUMI_HIP_SWEEP_MERGE=0), a row / step launch merged with the NEXT diagonal block's where rows beyond that
        # block stream on (1, the default: nsb - 2 merges per sweep) or wherever a next block exists (2: nsb - 1) -- one-front levels, W > 256, no
        # refinement launches.  Same arithmetic per row: bit-identical results where every block is within the growth bound
        levels = (1, 0, 2) if W > 256 and nsb >= 2 else (1,)
        first = None
        for level in levels:
            os.environ["SEDUMI_HIP_SWEEP_MERGE"] = str(level)
            try:
                plan.kprof(True)
                ys = []
                for r in rhss if level == 1 else rhss[:2]:                # (the other levels: two of the three right-hand sides)
                    plan.upload("rhs", r); plan.ldlsolve(); ys.append(plan.download("y"))
                prof = plan.kprof_summary()
                plan.kprof(False)
            finally:
                del os.environ["SEDUMI_HIP_SWEEP_MERGE"]
            nl = sum(v[0] for k, v in prof.items() if k.startswith("k_sfw") or k.startswith("k_sbw"))
            nm = 0 if W <= 256 or level == 0 else 2 * max(0, nsb - (2 if level == 1 else 1))
            lean, robust = 2 * (2 * nsb - 1), 2 * (2 * nsb - 1) + 8 * nsb
            if refine == 2:
                assert nl == len(ys) * robust, (width, refine, level, prof)
            elif refine == 1 and thr is not None and bad > 0:
                # the download after the first solve made the note visible (the emulator runs a launch to its end at once: there the
                # first solve's backward sweep already sees the note its forward sweep left)
                assert lean - nm + (len(ys) - 1) * robust <= nl <= len(ys) * robust, (width, refine, level, prof)
            else:
                assert nl == len(ys) * (lean - nm), (width, refine, level, prof)
                if nm:
                    assert any("rows_diag" in k for k in prof) and any("step_diag" in k for k in prof), prof
            for y, want in zip(ys, wants):
                assert relerr(y, want) < 1e-9, (width, refine, level, relerr(y, want))
            # the two sweeps on their own (fwblkslv / bwblkslv: no ./d folded into the forward sweep, the backward sweep from a fresh vector)
            os.environ["SEDUMI_HIP_SWEEP_MERGE"] = str(level)
            try:
                plan.upload("rhs", rhss[0]); plan.fwsolve(); yfw = plan.download("y")
                plan.upload("rhs", rhss[1]); plan.bwsolve(); ybw = plan.download("y")
            finally:
                del os.environ["SEDUMI_HIP_SWEEP_MERGE"]
            if thr is None:
                assert relerr(yfw, np.linalg.solve(Lv, rhss[0])) < 1e-9 and relerr(ybw, np.linalg.solve(Lv.T, rhss[1])) < 1e-9, (width, level)
            if first is None:
                first = (ys, yfw, ybw)
            elif bad == 0:                                          # (a block beyond the bound: the merged launch substitutes in tiles of 16, not 32)
                assert all(np.array_equal(u, v) for u, v in zip(first[0], ys)) and np.array_equal(first[1], yfw) and np.array_equal(first[2], ybw), (width, refine, level)
        plan.close()


def check_refined_solve_accuracy(m=320, seed=3, glo=1e5, ghi=1e7, cancelling=False):
    """An ILL-CONDITIONED unit lower factor (growth max|inv(L)| max|L| between the default bound 1e4 and 1e8, as in the late iterations of an
    interior-point run): the solve by substitution (sdm_plan_set_refinement mode 0), by the explicit inverse refined twice against the
    factor (mode 2) and by the bare inverse (bound lifted), each against the solution in extended precision.  The refined result is as
    accurate as the substitution it replaces (cancelling: a right-hand side L * ones, for which inv(L) b cancels by the size of inv(L)'s
    entries -- there the bare inverse is up to 10x worse than either); the errors are returned for the record."""
    from sedumi_amd import problem
    from sedumi_amd.plan import Plan
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    G = np.tril(rng.standard_normal((m, m)), -1)
    d = 0.5 + rng.random(m)
    b = rng.standard_normal(m)

    def make(alpha, mode, bound):
        Lv = np.eye(m) + alpha * G
        plan = Plan(0)
        plan.set_refinement(mode)
        if bound is not None:
            plan.set_growth_max(bound)
        plan.set_chol(problem.dense_symbolic(m), problem.dense_pattern(m))
        plan.load_factor(sp.csc_matrix(np.tril(np.where(Lv != 0, Lv, 0.0)) + 0.0 * np.tril(np.ones((m, m)))), d)
        return plan, Lv

    alpha = 0.02
    for _ in range(40):                                     # (growth rises monotonically with alpha)
        plan, Lv = make(alpha, 0, None)
        growth = plan.solve_stats()[2]
        plan.close()
        if glo <= growth <= ghi:
            break
        alpha *= 1.15 if growth < glo else 0.93
    assert glo <= growth <= ghi, growth
    if cancelling:                                          # b = L * ones: inv(L) b cancels from the size of inv(L)'s entries down to 1
        b = np.asarray(Lv.astype(np.longdouble) @ np.ones(m, dtype=np.longdouble), dtype=np.float64)
    Lx = Lv.astype(np.longdouble)
    y = b.astype(np.longdouble).copy()
    for i in range(m):
        y[i] -= Lx[i, :i] @ y[:i]
    z = y / d.astype(np.longdouble)
    for i in range(m - 1, -1, -1):
        z[i] -= Lx[i + 1:, i] @ z[i + 1:]
    want = z
    errs = {}
    for name, mode, bound in (("substitution", 0, None), ("refined", 2, None), ("bare_inverse", 0, 1e300)):
        plan, _ = make(alpha, mode, bound)
        nb, bad, g = plan.solve_stats()
        assert bad == (0 if bound else 1), (name, bad)
        plan.upload("rhs", b); plan.ldlsolve()
        got = plan.download("y").astype(np.longdouble)
        errs[name] = float(np.linalg.norm(got - want) / np.linalg.norm(want))
        plan.close()
    # (same accuracy class: both are bounded by eps * | |inv(L)| |L| |x| |, with different constants and summation orders)
    assert errs["refined"] <= 8 * errs["substitution"] + 4e-15, (growth, errs)
    return growth, errs


def check_refinement_prediction(m=300):
    """sdm_plan_set_refinement mode 1 over a sequence of factorisations: the first sweeps that meet a block beyond the bound substitute
    (and say so to the host), the later ones refine -- also across the next factorisations -- and once the factors are within the
    bound again the extra launches are dropped after a few sweeps."""
    from sedumi_amd import problem
    from sedumi_amd.plan import Plan
    rng = np.random.default_rng(m)
    Lv = np.tril(rng.standard_normal((m, m)) * (0.5 / np.sqrt(m)), -1) + np.eye(m)
    X = Lv @ np.diag(0.5 + rng.random(m)) @ Lv.T
    b = rng.standard_normal(m)
    want = np.linalg.solve(X, b)
    plan = Plan(0)
    plan.set_chol(problem.dense_symbolic(m), problem.dense_pattern(m))
    plan.upload("ada", X.ravel(order="F")); plan.upload("rhs", b)
    lean, robust = 2, 10                                              # one super-block: launches per solve

    def solves(n):
        counts = []
        for _ in range(n):
            plan.kprof(True); plan.ldlsolve(); y = plan.download("y"); prof = plan.kprof_summary(); plan.kprof(False)
            assert relerr(y, want) < 1e-9
            counts.append(sum(v[0] for k, v in prof.items() if k.startswith("k_sfw") or k.startswith("k_sbw")))
        return counts
    plan.blkchol(None, False)
    assert solves(2) == [lean, lean]                                  # within the bound
    plan.set_growth_max(0.0)
    plan.blkchol(None, False)
    c = solves(3)
    assert lean <= c[0] <= robust and c[1:] == [robust, robust], c    # (the first solve's sweeps may or may not have the news yet)
    plan.blkchol(None, False)
    assert solves(2) == [robust, robust]                              # stays on across factorisations
    plan.set_growth_max(1e4)
    plan.blkchol(None, False)
    c = solves(4)
    assert c[0] >= c[1] >= c[2] >= c[3] and c[2:] == [lean, lean], c  # dropped after a few clean sweeps
    plan.close()


def check_direct_columns_kernel(n):
    """k_psd_direct_cols (full columns of a dense ADA' pattern, 8 columns per workgroup: the resident plan on MAXCUT-shaped problems) against
    k_psd_direct (one column per workgroup: what the MEX route with its Aord permutations runs): the same bits, also for a column panel
    (sdm_plan_getada_cols, the multi-GPU hook) whose width is not a multiple of 8."""
    import scipy.sparse as sp
    from sedumi_amd import mex, problem
    from sedumi_amd.plan import Plan
    P = problem.maxcut(n)
    d, ud = problem.spd_scaling(P.K, seed=5)
    L, ADA = problem.dense_symbolic(P.m), problem.dense_pattern(P.m)
    plan = Plan(0)
    plan.set_chol(L, ADA); plan.set_ada(P.At, P.Ablkjc, P.K, problem.lorentz_pattern(P))
    plan.upload("dl", d["l"]); plan.upload("ddet", d["det"]); plan.upload("udsqr", ud)
    plan.kprof(True); plan.getada(); prof = plan.kprof_summary(); plan.kprof(False)
    assert "k_psd_direct_cols" in prof and "k_psd_direct" not in prof, sorted(prof)
    a1, s1 = plan.download("ada"), plan.download("absd")
    perm1 = np.arange(1, P.m + 1, dtype=np.float64)
    A0 = mex.getada1(ADA, P.At, P.Ablkjc[:, 2], perm1, d, P.K["qblkstart"])
    A3, absd = mex.getada3(A0, P.At, P.Ablkjc[:, 2], {"sperm": perm1}, ud, P.K)
    assert np.array_equal(a1, sp.csc_matrix(A3).toarray().ravel(order="F")) and np.array_equal(s1, np.asarray(absd).ravel())
    plan.upload("ada", np.full(plan.nnzADA, np.nan)); plan.getada_cols(5, n - 7)
    a3 = plan.download("ada").reshape(n, n, order="F")
    assert np.array_equal(a3[:, 5:n - 7], a1.reshape(n, n, order="F")[:, 5:n - 7]) and np.isnan(a3[:, :5]).all() and np.isnan(a3[:, n - 7:]).all()

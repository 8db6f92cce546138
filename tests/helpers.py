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

"""Dense-column path: symbfwblk, finsymbden, dpr1fact, fwdpr1, bwdpr1 against the
compiled reference MEX, on LPs with a few dense variables (the case the shipped examples never reach).
CPU run = fiber emulator (kernel logic + host code); the same body runs on the GPU in test_gpu_parity.py."""
import numpy as np
import pytest
import scipy.sparse as sp

from helpers import TOL, relerr, use_emu


@pytest.fixture(scope="module", autouse=True)
def _emu():
    use_emu()


def dense_case(refmex, glue, m, n, ndense, seed, zero_d=0, maxuden=500.0):
    """LP with `ndense` dense variables: reference pipeline up to the dpr1fact inputs (symbcholden.m:43-55,
    deninfac.m:58-72 restated), returning everything both implementations consume."""
    from oracle import glue as gl
    from oracle.refmex import RawSparse
    from sedumi_amd import mex, problem
    rng = np.random.default_rng(seed)
    P = problem.lp_dense_cols(m=m, n=n, dens=0.01, ndense=ndense, seed=seed)
    At = sp.csc_matrix(P.At)                                   # N x m, rows = variables
    rows_dense = 1 + np.arange(ndense)                         # the dense variables (row 0 is x0)
    denseA = sp.csc_matrix(At[rows_dense, :].T)                # m x ndense      (sedumi.m:359)
    keep = np.ones(At.shape[0]); keep[rows_dense] = 0.0
    Asp = sp.csc_matrix(sp.diags(keep) @ At)                   # sedumi.m:360
    dl = 10.0 ** rng.uniform(-1, 1, At.shape[0])
    ADA = sp.csc_matrix(Asp.T @ sp.diags(dl) @ Asp)            # getada.m:13-40 (LP only)
    ADA.sort_indices()
    L = glue.symbchol(sp.csc_matrix((np.ones(ADA.nnz), ADA.indices, ADA.indptr), shape=ADA.shape))
    pars = gl.default_pars_chol()
    LL, Ld, Lskip, Ladd = refmex.call("blkchol", 4, L, ADA, pars)
    Lf = dict(L); Lf["L"] = LL
    Ld = np.asarray(Ld).ravel().copy()
    if zero_d:
        Ld[rng.choice(m, zero_d, replace=False)] = 0.0         # dependent rows (d = 0): the partition branch of dodpr1fact
    LADsym = refmex.call("symbfwblk", 1, L, denseA)            # symbcholden.m:45-46, LP columns only
    perm, dz = refmex.call("incorder", 2, LADsym)
    sym_ref = refmex.call("finsymbden", 1, LADsym, perm, RawSparse(dz), float(ndense + 1))
    smult = dl[rows_dense]                                     # deninfac.m:61
    LAD = refmex.call("fwblkslv", 1, Lf, denseA, sym_ref["LAD"])    # sparfwslv.m:55-57
    return dict(L=L, Lf=Lf, Ld=Ld, denseA=denseA, LADsym=LADsym, perm=perm, dz=dz, sym_ref=sym_ref, smult=smult, LAD=LAD,
                maxuden=maxuden, rng=rng)


def check_dense_case(refmex, c):
    from oracle.refmex import RawSparse
    from sedumi_amd import mex
    # ---- symbolic
    X = mex.symbfwblk(c["L"], c["denseA"])
    assert np.array_equal(X.indptr, c["LADsym"].indptr) and np.array_equal(X.indices, c["LADsym"].indices)
    sym = mex.finsymbden(c["LADsym"], c["perm"], c["dz"], float(c["denseA"].shape[1] + 1))
    r = c["sym_ref"]
    assert np.array_equal(sym["perm"].ravel(), np.asarray(r["perm"]).ravel())
    assert np.array_equal(sym["first"].ravel(), np.asarray(r["first"]).ravel())
    assert np.array_equal(sym["dz"].indptr, r["dz"].indptr) and np.array_equal(sym["dz"].indices, r["dz"].indices)
    # ---- numeric: product-form factorisation
    sref = {"dz": RawSparse(r["dz"]), "perm": r["perm"], "first": r["first"]}
    Lden_r, Ld_r = refmex.call("dpr1fact", 2, c["LAD"], c["Ld"].reshape(-1, 1), sref, c["smult"].reshape(-1, 1), c["maxuden"])
    Lden, Ld = mex.dpr1fact(c["LAD"], c["Ld"], sym, c["smult"], c["maxuden"])
    same_order = np.array_equal(Lden["pivperm"].ravel(), np.asarray(Lden_r["pivperm"]).ravel()) and \
        np.array_equal(Lden["dopiv"].ravel(), np.asarray(Lden_r["dopiv"]).ravel())
    if same_order:
        assert np.array_equal(Lden["betajc"].ravel(), np.asarray(Lden_r["betajc"]).ravel())
    else:
        # (a different order of the postponed rows of column k gives another -- equally valid -- d and forward-solved later columns,
        # so the decisions of the columns AFTER k may differ: lengths and flags are compared up to and including k)
        ka = int(np.flatnonzero(np.asarray(Lden_r["dopiv"]).ravel())[0])
        assert np.array_equal(Lden["dopiv"].ravel()[:ka + 1], np.asarray(Lden_r["dopiv"]).ravel()[:ka + 1])
        assert np.array_equal(Lden["betajc"].ravel()[:ka + 2], np.asarray(Lden_r["betajc"]).ravel()[:ka + 2])
    if same_order:
        assert relerr(Lden["p"], Lden_r["p"]) < TOL and relerr(Lden["beta"], Lden_r["beta"]) < TOL and relerr(Ld, Ld_r) < TOL
    else:
        # The order of the POSTPONED rows of a reordered column comes from kdsortdec (dpr1fact.c:349,455): qsort with a
        # comparator that returns `char` through an `int (*)()` pointer (sdmauxCmp.c:60, blksdp.h:140) -- undefined
        # behaviour; the gcc build of the reference leaves the upper bytes of the result to chance (here it reverses the
        # list).  We sort by decreasing p_j^2 as documented.  First-round decisions do not depend on the sort: the
        # accepted prefix and the SET of postponed rows of the first reordered column must agree.
        dz = r["dz"]
        k = int(np.flatnonzero(np.asarray(Lden_r["dopiv"]).ravel())[0])
        mk = int(dz.indptr[k + 1])
        a, b_ = Lden["pivperm"].ravel()[:mk], np.asarray(Lden_r["pivperm"]).ravel()[:mk]
        ndiff = int(np.argmax(a != b_)) if np.any(a != b_) else mk
        assert sorted(a[ndiff:]) == sorted(b_[ndiff:]) and np.array_equal(a[:ndiff], b_[:ndiff])
    # ---- the factorisation must reproduce  diag(d) + LAD diag(smult) LAD'  (deninfac.m:67-72): solve check
    m = c["Ld"].size
    LADd = np.asarray(c["LAD"].todense())
    Xfull = np.diag(c["Ld"]) + LADd @ np.diag(c["smult"]) @ LADd.T
    Lmine = dict(Lden); Lmine["dz"] = sym["dz"]
    rhs = c["rng"].standard_normal((m, 2))
    ok = np.asarray(Ld).ravel() > 0
    if ok.all():
        sol = mex.bwdpr1(Lmine, mex.fwdpr1(Lmine, rhs) / np.asarray(Ld).reshape(-1, 1))
        assert relerr(Xfull @ sol, rhs) < 1e-8
    # ---- solves with the reference factor (stage-wise parity)
    Lr = dict(Lden_r); Lr["dz"] = r["dz"]
    Lr_ref = dict(Lden_r); Lr_ref["dz"] = RawSparse(r["dz"])
    b = c["rng"].standard_normal((m, 3))
    yf_r, yb_r = refmex.call("fwdpr1", 1, Lr_ref, b), refmex.call("bwdpr1", 1, Lr_ref, b)
    assert relerr(mex.fwdpr1(Lr, b), yf_r) < TOL
    assert relerr(mex.bwdpr1(Lr, b), yb_r) < TOL
    return Lden_r


@pytest.mark.parametrize("m,n,ndense,seed,zero_d,maxuden", [(60, 400, 3, 1, 0, 500.0), (120, 900, 6, 2, 0, 500.0),
                                                            (90, 700, 4, 3, 2, 500.0), (80, 600, 5, 4, 0, 1.5),
                                                            (700, 3000, 4, 5, 0, 500.0)])
def test_dense_column_pipeline(refmex, glue, m, n, ndense, seed, zero_d, maxuden):
    c = dense_case(refmex, glue, m, n, ndense, seed, zero_d, maxuden)
    Lden_r = check_dense_case(refmex, c)
    if maxuden < 2.0:
        assert np.asarray(Lden_r["dopiv"]).sum() > 0, "case meant to exercise the reordered (pivoted) factors"


def test_no_dense_columns_is_identity():
    from sedumi_amd import mex
    b = np.arange(6.0).reshape(3, 2)
    Lden = {"betajc": np.array([[1.0]])}                       # deninfac.m:81 -> betajc = 0 in MATLAB is 1 entry: nden = 0
    assert np.array_equal(mex.fwdpr1(Lden, b), b) and np.array_equal(mex.bwdpr1(Lden, b), b)


def check_resident_dense_unit(refmex, c, expect_host=None):
    """The dense-column unit resident on the plan ("+ sparse fwblkslv + dpr1fact + 4 x (fwdpr1+bwdpr1)"):
    LAD = L \\ Ad(perm,:) for all columns at once, dpr1fact on the device (host algorithm when a column needs the general
    path), then the whole wrapPcg.m:56-59 body -- against the reference chain fwblkslv(L,Ad,ysymb) -> dpr1fact ->
    fwblkslv, fwdpr1, ./Ld, bwdpr1, bwblkslv."""
    from oracle.refmex import RawSparse
    from sedumi_amd.plan import Plan
    r = c["sym_ref"]
    L, Lf, m = c["L"], c["Lf"], c["Ld"].size
    nden = c["denseA"].shape[1]
    plan = Plan(0)
    plan.set_chol(L, sp.identity(m, format="csc"))              # the ADA' pattern plays no role here
    plan.load_factor(Lf["L"], c["Ld"])
    plan.set_dense({"LAD": r["LAD"], "dz": r["dz"], "perm": r["perm"], "first": r["first"]})
    plan.upload("ad", np.asarray(c["denseA"].todense()).ravel(order="F"))
    host = plan.deninfac(c["smult"], c["maxuden"])
    if expect_host is not None:
        assert host == expect_host
    assert relerr(plan.download("lad", m * nden).reshape(m, nden, order="F"), c["LAD"]) < TOL
    sref = {"dz": RawSparse(r["dz"]), "perm": r["perm"], "first": r["first"]}
    Lden_r, Ld_r = refmex.call("dpr1fact", 2, c["LAD"], c["Ld"].reshape(-1, 1), sref, c["smult"].reshape(-1, 1), c["maxuden"])
    Lden, Ld = plan.lden()
    same = np.array_equal(Lden["pivperm"], np.asarray(Lden_r["pivperm"]).ravel()) and np.array_equal(Lden["dopiv"], np.asarray(Lden_r["dopiv"]).ravel())
    if not same:                                                    # (see check_dense_case: the reference's sort of the postponed rows is undefined behaviour)
        ka = int(np.flatnonzero(np.asarray(Lden_r["dopiv"]).ravel())[0])
        assert np.array_equal(Lden["dopiv"][:ka + 1], np.asarray(Lden_r["dopiv"]).ravel()[:ka + 1])
        assert np.array_equal(Lden["betajc"][:ka + 2], np.asarray(Lden_r["betajc"]).ravel()[:ka + 2])
        # ... and the stateless entry point gives the same factors as the resident unit
        from sedumi_amd import mex
        Lm, Ldm = mex.dpr1fact(c["LAD"], c["Ld"], {"dz": r["dz"], "perm": r["perm"], "first": r["first"]}, c["smult"], c["maxuden"])
        assert np.array_equal(Lden["pivperm"], Lm["pivperm"].ravel()) and relerr(Lden["beta"], Lm["beta"].ravel()) < 1e-13 and relerr(Ld, Ldm.ravel()) < 1e-13
    if same:
        assert np.array_equal(Lden["betajc"], np.asarray(Lden_r["betajc"]).ravel())
        rv = lambda a: np.asarray(a).ravel()
        assert relerr(Lden["p"], rv(Lden_r["p"])) < TOL and relerr(Lden["beta"], rv(Lden_r["beta"])) < TOL and relerr(Ld, rv(Ld_r)) < TOL
        # the complete solve of wrapPcg.m:56-59 with the reference's factors
        Lr_ref = dict(Lden_r); Lr_ref["dz"] = RawSparse(r["dz"])
        rhs = c["rng"].standard_normal((m, 1))
        Ldr = np.asarray(Ld_r).reshape(-1, 1)
        if np.all(Ldr > 0):
            pvec = refmex.call("fwdpr1", 1, Lr_ref, refmex.call("fwblkslv", 1, Lf, rhs))
            want = refmex.call("bwblkslv", 1, Lf, refmex.call("bwdpr1", 1, Lr_ref, pvec / Ldr))
            plan.upload("rhs", rhs.ravel()); plan.ldlsolve()
            assert relerr(plan.download("y"), want.ravel()) < TOL
    plan.close()
    return host


@pytest.mark.parametrize("m,n,ndense,seed,zero_d,maxuden,expect_host", [(60, 400, 3, 1, 0, 500.0, False), (120, 900, 6, 2, 0, 500.0, False),
                                                                        (90, 700, 4, 3, 2, 500.0, False), (80, 600, 5, 4, 0, 1.5, False),
                                                                        (700, 3000, 4, 5, 0, 500.0, False)])
def test_resident_dense_column_unit(refmex, glue, m, n, ndense, seed, zero_d, maxuden, expect_host):
    c = dense_case(refmex, glue, m, n, ndense, seed, zero_d, maxuden)
    check_resident_dense_unit(refmex, c, expect_host)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_adendotd_and_adenscale_match_reference(refmex, seed):
    """SURVEY 8f N3, dense-column half of getDAtm.m:45 / deninfac.m:61: dense Lorentz blocks (their trace columns and
    some dense norm-bound columns) on a synthetic `dense` structure, bit for bit against adendotd.c / adenscale.c."""
    from sedumi_amd import mex
    rng = np.random.default_rng(seed)
    m, lorN = 40, 6
    qdims = rng.integers(3, 7, lorN)
    firstQ = 20                                                    # 1-based subscript of the first norm-bound variable = blkstart(1)
    blkstart = np.concatenate(([firstQ], firstQ + np.cumsum(qdims - 1))).astype(np.float64)
    dq = np.sort(rng.choice(lorN, 3, replace=False))               # dense Lorentz blocks (0-based)
    nl = 2
    dencols = []                                                    # dense norm-bound columns: global subscripts inside the dense blocks
    for k in dq:
        lo, hi = int(blkstart[k]), int(blkstart[k + 1])
        dencols += sorted(rng.choice(np.arange(lo, hi), min(2, hi - lo), replace=False).tolist())
    nden = len(dencols)
    cols = np.concatenate((np.arange(1, nl + 1), 10 + dq + 1, np.asarray(dencols))).astype(np.float64)
    A = sp.random(m, nl + dq.size + nden, density=0.4, random_state=rng, format="csc"); A.sort_indices()
    dense = {"l": float(nl), "q": (dq + 1.0).reshape(-1, 1), "cols": cols.reshape(-1, 1), "A": A}
    d = {"q1": 1.0 + rng.random(lorN), "q2": rng.standard_normal(int(blkstart[-1] - firstQ)), "det": 0.5 + rng.random(lorN)}
    sparAd = sp.random(m, dq.size, density=0.3, random_state=rng, format="csc"); sparAd.sort_indices()
    Ablk = sp.csc_matrix(((abs(sparAd) + abs(A[:, nl:nl + dq.size]) + sum(abs(A[:, nl + dq.size + j]) @ sp.csc_matrix(([1.0], ([0], [int(np.searchsorted(blkstart[dq + 1], dencols[j], side="right"))])), shape=(1, dq.size)) for j in range(nden))) != 0).astype(np.float64))
    Ablk.sort_indices()
    dm = {"q1": d["q1"].reshape(-1, 1), "q2": d["q2"].reshape(-1, 1), "det": d["det"].reshape(-1, 1)}
    want = refmex.call("adendotd", 1, dense, dm, sparAd, Ablk, blkstart.reshape(-1, 1))
    got = mex.adendotd(dense, d, sparAd, Ablk, blkstart)
    assert np.array_equal(got.indices, want.indices) and np.array_equal(got.data, want.data)
    assert np.array_equal(mex.adenscale(dense, d, blkstart), refmex.call("adenscale", 1, dense, dm, blkstart.reshape(-1, 1)))


@pytest.mark.parametrize("seed,zero_d,maxuden", [(11, 0, 1.0), (12, 1, 1.2), (13, 3, 3.0), (14, 0, 1.05), (15, 5, 500.0), (16, 2, 1.0)])
def test_general_dpr1fact_on_the_device(refmex, glue, seed, zero_d, maxuden):
    """Postponed pivots (maxu close to 1: many rows fail the stability test and are sorted into a second round), dependent rows
    (d = 0: the partition branch, dpr1fact.c:371-476) and their combinations -- the data-dependent parts of dodpr1fact, which
    run on the device as scans / prefix counts / a bitonic sort (k_dpr1_general) -- against the compiled reference."""
    c = dense_case(refmex, glue, 110, 800, 5, seed, zero_d, maxuden)
    Lden_r = check_dense_case(refmex, c)
    if maxuden < 2.0:
        assert np.asarray(Lden_r["dopiv"]).sum() > 0
    check_resident_dense_unit(refmex, c, expect_host=False)


@pytest.mark.parametrize("seed,cfac", [(49, 4.0), (57, 4.0), (73, 4.0), (75, 0.9), (79, 0.9), (79, 1.5)])
def test_negative_multiple_brings_a_removed_dependency_back(refmex, glue, seed, cfac):
    """Dependent rows (d = 0) whose dependency an earlier column removes, then a LATE column with a negative multiple that drives
    that d back to <= 0: prodformfact calls findnewdep after EVERY column with smult < 0 (dpr1fact.c:575-576, :495-512), also when
    dodpr1fact took the natural-order path (:330-333).  (The cases are the ones of a seed sweep on which a device kernel that
    returned before findnewdep on that path differed from the reference: betajc / pivperm / d depend on dep[].)"""
    from oracle.refmex import RawSparse
    from sedumi_amd import mex
    c = dense_case(refmex, glue, 60, 400, 5, seed, 2, 500.0)
    r = c["sym_ref"]
    perm = np.asarray(r["perm"]).ravel().astype(int) - 1
    k0 = perm[len(perm) - 1 - (seed % 2)]
    pcol = np.asarray(c["LAD"][:, k0].todense()).ravel()
    c["smult"] = c["smult"].copy()
    c["smult"][k0] = -cfac / float(np.sum(pcol ** 2 / np.where(c["Ld"] > 0, c["Ld"], np.inf)))
    sref = {"dz": RawSparse(r["dz"]), "perm": r["perm"], "first": r["first"]}
    sym = mex.finsymbden(c["LADsym"], c["perm"], c["dz"], float(c["denseA"].shape[1] + 1))
    Lr, Ldr = refmex.call("dpr1fact", 2, c["LAD"], c["Ld"].reshape(-1, 1), sref, c["smult"].reshape(-1, 1), c["maxuden"])
    Lo, Ldo = mex.dpr1fact(c["LAD"], c["Ld"], sym, c["smult"], c["maxuden"])
    for key in ("betajc", "dopiv", "pivperm"):
        assert np.array_equal(np.asarray(Lo[key]).ravel(), np.asarray(Lr[key]).ravel()), key
    assert relerr(Lo["beta"], Lr["beta"]) < TOL and relerr(Lo["p"], Lr["p"]) < TOL and relerr(Ldo, Ldr) < TOL


@pytest.mark.parametrize("seed,which", [(21, 0), (22, 0), (23, 1), (24, 2)])
def test_dpr1fact_with_a_negative_multiple(refmex, glue, seed, which):
    """smult < 0 (the Lorentz trace columns of deninfac.m:62): D - |s| p p' with |s| small enough to stay positive definite;
    t starts negative (dpr1fact.c:293-300) and findnewdep (:495-512) runs after the step.  `which`: the position of that column in
    the factorisation order (later ones have been through the earlier factors' forward solves, which only shrink p' D^-1 p)."""
    c = dense_case(refmex, glue, 90, 700, 4, seed, 0, 500.0)
    r = c["sym_ref"]
    k0 = int(np.asarray(r["perm"]).ravel()[which]) - 1
    pcol = np.asarray(c["LAD"][:, k0].todense()).ravel()
    c["smult"] = c["smult"].copy()
    c["smult"][k0] = -0.5 / float(np.sum(pcol ** 2 / c["Ld"]))
    Lden_r = check_dense_case(refmex, c)
    assert np.asarray(Lden_r["betajc"]).ravel()[-1] > 1
    check_resident_dense_unit(refmex, c, expect_host=False)

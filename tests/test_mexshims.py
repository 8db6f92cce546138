"""The mexFunction shims of sedumi_amd/mexshims: compiled against the declaration-only MEX header
of the package's MEX host (sedumi_amd/mexhost), linked to the emulated build of the C ABI, and driven through the same mxArray marshalling as
the reference MEX -- so `prhs/plhs` handling, field lookups, 1-based conversions and sparse outputs are tested
end to end without MATLAB/Octave."""
import glob
import os
import subprocess

import numpy as np
import pytest
import scipy.sparse as sp

from helpers import ROOT, TOL, ref_scaling, relerr, spd_pattern

from sedumi_amd.build import SHIMS


def build_shims(lib, out):
    """Every mexFunction shim as a shared object of its own, linked to the C-ABI library `lib` and to the package's MEX host
    (sedumi_amd.build.build_mexshims: g++ only, MATLAB / Octave are not needed); driven through the same marshalling as the
    reference MEX."""
    from sedumi_amd import build as b
    b.build_mexshims(lib, out)
    from oracle.refmex import RefMex, REF_DIR
    return RefMex(REF_DIR, mex_dir=out)


@pytest.fixture(scope="module")
def shimlib():
    """The C-ABI library the shims are linked to: here the emulated build (tests/test_mexshims_gpu.py: libsedumi_hip.so)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))
    import build_emu
    return build_emu.build()


@pytest.fixture(scope="module")
def shimmex(refmex, shimlib):
    import ctypes
    ctypes.CDLL(shimlib).sdm_mexcache_set_lazy(0)      # these tests compare every gateway's arrays with the reference's (the library's default is level 2)
    return build_shims(shimlib, os.path.join(ROOT, "tests", "hipemu", "_mexshims"))


def test_every_hot_path_mex_has_a_shim():
    have = {os.path.basename(f)[:-4] for f in glob.glob(os.path.join(ROOT, "sedumi_amd", "mexshims", "*.cpp"))} - {"mexcommon"}
    assert set(SHIMS) <= have


def test_shims_reproduce_an_iteration_unit(glue, refmex, shimmex):
    from oracle import glue as gl
    from sedumi_amd import problem
    P = problem.random_sdp(m=28, seed=21)
    S = glue.setup(P.At, P.K)
    d, ud = ref_scaling(P, 2)
    it = glue.iteration_ref(S, d, ud)
    K = P.K
    dstruct = {"l": d["l"].reshape(-1, 1), "det": d["det"].reshape(-1, 1)}
    A1 = shimmex.call("getada1", 1, S["ADA"], S["A"], S["Ablkjc"][:, 2], S["Aord"]["lqperm"], dstruct, K["qblkstart"])
    assert relerr(A1, it["ADA1"]) < TOL
    A2 = shimmex.call("getada2", 1, A1, it["DAt"], S["Aord"], K)
    assert relerr(A2, it["ADA2"]) < TOL
    A3, absd = shimmex.call("getada3", 2, A2, S["A"], S["Ablkjc"][:, 2], S["Aord"], ud.reshape(-1, 1), K)
    assert relerr(A3, it["ADA"]) < TOL and relerr(absd, it["absd"]) < TOL
    pars = gl.default_pars_chol()
    LL, Ld, Lskip, Ladd = shimmex.call("blkchol", 4, S["L"], A3, pars, absd)
    assert relerr(LL, it["LL"]) < TOL and relerr(Ld, it["Ld"]) < TOL
    assert Lskip.nnz == it["Lskip"].nnz and Ladd.nnz == it["Ladd"].nnz
    L = dict(S["L"]); L["L"] = LL
    rhs = np.random.default_rng(0).standard_normal((P.m, 2))
    y = shimmex.call("bwblkslv", 1, L, shimmex.call("fwblkslv", 1, L, rhs) / Ld)
    yr = refmex.call("bwblkslv", 1, L, refmex.call("fwblkslv", 1, L, rhs) / Ld)
    assert relerr(y, yr) < TOL
    only_L = shimmex.call("blkchol", 1, S["L"], A3, pars, absd)          # nlhs = 1: other outputs destroyed
    assert relerr(only_L, it["LL"]) < TOL


def test_shims_symbolic_bit_exact(refmex, shimmex):
    rng = np.random.default_rng(4)
    X = spd_pattern("rand", 90, rng, 0.05)
    pr = refmex.call("ordmmdmex", 1, X)
    assert np.array_equal(shimmex.call("ordmmdmex", 1, X), pr)
    Lr, Lo = refmex.call("symfctmex", 1, X, pr), shimmex.call("symfctmex", 1, X, pr)
    assert np.array_equal(Lo["perm"], Lr["perm"]) and np.array_equal(Lo["xsuper"], Lr["xsuper"])
    assert np.array_equal(Lo["L"].indices, Lr["L"].indices) and np.array_equal(Lo["L"].indptr, Lr["L"].indptr)
    assert np.array_equal(shimmex.call("choltmpsiz", 1, Lr), refmex.call("choltmpsiz", 1, Lr))
    assert np.array_equal(shimmex.call("cholsplit", 1, Lr, 0.3), refmex.call("cholsplit", 1, Lr, 0.3))


def test_shim_incorder(refmex, shimmex):
    rng = np.random.default_rng(8)
    At = sp.random(120, 50, density=0.08, random_state=rng, format="csc"); At.sort_indices()
    Ajc1 = np.array([At.indptr[j] + np.searchsorted(At.indices[At.indptr[j]:At.indptr[j + 1]], 30) for j in range(50)], dtype=np.float64)
    for args in ((At,), (At, Ajc1.reshape(-1, 1), 31.0)):
        (pr, dzr), (po, dzo) = refmex.call("incorder", 2, *args), shimmex.call("incorder", 2, *args)
        assert np.array_equal(po, pr) and np.array_equal(dzo.indptr, dzr.indptr) and np.array_equal(dzo.indices, dzr.indices)


def test_shims_dense_column_path(refmex, glue, shimmex):
    """symbfwblk, finsymbden, dpr1fact, fwdpr1, bwdpr1 through their mexFunction shims (struct outputs, 1-based
    conversions, the unsorted `dz`) against the reference gateways."""
    from oracle.refmex import RawSparse
    from test_dense_columns import dense_case
    c = dense_case(refmex, glue, 70, 500, 4, 9)
    X = shimmex.call("symbfwblk", 1, c["L"], c["denseA"])
    assert np.array_equal(X.indptr, c["LADsym"].indptr) and np.array_equal(X.indices, c["LADsym"].indices)
    r = c["sym_ref"]
    sym = shimmex.call("finsymbden", 1, c["LADsym"], c["perm"], RawSparse(c["dz"]), 5.0)
    assert np.array_equal(sym["perm"], r["perm"]) and np.array_equal(sym["first"], r["first"])
    assert np.array_equal(sym["dz"].indptr, r["dz"].indptr) and np.array_equal(sym["dz"].indices, r["dz"].indices)
    sref = {"dz": RawSparse(r["dz"]), "perm": r["perm"], "first": r["first"]}
    args = (c["LAD"], c["Ld"].reshape(-1, 1), sref, c["smult"].reshape(-1, 1), c["maxuden"])
    (Lr, Ldr), (Lo, Ldo) = refmex.call("dpr1fact", 2, *args), shimmex.call("dpr1fact", 2, *args)
    for k in ("betajc", "dopiv", "pivperm"):
        assert np.array_equal(Lo[k], Lr[k])
    assert relerr(Lo["p"], Lr["p"]) < TOL and relerr(Lo["beta"], Lr["beta"]) < TOL and relerr(Ldo, Ldr) < TOL
    Lr2 = dict(Lr); Lr2["dz"] = RawSparse(r["dz"])
    b = np.random.default_rng(1).standard_normal((c["Ld"].size, 2))
    assert relerr(shimmex.call("fwdpr1", 1, Lr2, b), refmex.call("fwdpr1", 1, Lr2, b)) < TOL
    assert relerr(shimmex.call("bwdpr1", 1, Lr2, b), refmex.call("bwdpr1", 1, Lr2, b)) < TOL
    assert np.array_equal(shimmex.call("fwdpr1", 1, {"betajc": np.array([[1.0]])}, b), b)     # no dense columns


def test_shim_errors_go_through_mexErrMsgTxt(shimmex):
    from oracle.refmex import RefMexError
    L = {"L": sp.csc_matrix(np.tril(np.ones((4, 4)))), "perm": np.arange(1, 5.0)}
    with pytest.raises(RefMexError, match="Missing field L.xsuper"):
        shimmex.call("fwblkslv", 1, L, np.ones((4, 1)))


def test_shim_invcholfac(refmex, shimmex):
    """y = invcholfac(u, K, perm) through its mexFunction shim against the reference gateway (real + Hermitian blocks,
    with and without the permutation argument)."""
    from test_invcholfac import scaling_factor_case
    from sedumi_amd import problem
    K = problem.make_K(1, [], [70, 5], hs=[9])
    u, perm = scaling_factor_case(K, seed=3)
    for args in ((u.reshape(-1, 1), K, perm.reshape(-1, 1)), (u.reshape(-1, 1), K)):
        assert relerr(shimmex.call("invcholfac", 1, *args), refmex.call("invcholfac", 1, *args)) < TOL


def _factor_probe(lib, L, m):
    """resident(pattern, values) -> does sdm_mexcache_factor_plan take the host arrays GIVEN (no copies: the addresses are the caller's)
    for the factor the device holds?"""
    import ctypes
    lib.sdm_mexcache_factor_plan.restype = ctypes.c_void_p
    perm = (np.asarray(L["perm"]).ravel() - 1).astype(np.int64)
    xs = (np.asarray(L["xsuper"]).ravel() - 1).astype(np.int64)
    P = ctypes.POINTER(ctypes.c_int64)

    def resident(jc, ir, pr):
        assert jc.dtype == np.int64 and ir.dtype == np.int64 and pr.dtype == np.float64
        return bool(lib.sdm_mexcache_factor_plan(ctypes.c_int64(m), jc.ctypes.data_as(P), ir.ctypes.data_as(P),
                                                 pr.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), perm.ctypes.data_as(P),
                                                 ctypes.c_int64(xs.size - 1), xs.ctypes.data_as(P)))
    return resident


def _csc_arrays(LLv):
    LL = sp.csc_matrix(LLv); LL.sort_indices()
    return LL.indptr.astype(np.int64), LL.indices.astype(np.int64), np.array(LL.data, dtype=np.float64)


def test_factor_cache_is_shared_between_mex_binaries_and_validated_by_content(glue, refmex, shimmex, shimlib):
    """blkchol.mex leaves the factor resident in the plan cached inside the library (sdm_mexcache_*); fwblkslv.mex /
    bwblkslv.mex -- separate shared objects -- reuse it only when the L.L values they are handed ARE that factor.
    Two blkchol calls on the same symbolic factor, then solves with the FIRST factor's values: the cache holds the
    second factor, so the content check must reject it and the stateless path must give the first factor's answer.
    Deterministic whatever the allocator does: a copy of the resident factor is accepted at ITS address, then one word of that
    very array (one the 512-word sample does not look at) is changed in place -- an address the cache has accepted content at
    -- and must be rejected: arrays of this size are checksummed completely at every presentation."""
    import ctypes
    from oracle import glue as gl
    lib = ctypes.CDLL(shimlib)
    lib.sdm_mexcache_set_strict(0)
    lib.sdm_mexcache_set_full_below(ctypes.c_int64(1 << 16))
    rng = np.random.default_rng(12)
    X1 = spd_pattern("rand", 120, rng, 0.05)
    X2 = sp.csc_matrix(X1 + sp.diags(rng.random(120) + 0.5)); X2.sort_indices()
    L = glue.symbchol(X1)
    pars = gl.default_pars_chol()
    LL1, Ld1, _, _ = shimmex.call("blkchol", 4, L, X1, pars)
    LL2, Ld2, _, _ = shimmex.call("blkchol", 4, L, X2, pars)
    assert relerr(LL2, refmex.call("blkchol", 4, L, X2, pars)[0]) < TOL
    rhs = rng.standard_normal((120, 2))
    resident = _factor_probe(lib, L, 120)
    jc, ir, pr2 = _csc_arrays(LL2)
    _, _, pr1 = _csc_arrays(LL1)
    assert resident(jc, ir, pr2) and not resident(jc, ir, pr1)     # the second factor is resident, the first is not
    k = pr2.size // 3
    assert pr2.size > 512 and k % -(-pr2.size // 512) != 0         # (a word the sampled hash does not look at)
    for _ in range(3):
        assert resident(jc, ir, pr2)                                # accepted at this address ...
        keep = pr2[k]; pr2[k] += 1e-9
        assert not resident(jc, ir, pr2)                            # ... and the same address with one value changed is not
        pr2[k] = keep
    keep = ir[k]; ir[k] = ir[k - 1] if ir[k - 1] != ir[k] else ir[k] + 1   # the pattern likewise
    assert not resident(jc, ir, pr2)
    ir[k] = keep
    assert resident(jc, ir, pr2)
    for LLv in (LL2, LL1):                                    # resident path, then content-check fallback
        Lf = dict(L); Lf["L"] = LLv
        assert relerr(shimmex.call("fwblkslv", 1, Lf, rhs), refmex.call("fwblkslv", 1, Lf, rhs)) < TOL
        assert relerr(shimmex.call("bwblkslv", 1, Lf, rhs), refmex.call("bwblkslv", 1, Lf, rhs)) < TOL
    lib.sdm_mexcache_clear()
    assert not resident(jc, ir, pr2)


def test_factor_cache_shortcut_for_large_arrays_is_exactly_the_documented_one(glue, refmex, shimmex, shimlib):
    """sdm_mexcache.hip's header, DESIGN 1a, INTEGRATION.md: an array LARGER than `full_below` words is checksummed completely the
    first time an address presents it in an epoch (= between two blkchol calls); later presentations at that address in that epoch
    pass on length + the sampled hash.  So (the threshold lowered to make this factor `large`):
      * a changed copy at an address the cache has never accepted is rejected -- always;
      * a word the sample looks at, changed in place at a trusted address, is rejected;
      * a word the sample does NOT look at, changed in place at a trusted address inside the epoch, is NOT noticed (the documented
        limit) -- and IS noticed in strict mode, and after the next blkchol (new epoch: nothing is trusted any more)."""
    import ctypes
    from oracle import glue as gl
    lib = ctypes.CDLL(shimlib)
    lib.sdm_mexcache_set_strict(0)
    rng = np.random.default_rng(13)
    X = spd_pattern("rand", 120, rng, 0.05)
    L = glue.symbchol(X)
    pars = gl.default_pars_chol()
    try:
        lib.sdm_mexcache_set_full_below(ctypes.c_int64(100))
        LL, _, _, _ = shimmex.call("blkchol", 4, L, X, pars)
        resident = _factor_probe(lib, L, 120)
        jc, ir, pr = _csc_arrays(LL)
        n = pr.size
        step = -(-n // 512)
        k = n // 3
        assert n > 512 and k % step != 0
        other = pr.copy(); other[k] += 1e-9
        assert not resident(jc, ir, other)                          # an address never accepted: complete checksum, rejected
        assert resident(jc, ir, pr)                                 # complete checksum at this address: trusted for the epoch
        pr[step * 5] += 1e-9
        assert not resident(jc, ir, pr)                             # a sampled word
        pr[step * 5] -= 1e-9
        keep = pr[k]; pr[k] += 1e-9
        assert resident(jc, ir, pr)                                 # THE LIMIT: unsampled word, trusted address, same epoch
        lib.sdm_mexcache_set_strict(1)
        assert not resident(jc, ir, pr)                             # strict: every presentation is checksummed completely
        lib.sdm_mexcache_set_strict(0)
        pr[k] = keep
        # a new epoch with the same factor values on the device: the edited array has to present its complete content again
        LLb, _, _, _ = shimmex.call("blkchol", 4, L, X, pars)
        assert np.array_equal(_csc_arrays(LLb)[2], pr)
        pr[k] += 1e-9
        assert not resident(jc, ir, pr)
        pr[k] = keep
        assert resident(jc, ir, pr)
    finally:
        lib.sdm_mexcache_set_full_below(ctypes.c_int64(1 << 16))
        lib.sdm_mexcache_set_strict(0)
        lib.sdm_mexcache_clear()


def test_getada_shim_updates_the_global(glue, refmex, shimmex):
    """getada.mex shadows getada.m (sedumi.m:446-448, problems without PSD blocks): it reads the pattern of the GLOBAL
    ADA_sedumi_ (mexGetVariablePtr), writes ADA' back into it (mexPutVariable) and returns absd = diag(ADA')."""
    from oracle.refmex import RefMexError
    from sedumi_amd import problem
    P = problem.random_sdp(m=30, lp=7, q=(4, 5, 3), s=(), seed=33)
    S = glue.setup(P.At, P.K)
    d, ud = ref_scaling(P, 6)
    it = glue.iteration_ref(S, d, ud)
    dstruct = {"l": d["l"].reshape(-1, 1), "det": d["det"].reshape(-1, 1)}
    shimmex.set_global("ADA_sedumi_", S["ADA"])
    absd = shimmex.call("getada", 1, S["A"], P.K, dstruct, it["DAt"])
    ADA = shimmex.get_global("ADA_sedumi_")
    assert relerr(ADA, it["ADA"]) < TOL and relerr(absd.ravel(), it["ADA"].diagonal()) < TOL
    # a full (dense) DAt.q is accepted too (getada.m:21-24 densifies it itself)
    shimmex.set_global("ADA_sedumi_", S["ADA"])
    absd2 = shimmex.call("getada", 1, S["A"], P.K, dstruct, {"q": sp.csc_matrix(it["DAt"]["q"]).toarray()})
    assert relerr(shimmex.get_global("ADA_sedumi_"), it["ADA"]) < TOL and relerr(absd2, absd) < 1e-14
    shimmex.set_global("ADA_sedumi_", None)
    with pytest.raises(RefMexError, match="ADA_sedumi_ does not exist"):
        shimmex.call("getada", 1, S["A"], P.K, dstruct, it["DAt"])


def mexcache_stats(shimlib):
    import ctypes
    lib = ctypes.CDLL(shimlib)
    out = (ctypes.c_int64 * 16)()
    lib.sdm_mexcache_stats(out, ctypes.c_int64(16))
    keys = ("ada_build", "ada_reuse", "ada_upload", "ada_resident", "chol_build", "chol_reuse", "x_upload", "x_resident",
            "solve_resident", "solve_stateless", "at_upload")
    return dict(zip(keys, list(out)))


def run_units_by_reference(host, S, K, d, DAt, ud, pars, rhs, nunits, ref_it=None):
    """`nunits` iteration units through the mexFunction shims, arrays handed on BY REFERENCE (sedumi_amd.mexhost.iteration_units),
    each unit's ADA', absd, L.L, L.d checked against the reference's when ref_it is given."""
    from sedumi_amd.mexhost import iteration_units
    dstruct = {"l": np.asarray(d["l"]).reshape(-1, 1), "det": np.asarray(d["det"]).reshape(-1, 1)}

    def check(A3, absd, LL, Ld):
        assert relerr(A3, ref_it["ADA"]) < TOL and relerr(absd, ref_it["absd"]) < TOL
        assert relerr(LL, ref_it["LL"]) < TOL and relerr(Ld, ref_it["Ld"]) < TOL
    times, y = iteration_units(host, S["A"], S["Ablkjc"][:, 2], S["Aord"], K, dstruct, DAt, ud, S["L"], S["ADA"], pars, rhs, nunits,
                               check=check if ref_it is not None else None)
    return [sum(t.values()) for t in times], y


@pytest.mark.parametrize("lorentz", [True, False])
def test_iteration_units_by_reference_reuse_the_device_state(glue, refmex, shimmex, shimlib, lorentz):
    """(lorentz = False: getada2.mex has nothing to add and returns a COPY of its input, getada2.c:153-155 -- the copy must be the same
    ADA' to the cache, values and pattern.)  What an unmodified sedumi.m does every iteration: the same At / K / patterns, each gateway's output handed to the next.  From
    the second unit on nothing is analysed again (no ada_build, no chol_build), no ADA' values and no factor cross PCIe towards the
    device (ADA resident for getada2 / getada3 / blkchol, L.L resident for the eight solves)."""
    import ctypes
    from oracle import glue as gl
    from sedumi_amd import problem
    ctypes.CDLL(shimlib).sdm_mexcache_clear()
    P = problem.random_sdp(m=28, seed=21) if lorentz else problem.random_sdp(m=28, lp=4, q=(), s=(6, 5), seed=22)
    S = glue.setup(P.At, P.K)
    d, ud = ref_scaling(P, 2)
    it = glue.iteration_ref(S, d, ud)
    pars = gl.default_pars_chol()
    rhs = np.random.default_rng(0).standard_normal(P.m)
    assert (P.K["q"].size > 0) == lorentz
    s0 = mexcache_stats(shimlib)
    _, y = run_units_by_reference(shimmex, S, P.K, d, it["DAt"], ud, pars, rhs, 1, it)
    s1 = mexcache_stats(shimlib)
    _, y2 = run_units_by_reference(shimmex, S, P.K, d, it["DAt"], ud, pars, rhs, 2, it)
    s2 = mexcache_stats(shimlib)
    L = dict(S["L"]); L["L"] = it["LL"]
    yr = refmex.call("bwblkslv", 1, L, refmex.call("fwblkslv", 1, L, rhs.reshape(-1, 1)) / np.where(it["Ld"] > 0, it["Ld"], 1.0))
    assert relerr(y, yr.ravel()) < TOL and relerr(y2, yr.ravel()) < TOL
    nq = int(P.K["q"].size > 0)                                   # getada2 runs on the device only when there are Lorentz cones
    assert s1["ada_build"] - s0["ada_build"] == 2 + nq and s1["chol_build"] - s0["chol_build"] == 1
    # the later units: the problem data are re-marshalled by the test (new addresses, same content): recognised by content
    assert s2["ada_build"] == s1["ada_build"] and s2["chol_build"] == s1["chol_build"], (s1, s2)
    assert s2["at_upload"] == s1["at_upload"]
    assert s2["ada_upload"] == s1["ada_upload"] and s2["x_upload"] == s1["x_upload"], (s1, s2)
    assert s2["ada_resident"] - s1["ada_resident"] == 2 * (1 + nq) and s2["x_resident"] - s1["x_resident"] == 2
    assert s2["solve_resident"] - s1["solve_resident"] == 16 and s2["solve_stateless"] == s1["solve_stateless"]


@pytest.mark.parametrize("level,lorentz", [(1, True), (1, False), (2, True), (2, False)])
def test_lazy_intermediates_leave_ada_on_the_device(glue, refmex, shimmex, shimlib, level, lorentz):
    """Opt-in (SEDUMI_HIP_LAZY / sdm_mexcache_set_lazy): getada1.mex / getada2.mex return a token instead of ADA' (level 2: getada3.mex too)
    and the next gateway takes the device's values.  What sedumi.m uses is unchanged: level 1 -- ADA' and absd after getada3, L.L, L.d and
    the solves are the reference's; level 2 -- absd, L.L, L.d and the solves.  No ADA' value crosses PCIe towards the device, and a token
    that is not the current one is refused loudly.  (lorentz = False: getada2.mex has nothing to add and copies the token, getada2.c:153-155.)"""
    import ctypes
    from oracle import glue as gl
    from oracle.refmex import RefMexError
    from sedumi_amd import problem
    from sedumi_amd.mexhost import iteration_units
    lib = ctypes.CDLL(shimlib)
    lib.sdm_mexcache_token_base.restype = ctypes.c_double
    lib.sdm_mexcache_clear()
    P = problem.random_sdp(m=28, seed=21) if lorentz else problem.random_sdp(m=28, lp=4, q=(), s=(6, 5), seed=22)
    S = glue.setup(P.At, P.K)
    d, ud = ref_scaling(P, 2)
    it = glue.iteration_ref(S, d, ud)
    pars = gl.default_pars_chol()
    rhs = np.random.default_rng(0).standard_normal(P.m)
    dstruct = {"l": np.asarray(d["l"]).reshape(-1, 1), "det": np.asarray(d["det"]).reshape(-1, 1)}
    seen = []

    def check(A3, absd, LL, Ld):
        seen.append(A3)
        if level == 1:
            assert relerr(A3, it["ADA"]) < TOL
        else:
            A3 = sp.csc_matrix(A3)
            assert A3.nnz == 1 and A3[0, 0] > lib.sdm_mexcache_token_base()        # a token, not ADA'
        assert relerr(absd, it["absd"]) < TOL and relerr(LL, it["LL"]) < TOL and relerr(Ld, it["Ld"]) < TOL
    try:
        lib.sdm_mexcache_set_lazy(level)
        s0 = mexcache_stats(shimlib)
        _, y = iteration_units(shimmex, S["A"], S["Ablkjc"][:, 2], S["Aord"], P.K, dstruct, it["DAt"], ud, S["L"], S["ADA"], pars, rhs, 3, check=check)
        s1 = mexcache_stats(shimlib)
        L = dict(S["L"]); L["L"] = it["LL"]
        yr = refmex.call("bwblkslv", 1, L, refmex.call("fwblkslv", 1, L, rhs.reshape(-1, 1)) / np.where(it["Ld"] > 0, it["Ld"], 1.0))
        assert relerr(y, yr.ravel()) < TOL and len(seen) == 3
        assert s1["ada_upload"] == s0["ada_upload"] and s1["x_upload"] - s0["x_upload"] == 0, (s0, s1)
        # a token of an earlier call is not the current one: refused, not silently wrong
        tok = shimmex.call("getada1", 1, S["ADA"], S["A"], S["Ablkjc"][:, 2], S["Aord"]["lqperm"], dstruct, P.K["qblkstart"])
        assert sp.csc_matrix(tok).nnz == 1
        shimmex.call("getada1", 1, S["ADA"], S["A"], S["Ablkjc"][:, 2], S["Aord"]["lqperm"], dstruct, P.K["qblkstart"])   # ... a newer one exists now
        with pytest.raises(RefMexError, match="not the current one"):
            shimmex.call("getada3", 2, tok, S["A"], S["Ablkjc"][:, 2], S["Aord"], ud.reshape(-1, 1), P.K)
    finally:
        lib.sdm_mexcache_set_lazy(0)
        lib.sdm_mexcache_clear()


def test_checksum_threads_give_the_value_of_one_thread(shimlib):
    """sdm_mexcache_checksum: arrays from 128K words on are summed by a few helper threads that sleep in between (eight wrap-around sums:
    any partition gives the same value); clearing the cache joins them, the next big array starts them again."""
    import ctypes
    lib = ctypes.CDLL(shimlib)
    lib.sdm_mexcache_checksum.restype = ctypes.c_uint64
    a = np.random.default_rng(3).standard_normal(300_001)
    p = a.ctypes.data_as(ctypes.c_void_p)
    vals = []
    for nthreads in (1, 4, 3, -1):
        lib.sdm_mexcache_set_threads(nthreads)
        vals.append(lib.sdm_mexcache_checksum(p, ctypes.c_int64(a.size)))
        lib.sdm_mexcache_clear()                                       # joins the helpers
    lib.sdm_mexcache_set_threads(-1)
    assert len(set(vals)) == 1
    a[123_457] += 1e-12
    assert lib.sdm_mexcache_checksum(p, ctypes.c_int64(a.size)) != vals[0]
    lib.sdm_mexcache_clear()


def test_units_on_a_problem_without_lp_or_lorentz_part(glue, refmex, shimmex, shimlib):
    """MAXCUT's shape: one PSD block, nothing else.  getada1.mex has nothing to add (ADA' stays the zero matrix: no sparse dots over
    its pattern), getada2.mex copies, getada3.mex is handed the device's zero matrix and skips the symmetrisation of its input -- ADA',
    absd, the factor and the solves are the reference's, unit after unit."""
    import ctypes
    from oracle import glue as gl
    from sedumi_amd import problem
    ctypes.CDLL(shimlib).sdm_mexcache_clear()
    P = problem.maxcut(14)
    S = glue.setup(P.At, P.K)
    assert P.K["q"].size == 0 and np.all(np.asarray(S["Ablkjc"])[:, 2] == sp.csc_matrix(S["A"]).indptr[:-1])     # no LP / Lorentz nonzero in any constraint
    d, ud = ref_scaling(P, 4)
    it = glue.iteration_ref(S, d, ud)
    assert abs(it["ADA1"]).sum() == 0                               # (the reference's getada1 leaves zeros, too)
    pars = gl.default_pars_chol()
    rhs = np.random.default_rng(1).standard_normal(P.m)
    s0 = mexcache_stats(shimlib)
    _, y = run_units_by_reference(shimmex, S, P.K, d, it["DAt"], ud, pars, rhs, 3, it)
    s1 = mexcache_stats(shimlib)
    L = dict(S["L"]); L["L"] = it["LL"]
    yr = refmex.call("bwblkslv", 1, L, refmex.call("fwblkslv", 1, L, rhs.reshape(-1, 1)) / np.where(it["Ld"] > 0, it["Ld"], 1.0))
    assert relerr(y, yr.ravel()) < TOL
    assert s1["ada_upload"] == s0["ada_upload"] and s1["ada_resident"] - s0["ada_resident"] == 3
    # (an ADA' with values handed to getada3.mex -- uploaded, symmetrised as before -- is test_shims_reproduce_an_iteration_unit's case)


def test_lazy_level_2_is_the_default_and_the_environment_turns_it_off(shimlib):
    """sdm_mexcache_set_lazy(-1) = "as SEDUMI_HIP_LAZY says"; unset means level 2 (every reference call site hands ADA' on untouched)."""
    import subprocess
    import sys
    code = ("import ctypes,sys; lib = ctypes.CDLL(sys.argv[1]); lib.sdm_mexcache_set_lazy(-1); print(lib.sdm_mexcache_lazy())")
    env = {k: v for k, v in os.environ.items() if k != "SEDUMI_HIP_LAZY"}
    assert subprocess.check_output([sys.executable, "-c", code, shimlib], env=env).split()[-1] == b"2"
    for val in ("0", "1"):
        assert subprocess.check_output([sys.executable, "-c", code, shimlib], env=dict(env, SEDUMI_HIP_LAZY=val)).split()[-1] == val.encode()


@pytest.mark.parametrize("name", ["quantum"])           # (tests/test_mexshims_gpu.py: + nb, arch0, control07)
def test_whole_solves_through_the_shims_are_the_same_at_lazy_level_0_and_2(shimmex, shimlib, name):
    check_whole_solve_at_lazy_levels(shimmex, shimlib, name)


def check_whole_solve_at_lazy_levels(shimmex, shimlib, name):
    """the default of the drop-in tier.  A whole interior-point solve (tests/driver, sedumi.m's loop) with its hot path through the
    built mexFunction shims, ADA_sedumi_ handed from iteration to iteration the way sedumi.m:450-452 does -- once with every gateway returning the
    reference's arrays (level 0), once with getada1 / getada2 / getada3 returning tokens (level 2): the same iteration log, bit for bit (the device
    does the same arithmetic; only what crosses PCIe differs), and the optimal value of examples/test_sedumi.m."""
    import ctypes
    import test_driver as td
    from driver import sedumi_loop as sl
    lib = ctypes.CDLL(shimlib)
    runs = {}
    try:
        for level in (0, 2):
            lib.sdm_mexcache_clear()
            lib.sdm_mexcache_set_lazy(level)
            runs[level] = td.run(name, sl.ShimHot(shimmex))
    finally:
        lib.sdm_mexcache_set_lazy(0)
        lib.sdm_mexcache_clear()
    a, b = runs[0], runs[2]
    assert a["hot"] == b["hot"] == "sedumi_amd.mexshims" and a["iter"] > 3
    td.check_objectives(name, b)
    assert a["iter"] == b["iter"] and len(a["rows"]) == len(b["rows"])
    for ra, rb in zip(a["rows"], b["rows"]):
        assert ra == rb, (ra, rb)
    assert a["cx"] == b["cx"] and a["by"] == b["by"]
    td.check_log(name, b, td.reference_run(name))


def test_lazy_tokens_stale_consumed_and_one_by_one(glue, refmex, shimmex, shimlib):
    """  A 1 x 1 problem never gets a token (it could not be told from a genuine ADA'); a token is consumed when real arrays come back
    for it (re-presenting it is refused); getada1.mex -- which needs only the PATTERN of what it is given -- accepts a token that is no
    longer the current one (sedumi.m's global after optstep.m:68-76 ran the gateways on its own copy)."""
    import ctypes
    from oracle.refmex import RefMexError
    from sedumi_amd import problem
    lib = ctypes.CDLL(shimlib)
    lib.sdm_mexcache_token_base.restype = ctypes.c_double
    P = problem.random_sdp(m=28, seed=21)
    S = glue.setup(P.At, P.K)
    d, ud = ref_scaling(P, 2)
    it = glue.iteration_ref(S, d, ud)
    dstruct = {"l": np.asarray(d["l"]).reshape(-1, 1), "det": np.asarray(d["det"]).reshape(-1, 1)}
    a1 = (S["A"], S["Ablkjc"][:, 2], S["Aord"]["lqperm"], dstruct, P.K["qblkstart"])
    a3 = (S["A"], S["Ablkjc"][:, 2], S["Aord"], ud.reshape(-1, 1), P.K)
    try:
        lib.sdm_mexcache_clear()
        lib.sdm_mexcache_set_lazy(1)
        t1 = shimmex.call("getada1", 1, S["ADA"], *a1)
        t2 = shimmex.call("getada2", 1, t1, it["DAt"], S["Aord"], P.K)
        ADA, absd = shimmex.call("getada3", 2, t2, *a3)                      # level 1: materialises
        assert relerr(ADA, it["ADA"]) < TOL and relerr(absd, it["absd"]) < TOL
        with pytest.raises(RefMexError, match="not the current one"):          # t2 was consumed: not added a second time
            shimmex.call("getada3", 2, t2, *a3)
        lib.sdm_mexcache_set_lazy(2)
        t1 = shimmex.call("getada1", 1, S["ADA"], *a1)
        t3, _ = shimmex.call("getada3", 2, shimmex.call("getada2", 1, t1, it["DAt"], S["Aord"], P.K), *a3)
        t1b = shimmex.call("getada1", 1, t3, *a1)                              # the global of the iteration before: current token in
        t3b, absd2 = shimmex.call("getada3", 2, shimmex.call("getada2", 1, t1b, it["DAt"], S["Aord"], P.K), *a3)
        assert sp.csc_matrix(t3b).nnz == 1 and relerr(absd2, it["absd"]) < TOL
        t1c = shimmex.call("getada1", 1, t3, *a1)                              # t3 is stale by now: still good for its pattern
        _, absd3 = shimmex.call("getada3", 2, shimmex.call("getada2", 1, t1c, it["DAt"], S["Aord"], P.K), *a3)
        assert relerr(absd3, it["absd"]) < TOL
        with pytest.raises(RefMexError, match="not the current one"):          # ... but not for its values
            shimmex.call("getada2", 1, t3, it["DAt"], S["Aord"], P.K)
        # m = 1: real arrays at every level
        P1 = problem.random_sdp(m=1, lp=3, q=(), s=(3,), seed=5)
        S1 = glue.setup(P1.At, P1.K)
        d1, ud1 = ref_scaling(P1, 2)
        it1 = glue.iteration_ref(S1, d1, ud1)
        ds1 = {"l": np.asarray(d1["l"]).reshape(-1, 1), "det": np.asarray(d1["det"]).reshape(-1, 1)}
        A1 = shimmex.call("getada1", 1, S1["ADA"], S1["A"], S1["Ablkjc"][:, 2], S1["Aord"]["lqperm"], ds1, P1.K["qblkstart"])
        assert float(sp.csc_matrix(A1)[0, 0]) < lib.sdm_mexcache_token_base()
        A2 = shimmex.call("getada2", 1, A1, it1["DAt"], S1["Aord"], P1.K)
        A3, ab = shimmex.call("getada3", 2, A2, S1["A"], S1["Ablkjc"][:, 2], S1["Aord"], ud1.reshape(-1, 1), P1.K)
        assert relerr(A3, it1["ADA"]) < TOL and relerr(ab, it1["absd"]) < TOL
    finally:
        lib.sdm_mexcache_set_lazy(0)
        lib.sdm_mexcache_clear()

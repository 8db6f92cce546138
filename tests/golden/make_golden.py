"""tests/golden/make_golden.py -- regenerates the golden fixtures from the reference's own example problems.

Runs ONLY in the build container (needs /root/reference and oracle/_ref): loads examples/{arch0,control07,nb}.mat
(the problems of examples/test_sedumi.m:22-28; nb.mat is the Lorentz-cone example, BASELINE.json configs[2]), applies the restated pretransfo / setup glue, runs the
UNMODIFIED reference MEX (getada1/2/3, blkchol, fwblkslv, bwblkslv) and stores the hot-path inputs plus
reference outputs as compressed .npz so that the parity tests can run where /root/reference does not exist.
ADA' (upper triangle) and L (lower triangle) are stored IN FULL (control07: for the "rand" scaling only, 2 x 1.8 MB),
next to their Frobenius norms and a fixed random sample of entries.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import scipy.io as sio
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
from oracle import glue  # noqa: E402


def main():
    G = glue.Glue()
    rng = np.random.default_rng(2026)
    for name in ("arch0", "control07", "nb"):
        d = sio.loadmat(f"/root/reference/examples/{name}.mat")
        K = {k: d["K"][k][0, 0].astype(float).ravel() for k in d["K"].dtype.names}
        At, b, c, Ki = glue.pretransfo_real(d["At"], d["b"], d["c"], K)
        S = G.setup(At, Ki)
        m = At.shape[1]
        A = sp.csc_matrix(S["A"])
        out = {"At_data": A.data, "At_indices": A.indices.astype(np.int32), "At_indptr": A.indptr.astype(np.int64),
               "At_shape": np.array(A.shape), "Ablkjc": S["Ablkjc"], "lqperm": S["Aord"]["lqperm"], "qperm": S["Aord"]["qperm"],
               "sperm": S["Aord"]["sperm"], "K_l": Ki["l"], "K_q": Ki["q"], "K_s": Ki["s"], "K_blkstart": Ki["blkstart"],
               "nsuper": S["L"]["xsuper"].size - 1, "ADA_nnz": S["ADA"].nnz}
        assert S["ADA"].nnz == m * m and S["L"]["xsuper"].size == 2, "expected the dense shortcut (symbchol.m:75-77)"
        rhs = rng.standard_normal(m)
        out["rhs"] = rhs
        for tag, (dd, ud) in (("init", glue.sdinit_scaling(Ki, b, c)), ("rand", glue.random_scaling(Ki, seed=7, cond=1e4))):
            it = G.iteration_ref(S, dd, ud)
            if not Ki["s"].size:
                # no PSD blocks: absd = diag(ADA') as getada3.c:549-552 documents it (the compiled cpspdiag is
                # undefined behaviour there, see tests/helpers.py check_iteration); blkchol re-run with it
                it["absd"] = it["ADA"].diagonal().reshape(-1, 1)
                it["LL"], it["Ld"], it["Lskip"], it["Ladd"] = G.ref.call("blkchol", 4, S["L"], it["ADA"],
                                                                           glue.default_pars_chol(), it["absd"])
            Q = sp.csc_matrix(it["DAt"]["q"]); Q.sort_indices()
            out.update({f"{tag}_DAtq_data": Q.data, f"{tag}_DAtq_indices": Q.indices.astype(np.int32),
                        f"{tag}_DAtq_indptr": Q.indptr.astype(np.int64), f"{tag}_DAtq_shape": np.array(Q.shape),
                        f"{tag}_ADA2_fro": np.linalg.norm(it["ADA2"].toarray())})
            y = G.solve_ref(S, it, rhs)
            Ld = it["Ld"].ravel()
            L = dict(S["L"]); L["L"] = it["LL"]
            yfw = G.ref.call("fwblkslv", 1, L, rhs.reshape(-1, 1)).ravel()
            ADA = it["ADA"].toarray(); LL = it["LL"].toarray()
            si = rng.integers(0, m, size=400); sj = rng.integers(0, m, size=400)
            lo = np.maximum(si, sj), np.minimum(si, sj)
            out.update({f"{tag}_dl": dd["l"], f"{tag}_ddet": dd["det"], f"{tag}_udsqr": ud.astype(np.float64),
                        f"{tag}_absd": it["absd"].ravel(), f"{tag}_Ld": Ld, f"{tag}_y": y.ravel(), f"{tag}_yfw": yfw,
                        f"{tag}_ADA_fro": np.linalg.norm(ADA), f"{tag}_L_fro": np.linalg.norm(LL),
                        f"{tag}_si": si, f"{tag}_sj": sj, f"{tag}_ADA_s": ADA[si, sj], f"{tag}_L_s": LL[lo[0], lo[1]],
                        f"{tag}_nskip": it["Lskip"].nnz, f"{tag}_nadd": it["Ladd"].nnz,
                        f"{tag}_ADA1_fro": np.linalg.norm(it["ADA1"].toarray()), f"{tag}_ADA_diag": np.diag(ADA)})
            if name != "control07" or tag == "rand":
                assert np.array_equal(ADA, ADA.T)
                out.update({f"{tag}_ADA_triu": ADA[np.triu_indices(m)], f"{tag}_L_tril": LL[np.tril_indices(m)]})
        path = os.path.join(HERE, f"{name}.npz")
        np.savez_compressed(path, **out)
        print(name, "->", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()

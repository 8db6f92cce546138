"""tests/golden/make_driver_golden.py -- the N4 fixtures: what the loop restatement (tests/driver/sedumi_loop.py) logs
when its hot path is the REFERENCE's own MEX (getada1/2/3 | getada.m, blkchol, fwblkslv, bwblkslv, invcholfac from
oracle/_ref).  Runs ONLY in the build container (needs /root/reference for the example problems).

Stored per problem (driver_<name>.npz): b and the internal c (At and K are in <name>.npz already), the iteration
count, STOP code, objective values, and the columns sedumi.m:511-512 prints, one row per iteration.  trto3 and
OH_2Pi_STO-6GN9r12g1T2 (examples/test_sedumi.m:26-27; six minutes of reference hot path each) and the complex quantum
(test_sedumi.m:28: two Hermitian PSD blocks of order 5, complex constraints) have no <name>.npz: their driver file also
carries the internal At (3902, 66180 and 341 nonzeros) and K.

    python tests/golden/make_driver_golden.py [names...]

    python tests/golden/make_driver_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from driver import sedumi_loop as sl  # noqa: E402

COLS = ("by_x0", "gap", "delta", "rate", "tP", "tD", "feas", "kcg1", "kcg2", "prec", "nskip", "nadd")


SHORT = {"OH_2Pi_STO-6GN9r12g1T2": "OH_2Pi"}


def main():
    for name in (sys.argv[1:] or ("arch0", "control07", "nb")):
        At, b, c, K = sl.load_example(name)
        S = sl.Sedumi(At, b, c, K)
        r = S.solve()
        rows = np.array([[row[k] for k in COLS] for row in r["rows"]], dtype=np.float64)
        path = os.path.join(HERE, f"driver_{SHORT.get(name, name)}.npz")
        extra = {}
        if not os.path.exists(os.path.join(HERE, f"{name}.npz")):
            A = S.A.tocsc(); A.sort_indices()
            extra = {"At_data": A.data, "At_indices": A.indices.astype(np.int32), "At_indptr": A.indptr.astype(np.int64), "At_shape": np.array(A.shape),
                     "K_l": S.K["l"], "K_q": S.K["q"], "K_s": S.K["s"], "K_rsdpN": S.K["rsdpN"]}
        np.savez_compressed(path, b=S.b, c=S.c, iter=r["iter"], STOP=r["STOP"], cx=r["cx"], by=r["by"], rows=rows, cols=np.array(COLS), **extra)
        print(name, "iter", r["iter"], "STOP", r["STOP"], "cx", r["cx"], "by", r["by"], "->", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()

"""tools/reference_sensitivity.py -- how much the REFERENCE's own L.L moves on rank-deficient fronts (helpers.rank_deficient_front_case) when
only its BLAS-1 is exchanged (naive loops -> OpenBLAS: another summation order): the yardstick for the library's deviation on the same
fronts (tools/l_error_probe.py).  CPU only (oracle/_ref)."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, scipy.sparse as sp
from helpers import rank_deficient_front_case, relerr
from oracle.refmex import RefMex, REF_DIR, find_openblas
ref = RefMex(REF_DIR)
rng = np.random.default_rng(99)
errs=[]
for case in range(120):
    args = rank_deficient_front_case(rng, 100, 340)
    ref.use_blas(None); r0 = ref.call("blkchol", 4, *args)
    ok = ref.use_blas(find_openblas()); r1 = ref.call("blkchol", 4, *args)
    same = np.array_equal(r0[2].indices, r1[2].indices) and np.array_equal(r0[3].indices, r1[3].indices)
    errs.append((relerr(sp.csc_matrix(r1[0]).data, sp.csc_matrix(r0[0]).data), relerr(r1[1], r0[1]), same))
ref.use_blas(None)
e=np.array([(a,b) for a,b,_ in errs])
print("blas switched:", ok, "cases", len(errs), "decisions equal", sum(x[2] for x in errs), "L diff max %.3e median %.3e above 1e-10: %d" % (e[:,0].max(), np.median(e[:,0]), (e[:,0]>1e-10).sum()), "d diff max %.3e" % e[:,1].max())

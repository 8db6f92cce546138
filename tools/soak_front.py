"""tools/soak_front.py -- randomised soak of k_ldl_front against the launch-per-panel path on the GPU: 40 dense fronts of 320 ... 1024
rows, full rank and rank deficient, maxu 5e5 / 30 / 2 (skip, column-probe and added-diagonal paths): L, d and the pivot lists
must agree bit for bit.  Last run (MI355X, round 2): 0 mismatches."""
import sys, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, scipy.sparse as sp
import helpers; helpers.use_hip()
from sedumi_amd import problem
rng=np.random.default_rng(123)
bad=0
for it in range(40):
    m=int(rng.integers(320,1025))
    B=rng.standard_normal((m, m if it%3 else m//2))
    X=B@B.T + (m if it%3 else 1e-9*m)*np.eye(m)
    X=sp.csc_matrix(X); X.sort_indices()
    L=problem.dense_symbolic(m)
    pars={"canceltol":1e-12,"maxu":[5e5,30.0,2.0][it%3],"abstol":1e-20}
    (l1,d1,p1,y1,k1),(l2,d2,p2,y2,k2)=helpers._factor_both_ways(X,L,pars,rng.standard_normal(m))
    ok=np.array_equal(l1,l2,equal_nan=True) and np.array_equal(d1,d2,equal_nan=True) and np.array_equal(p1[0][0],p2[0][0]) and np.array_equal(p1[1][0],p2[1][0])
    if not ok: bad+=1; print("MISMATCH m",m,"it",it)
    assert "k_ldl_front" in k1 and "k_ldl_front" not in k2
print("soak done, mismatches:", bad)

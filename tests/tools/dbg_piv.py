import sys, numpy as np, scipy.sparse as sp; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from helpers import use_hip, use_emu, relerr
(use_emu if len(sys.argv)>1 and sys.argv[1]=='emu' else use_hip)()
from oracle import glue as gl, refmex as rm
from sedumi_amd import mex
refmex=rm.RefMex(); glue=gl.Glue(refmex)
case=5
rng = np.random.default_rng(50 + case)
m = [40, 90, 150, 200, 64, 333][case]
S = sp.random(m, m, density=0.05, random_state=rng, format="csc"); S = S + S.T
sc = 10.0 ** rng.uniform(-7, 3, m)
X = sp.diags(sc) @ (S + sp.diags(np.asarray(abs(S).sum(axis=1)).ravel() * rng.choice([1.0, 1.0, 0.5], m) + 1e-3)) @ sp.diags(sc)
X = sp.csc_matrix(X); X.sort_indices()
L = glue.symbchol(X)
print("xsuper", L["xsuper"].ravel())
for maxu in (5e5, 30.0, 2.0):
    pars = dict(gl.default_pars_chol()); pars["maxu"] = maxu
    absd = np.abs(X.diagonal()) * rng.choice([1.0, 1e3, 1e8], m)
    r = refmex.call("blkchol", 4, L, X, pars, absd)
    o = mex.blkchol(L, X, pars, absd)
    do=np.asarray(o[1]).ravel(); dr=np.asarray(r[1]).ravel()
    print(maxu, "skip diff", set(o[2].indices)^set(r[2].indices), "add diff", set(o[3].indices)^set(r[3].indices), "nadd", r[3].nnz)
    for i in sorted(set(o[3].indices)^set(r[3].indices)): print("   idx", i, "ours d", do[i], "ref d", dr[i], "ref add", r[3][i,0] if i in r[3].indices else None, "our add", o[3][i,0] if i in o[3].indices else None)
    bad=np.where(np.abs(do-dr)>1e-9*np.abs(dr))[0]
    print(" n bad d", len(bad), "first", bad[:5])

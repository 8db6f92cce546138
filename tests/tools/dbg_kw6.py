import sys, numpy as np; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from helpers import use_hip, use_emu
(use_emu if len(sys.argv)>1 and sys.argv[1]=='emu' else use_hip)()
from oracle import glue as gl, refmex as rm
from sedumi_amd import problem, mex
G=gl.Glue(rm.RefMex())
P = problem.random_sdp(seed=6, **dict(m=200, lp=10, q=(5,), s=(33, 10), dens=0.05, block_local=True))
S = G.setup(P.At, P.K)
from helpers import ref_scaling
d, ud = ref_scaling(P, 6)
import inspect
from helpers import check_iteration
src = inspect.getsource(check_iteration)
pars = gl.default_pars_chol()
it = G.iteration_ref(S, d, ud, dict(pars))
LL, Ld, Lskip, Ladd = mex.blkchol(S["L"], it["ADA"], pars, it["absd"])
print("ours skip", Lskip.indices, Lskip.data)
print("ref  skip", it["Lskip"].indices, it["Lskip"].data)
print("ours add", Ladd.indices, Ladd.data); print("ref add", it["Ladd"].indices, it["Ladd"].data)
Ld=np.asarray(Ld).ravel(); rd=np.asarray(it["Ld"]).ravel()
bad=np.argsort(-np.abs(Ld-rd)/np.maximum(np.abs(rd),1e-300))[:8]
print(bad, Ld[bad], rd[bad])
print("xsuper", S["L"]["xsuper"].ravel()[:40], "nsuper", S["L"]["xsuper"].size-1)
lb = 1e-12*np.asarray(it["absd"]).ravel()[(S["L"]["perm"].ravel()-1).astype(int)]
print("lb at bad", lb[bad])

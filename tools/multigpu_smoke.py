"""tools/multigpu_smoke.py -- the N > 1 paths of sedumi_amd.dist on REAL GPUs over RCCL (what tests/test_distributed.py runs at
world_size 2 / 4 on gloo with the emulator).  One process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port 29533 tools/multigpu_smoke.py

Every case builds its problem on all ranks, runs the sharded object (RCCL collectives on device buffers of the plans) and the
single-plan reference on the rank's own GPU, and rank 0 prints one JSON line per case: {"case", "world", "max_rel_err", "ok"}.
Expected: every line "ok": true with max_rel_err < 1e-12 (blockcyclic: 0, it is compared bit for bit) (tools/multigpu_smoke.expected.txt)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import scipy.sparse as sp
    import torch
    import torch.distributed as dist
    from sedumi_amd import capi, dist as sd, mex, problem
    from sedumi_amd.plan import Plan
    rank, world, lr = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    assert capi.backend() == "hip-gfx950" and capi.device_count() > lr
    torch.cuda.set_device(lr)
    dev = torch.device("cuda", lr)
    dist.init_process_group(backend="nccl", device_id=dev)
    pars = {"canceltol": 1e-12, "maxu": 5e5, "abstol": 1e-20}

    def report(case, err):
        t = torch.tensor([err], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if rank == 0:
            print(json.dumps({"case": case, "world": world, "max_rel_err": float(t.item()), "ok": bool(t.item() < 1e-12)}), flush=True)

    # ---- ADA' by column panels (one all-gather)
    P = problem.random_sdp(m=45 * world, lp=6, q=(4, 3), s=(7, 5), dens=0.4, seed=5)
    d, ud = problem.spd_scaling(P.K, seed=2)
    ADApat = problem.symb_ada(P); L = mex.symbchol(ADApat); Q = problem.lorentz_pattern(P)
    qv = np.random.default_rng(0).standard_normal(Q.nnz)

    def make():
        pl = Plan(lr); pl.set_chol(L, ADApat); pl.set_ada(P.At, P.Ablkjc, P.K, Q)
        pl.upload("dl", d["l"]); pl.upload("ddet", d["det"]); pl.upload("udsqr", ud); pl.upload("qpr", qv)
        return pl
    ref = make(); ref.getada()
    sh = make(); sh.upload("ada", np.full(sh.nnzADA, np.nan)); sh.upload("absd", np.full(sh.m, np.nan))
    sd.ColumnShardedAda(sh, device=dev).getada()
    a0 = ref.download("ada")
    report("columns", float(max(np.abs(sh.download("ada") - a0).max() / np.abs(a0).max(), np.abs(sh.download("absd") - ref.download("absd")).max() / np.abs(ref.download("absd")).max())))
    # ---- ADA' by PSD blocks (one all-reduce), replicated factor
    P = problem.random_sdp(m=40, lp=5, q=(4, 3), s=tuple([8, 5, 6, 7, 4, 9, 5, 6][:max(3, world)]), dens=0.3, seed=9)
    d, ud = problem.spd_scaling(P.K, seed=2)
    ADApat = problem.symb_ada(P); L = mex.symbchol(ADApat); Q = problem.lorentz_pattern(P)
    qv = np.random.default_rng(0).standard_normal(Q.nnz)
    ref = make(); ref.getada(); ref.blkchol(pars, True)
    bs = sd.BlockShardedAda(P, L, ADApat, device_index=lr, device=dev)
    bs.upload_scaling(d, ud, qv); bs.getada(); bs.plan.blkchol(pars, True)
    report("blocks", float(max(np.abs(bs.plan.download("ada") - ref.download("ada")).max() / np.abs(ref.download("ada")).max(),
                               np.abs(bs.plan.download("d") - ref.download("d")).max() / np.abs(ref.download("d")).max())))
    # ---- independent subtrees (all-gather of the solution)
    P = problem.blockdiag_sdp(nblk=max(5, 2 * world), n=9, mper=7, nnz=5, seed=3)
    d, ud = problem.spd_scaling(P.K, seed=4)
    rhs = np.random.default_rng(1).standard_normal(P.m)
    solver = sd.SubtreeShardedSolver(P, device_index=lr, device=dev, pars=pars)
    solver.upload_scaling(d, ud, P); solver.factor()
    y = solver.solve(rhs)
    ADApat = problem.symb_ada(P); L = mex.symbchol(ADApat)
    pl = Plan(lr); pl.set_chol(L, ADApat); pl.set_ada(P.At, P.Ablkjc, P.K, problem.lorentz_pattern(P))
    pl.upload("dl", d["l"]); pl.upload("ddet", d["det"]); pl.upload("udsqr", ud); pl.upload("rhs", rhs)
    pl.getada(); pl.blkchol(pars, True); pl.ldlsolve()
    y1 = pl.download("y")
    report("subtrees", float(np.abs(np.asarray(y) - y1).max() / np.abs(y1).max()))
    # ---- subtrees joined by separators (reduce of fronts, reduce of update vectors, broadcast of the top's solution)
    n = 14 * max(2, world // 2)
    T = sp.diags([-1.0, -1.0], [1, -1], shape=(n, n))
    X = sp.csc_matrix(sp.kron(sp.eye(n), T) + sp.kron(T, sp.eye(n)) + 4.5 * sp.eye(n * n)); X.sort_indices()
    rhs = np.random.default_rng(2).standard_normal(n * n)
    ss = sd.SeparatorShardedSolver(X, device_index=lr, device=dev)
    ss.factor(X.data, pars)
    xs = ss.solve(rhs).cpu().numpy()
    one = Plan(lr); one.set_chol(ss.L, X); one.upload("ada", X.data); one.upload("rhs", rhs)
    one.blkchol(pars, False); one.ldlsolve()
    x1 = one.download("y")
    report("separator_grid", float(np.abs(xs - x1).max() / np.abs(x1).max()))
    # ---- ONE dense front, block-column-cyclic over the ranks (a broadcast per 64-column panel): bit for bit the single plan
    mm = 900
    rng = np.random.default_rng(mm)
    B = rng.standard_normal((mm, mm - 9))                              # nine dependent columns: the skip / add decisions travel in the records
    Xd = B @ B.T / mm
    absd = np.abs(Xd).sum(axis=1)
    Lsym, pat = problem.dense_symbolic(mm), problem.dense_pattern(mm)
    vals = Xd.ravel(order="F")
    one = Plan(lr); one.set_one_launch_fronts(False); one.set_chol(Lsym, pat); one.upload("ada", vals); one.upload("absd", absd); one.blkchol(pars, True)
    bc = sd.BlockCyclicFactor(Lsym, pat, device_index=lr, device=dev)
    bc.factor(vals, pars, absd)
    same = np.array_equal(bc.plan.download("lpr"), one.download("lpr")) and np.array_equal(bc.plan.download("d"), one.download("d"))
    report("blockcyclic", 0.0 if same else 1.0)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

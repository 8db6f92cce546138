"""N > 1 paths of sedumi_amd.dist on CPU: world_size 2, gloo backend, the fiber-emulated kernels standing in for
the GPU (one emulator library per process).  Checks that the sharded runs reproduce the single-process results:
  * ColumnShardedAda: ADA' / absd assembled from two column panels + one all-gather
  * SubtreeShardedSolver: independent etree subtrees dealt to the ranks, local factor + solve, all-gather of y
  * BlockCyclicFactor: ONE dense front block-column-cyclically over 2 / 3 ranks, a broadcast per panel: L, d, pivot lists bit for bit
"""
import os
import socket
import sys

import numpy as np
import pytest

from helpers import ROOT


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, case, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
        import torch
        import torch.distributed as dist
        from helpers import use_emu
        use_emu()
        from sedumi_amd import dist as sd, mex, problem
        from sedumi_amd.plan import Plan
        dist.init_process_group("gloo", rank=rank, world_size=world)
        pars = {"canceltol": 1e-12, "maxu": 5e5, "abstol": 1e-20}
        if case == "columns":
            P = problem.random_sdp(m=45, lp=6, q=(4, 3), s=(7, 5), dens=0.4, seed=5)
            d, ud = problem.spd_scaling(P.K, seed=2)
            ADApat = problem.symb_ada(P)
            L = mex.symbchol(ADApat)
            Q = problem.lorentz_pattern(P)
            qv = np.random.default_rng(0).standard_normal(Q.nnz)

            def make():
                pl = Plan(0); pl.set_chol(L, ADApat); pl.set_ada(P.At, P.Ablkjc, P.K, Q)
                pl.upload("dl", d["l"]); pl.upload("ddet", d["det"]); pl.upload("udsqr", ud); pl.upload("qpr", qv)
                return pl
            ref = make(); ref.getada()
            sh = make(); sh.upload("ada", np.full(sh.nnzADA, np.nan)); sh.upload("absd", np.full(sh.m, np.nan))
            cs = sd.ColumnShardedAda(sh)
            cs.getada()
            err = max(np.abs(sh.download("ada") - ref.download("ada")).max(), np.abs(sh.download("absd") - ref.download("absd")).max())
            q.put((rank, float(err), [int(c) for c in cs.cols]))
        elif case.startswith("separator"):
            # ONE connected elimination tree: subtrees to the ranks, the top of the tree to rank 0; one reduce of the roots' fronts,
            # one reduce of their update vectors, one broadcast of the top's solution
            import scipy.sparse as sp
            from helpers import spd_pattern
            kind = case.split("_", 1)[1]
            rng = np.random.default_rng(23)
            if kind == "bordered":                           # three dense diagonal blocks under a dense border: three subtrees, one separator front
                sizes, nc = [40, 35, 30], 10
                mm = sum(sizes) + nc
                X = np.zeros((mm, mm)); o = 0
                for n in sizes:
                    X[o:o + n, o:o + n] = rng.standard_normal((n, n)); o += n
                X[o:, :] = rng.standard_normal((nc, mm))
                X = 0.1 * (X + X.T) / np.sqrt(mm)
                X = X + np.diag(np.abs(X).sum(axis=1) + 1.0)
            else:
                X = spd_pattern(kind, {"arrow": 90, "grid": 196, "rand": 150}[kind], rng, 0.03)
            X = sp.csc_matrix(X); X.sort_indices()
            m = X.shape[0]
            solver = sd.SeparatorShardedSolver(X)
            assert solver.top.any() and len(solver.roots) >= 2, (solver.top.sum(), solver.roots)       # (rank 0 owns the top; the subtrees go where the load is lowest)
            rhs = rng.standard_normal(m)
            xs = []
            for rep in range(2):                             # twice: the arenas are reused
                solver.factor(X.data, pars)
                xs.append(solver.solve(rhs).cpu().numpy().copy())
            assert np.array_equal(xs[0], xs[1]), (float(np.abs(xs[0] - xs[1]).max()), float(np.abs(xs[0]).max()), int(np.isnan(xs[0]).sum()), int(np.isnan(xs[1]).sum()))
            one = Plan(0); one.set_chol(solver.L, X); one.upload("ada", X.data); one.upload("rhs", rhs)
            one.blkchol(pars, False); one.ldlsolve()
            x1 = one.download("y")
            want = np.linalg.solve(X.toarray(), rhs)
            err = max(np.abs(xs[0] - x1).max() / np.abs(x1).max(), np.abs(xs[0] - want).max() / np.abs(want).max())
            q.put((rank, float(err), [int(solver.top.sum()), len(solver.roots)]))
        elif case.startswith("blockcyclic"):
            # ONE dense front, block-column-cyclic over the ranks (SURVEY.md 8e row blkchol): every panel factored by the owner of its tile
            # column and broadcast, every tile updated by the owner of its column -- L, d and the pivot lists BIT FOR BIT those of one plan
            _, mm, blk, deficient = case.split("_")
            mm, blk = int(mm), int(blk)
            rng = np.random.default_rng(mm)
            B = rng.standard_normal((mm, mm - 9 if deficient == "rank" else mm + 5))
            X = B @ B.T / mm + (0.0 if deficient == "rank" else 0.05) * np.eye(mm)       # ("rank": nine dependent columns -- the skip / add decisions)
            absd = np.abs(X).sum(axis=1)
            Lsym, pat = problem.dense_symbolic(mm), problem.dense_pattern(mm)
            vals = X.ravel(order="F")
            one = Plan(0); one.set_one_launch_fronts(False); one.set_chol(Lsym, pat); one.upload("ada", vals); one.upload("absd", absd)
            one.blkchol(pars, True)
            bc = sd.BlockCyclicFactor(Lsym, pat, blk=blk)
            outs = []
            for rep in range(2):                            # twice: the arenas and counters are reused
                bc.factor(vals, pars, absd)
                outs.append((bc.plan.download("lpr"), bc.plan.download("d"), bc.plan.pivots()))
            l1, d1, (s1, a1) = one.download("lpr"), one.download("d"), one.pivots()
            for lpr, dd, (sk, ad) in outs:
                assert np.array_equal(lpr, l1) and np.array_equal(dd, d1), (float(np.abs(lpr - l1).max()), float(np.abs(dd - d1).max()))
                assert all(np.array_equal(x, y) for x, y in zip(sk + ad, s1 + a1))
            rhs = rng.standard_normal(mm)
            bc.plan.upload("rhs", rhs); bc.plan.ldlsolve(); one.upload("rhs", rhs); one.ldlsolve()
            assert np.array_equal(bc.plan.download("y"), one.download("y"))
            q.put((rank, 0.0, [int(s1[0].size), int(a1[0].size), bc.npanel]))
        elif case in ("blocks", "blocks_confined"):
            # ADA' = sum of the PSD blocks' contributions: blocks dealt to the ranks, one all-reduce of [values | absd]
            from helpers import ref_scaling
            P = problem.random_sdp(m=40, lp=5, q=(4, 3), s=(8, 5, 6), hs=(4,), dens=0.3, seed=9)
            if case == "blocks_confined":
                # columns whose PSD nonzeros are confined to ONE block each (every block in turn, so some of them sit on a rank
                # other than 0 whatever the deal) while they also have LP entries: absd's LP / Lorentz term must still arrive
                import scipy.sparse as sp
                start, ns = problem._psd_rows(P.K)
                nreal = P.K["s"].size - 1
                A = sp.lil_matrix(P.At)
                for j in range(16):
                    keep = j % len(ns)
                    for k, n in enumerate(ns):
                        if k != keep:
                            A[start[k]:start[k] + (n * n if k < nreal else 2 * n * n), j] = 0.0
                    A[start[keep], j] = 1.5
                    A[1, j] = 1.0 + 0.1 * j
                A = sp.csc_matrix(A); A.eliminate_zeros()
                P = problem.Problem(A, P.K, "confined")
            d, ud = ref_scaling(P, 2)
            ADApat = problem.symb_ada(P)
            L = mex.symbchol(ADApat)
            Q = problem.lorentz_pattern(P)
            qv = np.random.default_rng(0).standard_normal(Q.nnz)
            ref = Plan(0); ref.set_chol(L, ADApat); ref.set_ada(P.At, P.Ablkjc, P.K, Q)
            ref.upload("dl", d["l"]); ref.upload("ddet", d["det"]); ref.upload("udsqr", ud); ref.upload("qpr", qv)
            ref.getada()
            bs = sd.BlockShardedAda(P, L, ADApat)
            bs.upload_scaling(d, ud, qv)
            bs.getada()
            err = max(np.abs(bs.plan.download("ada") - ref.download("ada")).max() / np.abs(ref.download("ada")).max(),
                      np.abs(bs.plan.download("absd") - ref.download("absd")).max() / np.abs(ref.download("absd")).max())
            bs.plan.blkchol(pars, True)                  # the replicated factor sees the assembled ADA'
            ref.blkchol(pars, True)
            err = max(err, np.abs(bs.plan.download("d") - ref.download("d")).max() / np.abs(ref.download("d")).max())
            q.put((rank, float(err), [int(b.size) for b in bs.blocks_of]))
        else:
            if case == "subtrees_lorentz":
                # components with Lorentz cones: the getada2 term needs every rank's own DAt.q (formed on the device
                # from its d.q1 / d.q2 in upload_scaling)
                from helpers import ref_scaling
                P = problem.random_sdp(m=30, lp=0, q=(8, 7, 9), s=(6, 5), dens=0.5, seed=7, block_local=True)
                d, ud = ref_scaling(P, 4)
            else:
                P = problem.blockdiag_sdp(nblk=5, n=9, mper=7, nnz=5, seed=3)
                d, ud = problem.spd_scaling(P.K, seed=4)
            rhs = np.random.default_rng(1).standard_normal(P.m)
            solver = sd.SubtreeShardedSolver(P, pars=pars)
            solver.upload_scaling(d, ud, P)
            solver.factor()
            y = solver.solve(rhs)
            solver.upload_rhs(rhs)                      # the resident form: gathered segments stay in solver.recv
            recv = solver.solve_resident().cpu().numpy()
            seg = solver.send.numel()
            for qq, c in enumerate(solver.cols_of):
                assert np.array_equal(recv[qq * seg:qq * seg + c.size], y[c])
            # single-process answer on the whole problem
            ADApat = problem.symb_ada(P); L = mex.symbchol(ADApat)
            pl = Plan(0); pl.set_chol(L, ADApat); pl.set_ada(P.At, P.Ablkjc, P.K, problem.lorentz_pattern(P))
            pl.upload("dl", d["l"]); pl.upload("ddet", d["det"]); pl.upload("udsqr", ud); pl.upload("rhs", rhs)
            if case == "subtrees_lorentz":
                pl.upload("q1", d["q1"]); pl.upload("q2", d["q2"]); pl.getdatq()
            pl.getada(); pl.blkchol(pars, True); pl.ldlsolve()
            y1 = pl.download("y")
            q.put((rank, float(np.abs(y - y1).max() / np.abs(y1).max()), [int(c.size) for c in solver.cols_of]))
        dist.destroy_process_group()
    except Exception as e:  # surface the failure in the parent
        import traceback
        q.put((rank, "ERR " + traceback.format_exc(), None))


@pytest.mark.parametrize("case,world", [("columns", 2), ("subtrees", 2), ("subtrees_lorentz", 2), ("blocks", 2), ("blocks_confined", 2),
                                        ("separator_arrow", 2), ("separator_grid", 2), ("separator_bordered", 2), ("separator_grid", 4), ("separator_rand", 4),
                                        ("blockcyclic_700_1_full", 2), ("blockcyclic_450_2_rank", 2), ("blockcyclic_530_1_rank", 3)])
def test_ranks_gloo(case, world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, err, info in res:
        assert not isinstance(err, str), err
        assert err < 1e-12, (rank, err)
        if case.startswith("separator"):
            assert info is not None and info[0] >= 1 and info[1] >= 2
            continue
        if case.startswith("blockcyclic"):
            assert info is not None and info[2] >= 8 and (info[0] + info[1] > 0) == case.endswith("rank"), info
            continue
        if case in ("blocks", "blocks_confined"):
            assert info is not None and sum(info) == 4 and min(info) >= 1
            continue
        assert info is not None and (len(info) == 3 if case == "columns" else sum(info) == (35 if case == "subtrees" else 30) and min(info) > 0)


def test_split_problem_components():
    sys.path.insert(0, ROOT)
    from sedumi_amd import dist as sd, problem
    P = problem.blockdiag_sdp(nblk=6, n=8, mper=5, nnz=4, seed=1)
    ncomp, lab, _ = sd.components(P)
    assert ncomp == 6 and np.array_equal(np.bincount(lab), np.full(6, 5))
    parts = sd.split_problem(P, 4)
    allc = np.sort(np.concatenate([c for (_, c, _) in parts]))
    assert np.array_equal(allc, np.arange(P.m))
    for sub, cols, rows in parts:
        assert sub.K["s"].size == cols.size // 5 and sub.At.shape == (1 + 64 * (cols.size // 5), cols.size)

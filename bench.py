#!/usr/bin/env python
"""bench.py -- IPM iterations per second of the normal-equations hot path on MI355X.

One "step" = one IPM iteration unit (sedumi.m:450-473) on frozen, HBM-resident inputs:
    1 x (getada1 + getada2 + getada3)  ->  1 x blkchol  ->  4 x (fwblkslv, ./L.d, bwblkslv)   (single RHS each)
Workload at N=1: the control07-shaped SDP of BASELINE.json configs[1] (m=666, K.s=[70 35], dense ADA',
synthetic data of that shape -- the reference's examples do not travel to the GPU box).

N>1 (launched with torch.distributed.run, one rank per GPU): the single dense supernode of this workload
does not shard, so every rank runs an independent replica of the unit ("replicas
only") -- weak scaling, no data-path collective; barrier + max-over-ranks timing.

Prints ONE JSON line (rank 0) with the fields the driver expects plus `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PARS = {"canceltol": 1e-12, "maxu": 5e5, "abstol": 1e-20}          # checkpars.m:144-168
HBM_PEAK_GBS = 8000.0                                               # MI355X_MICROARCH.md: HBM3E 8 TB/s
NSOLVE = 4
CONFIG_NOTE = {"contro": "control07-shaped SDP (BASELINE.json configs[1])", "nb": "nb-shaped SOCP (BASELINE.json configs[2])",
               "maxcut": "MAXCUT SDP, one dense PSD block (BASELINE.json configs[3])"}


def build_workload(name, seed):
    from sedumi_amd import problem
    if name == "control07":
        P = problem.control_like(seed=seed)
    elif name.startswith("maxcut"):
        P = problem.maxcut(int(name[6:] or 4000))
    elif name == "nb":                                  # BASELINE.json configs[2] shape: 793 Lorentz cones of dimension 3, m = 123
        P = problem.random_sdp(m=123, lp=4, q=(3,) * 793, s=(), dens=0.66, seed=31 + seed)
        P.name = "nb_like(m=123,q=793x3)"
    else:
        raise SystemExit("unknown workload " + name)
    L, ADA, Q = problem.dense_symbolic(P.m), problem.dense_pattern(P.m), problem.lorentz_pattern(P)
    d, ud = problem.spd_scaling(P.K, seed=seed + 5)
    rhs = np.random.default_rng(seed).standard_normal(P.m)
    return P, L, ADA, Q, d, ud, rhs


def cpu_baseline(P, d, ud, rhs, budget_s=12.0):
    """The compiled reference MEX (oracle/_ref) timed on this host, one thread, on a bounded sample of the
    same workload: repeated iteration units until ~budget_s of CPU time is spent."""
    try:
        from oracle import glue as gl, refmex
        if not refmex.available():
            return None
        G = gl.Glue()
        ref = G.ref
        S = G.setup(P.At, P.K)
        K = P.K
        dd = {"l": d["l"], "det": d["det"], "q1": np.ones(K["q"].size),
              "q2": np.zeros(int(K["mainblks"].ravel()[2] - K["mainblks"].ravel()[1]))}
        DAt = G.getDAtm(S, dd)
        dstruct = {"l": dd["l"].reshape(-1, 1), "det": dd["det"].reshape(-1, 1)}
        units, tot = 0, 0.0
        t_wall = time.perf_counter()
        while True:
            t1, ADA1 = ref.timed_call("getada1", 1, (S["ADA"], S["A"], S["Ablkjc"][:, 2], S["Aord"]["lqperm"], dstruct, K["qblkstart"]))
            t2, ADA2 = ref.timed_call("getada2", 1, (ADA1, DAt, S["Aord"], K))
            t3, (ADA3, absd) = ref.timed_call("getada3", 2, (ADA2, S["A"], S["Ablkjc"][:, 2], S["Aord"], ud.reshape(-1, 1), K))
            t4, (LL, Ld, _, _) = ref.timed_call("blkchol", 4, (S["L"], ADA3, dict(PARS), absd))
            L = dict(S["L"]); L["L"] = LL
            ts = 0.0
            for _ in range(NSOLVE):
                tf, p = ref.timed_call("fwblkslv", 1, (L, rhs.reshape(-1, 1)))
                tb, _y = ref.timed_call("bwblkslv", 1, (L, p / Ld))
                ts += tf[0] + tb[0]
            tot += t1[0] + t2[0] + t3[0] + t4[0] + ts
            units += 1
            if tot >= budget_s or time.perf_counter() - t_wall > 3 * budget_s or units >= 400:
                break
        return {"value": units / tot, "unit": "IPM iters/s", "cores": 1, "kind": "reference",
                "sample": f"{units} iteration units of {P.name} through oracle/_ref (unmodified reference C, gcc -O2, "
                          f"naive BLAS-1), MEX calls only ({tot:.1f} s CPU)"}
    except Exception as e:  # the baseline must never break the bench line
        return {"value": None, "unit": "IPM iters/s", "cores": 1, "kind": "reference", "sample": f"failed: {e}"}


def bench_subtrees(args, rank, local_rank, world, torch, dist, coll_dev):
    """BASELINE.json configs[4]: block-diagonal SDP (default 64 PSD blocks of order 200, 150 constraints each).  The
    64 independent elimination-tree subtrees are dealt to the ranks (sedumi_amd.dist.SubtreeShardedSolver): ADA',
    factor and solves of a subtree never leave its rank; the only exchange is the all-gather of the solution
    segments after each solve.  Total work is fixed: strong scaling."""
    from sedumi_amd import dist as sd, problem
    parts = args.workload.split(":")
    nblk, n, mper = (int(parts[1]), int(parts[2]), int(parts[3])) if len(parts) == 4 else (64, 200, 150)
    P = problem.blockdiag_sdp(nblk=nblk, n=n, mper=mper, nnz=20, seed=4)
    d, ud = problem.spd_scaling(P.K, seed=5)
    rhs = np.random.default_rng(0).standard_normal(P.m)
    dev = coll_dev if dist is not None else torch.device("cpu")
    solver = sd.SubtreeShardedSolver(P, device_index=local_rank, device=dev, pars=PARS)
    solver.upload_scaling(d, ud, P)
    solver.upload_rhs(rhs)                             # inputs resident in HBM before the timed region

    def step():
        solver.factor()
        for _ in range(NSOLVE):
            solver.solve_resident()

    def sync():
        if solver.plan is not None:
            solver.plan.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if solver.plan is not None:
        solver.plan.sync()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        print(json.dumps({
            "metric": "IPM iters/sec (ADA' form+factor+solve)", "value": args.steps / elapsed, "unit": "IPM iters/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{P.name}: block-diagonal SDP (BASELINE.json configs[4]), m={P.m}, {nblk} independent subtrees; "
                                   f"unit = getada1+2+3, blkchol, {NSOLVE}x(fwblkslv,./d,bwblkslv)",
                       "parallelism": f"{world} rank(s): subtrees per rank {[int(c.size // mper) for c in solver.cols_of]}, all-gather of y only"},
            "roofline": None, "cpu_baseline": None}), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="control07",
                    help="control07 (default, BASELINE configs[1]) | nb (configs[2] shape) | maxcut<n> (configs[3]) | blockdiag[:nblk:n:mper] (configs[4])")
    ap.add_argument("--shard", default="auto", choices=["auto", "replicas", "columns"],
                    help="N>1: replicas = independent units per rank (default for single-supernode workloads); "
                         "columns = ONE unit per step, ADA' column panels per rank + RCCL all-gather, factor/solves replicated")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured hipGraph (one launch per step)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    # BENCH_SHARE_GPU=1 (smoke tests on a 1-GPU box only): all ranks use device 0 and the process group runs on gloo
    share = os.environ.get("BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    dist = None
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        if share:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    coll_dev = torch.device("cpu") if share else torch.device("cuda", local_rank)

    from sedumi_amd.plan import Plan
    if args.workload.startswith("blockdiag"):
        return bench_subtrees(args, rank, local_rank, world, torch, dist, coll_dev)
    shard_cols = args.shard == "columns" and world > 1
    P, L, ADA, Q, d, ud, rhs = build_workload(args.workload, seed=0 if shard_cols else rank)
    plan = Plan(local_rank)
    plan.set_chol(L, ADA)
    plan.set_ada(P.At, P.Ablkjc, P.K, Q)
    plan.upload("dl", d["l"]); plan.upload("ddet", d["det"]); plan.upload("udsqr", ud); plan.upload("rhs", rhs)
    if Q.nnz:                                  # Lorentz cones: DAt.q values in the order of the pattern (getDAtm.m's product)
        plan.upload("qpr", 0.1 * np.random.default_rng(3).standard_normal(Q.nnz))

    cs = None
    if shard_cols:
        from sedumi_amd import dist as sd
        cs = sd.ColumnShardedAda(plan, device=coll_dev)

    def step():
        if cs is not None:
            cs.getada()
        else:
            plan.getada()
        plan.blkchol(PARS, True)
        for _ in range(NSOLVE):
            plan.ldlsolve()

    if args.graph and cs is None:
        eager_step = step
        for _ in range(2):
            eager_step()                       # first-use work (function attributes, allocations) outside the capture
        plan.sync()
        gid = plan.graph_capture(eager_step)

        def step():                            # noqa: F811
            plan.graph_launch(gid)

    def barrier():
        plan.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    plan.sync()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- roofline leg: the dominant kernel's launches timed with HIP events on the plan's stream
    # (same steps, events around every launch; the un-instrumented loop above gives `value`).
    plan.kprof(True)
    prof_step = eager_step if (args.graph and cs is None) else step
    for _ in range(min(args.steps, 50)):
        prof_step()
    prof = plan.kprof_summary()
    plan.kprof(False)
    nprof = min(args.steps, 50)
    m, nnzL = plan.m, plan.nnzL
    # ALGORITHMIC work per launch of each hot kernel (SURVEY.md 8(d) per-unit figures divided
    # by the launches per unit).  bound "hbm": bytes, peak 8 TB/s; bound "mfma": FP64 flops, peak 78.6 TFLOP/s
    # (on CDNA4 the FP64 matrix rate equals the FP64 vector rate).
    solve_bytes = 8.0 * nnzL + 8.0 * m + 16.0 * m       # per triangular solve: 8*nnz(L) + 8*len(lindx) + 16*m
    npan = (m + 63) // 64
    model = {
        "k_ldl_single": ("hbm", 2.0 * solve_bytes),                                  # forward + backward sweep in one launch
        "k_fw_level": ("hbm", solve_bytes), "k_bw_level": ("hbm", solve_bytes),
        "k_ldl_update": ("mfma", (m ** 3 / 3.0) / max(1, npan - 1)),                 # trailing updates carry the m^3/3 of the LDL'
        # one launch per panel: 64x64 LDL' (latency bound), row solves and the previous panel's trailing update
        "k_ldl_panel": ("mfma", 2.0 * 64 ** 3 / 3.0 + (0.0 if "k_ldl_update" in prof else (m ** 3 / 3.0) / max(1, npan - 1))),
        "k_psd_stage1_mfma": ("hbm", 8.0 * (ud.size + P.At.nnz + plan.nnzADA)),      # SURVEY.md 8(d) getada3 lower bound
        "k_psd_stage1": ("hbm", 8.0 * (ud.size + P.At.nnz + plan.nnzADA)),
        "k_psd_stage2": ("hbm", 8.0 * (P.At.nnz + plan.nnzADA)),
        "k_ada_spdot": ("hbm", 8.0 * (P.At.nnz + plan.nnzADA)),
    }
    peaks = {"hbm": (HBM_PEAK_GBS, "GB/s", 1e9), "mfma": (78.6, "TFLOP/s", 1e12)}
    dom = max(prof.items(), key=lambda kv: kv[1][1])[0] if prof else None
    roof = None
    if dom:
        calls, ms = prof[dom]
        avg_s = ms / calls * 1e-3
        key = dom.split("<")[0]
        if key.startswith("k_psd_stage2"):
            key = "k_psd_stage2"
        bound, work = model.get(key, ("hbm", 8.0 * (plan.nnzADA / 2 + 2 * nnzL)))
        peak, unit, scale = peaks[bound]
        ach = work / avg_s / scale
        roof = {"kernel": dom, "bound": bound, "achieved": ach, "peak": peak, "unit": unit,
                "frac": ach / peak, "traffic": None, "avg_launch_us": avg_s * 1e6, "launches_per_step": calls / nprof,
                "algorithmic_work_per_launch": work,
                "stage_ms_per_step": {k: v[1] / nprof for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])}}
        # the solve kernel is the path's HBM-bound kernel (north_star): always reported beside the dominant one
        for sk in ("k_ldl_single", "k_fw_level"):
            if sk in prof:
                c2, ms2 = prof[sk]
                b2 = model[sk][1] * (1.0 if sk == "k_ldl_single" else 1.0)
                roof["solve_kernel"] = {"kernel": sk, "avg_launch_us": ms2 / c2 * 1e3, "achieved_GBs": b2 / (ms2 / c2 * 1e-3) / 1e9,
                                        "frac_of_hbm_peak": b2 / (ms2 / c2 * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": b2}
                break
        import glob
        pmcs = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")))
        if pmcs:
            try:
                roof["traffic"] = json.load(open(pmcs[-1])).get(dom)
                roof["traffic_source"] = os.path.basename(pmcs[-1])
            except Exception:
                pass

    if rank == 0:
        base = None
        if world == 1 and not args.no_cpu_baseline:
            base = cpu_baseline(P, d, ud, rhs)
        out = {
            "metric": "IPM iters/sec (ADA' form+factor+solve)", "value": (1 if shard_cols else world) * args.steps / elapsed, "unit": "IPM iters/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "strong" if shard_cols else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{P.name}: {CONFIG_NOTE.get(args.workload[:6], CONFIG_NOTE['contro'])}, m={P.m}, nnz(At)={P.At.nnz}, "
                                   f"dense ADA' {P.m}x{P.m}, nnz(L)={nnzL}; unit = getada1+2+3, blkchol, {NSOLVE}x(fwblkslv,./d,bwblkslv)",
                       "parallelism": ("ADA' column panels + RCCL all-gather, factor/solves replicated" if shard_cols else "replicas") if world > 1 else "single GPU"},
            "roofline": roof, "cpu_baseline": base,
        }
        if base and base.get("value"):
            out["speedup_vs_cpu_reference"] = out["value"] / (1 if shard_cols else world) / base["value"]
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

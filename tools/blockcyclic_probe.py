"""tools/blockcyclic_probe.py <workload> [world] [blk] -- what the block-column-cyclic LDL' of ONE dense front (sedumi_amd.dist.BlockCyclicFactor,
SURVEY.md 8e row blkchol) would take on `world` GPUs, measured on ONE: the `world` plans of the ranks live in this process on the same device
and take turns, every rank's panel launch timed by itself with HIP events (the plan's timers), the panel handed on by device-to-device copies.

    projected factor time  =  sum over panels of [ max over ranks of the rank's launch  +  the panel's copy out and in (a stand-in for the broadcast:
                              the same bytes at HBM speed and no link latency -- a LOWER bound for RCCL over xGMI) ]

next to the single plan's factor on the same path (launch per panel) and on its default path.  The result is also checked: L and d of every
rank bit for bit those of the single plan.  One JSON line."""
import json
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "maxcut4000"
world = int(sys.argv[2]) if len(sys.argv) > 2 else 2
blk = int(sys.argv[3]) if len(sys.argv) > 3 else 1
P, L, ADA, Q, d, ud, rhs, qpr, note = bench.build_workload(name, 0)
dev = torch.device("cuda", 0)


def timed(plan, fn, reps=1):
    plan.sync(); plan.timer_begin(3)
    for _ in range(reps):
        fn()
    plan.timer_end(3)
    return plan.timer_ms(3) * 1e3 / reps


default = bench.make_plan(0, P, L, ADA, Q, d, ud, rhs, qpr)
default.getada()
for _ in range(3):
    default.blkchol(bench.PARS, True)
t_default = timed(default, lambda: default.blkchol(bench.PARS, True), 5)
vals, absd = default.download("ada"), default.download("absd")
one = bench.make_plan(0, P, L, ADA, Q, d, ud, rhs, qpr, one_launch_fronts=False)
one.upload("ada", vals); one.upload("absd", absd)
for _ in range(3):
    one.blkchol(bench.PARS, True)
t_one = timed(one, lambda: one.blkchol(bench.PARS, True), 5)
l1, d1 = one.download("lpr"), one.download("d")
npanel = (P.m + 63) // 64
nrec = 4 * 64 + 2 + 64 * 64
plans = []
for r in range(world):
    p = bench.make_plan(0, P, L, ADA, Q, d, ud, rhs, qpr, one_launch_fronts=False)
    p.set_column_owner(world, r, blk); p.upload("ada", vals); p.upload("absd", absd)
    plans.append(p)
buf = torch.zeros(plans[0].panel_slice(0)[1] + nrec, dtype=torch.float64, device=dev)
best = None
for rep in range(3):
    launch = np.zeros((npanel, world)); xfer = np.zeros(npanel)
    t_begin = max(timed(p, lambda p=p: p.blkchol_begin(bench.PARS, True)) for p in plans)
    for q in range(npanel):
        for r, p in enumerate(plans):
            launch[q, r] = timed(p, lambda p=p: p.blkchol_panels(0, 1, q, q + 1))
        src = (q // blk) % world
        off, n = plans[src].panel_slice(q)
        ps = plans[src]
        ps.sync(); ps.timer_begin(3)
        ps.panel_record(q, False); ps.copy("fronts", buf, off, n, False); ps.copy("panelrec", buf[n:], 0, nrec, False)
        ps.timer_end(3); xfer[q] = ps.timer_ms(3) * 1e3
        t_in = 0.0
        for r, p in enumerate(plans):
            if r != src:
                p.sync(); p.timer_begin(3)
                p.copy("fronts", buf, off, n, True); p.copy("panelrec", buf[n:], 0, nrec, True); p.panel_record(q, True)
                p.timer_end(3); t_in = max(t_in, p.timer_ms(3) * 1e3)
        xfer[q] += t_in
    t_end = max(timed(p, lambda p=p: p.blkchol_end()) for p in plans)
    tot = t_begin + float(launch.max(axis=1).sum()) + float(xfer.sum()) + t_end
    if best is None or tot < best["projected_us"]:
        best = {"projected_us": tot, "begin_us": t_begin, "launches_us_max_over_ranks": float(launch.max(axis=1).sum()),
                "launches_us_per_rank": [float(x) for x in launch.sum(axis=0)], "panel_copies_us": float(xfer.sum()), "inverse_prep_us": t_end}
same = all(np.array_equal(p.download("lpr"), l1) and np.array_equal(p.download("d"), d1) for p in plans)
print(json.dumps({"workload": name, "m": int(P.m), "world": world, "blk": blk, "panels": npanel, "bits_equal_to_single_plan": bool(same),
                  "single_plan_default_path_us": t_default, "single_plan_launch_per_panel_us": t_one, **best,
                  "bytes_broadcast_per_factor": 8.0 * sum(plans[0].panel_slice(q)[1] + nrec for q in range(npanel)),
                  "note": "ranks take turns on ONE device; every launch timed with events and a stream sync around it (the single plan's figures are back-to-back "
                          "launches); the panel copies are device-to-device: no link latency, HBM speed"}))

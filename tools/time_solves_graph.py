"""tools/time_solves_graph.py <workload>... -- microseconds per fw + ./d + bw solve, launched eagerly (one hipLaunchKernel per launch, the
host running ahead of the device) against replayed from a captured hipGraph of 4 solves (sdm_plan_graph_*): what the launch path itself costs
per dependent launch.  One JSON line per workload."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
if os.environ.get("SDM_LIB"):                      # a measurement variant (python -m sedumi_amd.build --variant <tag> <flags>)
    from sedumi_amd import capi
    capi.use_library(os.path.join(ROOT, "sedumi_amd", "lib", os.environ["SDM_LIB"]))
import bench  # noqa: E402

for name in sys.argv[1:] or ["control07"]:
    P, L, ADA, Q, d, ud, rhs, qpr, note = bench.build_workload(name, 0)
    plan = bench.make_plan(0, P, L, ADA, Q, d, ud, rhs, qpr)
    plan.getada(); plan.blkchol(bench.PARS, True)
    for _ in range(8):
        plan.ldlsolve()
    plan.sync()

    def four():
        for _ in range(4):
            plan.ldlsolve()
    reps = 50
    plan.timer_begin(2)
    for _ in range(reps):
        four()
    plan.timer_end(2)
    eager = plan.timer_ms(2) * 1e3 / (4 * reps)
    t0 = time.perf_counter()
    for _ in range(reps):
        four()
    t_host = (time.perf_counter() - t0) * 1e6 / (4 * reps)      # host time to ENQUEUE one solve
    plan.sync()
    gid = plan.graph_capture(four)
    for _ in range(3):
        plan.graph_launch(gid)
    plan.sync()
    plan.timer_begin(2)
    for _ in range(reps):
        plan.graph_launch(gid)
    plan.timer_end(2)
    graph = plan.timer_ms(2) * 1e3 / (4 * reps)
    plan.kprof(True)
    four()
    nl = sum(v[0] for v in plan.kprof_summary().values()) / 4.0
    plan.kprof(False)
    print(json.dumps({"workload": name, "m": int(P.m), "launches_per_solve": nl, "us_per_solve_eager": eager, "us_per_solve_graph": graph,
                      "host_us_to_enqueue_a_solve": t_host, "us_per_launch_eager": eager / nl, "us_per_launch_graph": graph / nl}), flush=True)
    plan.close()

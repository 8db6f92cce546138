#!/bin/bash
# bench line with every reference example at both scalings; the separator-sharded solver on one GPU and as two ranks sharing it (gloo)
OUT=gpurun_out/r03m; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python bench.py --steps 100 --warmup 10 > $OUT/bench.json 2> $OUT/bench.err
python - <<'PY'
import json
b=json.load(open("gpurun_out/r03m/bench.json"))
print(b["value"], b["ms_per_step"], b["roofline"]["kernel"], b["roofline"]["avg_launch_us"], b["phases_ms_per_step"]["ada_ms"], b["phases_ms_per_step"]["factor_ms"], b["phases_ms_per_step"]["solves_ms"], b["cpu_baseline"]["value"])
for o in b.get("other_configs", []):
    print(o.get("workload"), o.get("ms_per_step"), o.get("phases_ms_per_step"), (o.get("solve") or {}).get("frac_of_hbm_peak"), o.get("error"))
PY
timeout 300 python bench.py --workload grid:120 --steps 10 --warmup 2 > $OUT/grid1.json 2> $OUT/grid1.err; cut -c1-700 $OUT/grid1.json; tail -2 $OUT/grid1.err
BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --workload grid:60 --steps 3 --warmup 1 > $OUT/grid2.json 2> $OUT/grid2.err; cut -c1-700 $OUT/grid2.json; tail -3 $OUT/grid2.err

#!/bin/bash
# round 4, run p: timeline of the bench unit (kernel trace) -- ADA' cleared inside stage 1, assemble + pivot bounds as one launch
mkdir -p gpurun_out/r04p
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04p/kt -o kt -- python $GRAFT_REPO_ROOT/tools/trace_units.py > /dev/null 2>$GRAFT_REPO_ROOT/gpurun_out/r04p/err.txt
cd $GRAFT_REPO_ROOT
python tools/unit_timeline.py gpurun_out/r04p/kt/kt_kernel_trace.csv | tee gpurun_out/r04p/timeline.txt
python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-other-configs > gpurun_out/r04p/bench.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r04p/bench.json')); print(d['value'], d['ms_per_step'], d['phases_ms_per_step']['ada_ms'], d['phases_ms_per_step']['factor_ms'], d['phases_ms_per_step']['solves_ms'])"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "golden or iteration or one_launch_front or resident or pivot or maxcut or blockdiag" 2>&1 | tail -3

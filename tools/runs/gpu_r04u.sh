#!/bin/bash
# round 4, run u: MAXCUT-4000 through the MEX shims, counters per unit, with and without the host's memory policy
mkdir -p gpurun_out/r04u
python tools/mex_counters.py maxcut4000 4 > gpurun_out/r04u/keep_freed.txt 2>&1; cat gpurun_out/r04u/keep_freed.txt | tail -9
MEXHOST_DEFAULT_MALLOC=1 python tools/mex_counters.py maxcut4000 4 > gpurun_out/r04u/default_malloc.txt 2>&1; cat gpurun_out/r04u/default_malloc.txt | tail -9

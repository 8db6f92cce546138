#!/bin/bash
# round 4, run q: where the gap in front of k_ldl_front comes from -- the unit without the inverse behind the factor (no fork / join)
mkdir -p gpurun_out/r04q
cd /tmp && export TMPDIR=/tmp
SDM_LIB=libsedumi_hip_nofollow.so rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04q/kt -o kt -- python $GRAFT_REPO_ROOT/tools/trace_units.py > /dev/null 2>$GRAFT_REPO_ROOT/gpurun_out/r04q/err.txt
cd $GRAFT_REPO_ROOT
python tools/unit_timeline.py gpurun_out/r04q/kt/kt_kernel_trace.csv | tee gpurun_out/r04q/timeline_nofollow.txt

#!/bin/bash
# round 4, run w: unit timelines of the other configs (where are the gaps?)
mkdir -p gpurun_out/r04w
export TMPDIR=/tmp
for spec in "arch0 k_dsqr" "nb k_lq_q_prep" "blockdiag k_dsqr" "maxcut4000 k_psd_direct"; do
  set -- $spec
  (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04w/kt_$1 -o kt -- python $GRAFT_REPO_ROOT/tools/trace_units.py $1 > /dev/null 2>$GRAFT_REPO_ROOT/gpurun_out/r04w/err_$1.txt)
  echo "== $1"; python tools/unit_timeline.py gpurun_out/r04w/kt_$1/kt_kernel_trace.csv $2 2>&1 | awk '{ if (NR<=40) print }' | tee gpurun_out/r04w/timeline_$1.txt | cut -c1-130
done

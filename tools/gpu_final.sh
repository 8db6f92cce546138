#!/bin/bash
# final runs of a round on the GPU box: the whole GPU suite, smoke, the bench line, rocprofv3 kernel stats + PMC passes (HBM traffic, matrix-core
# counters) of four workloads.  gpurun --timeout 2400 -- 'bash tools/gpu_final.sh <tag>'
TAG=${1:-final}
bash tools/gpu_run.sh $TAG tests "tests" run "python -c \"import __graft_entry__ as g; g.smoke()\"" bench "" prof control07 100 prof maxcut4000 20 prof blockdiag 40 prof nb 100

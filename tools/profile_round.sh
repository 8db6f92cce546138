#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats + PMC HBM traffic of one bench command.
# Usage: tools/profile_round.sh <tag> [workload] [steps] [nopmc]      outputs under gpurun_out/prof_<tag>/
#   workload: control07 (default) | control07_like | nb | maxcut<n> | blockdiag   (bench.py --workload)
# PMC counters are collected in their own passes (one counter each), never together with other trace domains.
set -u
TAG=${1:-r02}
WL=${2:-control07}
STEPS=${3:-100}
NOPMC=${4:-}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python $REPO/bench.py --workload $WL --steps $STEPS --warmup 5 --no-cpu-baseline --no-other-configs > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
if [ -z "$NOPMC" ]; then
  PS=$(( STEPS / 10 > 2 ? STEPS / 10 : 2 ))
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- python $REPO/bench.py --workload $WL --steps $PS --warmup 2 --no-cpu-baseline --no-other-configs > /dev/null 2> $OUT/fetch.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- python $REPO/bench.py --workload $WL --steps $PS --warmup 2 --no-cpu-baseline --no-other-configs > /dev/null 2> $OUT/write.err
  # matrix-core utilisation (north_star: "rocprof HBM GB/s and MFMA utilisation"): busy cycles of the MFMA pipes and the FP64 MFMA
  # operations issued, beside the cycles the GPU was active -- SQ and GRBM counters, their own pass
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma -o mfma -- python $REPO/bench.py --workload $WL --steps $PS --warmup 2 --no-cpu-baseline --no-other-configs > /dev/null 2> $OUT/mfma.err
fi
cd $REPO
python tools/summarize_prof.py $OUT $TAG

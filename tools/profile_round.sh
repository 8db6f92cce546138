#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats + PMC HBM traffic of the default bench command.
# Usage: tools/profile_round.sh <tag>      outputs under gpurun_out/prof_<tag>/
set -u
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python $REPO/bench.py --steps 100 --warmup 5 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2> $OUT/write.err
cd $REPO
python tools/summarize_prof.py $OUT $TAG

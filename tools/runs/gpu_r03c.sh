#!/bin/bash
# round-3 GPU run C: the whole GPU suite with the inverse behind the factorisation, then the bench line
OUT=gpurun_out/r03c; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/tests.txt 2>&1
tail -5 $OUT/tests.txt
timeout 400 python bench.py --steps 50 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
cut -c1-600 $OUT/bench.json; tail -3 $OUT/bench.err

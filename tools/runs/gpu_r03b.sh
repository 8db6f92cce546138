#!/bin/bash
# round-3 GPU run B: the inverse behind the factorisation (k_sinv_follow)
OUT=gpurun_out/r03b; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "one_launch or solve_widths or deterministic or golden or control07 or resident_plan or native_library" > $OUT/tests.txt 2>&1
tail -5 $OUT/tests.txt
timeout 300 python tools/time_solves.py control07 0 > $OUT/solves_control07.jsonl 2> $OUT/solves_control07.err
cat $OUT/solves_control07.jsonl; tail -3 $OUT/solves_control07.err
timeout 400 python bench.py --steps 50 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
cut -c1-1800 $OUT/bench.json; tail -3 $OUT/bench.err

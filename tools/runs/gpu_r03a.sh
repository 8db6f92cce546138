#!/bin/bash
# round-3 GPU run A: the wide-super-block solves -- parity subset, width sweep on control07 / MAXCUT-4000, bench line
OUT=gpurun_out/r03a; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "solve_widths or sparse_factor or sparse_rhs or golden or maxcut_big_front or sweeps_with_front or resident_dense or resident_plan or native_library" > $OUT/tests.txt 2>&1
tail -3 $OUT/tests.txt
timeout 300 python tools/time_solves.py control07 0 256 512 > $OUT/solves_control07.jsonl 2> $OUT/solves_control07.err
cat $OUT/solves_control07.jsonl
timeout 400 python tools/time_solves.py maxcut4000 256 512 1024 2048 > $OUT/solves_maxcut4000.jsonl 2> $OUT/solves_maxcut4000.err
cat $OUT/solves_maxcut4000.jsonl
timeout 400 python bench.py --steps 50 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
cut -c1-1500 $OUT/bench.json

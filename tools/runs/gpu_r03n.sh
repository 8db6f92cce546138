#!/bin/bash
# row-dot sweep kernels (transposed inverse + transposed copy of L)
OUT=gpurun_out/r03n; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "solve_widths or sparse_factor or sparse_rhs or golden or maxcut_big_front or sweeps_with_front or resident_dense or resident_plan or deterministic" > $OUT/tests.txt 2>&1; echo "rc=$?"
tail -3 $OUT/tests.txt
timeout 300 python tools/time_solves.py control07 0 > $OUT/solves_control07.jsonl 2> $OUT/solves_control07.err; cut -c1-420 $OUT/solves_control07.jsonl
timeout 400 python tools/time_solves.py maxcut4000 512 1024 2048 > $OUT/solves_maxcut4000.jsonl 2> $OUT/solves_maxcut4000.err; cut -c1-620 $OUT/solves_maxcut4000.jsonl

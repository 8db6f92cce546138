#!/bin/bash
OUT=gpurun_out/r03o; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python tools/time_solves.py control07_init 0 > $OUT/a.jsonl 2> $OUT/a.err; cut -c1-500 $OUT/a.jsonl
timeout 300 python tools/time_solves.py control07 0 > $OUT/b.jsonl 2> $OUT/b.err; cut -c1-500 $OUT/b.jsonl
timeout 400 python tools/time_solves.py maxcut4000 1024 2048 > $OUT/c.jsonl 2> $OUT/c.err; cut -c1-600 $OUT/c.jsonl

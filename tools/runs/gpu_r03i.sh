#!/bin/bash
OUT=gpurun_out/r03i; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python tools/phase_chain.py 666 > $OUT/phase666.txt 2>&1; cat $OUT/phase666.txt
timeout 300 python tools/phase_chain.py 896 > $OUT/phase896.txt 2>&1; cat $OUT/phase896.txt
timeout 300 python tools/time_solves.py control07 0 > $OUT/ts.jsonl 2>$OUT/ts.err; cut -c1-400 $OUT/ts.jsonl

#!/bin/bash
OUT=gpurun_out/r03k; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python tools/trace_front.py 666 > $OUT/trace666.txt 2>&1; cat $OUT/trace666.txt

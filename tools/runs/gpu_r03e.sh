#!/bin/bash
OUT=gpurun_out/r03e; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q > $OUT/tests.txt 2>&1
tail -15 $OUT/tests.txt

#!/bin/bash
OUT=gpurun_out/r03d; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python tools/diag_driver.py control07 0:1e4 256:1e4 0:1e2 0:0 > $OUT/diag_control07.txt 2>&1
grep "==" $OUT/diag_control07.txt; head -45 $OUT/diag_control07.txt | tail -42

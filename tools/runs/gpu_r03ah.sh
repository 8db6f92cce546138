#!/bin/bash
cd /root/repo
OUT=gpurun_out/r03ah; mkdir -p $OUT
timeout 300 python tools/time_ada.py blockdiag >> $OUT/ada.jsonl 2>> $OUT/err.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "ada or iteration or golden or hermitian or blockdiag or full_size or above or subtree" > $OUT/tests.txt 2>&1
tail -3 $OUT/tests.txt

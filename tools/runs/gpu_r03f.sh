#!/bin/bash
OUT=gpurun_out/r03f; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_mexshims_gpu.py -m gpu -q > $OUT/t1.txt 2>&1; echo "mexshims rc=$?"; tail -3 $OUT/t1.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "one_launch_front_matches or solve_widths" > $OUT/t2.txt 2>&1; echo "parity subset rc=$?"; tail -3 $OUT/t2.txt
timeout 600 python -m pytest tests/test_driver.py -m gpu -q -k "nb" > $OUT/t3.txt 2>&1; echo "driver subset rc=$?"; tail -3 $OUT/t3.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "dense_col or resident_dense or sparse_rhs" > $OUT/t4.txt 2>&1; echo "dense subset rc=$?"; tail -3 $OUT/t4.txt

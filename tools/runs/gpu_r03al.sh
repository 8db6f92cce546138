#!/bin/bash
cd /root/repo
OUT=gpurun_out/r03al; mkdir -p $OUT
timeout 300 python tools/time_ada.py arch0 >> $OUT/ada.jsonl 2>> $OUT/err.txt
timeout 300 python tools/time_ada.py nb_init >> $OUT/ada.jsonl 2>> $OUT/err.txt
timeout 300 python bench.py --workload lpdense --steps 20 --warmup 3 > $OUT/lpdense.json 2>> $OUT/err.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "ada or iteration or golden or getada or dense or lp" > $OUT/tests.txt 2>&1
tail -3 $OUT/tests.txt

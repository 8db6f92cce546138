#!/bin/bash
cd /root/repo
OUT=gpurun_out/r03as; mkdir -p $OUT
timeout 120 python tests/tools/soak_def.py 40 > $OUT/def_front.txt 2>&1
tail -3 $OUT/def_front.txt
timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "one_launch_front" 2>&1 | tail -3 | tee $OUT/front_tests.txt
timeout 100 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-other-configs > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json | head -c 1500

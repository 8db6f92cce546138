#!/bin/bash
OUT=gpurun_out/r03g; mkdir -p $OUT
export TMPDIR=/tmp
for t in test_shims_symbolic_bit_exact test_shim_incorder test_shims_reproduce_an_iteration_unit test_factor_cache_is_shared_between_mex_binaries_and_validated_by_content test_getada_shim_updates_the_global test_shim_invcholfac test_shims_dense_column_path test_shim_errors_go_through_mexErrMsgTxt test_adendotd_and_adenscale_match_reference; do
  timeout 300 python -X faulthandler -m pytest tests/test_mexshims_gpu.py -m gpu -q -k "$t" > $OUT/$t.txt 2>&1; echo "$t rc=$?"
done
grep -l "Segmentation\|Fatal" $OUT/*.txt | head; grep -h -A25 "Fatal Python error" $OUT/test_shims_reproduce_an_iteration_unit.txt | head -40

#!/bin/bash
# round 4, run h: stage 2 of ADA' (k_psd_stage2_ell) -- columns per workgroup, interleaved z layout, split of the row groups
mkdir -p gpurun_out/r04h
for v in "" s2i s2jb3 s2jb3i s2jb4 s2jb4i s2jb5i s2jb4ig2 s2jb4ig5 s2jb4g5; do
  if [ -z "$v" ]; then lib=""; else lib="libsedumi_hip_$v.so"; fi
  SDM_LIB=$lib python tools/time_ada.py control07 >> gpurun_out/r04h/ada_control07_stage2_variants.jsonl 2>gpurun_out/r04h/err_$v.txt
done
cat gpurun_out/r04h/ada_control07_stage2_variants.jsonl

#!/bin/bash
# same-box A/B of the general blind rotation after the addressing changes (scalar-base loads, thread index opaque per transform):
# lookahead depth x position of the first key request (m<LA><FIRST_AT>)
OUT=gpurun_out/r03_gen6; mkdir -p $OUT
for v in gen_m22 gen_m10; do
REDSEC_HIP_LIB=$PWD/variants/lib_$v.so timeout -k 10 400 python -m pytest tests/test_gpu_general.py -x -q 2>&1 | tail -3 | tee -a $OUT/general_ab_addressing.txt
done
for v in gen_base gen_m10 gen_m20 gen_m22 gen_m30 gen_m32 gen_base gen_m10 gen_m20 gen_m22 gen_m30 gen_m32; do
  echo "== $v" | tee -a $OUT/general_ab_addressing.txt
  REDSEC_HIP_LIB=$PWD/variants/lib_$v.so timeout -k 10 300 python tools/general_rate.py redsec_medium redsec_large 2>$OUT/err_$v.txt | tee -a $OUT/general_ab_addressing.txt
done

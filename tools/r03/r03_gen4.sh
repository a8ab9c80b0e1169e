#!/bin/bash
# same-box A/B of the general blind rotation after the addressing changes (scalar-base loads, thread index opaque per transform):
# lookahead depth x position of the first key request (n<LA><FIRST_AT>)
OUT=gpurun_out/r03_gen4; mkdir -p $OUT
REDSEC_HIP_LIB=$PWD/variants/lib_gen_n22.so timeout -k 10 400 python -m pytest tests/test_gpu_general.py -x -q 2>&1 | tail -3 | tee -a $OUT/general_ab_addressing.txt
for v in gen_base gen_n10 gen_n12 gen_n20 gen_n22 gen_n30 gen_n32 gen_base gen_n10 gen_n12 gen_n20 gen_n22 gen_n30 gen_n32; do
  echo "== $v" | tee -a $OUT/general_ab_addressing.txt
  REDSEC_HIP_LIB=$PWD/variants/lib_$v.so timeout -k 10 300 python tools/general_rate.py redsec_medium redsec_large 2>$OUT/err_$v.txt | tee -a $OUT/general_ab_addressing.txt
done

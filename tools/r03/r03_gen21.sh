#!/bin/bash
# N = 8192 after the row rotation: a = first key request behind the last exchange, b = three positions ahead (all at the tail),
# c = near twiddle levels in LDS, d = digit rotation only
OUT=gpurun_out/r03_gen21; mkdir -p $OUT
for v in gen_cur gen_xa gen_xb gen_xc gen_xd gen_cur gen_xa gen_xb gen_xc gen_xd; do
  echo "== $v" | tee -a $OUT/general_ab_large_after_rotation.txt
  REDSEC_HIP_LIB=$PWD/variants/lib_$v.so timeout -k 10 300 python tools/general_rate.py redsec_large 2>$OUT/err_$v.txt | tee -a $OUT/general_ab_large_after_rotation.txt
done

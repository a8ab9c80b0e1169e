#!/bin/bash
# general kernels: near twiddle levels staged in LDS for N <= 4096 (gen_tw12, default), for no ring (gen_tw9), for N = 8192 too (gen_tw13)
OUT=gpurun_out/r03_gen10; mkdir -p $OUT
for v in gen_tw12 gen_tw9 gen_tw13 gen_tw12 gen_tw9 gen_tw13; do
  echo "== $v" | tee -a $OUT/general_ab_twiddle_staging.txt
  REDSEC_HIP_LIB=$PWD/variants/lib_$v.so timeout -k 10 300 python tools/general_rate.py redsec_medium redsec_large 2>$OUT/err_$v.txt | tee -a $OUT/general_ab_twiddle_staging.txt
done

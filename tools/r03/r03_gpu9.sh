#!/bin/bash
OUT=gpurun_out/r03_gpu9; mkdir -p $OUT
for v in genpipe genpre genpipe genpre; do
  echo "== $v" | tee -a $OUT/general_ab.txt
  REDSEC_HIP_LIB=$PWD/variants/lib_$v.so timeout -k 10 300 python tools/general_rate.py redsec_medium redsec_large 2>/dev/null | tee -a $OUT/general_ab.txt
done
REDSEC_HIP_LIB=$PWD/variants/lib_genpipe.so timeout -k 10 300 python tools/general_rate.py redsec_small default128 2>/dev/null | tee -a $OUT/general_small_sets.txt

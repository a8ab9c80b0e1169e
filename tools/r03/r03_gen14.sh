#!/bin/bash
# general blind rotation: which pass's twiddles to keep at N = 8192 (gen_cur: pass 1; gen_kall2: pass 2; gen_kall3: pass 3 = the last)
OUT=gpurun_out/r03_gen14; mkdir -p $OUT
for v in gen_cur gen_kall2 gen_kall3 gen_cur gen_kall2 gen_kall3; do
  echo "== $v" | tee -a $OUT/general_ab_which_pass_kept.txt
  REDSEC_HIP_LIB=$PWD/variants/lib_$v.so timeout -k 10 300 python tools/general_rate.py redsec_medium redsec_large 2>$OUT/err_$v.txt | tee -a $OUT/general_ab_which_pass_kept.txt
done

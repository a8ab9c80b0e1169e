#!/bin/bash
# general blind rotation: de-phased workgroups (st*) against the current build, and the no-key-loads probe on the current build
OUT=gpurun_out/r03_gen7; mkdir -p $OUT
for v in gen_cur gen_st1 gen_st2 gen_st3 gen_nokey gen_cur gen_st1 gen_st2 gen_st3; do
  echo "== $v" | tee -a $OUT/general_ab_stagger.txt
  REDSEC_HIP_LIB=$PWD/variants/lib_$v.so timeout -k 10 300 python tools/general_rate.py redsec_medium redsec_large 2>$OUT/err_$v.txt | tee -a $OUT/general_ab_stagger.txt
done

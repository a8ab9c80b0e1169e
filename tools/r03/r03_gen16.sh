#!/bin/bash
# general blind rotation: two key positions requested at the tail hook (gen_fg2), and with three positions of lookahead (gen_fg2la3)
OUT=gpurun_out/r03_gen16; mkdir -p $OUT
for v in gen_cur gen_fg2 gen_fg2la3 gen_cur gen_fg2 gen_fg2la3; do
  echo "== $v" | tee -a $OUT/general_ab_two_positions_at_tail.txt
  REDSEC_HIP_LIB=$PWD/variants/lib_$v.so timeout -k 10 300 python tools/general_rate.py redsec_medium redsec_large 2>$OUT/err_$v.txt | tee -a $OUT/general_ab_two_positions_at_tail.txt
done

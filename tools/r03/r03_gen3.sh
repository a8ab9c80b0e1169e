#!/bin/bash
# same-box A/B of the general blind rotation: where the first key position of a row is requested (0 behind the transform,
# 2 in front of its last exchange, 3 behind its last exchange)
OUT=gpurun_out/r03_gen3; mkdir -p $OUT
for v in gen_base gen_fa0 gen_fa2 gen_fa3 gen_base gen_fa0 gen_fa2 gen_fa3; do
  echo "== $v" | tee -a $OUT/general_ab_first_position.txt
  REDSEC_HIP_LIB=$PWD/variants/lib_$v.so timeout -k 10 300 python tools/general_rate.py redsec_medium redsec_large 2>$OUT/err_$v.txt | tee -a $OUT/general_ab_first_position.txt
  tail -2 $OUT/err_$v.txt | cut -c1-300
done

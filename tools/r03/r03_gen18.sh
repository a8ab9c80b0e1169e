#!/bin/bash
# general blind rotation: row-order rotation modes (1 digits + component order, 2 digits only, 3 component order only)
OUT=gpurun_out/r03_gen18; mkdir -p $OUT
for v in gen_cur gen_rot1 gen_rot2 gen_rot3 gen_cur gen_rot1 gen_rot2 gen_rot3; do
  echo "== $v" | tee -a $OUT/general_ab_row_rotation_modes.txt
  REDSEC_HIP_LIB=$PWD/variants/lib_$v.so timeout -k 10 300 python tools/general_rate.py redsec_medium redsec_large 2>$OUT/err_$v.txt | tee -a $OUT/general_ab_row_rotation_modes.txt
done

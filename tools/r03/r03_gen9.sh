#!/bin/bash
# general blind rotation: the two inverse transforms of a column as a pair (gen_inv2) against one after the other (gen_inv1)
OUT=gpurun_out/r03_gen9; mkdir -p $OUT
REDSEC_HIP_LIB=$PWD/variants/lib_gen_inv2.so timeout -k 10 400 python -m pytest tests/test_gpu_general.py -x -q 2>&1 | tail -5 | tee $OUT/tests.txt
for v in gen_inv1 gen_inv2 gen_inv1 gen_inv2; do
  echo "== $v" | tee -a $OUT/general_ab_inverse_pairs.txt
  REDSEC_HIP_LIB=$PWD/variants/lib_$v.so timeout -k 10 300 python tools/general_rate.py redsec_medium redsec_large 2>/dev/null | tee -a $OUT/general_ab_inverse_pairs.txt
done

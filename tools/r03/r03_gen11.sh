#!/bin/bash
# general blind rotation: the key row of a step's first digit requested whole at the start of the step (gen_r0) against the current form
OUT=gpurun_out/r03_gen11; mkdir -p $OUT
REDSEC_HIP_LIB=$PWD/variants/lib_gen_r0.so timeout -k 10 400 python -m pytest tests/test_gpu_general.py -x -q 2>&1 | tail -4 | tee $OUT/tests.txt
for v in gen_cur gen_r0 gen_cur gen_r0; do
  echo "== $v" | tee -a $OUT/general_ab_first_row_ahead.txt
  REDSEC_HIP_LIB=$PWD/variants/lib_$v.so timeout -k 10 300 python tools/general_rate.py redsec_medium redsec_large 2>$OUT/err_$v.txt | tee -a $OUT/general_ab_first_row_ahead.txt
done

#!/bin/bash
# general blind rotation, rings with LDS-staged near levels: the LAST pass's twiddles (read from global memory) kept in registers (gen_kp3)
OUT=gpurun_out/r03_gen13; mkdir -p $OUT
REDSEC_HIP_LIB=$PWD/variants/lib_gen_kp3.so timeout -k 10 400 python -m pytest tests/test_gpu_general.py -x -q 2>&1 | tail -4 | tee $OUT/tests.txt
for v in gen_cur gen_kp3 gen_cur gen_kp3; do
  echo "== $v" | tee -a $OUT/general_ab_last_pass_twiddles_kept.txt
  REDSEC_HIP_LIB=$PWD/variants/lib_$v.so timeout -k 10 300 python tools/general_rate.py redsec_medium redsec_large 2>$OUT/err_$v.txt | tee -a $OUT/general_ab_last_pass_twiddles_kept.txt
done

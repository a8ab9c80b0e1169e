#!/bin/bash
# general blind rotation: pass 1's twiddles kept in registers (gen_k1) against fetched per transform (gen_cur)
OUT=gpurun_out/r03_gen12; mkdir -p $OUT
REDSEC_HIP_LIB=$PWD/variants/lib_gen_k1.so timeout -k 10 400 python -m pytest tests/test_gpu_general.py -x -q 2>&1 | tail -4 | tee $OUT/tests.txt
for v in gen_cur gen_k1 gen_cur gen_k1; do
  echo "== $v" | tee -a $OUT/general_ab_pass1_twiddles_kept.txt
  REDSEC_HIP_LIB=$PWD/variants/lib_$v.so timeout -k 10 300 python tools/general_rate.py redsec_medium redsec_large 2>$OUT/err_$v.txt | tee -a $OUT/general_ab_pass1_twiddles_kept.txt
done
